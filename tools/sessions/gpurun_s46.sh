set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s46
timeout 1800 python -m pytest tests -m gpu -x -q > gpurun_out/s46/pytest_gpu.txt 2>&1; tail -3 gpurun_out/s46/pytest_gpu.txt
for i in 1 2; do
for nf in 0 1; do
  if [ $nf = 1 ]; then export PCG_NO_FIXUP=1; else unset PCG_NO_FIXUP; fi
  for args in "--workload mixed" "--workload cstr_safe --graph" "--workload cstr_safe --integrator rk4g"; do
  python bench.py $args --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$args NO_FIXUP=$nf us/step %.1f value %.3e sane %s' % (d['ms_per_step']*1e3, d['value'], d['config']['sane']), flush=True)"
  done
done
done 2>&1 | tee gpurun_out/s46/two_launch_more.txt
