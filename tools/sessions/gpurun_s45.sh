set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s45
timeout 1200 python -m pytest tests/test_gpu_erk.py tests/test_gpu_rodas4.py -m gpu -x -q > gpurun_out/s45/pytest_guarded.txt 2>&1; tail -15 gpurun_out/s45/pytest_guarded.txt
for i in 1 2; do
for nf in 0 1; do
  if [ $nf = 1 ]; then export PCG_NO_FIXUP=1; else unset PCG_NO_FIXUP; fi
  python bench.py --workload cstr_safe --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('cstr_safe NO_FIXUP=$nf us/step %.1f value %.3e sane %s' % (d['ms_per_step']*1e3, d['value'], d['config']['sane']), flush=True)"
  python tools/default_cstr_probe.py 2>&1 | grep -v amdgpu | head -2 | sed "s/^/NO_FIXUP=$nf /"
done
done 2>&1 | tee gpurun_out/s45/two_launch_ab.txt
