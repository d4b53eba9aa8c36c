set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s51
timeout 1800 python -m pytest tests -m gpu -x -q > gpurun_out/s51/pytest_gpu.txt 2>&1; tail -3 gpurun_out/s51/pytest_gpu.txt
timeout 600 python tools/queue_soak.py 40 > gpurun_out/s51/queue_soak.txt 2>&1; tail -2 gpurun_out/s51/queue_soak.txt
for args in "--workload mixed" "--workload cstr_safe --graph" "--workload cstr_safe --integrator rk4g" "--workload me10" "--workload me20"; do
  timeout 300 python bench.py $args --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$args us/step %.1f value %.3e sane %s' % (d['ms_per_step']*1e3, d['value'], d['config']['sane']), flush=True)"
done 2>&1 | tee gpurun_out/s51/bench.txt
