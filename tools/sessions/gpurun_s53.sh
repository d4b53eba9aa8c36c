set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s53
timeout 600 python -m pytest tests/test_gpu_erk.py tests/test_gpu_rodas4.py -m gpu -x -q > gpurun_out/s53/pytest_guarded.txt 2>&1; tail -3 gpurun_out/s53/pytest_guarded.txt
for i in 1 2; do
for w in me20 me10 cstr_safe mixed; do
  timeout 300 python bench.py --workload $w --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$w us/step %.1f sane %s' % (d['ms_per_step']*1e3, d['config']['sane']), flush=True)"
done
done 2>&1 | tee gpurun_out/s53/bench.txt
timeout 300 python tools/default_cstr_probe.py 2>&1 | grep -v amdgpu | head -2 | tee -a gpurun_out/s53/bench.txt
