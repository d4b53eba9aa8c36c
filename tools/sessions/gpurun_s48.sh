set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s48
timeout 600 python -m pytest tests/test_gpu_erk.py tests/test_gpu_rodas4.py -m gpu -x -q > gpurun_out/s48/pytest_guarded.txt 2>&1; tail -8 gpurun_out/s48/pytest_guarded.txt
timeout 600 python -m pytest tests/test_gpu_round2.py -m gpu -x -q -k "queue or launch_shapes" > gpurun_out/s48/pytest_queue.txt 2>&1; tail -3 gpurun_out/s48/pytest_queue.txt
for i in 1 2; do
  timeout 300 python bench.py --workload cstr_safe --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('cstr_safe us/step %.1f value %.3e sane %s' % (d['ms_per_step']*1e3, d['value'], d['config']['sane']), flush=True)"
  timeout 300 python tools/default_cstr_probe.py 2>&1 | grep -v amdgpu | head -2
done 2>&1 | tee gpurun_out/s48/two_launch_compact.txt
PCGYM_HIP_LIB=_ab/qstats_a.so timeout 300 python tools/queue_probe.py cstr_safe 2>&1 | grep -v amdgpu | tail -16 | tee gpurun_out/s48/queue_probe_cstr_safe_fixup_compact.txt
