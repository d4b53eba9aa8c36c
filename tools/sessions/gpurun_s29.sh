# phases 1 and 3 of the work-queue kernel with the next slot requested ahead
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s29
PCGYM_HIP_LIB=_ab/qstats_j.so python tools/queue_probe.py me20 2>&1 | grep -v amdgpu | tail -15 | tee gpurun_out/s29/queue_probe_me20.txt
timeout 1500 python -m pytest tests -m gpu -x -q -k "queue or permut or mixed or rodas or extraction or round2" > gpurun_out/s29/pytest_gpu.txt 2>&1; tail -3 gpurun_out/s29/pytest_gpu.txt
for w in me20 me10 me10_ros4 mixed; do
  for i in 1 2; do
  python bench.py --workload $w --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=d['roofline']
print('$w', 'ms/step %.4f kernel %.1f us value %.3e' % (d['ms_per_step'], r['kernel_avg_us'], d['value']), flush=True)"
  done
done 2>&1 | tee gpurun_out/s29/bench.txt
