# the 20-state cascade on 512-thread workgroups (two waves per SIMD within 256 registers) against one wave per SIMD
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s32
PCGYM_HIP_LIB=_ab/qstats_j.so python tools/queue_probe.py me20 2>&1 | grep -v amdgpu | tail -15 | tee gpurun_out/s32/queue_probe_me20_wide.txt
timeout 1500 python -m pytest tests -m gpu -x -q -k "queue or permut or mixed or rodas or extraction or round2" > gpurun_out/s32/pytest_gpu.txt 2>&1; tail -3 gpurun_out/s32/pytest_gpu.txt
for i in 1 2; do
for w in me20; do
  for wide in 0 1; do
  PCG_Q_WIDE=$wide python bench.py --workload $w --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=d['roofline']
print('$w wide=$wide', 'ms/step %.4f kernel %.1f us value %.3e' % (d['ms_per_step'], r['kernel_avg_us'], d['value']), flush=True)"
  done
done
done 2>&1 | tee gpurun_out/s32/bench.txt
