# what the work-queue kernel's waves do (me20, me10): per-wave phase stamps and loop counts
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s27
PCGYM_HIP_LIB=_ab/qstats_j.so python tools/queue_probe.py me20 2>&1 | grep -v amdgpu | tee gpurun_out/s27/queue_probe_me20.txt
PCGYM_HIP_LIB=_ab/qstats_i.so python tools/queue_probe.py me10 2>&1 | grep -v amdgpu | tee gpurun_out/s27/queue_probe_me10.txt
for w in me20 me10; do
  python bench.py --workload $w --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=d['roofline']
print('$w', 'ms/step %.4f kernel %.1f us value %.3e' % (d['ms_per_step'], r['kernel_avg_us'], d['value']), d['config'].get('mean_attempts'), flush=True)"
done 2>&1 | tee gpurun_out/s27/bench.txt
