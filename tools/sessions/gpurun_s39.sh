# lean pipelined kernel against the classic kernel on the compute-heavy fixed-step plans
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s39
for i in 1 2 3; do
for w in cryst_cv8 cryst four_tank; do
  for v in 0 1; do
  PCG_VARIANT=$v python bench.py --workload $w --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=d['roofline']
print('$w variant=$v', 'us/step %.2f kernel %.2f us' % (d['ms_per_step']*1e3, r['kernel_avg_us']), flush=True)"
  done
done
done 2>&1 | tee gpurun_out/s39/variant_ab.txt
