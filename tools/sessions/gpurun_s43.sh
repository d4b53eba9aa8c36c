set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s43
for i in 1 2 3; do
for w in me10 mixed; do
  for qw in 16 20 24 28; do
  PCG_Q_W=$qw python bench.py --workload $w --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=d['roofline']
print('$w q_w=$qw', 'us/step %.1f' % (d['ms_per_step']*1e3), flush=True)"
  done
done
done 2>&1 | tee gpurun_out/s43/q_w_fine.txt
