# resident workgroups per CU of the lean pipelined kernel on the compute-heavier fixed-step plans
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s58
for i in 1 2; do
for w in four_tank cryst_cv8 cryst; do
  for b in 0 3 4 5 6 8; do
  PCG_BPC=$b timeout 300 python bench.py --workload $w --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$w PCG_BPC=$b us/step %.2f' % (d['ms_per_step']*1e3), flush=True)"
  done
done
done 2>&1 | tee gpurun_out/s58/bpc_sweep.txt
