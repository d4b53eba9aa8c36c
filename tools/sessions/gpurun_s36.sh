set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s36
run() { local name=$1; shift
  env "$@" python bench.py --workload me20 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=d['roofline']
print('$name', 'ms/step %.4f kernel %.1f us' % (d['ms_per_step'], r['kernel_avg_us']), flush=True)"
}
for i in 1 2 3 4; do
  run T1024_hbm A=0
  run T512_xlds PCG_Q_TILE=512
  run T768_hbm PCG_Q_TILE=768
done 2>&1 | tee gpurun_out/s36/me20_tile.txt
