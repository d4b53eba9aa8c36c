# me20: the tile's state in LDS (T = 512, two sub-tiles per workgroup) against the scattered 8-byte accesses of the
# full tile; the same for the 24-state heat exchanger if it takes the queue.  A/B interleaved, three rounds.
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s23
run() { # name, env...
  local name=$1; shift
  env "$@" python bench.py --workload me20 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=d['roofline']
print('$name', 'ms/step %.4f kernel %.1f us' % (d['ms_per_step'], r['kernel_avg_us']), d['config'].get('accuracy'), flush=True)"
}
for i in 1 2 3; do
  run T1024_hbm A=0
  run T512_xlds PCG_Q_TILE=512
  run T512_hbm PCG_Q_TILE=512 PCG_Q_NOXLDS=1
  run T512_xlds_refill8 PCG_Q_TILE=512 PCG_Q_REFILL=8
done 2>&1 | tee gpurun_out/s23/me20_tile.txt
