# driver-shaped headline run (--steps 20 --warmup 5): what the host's way of waiting costs.  Interleaved, 6 rounds.
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s30
one() { # name env...
  local name=$1; shift
  env "$@" python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=d['roofline']
print('$name', 'value %.3e us/step %.2f kernel %.2f' % (d['value'], d['ms_per_step']*1e3, r['kernel_avg_us']), flush=True)"
}
for i in 1 2 3 4 5 6; do
  one base A=0
  one active_wait_1ms ROC_ACTIVE_WAIT_TIMEOUT=1000
  one no_interrupt HSA_ENABLE_INTERRUPT=0
  one both ROC_ACTIVE_WAIT_TIMEOUT=1000 HSA_ENABLE_INTERRUPT=0
done 2>&1 | tee gpurun_out/s30/wait_modes.txt
