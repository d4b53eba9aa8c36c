set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5s32; mkdir -p $O
run() { local label=$1; shift
  ( "$@" > $O/b.json 2> $O/b.err ) ; python - "$label" $O/b.json <<'P'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[2]) if l.startswith('{')][-1]); r=d['roofline']
    print(f"{sys.argv[1]:34s} value {d['value']:.4e} ms/step {d['ms_per_step']*1e3:9.3f} us kernel {r['kernel_avg_us']:9.2f} us")
except Exception as e: print(sys.argv[1],'FAILED',e)
P
}
for rep in 1 2; do
for w in 0 2 3.56 6 10; do
run "mixed q_w $w"      env PCG_Q_W=$w timeout 600 python bench.py --workload mixed --no-cpu-baseline
run "me10_ros5 q_w $w"  env PCG_Q_W=$w timeout 600 python bench.py --workload me10_ros5 --no-cpu-baseline
done
for r in 2 4 16; do
run "mixed refill $r"   env PCG_Q_REFILL=$r timeout 600 python bench.py --workload mixed --no-cpu-baseline
done
done
