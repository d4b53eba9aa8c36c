set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5s17; mkdir -p $O
run() { local label=$1; shift
  ( "$@" > $O/b.json 2> $O/b.err ) ; python - "$label" $O/b.json <<'P'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[2]) if l.startswith('{')][-1]); r=d['roofline']
    print(f"{sys.argv[1]:44s} value {d['value']:.4e} ms/step {d['ms_per_step']*1e3:9.3f} us kernel {r['kernel_avg_us']:9.2f} us sane {d['config'].get('sane')}")
except Exception as e: print(sys.argv[1],'FAILED',e, open(sys.argv[2].replace('.json','.err')).read()[-600:])
P
}
for rep in 1 2; do
run "mixed w1cap 1500"                   env PCG_Q_W1CAP=1500 timeout 600 python bench.py --workload mixed --no-cpu-baseline
run "mixed w1cap 1500 prio 2048"         env PCG_Q_W1CAP=1500 PCG_Q_PRIO=2048 timeout 600 python bench.py --workload mixed --no-cpu-baseline
run "mixed w1cap 1500 prio 0"            env PCG_Q_W1CAP=1500 PCG_Q_PRIO=0 timeout 600 python bench.py --workload mixed --no-cpu-baseline
run "mixed w1cap 1500 prio 512"          env PCG_Q_W1CAP=1500 PCG_Q_PRIO=512 timeout 600 python bench.py --workload mixed --no-cpu-baseline
run "me10_ros5 B=349524 (ME segment alone, no Gauss)"  timeout 600 python bench.py --workload me10_ros5 --batch 349524 --no-cpu-baseline
run "me10_ros5 B=349524 w1cap 1500"      env PCG_Q_W1CAP=1500 timeout 600 python bench.py --workload me10_ros5 --batch 349524 --no-cpu-baseline
done
