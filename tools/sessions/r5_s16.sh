set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5s16; mkdir -p $O
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
run "mixed (default: rodas5)"            timeout 600 python bench.py --workload mixed --no-cpu-baseline
run "mixed w1cap 1500"                   env PCG_Q_W1CAP=1500 timeout 600 python bench.py --workload mixed --no-cpu-baseline
run "mixed prio 0"                       env PCG_Q_PRIO=0 timeout 600 python bench.py --workload mixed --no-cpu-baseline
run "mixed prio 256"                     env PCG_Q_PRIO=256 timeout 600 python bench.py --workload mixed --no-cpu-baseline
run "mixed nolean"                       env PCG_Q_NOLEAN=1 timeout 600 python bench.py --workload mixed --no-cpu-baseline
run "me10_ros5"                          timeout 600 python bench.py --workload me10_ros5 --no-cpu-baseline
run "me10_ros5 w1 off"                   env PCG_Q_W1=0 timeout 600 python bench.py --workload me10_ros5 --no-cpu-baseline
done
timeout 3000 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; echo "pytest rc $?" >> $O/pytest_gpu.txt
tail -8 $O/pytest_gpu.txt
