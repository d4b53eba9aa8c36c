set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5s15; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_rodas5.py -m gpu -x -q > $O/pytest_r5.txt 2>&1; echo "pytest rc $?" >> $O/pytest_r5.txt
tail -8 $O/pytest_r5.txt
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
run "me10 --integrator rodas4 (default coop)"  timeout 600 python bench.py --workload me10 --integrator rodas4 --no-cpu-baseline
run "me10 --integrator rodas5 (coop off)"      timeout 600 python bench.py --workload me10 --integrator rodas5 --no-cpu-baseline
run "me10 --integrator rodas5 coop 60"         timeout 600 python bench.py --workload me10 --integrator rodas5 --coop-thr 60 --no-cpu-baseline

run "mixed rodas4"                             timeout 600 python bench.py --workload mixed --integrator rodas4 --no-cpu-baseline
run "mixed rodas5"                             timeout 600 python bench.py --workload mixed --integrator rodas5 --no-cpu-baseline
run "mixed rodas5 coop 60"                     timeout 600 python bench.py --workload mixed --integrator rodas5 --coop-thr 60 --no-cpu-baseline
done
