set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5s8; mkdir -p $O
timeout 2400 python -m pytest tests/test_gpu_rodas4.py tests/test_gpu_seulex.py tests/test_gpu_parity.py tests/test_gpu_round2.py -m gpu -x -q > $O/pytest_gpu.txt 2>&1; echo "pytest rc $?" >> $O/pytest_gpu.txt
tail -5 $O/pytest_gpu.txt
for w in me10 me10_ros4 mixed me10 me10_ros4 mixed; do
  timeout 600 python bench.py --workload $w --no-cpu-baseline > $O/bench_$w.json 2> $O/bench_$w.err
  python - $w $O/bench_$w.json <<'P'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[2]) if l.startswith('{')][-1]); r=d['roofline']
    print(f"{sys.argv[1]:10s} value {d['value']:.4e} ms/step {d['ms_per_step']*1e3:9.3f} us kernel {r['kernel_avg_us']:9.2f} us frac {r['frac']:.3f} ({r['bound']}) sane {d['config']['sane']}")
except Exception as e: print(sys.argv[1],'FAILED',e)
P
done
