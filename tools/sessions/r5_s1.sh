set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5s1; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_seulex.py tests/test_gpu_rodas4.py -m gpu -x -q > $O/pytest_seulex.txt 2>&1; echo "pytest rc $?" >> $O/pytest_seulex.txt
tail -15 $O/pytest_seulex.txt
line() { python - "$1" "$2" <<'P'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[2]) if l.startswith('{')][-1]); r=d['roofline']
    print(f"{sys.argv[1]:22s} value {d['value']:.4e} ms/step {d['ms_per_step']*1e3:9.3f} us kernel {r.get('kernel_avg_us',0):9.2f} us frac {r['frac']:.3f} sane {d['config'].get('sane')}")
except Exception as e: print(sys.argv[1],'FAILED',e)
P
}
for thr in 0 40 44 48 52 56 62; do
  timeout 300 python bench.py --workload me10_ros4 --no-cpu-baseline --coop-thr $thr > $O/me10_ros4_thr$thr.json 2> $O/me10_ros4_thr$thr.err
  line "me10_ros4 thr $thr" $O/me10_ros4_thr$thr.json
done
for thr in 0 48; do
  timeout 300 python bench.py --workload mixed --no-cpu-baseline --coop-thr $thr > $O/mixed_thr$thr.json 2> $O/mixed_thr$thr.err
  line "mixed thr $thr" $O/mixed_thr$thr.json
done
