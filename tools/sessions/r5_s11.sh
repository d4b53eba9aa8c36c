set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5s11; mkdir -p $O
timeout 2700 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; echo "pytest rc $?" >> $O/pytest_gpu.txt
tail -4 $O/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
for w in cstr cstr_safe four_tank me10 me10_ros4 me20 cryst cryst_cv8 mixed; do
  timeout 600 python bench.py --workload $w $( [ $w = cstr ] || echo --no-cpu-baseline ) > $O/bench_$w.json 2> $O/bench_$w.err
  python - $w $O/bench_$w.json <<'P'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[2]) if l.startswith('{')][-1]); r=d['roofline']
    print(f"{sys.argv[1]:10s} value {d['value']:.4e} ms/step {d['ms_per_step']*1e3:9.3f} us kernel {r['kernel_avg_us']:9.2f} us frac {r['frac']:.3f} ({r['bound']}) traffic/alg {r.get('traffic_over_algorithmic')} issue-by-class {r.get('valu_issue_time_frac_by_class')} copy {r.get('copy_ceiling_GBps')} steps {d['steps']} sane {d['config']['sane']}")
except Exception as e: print(sys.argv[1],'FAILED',e)
P
done
for i in 1 2 3 4 5; do python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 >> $O/driver_shape_runs.jsonl; done
python - <<'P'
import json
for l in open('gpurun_out/r5s11/driver_shape_runs.jsonl'):
    d=json.loads(l); print('driver shape: %.4e env-steps/s  %.3f us per step  kernel %.2f us' % (d['value'], d['ms_per_step']*1e3, d['roofline']['kernel_avg_us']))
P
