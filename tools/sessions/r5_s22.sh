# the round's final evidence on the final build: profile set (rocprofv3 stats + PMC passes, all ten workloads), the GPU suite, smoke, every bench line
set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5s22; mkdir -p $O
python -c "from pcgym_amd import _lib; print('build', _lib.load().pcg_build_id().decode())" 2>/dev/null | tail -1
export PMC_ROUND=r5
bash tools/prof_all.sh 2>&1 | tail -70 > $O/prof_all.txt
cp gpurun_out/pmc.json profiles/r5/pmc.json   # (on the box: the bench lines below quote the counters of THIS build)
timeout 2700 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; echo "pytest rc $?" >> $O/pytest_gpu.txt
tail -4 $O/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
for w in cstr cstr_safe four_tank me10 me10_ros4 me10_ros5 me20 cryst cryst_cv8 mixed; do
  timeout 900 python bench.py --workload $w $( [ $w = cstr ] || echo --no-cpu-baseline ) > $O/bench_$w.json 2> $O/bench_$w.err
  python - $w $O/bench_$w.json <<'P'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[2]) if l.startswith('{')][-1]); r=d['roofline']
    print(f"{sys.argv[1]:10s} value {d['value']:.4e} ms/step {d['ms_per_step']*1e3:9.3f} us kernel {r['kernel_avg_us']:9.2f} us frac {r['frac']:.3f} ({r['bound']}) traffic/alg {r.get('traffic_over_algorithmic')} issue-by-class {r.get('valu_issue_time_frac_by_class')} copy {r.get('copy_ceiling_GBps')} steps {d['steps']} sane {d['config']['sane']}")
except Exception as e: print(sys.argv[1],'FAILED',e)
P
done | tee $O/bench_all.txt
cp $O/bench_cstr.json $O/bench_default.json
for i in 1 2 3; do python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_driver_shape_run$i.json; done
python - <<'P'
import json
for i in (1,2,3):
    d=json.loads(open(f'gpurun_out/r5s22/bench_driver_shape_run{i}.json').read()); print('driver shape: %.4e env-steps/s  %.3f us per step  kernel %.2f us' % (d['value'], d['ms_per_step']*1e3, d['roofline']['kernel_avg_us']))
P
timeout 600 python tools/queue_soak.py > $O/queue_soak.txt 2>&1; tail -2 $O/queue_soak.txt
