set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s18
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/s18/pytest_gpu.txt 2>&1; tail -4 gpurun_out/s18/pytest_gpu.txt
export PMC_ROUND=r4
bash tools/prof_all.sh > gpurun_out/s18/prof_all.txt 2>&1
python bench.py > gpurun_out/s18/bench_default.json 2> gpurun_out/s18/bench_default.err
for i in 1 2 3; do
  python bench.py --no-cpu-baseline > gpurun_out/s18/bench_default_run$i.json 2>/dev/null
  python bench.py --no-cpu-baseline --steps 20 --warmup 5 > gpurun_out/s18/bench_driver_shape_run$i.json 2>/dev/null
done
for w in cstr_safe four_tank me10 me10_ros4 me20 cryst cryst_cv8 mixed; do
  python bench.py --workload $w --no-cpu-baseline > gpurun_out/s18/bench_$w.json 2> gpurun_out/s18/bench_$w.err
done
python bench.py --graph --no-cpu-baseline > gpurun_out/s18/bench_graph.json 2>/dev/null
python bench.py --workload four_tank --integrator rk4 --no-cpu-baseline > gpurun_out/s18/bench_four_tank_rk4.json 2>/dev/null
python - <<'P' | tee gpurun_out/s18/bench_all.txt
import json,glob,os
for f in sorted(glob.glob('gpurun_out/s18/bench_*.json')):
    try:
        d=json.loads([l for l in open(f) if l.startswith('{')][-1]); r=d['roofline']
        print(f"{os.path.basename(f)[6:-5]:24s} value {d['value']:.4e} ms/step {d['ms_per_step']*1e3:9.3f} us kernel {r['kernel_avg_us']:9.2f} us frac {r['frac']:.3f} ({r['bound']}) traffic {r.get('traffic')} issue2p4 {r.get('valu_issue_time_frac_at_2p4ns')} steps {d['steps']} sane {d['config']['sane']}")
    except Exception as e: print(f,'FAILED',e)
P
python - <<'P'
import json
d=json.load(open('gpurun_out/pmc.json'))
for k,v in d.items():
    print(k, v.get('build_id'), v.get('traffic_bytes_per_launch'), v.get('valu_issue_frac'), v.get('valu_issue_time_frac_at_2p4ns'), v.get('rocprof_avg_us'))
P
