set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s57
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/s57/pytest_gpu.txt 2>&1; tail -4 gpurun_out/s57/pytest_gpu.txt
export PMC_ROUND=r4
bash tools/prof_all.sh > gpurun_out/s57/prof_all.txt 2>&1
cp gpurun_out/pmc.json profiles/r4/pmc.json   # (on the box only: the bench lines below quote the counters of THIS build)
python bench.py > gpurun_out/s57/bench_default.json 2> gpurun_out/s57/bench_default.err
for i in 1 2 3; do
  python bench.py --no-cpu-baseline > gpurun_out/s57/bench_default_run$i.json 2>/dev/null
  python bench.py --no-cpu-baseline --steps 20 --warmup 5 > gpurun_out/s57/bench_driver_shape_run$i.json 2>/dev/null
done
for w in cstr_safe four_tank me10 me10_ros4 me20 cryst cryst_cv8 mixed; do
  python bench.py --workload $w --no-cpu-baseline > gpurun_out/s57/bench_$w.json 2> gpurun_out/s57/bench_$w.err
done
python bench.py --graph --no-cpu-baseline > gpurun_out/s57/bench_graph.json 2>/dev/null
python bench.py --workload four_tank --integrator rk4 --no-cpu-baseline > gpurun_out/s57/bench_four_tank_rk4.json 2>/dev/null
python - <<'P' | tee gpurun_out/s57/bench_all.txt
import json,glob,os
for f in sorted(glob.glob('gpurun_out/s57/bench_*.json')):
    try:
        d=json.loads([l for l in open(f) if l.startswith('{')][-1]); r=d['roofline']
        tr=r.get('traffic_over_algorithmic'); ip=r.get('valu_issue_time_frac_at_2p4ns')
        print(f"{os.path.basename(f)[6:-5]:24s} value {d['value']:.4e} ms/step {d['ms_per_step']*1e3:9.3f} us kernel {r['kernel_avg_us']:9.2f} us frac {r['frac']:.3f} ({r['bound']}) traffic/alg {tr if tr is None else round(tr,3)} issue@2.4ns {ip if ip is None else round(ip,3)} steps {d['steps']} sane {d['config']['sane']}")
    except Exception as e: print(f,'FAILED',e)
P
python tools/default_cstr_probe.py 2>&1 | grep -v amdgpu > gpurun_out/s57/default_cstr_probe.txt; head -2 gpurun_out/s57/default_cstr_probe.txt
PCGYM_HIP_LIB=_ab/qstats_j.so python tools/queue_probe.py me20 2>&1 | grep -v amdgpu | tail -15 > gpurun_out/s57/queue_probe_me20.txt
PCGYM_HIP_LIB=_ab/qstats_i.so python tools/queue_probe.py me10 2>&1 | grep -v amdgpu | tail -15 > gpurun_out/s57/queue_probe_me10.txt
PCGYM_HIP_LIB=_ab/qstats_i.so python tools/queue_probe.py me10_ros4 2>&1 | grep -v amdgpu | tail -15 > gpurun_out/s57/queue_probe_me10_ros4.txt
PCGYM_HIP_LIB=_ab/qstats_a.so python tools/queue_probe.py cstr_safe 2>&1 | grep -v amdgpu | tail -15 > gpurun_out/s57/queue_probe_cstr_safe_fixup.txt
