# after the default largest end-point exponent of rodas5 plans went 12 -> 16 (Python-side default; same library build): the fifth-order pair's tests, the two workloads whose plan it is
set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5s23; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_rodas5.py tests/test_gpu_round2.py tests/test_gpu_seulex.py -m gpu -x -q > $O/pytest.txt 2>&1; echo "pytest rc $?" >> $O/pytest.txt; tail -3 $O/pytest.txt
export PMC_ROUND=r5
bash tools/prof_all.sh me10_ros5 mixed 2>&1 | tail -12 > $O/prof.txt
python - <<'P'
import json
old=json.load(open('profiles/r5/pmc.json')); new=json.load(open('gpurun_out/pmc.json'))
for k in ('me10_ros5','mixed'):
    assert new[k].get('build_id', new[k].get('segments',{}).get('multistage_extraction',{}).get('build_id'))  # (taken on this build)
    old[k]=new[k]
json.dump(old, open('profiles/r5/pmc.json','w'), indent=1)
json.dump(old, open('gpurun_out/r5s23/pmc_merged.json','w'), indent=1)
print({s:(v.get('traffic_bytes_per_launch'), v.get('rocprof_avg_us')) for s,v in old['mixed']['segments'].items()})
P
for rep in 1 2 3; do for w in me10_ros5 mixed; do
  timeout 900 python bench.py --workload $w --no-cpu-baseline > $O/bench_${w}_$rep.json 2> $O/bench_$w.err
  python - $w $O/bench_${w}_$rep.json <<'P'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[2]) if l.startswith('{')][-1]); r=d['roofline']
    print(f"{sys.argv[1]:10s} value {d['value']:.4e} ms/step {d['ms_per_step']*1e3:9.3f} us kernel {r['kernel_avg_us']:9.2f} us frac {r['frac']:.3f} ({r['bound']}) traffic/alg {r.get('traffic_over_algorithmic')} issue-by-class {r.get('valu_issue_time_frac_by_class')} steps {d['steps']} sane {d['config']['sane']}")
except Exception as e: print(sys.argv[1],'FAILED',e)
P
done; done | tee $O/bench_lines.txt
