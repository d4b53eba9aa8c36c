#!/bin/bash
# One bench line per workload (bench.py defaults), on the GPU box; run AFTER tools/prof_all.sh in the same call so that the
# lines quote the counter passes of this very build (gpurun_out/pmc.json is copied to profiles/r6/ first).
#   output: gpurun_out/bench_workloads.jsonl, gpurun_out/bench_all.txt
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd $ROOT
[ -f gpurun_out/pmc.json ] && mkdir -p profiles/r6 && cp gpurun_out/pmc.json profiles/r6/pmc.json
: > gpurun_out/bench_workloads.jsonl
for w in cstr cstr_safe cstr_rollout cstr_safe_rollout cstr_unc four_tank me10 me10_ros4 me10_ros5 me20 cryst cryst_cv8 mixed; do
  extra="--no-cpu-baseline"; [ $w = cstr ] && extra=""
  python bench.py --workload $w $extra 2>/dev/null | grep '^{' | tail -1 >> gpurun_out/bench_workloads.jsonl
done
python - <<'PY' > gpurun_out/bench_all.txt
import json
print("# bench.py --workload <w> at its defaults, one MI355X (tools/bench_all.sh); us = wall time per env step of the whole batch")
for l in open("gpurun_out/bench_workloads.jsonl"):
    d = json.loads(l); r = d["roofline"]
    print(f"{d['config']['workload'][:62]:62s} value {d['value']:.4e} us/step {d['ms_per_step']*1e3:9.2f} kernel {r['kernel_avg_us']:9.2f} "
          f"frac {r['frac']:.3f} ({r['bound']}) traffic/alg {r.get('traffic_over_algorithmic')} issue-by-class {r.get('valu_issue_time_frac_by_class')} "
          f"copy {r.get('copy_ceiling_GBps')} steps {d['steps']} sane {d['config']['sane']}")
PY
cat gpurun_out/bench_all.txt
