#!/bin/bash
# Refresh the bench lines and rocprofv3 summaries under gpurun_out/ (copied into profiles/rNN afterwards).  Run ON the GPU box.
set -u
python -c "import torch"
OUT=gpurun_out/evidence; mkdir -p $OUT
for i in 1 2 3; do python bench.py 2>/dev/null | tail -1 >> $OUT/bench_runs.jsonl; done
for w in four_tank me10 me20 cryst mixed; do python bench.py --workload $w 2>/dev/null | tail -1 >> $OUT/bench_workloads.jsonl; done
python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > $OUT/bench_driver_shape.json
for w in cstr four_tank cryst; do PROF_PMC_STEPS=590 PROF_PMC_WARMUP=59 bash tools/prof.sh $w --workload $w > /dev/null 2>&1; done
python - <<'PY'
import json
rows=[json.loads(l) for l in open("gpurun_out/evidence/bench_runs.jsonl")]
rows.sort(key=lambda d:d["value"])
json.dump(rows[1], open("gpurun_out/evidence/bench_default.json","w"))
for d in rows: print("cstr", "%.3e"%d["value"], d["ms_per_step"], d["roofline"]["frac"])
for l in open("gpurun_out/evidence/bench_workloads.jsonl"):
    d=json.loads(l); print(d["config"]["workload"], "%.3e"%d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["bound"])
PY
