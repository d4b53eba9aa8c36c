#!/bin/bash
# The second pass of the barrier-free rollout (pcg_rollout_flat.hpp) under its two launch parameters, on the GPU box:
# persistent waves per SIMD x cadence of the step boundaries; us per 2^20-env step of bench.py --workload cstr_safe_rollout.
for w in 2 3 4; do for m in 1 2 3 4 6; do
  r=$(PCG_FLAT_WPS=$w PCG_FLAT_EVERY=$m timeout 300 python bench.py --workload cstr_safe_rollout --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('%.1f' % (d['ms_per_step']*1e3))")
  echo "waves per SIMD $w, boundaries every $m: $r us per step"
done; done
echo "single-kernel rollout (PCG_NO_FLAT=1):"; PCG_NO_FLAT=1 timeout 300 python bench.py --workload cstr_safe_rollout --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('%.1f us per step' % (d['ms_per_step']*1e3))"
echo "stepping (--workload cstr_safe):"; timeout 300 python bench.py --workload cstr_safe --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('%.1f us per step' % (d['ms_per_step']*1e3))"
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/flatprof -o t -- python $GRAFT_REPO_ROOT/bench.py --workload cstr_safe_rollout --no-cpu-baseline > /dev/null 2>&1; python - <<'PY'
import csv, glob
for f in glob.glob('/tmp/flatprof/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        print(r['Name'][:90], 'calls', r['Calls'], 'avg_us', round(float(r['AverageNs'])/1e3,1), 'pct', r['Percentage'])
PY
