#!/bin/bash
# PMC A/B of bench.py variants (GPU box): usage tools/pmc_ab.sh <outdir> <workload> -- "ENV=... ENV=..." ["ENV=..." ...]
# one rocprofv3 --pmc pass (SQ counters) per variant; prints mean counters per dispatch of the step kernels
OUT=$1; WL=$2; shift 3
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for cfg in "$@"; do
  i=$((i+1))
  env $cfg rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_LDS --kernel-trace --output-format csv -d $OUT/v$i -o p -- python $ROOT/bench.py --workload $WL --steps 12 --warmup 3 --no-cpu-baseline > $OUT/v$i.json 2> $OUT/v$i.log
  echo "== $cfg"
  python - $OUT/v$i <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        agg[row["Kernel_Name"]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, cs in agg.items():
    if "step_kernel" not in k: continue
    print("  ", k[:70], "n=%d" % len(next(iter(cs.values()))), " ".join(f"{c}={sum(v)/len(v):.4g}" for c, v in sorted(cs.items())))
PY
done
