#!/bin/bash
# Same-box A/B of the headline workload (round 4).  Run ON the GPU box:
#   tools/headline_ab.sh <rounds> [tag=/path/to/libpcgym_hip.so ...]
# Every tag is the current tree's bench.py with that build of the library (PCGYM_HIP_LIB); "cur" = the tree's own library.
# AB_TREES=1 adds the round-1 / round-2 trees staged under _ab/r1, _ab/r2 (their own bench.py, package and library).
# Each round runs every variant once in the driver's shape (--steps 20 --warmup 5) and once in the default shape; a
# rocprofv3 kernel trace of the default shape follows.  Output: gpurun_out/ab/.
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/ab
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
R=${1:-3}; shift
VARS=("cur=" "$@")
line() { python3 - "$@" <<'P'
import json,sys
tag,path=sys.argv[1],sys.argv[2]
try:
    d=json.loads([l for l in open(path) if l.startswith('{')][-1])
    r=d.get('roofline',{})
    print(f"{tag:34s} value {d['value']:.4e}  ms/step {d['ms_per_step']*1e3:7.3f} us  kernel(events) {r.get('kernel_avg_us',float('nan')):6.2f} us  frac {r.get('frac',float('nan')):.3f}  steps {d['steps']} warmup {d['warmup']}")
except Exception as e:
    print(tag,'FAILED',e)
P
}
run() { # tag tree lib -- args
  local tag=$1 tree=$2 lib=$3; shift 4
  ( cd $tree && env ${lib:+PCGYM_HIP_LIB=$lib} timeout 300 python3 bench.py --no-cpu-baseline "$@" > $OUT/$tag.json 2> $OUT/$tag.err )
  line $tag $OUT/$tag.json
}
python3 -c "import torch; print(torch.cuda.get_device_name(0))" 2>/dev/null
for i in $(seq 1 $R); do
  echo "== round $i: driver shape (--steps 20 --warmup 5)"
  if [ "${AB_TREES:-0}" = 1 ]; then
    run r1_drv_$i $ROOT/_ab/r1 "" -- --steps 20 --warmup 5
    run r2_drv_$i $ROOT/_ab/r2 "" -- --steps 20 --warmup 5
  fi
  for v in "${VARS[@]}"; do run ${v%%=*}_drv_$i $ROOT "${v#*=}" -- --steps 20 --warmup 5 ${AB_ARGS:-}; done
  echo "== round $i: default shape (590 + 5900)"
  if [ "${AB_TREES:-0}" = 1 ]; then
    run r1_def_$i $ROOT/_ab/r1 "" --
    run r2_def_$i $ROOT/_ab/r2 "" --
  fi
  for v in "${VARS[@]}"; do run ${v%%=*}_def_$i $ROOT "${v#*=}" -- ${AB_ARGS:-}; done
done
echo "== rocprofv3 kernel trace, default shape"
prof() { # tag tree lib
  local tag=$1 tree=$2 lib=$3
  ( cd $tree && env ${lib:+PCGYM_HIP_LIB=$lib} timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$tag -o t -- python3 bench.py --no-cpu-baseline ${AB_ARGS:-} > $OUT/prof_$tag.json 2> $OUT/prof_$tag.err )
  python3 - $OUT/prof_$tag $tag <<'P'
import csv,glob,sys
d,tag=sys.argv[1],sys.argv[2]
for f in glob.glob(d+'/**/*kernel_stats.csv',recursive=True):
    for r in csv.DictReader(open(f)):
        if 'step_kernel' in r['Name']:
            print(f"{tag:18s} {r['Name'][:70]:70s} calls {r['Calls']:>6s} avg {float(r['AverageNs'])/1e3:7.3f} min {float(r['MinNs'])/1e3:7.3f} max {float(r['MaxNs'])/1e3:7.3f} us")
P
}
if [ "${AB_TREES:-0}" = 1 ]; then prof r1 $ROOT/_ab/r1 ""; prof r2 $ROOT/_ab/r2 ""; fi
for v in "${VARS[@]}"; do prof ${v%%=*} $ROOT "${v#*=}"; done
