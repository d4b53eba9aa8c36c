#!/bin/bash
# rocprofv3 recipe used for profiles/: run ON the GPU box (via gpurun).  usage: tools/prof.sh <tag> [bench args...]
# (round 5: one more SQ pass with the VALU instruction counts by class, for the issue time priced by class: tools/pmc_json.py)
# Pass 1: kernel trace + stats of the bench's default run.  Passes 2..n: one PMC group each (counters collected in their
# own runs, with --kernel-trace only; FETCH_SIZE and WRITE_SIZE cannot share a pass: TCC has 4 slots, they need 3+2).
# PROF_PMC_STEPS / PROF_PMC_WARMUP: length of the counter passes (counters are clock-independent: shorter runs).
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--no-cpu-baseline $*"
PMC_ARGS="--steps ${PROF_PMC_STEPS:-590} --warmup ${PROF_PMC_WARMUP:-59} --no-cpu-baseline $*"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python $ROOT/bench.py $ARGS > $OUT/bench_trace.json 2> $OUT/trace.log
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64" "GRBM_GUI_ACTIVE"; do
  name=$(echo $grp | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/pmc_$name -o p -- python $ROOT/bench.py $PMC_ARGS > /dev/null 2> $OUT/pmc_$name.log
done
python $ROOT/tools/prof_summary.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
