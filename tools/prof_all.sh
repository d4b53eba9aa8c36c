#!/bin/bash
# Round-3 profile set, run ON the GPU box (gpurun): for every bench workload one rocprofv3 --kernel-trace --stats pass of
# the default bench command and four PMC passes (FETCH_SIZE | WRITE_SIZE | SQ group | GRBM_GUI_ACTIVE -- separate runs:
# the TCC block has 4 counter slots, FETCH_SIZE takes 3 and WRITE_SIZE 2), then tools/pmc_json.py condenses them.
#   usage: tools/prof_all.sh [workload ...]      output: gpurun_out/prof_<workload>/summary.txt, gpurun_out/pmc.json
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
WLS=${@:-cstr cstr_safe cstr_rollout cstr_safe_rollout cstr_unc four_tank me10 me10_ros4 me10_ros5 me20 cryst cryst_cv8 mixed}
for w in $WLS; do
  extra="--workload $w"
  PROF_PMC_STEPS=${PROF_PMC_STEPS:-118} PROF_PMC_WARMUP=${PROF_PMC_WARMUP:-12} bash $ROOT/tools/prof.sh $w $extra > /dev/null 2>&1
  echo "== $w"; grep -E "step_kernel|rollout" $ROOT/gpurun_out/prof_$w/summary.txt | head -6
done
python $ROOT/tools/pmc_json.py $ROOT/gpurun_out $WLS > $ROOT/gpurun_out/pmc.json
