#!/bin/bash
# bench.py --workload cstr_rollout under the lean rollout kernel's switches (GPU box): trajectory stores non-temporal or
# plain (A/B library), one chunk per workgroup or a persistent grid of n workgroups per CU
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd $ROOT
for lib in pc-gym_amd/libpcgym_hip.so _ab/libpcgym_hip_nont.so; do for bpc in 0 2 3 4 6 8; do
  e=""; [ $bpc != 0 ] && e="PCG_ROLL_BPC=$bpc"
  r=$(env PCGYM_HIP_LIB=$ROOT/$lib $e python bench.py --workload cstr_rollout --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('%.2f us per step, kernel %.2f, frac %.3f' % (d['ms_per_step']*1e3, d['roofline']['kernel_avg_us'], d['roofline']['frac']))")
  echo "$lib workgroups per CU $bpc (0 = one chunk each): $r"
done; done
