#!/bin/bash
# Device-side half of the root-cause work on the round-5 bug (see ros_host_check.cpp for the host half): the heat-exchanger
# unit compiled with the UNROLLED Rosenbrock attempt (-DPCG_ROS_ROLLED_ABOVE=9999: the form that returned a garbage x[2])
# under different code-generation settings, each linked with the product's other objects into _ab/libpcgym_hip_<v>.so.
#   tools/hostcheck/device_variants.sh build      (here, no GPU: ~2 min per variant)
#   tools/hostcheck/device_variants.sh run        (on the GPU box: the integrator sweep of heat_exchanger per variant)
set -u
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
cd "$ROOT/pc-gym_amd/csrc"
HIPCC=/opt/rocm/bin/hipcc
BASE="--offload-arch=gfx950 -std=c++17 -fPIC -fvisibility=hidden -DPCG_ROS_ROLLED_ABOVE=9999"
declare -A V
V[unrolled_O3]="-O3"
V[unrolled_O1]="-O1"
V[unrolled_O3_basic_ra]="-O3 -mllvm -amdgpu-sgpr-regalloc=basic -mllvm -amdgpu-vgpr-regalloc=basic"
V[unrolled_O3_sgpr_to_mem]="-O3 -mllvm -amdgpu-spill-sgpr-to-vgpr=false"
V[unrolled_O3_no_sched]="-O3 -mllvm -amdgpu-disable-unclustered-high-rp-reschedule -mllvm -enable-misched=false"
if [ "${1:-build}" = build ]; then
  mkdir -p "$ROOT/_ab"
  for v in "${!V[@]}"; do
    ( $HIPCC $BASE ${V[$v]} -DPCG_SRC_HASH='"variant"' -c -o "$ROOT/_ab/inst_h_$v.o" pcg_inst_h.hip > "$ROOT/_ab/$v.log" 2>&1 \
      && $HIPCC --offload-arch=gfx950 -fPIC -shared -o "$ROOT/_ab/libpcgym_hip_$v.so" $(ls build/*.o | grep -v pcg_inst_h.o) "$ROOT/_ab/inst_h_$v.o" -lhiprtc \
      && echo "built $v" ) &
  done
  wait
else
  cd "$ROOT"
  for v in product "${!V[@]}"; do
    lib="$ROOT/_ab/libpcgym_hip_$v.so"; [ $v = product ] && lib="$ROOT/pc-gym_amd/libpcgym_hip.so"
    echo "=== $v"
    PCGYM_HIP_LIB=$lib python -m pytest tests/test_gpu_sweeps.py -q -m gpu -p no:cacheprovider \
      -k "test_integrator_sweep and heat_exchanger and (rodas4 or rodas5) and (auto or classic) and lean" 2>&1 | tail -3
    PCGYM_HIP_LIB=$lib python tools/hostcheck/lane_pattern.py 2>&1 | tail -6
  done
fi
