#!/bin/bash
# A/B builds of the second pass of the barrier-free rollout: only the cstr unit carries the kernel, so a variant is that unit
# recompiled and linked with the product's other objects (_ab/libpcgym_hip_flat_<v>.so).  build here, run on the GPU box.
ROOT=$(cd "$(dirname "$0")/.." && pwd); cd "$ROOT/pc-gym_amd/csrc"
declare -A V
V[base]=""
V[wpe4]="-DPCG_FLAT_WPE=4"
V[att2]="-DPCG_FLAT_ATT=2"
V[wpe4_att2]="-DPCG_FLAT_WPE=4 -DPCG_FLAT_ATT=2"
if [ "${1:-build}" = build ]; then
  mkdir -p "$ROOT/_ab"
  for v in "${!V[@]}"; do
    ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden ${V[$v]} -DPCG_SRC_HASH='"variant"' -c -o "$ROOT/_ab/inst_a_$v.o" pcg_inst_a.hip > "$ROOT/_ab/flat_$v.log" 2>&1 \
      && /opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -o "$ROOT/_ab/libpcgym_hip_flat_$v.so" $(ls build/*.o | grep -v pcg_inst_a.o) "$ROOT/_ab/inst_a_$v.o" -lhiprtc && echo "built $v" ) &
  done; wait
else
  cd "$ROOT"
  for v in "${!V[@]}"; do for w in 2 3 4; do for m in 1 2 3; do
    r=$(PCGYM_HIP_LIB=$ROOT/_ab/libpcgym_hip_flat_$v.so PCG_FLAT_WPS=$w PCG_FLAT_EVERY=$m timeout 300 python bench.py --workload cstr_safe_rollout --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('%.1f' % (d['ms_per_step']*1e3))")
    echo "$v: waves per SIMD $w, boundaries every $m: $r us per step"
  done; done; done
fi
