#!/bin/bash
# Measurement builds: recompile ONLY the cstr / four_tank unit (pcg_inst_a.hip) with extra -D flags and link it against the
# other objects of the last full build (UNIT=pcg_inst_j for another unit).  usage: tools/fastlib.sh <out.so> [-DFLAG ...]
set -e
cd "$(dirname "$0")/../pc-gym_amd/csrc"
OUT=$1; shift
UNIT=${UNIT:-pcg_inst_a}
H=$(cat $(ls *.hpp | sort) ../../include/pcgym_hip.h pcg_abi.hip $(ls pcg_inst_*.hip | sort) | sha256sum | cut -c1-32)
TMP=$(mktemp -d)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -DPCG_SRC_HASH="\"$H\"" "$@" -c -o $TMP/a.o $UNIT.hip 2>/dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -shared -o $OUT $TMP/a.o $(ls build/*.o | grep -v $UNIT.o) -lhiprtc
rm -rf $TMP
echo built $OUT
