#!/bin/bash
# kernel resource usage of one translation unit: tools/kres.sh pcg_inst_i.hip [filter-regex]
cd "$(dirname "$0")/../pc-gym_amd/csrc" || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -c -o /dev/null "$1" \
  -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c "
import sys,re
flt=re.compile(sys.argv[1]) if len(sys.argv)>1 else None
txt=sys.stdin.read()
for b in re.split(r'remark: [^\n]*Function Name: ',txt)[1:]:
    name=b.split('\n')[0]
    dem=name
    if flt and not flt.search(name): continue
    g=lambda k:(re.search(k+r': (\d+)',b) or [None,'?'])[1]
    print(name[:150],'| VGPR',g('VGPRs'),'AGPR',g('AGPRs'),'SGPR',g('SGPRs'),'scratch',g(r'ScratchSize \[bytes/lane\]'),'occ',g(r'Occupancy \[waves/SIMD\]'),'LDS',g(r'LDS Size \[bytes/block\]'))
" "${2:-.}"
