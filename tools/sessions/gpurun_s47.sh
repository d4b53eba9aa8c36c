set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s47
PCGYM_HIP_LIB=_ab/qstats_a.so python tools/queue_probe.py cstr_safe 2>&1 | grep -v amdgpu | tail -16 | tee gpurun_out/s47/queue_probe_cstr_safe_fixup.txt
