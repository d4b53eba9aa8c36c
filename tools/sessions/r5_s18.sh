set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5s18; mkdir -p $O
timeout 900 python tools/chunk_safe_probe.py cstr_safe 236 2>&1 | tail -8
timeout 900 python tools/chunk_safe_probe.py me10_ros5 118 2>&1 | tail -8
