cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/s6
{ echo "== default env"; python tools/region_probe.py; echo "== HSA_ENABLE_INTERRUPT=0"; HSA_ENABLE_INTERRUPT=0 python tools/region_probe.py; echo "== GPU_MAX_HW_QUEUES=1"; GPU_MAX_HW_QUEUES=1 python tools/region_probe.py; } > gpurun_out/s6/region.txt 2>&1
cat gpurun_out/s6/region.txt
