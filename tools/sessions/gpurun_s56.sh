set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s56
echo "== before" | tee gpurun_out/s56/tsit5_probe.txt
PCGYM_HIP_LIB=_ab/lib_before_t5acc.so timeout 600 python tools/tsit5_probe.py 2>&1 | grep -v amdgpu | tee -a gpurun_out/s56/tsit5_probe.txt
echo "== after (accumulator form for NX > 10)" | tee -a gpurun_out/s56/tsit5_probe.txt
timeout 600 python tools/tsit5_probe.py 2>&1 | grep -v amdgpu | tee -a gpurun_out/s56/tsit5_probe.txt
timeout 900 python -m pytest tests/test_gpu_tsit5.py tests/test_gpu_reference_engine.py -m gpu -x -q 2>&1 | tail -3
