# classic DOPRI5 in accumulator form for models with more than ten states: the registry's default paths, before / after
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s38
echo "== before (library of the previous commit)" | tee gpurun_out/s38/registry_sweep.txt
PCGYM_HIP_LIB=_ab/lib_before_acc.so python tools/registry_sweep.py 2>&1 | grep -v amdgpu | tee -a gpurun_out/s38/registry_sweep.txt
echo "== after" | tee -a gpurun_out/s38/registry_sweep.txt
python tools/registry_sweep.py 2>&1 | grep -v amdgpu | tee -a gpurun_out/s38/registry_sweep.txt
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/s38/pytest_gpu.txt 2>&1; tail -3 gpurun_out/s38/pytest_gpu.txt
