# hygiene on the final build: whole GPU suite with every buffer its own allocation and blocking launches; work-queue soak
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s55
PYTORCH_NO_CUDA_MEMORY_CACHING=1 HIP_LAUNCH_BLOCKING=1 timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/s55/pytest_gpu_nocache_blocking.txt 2>&1; tail -3 gpurun_out/s55/pytest_gpu_nocache_blocking.txt
timeout 900 python tools/queue_soak.py > gpurun_out/s55/queue_soak.txt 2>&1; tail -5 gpurun_out/s55/queue_soak.txt
