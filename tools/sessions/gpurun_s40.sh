set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s40
timeout 900 python -m pytest tests/test_gpu_round2.py -m gpu -x -q -k "launch_shapes or tile_sort" > gpurun_out/s40/pytest_new.txt 2>&1; tail -15 gpurun_out/s40/pytest_new.txt
