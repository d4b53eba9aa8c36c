set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5s7; mkdir -p $O
timeout 2700 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; echo "pytest rc $?" >> $O/pytest_gpu.txt
tail -8 $O/pytest_gpu.txt
