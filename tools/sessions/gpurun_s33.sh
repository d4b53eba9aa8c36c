set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s33
timeout 900 python -m pytest tests/test_bench_contract.py -m gpu -x -q > gpurun_out/s33/pytest_bench_contract.txt 2>&1; tail -15 gpurun_out/s33/pytest_bench_contract.txt
