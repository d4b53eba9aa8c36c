set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5s28; mkdir -p $O
timeout 2700 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; echo "pytest rc $?" >> $O/pytest_gpu.txt; tail -4 $O/pytest_gpu.txt
PCG_FUZZ_SEEDS=600 timeout 2400 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "random_configurations" > $O/fuzz.txt 2>&1; echo "pytest rc $?" >> $O/fuzz.txt; tail -4 $O/fuzz.txt
timeout 900 python tools/registry_sweep.py > $O/registry_sweep.txt 2>&1; tail -30 $O/registry_sweep.txt
