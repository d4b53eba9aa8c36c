set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5s24; mkdir -p $O
PCG_FUZZ_SEEDS=600 timeout 2400 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "random_configurations" > $O/fuzz.txt 2>&1; echo "pytest rc $?" >> $O/fuzz.txt; tail -15 $O/fuzz.txt
