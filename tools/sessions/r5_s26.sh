set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5s26; mkdir -p $O
timeout 1500 python tools/integrator_sweep.py > $O/ros_dense_sweep.txt 2>&1; echo "rc $?" >> $O/ros_dense_sweep.txt; grep -c "worst rel diff" $O/ros_dense_sweep.txt; grep -E "BAD|bad combinations|rc |Error" $O/ros_dense_sweep.txt | tail -8
timeout 1500 python -m pytest tests/test_gpu_rodas4.py tests/test_gpu_user_model.py -m gpu -x -q > $O/pytest.txt 2>&1; echo "pytest rc $?" >> $O/pytest.txt; tail -4 $O/pytest.txt
PCG_FUZZ_SEEDS=600 timeout 2400 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "random_configurations" > $O/fuzz.txt 2>&1; echo "pytest rc $?" >> $O/fuzz.txt; tail -4 $O/fuzz.txt
