set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5s27; mkdir -p $O
timeout 600 python tools/he_dbg.py 2>&1 | grep "^rodas" | cut -c1-200
timeout 1500 python tools/integrator_sweep.py > $O/ros_dense_sweep.txt 2>&1; echo "rc $?" >> $O/ros_dense_sweep.txt; grep -c "worst rel diff" $O/ros_dense_sweep.txt; grep -E "BAD|bad combinations|rc |Error" $O/ros_dense_sweep.txt | tail -8
