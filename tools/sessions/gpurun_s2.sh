set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s2
tools/issuebench > gpurun_out/s2/issuebench.txt 2>&1
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/s2/pytest_gpu.txt 2>&1; echo "pytest rc $?" >> gpurun_out/s2/pytest_gpu.txt
PCGYM_HIP_LIB=$PWD/_ab/var/lib_new_DIV_NR1.so timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_erk.py -m gpu -q > gpurun_out/s2/pytest_gpu_nr1.txt 2>&1; echo "pytest rc $?" >> gpurun_out/s2/pytest_gpu_nr1.txt
V=$PWD/_ab/var
bash tools/headline_ab.sh 3 head=$V/lib_head.so both=$V/lib_both.so nr1=$V/lib_new_DIV_NR1.so taylor13=$V/lib_new_EXP_TAYLOR13.so > gpurun_out/s2/ab.txt 2>&1
tail -3 gpurun_out/s2/pytest_gpu.txt; tail -3 gpurun_out/s2/pytest_gpu_nr1.txt; cat gpurun_out/s2/issuebench.txt; cat gpurun_out/s2/ab.txt
