set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s59
timeout 1500 python -m pytest tests -m gpu -x -q -p no:cacheprovider > gpurun_out/s59/pytest_gpu.txt 2>&1; echo "pytest rc $?" >> gpurun_out/s59/pytest_gpu.txt
tail -3 gpurun_out/s59/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
python bench.py --steps 20 --warmup 5 > gpurun_out/s59/bench_driver_shape.json 2>gpurun_out/s59/bench_driver_shape.err; tail -c 2500 gpurun_out/s59/bench_driver_shape.json
python bench.py > gpurun_out/s59/bench_default.json 2>gpurun_out/s59/bench_default.err; tail -c 2500 gpurun_out/s59/bench_default.json
