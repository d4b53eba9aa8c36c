set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s9
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/s9/pytest_gpu.txt 2>&1; echo "pytest rc $?" >> gpurun_out/s9/pytest_gpu.txt
tail -5 gpurun_out/s9/pytest_gpu.txt
for w in cstr cstr_safe four_tank me10 me10_ros4 me20 cryst cryst_cv8 mixed; do
  python bench.py --workload $w --no-cpu-baseline > gpurun_out/s9/bench_$w.json 2> gpurun_out/s9/bench_$w.err
  python - $w gpurun_out/s9/bench_$w.json <<'P'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[2]) if l.startswith('{')][-1]); r=d['roofline']
    print(f"{sys.argv[1]:10s} value {d['value']:.4e} ms/step {d['ms_per_step']*1e3:9.3f} us kernel {r['kernel_avg_us']:9.2f} us frac {r['frac']:.3f} ({r['bound']}) sane {d['config']['sane']}")
except Exception as e: print(sys.argv[1],'FAILED',e)
P
done
python bench.py --steps 20 --warmup 5 > gpurun_out/s9/bench_driver_shape.json 2>gpurun_out/s9/bench_driver_shape.err; tail -c 1500 gpurun_out/s9/bench_driver_shape.json
python tools/default_cstr_probe.py > gpurun_out/s9/default_cstr_probe.txt 2>&1; cat gpurun_out/s9/default_cstr_probe.txt | grep -v amdgpu
