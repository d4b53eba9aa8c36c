set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s10
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/s10/pytest_gpu.txt 2>&1; echo "pytest rc $?" >> gpurun_out/s10/pytest_gpu.txt
tail -6 gpurun_out/s10/pytest_gpu.txt
python tools/default_cstr_probe.py 2>&1 | grep -v amdgpu | tee gpurun_out/s10/default_cstr_probe.txt
PCG_T5G_ONE_LAUNCH=1 python tools/default_cstr_probe.py 2>&1 | grep -v amdgpu | head -1 | sed 's/^/one-launch form: /' | tee -a gpurun_out/s10/default_cstr_probe.txt
for i in 1 2 3; do
python bench.py --workload cstr_safe --no-cpu-baseline > gpurun_out/s10/bench_cstr_safe.$i.json 2>/dev/null
python bench.py --no-cpu-baseline --steps 20 --warmup 5 > gpurun_out/s10/bench_drv.$i.json 2>/dev/null
done
python - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/s10/bench_*.json')):
    d=json.loads([l for l in open(f) if l.startswith('{')][-1]); r=d['roofline']
    print(f"{f.split('/')[-1]:28s} value {d['value']:.4e} ms/step {d['ms_per_step']*1e3:9.3f} us kernel {r['kernel_avg_us']:9.2f} us")
P
