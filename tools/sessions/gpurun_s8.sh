set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s8
V=$PWD/_ab/var
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/s8/pytest_gpu.txt 2>&1; echo "pytest rc $?" >> gpurun_out/s8/pytest_gpu.txt
for r in 1 2 3 4 5; do
  for v in head cur; do
    lib=""; [ $v = head ] && lib=$V/lib_head.so
    env ${lib:+PCGYM_HIP_LIB=$lib} python bench.py --no-cpu-baseline --steps 20 --warmup 5 > gpurun_out/s8/${v}_drv.$r.json 2>gpurun_out/s8/${v}_drv.$r.err
    env ${lib:+PCGYM_HIP_LIB=$lib} python bench.py --no-cpu-baseline > gpurun_out/s8/${v}_def.$r.json 2>gpurun_out/s8/${v}_def.$r.err
  done
done
python - <<'P' > gpurun_out/s8/sweep.txt
import json,glob,os,statistics
rows={}
for f in sorted(glob.glob('gpurun_out/s8/*.json')):
    tag=os.path.basename(f).rsplit('.',2)[0]
    try:
        d=json.loads([l for l in open(f) if l.startswith('{')][-1]); rows.setdefault(tag,[]).append((d['value'],d['ms_per_step']*1e3,d['roofline']['kernel_avg_us']))
    except Exception as e: pass
for tag,v in sorted(rows.items()):
    print(f"{tag:12s} value median {statistics.median(x[0] for x in v):.4e} ms/step median {statistics.median(x[1] for x in v):6.2f} kernel median {statistics.median(x[2] for x in v):6.2f} | "+" ".join(f"{x[1]:5.2f}/{x[2]:5.2f}" for x in v))
P
cat gpurun_out/s8/sweep.txt; tail -4 gpurun_out/s8/pytest_gpu.txt
