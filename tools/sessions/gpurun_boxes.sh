# one fresh box per call: three driver-shaped and two default-shaped headline runs (how the number moves from box to box)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/boxes
tag=$(date +%H%M%S)
for i in 1 2 3; do python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | grep '^{' > gpurun_out/boxes/drv_${tag}_$i.json; done
for i in 1 2; do python bench.py --no-cpu-baseline 2>/dev/null | grep '^{' > gpurun_out/boxes/def_${tag}_$i.json; done
python - $tag <<'P'
import json,glob,sys
t=sys.argv[1]
d=[json.load(open(f)) for f in sorted(glob.glob(f'gpurun_out/boxes/drv_{t}_*.json'))]
e=[json.load(open(f)) for f in sorted(glob.glob(f'gpurun_out/boxes/def_{t}_*.json'))]
print(f"box {t}: driver shape value " + " ".join(f"{x['value']:.3e}" for x in d) + " | kernel us " + " ".join(f"{x['roofline']['kernel_avg_us']:.2f}" for x in d) + " || default shape value " + " ".join(f"{x['value']:.3e}" for x in e) + " | kernel us " + " ".join(f"{x['roofline']['kernel_avg_us']:.2f}" for x in e))
P
