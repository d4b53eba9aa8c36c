set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s7
V=$PWD/_ab/var
cat > /tmp/cfgs.txt <<'C'
head_both|$V/lib_head.so|X=1|--steps 20 --warmup 5
head_copy|$V/lib_head.so|PCG_BENCH_PREHEAT=copy|--steps 20 --warmup 5
head_matmul|$V/lib_head.so|PCG_BENCH_PREHEAT=matmul|--steps 20 --warmup 5
head_none|$V/lib_head.so|X=1|--steps 20 --warmup 5 --preheat-ms 0
head_none_w60|$V/lib_head.so|X=1|--steps 20 --warmup 60 --preheat-ms 0
new5_both|$V/lib_new.so|PCG_BPC=5|--steps 20 --warmup 5
new5_copy|$V/lib_new.so|PCG_BPC=5 PCG_BENCH_PREHEAT=copy|--steps 20 --warmup 5
new5_none|$V/lib_new.so|PCG_BPC=5|--steps 20 --warmup 5 --preheat-ms 0
new5_both_graphless_w590|$V/lib_new.so|PCG_BPC=5|--steps 20 --warmup 590
C
for r in 1 2 3 4 5; do
  while IFS='|' read -r tag lib envs args; do
    lib=$(eval echo $lib)
    env $envs PCGYM_HIP_LIB=$lib python bench.py --no-cpu-baseline $args > gpurun_out/s7/$tag.$r.json 2>gpurun_out/s7/$tag.$r.err
  done < /tmp/cfgs.txt
done
python - <<'P' > gpurun_out/s7/sweep.txt
import json,glob,os,statistics
rows={}
for f in sorted(glob.glob('gpurun_out/s7/*.json')):
    tag=os.path.basename(f).rsplit('.',2)[0]
    try:
        d=json.loads([l for l in open(f) if l.startswith('{')][-1]); rows.setdefault(tag,[]).append((d['ms_per_step']*1e3,d['roofline']['kernel_avg_us']))
    except Exception as e: pass
for tag,v in sorted(rows.items(), key=lambda kv: statistics.median(x[0] for x in kv[1])):
    print(f"{tag:26s} ms/step median {statistics.median(x[0] for x in v):6.2f} kernel median {statistics.median(x[1] for x in v):6.2f} | "+" ".join(f"{x[0]:5.2f}/{x[1]:5.2f}" for x in v))
P
cat gpurun_out/s7/sweep.txt
