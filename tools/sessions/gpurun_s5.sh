set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s5
V=$PWD/_ab/var
cat > /tmp/cfgs.txt <<'C'
head|$V/lib_head.so|X=1
both|$V/lib_both.so|X=1
new_bpc6|$V/lib_new.so|PCG_BPC=6
new_bpc6_late16B|$V/lib_new.so|PCG_BPC=6 PCG_LEAN_PRIO=1016B
new_bpc6_lateE50|$V/lib_new.so|PCG_BPC=6 PCG_LEAN_PRIO=10E50
new_bpc6_late1B|$V/lib_new.so|PCG_BPC=6 PCG_LEAN_PRIO=1001B
new_bpc6_late5AF|$V/lib_new.so|PCG_BPC=6 PCG_LEAN_PRIO=105AF
new_bpc4|$V/lib_new.so|PCG_BPC=4
new_bpc4_late1B|$V/lib_new.so|PCG_BPC=4 PCG_LEAN_PRIO=1001B
new_bpc4_lateE4|$V/lib_new.so|PCG_BPC=4 PCG_LEAN_PRIO=100E4
new_bpc5|$V/lib_new.so|PCG_BPC=5
new_bpc5_late6B|$V/lib_new.so|PCG_BPC=5 PCG_LEAN_PRIO=1006B
new_bpc8_late5AF|$V/lib_new.so|PCG_LEAN_PRIO=105AF
new_bpc8_late56B|$V/lib_new.so|PCG_LEAN_PRIO=1056B
C
for r in 1 2 3; do
  while IFS='|' read -r tag lib envs; do
    lib=$(eval echo $lib)
    env $envs PCGYM_HIP_LIB=$lib python bench.py --no-cpu-baseline > gpurun_out/s5/$tag.$r.json 2>gpurun_out/s5/$tag.$r.err
  done < /tmp/cfgs.txt
done
python - <<'P' > gpurun_out/s5/sweep.txt
import json,glob,os,statistics
rows={}
for f in sorted(glob.glob('gpurun_out/s5/*.json')):
    tag=os.path.basename(f).rsplit('.',2)[0]
    try:
        d=json.loads([l for l in open(f) if l.startswith('{')][-1]); rows.setdefault(tag,[]).append(d['roofline']['kernel_avg_us'])
    except Exception as e: rows.setdefault(tag,[]).append(float('nan'))
for tag,v in sorted(rows.items(), key=lambda kv: statistics.median(kv[1])):
    print(f"{tag:24s} median {statistics.median(v):6.2f}  runs "+" ".join(f"{x:6.2f}" for x in v))
P
cat gpurun_out/s5/sweep.txt
for cfg in "head_TL X=1" "both_TL X=1" "new_TL PCG_BPC=6" "new_TL PCG_BPC=6 PCG_LEAN_PRIO=1016B" "new_TL PCG_BPC=4 PCG_LEAN_PRIO=1001B"; do
  set -- $cfg; lib=$1; shift
  echo "=== timeline $lib $*"; env "$@" PCGYM_HIP_LIB=$V/lib_$lib.so python tools/timeline_probe.py
done > gpurun_out/s5/timeline.txt 2>&1
grep -v "^    [0-9]\|^   1[0-9]" gpurun_out/s5/timeline.txt
