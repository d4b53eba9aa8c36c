set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s4
V=$PWD/_ab/var
one() { # tag lib env...
  local tag=$1 lib=$2; shift 2
  for i in 1 2; do
    env "$@" ${lib:+PCGYM_HIP_LIB=$lib} python bench.py --no-cpu-baseline > gpurun_out/s4/$tag.$i.json 2>gpurun_out/s4/$tag.$i.err
    python - $tag gpurun_out/s4/$tag.$i.json <<'P'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[2]) if l.startswith('{')][-1]); r=d['roofline']
    print(f"{sys.argv[1]:28s} value {d['value']:.4e} ms/step {d['ms_per_step']*1e3:7.3f} kernel {r['kernel_avg_us']:6.2f} sane {d['config']['sane']}")
except Exception as e: print(sys.argv[1],'FAILED',e)
P
  done
}
{
one head $V/lib_head.so X=1
one both $V/lib_both.so X=1
one new_bpc6 "" PCG_BPC=6
for m in 16B 5AF 1B E50 333 FFF; do one new_bpc6_prio$m "" PCG_BPC=6 PCG_LEAN_PRIO=$m; done
one new_bpc8 "" X=1
for m in 5AF 56B 1B FFFF E4E4; do one new_bpc8_prio$m "" PCG_LEAN_PRIO=$m; done
one new_bpc4 "" PCG_BPC=4
for m in 1B E4 F; do one new_bpc4_prio$m "" PCG_BPC=4 PCG_LEAN_PRIO=$m; done
one new_bpc5_prio6B "" PCG_BPC=5 PCG_LEAN_PRIO=6B
one head_again $V/lib_head.so X=1
} > gpurun_out/s4/sweep.txt 2>&1
for cfg in "PCG_BPC=6" "PCG_BPC=6 PCG_LEAN_PRIO=16B" "PCG_BPC=6 PCG_LEAN_PRIO=1B" "PCG_BPC=4 PCG_LEAN_PRIO=1B"; do
  echo "=== timeline $cfg"; env $cfg PCGYM_HIP_LIB=$V/lib_new_TL.so python tools/timeline_probe.py
done > gpurun_out/s4/timeline.txt 2>&1
cat gpurun_out/s4/sweep.txt
