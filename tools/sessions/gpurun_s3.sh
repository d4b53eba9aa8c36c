set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s3
V=$PWD/_ab/var
one() { # tag lib env...
  local tag=$1 lib=$2; shift 2
  for i in 1 2; do
    env "$@" ${lib:+PCGYM_HIP_LIB=$lib} python bench.py --no-cpu-baseline > gpurun_out/s3/$tag.$i.json 2>gpurun_out/s3/$tag.$i.err
    python - $tag gpurun_out/s3/$tag.$i.json <<'P'
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
for nt in 1 3 5 7 0; do one new_nt$nt "" PCG_NT=$nt; done
for bpc in 4 5 6 7; do one new_bpc$bpc "" PCG_BPC=$bpc; one new_bpc${bpc}_nt3 "" PCG_BPC=$bpc PCG_NT=3; done
} > gpurun_out/s3/sweep.txt 2>&1
for cfg in "PCG_NT=1" "PCG_NT=3" "PCG_NT=1 PCG_BPC=6" "PCG_NT=3 PCG_BPC=6"; do
  echo "=== timeline $cfg"; env $cfg PCGYM_HIP_LIB=$V/lib_new_TL.so python tools/timeline_probe.py
done > gpurun_out/s3/timeline.txt 2>&1
cat gpurun_out/s3/sweep.txt; cat gpurun_out/s3/timeline.txt
