set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5s19; mkdir -p $O
X=$PWD/pc-gym_amd/libpcgym_hip_estrin.so
run() { local label=$1; shift
  ( "$@" > $O/b.json 2> $O/b.err ) ; python - "$label" $O/b.json <<'P'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[2]) if l.startswith('{')][-1]); r=d['roofline']
    print(f"{sys.argv[1]:44s} value {d['value']:.4e} ms/step {d['ms_per_step']*1e3:9.3f} us kernel {r['kernel_avg_us']:9.2f} us sane {d['config'].get('sane')} chain {r.get('chain')}")
except Exception as e: print(sys.argv[1],'FAILED',e, open(sys.argv[2].replace('.json','.err')).read()[-600:])
P
}
for rep in 1 2 3; do
run "cstr_safe horner"   timeout 600 python bench.py --workload cstr_safe --no-cpu-baseline
run "cstr_safe estrin"   env PCGYM_HIP_LIB=$X timeout 600 python bench.py --workload cstr_safe --no-cpu-baseline
run "cstr horner"        timeout 600 python bench.py --no-cpu-baseline
run "cstr estrin"        env PCGYM_HIP_LIB=$X timeout 600 python bench.py --no-cpu-baseline
done
