set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5s20; mkdir -p $O
X=$PWD/pc-gym_amd/libpcgym_hip_tail.so
run() { local label=$1; shift
  ( "$@" > $O/b.json 2> $O/b.err ) ; python - "$label" $O/b.json <<'P'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[2]) if l.startswith('{')][-1]); r=d['roofline']
    print(f"{sys.argv[1]:44s} value {d['value']:.4e} ms/step {d['ms_per_step']*1e3:9.3f} us kernel {r['kernel_avg_us']:9.2f} us sane {d['config'].get('sane')} heaviest {(r.get('chain') or {}).get('heaviest_env_attempts_last_step')}")
except Exception as e: print(sys.argv[1],'FAILED',e, open(sys.argv[2].replace('.json','.err')).read()[-600:])
P
}
for rep in 1 2 3; do
run "cstr_safe base"   timeout 600 python bench.py --workload cstr_safe --no-cpu-baseline
run "cstr_safe tail"   env PCGYM_HIP_LIB=$X timeout 600 python bench.py --workload cstr_safe --no-cpu-baseline
done
PCGYM_HIP_LIB=$X timeout 1500 python -m pytest tests -m gpu -x -q -k "guard or fixup or fix_up or t5g or rk4g or two_launch or cstr_safe or ignition or default_plan" > $O/pytest_tail.txt 2>&1; echo "pytest rc $?" >> $O/pytest_tail.txt
tail -5 $O/pytest_tail.txt
