set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5s9; mkdir -p $O
for rep in 1 2 3; do for lib in folded unfolded; do for w in me10 me10_ros4; do
  if [ $lib = unfolded ]; then export PCGYM_HIP_LIB=_ab/me_unfolded.so; else unset PCGYM_HIP_LIB; fi
  timeout 600 python bench.py --workload $w --no-cpu-baseline > $O/b.json 2> $O/b.err
  python - "$w $lib" $O/b.json <<'P'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[2]) if l.startswith('{')][-1]); r=d['roofline']
    print(f"{sys.argv[1]:22s} ms/step {d['ms_per_step']*1e3:9.3f} us attempts {r.get('attempted_steps_mean')}")
except Exception as e: print(sys.argv[1],'FAILED',e)
P
done; done; done
