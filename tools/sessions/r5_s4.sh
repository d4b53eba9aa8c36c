set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5s4; mkdir -p $O
line() { python - "$1" "$2" <<'P'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[2]) if l.startswith('{')][-1]); r=d['roofline']
    print(f"{sys.argv[1]:34s} value {d['value']:.4e} ms/step {d['ms_per_step']*1e3:9.3f} us kernel {r.get('kernel_avg_us',0):9.2f} us sane {d['config'].get('sane')}")
except Exception as e: print(sys.argv[1],'FAILED',e)
P
}
PCG_Q_R4WIDE=0 timeout 300 python bench.py --workload me10_ros4 --no-cpu-baseline --coop-thr 0 > $O/base.json 2> $O/base.err; line "me10_ros4 off" $O/base.json
for cw in 1 2; do for thr in 50 54 58 62 66; do
  PCG_Q_R4WIDE=0 PCG_Q_COOPW=$cw timeout 300 python bench.py --workload me10_ros4 --no-cpu-baseline --coop-thr $thr > $O/me10_cw${cw}_thr$thr.json 2> $O/me10_cw${cw}_thr$thr.err
  line "me10_ros4 coopw $cw thr $thr" $O/me10_cw${cw}_thr$thr.json
done; done
for cw in 1 2; do for thr in 54 60; do
  PCG_Q_R4WIDE=0 PCG_Q_COOPW=$cw timeout 300 python bench.py --workload mixed --no-cpu-baseline --coop-thr $thr > $O/mixed_cw${cw}_thr$thr.json 2> $O/mixed_cw${cw}_thr$thr.err
  line "mixed coopw $cw thr $thr" $O/mixed_cw${cw}_thr$thr.json
done; done
