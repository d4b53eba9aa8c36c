set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5s3; mkdir -p $O
line() { python - "$1" "$2" <<'P'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[2]) if l.startswith('{')][-1]); r=d['roofline']
    print(f"{sys.argv[1]:34s} value {d['value']:.4e} ms/step {d['ms_per_step']*1e3:9.3f} us kernel {r.get('kernel_avg_us',0):9.2f} us sane {d['config'].get('sane')}")
except Exception as e: print(sys.argv[1],'FAILED',e)
P
}
for wide in 0 1; do for thr in 0 48 56 62 66; do
  PCG_Q_R4WIDE=$wide timeout 300 python bench.py --workload me10_ros4 --no-cpu-baseline --coop-thr $thr > $O/me10_w${wide}_thr$thr.json 2> $O/me10_w${wide}_thr$thr.err
  line "me10_ros4 wide $wide thr $thr" $O/me10_w${wide}_thr$thr.json
done; done
for wide in 0 1; do for thr in 0 56; do
  PCG_Q_R4WIDE=$wide timeout 300 python bench.py --workload mixed --no-cpu-baseline --coop-thr $thr > $O/mixed_w${wide}_thr$thr.json 2> $O/mixed_w${wide}_thr$thr.err
  line "mixed wide $wide thr $thr" $O/mixed_w${wide}_thr$thr.json
done; done
PCG_Q_R4WIDE=1 PCG_Q_HALF_MIN4=5 timeout 300 python bench.py --workload mixed --no-cpu-baseline --coop-thr 56 > $O/mixed_w1_half.json 2> $O/mixed_w1_half.err
line "mixed wide 1 thr 56 half tiles" $O/mixed_w1_half.json
PCG_Q_R4WIDE=1 PCGYM_HIP_LIB=_ab/qstats_i.so timeout 300 python tools/queue_probe.py me10_ros4 56 > $O/probe_wide_thr56.txt 2>&1; tail -28 $O/probe_wide_thr56.txt
