set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5s2; mkdir -p $O
for thr in 0 48 62; do
  PCGYM_HIP_LIB=_ab/qstats_i.so timeout 300 python tools/queue_probe.py me10_ros4 $thr > $O/probe_thr$thr.txt 2>&1
  echo "=== thr $thr"; tail -28 $O/probe_thr$thr.txt
done
