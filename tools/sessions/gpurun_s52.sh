set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s52
for i in 1 2 3; do
  timeout 300 python bench.py --workload me20 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('me20 us/step %.1f' % (d['ms_per_step']*1e3), flush=True)"
  PCG_Q_NOXLDS=1 timeout 300 python bench.py --workload me20 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('me20 NOXLDS us/step %.1f' % (d['ms_per_step']*1e3), flush=True)"
done 2>&1 | tee gpurun_out/s52/me20.txt
PCGYM_HIP_LIB=_ab/qstats_j.so timeout 300 python tools/queue_probe.py me20 2>&1 | grep -v amdgpu | tail -15 | tee gpurun_out/s52/probe.txt
