set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s49
for i in 1 2; do
for pr in 0 32 64 128 256; do
  PCG_Q_PRIO=$pr timeout 300 python bench.py --workload cstr_safe --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('cstr_safe q_prio=$pr us/step %.1f value %.3e sane %s' % (d['ms_per_step']*1e3, d['value'], d['config']['sane']), flush=True)"
done
done 2>&1 | tee gpurun_out/s49/fixup_prio.txt
