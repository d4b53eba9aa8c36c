# sort-key weight of the transient term on the launch shapes of round 4 (it was tuned on 256-thread tiles of 1024 slots)
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s42
for i in 1 2; do
for w in me10 me20; do
  for qw in 0 10 20 33 50; do
  PCG_Q_W=$qw python bench.py --workload $w --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=d['roofline']
print('$w q_w=$qw', 'us/step %.1f' % (d['ms_per_step']*1e3), flush=True)"
  done
done
done 2>&1 | tee gpurun_out/s42/q_w_sweep.txt
python bench.py --workload cstr_safe --integrator dopri5 --work-queue --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('cstr_safe dopri5 1e-10 through the work queue: us/step %.1f value %.3e' % (d['ms_per_step']*1e3, d['value']))" | tee gpurun_out/s42/cstr_safe_queue.txt
python bench.py --workload cstr_safe --integrator dopri5 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('cstr_safe dopri5 1e-10 classic kernel: us/step %.1f value %.3e' % (d['ms_per_step']*1e3, d['value']))" | tee -a gpurun_out/s42/cstr_safe_queue.txt
