# DOPRI5 rows folded for NX > 4 only: full GPU suite, then the lines it touches
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s25
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/s25/pytest_gpu.txt 2>&1; tail -4 gpurun_out/s25/pytest_gpu.txt
for w in me20 me10 mixed cstr_safe; do
  for i in 1 2; do
  python bench.py --workload $w --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=d['roofline']
print('$w', 'ms/step %.4f kernel %.1f us value %.3e' % (d['ms_per_step'], r['kernel_avg_us'], d['value']), flush=True)"
  done
done 2>&1 | tee gpurun_out/s25/bench.txt
