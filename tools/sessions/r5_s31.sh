set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5s31; mkdir -p $O
for rep in 1 2 3; do timeout 600 python bench.py --workload cstr_safe --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('cstr_safe value %.4e  %.3f us per step' % (d['value'], d['ms_per_step']*1e3))"; done
timeout 600 python tools/queue_soak.py > $O/queue_soak.txt 2>&1; tail -1 $O/queue_soak.txt
timeout 2700 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; echo "pytest rc $?" >> $O/pytest_gpu.txt; tail -3 $O/pytest_gpu.txt
