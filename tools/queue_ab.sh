python -c "import torch"
timeout 900 python -m pytest tests -q -m gpu -k "work_queue or mixed or me_ or extraction or batched_step" -x 2>&1 | tail -5
for w in me10 me20 mixed; do for v in 0 1 0 1; do
  if [ $v = 1 ]; then export PCG_VARIANT=1; else unset PCG_VARIANT; fi
  timeout 300 python bench.py --workload $w 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$w','variant',$v,'%.4f ms'%d['ms_per_step'],'%.3e'%d['value'])"
done; done
