#!/bin/bash
# idle-lane threshold of a refill in the work-queue kernel (PCG_Q_REFILL): ms per step.  Run ON the GPU box.
python -c "import torch"
for w in me10 me20 mixed; do for r in 8 2 4 12 16 24 8; do
  PCG_Q_REFILL=$r timeout 300 python bench.py --workload $w --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$w','refill',$r,'%.4f ms'%d['ms_per_step'])"
done; done
