#!/bin/bash
# sort-key weight sweep for the work-queue kernel (PCG_Q_W): ms per step of me10 / me20 / mixed.  Run ON the GPU box.
python -c "import torch"
for w in me10 me20 mixed; do for qw in ${QW_LIST:-0 15 33 60 33 0}; do
  PCG_Q_W=$qw timeout 300 python bench.py --workload $w --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$w','q_w',$qw,'%.4f ms'%d['ms_per_step'],'%.3e'%d['value'])"
done; done
