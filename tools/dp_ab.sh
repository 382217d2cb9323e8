#!/bin/bash
# A/B on the GPU box: the data-parallel step with one rank (phase calls + collectives) with the collectives as direct RCCL calls
# (default) / through torch.distributed (COOT_DP_COLLECTIVES=torch) against the single-call step, interleaved.
#   bash tools/dp_ab.sh [rounds] [steps] [warmup]
R=${1:-3}; K=${2:-50}; W=${3:-10}
for i in $(seq 1 $R); do
  for V in direct torch; do
    COOT_DP_COLLECTIVES=$V python bench.py --steps $K --warmup $W --no-cpu-baseline --no-roofline --force-dp 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('dp', '$V', d['config'].get('collectives','?')[:5], d['ms_per_step'])"
  done
  python bench.py --steps $K --warmup $W --no-cpu-baseline --no-roofline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('single', d['ms_per_step'])"
done
