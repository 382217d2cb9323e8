#!/bin/bash
# HIP maps streams to 4 hardware queues in creation order: N unrelated streams created before the trainer (bench.py --extra-streams N)
# shift the mapping.  The step's streams are verified to run concurrently (coot_stream_create_concurrent), so ms_per_step must not
# depend on N — single call, and the data-parallel phase path with its collectives as direct RCCL calls / through torch.distributed.
#   bash tools/stream_queues.sh [steps] [warmup]   -> stdout (profiles/r06_stream_queues.txt)
K=${1:-30}; W=${2:-8}
line() { python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$1', d['ms_per_step'])"; }
for N in 0 1 2 3 4 5; do
  python bench.py --steps $K --warmup $W --no-cpu-baseline --no-roofline --extra-streams $N 2>/dev/null | line "extra_streams=$N single"
  for V in direct torch; do
    COOT_DP_COLLECTIVES=$V python bench.py --steps $K --warmup $W --no-cpu-baseline --no-roofline --force-dp --extra-streams $N 2>/dev/null | line "extra_streams=$N dp1 collectives=$V"
  done
done
