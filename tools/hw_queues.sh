#!/bin/bash
# The stream / hardware-queue A-B of profiles/r05_hw_queues.txt on one box: the step under GPU_MAX_HW_QUEUES = 1..8 (HIP multiplexes
# its streams onto that many hardware queues; the package keeps to the null stream + three side streams, so 4 and above must agree)
# and under 1..3 discriminator streams.
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
B="--steps 25 --warmup 4 --no-cpu-baseline --no-pointwise --no-companions"
for r in 1 2; do
  for q in 1 2 3 4 5 8; do
    echo -n "[GPU_MAX_HW_QUEUES=$q] "
    env GPU_MAX_HW_QUEUES=$q python3 bench.py $B 2>/dev/null | tail -1 | grep -o '"ms_per_step": [0-9.]*'
  done
  for d in 3 2 1; do
    echo -n "[SCAN_DIS_STREAMS=$d] "
    env SCAN_DIS_STREAMS=$d python3 bench.py $B 2>/dev/null | tail -1 | grep -o '"ms_per_step": [0-9.]*'
  done
done
