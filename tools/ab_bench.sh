#!/bin/bash
# same-box A/B of environment switches: tools/ab_bench.sh "<env A>" "<env B>" [rounds]  -> ms/step of alternating runs
A="$1"; B="$2"; R=${3:-3}
for r in $(seq $R); do
  for v in "$A" "$B"; do
    echo -n "[$v] "
    env $v python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-pointwise --no-companions 2>/dev/null | tail -1 | grep -o '"ms_per_step": [0-9.]*' | tr "\n" " "
    echo
  done
done
