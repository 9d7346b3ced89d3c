#!/bin/bash
# the other BASELINE.json workloads on one GPU: tools/other_workloads.sh [tag] -> gpurun_out/<tag>_other_workloads.txt
TAG=${1:-r04}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
F="--no-cpu-baseline --no-pointwise --no-companions --steps 5 --warmup 2"
run() {
  echo "## bench.py $*"
  timeout 600 python3 bench.py $F "$@" 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
r=d['roofline']
print('   %.2f ms/step  %.2f pairs/s  dominant %s %.1f TF = %.3f' % (d['ms_per_step'], d['value'], r['kernel'], r['achieved'], r['frac']))"
}
{
  run --model s2c
  run --model k2c_r50
  run --height 1333 --width 2666 --batch 8
  run --scaling strong --global-batch 8
  run --forward-target --ft-positives 0.01
  run --forward-target
} > gpurun_out/${TAG}_other_workloads.txt 2>&1
cat gpurun_out/${TAG}_other_workloads.txt
