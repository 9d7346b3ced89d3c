#!/bin/bash
# same-box A/B of an environment switch, per kernel record: tools/ab_kernel.sh "<env A>" "<env B>" <record-substring> [rounds]
#   -> ms/step and the HIP-event average of the matching kernel records (roofline.all_conv_kernels of bench.py's serial steps)
A="$1"; B="$2"; K="$3"; R=${4:-3}
for r in $(seq $R); do
  for v in "$A" "$B"; do
    echo -n "[$v] "
    env $v python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-pointwise --no-companions 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
ks=d['roofline']['all_conv_kernels']
print('%.2f ms/step ' % d['ms_per_step'] + '  '.join('%s %.4f ms x %d = %.2f TF' % (k, v['avg_ms'], v['launches'], v['tflops']) for k, v in ks.items() if '$K' in k))"
  done
done
