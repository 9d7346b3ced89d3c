for r in 1 2; do
for d in 3 2 1; do for f in 1 0; do
  echo -n "[DIS=$d FCOS=$f] "
  env SCAN_DIS_STREAMS=$d SCAN_FCOS_STREAM=$f python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-pointwise --no-companions 2>/dev/null | tail -1 | grep -o '"ms_per_step": [0-9.]*' | tr "\n" " "; echo
done; done; done
