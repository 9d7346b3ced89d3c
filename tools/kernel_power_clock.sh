# socket power and sclk while ONE conv kernel runs back to back (tools/conv_bench.py on one shape), per library build:
#   bash tools/kernel_power_clock.sh wgrad "conv3_x" "" noprod prod9 prod10 > gpurun_out/kernel_power_clock.txt
# ("" = the shipped library; names select scan_amd/libscan_hip_exp_<name>.so)
OP=$1; SHAPE=$2; shift 2
python -c "import torch" 2>/dev/null  # page the image in once
for L in "$@"; do
  if [ -n "$L" ]; then export SCAN_HIP_LIB=scan_amd/libscan_hip_exp_$L.so; else unset SCAN_HIP_LIB; fi
  echo "## lib: ${L:-shipped}"
  python tools/conv_bench.py --op $OP --mode bf16x6 --shapes "$SHAPE" --variants wgrad_v6=1 --reps 5000 --rounds 2 > gpurun_out/kpc_bench.txt 2>/dev/null &
  BP=$!
  sleep 16
  for i in $(seq 8); do rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk" | sed -e 's/.*: //' | tr "\n" " "; echo; sleep 0.4; done
  wait $BP
  grep -v "^op " gpurun_out/kpc_bench.txt
done
