# sustained bf16 MFMA rate, clock and socket power of this board (register-only MFMA loop, tools/mfma_peak.hip):
#   bash tools/mfma_peak.sh > gpurun_out/mfma_peak.txt                      (default series)
#   bash tools/mfma_peak.sh "2 2 0" "2 3 0"                                 (configs: "<waves per SIMD> <data mode> <shape>")
hipcc --offload-arch=gfx950 -O3 -o gpurun_out/mfma_peak tools/mfma_peak.hip || exit 1
if [ $# -eq 0 ]; then set -- "2 1 0" "2 1 1" "4 1 0" "2 0 0"; fi
for cfg in "$@"; do
  gpurun_out/mfma_peak 6 $cfg > gpurun_out/mfma_peak_run.txt &
  BP=$!
  sleep 3
  for i in $(seq 5); do rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk" | sed -e 's/.*: //' -e 's/=//g' | tr "\n" " "; echo; sleep 0.4; done
  wait $BP
  cat gpurun_out/mfma_peak_run.txt
done
rocm-smi --showmaxpower 2>/dev/null | grep -i power
