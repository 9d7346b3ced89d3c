# sustained bf16 MFMA rate, clock and socket power of this board (register-only MFMA loop): bash tools/mfma_peak.sh > gpurun_out/mfma_peak.txt
hipcc --offload-arch=gfx950 -O3 -o gpurun_out/mfma_peak tools/mfma_peak.hip || exit 1
for cfg in "2 1 0" "2 1 1" "4 1 0" "2 0 0"; do
  set -- $cfg
  gpurun_out/mfma_peak 6 $1 $2 $3 > gpurun_out/mfma_peak_run.txt &
  BP=$!
  sleep 3
  for i in $(seq 5); do rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk" | sed -e 's/.*: //' -e 's/=//g' | tr "\n" " "; echo; sleep 0.4; done
  wait $BP
  cat gpurun_out/mfma_peak_run.txt
done
rocm-smi --showmaxpower 2>/dev/null | grep -i power
