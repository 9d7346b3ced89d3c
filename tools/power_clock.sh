# samples socket power, sclk and junction temperature with rocm-smi while bench.py runs 300 steps: bash tools/power_clock.sh > gpurun_out/power_clock.txt
python bench.py --steps 300 --warmup 5 --no-cpu-baseline --no-pointwise --no-companions > gpurun_out/pw_bench.txt 2>/dev/null &
BP=$!
sleep 18
for i in $(seq 12); do rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -E "Power|sclk|mclk|Temperature \(Sensor (junction|edge)" | tr "\n" " "; echo; sleep 0.4; done
wait $BP
tail -1 gpurun_out/pw_bench.txt | cut -c1-200
rocm-smi --showmaxpower 2>/dev/null | grep -i power
