export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
python3 bench.py 2> gpurun_out/final_bench.err | tail -1 > gpurun_out/final_bench.json
python3 bench.py --model s2c --no-cpu-baseline --no-pointwise 2>/dev/null | tail -1 > gpurun_out/final_bench_s2c.json
python3 bench.py --model k2c_r50 --no-cpu-baseline --no-pointwise 2>/dev/null | tail -1 > gpurun_out/final_bench_k2c_r50.json
python3 bench.py --height 1333 --width 2666 --batch 8 --steps 3 --warmup 1 --no-cpu-baseline --no-pointwise 2>/dev/null | tail -1 > gpurun_out/final_bench_cfg5.json
python3 bench.py --scaling strong --global-batch 8 --steps 3 --warmup 1 --no-cpu-baseline --no-pointwise 2>/dev/null | tail -1 > gpurun_out/final_bench_strong8.json
python3 bench.py --forward-target --no-cpu-baseline --no-pointwise 2>/dev/null | tail -1 > gpurun_out/final_bench_ft.json
for f in gpurun_out/final_bench*.json; do echo -n "$f "; grep -o "ms_per_step\": [0-9.]*\|\"value\": [0-9.]*\|\"frac\": [0-9.]*" $f | head -3 | tr "\n" " "; echo; done
