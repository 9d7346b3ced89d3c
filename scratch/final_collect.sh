export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
python3 bench.py 2> gpurun_out/final_bench.err | tail -1 > gpurun_out/final_bench.json
cd /tmp
rocprofv3 --kernel-trace --stats -d /tmp/p_ov_$$ --output-format csv -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-pointwise > $R/gpurun_out/final_prof_overlap.log 2>&1
python3 $R/tools/gpu_idle.py $(find /tmp/p_ov_$$ -name "*kernel_trace.csv" | head -1) > $R/gpurun_out/final_gpu_idle.json 2>/dev/null
cp $(find /tmp/p_ov_$$ -name "*kernel_stats.csv" | head -1) $R/gpurun_out/final_overlap_kernel_stats.csv
rm -rf /tmp/p_ov_$$
rocprofv3 --kernel-trace --stats -d /tmp/p_se_$$ --output-format csv -- python3 $R/bench.py --serial-streams --steps 10 --warmup 3 --no-cpu-baseline --no-pointwise > $R/gpurun_out/final_prof_serial.log 2>&1
cp $(find /tmp/p_se_$$ -name "*kernel_stats.csv" | head -1) $R/gpurun_out/final_serial_kernel_stats.csv
rm -rf /tmp/p_se_$$
rocprofv3 --kernel-trace --stats -d /tmp/p_pw_$$ --output-format csv -- python3 $R/tools/pointwise_roofline.py --out $R/gpurun_out/final_pointwise.json > $R/gpurun_out/final_pointwise.log 2>&1
cp $(find /tmp/p_pw_$$ -name "*kernel_stats.csv" | head -1) $R/gpurun_out/final_pointwise_kernel_stats.csv
rm -rf /tmp/p_pw_$$
rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pmcF_$$ -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-pointwise > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/pmcW_$$ -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-pointwise > /dev/null 2>&1
python3 $R/tools/pmc_summarize.py $(find /tmp/pmcF_$$ -name "*counter_collection.csv*" | head -1) $(find /tmp/pmcW_$$ -name "*counter_collection.csv*" | head -1) > $R/gpurun_out/final_pmc_traffic.json 2> $R/gpurun_out/final_pmc.err
cd $R
for r in 1 2 3; do for b in 0 1; do echo -n "SCAN_BATCHED=$b SCAN_GROUPED_CLS=$b "; SCAN_BATCHED=$b SCAN_GROUPED_CLS=$b python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-pointwise 2>/dev/null | tail -1 | grep -o "ms_per_step\": [0-9.]*\|[0-9.]* ms/step serial" | tr "\n" " "; echo; done; done > gpurun_out/final_ab.txt
ls -la gpurun_out/final_*
