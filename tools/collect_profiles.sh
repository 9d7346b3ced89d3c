#!/bin/bash
# Collects the rocprofv3 evidence bench.py's figures are checked against, on a GPU box:
#   tools/collect_profiles.sh <tag> [commit]      -> gpurun_out/<tag>_*.{csv,json,log}
# kernel-trace + stats of the default (overlapped) run and of the serial-stream run (the per-kernel roofline figures),
# idle / exposed-small-kernel time per step, the pointwise tool's stats, and the two PMC passes (FETCH_SIZE, WRITE_SIZE:
# separate runs, counters only) summarised per kernel.  Copy what is to be judged into profiles/.
TAG=${1:-prof}
export SCAN_COMMIT=${2:-}
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out
mkdir -p $O
cd /tmp
B="--steps 10 --warmup 3 --no-cpu-baseline --no-pointwise --no-companions"
rocprofv3 --kernel-trace --stats -d /tmp/p_ov_$$ --output-format csv -- python3 $R/bench.py $B > $O/${TAG}_prof_overlap.log 2>&1
python3 $R/tools/gpu_idle.py $(find /tmp/p_ov_$$ -name "*kernel_trace.csv" | head -1) > $O/${TAG}_gpu_idle.json 2>/dev/null
cp $(find /tmp/p_ov_$$ -name "*kernel_stats.csv" | head -1) $O/${TAG}_bench_overlap_kernel_stats.csv
rm -rf /tmp/p_ov_$$
rocprofv3 --kernel-trace --stats -d /tmp/p_se_$$ --output-format csv -- python3 $R/bench.py --serial-streams $B > $O/${TAG}_prof_serial.log 2>&1
cp $(find /tmp/p_se_$$ -name "*kernel_stats.csv" | head -1) $O/${TAG}_bench_serial_kernel_stats.csv
rm -rf /tmp/p_se_$$
if [ "$3" != "quick" ]; then
rocprofv3 --kernel-trace --stats -d /tmp/p_pw_$$ --output-format csv -- python3 $R/tools/pointwise_roofline.py --out $O/${TAG}_pointwise_roofline.json > $O/${TAG}_pointwise.log 2>&1
cp $(find /tmp/p_pw_$$ -name "*kernel_stats.csv" | head -1) $O/${TAG}_pointwise_kernel_stats.csv
rm -rf /tmp/p_pw_$$
P="--steps 1 --warmup 1 --no-cpu-baseline --no-pointwise --no-companions"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pmcF_$$ -- python3 $R/bench.py $P > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/pmcW_$$ -- python3 $R/bench.py $P > /dev/null 2>&1
python3 $R/tools/pmc_summarize.py $(find /tmp/pmcF_$$ -name "*counter_collection.csv*" | head -1) $(find /tmp/pmcW_$$ -name "*counter_collection.csv*" | head -1) > $O/${TAG}_pmc_traffic.json 2> $O/${TAG}_pmc.err
rm -rf /tmp/pmcF_$$ /tmp/pmcW_$$
fi
ls -la $O/${TAG}_*
