#!/bin/bash
# Round-4 evidence beside tools/collect_profiles.sh:  tools/collect_extra.sh <tag>  -> gpurun_out/<tag>_*.txt
#   conv_bench tables (forward / data gradient / weight gradient, three and two pieces), SQ counters of the conv kernels,
#   per-launch-shape times of a step, the weight gradient's distance from fp64 over the split-K count, and the timing
#   experiments of the weight gradient (separate exp_* libraries: wrong results or other trade-offs, never the product).
# rocprofv3 summaries of the companion arithmetics (the same step on the exact fp32-MFMA kernels / the two-piece split):
#   bash tools/collect_extra.sh companions <tag>
if [ "$1" = "companions" ]; then
  TAG=${2:-prof}; export TMPDIR=/tmp; R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; O=$R/gpurun_out; cd /tmp
  for M in fp32 bf16x3; do
    rocprofv3 --kernel-trace --stats -d /tmp/p_$M --output-format csv -- python3 $R/bench.py --conv-mode $M --serial-streams --steps 5 --warmup 2 --no-cpu-baseline --no-pointwise --no-companions > $O/${TAG}_prof_$M.log 2>&1
    cp $(find /tmp/p_$M -name "*kernel_stats.csv" | head -1) $O/${TAG}_bench_${M}_kernel_stats.csv; rm -rf /tmp/p_$M
    tail -1 $O/${TAG}_prof_$M.log | cut -c1-300
  done
  exit 0
fi
TAG=${1:-r04}
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
export PYTHONPATH=$R
{
  echo "# tools/conv_bench.py, 4 frames of 1024x2048 (paired step); us per launch (three rounds) -> fp32-equivalent TFLOP/s"
  echo "# ceilings: bf16x6 416.7, bf16x3 833.3 TFLOP/s"
  for m in bf16x6 bf16x3; do for op in fwd dgrad wgrad; do
    timeout 300 python3 tools/conv_bench.py --mode $m --op $op --variants conv_bn256=1 2>&1 | grep -v amdgpu.ids
  done; done
} > $O/${TAG}_conv_bench.txt
{
  echo "# tools/sq_counters.sh bf16x6 / bf16x3: one --pmc pass per op over tools/conv_bench.py's shapes"
  echo "# mfma_busy assumes 2 resident waves per SIMD; the weight-gradient kernel runs 3 (x 1.5)"
  timeout 300 bash tools/sq_counters.sh bf16x6 2>&1 | grep -v amdgpu.ids
  timeout 300 bash tools/sq_counters.sh bf16x3 2>&1 | grep -v amdgpu.ids
} > $O/${TAG}_sq_counters.txt
timeout 300 python3 tools/layer_times.py 2>&1 | grep -v amdgpu.ids > $O/${TAG}_layer_times.txt
{
  echo "# weight gradient, conv3_2 (4 x 256x512 px, 256 -> 256): distance from an fp64 weight gradient (max, rms, signed bias;"
  echo "# relative to the largest element) over the split-K count (scan_tune wgrad_wgs) -- tools/wgrad_err.py"
  echo "## shipped library (temporary accumulator per 32-pixel step)"
  timeout 200 python3 tools/wgrad_err.py 2>&1 | grep -v amdgpu.ids
  if [ -f scan_amd/libscan_hip_exp_notchain.so ]; then
    echo "## make exp_wgrad_notchain (six piece products added to the running accumulator one by one)"
    SCAN_HIP_LIB=scan_amd/libscan_hip_exp_notchain.so timeout 200 python3 tools/wgrad_err.py 2>&1 | grep -v amdgpu.ids
  fi
} > $O/${TAG}_wgrad_error.txt
{
  echo "# weight-gradient timing experiments (tools/conv_bench.py --op wgrad, bf16x6): shipped library, then the exp_* builds"
  for L in "" scan_amd/libscan_hip_exp_notchain.so scan_amd/libscan_hip_exp_tg1.so scan_amd/libscan_hip_exp_nosplit.so; do
    [ -z "$L" ] || [ -f "$L" ] || continue
    echo "## library: ${L:-scan_amd/libscan_hip.so}"
    SCAN_HIP_LIB=$L timeout 250 python3 tools/conv_bench.py --op wgrad --shapes conv3,conv4,towers,dis --variants wgrad_tile=1,wgrad_tile=0 2>&1 | grep -v "amdgpu.ids"
  done
  echo "# exp_wgrad_nosplit: the producers store raw bits instead of converting (WRONG results): what the fp32 -> 3 x bf16 split costs"
  echo "# exp_wgrad_notchain: no temporary accumulator;  exp_wgrad_tg1: temporary over one column tile per block (150 registers, no scratch)"
} > $O/${TAG}_wgrad_exp.txt
ls -la $O/${TAG}_*.txt
