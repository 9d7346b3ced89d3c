#!/bin/bash
# SQ counters of the conv kernels over tools/conv_bench.py's layer shapes (one --pmc pass per op, counters only):
#   tools/sq_counters.sh [mode]     mode = bf16x6 (default) | bf16x3
# (eight counters: one more and rocprofv3 needs a second pass per kernel, which takes > 15 min on these shapes)
# SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_* count quad-cycles per wave, SQ_VALU_MFMA_BUSY_CYCLES cycles per SIMD.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
M=${1:-bf16x6}
cd /tmp
C="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS"
rm -rf /tmp/sqf /tmp/sqd /tmp/sqw
rocprofv3 --pmc $C --output-format csv -d /tmp/sqf -- python3 $R/tools/conv_bench.py --mode $M --op fwd --rounds 1 --reps 2 --variants conv_bn256=1 > /dev/null 2>&1
rocprofv3 --pmc $C --output-format csv -d /tmp/sqw -- python3 $R/tools/conv_bench.py --mode $M --op wgrad --rounds 1 --reps 2 --variants wgrad_v6=1 > /dev/null 2>&1
for d in /tmp/sqf /tmp/sqw; do python3 $R/tools/pmc_counters.py $(find $d -name "*counter_collection.csv*" | head -1) conv_split conv_wgrad waves_per_simd=2; done
