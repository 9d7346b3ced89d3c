# SQ counters of the conv kernels over tools/conv_bench.py's layer shapes (one --pmc pass per op, counters only)
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp
C="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS"
rocprofv3 --pmc $C --output-format csv -d /tmp/sqf -- python3 $R/tools/conv_bench.py --op fwd --rounds 1 --variants conv_w8=1,conv_w8=0 > /dev/null 2>&1
rocprofv3 --pmc $C --output-format csv -d /tmp/sqw -- python3 $R/tools/conv_bench.py --op wgrad --rounds 1 --variants wgrad_v6=1 > /dev/null 2>&1
for d in /tmp/sqf /tmp/sqw; do python3 $R/tools/pmc_counters.py $(find $d -name "*counter_collection.csv*" | head -1) conv; done
