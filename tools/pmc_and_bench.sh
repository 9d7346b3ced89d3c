export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; export SCAN_COMMIT=$1
cd /tmp
P="--steps 1 --warmup 1 --no-cpu-baseline --no-pointwise --no-companions"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pmcF_$$ -- python3 $R/bench.py $P > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/pmcW_$$ -- python3 $R/bench.py $P > /dev/null 2>&1
python3 $R/tools/pmc_summarize.py $(find /tmp/pmcF_$$ -name "*counter_collection.csv*" | head -1) $(find /tmp/pmcW_$$ -name "*counter_collection.csv*" | head -1) > $O/r03_pmc_traffic.json 2> $O/r03_pmc.err
mkdir -p $R/profiles; cp $O/r03_pmc_traffic.json $R/profiles/r03_pmc_traffic.json
cd $R; python3 bench.py > $O/r03_bench_line.json 2> $O/r03_bench.err
tail -1 $O/r03_bench_line.json | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline']['traffic'], d['roofline'].get('traffic_provenance'))"
