#!/bin/bash
# usage: tools/pmc_pass.sh <tag> "<counter list>" [kernel substrings...]  -- one rocprofv3 --pmc pass over a short inference bench
tag=$1; ctrs=$2; shift 2
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_$tag
timeout ${PMC_TIMEOUT:-240} rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d /tmp/pmc_$tag -o p -- python /root/repo/bench.py --steps 2 --warmup 1 --no-cpu-baseline ${BENCH_ARGS} > /root/repo/gpurun_out/pmc_$tag.log 2>&1
cd /root/repo
f=$(find /tmp/pmc_$tag -name "*counter_collection.csv" | head -1)
python tools/pmc_counters.py $f "$@" > gpurun_out/pmc_$tag.txt
cat gpurun_out/pmc_$tag.txt
