#!/bin/bash
# usage: tools/pmc_gemm.sh <tag> "<counters>" [gemm_bench args]  -- one rocprofv3 --pmc pass over the layer-0 GEMM micro-benchmark
tag=$1; ctrs=$2; shift 2
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmcg_$tag
timeout 300 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d /tmp/pmcg_$tag -o p -- python /root/repo/tools/gemm_bench.py "$@" > /root/repo/gpurun_out/pmcg_$tag.log 2>&1
cd /root/repo
python tools/pmc_counters.py $(find /tmp/pmcg_$tag -name "*counter_collection.csv" | head -1) unfold_gemm
