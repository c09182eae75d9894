#!/bin/bash
# usage: tools/pmc_hbm.sh <tag>   -- the two separate --pmc passes (FETCH_SIZE, WRITE_SIZE) over the default bench command line, then
# tools/pmc_traffic.py -> gpurun_out/<tag>_pmc_hbm_traffic.txt + gpurun_out/pmc_traffic.json (copy both into profiles/)
tag=$1
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  timeout ${PMC_TIMEOUT:-300} rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_$c -o p -- python /root/repo/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /root/repo/gpurun_out/pmc_$c.log 2>&1
done
cd /root/repo
python tools/pmc_traffic.py $(find /tmp/pmc_FETCH_SIZE -name "*counter_collection.csv" | head -1) $(find /tmp/pmc_WRITE_SIZE -name "*counter_collection.csv" | head -1) gpurun_out/${tag}_pmc_hbm_traffic.txt
cp gpurun_out/${tag}_pmc_hbm_traffic.json gpurun_out/pmc_traffic.json  # (bench.py reads the roofline kernel's `traffic` from profiles/pmc_traffic.json)
head -12 gpurun_out/${tag}_pmc_hbm_traffic.txt | cut -c1-60,100-150
