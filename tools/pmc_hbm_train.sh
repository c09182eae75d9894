#!/bin/bash
# usage: tools/pmc_hbm_train.sh <tag>   -- tools/pmc_hbm.sh for the TRAINING step (bench.py --mode train, 1 warm-up + 1 timed step per pass)
tag=$1
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmct_$c
  timeout ${PMC_TIMEOUT:-400} rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmct_$c -o p -- python /root/repo/bench.py --mode train --steps 1 --warmup 1 --no-cpu-baseline > /root/repo/gpurun_out/pmct_$c.log 2>&1
done
cd /root/repo
python tools/pmc_traffic.py $(find /tmp/pmct_FETCH_SIZE -name "*counter_collection.csv" | head -1) $(find /tmp/pmct_WRITE_SIZE -name "*counter_collection.csv" | head -1) gpurun_out/${tag}_train_pmc_hbm_traffic.txt
head -40 gpurun_out/${tag}_train_pmc_hbm_traffic.txt | cut -c1-70,100-150
