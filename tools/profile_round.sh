#!/bin/bash
# usage: tools/profile_round.sh <tag>   (on the GPU box)  ->  gpurun_out/<tag>_*: kernel statistics of the fp32 / bf16x3 forward and the fp32
# training step, the two HBM-traffic PMC passes (separate runs) and the MFMA-utilisation pass; copy what is to be judged into profiles/.
tag=${1:-r03}
TOPN=5 bash tools/prof_train.sh ${tag}_f32_infer --no-cpu-baseline --steps 10 --warmup 3 | tail -8
TOPN=5 bash tools/prof_train.sh ${tag}_bf16x3_infer --no-cpu-baseline --dtype bf16x3 --steps 10 --warmup 3 | tail -8
TOPN=5 bash tools/prof_train.sh ${tag}_train --no-cpu-baseline --mode train --steps 8 --warmup 2 | tail -8
bash tools/pmc_hbm.sh ${tag} | tail -14
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_mfma
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d /tmp/pmc_mfma -o p -- python /root/repo/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /root/repo/gpurun_out/pmc_mfma.log 2>&1
cd /root/repo
python tools/pmc_mfma_util.py $(find /tmp/pmc_mfma -name "*counter_collection.csv" | head -1) gpurun_out/${tag}_pmc_mfma_util.txt
head -14 gpurun_out/${tag}_pmc_mfma_util.txt | cut -c1-150
