#!/bin/bash
# same-box A/B of the large-batch residual kernel forms (RTFS_VARIANTS=resid:<n>: 2 = one 4-wave workgroup per CU, 3 = 8 waves, MFMA + memory roles) and
# a kernel trace of the default forward.  Runs on the GPU box; writes gpurun_out/<tag>_*.
tag=${1:-ab}
for v in 2 3 2 3; do
  RTFS_VARIANTS=resid:$v python bench.py --no-cpu-baseline --steps 30 --warmup 5 > gpurun_out/${tag}_resid_v$v.json 2>/dev/null
  python -c "import json,sys; r=json.loads(open('gpurun_out/${tag}_resid_v$v.json').read().strip().splitlines()[-1]); print('variant $v', round(r['ms_per_step'],3), round(r['ms_per_step_median'],3), round(r['roofline']['avg_launch_ms']*1e3,1), round(r['roofline']['frac'],4))"
done
for v in 2 3; do
  RTFS_VARIANTS=resid:$v python bench.py --no-cpu-baseline --dtype bf16x3 --steps 30 --warmup 5 > gpurun_out/${tag}_resid_bf16x3_v$v.json 2>/dev/null
  python -c "import json,sys; r=json.loads(open('gpurun_out/${tag}_resid_bf16x3_v$v.json').read().strip().splitlines()[-1]); print('bf16x3 variant $v', round(r['ms_per_step'],3), round(r['ms_per_step_median'],3), round(r['roofline']['avg_launch_ms']*1e3,1))"
done
TOPN=45 bash tools/prof_train.sh ${tag}_f32_infer --no-cpu-baseline --steps 10 --warmup 3
