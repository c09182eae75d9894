#!/bin/bash
# usage: [PROF_SCRIPT=tools/lip_bench.py] tools/prof_train.sh <tag> [script args...]   (runs on the GPU box; writes gpurun_out/<tag>_kernel_stats.txt + <tag>.log)
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$tag
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_$tag -o tr -- python /root/repo/${PROF_SCRIPT:-bench.py} "$@" > /root/repo/gpurun_out/$tag.log 2>&1
cd /root/repo
grep -a '"metric"' gpurun_out/$tag.log | cut -c1-330
python tools/rocprof_summary.py $(find /tmp/prof_$tag -name "*.db" | head -1) gpurun_out/${tag}_kernel_stats.txt
head -${TOPN:-40} gpurun_out/${tag}_kernel_stats.txt | cut -c1-70,120-170
tail -1 gpurun_out/${tag}_kernel_stats.txt
