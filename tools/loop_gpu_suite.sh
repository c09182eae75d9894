#!/bin/bash
# usage: tools/loop_gpu_suite.sh <n> [pytest selection...]  -- runs the GPU suite n times on the GPU box and keeps the report of every failing run under gpurun_out/
cd /root/repo
n=$1; shift
fails=0
for i in $(seq 1 $n); do
  python -m pytest ${@:-tests} -q -m gpu -x > /tmp/suite_$i.log 2>&1
  if grep -aq "failed" /tmp/suite_$i.log; then fails=$((fails+1)); cp /tmp/suite_$i.log gpurun_out/suite_fail_$i.log; fi
  grep -a "passed\|failed" /tmp/suite_$i.log | tail -1
done
echo "$fails failing runs of $n"
