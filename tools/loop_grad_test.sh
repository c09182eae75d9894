cd /root/repo
for mode in fused three; do
  if [ $mode = three ]; then export RTFS_DISABLE=srubwd; else unset RTFS_DISABLE; fi
  fails=0
  for i in $(seq 1 25); do
    out=$(python -m pytest tests/test_hip_backward.py -q -x -k "test_parameter_gradients and not split" 2>&1 | tail -30)
    if echo "$out" | grep -q "failed"; then fails=$((fails+1)); echo "$out" | grep -a "AssertionError\|worst tensor\|assert not bad" | head -5; fi
  done
  echo "mode $mode: $fails failures of 25"
done
