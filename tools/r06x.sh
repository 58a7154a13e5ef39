cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
for r in 1 2 3; do
  for k in 1 0; do
    APAMD_PRECISION=bf16 APAMD_NO_K7_WGRAD=$k python tools/train_bench.py 16 5 > /tmp/ab_train.log 2>&1
    echo "round $r APAMD_NO_K7_WGRAD=$k $(grep 'train step' /tmp/ab_train.log)"
    [ $r = 1 ] && grep "^{" /tmp/ab_train.log
  done
done > gpurun_out/r06x_ab_train.txt 2>&1
python -m pytest tests -x -q -m gpu > gpurun_out/r06x_tests.txt 2>&1
