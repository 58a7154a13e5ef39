# Same-box interleaved A/B of two builds of libapamd.so on the train step:  bash tools/ab_train.sh <libA> <libB> [rounds] [precision]
# (prints ms per step and the loss dictionary after the same number of steps: the two builds must agree on it)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
A=$1; B=$2; R=${3:-3}; P=${4:-bf16}
for r in $(seq 1 $R); do
  for lib in $A $B; do
    APAMD_PRECISION=$P APAMD_LIB=$ROOT/$lib python tools/train_bench.py 16 5 > /tmp/ab_train.log 2>&1
    echo "round $r $lib $(grep 'train step' /tmp/ab_train.log)"
    [ $r = 1 ] && grep "^{" /tmp/ab_train.log
  done
done
