# Same-box A/B of two builds of libapamd.so on the conv micro-benchmark, interleaved rounds:
#   bash tools/ab_conv.sh <libA> <libB> [rounds] [layer substring]
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
A=$1; B=$2; R=${3:-3}; L=${4:-}
for r in $(seq 1 $R); do
  for lib in $A $B; do
    echo "== round $r lib $lib"
    APAMD_LIB=$ROOT/$lib python tools/conv_bench.py 20 $L 2>&1 | grep -v amdgpu.ids
  done
done
