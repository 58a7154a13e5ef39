# Ablations of the bf16 weight-gradient kernel (whole operator incl. operand preparation; compare deltas).
# bits: 1 no DMA refill, 2 no stage barrier / wait, 16 no output.   usage: bash tools/ablate_wgrad.sh [bf16x3|bf16]
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
export APAMD_LIB=$ROOT/animateportrait_amd/libapamd_ablate.so APAMD_PRECISION=${1:-bf16}
for a in 0 1 2 3 16 19; do echo "ABLATE=$a"; APAMD_ABLATE=$a python tools/wgrad_bench.py 10 "res 256" 2>&1 | tail -1; done
for b in 128 256 512 1024; do echo "BLOCKS=$b"; APAMD_WGRAD_BLOCKS=$b python tools/wgrad_bench.py 10 "res 256" 2>&1 | tail -1; done
