# Ablations of the dominant 3x3 kernel with the experiment library (make -C animateportrait_amd/csrc ablate):
#   bits: 1 no DMA refill, 2 no stage barrier, 8 no epilogue.   usage: bash tools/ablate_conv.sh [bf16x3|bf16]
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
export APAMD_LIB=$ROOT/animateportrait_amd/libapamd_ablate.so APAMD_PRECISION=${1:-bf16x3}
for a in 0 1 2 8 9 11; do echo "ABLATE=$a"; APAMD_ABLATE=$a python tools/conv_bench.py 20 "res 256->256 k3 @64 (again" 2>&1 | tail -1; done
for b in 64 128 256 512; do echo "BLOCKS=$b"; APAMD_BF3_BLOCKS=$b python tools/conv_bench.py 20 "res 256->256 k3 @64 (again" 2>&1 | tail -1; done
