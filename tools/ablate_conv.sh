cd $GRAFT_REPO_ROOT
for a in 0 1 2 3 8 9 10 11; do echo "ABLATE=$a"; APAMD_ABLATE=$a python tools/conv_bench.py 20 "res 256->256 k3 @64 (again" 2>&1 | tail -1; done
for b in 64 128 192 256; do echo "BLOCKS=$b"; APAMD_BF3_BLOCKS=$b python tools/conv_bench.py 20 "res 256->256 k3 @64 (again" 2>&1 | tail -1; done
echo NOSTATS; APAMD_BENCH_NOSTATS=1 python tools/conv_bench.py 20 "res 256->256 k3 @64 (again" 2>&1 | tail -1
