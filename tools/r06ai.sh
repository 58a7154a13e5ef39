cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
for i in 1 2 3; do
python tools/conv_bench.py 30 "final 64->1" 2>&1 | tail -1
APAMD_LIB=$PWD/abl/libapamd_direct_wpe2.so python tools/conv_bench.py 30 "final 64->1" 2>&1 | tail -1
done > gpurun_out/r06ai_direct.txt 2>&1
