cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
export APAMD_LIB=$PWD/animateportrait_amd/libapamd_ablate.so APAMD_PRECISION=bf16
python tools/cycle_account.py res gpurun_out/r06v_cycle_bf16 > gpurun_out/r06v_cycle_bf16.txt 2>&1
for a in 0 1 2 8 16 9 11; do echo "ABLATE=$a"; APAMD_ABLATE=$a python tools/conv_bench.py 20 "res 256->256 k3 @64 (again" 2>&1 | tail -2; done > gpurun_out/r06v_ablate_bf16.txt 2>&1
for b in 128 256 384 512; do echo "BLOCKS=$b"; APAMD_BF3_BLOCKS=$b python tools/conv_bench.py 20 "res 256->256 k3 @64 (again" 2>&1 | tail -2; done >> gpurun_out/r06v_ablate_bf16.txt 2>&1
