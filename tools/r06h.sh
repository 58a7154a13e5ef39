cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py tests/test_bf16_gpu.py -x -q -m gpu > gpurun_out/r06h_tests.txt 2>&1
for i in 1 2 3; do APAMD_LIB=$PWD/abl/libapamd_r05.so python tools/gen_time.py; python tools/gen_time.py; done 2>&1 | grep frames > gpurun_out/r06h_gen.txt
APAMD_LIB=$PWD/animateportrait_amd/libapamd_ablate.so python tools/cycle_account.py res gpurun_out/r06h_epi_fast > gpurun_out/r06h_epi_fast.txt 2>&1
bash tools/ab_train.sh abl/libapamd_r05.so animateportrait_amd/libapamd.so 3 bf16 > gpurun_out/r06h_ab_train.txt 2>&1
bash tools/ab_train.sh abl/libapamd_r05.so animateportrait_amd/libapamd.so 2 bf16x3 > gpurun_out/r06h_ab_train_x3.txt 2>&1
