cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py tests/test_train_gpu.py tests/test_bf16_gpu.py -x -q -m gpu > gpurun_out/r06f_tests.txt 2>&1
bash tools/ab_train.sh abl/libapamd_r05.so animateportrait_amd/libapamd.so 3 bf16 > gpurun_out/r06f_ab_train.txt 2>&1
bash tools/ab_prof.sh abl/libapamd_r05.so animateportrait_amd/libapamd.so r06f bf16 > gpurun_out/r06f_ab_prof.txt 2>&1
