cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
python -m pytest tests/test_inbwd_split_gpu.py tests/test_bf16_gpu.py tests/test_gpu_parity.py tests/test_train_gpu.py -x -q -m gpu > gpurun_out/r06e_tests.txt 2>&1
bash tools/ab_train.sh abl/libapamd_r05.so animateportrait_amd/libapamd.so 3 bf16 > gpurun_out/r06e_ab_train.txt 2>&1
for lib in abl/libapamd_r05.so animateportrait_amd/libapamd.so abl/libapamd_r05.so animateportrait_amd/libapamd.so; do echo "== $lib"; APAMD_LIB=$PWD/$lib python tools/ob16_bench.py 100 2>&1 | grep -v amdgpu.ids; done > gpurun_out/r06e_ob16.txt 2>&1
HEAD=70 bash tools/train_prof.sh bf16 r06e_train_bf16 > gpurun_out/r06e_train_prof.txt 2>&1
for i in 1 2; do APAMD_LIB=$PWD/abl/libapamd_r05.so python tools/gen_time.py; python tools/gen_time.py; done 2>&1 | grep frames > gpurun_out/r06e_gen.txt
