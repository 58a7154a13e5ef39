cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
for i in 1 2 3; do APAMD_LIB=$PWD/abl/libapamd_r05.so APAMD_NO_OCTET_TRUNK=1 python tools/gen_time.py; python tools/gen_time.py; done 2>&1 | grep frames > gpurun_out/r06u_gen.txt
bash tools/ab_train.sh abl/libapamd_r05.so animateportrait_amd/libapamd.so 3 bf16 > gpurun_out/r06u_ab_train.txt 2>&1
python -m pytest tests -x -q -m gpu > gpurun_out/r06u_tests.txt 2>&1
