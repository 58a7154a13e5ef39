cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py tests/test_stream_gpu.py -x -q -m gpu > gpurun_out/r06r_tests.txt 2>&1
for i in 1 2 3; do APAMD_NO_OCTET_TRUNK=1 python tools/gen_time.py; python tools/gen_time.py; done 2>&1 | grep frames > gpurun_out/r06r_gen.txt
bash tools/gen_prof.sh r06r_gen > gpurun_out/r06r_gen_prof.log 2>&1
