cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py tests/test_inbwd_split_gpu.py -x -q -k "instnorm or inbwd or backward or grad" > gpurun_out/r06aq_tests.txt 2>&1
python -m pytest tests/test_train_gpu.py -x -q > gpurun_out/r06aq_tests2.txt 2>&1
bash tools/train_prof.sh bf16 r06aq_train_bf16 > gpurun_out/r06aq.log 2>&1
