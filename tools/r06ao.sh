cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py -x -q -k "act_bwd_with_the_bias" > gpurun_out/r06ao_tests.txt 2>&1
python -m pytest tests/test_train_gpu.py -x -q > gpurun_out/r06ao_tests2.txt 2>&1
bash tools/train_prof.sh bf16 r06ao_train_bf16 > gpurun_out/r06ao.log 2>&1
