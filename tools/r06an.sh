cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
python -m pytest tests/test_bf16_gpu.py -x -q -k "patchgan_output or final_layer" > gpurun_out/r06an_tests.txt 2>&1
bash tools/train_prof.sh bf16 r06an_train_bf16 > gpurun_out/r06an.log 2>&1
