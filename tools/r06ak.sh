cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py -x -q -k "discriminator or patchgan or conv" > gpurun_out/r06ak_tests.txt 2>&1
bash tools/train_prof.sh bf16 r06ak_train_bf16 > gpurun_out/r06ak.log 2>&1
