cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
python -m pytest tests/test_bf16_gpu.py -x -q -k "k7 or patchgan or stem_gradient or final_layer" > gpurun_out/r06ah_tests.txt 2>&1
