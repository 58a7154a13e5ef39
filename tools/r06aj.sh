cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
python -m pytest tests/test_bf16_gpu.py -x -q -k "k7 or patchgan or stem_gradient or final_layer" > gpurun_out/r06aj_tests.txt 2>&1
python tools/k7_bench.py 20 > gpurun_out/r06aj_k7_bench.txt 2>&1
