cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
python -m pytest tests/test_bf16_gpu.py -x -q -k "k7 or patchgan or stem_gradient or final_layer" > gpurun_out/r06ac_tests.txt 2>&1
python tools/k7_bench.py 20 > gpurun_out/r06ac_k7_bench.txt 2>&1
python -m pytest tests/test_train_gpu.py tests/test_bf16_gpu.py -x -q > gpurun_out/r06ac_tests2.txt 2>&1
for r in 1 2 3; do
  for k in 1 0; do
    APAMD_PRECISION=bf16 APAMD_NO_K7_WGRAD=$k python tools/train_bench.py 16 5 > /tmp/ab_train.log 2>&1
    echo "round $r APAMD_NO_K7_WGRAD=$k $(grep 'train step' /tmp/ab_train.log)"
  done
done > gpurun_out/r06ac_ab_train.txt 2>&1
