cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py tests/test_train_gpu.py -x -q > gpurun_out/r06ae_tests.txt 2>&1
bash tools/train_prof.sh bf16 r06ae_train_bf16 > gpurun_out/r06ae.log 2>&1
for r in 1 2 3; do
  APAMD_PRECISION=bf16 python tools/train_bench.py 16 5 > /tmp/ab_train.log 2>&1
  echo "round $r $(grep 'train step' /tmp/ab_train.log)"
done > gpurun_out/r06ae_train.txt 2>&1
