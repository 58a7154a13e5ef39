cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
python -m pytest tests/test_module1_gpu.py tests/test_stream_gpu.py -x -q -m gpu > gpurun_out/r06j_tests.txt 2>&1
for b in 16 32 48; do python bench.py --stream --stream-batch $b > gpurun_out/r06j_stream_b$b.json 2> gpurun_out/r06j_stream_b$b.err; done
python -m pytest tests/test_gpu_parity.py tests/test_train_gpu.py -x -q -m gpu > gpurun_out/r06j_tests2.txt 2>&1
