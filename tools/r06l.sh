cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_module1_gpu.py tests/test_stream_gpu.py -x -q -m gpu > gpurun_out/r06l_tests.txt 2>&1
echo "rc $?" >> gpurun_out/r06l_tests.txt
for b in 16 32; do timeout 600 python bench.py --stream --stream-batch $b > gpurun_out/r06l_stream_b$b.json 2> gpurun_out/r06l_stream_b$b.err; done
