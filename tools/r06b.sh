cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
( cd tools/hazard/repro && make > /dev/null 2>&1 && timeout 900 ./cohazard.bin 200 60 ) > gpurun_out/r06b_cohazard_repro.txt 2>&1
echo "repro exit $?" >> gpurun_out/r06b_cohazard_repro.txt
( hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -o tools/hazard/libhazard.so tools/hazard/hazard_kernels.hip 2>/dev/null; APAMD_NO_STREAM_FENCE=1 VICTIMS=warp AGGRESSORS=conv3x3,synth:11,synth:0 timeout 600 python tools/hazard/run_hazard.py ) > gpurun_out/r06b_lab.txt 2>&1
python tools/ob16_bench.py 100 > gpurun_out/r06b_ob16.txt 2>&1
python -m pytest tests/test_bf16_gpu.py tests/test_inbwd_split_gpu.py -x -q -m gpu > gpurun_out/r06b_tests.txt 2>&1
APAMD_PRECISION=bf16 python tools/train_bench.py 16 5 > gpurun_out/r06b_train_bf16.log 2>&1
