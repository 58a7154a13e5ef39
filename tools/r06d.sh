cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
( cd tools/hazard/repro && make > /dev/null 2>&1 && timeout 900 ./cohazard.bin 200 60 ) > gpurun_out/r06d_cohazard_repro.txt 2>&1
echo "repro exit $?" >> gpurun_out/r06d_cohazard_repro.txt
python -m pytest tests/test_inbwd_split_gpu.py tests/test_bf16_gpu.py tests/test_gpu_parity.py tests/test_train_gpu.py -x -q -m gpu > gpurun_out/r06d_tests.txt 2>&1
bash tools/ab_train.sh abl/libapamd_r05.so animateportrait_amd/libapamd.so 3 bf16 > gpurun_out/r06d_ab_train.txt 2>&1
HEAD=70 bash tools/train_prof.sh bf16 r06d_train_bf16 > gpurun_out/r06d_train_prof.txt 2>&1
