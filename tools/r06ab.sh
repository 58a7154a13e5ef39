cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
python -m pytest tests/test_bf16_gpu.py -x -q -k "patchgan_first" > gpurun_out/r06ab_tests.txt 2>&1
python -m pytest "tests/test_train_gpu.py::test_train_step_at_the_reported_size_equals_mean_of_single_sample_steps[bf16]" -x -q -s > gpurun_out/r06ab_on.txt 2>&1
APAMD_NO_D0_MFMA=1 python -m pytest "tests/test_train_gpu.py::test_train_step_at_the_reported_size_equals_mean_of_single_sample_steps[bf16]" -x -q -s > gpurun_out/r06ab_off.txt 2>&1
