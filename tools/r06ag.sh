cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py -x -q -k "generator or final or direct or conv" > gpurun_out/r06ag_tests.txt 2>&1
bash tools/gen_prof.sh r06ag_gen > gpurun_out/r06ag_gen_prof.log 2>&1
for i in 1 2 3; do python tools/gen_time.py; done 2>&1 | grep frames > gpurun_out/r06ag_gen.txt
