cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
python -m pytest tests -x -q -m gpu > gpurun_out/r06t_tests.txt 2>&1
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r06t_smoke.txt 2>&1
python bench.py > gpurun_out/r06t_bench.json 2> gpurun_out/r06t_bench.err
