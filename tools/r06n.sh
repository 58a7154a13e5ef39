cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
( time python bench.py ) > gpurun_out/r06n_bench.json 2> gpurun_out/r06n_bench.err
