cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
APAMD_PRECISION=bf16 python tools/dispatch_census.py 70 > gpurun_out/r06ap_census.txt 2>&1
