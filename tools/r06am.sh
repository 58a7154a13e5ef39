cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
APAMD_PRECISION=bf16 python tools/aten_census.py > gpurun_out/r06am_aten.txt 2>&1
