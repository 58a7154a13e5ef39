cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
bash tools/ablate_wgrad.sh bf16 > gpurun_out/r06m_ablate_wgrad.txt 2>&1
