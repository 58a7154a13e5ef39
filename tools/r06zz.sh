cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
bash tools/round_profiles.sh r06zz > gpurun_out/r06zz_round.log 2>&1
ROUTES=split bash tools/train_hbm.sh r06 > gpurun_out/r06_train_hbm.log 2>&1
