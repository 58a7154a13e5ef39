# end-of-round evidence set: profiles/r06zz_* (tools/round_profiles.sh), the train step's HBM bytes per kernel, the GPU suite
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
bash tools/round_profiles.sh r06zz > gpurun_out/r06zz_round.log 2>&1
ROUTES=split bash tools/train_hbm.sh r06 > gpurun_out/r06_train_hbm.log 2>&1
python -m pytest tests -x -q -m gpu > gpurun_out/r06zz_tests.txt 2>&1
