# rocprofv3 kernel stats of the train step (2 warm-up + 3 steps, B=16) in a given arithmetic mode:
#   bash tools/train_prof.sh bf16 r02g_train_bf16      -> gpurun_out/<tag>_kernel_stats.md
PREC=${1:-bf16x3}; TAG=${2:-r02_train_$PREC}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
APAMD_PRECISION=$PREC rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/$TAG -o prof -- python $ROOT/tools/train_bench.py 16 3 > $ROOT/gpurun_out/$TAG.log 2>&1
tail -3 $ROOT/gpurun_out/$TAG.log
DB=$(find $ROOT/gpurun_out/$TAG -name "*results.db" | head -1)
python $ROOT/tools/rocpd_summary.py stats $DB $ROOT/gpurun_out/${TAG}_kernel_stats.md | head -${HEAD:-70}
[ -n "$BYGRID" ] && python $ROOT/tools/rocpd_summary.py bygrid $DB $ROOT/gpurun_out/${TAG}_bygrid.md "$BYGRID"
rm -rf $ROOT/gpurun_out/$TAG
