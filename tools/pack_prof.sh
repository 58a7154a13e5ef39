cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py tests/test_train_gpu.py -x -q -m gpu 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
ROOT=$GRAFT_REPO_ROOT
rm -rf $ROOT/gpurun_out/tr_pack
APAMD_PRECISION=bf16 rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/tr_pack -o prof -- python $ROOT/tools/train_bench.py 16 5 > $ROOT/gpurun_out/tr_pack.log 2>&1
DB=$(find $ROOT/gpurun_out/tr_pack -name "*results.db" | head -1)
python $ROOT/tools/rocpd_summary.py stats $DB $ROOT/gpurun_out/tr_pack_kernel_stats.md > /dev/null
rm -rf $ROOT/gpurun_out/tr_pack
tail -2 $ROOT/gpurun_out/tr_pack.log
grep "pack_" $ROOT/gpurun_out/tr_pack_kernel_stats.md
