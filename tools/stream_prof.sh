ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf $ROOT/gpurun_out/r03z_st
rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/r03z_st -o prof -- python $ROOT/bench.py --stream > $ROOT/gpurun_out/r03z_stream.json 2> $ROOT/gpurun_out/r03z_stream.err
DB=$(find $ROOT/gpurun_out/r03z_st -name "*results.db" | head -1)
python $ROOT/tools/rocpd_summary.py stats $DB $ROOT/gpurun_out/r03z_stream_kernel_stats.md | head -30
rm -rf $ROOT/gpurun_out/r03z_st
