TAG=${1:-r03z}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf $ROOT/gpurun_out/${TAG}_st
rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/${TAG}_st -o prof -- python $ROOT/bench.py --stream > $ROOT/gpurun_out/${TAG}_stream.json 2> $ROOT/gpurun_out/${TAG}_stream.err
DB=$(find $ROOT/gpurun_out/${TAG}_st -name "*results.db" | head -1)
python $ROOT/tools/rocpd_summary.py stats $DB $ROOT/gpurun_out/${TAG}_stream_kernel_stats.md | head -30
rm -rf $ROOT/gpurun_out/${TAG}_st
