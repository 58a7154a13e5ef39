cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
ROOT=$PWD
cd /tmp && export TMPDIR=/tmp
rm -rf $ROOT/gpurun_out/r06al
rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/r06al -o prof -- python $ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-exact-fp32 --train-steps 0 --no-stream > /dev/null 2>&1
DB=$(find $ROOT/gpurun_out/r06al -name "*results.db" | head -1)
python $ROOT/tools/rocpd_summary.py bygrid $DB $ROOT/gpurun_out/r06al_gen_bygrid.md "" > /dev/null
rm -rf $ROOT/gpurun_out/r06al
BYGRID="" HEAD=5 bash $ROOT/tools/train_prof.sh bf16 r06al_train > /dev/null 2>&1
