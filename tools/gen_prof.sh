# rocprofv3 kernel stats of the headline bench (generator forward, B=16): bash tools/gen_prof.sh <tag>
TAG=${1:-r02_gen}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf $ROOT/gpurun_out/$TAG
rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/$TAG -o prof -- python $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-exact-fp32 --train-steps 0 --no-stream > $ROOT/gpurun_out/$TAG.json 2> $ROOT/gpurun_out/$TAG.err
DB=$(find $ROOT/gpurun_out/$TAG -name "*results.db" | head -1)
python $ROOT/tools/rocpd_summary.py stats $DB $ROOT/gpurun_out/${TAG}_kernel_stats.md | head -40
rm -rf $ROOT/gpurun_out/$TAG
python -c "import json; d=json.load(open('$ROOT/gpurun_out/$TAG.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_us'])"
