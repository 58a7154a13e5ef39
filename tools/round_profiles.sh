# Everything profiles/ holds for one tag, in one GPU call:  bash tools/round_profiles.sh r03z
#   <tag>_bench.json                       python bench.py (the driver's default command line)
#   <tag>_gen_kernel_stats.md              rocprofv3 --kernel-trace --stats of the generator leg
#   <tag>_pmc_{mfma,wait,fetch,write}.md   PMC passes (separate runs, --pmc with --kernel-trace only)
#   <tag>_hbm_per_kernel.md                HBM bytes / achieved TB/s per kernel from the fetch / write passes
#   <tag>_train_{bf16x3,bf16}_kernel_stats.md   rocprofv3 stats of the train step in both arithmetic modes
TAG=${1:-r03z}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
bash tools/gen_prof.sh ${TAG}_gen > gpurun_out/${TAG}_gen_prof.log 2>&1
mv gpurun_out/${TAG}_gen_kernel_stats.md gpurun_out/${TAG}_gen_kernel_stats.md 2>/dev/null
bash tools/pmc_prof.sh ${TAG} > gpurun_out/${TAG}_pmc.log 2>&1
python tools/hbm_table.py gpurun_out/${TAG}_pmc_fetch.json gpurun_out/${TAG}_pmc_write.json gpurun_out/${TAG}_gen_gen_kernel_stats.md gpurun_out/${TAG}_hbm_per_kernel.md 30 > /dev/null 2>&1 || \
python tools/hbm_table.py gpurun_out/${TAG}_pmc_fetch.json gpurun_out/${TAG}_pmc_write.json gpurun_out/${TAG}_gen_kernel_stats.md gpurun_out/${TAG}_hbm_per_kernel.md 30 > /dev/null 2>&1
cd /tmp && export TMPDIR=/tmp
for MODE in bf16x3 bf16; do
  rm -rf $ROOT/gpurun_out/${TAG}_tr_$MODE
  APAMD_PRECISION=$MODE rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/${TAG}_tr_$MODE -o prof -- python $ROOT/tools/train_bench.py 16 5 > $ROOT/gpurun_out/${TAG}_train_$MODE.log 2>&1
  DB=$(find $ROOT/gpurun_out/${TAG}_tr_$MODE -name "*results.db" | head -1)
  python $ROOT/tools/rocpd_summary.py stats $DB $ROOT/gpurun_out/${TAG}_train_${MODE}_kernel_stats.md > /dev/null
  rm -rf $ROOT/gpurun_out/${TAG}_tr_$MODE
  tail -2 $ROOT/gpurun_out/${TAG}_train_$MODE.log
done
ls $ROOT/gpurun_out | grep "^${TAG}"
