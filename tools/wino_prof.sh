# Winograd micro-benchmark: ablations + PMC passes (HBM fetch, L2 hit / miss, wait states) of the ResnetBlock layer.
#   bash tools/wino_prof.sh <tag>   -> gpurun_out/<tag>_wino_ablate.log, gpurun_out/<tag>_pmc_<group>.md
TAG=${1:-r03_wino}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
$ROOT/tools/wino_bench.bin 16 2 > $ROOT/gpurun_out/${TAG}_ablate.log 2>&1
cd /tmp && export TMPDIR=/tmp
run() {
  local name=$1; shift
  rm -rf $ROOT/gpurun_out/${TAG}_$name
  rocprofv3 --kernel-trace --pmc $PMC -d $ROOT/gpurun_out/${TAG}_$name -o prof -- $ROOT/tools/wino_bench.bin 16 1 > $ROOT/gpurun_out/${TAG}_$name.log 2>&1
  DB=$(find $ROOT/gpurun_out/${TAG}_$name -name "*results.db" | head -1)
  python $ROOT/tools/rocpd_summary.py pmc $DB $ROOT/gpurun_out/${TAG}_pmc_$name.md > /dev/null
  rm -rf $ROOT/gpurun_out/${TAG}_$name
  grep "wino" $ROOT/gpurun_out/${TAG}_pmc_$name.md
}
PMC="FETCH_SIZE" run fetch
PMC="WRITE_SIZE" run write
PMC="TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" run l2
PMC="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" run wait
PMC="SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU" run insts
cat $ROOT/gpurun_out/${TAG}_ablate.log
