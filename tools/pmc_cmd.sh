# rocprofv3 PMC passes over an arbitrary command, per-kernel averages of a few counter groups:
#   bash tools/pmc_cmd.sh <tag> <kernel-substring> <command...>   -> gpurun_out/<tag>_pmc_<group>.md
TAG=$1; PAT=$2; shift 2
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
run() {
  local name=$1; shift
  rm -rf $ROOT/gpurun_out/${TAG}_$name
  rocprofv3 --kernel-trace --pmc $PMC -d $ROOT/gpurun_out/${TAG}_$name -o prof -- "$@" > $ROOT/gpurun_out/${TAG}_$name.log 2>&1
  DB=$(find $ROOT/gpurun_out/${TAG}_$name -name "*results.db" | head -1)
  python $ROOT/tools/rocpd_summary.py pmc $DB $ROOT/gpurun_out/${TAG}_pmc_$name.md > /dev/null
  rm -rf $ROOT/gpurun_out/${TAG}_$name
  grep "$PAT" $ROOT/gpurun_out/${TAG}_pmc_$name.md
}
PMC="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" run wait "$@"
PMC="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES" run insts "$@"
PMC="SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU" run busy "$@"
PMC="SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_ATOMIC_RETURN SQ_INSTS_SMEM SQ_IFETCH" run lds "$@"
