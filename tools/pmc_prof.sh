# rocprofv3 PMC passes over the headline bench (generator forward, B=16), one counter group per pass as
# MI355X_MICROARCH.md prescribes (FETCH_SIZE and WRITE_SIZE do not fit one pass; --pmc only with --kernel-trace):
#   bash tools/pmc_prof.sh r02p      -> gpurun_out/<tag>_pmc_{mfma,wait,fetch,write}.md (+ .json)
TAG=${1:-r02_pmc}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
run() {  # name, counters...
  local name=$1; shift
  rm -rf $ROOT/gpurun_out/${TAG}_$name
  rocprofv3 --kernel-trace --pmc "$@" -d $ROOT/gpurun_out/${TAG}_$name -o prof -- python $ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-exact-fp32 --train-steps 0 --no-stream > $ROOT/gpurun_out/${TAG}_$name.log 2>&1
  DB=$(find $ROOT/gpurun_out/${TAG}_$name -name "*results.db" | head -1)
  python $ROOT/tools/rocpd_summary.py pmc $DB $ROOT/gpurun_out/${TAG}_pmc_$name.md $ROOT/gpurun_out/${TAG}_pmc_$name.json > /dev/null
  rm -rf $ROOT/gpurun_out/${TAG}_$name
  grep "Bf3Cfg<1, 3, 1, 2, 4, 4" $ROOT/gpurun_out/${TAG}_pmc_$name.md
}
run mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE
run wait SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_WAVE_CYCLES
run fetch FETCH_SIZE
run write WRITE_SIZE
