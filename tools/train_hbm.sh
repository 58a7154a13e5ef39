# HBM bytes per kernel of the plain-bf16 train step (B = 16, stand-in aux nets), with the InstanceNorm backward writing the
# matrix kernels' operands itself (default) and with APAMD_NO_INBWD_SPLIT=1 (the fp32-gradient route):
#   bash tools/train_hbm.sh r04   -> gpurun_out/<tag>_train_hbm_{split,fp32route}.md  (+ a total line per route)
# One rocprofv3 run per counter (FETCH_SIZE / WRITE_SIZE do not fit one pass; --pmc only with --kernel-trace) + one timing run.
TAG=${1:-r04}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
STEPS=3
for ROUTE in ${ROUTES:-split fp32route}; do
  OFF=0; [ $ROUTE = fp32route ] && OFF=1
  for PASS in time fetch write; do
    D=$ROOT/gpurun_out/${TAG}_thbm_${ROUTE}_$PASS
    rm -rf $D
    if [ $PASS = time ]; then ARGS="--kernel-trace --stats"; elif [ $PASS = fetch ]; then ARGS="--kernel-trace --pmc FETCH_SIZE"; else ARGS="--kernel-trace --pmc WRITE_SIZE"; fi
    APAMD_PRECISION=bf16 APAMD_NO_INBWD_SPLIT=$OFF rocprofv3 $ARGS -d $D -o prof -- python $ROOT/tools/train_bench.py 16 $STEPS > $D.log 2>&1
    DB=$(find $D -name "*results.db" | head -1)
    if [ $PASS = time ]; then python $ROOT/tools/rocpd_summary.py stats $DB $ROOT/gpurun_out/${TAG}_thbm_${ROUTE}_stats.md > /dev/null
    else python $ROOT/tools/rocpd_summary.py pmc $DB $ROOT/gpurun_out/${TAG}_thbm_${ROUTE}_$PASS.md $ROOT/gpurun_out/${TAG}_thbm_${ROUTE}_$PASS.json > /dev/null; fi
    grep "train step" $D.log
    rm -rf $D
  done
  python $ROOT/tools/hbm_table.py $ROOT/gpurun_out/${TAG}_thbm_${ROUTE}_fetch.json $ROOT/gpurun_out/${TAG}_thbm_${ROUTE}_write.json \
      $ROOT/gpurun_out/${TAG}_thbm_${ROUTE}_stats.md $ROOT/gpurun_out/${TAG}_train_hbm_${ROUTE}.md $((STEPS + 2)) > /dev/null
  python - <<PY
rows = [l.strip().strip('|').split('|') for l in open('$ROOT/gpurun_out/${TAG}_train_hbm_${ROUTE}.md') if l.startswith('| \`')]
rd = sum(float(r[1]) * float(r[3]) for r in rows); wr = sum(float(r[1]) * float(r[4]) for r in rows)
us = sum(float(r[1]) * float(r[2]) for r in rows)
line = 'TOTAL ${ROUTE}: %.2f GB read + %.2f GB written per step over the listed kernels (%.1f ms of kernel time)' % (rd / 1e3, wr / 1e3, us / 1e3)
print(line); open('$ROOT/gpurun_out/${TAG}_train_hbm_${ROUTE}.md', 'a').write('\n' + line + '\n')
PY
done
