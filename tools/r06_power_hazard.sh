# one GPU call: the power record and the library-free hazard reproducer (item 6)
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 600 python tools/power_record.py 10 > gpurun_out/r06_power.md 2> gpurun_out/r06_power.err
( cd tools/hazard/repro && make > /dev/null 2>&1 && timeout 600 ./cohazard.bin 200 ) > gpurun_out/r06_cohazard_repro.txt 2>&1
echo "repro exit $?" >> gpurun_out/r06_cohazard_repro.txt
