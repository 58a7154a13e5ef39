cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
for sw in "" "APAMD_NO_D0_MFMA=1" "APAMD_NO_K7_WGRAD=1"; do
  env $sw python bench.py --no-cpu-baseline --no-exact-fp32 --no-stream --steps 5 --warmup 2 > gpurun_out/r06af_bench_${sw:-default}.json 2> /dev/null
done
