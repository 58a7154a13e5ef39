cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
export APAMD_LIB=$PWD/animateportrait_amd/libapamd_ablate.so
python tools/cycle_account.py res gpurun_out/r06_epi_full > gpurun_out/r06_epi_full.txt 2>&1
APAMD_ABLATE=16 python tools/cycle_account.py res gpurun_out/r06_epi_nostores > gpurun_out/r06_epi_nostores.txt 2>&1
APAMD_ABLATE=8 python tools/cycle_account.py res gpurun_out/r06_epi_none > gpurun_out/r06_epi_none.txt 2>&1
unset APAMD_LIB
python bench.py --stream > gpurun_out/r06g_stream.json 2> gpurun_out/r06g_stream.err
