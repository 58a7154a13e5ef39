cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
bash tools/ab_train.sh abl/libapamd_r05.so animateportrait_amd/libapamd.so 3 bf16 > gpurun_out/r06i_ab_train.txt 2>&1
for i in 1 2; do APAMD_LIB=$PWD/abl/libapamd_r05.so python tools/gen_time.py; python tools/gen_time.py; done 2>&1 | grep frames > gpurun_out/r06i_gen.txt
python bench.py --stream > gpurun_out/r06i_stream.json 2> gpurun_out/r06i_stream.err
python -m pytest tests -x -q -m gpu > gpurun_out/r06i_tests.txt 2>&1
