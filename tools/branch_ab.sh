# A/B of the encoder-branch streams (generator forward, B=16): bash tools/branch_ab.sh
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "generator" 2>&1 | tail -5
for v in 1 0 1 0; do
  APAMD_BRANCH_STREAMS=$v python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-exact-fp32 --train-steps 0 --no-stream > gpurun_out/branch_$v.json 2> gpurun_out/branch_$v.err
  python -c "import json; d=json.load(open('gpurun_out/branch_$v.json')); print('branch_streams=$v', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_us'])"
done
