cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
for i in 1 2 3; do python bench.py --no-stream --no-exact-fp32 --no-cpu-baseline --steps 10 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('value',d['value'],'bf16',d['train_step_bf16']['ms_per_step'],'standin',d['train_step_bf16']['ms_per_step_with_standin_aux'],'netF',d['train_step_bf16']['netF_ms_per_step'],'x3',d['train_step']['ms_per_step'])
"; done > gpurun_out/r06o_bench_variance.txt 2>&1
for lib in abl/libapamd_r05.so animateportrait_amd/libapamd.so abl/libapamd_r05.so animateportrait_amd/libapamd.so; do echo "== $lib"; APAMD_LIB=$PWD/$lib python tools/norm_split_bench.py 50 2>&1 | grep -v amdgpu.ids | grep 16x256x64; done > gpurun_out/r06o_norm_split.txt 2>&1
