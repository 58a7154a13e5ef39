cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
pick='import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(sys.argv[1], "value",d["value"],"stream",d["stream"]["value"],d["stream"]["frames_per_s"],d["stream"]["stage_seconds_profiled"])'
python bench.py --no-exact-fp32 --no-cpu-baseline --train-steps 0 2>/dev/null | python -c "$pick" no-train > gpurun_out/r06p_order.txt 2>&1
python bench.py --no-exact-fp32 --no-cpu-baseline 2>/dev/null | python -c "$pick" after-train >> gpurun_out/r06p_order.txt 2>&1
python bench.py --no-exact-fp32 --no-cpu-baseline --train-steps 0 2>/dev/null | python -c "$pick" no-train >> gpurun_out/r06p_order.txt 2>&1
