# Per-kernel rocprofv3 stats of the train step with two builds of the library on ONE box:  bash tools/ab_prof.sh <libA> <libB> <tag> [precision]
#   -> gpurun_out/<tag>_A_kernel_stats.md, <tag>_B_kernel_stats.md and <tag>_diff.md (kernels whose time per step moved by > 0.05 ms)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
A=$1; B=$2; TAG=$3; P=${4:-bf16}
for X in A B; do
  LIB=$A; [ $X = B ] && LIB=$B
  APAMD_LIB=$ROOT/$LIB HEAD=0 bash $ROOT/tools/train_prof.sh $P ${TAG}_$X > /dev/null 2>&1
done
python3 - $ROOT/gpurun_out/${TAG}_A_kernel_stats.md $ROOT/gpurun_out/${TAG}_B_kernel_stats.md > $ROOT/gpurun_out/${TAG}_diff.md <<'PY'
import re, sys
def load(f):
    d = {}
    for l in open(f):
        m = re.match(r'\| `(.*)` \| (\d+) \| ([\d.]+) \| ([\d.]+) \|', l)
        if m: d[re.sub(r'(instnorm_bwd_split_kernel<\d+, \w+, \w+), \w+>', r'\1>', re.sub(r'(norm_split_kernel<\d, \w+), \d>', r'\1>', m.group(1)))] = d.get(m.group(1), (0, 0.0))
        if m:
            k = re.sub(r'(instnorm_bwd_split_kernel<\d+, \w+, \w+), \w+>', r'\1>', re.sub(r'(norm_split_kernel<\d, \w+), \d>', r'\1>', m.group(1)))
            c, t = d.get(k, (0, 0.0))
            d[k] = (c + int(m.group(2)), t + float(m.group(3)))
    return d
a, b = load(sys.argv[1]), load(sys.argv[2])
steps = 5          # tools/train_bench.py 16 3 under train_prof.sh: 2 warm-up + 3 timed steps, all profiled
ta, tb = sum(v[1] for v in a.values()) / steps / 1e3, sum(v[1] for v in b.values()) / steps / 1e3
print('| kernel | launches / step A | ms / step A | launches / step B | ms / step B | delta ms |\n|---|---|---|---|---|---|')
rows = []
for k in sorted(set(a) | set(b)):
    ca, xa = a.get(k, (0, 0.0)); cb, xb = b.get(k, (0, 0.0))
    rows.append((xb / steps / 1e3 - xa / steps / 1e3, k, ca / steps, xa / steps / 1e3, cb / steps, xb / steps / 1e3))
for d, k, ca, xa, cb, xb in sorted(rows):
    if abs(d) > 0.05: print('| `%s` | %.0f | %.3f | %.0f | %.3f | %+.3f |' % (k[:100], ca, xa, cb, xb, d))
print('\nTOTAL kernel time per step: A %.2f ms, B %.2f ms (%+.2f)' % (ta, tb, tb - ta))
PY
cat $ROOT/gpurun_out/${TAG}_diff.md
