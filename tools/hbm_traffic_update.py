#!/usr/bin/env python3
"""profiles/hbm_traffic.json from the FETCH_SIZE / WRITE_SIZE passes of tools/pmc_prof.sh:
    python tools/hbm_traffic_update.py <tag>      reads gpurun_out/<tag>_pmc_fetch.json and <tag>_pmc_write.json
Keys are the kernel names bench.py's LaunchProfiler uses (ap_conv2d_kernel_name; '+IN' for the in-kernel InstanceNorm form).
bytes = FETCH_SIZE KiB x 1024 x 2 (the gfx950 correction of MI355X_MICROARCH.md for wide coalesced reads) + WRITE_SIZE KiB x 1024."""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
fd = json.load(open(os.path.join(ROOT, 'gpurun_out', tag + '_pmc_fetch.json')))
wd = json.load(open(os.path.join(ROOT, 'gpurun_out', tag + '_pmc_write.json')))
path = os.path.join(ROOT, 'profiles', 'hbm_traffic.json')
tab = json.load(open(path))


def key_of(name):
    m = re.match(r'conv_bf16x3<(Bf3Cfg<[^>]*>)', name)
    if m:
        a = [x.strip() for x in m.group(1)[7:-1].split(',')]
        a += ['0', '0', '2', '0', '0'][len(a) - 6:] if len(a) < 11 else []
        fn = len(a) > 10 and a[10] == '1'
        core = a[:6] + ([a[6]] if (a[6] != '0' or a[7] != '0') else []) + ([a[7]] if a[7] != '0' else [])
        if a[1] == '0' and len(core) == 6:
            core.append(a[6])
        return 'Bf3Cfg<%s>%s' % (', '.join(core), ' +IN' if fn else '')
    m = re.match(r'conv_ph4<Ph4Cfg<(\d)', name)
    if m:
        return 'Ph4Cfg<%s>' % m.group(1)
    m = re.match(r'conv_direct_f32<(DirectCfg<[^>]*>)', name)
    if m:
        return m.group(1)
    return None


for name, v in fd.items():
    k = key_of(name)
    if k is None or name not in wd:
        continue
    f, w = v['FETCH_SIZE'], wd[name]['WRITE_SIZE']
    tab[k] = {'fetch_kib_raw': f, 'write_kib_raw': w, 'hbm_bytes_per_launch': int(round(2 * f * 1024 + w * 1024)), 'round': tag,
              'rocprof_kernel': name}
    print('%-40s %8.1f MB read x2, %8.1f MB written  <- %s' % (k, f * 1024 / 1e6, w * 1024 / 1e6, name))
# entries of other rounds are dropped: they belong to kernels as they were then (seven r01* keys of kernel names that no
# longer exist); a kernel the bench asks for and this round's passes did not see stops the bench (APAMD_BENCH_NO_TRAFFIC to develop)
stale = [k for k, v in tab.items() if v.get('round') != tag and not k.startswith('__')]      # '__train_step_*__': tools/train_hbm.sh totals
for k in stale:
    del tab[k]
if stale:
    print('dropped entries of other rounds: %s' % ', '.join(sorted(stale)))
json.dump(tab, open(path, 'w'), indent=1, sort_keys=True)
