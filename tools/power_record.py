#!/usr/bin/env python3
"""Is the matrix pipe power-limited?  The chip's own throttle accumulators around three workloads:
    python tools/power_record.py [seconds per workload] > gpurun_out/r06_power.md
  (i)   ONLY the dominant kernel conv_bf16x3<Bf3Cfg<1,3,1,2,4,4>> (the 256 -> 256 3x3 layer at B = 16), back to back
  (ii)  the whole generator forward at B = 16, back to back
  (iii) tools/mfma_peak.hip `loop`: a register-only v_mfma_f32_32x32x16_bf16 stream on all CUs, real split-bf16 operand data
  (iv)  the same stream on all-ones data (the control: same instruction stream, less toggling)
Around each: `amd-smi metric -v -E -p -c --json` (violation accumulators: ppt / thermal / hbm / prochot, per-XCD
"gfx clock below host limit because of power / thermal"; energy counter), and while it runs socket power from the hwmon file
of the same device sampled every millisecond plus `amd-smi metric -c -p` every ~0.3 s for the clocks.
Accumulators count in units of `accumulation_counter`: fraction of the interval = delta(acc) / delta(accumulation_counter)."""
import glob
import json
import os
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SECONDS = float(sys.argv[1]) if len(sys.argv) > 1 else 10.0


def smi(args):
    out = subprocess.run(['amd-smi', 'metric', '-g', '0'] + args + ['--json'], capture_output=True, text=True).stdout
    i = min([x for x in (out.find('['), out.find('{')) if x >= 0])
    d = json.loads(out[i:])
    g = d[0] if isinstance(d, list) else d
    if 'gpu_data' in g:
        g = g['gpu_data'][0]
    return g


def hwmon_power_file():
    """hwmon power1_input of the device amd-smi calls GPU 0 (matched by PCI address)."""
    out = subprocess.run(['amd-smi', 'list', '--json'], capture_output=True, text=True).stdout
    try:
        bdf = json.loads(out[out.find('['):])[0]['bdf'].lower()
    except Exception:
        bdf = None
    for f in glob.glob('/sys/class/drm/card*/device/hwmon/hwmon*/power1_input'):
        dev = os.path.realpath(os.path.dirname(os.path.dirname(os.path.dirname(f))))
        if bdf and dev.lower().endswith(bdf):
            return f, bdf
    return None, bdf


class Sampler(threading.Thread):
    def __init__(self, pfile):
        super().__init__(daemon=True)
        self.pfile, self.stop, self.p, self.clk = pfile, False, [], []

    def run(self):
        nxt = 0.0
        while not self.stop:
            t = time.perf_counter()
            if self.pfile:
                try:
                    self.p.append((t, int(open(self.pfile).read()) / 1e6))
                except Exception:
                    pass
            if t >= nxt:
                try:
                    g = smi(['-c', '-p'])
                    gfx = [v['clk']['value'] for k, v in g['clock'].items() if k.startswith('gfx_') and isinstance(v['clk'], dict)]
                    self.clk.append((t, sum(gfx) / len(gfx), min(gfx), max(gfx), g['power']['socket_power']['value']))
                except Exception:
                    pass
                nxt = time.perf_counter() + 0.25
            time.sleep(0.001)


def acc_of(g):
    t = g['throttle']
    def xs(k):
        v = t.get(k)
        if isinstance(v, dict):
            v = list(v.values())[0]
        return v
    return {'counter': t['accumulation_counter'], 'ppt': t['ppt_accumulated'], 'prochot': t['prochot_accumulated'],
            'socket_thermal': t['socket_thermal_accumulated'], 'vr_thermal': t['vr_thermal_accumulated'],
            'hbm_thermal': t['hbm_thermal_accumulated'],
            'below_host_power': xs('gfx_clk_below_host_limit_power_accumulated'),
            'below_host_thermal': xs('gfx_clk_below_host_limit_thermal_accumulated'),
            'below_host_total': xs('total_gfx_clk_below_host_limit_accumulated'),
            'low_util': xs('low_utilization_accumulated'),
            'energy_J': g['energy']['total_energy_consumption']['value']}


def run_workload(name, cmd, pfile, env=None):
    e = dict(os.environ)
    e.update(env or {})
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=e, cwd=ROOT)
    # the workload prints READY when its warm-up is done and runs SECONDS from there
    lines = []
    while True:
        l = proc.stdout.readline()
        if not l:
            break
        lines.append(l.rstrip())
        if l.startswith('READY'):
            break
    before = acc_of(smi(['-v', '-E']))
    t0 = time.perf_counter()
    s = Sampler(pfile)
    s.start()
    rest = proc.stdout.read()
    proc.wait()
    t1 = time.perf_counter()
    s.stop = True
    s.join()
    after = acc_of(smi(['-v', '-E']))
    lines += rest.splitlines()
    return {'name': name, 'before': before, 'after': after, 'power': s.p, 'clk': s.clk, 'wall': t1 - t0,
            'out': [l for l in lines if l.startswith('RESULT')]}


def frac(b, a, k):
    dc = a['counter'] - b['counter']
    v0, v1 = b[k], a[k]
    if isinstance(v0, list):
        ds = [(y - x) / dc for x, y in zip(v0, v1) if isinstance(x, (int, float))]
        return '%.1f-%.1f %%' % (100 * min(ds), 100 * max(ds)) if ds else 'n/a'
    if not isinstance(v0, (int, float)):
        return 'n/a'
    return '%.1f %%' % (100.0 * (v1 - v0) / dc)


def main():
    pfile, bdf = hwmon_power_file()
    peak = os.path.join(ROOT, 'gpurun_out', 'mfma_peak')
    subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-Wno-unused-value', '-Wno-unused-result', os.path.join(ROOT, 'tools', 'mfma_peak.hip'), '-o', peak])
    py = sys.executable
    loads = [
        ('(i) only `conv_bf16x3<Bf3Cfg<1,3,1,2,4,4>>` (256 -> 256 3x3 @ 64 x 64, B = 16), back to back', [py, 'tools/power_loads.py', 'dominant', str(SECONDS)]),
        ('(ii) the whole generator forward, B = 16, back to back', [py, 'tools/power_loads.py', 'forward', str(SECONDS)]),
        ('(iii) `mfma_peak loop`: register-only MFMA stream, all CUs, real split-bf16 operand data', [peak, 'loop', str(SECONDS), '2']),
        ('(iv) the same stream on all-ones data (control)', [peak, 'loop', str(SECONDS), '0']),
        ('(v) plain-bf16 train step, B = 16, stand-in aux nets', [py, 'tools/power_loads.py', 'train', str(SECONDS)]),
    ]
    print('# Power limit or not: the chip\'s own throttle accumulators (round 6)\n')
    print('`python tools/power_record.py %g` on one MI355X (GPU 0 = %s, hwmon file `%s`, cap %s W).' % (
        SECONDS, bdf, pfile, (int(open(pfile.replace('power1_input', 'power1_cap')).read()) // 1000000) if pfile else '?'))
    print('Accumulators from `amd-smi metric -v` before / after each workload; a residency = delta(accumulator) / delta(`accumulation_counter`).')
    print('Power: hwmon `power1_input` read every millisecond (the firmware refreshes it more slowly: distinct values are counted); clocks: `amd-smi metric -c` every ~0.3 s, mean over the 8 XCDs.\n')
    rows = []
    for name, cmd in loads:
        time.sleep(3.0)             # let the board cool to the same idle state
        r = run_workload(name, cmd, pfile)
        b, a = r['before'], r['after']
        pw = [v for _, v in r['power']]
        pw_sorted = sorted(pw)
        distinct = sum(1 for i in range(1, len(pw)) if pw[i] != pw[i - 1])
        ck = r['clk']
        print('## %s\n' % name)
        for l in r['out']:
            print('    ' + l)
        print()
        print('| quantity | value |\n|---|---|')
        print('| interval | %.2f s, `accumulation_counter` +%d (%.0f counts/s) |' % (r['wall'], a['counter'] - b['counter'], (a['counter'] - b['counter']) / r['wall']))
        print('| **PPT (socket power limit) residency** | **%s** |' % frac(b, a, 'ppt'))
        print('| gfx clock below host limit because of POWER, per XCD | %s |' % frac(b, a, 'below_host_power'))
        print('| gfx clock below host limit because of TEMPERATURE, per XCD | %s |' % frac(b, a, 'below_host_thermal'))
        print('| gfx clock below host limit, any reason, per XCD | %s |' % frac(b, a, 'below_host_total'))
        print('| low-utilisation residency, per XCD | %s |' % frac(b, a, 'low_util'))
        print('| PROCHOT / socket thermal / VR thermal / HBM thermal residency | %s / %s / %s / %s |' % (
            frac(b, a, 'prochot'), frac(b, a, 'socket_thermal'), frac(b, a, 'vr_thermal'), frac(b, a, 'hbm_thermal')))
        print('| energy counter | %.0f J over the interval = %.0f W average |' % (a['energy_J'] - b['energy_J'], (a['energy_J'] - b['energy_J']) / r['wall']))
        if pw:
            print('| hwmon socket power | %d reads, %d value changes (%.0f Hz); min %.0f / median %.0f / p95 %.0f / max %.0f W |' % (
                len(pw), distinct, distinct / r['wall'], pw_sorted[0], pw_sorted[len(pw) // 2], pw_sorted[int(len(pw) * 0.95)], pw_sorted[-1]))
        if ck:
            ck2 = ck[len(ck) // 5:]     # the ramp-up fifth left out
            print('| gfx clock (amd-smi, %d samples after ramp-up) | mean %.0f MHz, slowest XCD %.0f, fastest %.0f; amd-smi socket power mean %.0f W |' % (
                len(ck2), sum(c[1] for c in ck2) / len(ck2), min(c[2] for c in ck2), max(c[3] for c in ck2), sum(c[4] for c in ck2) / len(ck2)))
        print()
        rows.append(r)


if __name__ == '__main__':
    main()
