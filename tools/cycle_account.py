#!/usr/bin/env python3
"""Cycle account of the persistent split-bf16 convolution (conv_bf16x3): where the cycles of a tile go.

Needs the experiment library (make -C animateportrait_amd/csrc ablate); every wave stamps s_memtime at the stage
synchronisation points (arrival, DMA landed, barrier passed), at the end of the MFMA stream, inside and after the epilogue
(conv_bf16x3.h, AP_STAMP).  Usage:
    APAMD_LIB=animateportrait_amd/libapamd_ablate.so python tools/cycle_account.py [layer substring] [out prefix]
Writes <prefix>.npz (raw stamps of the last launch) and prints the table that profiles/r05_dominant_cycle_account.md holds."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

# name, source segments, cout, k, stride, pad, reflection padding, transposed, H
LAYERS = {
    'res': ('res 256->256 k3 @64', [256], 256, 3, 1, 1, True, False, 64),
    'res2': ('res2 288->256 k3 @64', [256, 16, 16], 256, 3, 1, 1, True, False, 64),
    'merge': ('merge 768->256 k3 @64', [256, 256, 256], 256, 3, 1, 1, False, False, 64),
    'stem': ('stem 3->64 k7 @256 (row form)', [3], 64, 7, 1, 3, True, False, 256),
    'down': ('down 64->128 k3 s2 @256 (space-to-depth form)', [64], 128, 3, 2, 1, False, False, 256),
}
KIND = {0: 'A arrive', 1: 'B dma landed', 2: 'C barrier passed', 3: 'last MFMA issued', 4: 'epilogue done', 5: 'sync after epilogue',
        6: 'next chunk-1 DMA issued', 7: 'entry', 8: 'prologue landed', 9: 'stores issued', 10: 'stores acknowledged'}


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else 'res'
    prefix = sys.argv[2] if len(sys.argv) > 2 else 'gpurun_out/cycle_account_' + which
    words = int(os.environ.get('APAMD_STAMP_WORDS', '512'))
    import torch
    from animateportrait_amd import ops
    from animateportrait_amd.networks import ConvLayer
    name, segs, cout, k, stride, pad, refl, tr, h = LAYERS[which]
    mode = ops.PAD_REFLECT if refl else ops.PAD_ZERO
    dev = torch.device('cuda:0')
    n = int(os.environ.get('BATCH', '16'))
    layer = ConvLayer(segs, cout, k, stride, pad, mode, tr, 0).to(dev)
    torch.nn.init.normal_(layer.weight, 0, 0.02)
    srcs = []
    for c in segs:
        x = torch.randn(n, c, h, h, device=dev)
        srcs.append(ops.Feat(x, torch.zeros(n * c, device=dev), torch.ones(n * c, device=dev), ops.ACT_RELU))
    kw = dict(norm_act=ops.ACT_RELU)
    nwg = 512
    buf = torch.zeros(nwg * 4 * words, dtype=torch.int32, device=dev)
    # un-stamped timing first (same library), then the stamped launches
    for _ in range(3):
        layer.run(srcs, **kw)
    def timed(iters):
        prof = ops.LaunchProfiler()
        ops.PROFILER = prof
        for _ in range(iters):
            layer.run(srcs, **kw)
        ops.PROFILER = None
        out = {}
        for kn, v in prof.summary().items():
            out[kn] = v['ms'] * 1e3 / iters
        return out
    plain = timed(20)
    os.environ['APAMD_STAMP_BUF'] = hex(buf.data_ptr())
    os.environ['APAMD_STAMP_WORDS'] = str(words)
    for _ in range(3):
        layer.run(srcs, **kw)
    stamped = timed(20)
    buf.zero_()
    layer.run(srcs, **kw)
    torch.cuda.synchronize()
    del os.environ['APAMD_STAMP_BUF']
    raw = buf.cpu().numpy().astype(np.uint32).reshape(nwg, 4, words // 2, 2)
    np.savez_compressed(prefix + '.npz', raw=raw, plain_us=np.array(list(plain.values())), stamped_us=np.array(list(stamped.values())))
    print('layer', name, 'B', n)
    print('plain launch (us):', plain)
    print('stamped launch (us):', stamped)
    us = [v for kn, v in stamped.items() if 'conv_bf16x3' in kn or 'Bf3' in kn]
    report(raw, us[0] if us else None)


def report(raw, kernel_us=None, out=sys.stdout):
    nwg = raw.shape[0]
    live = [b for b in range(nwg) if raw[b, 0, 0, 0] != 0 or raw[b, 0, 0, 1] != 0]
    if not live:
        print('no stamps (does the launch leave 16 * APAMD_STAMP_WORDS bytes of LDS behind the stage buffers?)', file=out)
        return {}
    p = lambda *a: print(*a, file=out)
    p('workgroups with stamps: %d' % len(live))
    seg = {}       # name -> list of cycles (per wave, per tile)
    def add(nm, v):
        seg.setdefault(nm, []).append(float(v))
    spans = []
    per_stage_body, per_stage_wait, per_stage_bar = {}, {}, {}
    ntiles_seen = 0
    for b in live:
        for w in range(4):
            st = raw[b, w]
            cnt = 0                                            # (LDS behind the last stamp is not initialised: stop at the exit stamp)
            while cnt < st.shape[0] and (int(st[cnt, 0]) & 0xff) != 10:
                cnt += 1
            cnt = min(cnt + 1, st.shape[0])
            ev = [(int(c) & 0xff, (int(c) >> 8) & 0xff, int(c) >> 16, int(t)) for c, t in st[:cnt]]
            if not ev:
                continue
            d = lambda a, b_: (b_ - a) & 0xffffffff
            t_entry = ev[0][3]
            spans.append(d(t_entry, ev[-1][3]))
            prev_t, prev_kind = None, None
            tile_start = None
            for kind, c, tile, t in ev:
                if kind == 7:
                    pass
                elif kind == 8:
                    add('prologue: entry -> first two stages landed (first tile only)', d(prev_t, t))
                    tile_start = prev_t
                elif kind == 0:
                    nm = 'MFMA body: last tap of the previous stage + taps 0..T-2 of this one'
                    if c == 0:
                        nm = 'MFMA body of stage 0 (taps 0..T-2; after the prologue / the previous tile)'
                    add(nm, d(prev_t, t))
                    per_stage_body.setdefault(c, []).append(d(prev_t, t))
                elif kind == 1:
                    add('wait: vmcnt(0) (next stage landed; also the previous tile\'s stores acknowledged)', d(prev_t, t))
                    per_stage_wait.setdefault(c, []).append(d(prev_t, t))
                elif kind == 2:
                    add('wait: s_barrier (skew between the four waves)', d(prev_t, t))
                    per_stage_bar.setdefault(c, []).append(d(prev_t, t))
                elif kind == 3:
                    add('last tap of the last stage (+ DMA issue of the next tile\'s chunk 0)', d(prev_t, t))
                elif kind == 9:
                    add('epilogue: accumulators -> LDS transpose -> 16-byte stores issued, row sums', d(prev_t, t))
                elif kind == 4:
                    add('epilogue: statistics (barrier, cross-wave sums, partial store)', d(prev_t, t))
                    if w == 0:
                        ntiles_seen += 1
                elif kind == 5:
                    add('barrier before the stage buffer is refilled', d(prev_t, t))
                elif kind == 6:
                    add('DMA issue of the next tile\'s chunk 1', d(prev_t, t))
                elif kind == 10:
                    add('exit: last stores acknowledged', d(prev_t, t))
                prev_t, prev_kind = t, kind
    span = np.array(spans, dtype=np.float64)
    p('kernel span per wave (entry -> stores acknowledged): mean %.0f  min %.0f  max %.0f ticks' % (span.mean(), span.min(), span.max()))
    if kernel_us:
        p('stamped kernel %.1f us -> %.3f ticks/ns (s_memtime rate)' % (kernel_us, span.max() / (kernel_us * 1e3)))
    tiles_per_wave = ntiles_seen / max(1, len(live))
    p('tiles per workgroup: %.2f' % tiles_per_wave)
    p('')
    p('| segment | events per wave | mean ticks | median | p90 | ticks per wave and launch | share of span |')
    p('|---|---|---|---|---|---|---|')
    nw = len(live) * 4
    total = 0.0
    for nm, v in seg.items():
        a = np.array(v)
        per_wave = a.sum() / nw
        total += per_wave
        p('| %s | %.1f | %.0f | %.0f | %.0f | %.0f | %.1f %% |' % (nm, len(a) / nw, a.mean(), np.median(a), np.percentile(a, 90), per_wave,
                                                                  100 * per_wave / span.mean()))
    p('| sum | | | | | %.0f | %.1f %% |' % (total, 100 * total / span.mean()))
    p('')
    p('per stage (chunk) means: body / vmcnt wait / barrier wait')
    for c in sorted(per_stage_body):
        p('  chunk %2d: %7.0f %7.0f %7.0f' % (c, np.mean(per_stage_body[c]), np.mean(per_stage_wait.get(c, [0])), np.mean(per_stage_bar.get(c, [0]))))
    return seg


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1].endswith('.npz'):
        z = np.load(sys.argv[1])
        report(z['raw'], float(z['stamped_us'].max()))
    else:
        main()
