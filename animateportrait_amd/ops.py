"""Operator layer: thin Python wrappers that hand raw device pointers to the HIP kernels.

``Feat`` is the unit the layers exchange: a feature map that may still be *virtual*,
i.e. a raw convolution output plus the per-(n,c) InstanceNorm statistics and the
activation that the CONSUMER applies while it stages the tensor into LDS.  This is how
``Conv2d -> InstanceNorm2d -> ReLU`` chains of the reference
(Module2/models/networks.py:1218-1282) run as one kernel per convolution.
"""
import ctypes
import os

import torch

from . import _capi as C
from ._capi import ACT_NONE, ACT_RELU, ACT_LRELU, ACT_TANH, PAD_ZERO, PAD_REFLECT, W_OIHW, W_IOHW  # noqa: F401

EPS = 1e-5  # nn.InstanceNorm2d default (networks.py:33-34)

# bumped by optimisers that update parameters through raw pointers (optim.FlatAdam); part of the key of the
# per-layer packed-weight caches
WEIGHTS_EPOCH = 0


def weight_key(w):
    """What a packed copy of parameter ``w`` is valid for: torch's version counter (in-place torch ops), the address, the
    package-wide epoch (raw-storage updates such as the start-up broadcast) and the epoch of the FlatAdam that owns the
    parameter (its update goes through a raw pointer) -- per optimizer, so that a discriminator step does not throw
    the generator's packed weights away and vice versa."""
    owner = getattr(w, '_flat_owner', None)
    return (w._version, w.data_ptr(), WEIGHTS_EPOCH, owner.epoch if owner is not None else 0)

# Arithmetic of the wide 3x3 layers (include/animateportrait_amd.h, ap_conv_desc.precision):
#   'fp32'   exact fp32 MFMA everywhere;
#   'bf16x3' operands split into bf16 head + tail, three bf16 MFMAs per tile, fp32 accumulation: fp32-class
#            results (generator L-inf ~1e-4 vs the fp32 reference, inside the 1e-3 budget) at ~5x the MFMA rate.
# Picked up by every ConvSpec created afterwards (APAMD_PRECISION=fp32 forces exact arithmetic).
#   'bf16'   plain bf16 operands (the head parts only), ONE MFMA per tile, fp32 accumulation, fp32 master weights: the
#            training configurations (BASELINE configs[2-3]); ~3x less matrix work than bf16x3, bf16-autocast accuracy.
PRECISION_FP32, PRECISION_BF16X3, PRECISION_BF16 = 0, 1, 2
PRECISION_BY_NAME = {'fp32': PRECISION_FP32, 'bf16x3': PRECISION_BF16X3, 'bf16': PRECISION_BF16}
DEFAULT_PRECISION = PRECISION_BY_NAME.get(os.environ.get('APAMD_PRECISION', 'bf16x3'), PRECISION_BF16X3)


_LAST_STREAM = {}          # device index -> torch stream of this package's previous launch
_STREAM_FENCE = os.environ.get('APAMD_NO_STREAM_FENCE', '0') != '1'


def _stream():
    """The stream a launch goes to (torch's current one) -- and the fence of DESIGN.md section 3.9: no two kernels of this library
    may share compute units, or the second one can compute wrong values.  Callers that move between streams (graph capture warm-ups,
    user side streams) are therefore serialised HERE: when the current stream differs from the one of the previous launch on this
    device, it first waits for an event recorded behind that launch.  Costs nothing while the stream stays the same; inside a
    stream capture no foreign event may be waited for, so captures start from a synchronised device (flow_unet_hip, aux_nets)."""
    s = torch.cuda.current_stream()
    if _STREAM_FENCE:
        dev = s.device.index
        last = _LAST_STREAM.get(dev)
        if last is not None and last.cuda_stream != s.cuda_stream and not torch.cuda.is_current_stream_capturing():
            ev = torch.cuda.Event()
            ev.record(last)
            s.wait_event(ev)
        _LAST_STREAM[dev] = s
    return ctypes.c_void_p(s.cuda_stream)


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def _require_device(t, name, allow_bf16=False):
    """allow_bf16: the caller's kernel also reads a tensor of bf16 values (a raw convolution output / a data gradient stored by
    ap_conv2d_fwd_bf16out in the plain-bf16 train step) and is told so; everybody else refuses one loudly."""
    if not t.is_cuda:
        raise RuntimeError('animateportrait_amd: %s must live on the MI355X (got a %s tensor); '
                           'this package has no CPU path' % (name, t.device))
    if not t.is_contiguous() or not (t.dtype == torch.float32 or (allow_bf16 and t.dtype == torch.bfloat16)):
        raise RuntimeError('animateportrait_amd: %s must be contiguous fp32%s (got %s)' % (name, ' or bf16' if allow_bf16 else '', t.dtype))


# Plain-bf16 training (configs[2-3]): the raw outputs of the ResNet trunk's main-branch convolutions and the gradients that leave
# their data-gradient convolutions are STORED as bf16 (ap_conv2d_fwd_bf16out; every reader rounds them to bf16 anyway);
# APAMD_NO_BF16_RAW=1 keeps them fp32 (A/B).  Measured (round 5, profiles/r05p_bf16_raw.md): 148.7 -> 143.6 GB of HBM traffic per step,
# -1.1 GB of memory, and only -0.4 ms of 56 -- the passes that read these tensors (norm_split, instnorm_bwd_split, the padded-row
# operand transposition) run at 3.5-4.4 TB/s because of latency / issue, not bandwidth: with 8-byte instead of 16-byte loads per
# lane they take as long per launch.
BF16_RAW = os.environ.get('APAMD_NO_BF16_RAW', '0') != '1'
# the weight gradients' shifted operand re-tiled from the forward pass's split copies (ap_wgrad_desc.src_xs); 1 turns it off
XS_WGRAD = os.environ.get('APAMD_NO_XS_WGRAD', '0') != '1'
# ... and, where the gradient's split copy exists too, both operands read as split copies by the kernel itself (ap_conv2d_wgrad_xs)
XS_DIRECT = XS_WGRAD and os.environ.get('APAMD_NO_XS_DIRECT', '0') != '1'


class Feat:
    """act((data - mean) * rstd) with mean/rstd of shape (N*C,), or a plain tensor when mean is None.

    A convolution hands its InstanceNorm statistics over as per-tile partial sums (``pending``); they are
    finalised by whichever consumer comes first -- inside the fused norm/residual/split pass when that is the
    consumer (ap_norm_apply_split), by a standalone ap_instnorm_finalize when ``mean`` / ``rstd`` are read."""
    __slots__ = ('data', '_mean', '_rstd', 'act', 'xs', 'pending', 'xs_rows', 'xs_heads_only', 's2d', 'oct')

    def __init__(self, data, mean=None, rstd=None, act=ACT_NONE, pending=None):
        self.data, self._mean, self._rstd, self.act = data, mean, rstd, act
        self.pending = pending   # (partials [N*C, tiles, 2], tiles) of the producing convolution, or None
        self.xs = None           # split-bf16 copy (ap_split_prepass), made on first use and shared by all consumers
        self.xs_rows = None      # {(k, pad, pad_mode): row expansion for k x k stems (ap_split_prepass_rows)}
        self.xs_heads_only = False   # the split copy was written without its tail planes (package mode plain bf16)
        self.s2d = None              # space-to-depth split copy (a split-only Feat) made by the producer (warp_concat s2d=True)
        self.oct = None              # the fp32 values in the channel-octet layout [N, C/8, H*W, 8] (conv2d out_octet=True);
                                     # ``data`` is then a storage-less stand-in: only warp_concat reads such a feature

    @property
    def shape(self):
        return self.data.shape

    @classmethod
    def split_only(cls, shape, xs):
        """A feature that exists only as its split-bf16 copy (inference: the fp32 tensor is never written).
        ``data`` is a storage-less stand-in that carries shape and device; fp32 consumers reject it."""
        f = cls(torch.empty(1, dtype=torch.float32, device=xs.device).expand(shape))
        f.xs = xs
        return f

    @property
    def is_split_only(self):
        """No fp32 NCHW tensor behind ``data`` (a storage-less stand-in): the feature lives as its split copy, its
        space-to-depth copy and / or its channel-octet form only."""
        return ((self.xs is not None or self.s2d is not None or self.oct is not None) and self.data.stride(0) == 0 and
                self.data.numel() > 1)

    @property
    def virtual(self):
        return self._mean is not None or self.pending is not None

    def _finalize(self):
        if self.pending is not None:
            partial, tiles = self.pending
            n, c, h, w = self.data.shape
            self._alloc_stats()
            if self.oct is not None:
                C.check(C.lib().ap_instnorm_finalize_octet(_ptr(partial), _ptr(self.oct), n, c, tiles, h * w, EPS,
                                                           _ptr(self._mean), _ptr(self._rstd), _stream()), 'instnorm_finalize_octet')
                self.pending = None
                return
            if self.data.dtype != torch.float32:
                raise RuntimeError('a bf16 raw output finalises its statistics inside ap_norm_apply_split (its first consumer), not here')
            C.check(C.lib().ap_instnorm_finalize(_ptr(partial), _ptr(self.data), n * c, tiles, h * w, EPS, _ptr(self._mean),
                                                 _ptr(self._rstd), _stream()), 'instnorm_finalize')
            self.pending = None

    def _alloc_stats(self):
        n, c = self.data.shape[:2]
        self._mean = torch.empty(n * c, dtype=torch.float32, device=self.data.device)
        self._rstd = torch.empty_like(self._mean)

    @property
    def mean(self):
        self._finalize()
        return self._mean

    @property
    def rstd(self):
        self._finalize()
        return self._rstd

    def batch_slice(self, lo, hi):
        if self.oct is not None:
            raise RuntimeError('a channel-octet feature is read by warp_concat only')
        c = self.data.shape[1]
        return Feat(self.data[lo:hi], None if self.mean is None else self.mean[lo * c:hi * c],
                    None if self.rstd is None else self.rstd[lo * c:hi * c], self.act)


class ConvSpec:
    """Static part of one convolution-like operator (everything except N, H, W and pointers)."""

    def __init__(self, cin_segments, cout, k, stride=1, pad=0, pad_mode=PAD_ZERO, transposed=False,
                 output_padding=0, w_layout=W_OIHW, w_flip=False):
        self.cin_segments = tuple(cin_segments)
        self.cout, self.k, self.stride, self.pad, self.pad_mode = cout, k, stride, pad, pad_mode
        self.transposed, self.output_padding = transposed, output_padding
        self.w_layout, self.w_flip = w_layout, w_flip
        self.precision = DEFAULT_PRECISION
        self.kh = None            # 1: the 1 x k row form of a k x k stem (see stem_rows_spec)

    def desc(self, n, h, w, srcs=None, act=ACT_NONE):
        d = C.ApConvDesc()
        d.N, d.H, d.W, d.Cout = n, h, w, self.cout
        d.KH, d.KW = (self.kh or self.k), self.k
        d.stride, d.pad, d.pad_mode = self.stride, self.pad, self.pad_mode
        d.transposed, d.output_padding = int(self.transposed), self.output_padding
        d.w_layout, d.w_flip, d.act = self.w_layout, int(self.w_flip), act
        d.nsrc = len(self.cin_segments)
        d.precision = self.precision
        d.s2d_k = getattr(self, 's2d_k', 0)
        for i, c in enumerate(self.cin_segments):
            d.src[i].C = c
        if srcs is not None:
            self.fill_sources(d, srcs)
        return d

    def fill_sources(self, d, srcs):
        """Point the descriptor at (virtual) fp32 sources; reading mean / rstd finalises pending statistics."""
        for i, f in enumerate(srcs):
            d.src[i].data = f.data.data_ptr()
            d.src[i].mean = f.mean.data_ptr() if f.virtual else None
            d.src[i].rstd = f.rstd.data_ptr() if f.virtual else None
            d.src[i].act = f.act

    def out_size(self, h, w):
        d = self.desc(1, h, w)
        ho, wo = ctypes.c_int32(), ctypes.c_int32()
        C.check(C.lib().ap_conv2d_out_size(ctypes.byref(d), ctypes.byref(ho), ctypes.byref(wo)), 'conv2d_out_size')
        return ho.value, wo.value


class LaunchProfiler:
    """Optional per-launch timing for bench.py: brackets every convolution launch with events on the
    launch stream (the kernels run on torch's current stream, so torch.cuda.Event is the HIP event on
    that stream) and records the kernel instantiation and the layer's algorithmic FLOPs."""

    def __init__(self):
        self.records = []   # (kernel name, flops, start event, end event)
        self.calls = {}     # un-timed launches worth knowing about: name -> count (tests read which plans ran)

    def note(self, name):
        self.calls[name] = self.calls.get(name, 0) + 1

    def summary(self):
        """Per kernel instantiation: launches, total ms, FLOPs.  A bracket far above its group's median (same kernel, same
        FLOPs) is a scheduling hiccup between the two events (seen: one 22 ms bracket around a 0.2 ms launch), not kernel
        time: it is replaced by the group's median and counted in ``outliers``."""
        torch.cuda.synchronize()
        groups = {}
        for name, flops, e0, e1 in self.records:
            groups.setdefault((name, flops), []).append(e0.elapsed_time(e1))
        agg = {}
        for (name, flops), ts in groups.items():
            med = sorted(ts)[len(ts) // 2]
            a = agg.setdefault(name, dict(launches=0, ms=0.0, flops=0.0, outliers=0))
            for t in ts:
                if t > 4.0 * med:
                    a['outliers'] += 1
                    t = med
                a['ms'] += t
            a['launches'] += len(ts)
            a['flops'] += flops * len(ts)
        return agg


PROFILER = None


def pack_weights(spec, weight):
    """Re-lay ``weight`` (nn.Conv2d OIHW or nn.ConvTranspose2d IOHW) into the kernel's LDS image."""
    _require_device(weight, 'weight')
    d = spec.desc(1, 64, 64)
    n = C.check(C.lib().ap_conv2d_packed_floats(ctypes.byref(d)), 'conv2d_packed_floats')
    packed = torch.empty(n, dtype=torch.float32, device=weight.device)
    C.check(C.lib().ap_conv2d_pack_weights(ctypes.byref(d), _ptr(weight), _ptr(packed), _stream()), 'pack_weights')
    return packed


class WeightView:
    """A packer operand: ``t`` = a (possibly non-contiguous) 4-D view of a layer's parameter, plus the derived forms
    (``s2d_c``: space-to-depth form over 4 * s2d_c channels; ``rows_c``: 1 x K row form over rows_c channels)."""
    __slots__ = ('t', 's2d_c', 'rows_c')

    def __init__(self, t, s2d_c=0, rows_c=0):
        self.t, self.s2d_c, self.rows_c = t, s2d_c, rows_c

    def dense(self):
        """The operand as the contiguous tensor ap_conv2d_pack_weights takes (the unbatched path)."""
        if self.s2d_c:
            return s2d_weight(self.t)
        if self.rows_c:
            return stem_rows_weight(self.t)
        return self.t.contiguous()

    def c_view(self, spec):
        v = C.ApWeightView()
        t = self.t
        v.w = t.data_ptr()
        derived = bool(self.s2d_c or self.rows_c)
        oihw = derived or spec.w_layout == W_OIHW            # the derived forms index the layer's own OIHW parameter
        v.s_co, v.s_ci = (t.stride(0), t.stride(1)) if oihw else (t.stride(1), t.stride(0))
        v.s_ky, v.s_kx = t.stride(2), t.stride(3)
        v.s2d_c, v.rows_c, v.ksrc = self.s2d_c, self.rows_c, t.shape[2]
        return v


class PackedSlot:
    __slots__ = ('buf', 'key', 'batched', 'owner')

    def __init__(self, buf, batched):
        self.buf, self.key, self.batched, self.owner = buf, None, batched, None


class PackSet:
    """Every split-bf16 packed image of the layers whose parameters ONE optimiser (optim.FlatAdam) owns, as a device-resident
    table of packer entries: after an optimiser step all of them are stale at once and are rebuilt by ONE launch
    (ap_conv2d_pack_run) instead of one per (layer, operator) -- 185 launches per train step before.  The pointers in the
    table are stable: parameters are views of the optimiser's flat buffer, the packed images are allocated once."""

    def __init__(self):
        self.host = bytearray()
        self.count = 0
        self.table = None            # device copy of ``host``; None: to be uploaded
        self.members = []            # (PackedSlot, weight Parameter)

    def register(self, slot, weight, entry_bytes, n):
        self.host += entry_bytes
        self.count += n
        self.table = None
        self.members.append((slot, weight))

    def refresh(self, device):
        if self.table is None:
            import numpy as np
            self.table = torch.from_numpy(np.frombuffer(bytes(self.host), dtype=np.uint8).copy()).to(device)
        C.check(C.lib().ap_conv2d_pack_run(_ptr(self.table), self.count, _stream()), 'pack_run')
        for slot, w in self.members:
            slot.key = weight_key(w)


def packed_slot(spec, weight, view, slot):
    """The packed image of operator ``spec`` over ``view`` of parameter ``weight``, current for the weight's present value.
    slot: the PackedSlot of an earlier call or None.  Parameters owned by a FlatAdam are packed through its PackSet."""
    key = weight_key(weight)
    if slot is not None and slot.key == key:
        return slot
    if slot is not None and (slot.buf.device != weight.device or
                             (slot.batched and slot.owner is not getattr(weight, '_flat_owner', None))):
        slot = None                      # the module moved to another device / to another optimiser: start over
    owner = getattr(weight, '_flat_owner', None)
    lib = C.lib()
    if slot is None:
        d = spec.desc(1, 64, 64)
        n = C.check(lib.ap_conv2d_packed_floats(ctypes.byref(d)), 'conv2d_packed_floats')
        buf = torch.empty(n, dtype=torch.float32, device=weight.device)
        slot = PackedSlot(buf, False)
        if owner is not None and weight.is_cuda:
            eb = lib.ap_conv2d_pack_entry_bytes()
            room = (ctypes.c_ubyte * (4 * eb))()
            cv = view.c_view(spec)
            cnt = C.check(lib.ap_conv2d_pack_entries(ctypes.byref(d), ctypes.byref(cv), _ptr(buf), room, 4), 'pack_entries')
            if cnt > 0:
                ps = getattr(owner, '_packset', None)
                if ps is None:
                    ps = owner._packset = PackSet()
                ps.register(slot, weight, bytes(room)[:cnt * eb], cnt)
                slot.batched, slot.owner = True, owner
    if slot.batched and slot.key is not None:
        owner._packset.refresh(weight.device)             # stale by an optimiser step: every image of the set in one launch
    else:
        # the unbatched packer -- also for the first image of a table member (the table run would rebuild all the others)
        d = spec.desc(1, 64, 64)
        dense = view.dense()
        _require_device(dense, 'weight')
        C.check(lib.ap_conv2d_pack_weights(ctypes.byref(d), _ptr(dense), _ptr(slot.buf), _stream()), 'pack_weights')
        slot.key = key
    return slot


def presplit(f, precision=None):
    """Split-bf16 copy of a (virtual) feature: XS[n][head|tail][C/8][H*W][8 x bf16] with the producer's
    InstanceNorm + activation applied (ap_split_prepass).  Cached on the Feat: one pass serves every consumer.
    precision: arithmetic of the consuming layer -- a bf16x3 consumer reads the tail planes, so a copy that was written
    without them (package mode plain bf16 at the time) is refused instead of read uninitialised."""
    if f.xs is None:
        _norm_apply_split(f, None, want_y=False, want_xs=True)
    if precision == PRECISION_BF16X3 and f.xs_heads_only:
        raise RuntimeError('a split-bf16 (bf16x3) layer was handed a split copy written without its tail planes '
                           '(made while the package mode was plain bf16)')
    return f.xs


ROW_CHANNELS = 32   # channels of the row expansion (k * C <= 32)


def stem_rows_eligible(spec):
    """7x7 'same' stems with <= 4 input channels (networks.py:1218, 1231, 1244) run as a 1x7 split-bf16 convolution
    over the row expansion of their input."""
    return (spec.precision != PRECISION_FP32 and spec.k == 7 and spec.stride == 1 and spec.pad == 3 and
            not spec.transposed and len(spec.cin_segments) == 1 and spec.cin_segments[0] <= 4 and spec.cout >= 32 and
            spec.w_layout == W_OIHW and not spec.w_flip)


def stem_rows_spec(spec):
    s = ConvSpec([ROW_CHANNELS], spec.cout, spec.k, 1, spec.pad, spec.pad_mode)
    s.kh = 1
    s.precision = spec.precision
    s.alg_macs = spec.cin_segments[0] * spec.k * spec.k     # algorithmic MACs per output value (profiler accounting)
    return s


def stem_rows_weight(weight):
    """W[co][c][ky][kx] -> W'[co][ky*C + c][0][kx], zero for the unused row channels."""
    co, c, k, _ = weight.shape
    w = weight.permute(0, 2, 1, 3).reshape(co, k * c, 1, k)
    out = torch.zeros((co, ROW_CHANNELS, 1, k), dtype=weight.dtype, device=weight.device)
    out[:, :k * c] = w
    return out


def presplit_rows(f, k, pad, pad_mode):
    """Row expansion of a (virtual) stem input as a split-only 32-channel Feat; cached on the source, so the three
    stems of the generator share one pass."""
    key = (k, pad, pad_mode)
    if f.xs_rows is None:
        f.xs_rows = {}
    hit = f.xs_rows.get(key)
    if hit is None:
        x = f.data
        n, c, h, w = x.shape
        _require_device(x, 'stem input')
        s = C.ApSrc()
        s.data, s.C, s.act = x.data_ptr(), c, f.act
        if f.virtual:
            s.mean, s.rstd = f.mean.data_ptr(), f.rstd.data_ptr()
        nbytes = C.check(C.lib().ap_split_prepass_bytes(n, ROW_CHANNELS, h, w), 'split_prepass_bytes')
        xs = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
        C.check(C.lib().ap_split_prepass_rows(ctypes.byref(s), n, h, w, k, pad, pad_mode, _ptr(xs), _stream()),
                'split_prepass_rows')
        hit = Feat.split_only((n, ROW_CHANNELS, h, w), xs)
        f.xs_rows[key] = hit
    return hit


def s2d_eligible(spec, h, w):
    """4x4 stride-2 pad-1 layers with >= 32 input channels (PatchGAN body, networks.py:2620-2636) -- and, in split-bf16
    arithmetic, the 3x3 stride-2 encoder layers (7 of their 16 space-to-depth taps are zero and skipped, desc.s2d_k) --
    run as a 2x2 stride-1 split-bf16 convolution over the space-to-depth copy of their input."""
    # (3x3: split arithmetic only -- measured +1 % on the generator forward; in plain bf16 the stride-2 kernel already
    # shares a CU with a second workgroup and the space-to-depth copy costs more than it saves: 60.3 -> 62.0 ms train step)
    k_ok = spec.k == 4 or (spec.k == 3 and spec.precision == PRECISION_BF16X3 and spec.cin_segments[0] % 16 == 0 and
                           not os.environ.get('APAMD_NO_S2D3'))
    return (spec.precision != PRECISION_FP32 and k_ok and spec.stride == 2 and spec.pad == 1 and
            spec.pad_mode == PAD_ZERO and not spec.transposed and len(spec.cin_segments) == 1 and
            spec.cin_segments[0] >= 32 and spec.cin_segments[0] % 8 == 0 and spec.cout >= 48 and h % 2 == 0 and
            w % 2 == 0 and spec.w_layout == W_OIHW and not spec.w_flip and not os.environ.get('APAMD_NO_S2D'))


def s2d_spec(spec):
    s = ConvSpec([4 * spec.cin_segments[0]], spec.cout, 2, 1, 0, PAD_ZERO)
    s.precision = spec.precision
    s.s2d_k = 3 if spec.k == 3 else 0
    return s


def s2d_weight(weight):
    """W[co][c][2 ty + ry][2 tx + rx] -> W'[co][(ry*2 + rx)*C + c][ty][tx]."""
    co, c, k = weight.shape[:3]
    if k == 3:                                   # a 3x3 layer: the fourth row / column of taps does not exist
        weight = torch.nn.functional.pad(weight, (0, 1, 0, 1))
    return weight.reshape(co, c, 2, 2, 2, 2).permute(0, 3, 5, 1, 2, 4).reshape(co, 4 * c, 2, 2).contiguous()


def presplit_s2d(f):
    """Space-to-depth split copy of a (virtual) feature as a split-only Feat of shape (N, 4C, H/2 + 1, W/2 + 1)."""
    x = f.data
    n, c, h, w = x.shape
    _require_device(x, 'space-to-depth source')
    s = C.ApSrc()
    s.data, s.C, s.act = x.data_ptr(), c, f.act
    if f.virtual:
        s.mean, s.rstd = f.mean.data_ptr(), f.rstd.data_ptr()
    shape = (n, 4 * c, h // 2 + 1, w // 2 + 1)
    nbytes = C.check(C.lib().ap_split_prepass_bytes(*shape), 'split_prepass_bytes')
    xs = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
    C.check(C.lib().ap_split_prepass_s2d(ctypes.byref(s), n, h, w, _ptr(xs), _stream()), 'split_prepass_s2d')
    return Feat.split_only(shape, xs)


def _alloc_xs(x):
    n, c, h, w = x.shape
    nbytes = C.check(C.lib().ap_split_prepass_bytes(n, c, h, w), 'split_prepass_bytes')
    return torch.empty(nbytes, dtype=torch.uint8, device=x.device)


def _norm_apply_split(f, residual, want_y, want_xs, xs_relu=False):
    """One ap_norm_apply_split pass over ``f``: finalises pending statistics on the way, returns the fp32
    tensor ``act(IN(f)) [+ IN(residual)]`` (want_y) and / or its split-bf16 copy (want_xs).  Without a residual the
    split copy is f's own and is cached on it."""
    x = f.data
    n, c, h, w = x.shape
    oct_src = f.oct is not None          # a channel-octet raw output (conv2d out_octet=True): split copy only, residual none / split-only
    if oct_src:
        if want_y or not want_xs or (residual is not None and not residual.is_split_only):
            raise RuntimeError('a channel-octet raw output is read as a split copy only (no fp32 result, no fp32 residual)')
        _require_device(f.oct, 'norm/split source')
    else:
        _require_device(x, 'norm/split source', allow_bf16=True)
    s = C.ApSrc()
    s.data, s.C, s.act = (f.oct if oct_src else x).data_ptr(), c, f.act
    partial, tiles, mo, ro = None, 0, None, None
    if f.pending is not None:
        partial, tiles = f.pending
        f._alloc_stats()
        mo, ro = f._mean, f._rstd
    elif f._mean is not None:
        s.mean, s.rstd = f._mean.data_ptr(), f._rstd.data_ptr()
    r = None
    res_flag = 0
    if residual is not None:
        if residual.act != ACT_NONE and residual.virtual:
            raise ValueError('a normalised residual cannot carry an activation')
        if residual.data.shape != x.shape:
            raise ValueError('residual shape mismatch')
        r = C.ApSrc()
        r.C, r.act = c, ACT_NONE
        if residual.is_split_only:
            # the residual exists only as its split copy (materialize(keep_fp32=False) of the previous block): head + tail
            if residual.xs is None or residual.xs_heads_only:
                raise RuntimeError('a split-only residual needs both planes of its split copy')
            r.data = residual.xs.data_ptr()
            res_flag = 4
        else:
            r.data = residual.data.data_ptr()
            if residual.virtual:
                r.mean, r.rstd = residual.mean.data_ptr(), residual.rstd.data_ptr()
    y = torch.empty(x.shape, dtype=torch.float32, device=x.device) if want_y else None
    xs = _alloc_xs(x) if want_xs else None
    # plain-bf16 mode: no kernel reads tail planes, so they are not written (the package-wide mode decides: split
    # copies are shared by every consumer of a feature)
    flags = ((1 if DEFAULT_PRECISION == PRECISION_BF16 else 0) | (2 if xs_relu else 0) | res_flag |
             (8 if (x.dtype == torch.bfloat16 and not oct_src) else 0) | (16 if oct_src else 0))
    C.check(C.lib().ap_norm_apply_split_ex(ctypes.byref(s), _ptr(partial), tiles, EPS, _ptr(mo), _ptr(ro),
                                           ctypes.byref(r) if r is not None else None, n, h, w, _ptr(y), _ptr(xs),
                                           flags, _stream()), 'norm_apply_split')
    f.pending = None
    if want_xs and residual is None:
        f.xs = xs
        f.xs_heads_only = bool(flags & 1)
    return y, xs


def trunk_octet_ok(c, residual=None, keep_fp32=False):
    """Inference: may a trunk convolution whose raw output is read by ONE ap_norm_apply_split pass -- split copy out, residual none
    or the previous block's split copy -- write the channel-octet layout (no LDS transposition in its epilogue, no LDS staging in
    the pass)?  Mirrors the conditions under which materialize() keeps only the split copy."""
    if (DEFAULT_PRECISION != PRECISION_BF16X3 or FUSED_NORM or not RESIDUAL_AS_SPLIT or not OCTET_TRUNK or keep_fp32 or
            not wants_split(c) or c % 8):
        return False
    return residual is None or (residual.is_split_only and residual.xs is not None and not residual.xs_heads_only and residual.oct is None)


# the inference trunk's raw outputs in the channel-octet layout (round 6; tests flip it to compare with the NCHW route)
OCTET_TRUNK = os.environ.get('APAMD_NO_OCTET_TRUNK', '0') != '1'


def wants_split(c):
    """Whether a materialised residual-trunk feature of ``c`` channels is going to be staged by a split-bf16
    convolution: its consumers are c -> c 3x3 layers, so this is the C library's eligibility rule (>= 48 outputs,
    >= 32 inputs in 16-channel segments).  A wrong guess costs the unused copy; where the fp32 tensor is dropped on the
    strength of it (materialize keep_fp32=False) the caller checks the real consumer (autograd.materialize_forward)."""
    return DEFAULT_PRECISION != PRECISION_FP32 and c >= 48 and c % 16 == 0


def takes_split(spec, n, h, w):
    """Whether the convolution ``spec`` stages its sources as split-bf16 copies at this input size."""
    if spec.precision == PRECISION_FP32:
        return False
    d = spec.desc(n, h, w)
    return bool(C.check(C.lib().ap_conv2d_wants_presplit(ctypes.byref(d)), 'wants_presplit'))


def conv2d(spec, srcs, packed, bias=None, act=ACT_NONE, want_stats=False, out_act=ACT_NONE, out_octet=False, out_bf16=False):
    """Run one convolution.  Returns a Feat:
    * want_stats=False: materialised ``act(conv + bias)``;
    * want_stats=True : raw conv output with its InstanceNorm statistics, to be consumed as
      ``out_act(IN(raw))`` by the next layer.
    out_octet: the only reader is warp_concat (inference) -- where the layer's kernel can (ap_conv2d_octet_ok), the output
    is written in the channel-octet layout [N, C/8, H*W, 8] and returned as ``.oct`` of a Feat without an NCHW tensor."""
    x0 = srcs[0].data
    n, _, h, w = x0.shape
    for f, c in zip(srcs, spec.cin_segments):
        if not f.is_split_only:
            _require_device(f.data, 'conv input', allow_bf16=True)      # (a bf16 source is legal only where it is staged as a split copy: below)
        if f.data.shape[1] != c or f.data.shape[0] != n or f.data.shape[2:] != x0.shape[2:]:
            raise ValueError('conv2d: source of shape %s does not match segment C=%d' % (tuple(f.data.shape), c))
    d = spec.desc(n, h, w, None, act)
    lib = C.lib()
    if spec.precision != PRECISION_FP32 and C.check(lib.ap_conv2d_wants_presplit(ctypes.byref(d)), 'wants_presplit'):
        # this layer runs on the split-bf16 matrix path: hand it the split copies of its sources (made, together
        # with the sources' pending InstanceNorm statistics, in one pass each)
        d.presplit = 1
        if spec.precision == PRECISION_BF16X3 and DEFAULT_PRECISION == PRECISION_BF16:
            raise RuntimeError('a split-bf16 (bf16x3) layer cannot run while the package mode is plain bf16: split copies '
                               'are then written without their tail planes')
        for i, f in enumerate(srcs):
            d.src[i].data = presplit(f, spec.precision).data_ptr()
            d.src[i].mean = d.src[i].rstd = None
            d.src[i].act = ACT_NONE
    else:
        if any(f.is_split_only for f in srcs):
            raise RuntimeError('conv2d: a source exists only as its split-bf16 copy but this layer reads fp32')
        if any(f.data.dtype != torch.float32 for f in srcs):
            raise RuntimeError('conv2d: a source holds bf16 values but this layer reads fp32')
        spec.fill_sources(d, srcs)
    ho, wo = ctypes.c_int32(), ctypes.c_int32()
    C.check(lib.ap_conv2d_out_size(ctypes.byref(d), ctypes.byref(ho), ctypes.byref(wo)), 'conv2d_out_size')
    out_octet = bool(out_octet) and d.presplit == 1 and lib.ap_conv2d_octet_ok(ctypes.byref(d)) == 1
    out_bf16 = bool(out_bf16) and not out_octet and d.presplit == 1 and lib.ap_conv2d_bf16out_ok(ctypes.byref(d)) == 1
    if out_octet:
        y = torch.empty((n, spec.cout // 8, ho.value * wo.value, 8), dtype=torch.float32, device=x0.device)
    else:
        y = torch.empty((n, spec.cout, ho.value, wo.value), dtype=torch.bfloat16 if out_bf16 else torch.float32, device=x0.device)
    partial = None
    if want_stats:
        tiles = C.check(lib.ap_conv2d_stat_tiles(ctypes.byref(d)), 'conv2d_stat_tiles')
        partial = torch.empty((n * spec.cout, tiles, 2), dtype=torch.float32, device=x0.device)
    if PROFILER is not None:
        buf = ctypes.create_string_buffer(96)
        C.check(lib.ap_conv2d_kernel_name(ctypes.byref(d), buf, 96), 'conv2d_kernel_name')
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    fwd = lib.ap_conv2d_fwd_octet if out_octet else (lib.ap_conv2d_fwd_bf16out if out_bf16 else lib.ap_conv2d_fwd)
    C.check(fwd(ctypes.byref(d), _ptr(packed), _ptr(bias), _ptr(y), _ptr(partial), _stream()), 'conv2d_fwd')
    if PROFILER is not None:
        e1.record()
        # algorithmic FLOPs = 2 * MACs of the dense operator (transposed: every input pixel x k*k taps)
        px = h * w if spec.transposed else ho.value * wo.value
        macs = getattr(spec, 'alg_macs', None) or sum(spec.cin_segments) * spec.k ** 2
        PROFILER.records.append((buf.value.decode(), 2.0 * n * px * spec.cout * macs, e0, e1))
    if out_octet:
        res = Feat(torch.empty(1, dtype=torch.float32, device=x0.device).expand((n, spec.cout, ho.value, wo.value)),
                   act=out_act if want_stats else ACT_NONE, pending=(partial, tiles) if want_stats else None)
        res.oct = y
        return res
    if not want_stats:
        return Feat(y)
    return Feat(y, act=out_act, pending=(partial, tiles))


_FNORM_FLAGS = []          # counter buffers of recent ap_conv2d_fwd_norm launches (their last element is the kernel's error flag)


_COUNTER_POOL = {}         # device -> [zero-initialised int32 pool, next free offset]


def _zero_counters(n, device):
    """n zero int32 counters for one ap_conv2d_fwd_norm launch, cut from a pool that is zeroed once (a torch.zeros per launch was
    a 5 us fill kernel in front of each of the 21 trunk convolutions).  A slice is used by exactly one launch."""
    pool = _COUNTER_POOL.get(device)
    if pool is None or pool[1] + n > pool[0].numel():
        pool = _COUNTER_POOL[device] = [torch.zeros(1 << 20, dtype=torch.int32, device=device), 0]
    out = pool[0][pool[1]:pool[1] + n]
    pool[1] += (n + 3) & ~3
    return out


# Convolution + InstanceNorm in one launch (ap_conv2d_fwd_norm) for the trunk of the generators in inference: OPT-IN.  Measured at
# B = 16 it removes the 18 norm_split passes of a forward (0.6 ms) and gives the time back in its own epilogue -- every workgroup
# bursts its (larger) output at the same moment and the matrix pipe idles meanwhile: 2131 vs 2110 and 2207 vs 2204 frames/s on two
# boxes (HISTORY.md section 3.11).  Not worth a kernel that waits on its peers by default.
FUSED_NORM = os.environ.get('APAMD_FUSED_NORM', '0') == '1'


# Inference: the ResNet trunk's residual stream is kept only as split copies (materialize keep_fp32=False; HISTORY.md section 3.10).
# (tests flip the flag to compare with the fp32 stream)
RESIDUAL_AS_SPLIT = True
# plain-bf16 arithmetic: the 7x7 edge layers' weight gradients on the bf16 matrix pipe (wgrad_k7.h); 0: the fp32 / vector-ALU kernels (A/B)
K7_WGRAD = os.environ.get('APAMD_NO_K7_WGRAD', '0') != '1'
# ... and the PatchGAN's first layer as an output stream on the matrix pipe (conv_d0.h); 0: the fp32 implicit-GEMM kernel (A/B)
D0_MFMA = K7_WGRAD and os.environ.get('APAMD_NO_D0_MFMA', '0') != '1'


def fused_norm_ok(spec, srcs):
    """Can this layer, at this shape, normalise its own output in the epilogue (ap_conv2d_fwd_norm), and is that form switched on?"""
    if not FUSED_NORM or spec.precision != PRECISION_BF16X3 or DEFAULT_PRECISION != PRECISION_BF16X3:
        return False
    n, _, h, w = srcs[0].data.shape
    d = spec.desc(n, h, w, None, ACT_NONE)
    d.presplit = 1
    return C.lib().ap_conv2d_fused_norm_ok(ctypes.byref(d)) == 1


def conv2d_norm(spec, srcs, packed, act=ACT_NONE, residual=None, want_oct=False, want_xs=True):
    """``act(InstanceNorm(conv(srcs))) [+ residual]`` in ONE launch (inference: nothing is kept for a backward pass).
    residual: a materialised Feat -- its channel-octet fp32 copy (``.oct``) when it has one, else its NCHW tensor.
    Returns a materialised Feat that exists as its split-bf16 copy (``.xs``, want_xs) and / or as channel-octet fp32 (``.oct``,
    want_oct: what the next block's residual add reads); its NCHW tensor is never written."""
    x0 = srcs[0].data
    n, _, h, w = x0.shape
    d = spec.desc(n, h, w, None, ACT_NONE)
    d.presplit = 1
    for i, f in enumerate(srcs):
        d.src[i].data = presplit(f, spec.precision).data_ptr()
        d.src[i].mean = d.src[i].rstd = None
        d.src[i].act = ACT_NONE
    lib = C.lib()
    ho, wo = ctypes.c_int32(), ctypes.c_int32()
    C.check(lib.ap_conv2d_out_size(ctypes.byref(d), ctypes.byref(ho), ctypes.byref(wo)), 'conv2d_out_size')
    dev, cout, hw = x0.device, spec.cout, ho.value * wo.value
    tiles = C.check(lib.ap_conv2d_stat_tiles(ctypes.byref(d)), 'conv2d_stat_tiles')
    nctr = C.check(lib.ap_conv2d_fused_norm_counters(ctypes.byref(d)), 'fused_norm_counters')
    fn = C.ApFusedNorm()
    fn.act, fn.eps = act, EPS
    partial = torch.empty((n * cout, tiles, 2), dtype=torch.float32, device=dev)
    counters = _zero_counters(nctr, dev)
    mean = torch.empty(n * cout, dtype=torch.float32, device=dev)
    rstd = torch.empty_like(mean)
    fn.partials, fn.counters, fn.mean, fn.rstd = partial.data_ptr(), counters.data_ptr(), mean.data_ptr(), rstd.data_ptr()
    if residual is not None:
        if residual.virtual:
            raise RuntimeError('conv2d_norm: the residual must be a materialised feature')
        if residual.oct is not None:
            fn.res_oct = residual.oct.data_ptr()
        else:
            if residual.is_split_only:
                raise RuntimeError('conv2d_norm: the residual exists only as its split-bf16 copy')
            _require_device(residual.data, 'residual')
            fn.res_nchw = residual.data.data_ptr()
    y_oct = xs = None
    if want_oct:
        y_oct = torch.empty((n, cout // 8, hw, 8), dtype=torch.float32, device=dev)
        fn.y_oct = y_oct.data_ptr()
    if want_xs:
        nbytes = C.check(lib.ap_split_prepass_bytes(n, cout, ho.value, wo.value), 'split_prepass_bytes')
        xs = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        fn.xs = xs.data_ptr()
    if PROFILER is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    C.check(lib.ap_conv2d_fwd_norm(ctypes.byref(d), _ptr(packed), ctypes.byref(fn), _stream()), 'conv2d_fwd_norm')
    if PROFILER is not None:
        e1.record()
        macs = sum(spec.cin_segments) * spec.k ** 2
        PROFILER.records.append(('Bf3Cfg<1, 3, 1, 2, 4, 4> +IN', 2.0 * n * hw * cout * macs, e0, e1))
    _FNORM_FLAGS.append(counters)
    if len(_FNORM_FLAGS) > 256:
        check_fused_norm()
    res = Feat(torch.empty(1, dtype=torch.float32, device=dev).expand((n, cout, ho.value, wo.value)))
    res.xs, res.oct = xs, y_oct
    return res


def check_fused_norm():
    """Deferred check of the error flags of the ap_conv2d_fwd_norm launches since the last call (one device read): a set flag
    means a workgroup gave up waiting for its group -- the device was shared with other work -- and the results are invalid."""
    global _FNORM_FLAGS
    flags, _FNORM_FLAGS = _FNORM_FLAGS, []
    if flags and int(torch.stack([c[-1] for c in flags]).max()) != 0:
        raise RuntimeError('animateportrait_amd: a convolution with in-kernel InstanceNorm timed out waiting for its peer workgroups '
                           '(is the GPU shared with another process or stream?); set APAMD_NO_FUSED_NORM=1')


def _conv2d_view(spec, srcs, packed, out, view):
    """ap_conv2d_fwd_view: the split-bf16 convolution ``spec`` of ``srcs`` into a window of ``out`` (no bias, no
    activation, no statistics)."""
    x0 = srcs[0].data
    n, _, h, w = x0.shape
    d = spec.desc(n, h, w, None, ACT_NONE)
    d.presplit = 1
    for i, f in enumerate(srcs):
        d.src[i].data = presplit(f, spec.precision).data_ptr()
        d.src[i].mean = d.src[i].rstd = None
        d.src[i].act = ACT_NONE
    fn = C.lib().ap_conv2d_fwd_view_bf16out if out.dtype == torch.bfloat16 else C.lib().ap_conv2d_fwd_view
    C.check(fn(ctypes.byref(d), ctypes.byref(view), _ptr(packed), None, _ptr(out), _stream()), 'conv2d_fwd_view')


def dgrad_strip_eligible(spec, g):
    """spec: data-gradient operator of a reflection-padded 3x3 stride-1 layer (full correlation, pad 2), g: the plain
    gradient Feat.  True when the padded-coordinate output is 32 m + 2 columns wide (the 64 x 64 maps of the ResNet
    blocks: 66) and the operator runs on the split-bf16 path: see conv2d_dgrad_strip."""
    if (spec.precision == PRECISION_FP32 or spec.k != 3 or spec.stride != 1 or spec.pad != 2 or spec.transposed or
            len(spec.cin_segments) != 1 or g.virtual):
        return False
    n, c, h, w = g.data.shape
    if (w + 2) % 32 != 2 or w < 32 or h < 4 or c % 8:
        return False
    d = spec.desc(n, h, w, None, ACT_NONE)
    return bool(C.check(C.lib().ap_conv2d_wants_presplit(ctypes.byref(d)), 'wants_presplit'))


def conv2d_dgrad_strip(spec, g, packed, packed_t, strip=None, out_bf16=False):
    """Padded-coordinate data gradient (N, Cin, H+2, W+2) of a reflection-padded 3x3 layer in two launches instead of a
    three-tile-column one: W+2 = 32 m + 2, so the last tile column of a plain launch would hold 2 of 32 columns.
    * the two last padded columns depend on the two last gradient columns only: the same operator on their
      TRANSPOSE (N, C, 2, H) with transposed taps (``packed_t``) yields a (4, H+2) map whose rows are the padded
      columns W-2 .. W+1; it is stored transposed into place (rows 0, 1 are partial sums of columns W-2, W-1);
    * the main launch then writes columns 0 .. W-1 (whole tile columns), overwriting those two."""
    n, c, h, w = g.data.shape
    hp, wp = h + 2, w + 2
    # out_bf16: the gradient is stored as bf16 (plain-bf16 train step; its only reader is ap_instnorm_bwd_split, which is told)
    d0 = spec.desc(n, h, w, None, ACT_NONE)
    d0.presplit = 1
    out_bf16 = bool(out_bf16) and C.lib().ap_conv2d_bf16out_ok(ctypes.byref(d0)) == 1
    out = torch.empty((n, spec.cout, hp, wp), dtype=torch.bfloat16 if out_bf16 else torch.float32, device=g.data.device)
    if strip is None and g.is_split_only:
        raise RuntimeError('conv2d_dgrad_strip: the gradient exists only as its split copy and no column strip was prepared '
                           '(ap_instnorm_bwd_split writes it when the backward plan asks for one)')
    t = strip if strip is not None else Feat(g.data[:, :, :, w - 2:].transpose(2, 3).contiguous())   # strip: (N, C, 2, H) from instnorm_bwd_split
    v = C.ApOutView()
    v.nstride, v.cstride = spec.cout * hp * wp, hp * wp
    v.rstride, v.xstride, v.y_off, v.x_off, v.OH, v.OW = 1, wp, w - 2, 0, 4, hp
    _conv2d_view(spec, [t], packed_t, out, v)
    v2 = C.ApOutView()
    v2.nstride, v2.cstride = spec.cout * hp * wp, hp * wp
    v2.rstride, v2.xstride, v2.y_off, v2.x_off, v2.OH, v2.OW = wp, 1, 0, 0, hp, w
    _conv2d_view(spec, [g], packed, out, v2)
    return out


def materialize(f, residual=None, emit_xs=None, keep_fp32=True):
    """out = act(IN(f.data)) [+ residual]; residual may itself be a virtual Feat (act must be NONE).
    emit_xs: also write the split-bf16 copy of ``out`` in the same pass (default: when a split-bf16 convolution
    can consume it).  keep_fp32=False (inference, fp32-class mode): when the split copy is written the fp32 tensor is not --
    the result exists only as its split copy (head + tail, 2^-17 relative), which is also what the next block's residual add
    reads (RESIDUAL_AS_SPLIT; the trunk's nine fp32 writes of 67 MB each at B = 16 disappear)."""
    if not f.virtual:
        raise ValueError('materialize: feature is already plain')
    n, c, h, w = f.data.shape
    if emit_xs is None:
        emit_xs = wants_split(c)
    if c % 8 == 0:
        # (with the opt-in in-kernel InstanceNorm the next block's fused epilogue reads its residual as fp32: keep it)
        split_only = (bool(emit_xs) and not keep_fp32 and RESIDUAL_AS_SPLIT and not FUSED_NORM and
                      DEFAULT_PRECISION == PRECISION_BF16X3)
        if residual is not None and residual.is_split_only and not split_only and residual.xs_heads_only:
            raise RuntimeError('materialize: the residual exists only as a head-only split copy')
        y, xs = _norm_apply_split(f, residual, want_y=not split_only, want_xs=bool(emit_xs))
        out = Feat.split_only((n, c, h, w), xs) if split_only else Feat(y)
        out.xs = xs
        out.xs_heads_only = xs is not None and DEFAULT_PRECISION == PRECISION_BF16
        return out
    x = f.data
    out = torch.empty_like(x)
    res = rm = rr = None
    if residual is not None:
        if residual.act != ACT_NONE and residual.virtual:
            raise ValueError('materialize: a normalised residual cannot carry an activation')
        res, rm, rr = residual.data, residual.mean, residual.rstd
        if res.shape != x.shape:
            raise ValueError('materialize: residual shape mismatch')
    C.check(C.lib().ap_instnorm_apply(_ptr(x), _ptr(f.mean), _ptr(f.rstd), f.act, _ptr(res), _ptr(rm), _ptr(rr),
                                      _ptr(out), n * c, h * w, _stream()), 'instnorm_apply')
    return Feat(out)


def warp_concat(f, motion, flow, ifmask, level, emit_xs=False, keep_fp32=True, s2d=False):
    """double_feature_warping (networks.py:1298-1313) on a (possibly virtual) feature map.  emit_xs: the 2C-channel
    concat is going to be staged by a split-bf16 convolution, so the kernel also writes its split copy; with
    keep_fp32=False (inference) that copy is the only output and the result is a split-only Feat.
    s2d (with emit_xs): the consumer is a stride-2 3x3 layer that runs as a 2x2 layer over the space-to-depth copy
    (s2d_eligible): the kernel writes THAT layout (zero padding ring included) and the result carries it as ``.s2d``."""
    x = f.data
    n, c, h, w = x.shape
    if f.oct is not None:
        x = f.oct                    # channel-octet layout (conv2d out_octet=True): flags bit 1
    for t, name in ((x, 'x'), (motion, 'motion'), (flow, 'flow'), (ifmask, 'ifmask')):
        _require_device(t, name)
    s = motion.shape[1]
    if motion.shape != (n, s, s, 2) or flow.shape != (n, 2, s, s) or ifmask.shape != (n, 1, s, s):
        raise ValueError('warp_concat: motion/flow/ifmask shapes %s %s %s' % (motion.shape, flow.shape, ifmask.shape))
    if h != s >> level or w != s >> level:
        raise ValueError('warp_concat: level %d expects %d px features, got %dx%d' % (level, s >> level, h, w))
    emit_xs = bool(emit_xs) and c % 8 == 0
    s2d = bool(s2d) and emit_xs and h % 2 == 0 and w % 2 == 0
    out = xs = None
    if keep_fp32 or not emit_xs:
        out = torch.empty((n, 2 * c, h, w), dtype=torch.float32, device=x.device)
    s2d_shape = (n, 8 * c, h // 2 + 1, w // 2 + 1)
    if emit_xs:
        nbytes = C.check(C.lib().ap_split_prepass_bytes(*(s2d_shape if s2d else (n, 2 * c, h, w))), 'split_prepass_bytes')
        xs = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
    C.check(C.lib().ap_warp_concat_fwd_ex(_ptr(x), _ptr(f.mean), _ptr(f.rstd), f.act, _ptr(motion), _ptr(flow),
                                          _ptr(ifmask), _ptr(out), _ptr(xs), n, c, h, w, s, 1.0 / (1 << level),
                                          (1 if s2d else 0) | (2 if f.oct is not None else 0), _stream()), 'warp_concat_fwd')
    if s2d:
        # the plain split copy does not exist: only the stride-2 consumer (through .s2d) or fp32 readers can use this
        res = Feat(out) if out is not None else Feat(torch.empty(1, dtype=torch.float32, device=x.device).expand((n, 2 * c, h, w)))
        res.s2d = Feat.split_only(s2d_shape, xs)
        return res
    if out is None:
        return Feat.split_only((n, 2 * c, h, w), xs)
    res = Feat(out)
    res.xs = xs
    return res


def resize_bilinear(x, size):
    """F.interpolate(x, size, mode='bilinear', align_corners=False) (geomcgt_ifw_test_model.py:283, 285)."""
    _require_device(x, 'resize input')
    n, c, h, w = x.shape
    oh, ow = size
    y = torch.empty((n, c, oh, ow), dtype=torch.float32, device=x.device)
    C.check(C.lib().ap_resize_bilinear(_ptr(x), n * c, h, w, oh, ow, _ptr(y), _stream()), 'resize_bilinear')
    return y


def pixel_shuffle2(x):
    """nn.PixelShuffle(2) (intrinsic_flow_models/networks.py:696): (N, 4C, H, W) -> (N, C, 2H, 2W)."""
    _require_device(x, 'pixel_shuffle input')
    n, c4, h, w = x.shape
    if c4 % 4:
        raise ValueError('pixel_shuffle2: %d channels' % c4)
    y = torch.empty((n, c4 // 4, 2 * h, 2 * w), dtype=torch.float32, device=x.device)
    C.check(C.lib().ap_pixel_shuffle2(_ptr(x), n, c4 // 4, h, w, _ptr(y), _stream()), 'pixel_shuffle2')
    return y


def grid_sample(x, grid, align_corners=False):
    """F.grid_sample(x, grid, mode='bilinear', padding_mode='zeros', align_corners=...) (geomcgt_ifw_test_model.py:294)."""
    _require_device(x, 'grid_sample input')
    _require_device(grid, 'grid_sample grid')
    n, c, h, w = x.shape
    if grid.dim() != 4 or grid.shape[0] != n or grid.shape[3] != 2:
        raise ValueError('grid_sample: grid of shape %s for input %s' % (tuple(grid.shape), tuple(x.shape)))
    oh, ow = grid.shape[1], grid.shape[2]
    y = torch.empty((n, c, oh, ow), dtype=torch.float32, device=x.device)
    C.check(C.lib().ap_grid_sample(_ptr(x), _ptr(grid), n, c, h, w, oh, ow, int(bool(align_corners)), _ptr(y), _stream()),
            'grid_sample')
    return y


# =============================================================================== backward ops
def _grad_out(out, shape, device):
    """The tensor a gradient kernel writes: the caller's slot (a contiguous view of a gradient block) or a new one."""
    if out is None:
        return torch.empty(shape, dtype=torch.float32, device=device)
    if tuple(out.shape) != tuple(shape) or not out.is_contiguous():
        raise ValueError('gradient slot of shape %s for a gradient of shape %s' % (tuple(out.shape), tuple(shape)))
    return out


def wgrad(k, stride, pad, pad_mode, g, srcs, out_shape, precision=None, out=None, g_t=None, g_xs=None):
    """Weight gradient (see include/animateportrait_amd.h: ap_conv2d_wgrad).  g: Feat of the M-role tensor,
    srcs: Feats of the shifted tensor's segments.  Returns a tensor of ``out_shape`` (OIHW / IOHW): ``out`` when given
    (the layer's slot in the network's contiguous gradient block), else a new tensor.
    precision: PRECISION_* (default: the package default, i.e. split-bf16 for the wide stride-1 layers).
    g_t: the M-role operand as instnorm_bwd_split wrote it (wgrad_gt_dims); ``g`` then only carries the shape.
    g_xs: the split copy of the gradient (instnorm_bwd_split's ``xs``): where the layer is served by ap_conv2d_wgrad_xs (both operands
    read as the convolutions' split copies, no preparation) that route is taken; ``g`` again only carries the shape."""
    n, m, gh, gw = g.data.shape
    cin = sum(f.data.shape[1] for f in srcs)
    if (m == 1 and k == 4 and stride == 1 and len(srcs) == 1 and cin >= 64 and not g.virtual and g.act == ACT_NONE and
            tuple(out_shape) == (1, cin, k, k) and srcs[0].data.shape[2] <= 32 and srcs[0].data.shape[3] <= 31 and
            pad == 1 and pad_mode == PAD_ZERO):
        # PatchGAN output layer: one workgroup per input channel (conv_head.h)
        f = srcs[0]
        _require_device(f.data, 'wgrad source')
        _require_device(g.data, 'wgrad gradient')
        s = C.ApSrc()
        s.data, s.C, s.act = f.data.data_ptr(), cin, f.act
        if f.virtual:
            s.mean, s.rstd = f.mean.data_ptr(), f.rstd.data_ptr()
        dw = _grad_out(out, out_shape, g.data.device)
        C.check(C.lib().ap_conv_head_wgrad(ctypes.byref(s), _ptr(g.data), n, f.data.shape[2], f.data.shape[3], k, pad,
                                           _ptr(dw), _stream()), 'conv_head_wgrad')
        return dw
    prec = DEFAULT_PRECISION if precision is None else precision
    if (K7_WGRAD and prec == PRECISION_BF16 and k == 7 and stride == 1 and pad == 3 and pad_mode == PAD_REFLECT and len(srcs) == 1 and
            g_t is None and g_xs is None and not g.virtual and g.act == ACT_NONE and srcs[0].data.dtype == torch.float32):
        # the 7x7 edge layers at full resolution, plain-bf16 arithmetic: one pass over the wide tensor on the bf16 matrix pipe
        # (wgrad_k7.h) -- the stems (wide = the gradient) and the last layer (wide = the input)
        f = srcs[0]
        h, w = f.data.shape[2:]
        final_form = 1 if (m == 1 and cin >= 32) else 0
        wide, narrow = (f, g) if final_form else (g, f)
        ok = ((final_form or not f.virtual and f.act == ACT_NONE) and tuple(out_shape) == (m, cin, k, k) and
              (g.data.dtype == torch.float32 or not final_form) and
              C.lib().ap_wgrad_k7_bf16_ok(n, wide.data.shape[1], narrow.data.shape[1], h, w, final_form) == 1)
        if ok:
            _require_device(f.data, 'wgrad source')
            _require_device(g.data, 'wgrad gradient', allow_bf16=not final_form)
            sw, sn = C.ApSrc(), C.ApSrc()
            # ap_src.act bit 8: the stems' gradient as instnorm_bwd(out_bf16=True) stored it
            sw.data, sw.C, sw.act = wide.data.data_ptr(), wide.data.shape[1], wide.act | (0x100 if wide.data.dtype == torch.bfloat16 else 0)
            if wide.virtual:
                sw.mean, sw.rstd = wide.mean.data_ptr(), wide.rstd.data_ptr()
            sn.data, sn.C, sn.act = narrow.data.data_ptr(), narrow.data.shape[1], ACT_NONE
            ws = torch.empty(C.check(C.lib().ap_wgrad_k7_bf16_workspace_floats(n, sw.C, sn.C, h, w, final_form), 'wgrad_k7_ws'),
                             dtype=torch.float32, device=g.data.device)
            dw = _grad_out(out, out_shape, g.data.device)
            if PROFILER is not None:
                PROFILER.note('wgrad_k7<%s>' % ('final' if final_form else 'stem'))
            C.check(C.lib().ap_wgrad_k7_bf16(ctypes.byref(sw), ctypes.byref(sn), n, h, w, final_form, _ptr(ws), _ptr(dw), _stream()),
                    'wgrad_k7_bf16')
            return dw
    if g.data.dtype == torch.bfloat16 and g_t is None and g_xs is None:
        g = Feat(g.data.float(), g._mean, g._rstd, g.act)       # (only wgrad_k7 reads a bf16-stored gradient)
    if (D0_MFMA and prec == PRECISION_BF16 and k == 4 and stride == 2 and pad == 1 and pad_mode == PAD_ZERO and len(srcs) == 1 and
            g_t is None and g_xs is None and not g.virtual and g.act == ACT_NONE and not srcs[0].virtual and srcs[0].act == ACT_NONE and
            not srcs[0].is_split_only and srcs[0].data.dtype == torch.float32 and tuple(out_shape) == (m, cin, k, k)):
        # the PatchGAN's first layer, plain-bf16 arithmetic: one pass over the gradient on the bf16 matrix pipe (wgrad_k7.h, form 2)
        f = srcs[0]
        h, w = f.data.shape[2:]
        if (gh, gw) == (h // 2, w // 2) and C.lib().ap_wgrad_d0_bf16_ok(n, m, cin, h, w) == 1:
            _require_device(f.data, 'wgrad source')
            _require_device(g.data, 'wgrad gradient')
            ws = torch.empty(C.check(C.lib().ap_wgrad_d0_bf16_workspace_floats(n, m, cin, h, w), 'wgrad_d0_ws'),
                             dtype=torch.float32, device=g.data.device)
            dw = _grad_out(out, out_shape, g.data.device)
            if PROFILER is not None:
                PROFILER.note('wgrad_k7<d0>')
            C.check(C.lib().ap_wgrad_d0_bf16(_ptr(g.data), _ptr(f.data), n, m, cin, h, w, _ptr(ws), _ptr(dw), _stream()), 'wgrad_d0_bf16')
            return dw
    if (m == 1 and k == 7 and stride == 1 and pad == 3 and len(srcs) == 1 and cin >= 16 and not g.virtual and
            g.act == ACT_NONE and tuple(out_shape) == (1, cin, k, k)):
        # the generator's last layer: vector-ALU kernel, window through LDS (wgrad_final.h)
        f = srcs[0]
        _require_device(f.data, 'wgrad source')
        _require_device(g.data, 'wgrad gradient')
        s = C.ApSrc()
        s.data, s.C, s.act = f.data.data_ptr(), cin, f.act
        if f.virtual:
            s.mean, s.rstd = f.mean.data_ptr(), f.rstd.data_ptr()
        h, w = f.data.shape[2:]
        ws = torch.empty(C.check(C.lib().ap_conv_final_wgrad_workspace_floats(n, cin, h, w), 'conv_final_wgrad_ws'),
                         dtype=torch.float32, device=g.data.device)
        dw = _grad_out(out, out_shape, g.data.device)
        C.check(C.lib().ap_conv_final_wgrad(ctypes.byref(s), _ptr(g.data), n, h, w, k, pad, pad_mode, _ptr(ws), _ptr(dw),
                                            _stream()), 'conv_final_wgrad')
        return dw
    if m <= 4 and stride == 1 and cin >= 16 and tuple(out_shape) == (m, cin, k, k) and 2 * pad == k - 1:
        dw = _wgrad_few_outputs(k, pad, pad_mode, g, srcs, out_shape)
        return dw if out is None else out.copy_(dw)
    d = _wgrad_desc(k, stride, pad, pad_mode, (n, m, gh, gw), None if (g_t is not None or g_xs is not None) else g, srcs, precision)
    lib = C.lib()
    nws = C.check(lib.ap_conv2d_wgrad_workspace_floats(ctypes.byref(d)), 'wgrad_workspace_floats')
    ws = torch.empty(nws, dtype=torch.float32, device=srcs[0].data.device)
    dw = _grad_out(out, out_shape, srcs[0].data.device)
    assert dw.numel() == m * cin * k * k
    if g_xs is not None:
        if lib.ap_conv2d_wgrad_xs_ok(ctypes.byref(d)) != 1:
            raise RuntimeError('wgrad: the split copy of the gradient was passed but the layer is not served by ap_conv2d_wgrad_xs')
        if PROFILER is not None:
            PROFILER.note('wgrad_xs<%d> (split copies)' % k)
        C.check(lib.ap_conv2d_wgrad_xs(ctypes.byref(d), _ptr(g_xs), _ptr(ws), _ptr(dw), _stream()), 'conv2d_wgrad_xs')
    elif g_t is not None:
        if PROFILER is not None:
            PROFILER.note('wgrad_bf16x3<%d> (prepared operand)' % k)  # only that kernel takes the operand instnorm_bwd_split wrote
        C.check(lib.ap_conv2d_wgrad_pre(ctypes.byref(d), _ptr(g_t), _ptr(ws), _ptr(dw), _stream()), 'conv2d_wgrad_pre')
    else:
        C.check(lib.ap_conv2d_wgrad(ctypes.byref(d), _ptr(ws), _ptr(dw), _stream()), 'conv2d_wgrad')
    return dw


def _wgrad_desc(k, stride, pad, pad_mode, g_shape, g, srcs, precision):
    """ap_wgrad_desc of a layer; g None: the M-role operand comes prepared (ap_conv2d_wgrad_pre) and its pointers stay empty."""
    n, m, gh, gw = g_shape
    d = C.ApWgradDesc()
    d.N, d.M, d.GH, d.GW = n, m, gh, gw
    d.H, d.W = srcs[0].data.shape[2], srcs[0].data.shape[3]
    d.K, d.stride, d.pad, d.pad_mode = k, stride, pad, pad_mode
    d.nsrc = len(srcs)
    d.precision = DEFAULT_PRECISION if precision is None else precision
    d.g.C, d.g.act = m, ACT_NONE
    if g is not None:
        _require_device(g.data, 'wgrad gradient')
        d.g.data = g.data.data_ptr()
        d.g.mean = g.mean.data_ptr() if g.mean is not None else None
        d.g.rstd = g.rstd.data_ptr() if g.rstd is not None else None
        d.g.act = g.act
    for i, f in enumerate(srcs):
        _require_device(f.data, 'wgrad source', allow_bf16=True)
        d.src[i].data = f.data.data_ptr()
        d.src[i].mean = f.mean.data_ptr() if f.mean is not None else None
        d.src[i].rstd = f.rstd.data_ptr() if f.rstd is not None else None
        # ap_src.act bit 8: the segment holds bf16 values (the C side refuses it where its kernels cannot read them)
        d.src[i].C, d.src[i].act = f.data.shape[1], f.act | (0x100 if f.data.dtype == torch.bfloat16 else 0)
    # the split copies the forward pass staged of the same sources (still alive on the tape): the C side re-tiles the shifted
    # operand from them instead of normalising + splitting the fp32 tensors again (ap_wgrad_desc.src_xs)
    if XS_WGRAD:
        heads = DEFAULT_PRECISION == PRECISION_BF16 or any(f.xs_heads_only for f in srcs)
        d.xs_parts = 1 if heads else 2
        if all(f.xs is not None for f in srcs):
            for i, f in enumerate(srcs):
                d.src_xs[i] = f.xs.data_ptr()
        if len(srcs) == 1 and srcs[0].s2d is not None and srcs[0].s2d.xs is not None:
            d.src_xs_s2d = srcs[0].s2d.xs.data_ptr()
    return d


def wgrad_xs_ok(k, stride, pad, pad_mode, g_shape, srcs, precision=None):
    """Is this weight gradient served with both operands as split copies (ap_conv2d_wgrad_xs)?  Needs the sources' forward copies."""
    if not XS_DIRECT or g_shape[1] <= 4:
        return False
    # the answer depends on the layer geometry and on WHICH forward copies the sources carry, not on their contents: cached, so
    # that the backward pass of a launch-bound train step does not rebuild a descriptor and re-plan per layer and step
    key = (k, stride, pad, pad_mode, tuple(g_shape), precision,
           tuple((tuple(f.data.shape), f.xs is not None, f.s2d is not None, f.is_split_only) for f in srcs))
    hit = _WGRAD_XS_OK.get(key)
    if hit is None:
        d = _wgrad_desc(k, stride, pad, pad_mode, g_shape, None, srcs, precision)
        hit = _WGRAD_XS_OK[key] = C.lib().ap_conv2d_wgrad_xs_ok(ctypes.byref(d)) == 1
    return hit


_WGRAD_XS_OK = {}


def wgrad_gt_dims(k, stride, pad, pad_mode, g_shape, srcs, precision=None):
    """(rows, pixel octets per row, padded channels) of the M-role operand when this weight gradient runs on the bf16 matrix kernel
    and therefore takes the operand prepared by its producer (instnorm_bwd_split -> wgrad(..., g_t=)); None when it does not."""
    if g_shape[1] <= 4:
        return None          # the narrow-output special cases of wgrad()
    d = _wgrad_desc(k, stride, pad, pad_mode, g_shape, None, srcs, precision)
    dims = (ctypes.c_int32 * 3)()
    if C.check(C.lib().ap_conv2d_wgrad_gt_dims(ctypes.byref(d), dims), 'wgrad_gt_dims') != 1:
        return None
    return tuple(dims)


def pad_materialize(srcs, pad, pad_mode, hp=None, wp=None):
    """Padded, concatenated, normalised + activated copy of the (virtual) sources: (N, sum C, Hp, Wp)."""
    x0 = srcs[0].data
    n, _, h, w = x0.shape
    hp = h + 2 * pad if hp is None else hp
    wp = w + 2 * pad if wp is None else wp
    arr = (C.ApSrc * len(srcs))()
    for i, f in enumerate(srcs):
        _require_device(f.data, 'pad_materialize source')
        arr[i].data = f.data.data_ptr()
        arr[i].mean = f.mean.data_ptr() if f.mean is not None else None
        arr[i].rstd = f.rstd.data_ptr() if f.rstd is not None else None
        arr[i].C, arr[i].act = f.data.shape[1], f.act
    out = torch.empty((n, sum(f.data.shape[1] for f in srcs), hp, wp), dtype=torch.float32, device=x0.device)
    C.check(C.lib().ap_pad_materialize(arr, len(srcs), n, h, w, pad, pad_mode, hp, wp, _ptr(out), _stream()),
            'pad_materialize')
    return out


def _wgrad_few_outputs(k, pad, pad_mode, g, srcs, out_shape):
    """Weight gradient of a 'same' convolution with 1..4 output channels (the generator's last layer,
    networks.py:1277-1279).  With M = Cout the MFMA rows would be >96 % padding, so the roles are swapped:
        dW[co][ci][ky][kx] = sum_p' a_pad[ci][p'] * g[co][p' + (k-1-ky) - (k-1)]
    i.e. a weight gradient with M-role = the PADDED input (Cin rows), shifted tensor = the one-channel gradient,
    zero padding k-1, and mirrored taps."""
    a_pad = pad_materialize(srcs, pad, pad_mode)                      # (N, Cin, H+2p, W+2p), plain
    m = g.data.shape[1]
    cin = a_pad.shape[1]
    outs = []
    for co in range(m):
        gc = g.data[:, co:co + 1].contiguous() if m > 1 else g.data
        dwt = wgrad(k, 1, k - 1, PAD_ZERO, Feat(a_pad), [Feat(gc)], (cin, 1, k, k))   # [ci][0][ky'][kx']
        outs.append(dwt.flip(2, 3).permute(1, 0, 2, 3))
    return torch.cat(outs, 0).contiguous().view(out_shape)


def _split_contribs(contribs):
    """contribs: list of (tensor, pad).  Reduce to (g1, pad1, g2) with g2 plain, folding the extras."""
    plain = [t for t, p in contribs if p == 0]
    padded = [(t, p) for t, p in contribs if p > 0]
    while True:
        if len(padded) > 1:
            t, p = padded.pop()
            plain.append(fold_add(t, p, plain.pop() if plain else None))
            continue
        if len(plain) > (1 if padded else 2):
            a, b = plain.pop(), plain.pop()
            plain.append(fold_add(a, 0, b))
            continue
        break
    if padded:
        return padded[0][0], padded[0][1], (plain[0] if plain else None)
    return plain[0], 0, (plain[1] if len(plain) > 1 else None)


def _as_fp32_grad(t):
    """A gradient stored as bf16 (conv2d_dgrad_strip(out_bf16=...)) has ONE reader that understands it, ap_instnorm_bwd_split.
    Every fp32 kernel that is handed such a tensor would read a half-sized buffer as floats: convert (once) instead."""
    return t.float() if (t is not None and t.dtype == torch.bfloat16) else t


def fold_add(g1, pad, g2):
    """out = fold(g1) + g2 (reflection-pad backward fused)."""
    g1, g2 = _as_fp32_grad(g1), _as_fp32_grad(g2)
    n, c, hp, wp = g1.shape
    h, w = hp - 2 * pad, wp - 2 * pad
    out = torch.empty((n, c, h, w), dtype=torch.float32, device=g1.device)
    C.check(C.lib().ap_act_bwd(_ptr(g1), pad, _ptr(g2), None, ACT_NONE, n * c, h, w, _ptr(out), _stream()), 'act_bwd')
    return out


def instnorm_bwd(contribs, f, out_bf16=False):
    """Gradient w.r.t. the raw conv output y of the virtual feature f = act(IN(y)).
    out_bf16: the caller's ONLY reader of dy is the stems' weight gradient on the bf16 matrix pipe (k7_stem_wgrad_ok): where the
    big-plane kernel serves the call, dy is stored as bf16 (half the bytes written and read back)."""
    g1, pad, g2 = _split_contribs(contribs)
    g1, g2 = _as_fp32_grad(g1), _as_fp32_grad(g2)
    n, c, h, w = f.data.shape
    b16 = bool(out_bf16) and pad == 0 and 16384 < h * w <= 65536 and w % 4 == 0
    dy = torch.empty(f.data.shape, dtype=torch.bfloat16 if b16 else torch.float32, device=f.data.device)
    ws = torch.empty(n * c * 2, dtype=torch.float32, device=dy.device)
    C.check(C.lib().ap_instnorm_bwd(_ptr(g1), pad, _ptr(g2), _ptr(f.data), _ptr(f.mean), _ptr(f.rstd), f.act | (0x100 if b16 else 0),
                                    n * c, h, w, _ptr(ws), _ptr(dy), _stream()), 'instnorm_bwd')
    return dy


def conv_d0_ok(spec, srcs, act):
    """Is this layer the PatchGAN's first (1 | 2 -> 64 channels, 4x4 stride 2 on a plain image) in plain-bf16 arithmetic, served as
    an output stream on the bf16 matrix pipe (ap_conv_d0_fwd_bf16)?"""
    if not (D0_MFMA and DEFAULT_PRECISION == PRECISION_BF16 and spec.precision == PRECISION_BF16 and not spec.transposed and
            spec.k == 4 and spec.stride == 2 and spec.pad == 1 and spec.pad_mode == PAD_ZERO and len(srcs) == 1 and act in (ACT_NONE, ACT_RELU, ACT_LRELU)):
        return False
    f = srcs[0]
    if f.is_split_only or f.virtual or f.act != ACT_NONE or f.data.dtype != torch.float32:
        return False
    n, c, h, w = f.data.shape
    return C.lib().ap_conv_d0_fwd_bf16_ok(n, c, spec.cout, h, w) == 1


def conv_d0(f, weight, bias, act):
    n, c, h, w = f.data.shape
    _require_device(f.data, 'conv_d0 input')
    y = torch.empty((n, weight.shape[0], h // 2, w // 2), dtype=torch.float32, device=f.data.device)
    wt = weight.detach().contiguous()
    if PROFILER is not None:
        PROFILER.note('conv_d0<%d>' % c)
    C.check(C.lib().ap_conv_d0_fwd_bf16(_ptr(f.data), _ptr(wt), _ptr(bias), n, c, weight.shape[0], h, w, act, _ptr(y), _stream()),
            'conv_d0_fwd_bf16')
    return Feat(y)


def final_dgrad_k7_ok(spec, g, f):
    """Is the data gradient of this layer the last layer's, served on the bf16 matrix pipe (ap_conv_final_dgrad_bf16)?"""
    n, m, h, w = g.data.shape
    return (K7_WGRAD and DEFAULT_PRECISION == PRECISION_BF16 and spec.precision == PRECISION_BF16 and not spec.transposed and
            spec.k == 7 and spec.stride == 1 and spec.pad == 3 and spec.pad_mode == PAD_REFLECT and m == 1 and not g.virtual and
            g.act == ACT_NONE and g.data.dtype == torch.float32 and tuple(f.data.shape[2:]) == (h, w) and
            C.lib().ap_conv_final_dgrad_bf16_ok(n, f.data.shape[1], h, w) == 1)


def final_dgrad_k7(g, weight):
    """Gradient w.r.t. the reflection-padded input of Conv2d(C, 1, 7): (N, C, H + 6, W + 6), to be folded with pad 3."""
    n, _, h, w = g.data.shape
    c = weight.shape[1]
    _require_device(g.data, 'final dgrad gradient')
    wt = weight.detach().contiguous()
    ws = torch.empty(C.check(C.lib().ap_conv_final_dgrad_bf16_workspace_floats(n, c, h, w), 'conv_final_dgrad_ws'),
                     dtype=torch.float32, device=g.data.device)
    gp = torch.empty((n, c, h + 6, w + 6), dtype=torch.float32, device=g.data.device)
    if PROFILER is not None:
        PROFILER.note('dgrad_k7<final>')
    C.check(C.lib().ap_conv_final_dgrad_bf16(_ptr(g.data), _ptr(wt), n, c, h, w, _ptr(ws), _ptr(gp), _stream()), 'conv_final_dgrad_bf16')
    return gp


def head_dgrad_ok(spec, g, f):
    """Is the data gradient of this layer the PatchGAN output layer's, served on the bf16 matrix pipe (ap_conv_head_dgrad_bf16)?"""
    n, m, gh, gw = g.data.shape
    h, w = f.data.shape[2:]
    return (D0_MFMA and DEFAULT_PRECISION == PRECISION_BF16 and spec.precision == PRECISION_BF16 and not spec.transposed and
            spec.k == 4 and spec.stride == 1 and spec.pad == 1 and spec.pad_mode == PAD_ZERO and m == 1 and not g.virtual and
            g.act == ACT_NONE and g.data.dtype == torch.float32 and (gh, gw) == (h - 1, w - 1) and
            C.lib().ap_conv_head_dgrad_bf16_ok(n, f.data.shape[1], h, w) == 1)


def head_dgrad(g, weight, h, w):
    """Gradient w.r.t. the input of Conv2d(C, 1, 4, 1, 1): (N, C, h, w)."""
    n = g.data.shape[0]
    c = weight.shape[1]
    _require_device(g.data, 'head dgrad gradient')
    wt = weight.detach().contiguous()
    gx = torch.empty((n, c, h, w), dtype=torch.float32, device=g.data.device)
    if PROFILER is not None:
        PROFILER.note('dgrad_head')
    C.check(C.lib().ap_conv_head_dgrad_bf16(_ptr(g.data), _ptr(wt), n, c, h, w, _ptr(gx), _stream()), 'conv_head_dgrad_bf16')
    return gx


def k7_stem_wgrad_ok(spec, g_shape, srcs):
    """Does wgrad() serve this layer's weight gradient in the stem form of ap_wgrad_k7_bf16 (which also reads a bf16-stored gradient)?"""
    n, m, h, w = g_shape
    f = srcs[0]
    return (K7_WGRAD and DEFAULT_PRECISION == PRECISION_BF16 and spec.precision == PRECISION_BF16 and not spec.transposed and
            spec.k == 7 and spec.stride == 1 and spec.pad == 3 and spec.pad_mode == PAD_REFLECT and len(srcs) == 1 and
            not f.virtual and f.act == ACT_NONE and f.data.dtype == torch.float32 and tuple(f.data.shape[2:]) == (h, w) and
            not (m == 1 and f.data.shape[1] >= 32) and C.lib().ap_wgrad_k7_bf16_ok(n, m, f.data.shape[1], h, w, 0) == 1)


def instnorm_bwd_split_ok(f, pad):
    n, c, h, w = f.data.shape
    return C.lib().ap_instnorm_bwd_split_ok(c, h, w, pad) == 1


def instnorm_bwd_split(red, f, gt_dims=None, want_xs=True, want_strip=False, want_dy=False):
    """instnorm_bwd for a layer whose gradient only feeds the bf16 matrix kernels (ap_instnorm_bwd_split): no fp32 dy; the kernel
    writes the split copy the data-gradient convolution stages, the weight gradient's M-role operand (gt_dims from wgrad_gt_dims)
    and, for conv2d_dgrad_strip, the transposed two-column strip.  red = _split_contribs(contribs).
    Returns (gradient Feat that exists as its split copy, gt or None, strip Feat or None)."""
    g1, pad, g2 = red
    n, c, h, w = f.data.shape
    dev = f.data.device
    heads_only = DEFAULT_PRECISION == PRECISION_BF16
    xs = _alloc_xs(f.data) if want_xs else None
    gt = dims = None
    if gt_dims is not None:
        ghp, gx8, mp = gt_dims
        alloc = torch.empty if (ghp == h and gx8 * 8 == w and mp == c) else torch.zeros
        gt = alloc(n * 2 * ghp * gx8 * mp * 16, dtype=torch.uint8, device=dev)
        dims = (ctypes.c_int32 * 3)(ghp, gx8, mp)
    strip = torch.empty((n, c, 2, h), dtype=torch.float32, device=dev) if want_strip else None
    dy = torch.empty(f.data.shape, dtype=torch.float32, device=dev) if want_dy else None
    # bit 1: y holds bf16 values (ap_conv2d_fwd_bf16out); bit 2: so does the gradient g1 (ap_conv2d_fwd_view_bf16out)
    flags = (1 if heads_only else 0) | (2 if f.data.dtype == torch.bfloat16 else 0) | (4 if g1.dtype == torch.bfloat16 else 0)
    if g2 is not None and g2.dtype != torch.float32:
        raise RuntimeError('instnorm_bwd_split: only the first gradient contribution may hold bf16 values')
    if PROFILER is not None:
        PROFILER.note('instnorm_bwd_split<%d>' % (h * w // 4))       # threads per (image, channel octet) item: 4 pixels each
    C.check(C.lib().ap_instnorm_bwd_split(_ptr(g1), pad, _ptr(g2), _ptr(f.data), _ptr(f.mean), _ptr(f.rstd), f.act, n, c, h, w,
                                          _ptr(xs), _ptr(gt), dims, _ptr(strip), _ptr(dy), flags, _stream()),
            'instnorm_bwd_split')
    if dy is not None:
        gf = Feat(dy)
        gf.xs = xs
    else:
        gf = Feat.split_only((n, c, h, w), xs)
    gf.xs_heads_only = heads_only
    return gf, gt, (Feat(strip) if strip is not None else None)


def act_bwd(contribs, out, act):
    """Gradient w.r.t. the pre-activation of a plain layer output ``out = act(pre)``."""
    g1, pad, g2 = _split_contribs(contribs)
    g1, g2 = _as_fp32_grad(g1), _as_fp32_grad(g2)
    if act == ACT_NONE and pad == 0 and g2 is None:
        return g1
    n, c, h, w = out.shape
    dy = torch.empty_like(out)
    C.check(C.lib().ap_act_bwd(_ptr(g1), pad, _ptr(g2), _ptr(out), act, n * c, h, w, _ptr(dy), _stream()), 'act_bwd')
    return dy


def act_bwd_bias(contribs, out, act, db_out=None):
    """act_bwd of a layer without InstanceNorm AND its bias gradient (sum of dy over n, y, x) in one pass: (dy, db).
    Served for one plain-or-pad-1 contribution pattern that ap_act_bwd's general kernel takes; otherwise the two operators separately."""
    g1, pad, g2 = _split_contribs(contribs)
    g1, g2 = _as_fp32_grad(g1), _as_fp32_grad(g2)
    n, c, h, w = out.shape
    fused = not (pad == 1 and w % 4 == 0 and h >= 3 and w >= 4) and n * c <= 65535 and h * w >= 4096
    if not fused:
        dy = act_bwd(contribs, out, act)
        return dy, bias_grad(dy, out=db_out)
    dy = torch.empty_like(out)
    db = _grad_out(db_out, (c,), out.device)
    ws = torch.empty(C.check(C.lib().ap_act_bwd_bias_workspace_floats(n, c, h, w), 'act_bwd_bias_ws'), dtype=torch.float32, device=out.device)
    C.check(C.lib().ap_act_bwd_bias(_ptr(g1), pad, _ptr(g2), _ptr(out), act, n, c, h, w, _ptr(dy), _ptr(ws), _ptr(db), _stream()), 'act_bwd_bias')
    return dy, db


def bias_grad(dy, out=None):
    n, c, h, w = dy.shape
    db = _grad_out(out, (c,), dy.device)
    lib = C.lib()
    if c >= 1024:
        C.check(lib.ap_bias_grad(_ptr(dy), n, c, h * w, _ptr(db), _stream()), 'bias_grad')
    else:   # one workgroup per channel leaves the GPU idle below ~1 K channels (256 channels x 32 x 64^2: 210 us, 0.6 TB/s):
            # slices of every (n, c) plane in parallel, then a fixed-order sum per channel
        nws = C.check(lib.ap_bias_grad_workspace_floats(n, c, h * w), 'bias_grad_workspace_floats')
        ws = torch.empty(nws, dtype=torch.float32, device=dy.device)
        C.check(lib.ap_bias_grad_ws(_ptr(dy), n, c, h * w, _ptr(ws), _ptr(db), _stream()), 'bias_grad')
    return db


def warp_concat_bwd(gout, motion, flow, ifmask, level):
    n, c2, h, w = gout.shape
    c = c2 // 2
    s = motion.shape[1]
    dx = torch.empty((n, c, h, w), dtype=torch.float32, device=gout.device)
    C.check(C.lib().ap_warp_concat_bwd(_ptr(gout), _ptr(motion), _ptr(flow), _ptr(ifmask), _ptr(dx), n, c, h, w, s,
                                       1.0 / (1 << level), _stream()), 'warp_concat_bwd')
    return dx
