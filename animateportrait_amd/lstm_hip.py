"""nn.LSTM forward (inference) with the recurrence on libapamd.so: per layer and direction ONE library GEMM for the input
projections of all time steps and ONE launch of ``ap_lstm_recurrence`` (csrc/lstm.hip) for the time loop -- instead of a few tiny
library kernels per time step (the AutoVC converter's ~830-frame sequences: ~11 k launches, 0.20 s of host time per 10 s clip).

Same parameters, same gate order, same results as ``nn.LSTM`` (tests/test_module1_gpu.py holds it to the library op on the device).
Falls back to the module itself where the kernel does not apply (CPU tensors, training mode, projections, unsupported sizes)."""
import ctypes

import torch

from . import _capi as C
from . import ops

_WS = {}


def _workspace(h, dev):
    key = (h, str(dev))
    ws = _WS.get(key)
    if ws is None:
        n = int(C.lib().ap_lstm_workspace_bytes(h))
        ws = _WS[key] = torch.zeros(n, dtype=torch.uint8, device=dev)
    return ws


def supported(lstm, x):
    h = lstm.hidden_size
    return (isinstance(lstm, torch.nn.LSTM) and x.is_cuda and x.dtype == torch.float32 and not lstm.training and lstm.batch_first and
            lstm.bias and getattr(lstm, 'proj_size', 0) == 0 and x.dim() == 3 and not torch.is_grad_enabled() and
            (h <= 64 or (h in (256, 512) and x.shape[0] == 1)))


def lstm_forward(lstm, x):
    """(B, T, I) -> (B, T, D * H): ``lstm(x)[0]`` with zero initial state."""
    if not supported(lstm, x):
        return lstm(x)[0]
    lib = C.lib()
    b, t, _ = x.shape
    h, nd = lstm.hidden_size, 2 if lstm.bidirectional else 1
    inp = x.contiguous()
    for layer in range(lstm.num_layers):
        out = torch.empty((b, t, nd * h), dtype=torch.float32, device=x.device)
        for d in range(nd):
            sfx = '_l%d%s' % (layer, '_reverse' if d else '')
            w_ih, w_hh = getattr(lstm, 'weight_ih' + sfx), getattr(lstm, 'weight_hh' + sfx)
            bias = getattr(lstm, 'bias_ih' + sfx) + getattr(lstm, 'bias_hh' + sfx)
            xproj = torch.addmm(bias, inp.reshape(b * t, -1), w_ih.t())              # (B T, 4H): every time step in one GEMM
            ws = _workspace(h, x.device) if h > 64 else None
            C.check(lib.ap_lstm_recurrence(ops._ptr(xproj), ops._ptr(w_hh.contiguous()), None, None, ops._ptr(out), None, None, b, t, h, d,
                                           nd * h, d * h, ops._ptr(ws) if ws is not None else None, ops._stream()), 'lstm_recurrence')
        inp = out
    return inp


def check_timeouts():
    """After a synchronisation: did a distributed launch give up waiting for its peer workgroups (a device shared with other work)?"""
    for (h, _), ws in _WS.items():
        if C.lib().ap_lstm_timed_out(ctypes.c_void_p(ws.data_ptr()), h) == 1:
            raise RuntimeError('ap_lstm_recurrence (H = %d): the workgroups of a launch never met; its output is invalid' % h)
