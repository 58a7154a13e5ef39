"""Flat-buffer Adam: all parameters of an optimiser live in ONE contiguous fp32 buffer (the
``nn.Parameter``s are views into it), so the update is a single HIP launch (``ap_adam_step``) and the
data-parallel gradient exchange is a single all-reduce over one contiguous gradient buffer.

Semantics: ``torch.optim.Adam(params, lr, betas=(beta1, 0.999))`` as constructed at
Module2/models/geomgm_ifw_fore_model.py:346-360 (no weight decay, no amsgrad).  It subclasses
``torch.optim.Optimizer`` so the reference's ``lr_scheduler.LambdaLR`` (networks.py:55-59) drives ``lr``.
"""
import ctypes

import torch

from . import _capi as C
from . import ops
from .ops import _ptr, _stream


class FlatAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=2e-4, betas=(0.5, 0.999), eps=1e-8):
        params = [p for p in params]
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))
        self.epoch = 0
        self._params = params
        n = sum(p.numel() for p in params)
        dev = params[0].device
        self.flat = torch.empty(n, dtype=torch.float32, device=dev)
        self.flat_grad = torch.zeros(n, dtype=torch.float32, device=dev)
        self.exp_avg = torch.zeros(n, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(n, dtype=torch.float32, device=dev)
        self.step_count = 0
        off = 0
        with torch.no_grad():
            for p in params:
                k = p.numel()
                self.flat[off:off + k].copy_(p.detach().reshape(-1))
                p.data = self.flat[off:off + k].view_as(p)          # parameter becomes a view of the flat buffer
                p.grad = self.flat_grad[off:off + k].view_as(p)     # so does its gradient
                p._flat_owner, p._flat_off = self, off              # autograd.Tape.grad_block writes there directly
                off += k

    def zero_grad(self, set_to_none=False):
        self.flat_grad.zero_()
        self._rebind()

    def _rebind(self):
        """Keep every p.grad a view of the flat gradient buffer (autograd may have replaced it)."""
        off = 0
        for p in self._params:
            k = p.numel()
            view = self.flat_grad[off:off + k].view_as(p)
            if p.grad is None:
                p.grad = view
            elif p.grad.data_ptr() != view.data_ptr():
                view.copy_(p.grad)
                p.grad = view
            off += k

    @torch.no_grad()
    def step(self, closure=None):
        self._rebind()
        self.step_count += 1
        g = self.param_groups[0]
        if not self.flat.is_cuda:
            raise RuntimeError('FlatAdam: parameters must live on the MI355X (no CPU path)')
        C.check(C.lib().ap_adam_step(_ptr(self.flat), _ptr(self.flat_grad), _ptr(self.exp_avg), _ptr(self.exp_avg_sq),
                                     self.flat.numel(), float(g['lr']), float(g['betas'][0]), float(g['betas'][1]),
                                     float(g['eps']), self.step_count, _stream()), 'adam_step')
        # the update went through a raw pointer: tell the layers of THIS optimizer's networks that their packed-weight
        # caches are stale (ops.weight_key)
        self.epoch += 1
