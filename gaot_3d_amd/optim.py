"""Fused AdamW for the hot path: same constructor arguments, defaults, param_groups and state_dict layout
(``step`` / ``exp_avg`` / ``exp_avg_sq`` per parameter) as ``torch.optim.AdamW`` as the reference uses it
(src/trainer/optimizers.py:210: ``AdamW(params, lr=config.lr, weight_decay=config.weight_decay)``), one HIP launch per
48 parameter tensors instead of ~15 multi-tensor ATen kernels.  The step counter and the learning rate are device
scalars, so ``step()`` can sit inside a captured hipGraph; a host LR schedule that edits ``param_groups[i]["lr"]``
(optimizers.py:226-246) is picked up by the next ``step()`` outside capture, or by ``sync_lr()`` before a replay."""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib
from ._lib import GaotError, check


class _Entry(C.Structure):   # gaot_adamw_tensor_t
    _fields_ = [("param", C.c_void_p), ("grad", C.c_void_p), ("exp_avg", C.c_void_p), ("exp_avg_sq", C.c_void_p),
                ("numel", C.c_int64)]


class AdamW(torch.optim.Optimizer):
    def __init__(self, params, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 1e-2,
                 amsgrad: bool = False):
        if amsgrad:
            raise NotImplementedError("amsgrad is not implemented on the HIP path (the reference does not use it)")
        if lr < 0 or eps < 0 or weight_decay < 0 or not (0 <= betas[0] < 1 and 0 <= betas[1] < 1):
            raise ValueError("invalid AdamW hyper-parameter")
        super().__init__(params, dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay, amsgrad=False))
        self._dev = {}   # group index -> (lr tensor, step tensor)

    def _scalars(self, gi: int, group, device):
        if gi not in self._dev:
            self._dev[gi] = (torch.tensor([group["lr"]], dtype=torch.float32, device=device),
                             torch.zeros(1, dtype=torch.float32, device=device))
            self._host_lr = getattr(self, "_host_lr", {})
            self._host_lr[gi] = group["lr"]
        return self._dev[gi]

    def sync_lr(self) -> None:
        """push the param_groups' learning rates to the device scalars (call before replaying a captured step)"""
        for gi, group in enumerate(self.param_groups):
            if gi in self._dev and self._host_lr.get(gi) != group["lr"]:
                self._dev[gi][0].fill_(group["lr"])
                self._host_lr[gi] = group["lr"]

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = _lib.load()
        capturing = torch.cuda.is_available() and torch.cuda.is_current_stream_capturing()
        for gi, group in enumerate(self.param_groups):
            ps = [p for p in group["params"] if p.grad is not None]
            if not ps:
                continue
            dev = ps[0].device
            if dev.type != "cuda":
                raise GaotError("gaot_3d_amd.optim.AdamW runs on the HIP device only (no CPU fallback)")
            lr_t, step_t = self._scalars(gi, group, dev)
            if not capturing:
                self.sync_lr()
            entries = (_Entry * len(ps))()
            for i, p in enumerate(ps):
                if p.dtype != torch.float32 or p.grad.dtype != torch.float32 or not p.is_contiguous():
                    raise GaotError("AdamW: parameters and gradients must be contiguous fp32")
                g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                st = self.state[p]
                if len(st) == 0:
                    if capturing:
                        raise GaotError("AdamW: run one step before capturing (optimizer state is allocated lazily)")
                    st["step"] = step_t          # shared device counter (torch keeps one per parameter; same value)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                elif st["step"] is not step_t:   # state loaded from a torch.optim.AdamW checkpoint: adopt its counter
                    if capturing:
                        raise GaotError("AdamW: run one step after load_state_dict before capturing")
                    step_t.fill_(float(st["step"]))
                    st["step"] = step_t
                    st["exp_avg"] = st["exp_avg"].to(dev, torch.float32).contiguous()
                    st["exp_avg_sq"] = st["exp_avg_sq"].to(dev, torch.float32).contiguous()
                entries[i] = _Entry(p.data_ptr(), g.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), p.numel())
            b1, b2 = group["betas"]
            check(lib.gaot_adamw_step(entries, len(ps), C.c_void_p(lr_t.data_ptr()), C.c_void_p(step_t.data_ptr()), float(b1),
                                      float(b2), float(group["eps"]), float(group["weight_decay"]),
                                      C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)), "gaot_adamw_step")
        return loss
