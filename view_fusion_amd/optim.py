"""Adam for the ViewFusion training step as ONE multi-tensor HIP launch.

Drop-in for `torch.optim.Adam(params, lr)` as the reference uses it (experiment.py:118-120: default
betas/eps, no weight decay, no amsgrad); `state_dict()` uses torch's keys (`step`, `exp_avg`,
`exp_avg_sq`) so optimizer checkpoints interchange.  The learning rate is read from
`param_group["lr"]` every step (the reference sets it from its LrScheduler before each step).
"""
import ctypes

import torch

from . import _lib


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for group in self.param_groups:
            rows, first, t = [], 0, None
            for p in group["params"]:
                if p.grad is None:
                    continue
                if not p.is_cuda or p.dtype != torch.float32 or not p.is_contiguous():
                    raise _lib.VFHipError("FusedAdam needs contiguous float32 GPU parameters")
                st = self.state[p]
                if not st:
                    st["step"] = torch.tensor(0.0)
                    st["exp_avg"] = torch.zeros_like(p)
                    st["exp_avg_sq"] = torch.zeros_like(p)
                st["step"] += 1
                t = float(st["step"])
                g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                rows.append([p.data_ptr(), g.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(),
                             p.numel(), first])
                first += (p.numel() + 1023) // 1024
            if not rows:
                continue
            b1, b2 = group["betas"]
            dev = group["params"][0].device
            desc = torch.tensor(rows, dtype=torch.int64).pin_memory().to(dev, non_blocking=True)
            _lib.call("vf_adam_multi", ctypes.c_void_p(desc.data_ptr()), len(rows), first, float(group["lr"]),
                      float(b1), float(b2), float(group["eps"]), 1.0 - b1 ** t, 1.0 - b2 ** t,
                      ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        return loss
