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
        self._plans = {}          # per param group: cached descriptor rows + pinned staging buffers

    def _plan(self, gi, group):
        """Static part of the descriptor table {param, grad, exp_avg, exp_avg_sq, numel, first_block}: built once
        (state tensors never move); only the gradient pointers change from step to step."""
        params = [p for p in group["params"] if p.grad is not None]
        key = tuple(id(p) for p in params)
        plan = self._plans.get(gi)
        if plan is not None and plan["key"] == key:
            return plan
        rows, first = [], 0
        for p in params:
            if not p.is_cuda or p.dtype != torch.float32 or not p.is_contiguous():
                raise _lib.VFHipError("FusedAdam needs contiguous float32 GPU parameters")
            st = self.state[p]
            if not st:
                st["step"] = torch.tensor(0.0)
                st["exp_avg"] = torch.zeros_like(p)
                st["exp_avg_sq"] = torch.zeros_like(p)
            rows.append([p.data_ptr(), 0, st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), p.numel(), first])
            first += (p.numel() + 1023) // 1024
        dev = params[0].device
        plan = dict(key=key, params=params, blocks=first, t=int(float(self.state[params[0]]["step"])),
                    host=[torch.tensor(rows, dtype=torch.int64).pin_memory() for _ in range(2)],
                    dev=[torch.empty(len(rows), 6, dtype=torch.int64, device=dev) for _ in range(2)], flip=0)
        self._plans[gi] = plan
        return plan

    def state_dict(self):
        for plan in self._plans.values():        # the per-parameter step counters are kept lazily
            for p in plan["params"]:
                self.state[p]["step"] = torch.tensor(float(plan["t"]))
        return super().state_dict()

    def load_state_dict(self, sd):
        super().load_state_dict(sd)
        self._plans = {}

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for gi, group in enumerate(self.param_groups):
            if not any(p.grad is not None for p in group["params"]):
                continue
            plan = self._plan(gi, group)
            plan["t"] += 1
            t = plan["t"]
            plan["flip"] ^= 1                     # two staging buffers: the previous step's copy may still be queued
            host, desc = plan["host"][plan["flip"]], plan["dev"][plan["flip"]]
            grads = [p.grad if p.grad.is_contiguous() else p.grad.contiguous() for p in plan["params"]]
            host[:, 1] = torch.tensor([g.data_ptr() for g in grads], dtype=torch.int64)
            desc.copy_(host, non_blocking=True)
            b1, b2 = group["betas"]
            _lib.call("vf_adam_multi", ctypes.c_void_p(desc.data_ptr()), len(grads), plan["blocks"],
                      float(group["lr"]), float(b1), float(b2), float(group["eps"]), 1.0 - b1 ** t, 1.0 - b2 ** t,
                      ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        return loss
