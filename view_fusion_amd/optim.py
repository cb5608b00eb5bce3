"""Adam for the ViewFusion training step as ONE multi-tensor HIP launch.

Drop-in for `torch.optim.Adam(params, lr)` as the reference uses it (experiment.py:118-120: default
betas/eps, no weight decay, no amsgrad); `state_dict()` uses torch's keys (`step`, `exp_avg`,
`exp_avg_sq`) so optimizer checkpoints interchange.  The learning rate is read from
`param_group["lr"]` every step (the reference sets it from its LrScheduler before each step).

Step counts are per parameter, as in torch: the parameters of a group that share a step count form one
"bucket" = one launch (in the ViewFusion UNet every parameter receives a gradient every iteration, so
there is exactly one bucket); a parameter that joins later (first gradient at iteration k) gets a bucket
of its own with its own bias corrections.
"""
import ctypes

import torch

from . import _lib, ops


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))
        self._plans = {}          # per param group: buckets of {descriptor rows, staging buffers, step count}
        self.graph_epoch = 0      # bumped whenever the state tensors a captured step addresses are replaced

    # -- descriptor tables ---------------------------------------------------------------------------------------
    def _flush_steps(self, gi):
        """Write the buckets' step counters back into the per-parameter state (torch's `step` entries)."""
        for b in self._plans.get(gi, {}).get("buckets", ()):
            for p in b["params"]:
                self.state[p]["step"] = torch.tensor(float(b["t"]))
        ext = getattr(self, "_ext", None)
        if ext is not None and gi == 0 and ext["epoch"] == self.graph_epoch:
            for p in ext["params"]:
                self.state[p]["step"] = torch.tensor(float(ext["t"]))

    # -- an exchange step that applies the update itself (reducer.XgmiArena: all-reduce fused with Adam) -----------
    def external_begin(self):
        """One iteration of a reducer that runs this optimizer's update inside its own kernels: state for EVERY
        parameter (created if missing), one common step count, advanced by one here.  Returns (handle, (lr, 1-b1^t,
        1-b2^t, b1, b2, eps)); the caller's kernels update p / exp_avg / exp_avg_sq in place (adam.hip's arithmetic), so
        `state_dict()` stays torch's and a later plain `step()` continues from the same counters."""
        if len(self.param_groups) != 1:
            raise _lib.VFHipError("FusedAdam.external_begin: one parameter group")
        group = self.param_groups[0]
        ext = getattr(self, "_ext", None)
        if ext is None or ext["epoch"] != self.graph_epoch:
            for gi in self._plans:
                self._flush_steps(gi)
            steps = set()
            for p in group["params"]:
                if not p.is_cuda or p.dtype != torch.float32 or not p.is_contiguous():
                    raise _lib.VFHipError("FusedAdam needs contiguous float32 GPU parameters")
                st = self.state[p]
                if not st:
                    st["step"] = torch.tensor(0.0)
                    st["exp_avg"] = torch.zeros_like(p)
                    st["exp_avg_sq"] = torch.zeros_like(p)
                steps.add(int(float(st["step"])))
            if len(steps) != 1:
                raise _lib.VFHipError("FusedAdam.external_begin: the parameters carry different step counts")
            self._plans = {}                               # (a later plain step() re-reads the counters from the state)
            ext = self._ext = dict(params=list(group["params"]), t=steps.pop(), epoch=self.graph_epoch)
        ext["t"] += 1
        b1, b2 = group["betas"]
        t = ext["t"]
        return ext, (float(group["lr"]), 1.0 - b1 ** t, 1.0 - b2 ** t, float(b1), float(b2), float(group["eps"]))

    def _plan(self, gi, group):
        """Static part of the descriptor tables {param, grad, exp_avg, exp_avg_sq, numel, first_block}: built once
        (state tensors never move); only the gradient pointers can change from step to step."""
        params = [p for p in group["params"] if p.grad is not None]
        key = tuple(id(p) for p in params)
        plan = self._plans.get(gi)
        if plan is not None and plan["key"] == key:
            return plan
        self._flush_steps(gi)                          # a changed parameter set must not restart the bias correction
        self._ext = None                               # (an external reducer's counters have just been flushed)
        self.graph_epoch += 1
        by_step = {}
        for p in params:
            if not p.is_cuda or p.dtype != torch.float32 or not p.is_contiguous():
                raise _lib.VFHipError("FusedAdam needs contiguous float32 GPU parameters")
            st = self.state[p]
            if not st:
                st["step"] = torch.tensor(0.0)
                st["exp_avg"] = torch.zeros_like(p)
                st["exp_avg_sq"] = torch.zeros_like(p)
            by_step.setdefault(int(float(st["step"])), []).append(p)
        buckets = []
        for t, ps in sorted(by_step.items()):
            rows, first = [], 0
            for p in ps:
                st = self.state[p]
                rows.append([p.data_ptr(), 0, st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), p.numel(), first])
                first += (p.numel() + 1023) // 1024
            dev = ps[0].device
            buckets.append(dict(params=ps, blocks=first, t=t, numel=sum(p.numel() for p in ps),
                                host=[torch.tensor(rows, dtype=torch.int64).pin_memory() for _ in range(2)],
                                dev=[torch.empty(len(rows), 6, dtype=torch.int64, device=dev) for _ in range(2)],
                                ptrs=[None, None], done=[None, None], flip=0))
        plan = dict(key=key, buckets=buckets)
        self._plans[gi] = plan
        return plan

    def state_dict(self):
        for gi in set(self._plans) | {0}:              # the per-parameter step counters are kept lazily
            self._flush_steps(gi)
        return super().state_dict()

    def load_state_dict(self, sd):
        super().load_state_dict(sd)
        self._plans = {}
        self.graph_epoch += 1                          # new exp_avg / exp_avg_sq tensors: captured steps are stale

    # -- HIP-graph capture of a whole training step (train.Trainer) ------------------------------------------------
    # A captured step cannot carry step-dependent launch arguments, so the update reads {lr, 1-b1^t, 1-b2^t} from three
    # device floats that `graph_tick` refreshes (as launch arguments of a one-thread kernel) before every replay.
    def graph_begin(self):
        """Handle for capturing `step_captured`, or None when the update is not ONE launch over ALL parameters (more
        than one group or bucket, a parameter without state yet): the caller then stays on the eager path."""
        if len(self.param_groups) != 1:
            return None
        plan = self._plans.get(0)
        if plan is None or len(plan["buckets"]) != 1:
            return None
        b = plan["buckets"][0]
        if len(b["params"]) != len(self.param_groups[0]["params"]):
            return None
        rows = b["host"][0].clone()
        return dict(bucket=b, rows=rows, epoch=self.graph_epoch, dev=torch.empty(rows.shape, dtype=torch.int64, device=b["params"][0].device))

    @torch.no_grad()
    def step_captured(self, h, scalars):
        """Inside the capture, after backward(): the one multi-tensor launch.  The descriptor table's contents (the
        gradient addresses this capture allocated) are uploaded by `graph_end` once the capture has ended."""
        b, group = h["bucket"], self.param_groups[0]
        b1, b2 = group["betas"]
        _lib.call("vf_adam_multi_dev", ctypes.c_void_p(h["dev"].data_ptr()), len(b["params"]), b["blocks"],
                  ctypes.c_void_p(scalars.data_ptr()), float(b1), float(b2), float(group["eps"]),
                  ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))

    def graph_end(self, h):
        grads = [p.grad for p in h["bucket"]["params"]]
        if any(g is None or not g.is_contiguous() for g in grads):
            raise _lib.VFHipError("captured training step: a parameter received no (or a strided) gradient")
        h["rows"][:, 1] = torch.tensor([g.data_ptr() for g in grads], dtype=torch.int64)
        h["dev"].copy_(h["rows"])
        h["grads"] = grads                             # the graph writes these allocations on every replay
        return h

    def graph_tick(self, h, scalars):
        """Before a replay: advance the step count and hand the replay its learning rate and bias corrections."""
        b, group = h["bucket"], self.param_groups[0]
        b["t"] += 1
        b1, b2 = group["betas"]
        _lib.call("vf_adam_set_scalars", ctypes.c_void_p(scalars.data_ptr()), float(group["lr"]), 1.0 - b1 ** b["t"],
                  1.0 - b2 ** b["t"], ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        torch.autograd.graph.increment_version(b["params"])

    # -- the step ------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        stream = torch.cuda.current_stream()
        raw = ctypes.c_void_p(stream.cuda_stream)
        for gi, group in enumerate(self.param_groups):
            if not any(p.grad is not None for p in group["params"]):
                continue
            b1, b2 = group["betas"]
            for b in self._plan(gi, group)["buckets"]:
                b["t"] += 1
                t = b["t"]
                grads = [p.grad if p.grad.is_contiguous() else p.grad.contiguous() for p in b["params"]]
                ptrs = [g.data_ptr() for g in grads]
                f = b["flip"]
                if b["ptrs"][f] != ptrs:               # gradient arena / stable allocations: the table is reused as is
                    f = b["flip"] = f ^ 1
                    if b["ptrs"][f] != ptrs:
                        # two staging buffers, each guarded by an event: the H2D copy that last read this pinned
                        # buffer (two steps ago) must have executed before the host overwrites it -- the host may
                        # run several iterations ahead of the GPU
                        if b["done"][f] is not None:
                            b["done"][f].synchronize()
                        b["host"][f][:, 1] = torch.tensor(ptrs, dtype=torch.int64)
                        b["dev"][f].copy_(b["host"][f], non_blocking=True)
                        ev = torch.cuda.Event()
                        ev.record(stream)
                        b["done"][f], b["ptrs"][f] = ev, ptrs
                ops._launch("adam", 0.0, "vf_adam_multi", ctypes.c_void_p(b["dev"][f].data_ptr()), len(grads), b["blocks"],
                            float(group["lr"]), float(b1), float(b2), float(group["eps"]), 1.0 - b1 ** t, 1.0 - b2 ** t,
                            raw, nbytes=28.0 * b["numel"])   # 4 reads + 3 writes
                # the kernel writes through raw pointers: tell autograd (and the packed-weight caches of ops/packing.py,
                # which are keyed on the version counter) that the parameters changed
                torch.autograd.graph.increment_version(b["params"])
        return loss
