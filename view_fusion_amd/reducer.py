"""Gradient arena: the data-parallel exchange step of the training iteration (reference
experiment.py:104-107 wraps the model in DistributedDataParallel over NCCL).

torch's DDP reducer copies every parameter gradient into its bucket with one small kernel per
parameter (~400 launches, +1.9 ms on a 43 ms iteration).  Here the gradients are BORN in the
communication buffer instead: one flat fp32 arena holds a slot per parameter, the backward
kernels of `ops` write dW/db/dgamma/dbeta straight into the slot (`slot()`), autograd's
AccumulateGrad adopts the returned alias as `param.grad` without a copy, and a per-parameter
post-accumulate hook counts a segment down; when the last gradient of a segment has been
enqueued, ONE asynchronous all-reduce (RCCL `ncclAvg` over xGMI; SUM + scale on gloo) is
launched for the whole segment while the backward pass keeps running.  `finish()` joins the
collectives before the optimizer step.

The slot order is the order in which gradients become ready, observed on the first iteration
(rank 0's order is broadcast so that every rank uses the same layout), module by module so
that the parameters of one layer stay adjacent; the first iteration itself reduces from a copy.
A gradient produced outside our kernels (or by a stand-in torch module in the CPU tests) is
moved into its slot by the hook -- the arena is a correct reducer for any module, the zero-copy
path is an optimisation on top.  A parameter used by several layers of one graph gets its slot
handed out once per iteration; the other uses return ordinary tensors that autograd adds.
Gradient accumulation over several backward passes per optimizer step is NOT supported (the
reference never does it, experiment.py:286-293) and raises.

Single-process training (world 1) never constructs an arena.
"""
import functools

import torch
import torch.distributed as dist

ACTIVE = None            # the arena the backward kernels should write into (set by Trainer)
_ALIGN = 64              # floats: a layer's slots start 256-byte aligned ...
_ALIGN_IN = 4            # ... and within a layer every slot is 16-byte aligned (the Adam kernel reads float4), so
                         # GroupNorm's gamma | beta stay adjacent and are written as one (2, C) block


class GradArena:
    def __init__(self, module, world, segments=6, group=None):
        self.params = [p for p in module.parameters() if p.requires_grad]
        assert self.params and all(p.dtype == torch.float32 for p in self.params)
        self.world, self.group, self.nseg = dist.get_world_size(group), group, segments
        self.index = {id(p): i for i, p in enumerate(self.params)}
        # owner[i] = ordinal of the nn.Module that directly holds parameter i
        self.owner = [None] * len(self.params)
        for m_ord, m in enumerate(module.modules()):
            for p in m._parameters.values():
                i = self.index.get(id(p)) if p is not None else None
                if i is not None and self.owner[i] is None:
                    self.owner[i] = m_ord
        self.avg = dist.get_backend(group) == "nccl"          # RCCL has ncclAvg; gloo only SUM
        self.flat = None                                      # laid out at the end of the first iteration
        self.fired = []                                       # first iteration: parameter indices in ready order
        self.got = [False] * len(self.params)
        self.handed = set()
        self.works = []
        self.copied = 0                                       # gradients not born in their slot (stats, per step)
        for i, p in enumerate(self.params):
            p.register_post_accumulate_grad_hook(functools.partial(self._ready, i))
        dev = self.params[0].device
        for t in list(module.parameters()) + list(module.buffers()):      # rank 0's state everywhere (as DDP does)
            if t.device == dev:
                dist.broadcast(t.data, 0, group=self.group)

    # -- layout -------------------------------------------------------------------------------------------------
    def _lay_out(self):
        n = len(self.params)
        rank_of = [n] * n                                     # never fired -> placed last
        for r, i in enumerate(self.fired):
            rank_of[i] = min(rank_of[i], r)
        order = torch.tensor(rank_of, dtype=torch.int64, device=self.params[0].device)
        dist.broadcast(order, 0, group=self.group)            # one layout for all ranks
        rank_of = order.tolist()
        first = {}
        for i in range(n):
            first[self.owner[i]] = min(first.get(self.owner[i], n), rank_of[i])
        seq = sorted(range(n), key=lambda i: (first[self.owner[i]], self.owner[i], i))
        self.off, off, prev = [0] * n, 0, None
        for i in seq:
            al = _ALIGN_IN if self.owner[i] == prev else _ALIGN
            off = (off + al - 1) // al * al
            self.off[i], prev = off, self.owner[i]
            off += self.params[i].numel()
        off = (off + _ALIGN - 1) // _ALIGN * _ALIGN
        self.flat = torch.zeros(off, device=self.params[0].device, dtype=torch.float32)
        self.base = self.flat.data_ptr()
        self.seg_of, self.seg_range, k, lo = [0] * n, [], 0, 0
        for pos, i in enumerate(seq):
            self.seg_of[i] = k
            end = self.off[seq[pos + 1]] if pos + 1 < n else off
            if end * self.nseg >= off * (k + 1):
                self.seg_range.append((lo, end))
                lo, k = end, k + 1
        self.seg_count = [self.seg_of.count(s) for s in range(len(self.seg_range))]
        self._reset()

    def _reset(self):
        self.pending = list(self.seg_count)
        self.launched = [False] * len(self.seg_range)
        self.got = [False] * len(self.params)
        self.handed = set()                                   # slots given to a backward kernel this iteration

    def _view(self, i):
        p, o = self.params[i], self.off[i]
        return self.flat[o:o + p.numel()].view(p.shape)

    # -- called from the backward kernels' host code ------------------------------------------------------------
    def slot(self, p):
        """Fresh alias of p's gradient slot, or None when there is no layout yet, p is unknown, or p already holds
        a gradient (accumulation over several backward passes must add, not overwrite)."""
        i = self.index.get(id(p))
        if i is None or self.flat is None or p.grad is not None or i in self.handed:
            return None                                       # second use of a shared layer: autograd must ADD
        self.handed.add(i)
        return self._view(i)

    def slot_pair(self, p, q):
        """One (2, n) alias covering the adjacent, equally sized slots of p and q (GroupNorm gamma / beta), or None."""
        i, j = self.index.get(id(p)), self.index.get(id(q))
        if i is None or j is None or self.flat is None or p.grad is not None or q.grad is not None:
            return None
        n = p.numel()
        if q.numel() != n or self.off[j] != self.off[i] + n or i in self.handed or j in self.handed:
            return None
        self.handed.update((i, j))
        return self.flat[self.off[i]:self.off[i] + 2 * n].view(2, n)

    # -- autograd thread ----------------------------------------------------------------------------------------
    def _ready(self, i, p):
        if self.flat is None:
            self.got[i] = True
            self.fired.append(i)
            return
        if self.got[i]:
            # a second backward pass before finish(): the slot's segment may already have been averaged in place
            raise RuntimeError("GradArena does not support gradient accumulation over several backward passes per "
                               "optimizer step (use VF_REDUCER=ddp with no_sync() for that)")
        self.got[i] = True
        g = p.grad
        if g.data_ptr() != self.base + 4 * self.off[i] or not g.is_contiguous():
            s = self._view(i)
            s.copy_(g)
            p.grad = s
            self.copied += 1
        k = self.seg_of[i]
        self.pending[k] -= 1
        if self.pending[k] == 0:
            self._launch(k)

    def _launch(self, k):
        lo, hi = self.seg_range[k]
        t = self.flat[lo:hi]
        op = dist.ReduceOp.AVG if self.avg else dist.ReduceOp.SUM
        self.works.append((dist.all_reduce(t, op=op, group=self.group, async_op=True), t))
        self.launched[k] = True

    # -- main thread, after loss.backward() ---------------------------------------------------------------------
    def finish(self):
        """Reduce what is still pending (a parameter without a gradient this step contributes zeros), then make
        the current stream wait for every collective."""
        if self.flat is None:                                 # first iteration: lay out, move the gradients in
            got = self.got
            self._lay_out()
            for i, p in enumerate(self.params):
                if got[i]:
                    s = self._view(i)
                    s.copy_(p.grad)
                    p.grad = s
            self.got = got
        for k, done in enumerate(self.launched):
            if not done:
                for i, p in enumerate(self.params):
                    if self.seg_of[i] == k and not self.got[i]:
                        self._view(i).zero_()
                self._launch(k)
        for w, t in self.works:
            w.wait()
            if not self.avg:
                t.mul_(1.0 / self.world)
        self.works = []
        self._reset()
