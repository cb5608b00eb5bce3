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

Order of the collectives: segment k's all-reduce is issued only when segments 0..k-1 have been issued, on every rank and
in every launch mode (eager hooks, captured graph, replay + reduce_all()).  Ranks may therefore run different launch
modes in the same iteration (one rank replays a graph while another meets a new batch geometry and runs eagerly) and
still present RCCL with the same sequence of six collectives.

Failure flag: the last segment carries one extra float that every rank fills before the segment goes -- 0, or 1 when
this rank's training-step capture has failed -- and the averaged value is added to a device accumulator after the join.
train.Trainer reads the accumulator at iteration numbers every rank computes alike and, when it is non-zero, ALL ranks
step down together (captured collectives -> split replay -> eager): agreement about the launch mode without an extra
collective and without a per-step host sync.

Deferred GroupNorm column sums (ops._colsum): with the arena the ~70 per-layer launches of a backward pass collapse into
one multi-tensor launch per SEGMENT -- `_ready` flushes what is pending right before it lets a segment's all-reduce go.

Captured iteration (train.Trainer with a HIP graph, world > 1): the forward runs on leaf aliases of the parameters and
the gradients come from autograd.grad, so AccumulateGrad and its hooks never run.  `capture_begin(leaves)` maps the
aliases onto the slots and puts a tensor hook on every alias that does what `_ready` does -- on RCCL the segment
all-reduces are issued (and recorded into the graph, on RCCL's own stream: a parallel branch) while the backward pass
is still being captured; `capture_finish()` joins them in front of the Adam launch.  A transport whose collectives are
host-driven (gloo) cannot be captured: there the graph ends with the backward pass, and `reduce_all()` + the Adam launch
follow every replay eagerly (no overlap of the exchange with the backward pass, ~8 host calls instead of ~1000).
"""
import functools
import os

import torch
import torch.distributed as dist

ACTIVE = None            # the arena the backward kernels should write into (set by Trainer)
_ALIGN = 64              # floats: a layer's slots start 256-byte aligned ...
_ALIGN_IN = 4            # ... and within a layer every slot is 16-byte aligned (the Adam kernel reads float4), so
                         # GroupNorm's gamma | beta stay adjacent and are written as one (2, C) block


class GradArena:
    flushes_colsums = True      # ops._colsum may defer: every path that releases a gradient flushes first

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
        # RCCL collectives are stream work and can be recorded into a HIP graph; gloo's run on the host
        # Default: inside the graph only for a group of ONE (nothing to wait for).  With real peers the captured-collective
        # path has never run on hardware (one-GPU pool), and a mismatch there is a hang no except-clause can catch: a
        # multi-rank run replays forward + backward and issues the six collectives + Adam eagerly after each replay
        # unless VF_CAPTURE_COLLECTIVES=1 asks for the fully captured iteration.
        want = os.environ.get("VF_CAPTURE_COLLECTIVES", "1" if self.world == 1 else "0")
        self.capturable = self.avg and want == "1"
        self.flag_value = 0.0                                 # this rank's entry of the failure flag (see module doc)
        self.flag_acc = None
        self._alias = {}                                      # id(leaf alias of a parameter) -> index (captured step)
        self._hooks = []
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
        self.flag_off = off                                   # the failure flag: one float behind the last slot,
        self.flat = self._alloc_flat(off + _ALIGN)            # (the flag lives inside the last segment)
        self.flag_acc = torch.zeros(1, device=self.params[0].device, dtype=torch.float32)
        self.base = self.flat.data_ptr()
        self.seg_of, self.seg_range, k, lo = [0] * n, [], 0, 0
        for pos, i in enumerate(seq):
            self.seg_of[i] = k
            end = self.off[seq[pos + 1]] if pos + 1 < n else off + _ALIGN
            if end * self.nseg >= off * (k + 1) or pos + 1 == n:
                self.seg_range.append((lo, end))
                lo, k = end, k + 1
        self.seg_count = [self.seg_of.count(s) for s in range(len(self.seg_range))]
        self._reset()

    def _alloc_flat(self, n):
        return torch.zeros(n, device=self.params[0].device, dtype=torch.float32)

    def _after_layout(self):
        """(subclass hook: the arena exists, the first iteration's gradients have been moved in, nothing is launched yet)"""

    def _reset(self):
        self.pending = list(self.seg_count)
        self.launched = [False] * len(self.seg_range)
        self.next_seg = 0
        self.got = [False] * len(self.params)
        self.handed = set()                                   # slots given to a backward kernel this iteration

    def _view(self, i):
        p, o = self.params[i], self.off[i]
        return self.flat[o:o + p.numel()].view(p.shape)

    def _idx(self, p):
        i = self.index.get(id(p))
        return i if i is not None else self._alias.get(id(p))

    # -- called from the backward kernels' host code ------------------------------------------------------------
    def slot(self, p):
        """Fresh alias of p's gradient slot, or None when there is no layout yet, p is unknown, or p already holds
        a gradient (accumulation over several backward passes must add, not overwrite)."""
        i = self._idx(p)
        if i is None or self.flat is None or p.grad is not None or i in self.handed:
            return None                                       # second use of a shared layer: autograd must ADD
        self.handed.add(i)
        return self._view(i)

    def slot_pair(self, p, q):
        """One (2, n) alias covering the adjacent, equally sized slots of p and q (GroupNorm gamma / beta), or None."""
        i, j = self._idx(p), self._idx(q)
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
        g = p.grad
        s = self._arrived(i, g)
        if s is not g:
            p.grad = s

    def _arrived(self, i, g):
        """Gradient g of parameter i exists (enqueued): move it into its slot unless it was born there, count its
        segment down and let the segment go when it is complete.  Returns the slot view."""
        if self.got[i]:
            # a second backward pass before finish(): the slot's segment may already have been averaged in place
            raise RuntimeError("GradArena does not support gradient accumulation over several backward passes per "
                               "optimizer step (use VF_REDUCER=ddp with no_sync() for that)")
        self.got[i] = True
        if g.data_ptr() != self.base + 4 * self.off[i] or not g.is_contiguous():
            from . import ops
            ops.flush_colsums()                               # g may be a deferred destination: fill it before reading
            s = self._view(i)
            s.copy_(g)
            g = s
            self.copied += 1
        self.pending[self.seg_of[i]] -= 1
        self._launch_complete()
        return g

    def _launch_complete(self):
        """Issue, in index order, every segment whose gradients have all arrived (never segment k before k-1)."""
        if not self._may_launch():
            return
        k = self.next_seg
        while k < len(self.pending) and self.pending[k] == 0:
            self._launch(k)
            k += 1

    def _may_launch(self):
        return self.capturable or not (self.flat.is_cuda and torch.cuda.is_current_stream_capturing())

    def _launch(self, k):
        from . import ops
        ops.flush_colsums()                                   # deferred GroupNorm sums of this (and earlier) segments
        assert k == self.next_seg, (k, self.next_seg)       # the order every rank relies on
        lo, hi = self.seg_range[k]
        if k == len(self.seg_range) - 1:
            self.flat[self.flag_off:self.flag_off + 1].fill_(self.flag_value)
        t = self.flat[lo:hi]
        op = dist.ReduceOp.AVG if self.avg else dist.ReduceOp.SUM
        self.works.append((dist.all_reduce(t, op=op, group=self.group, async_op=True), t))
        self.launched[k] = True
        self.next_seg = k + 1

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
            self._after_layout()
        for k, done in enumerate(self.launched):
            if not done:
                for i, p in enumerate(self.params):
                    if self.seg_of[i] == k and not self.got[i]:
                        self._view(i).zero_()
                self._launch(k)
        self._join()

    def _join(self):
        for w, t in self.works:
            w.wait()
            if not self.avg:
                t.mul_(1.0 / self.world)
        if self.works:
            self.flag_acc.add_(self.flat[self.flag_off:self.flag_off + 1])
        self.works = []
        self._reset()

    # -- captured iteration (train.Trainer) ---------------------------------------------------------------------
    def capture_begin(self, leaves):
        """`leaves`: fresh leaf aliases of self.params (same order) that the captured forward runs on."""
        assert self.flat is not None and len(leaves) == len(self.params)
        self._reset()
        self._alias = {id(l): i for i, l in enumerate(leaves)}
        self._hooks = [l.register_hook(functools.partial(self._arrived, i)) for i, l in enumerate(leaves)]

    def capture_finish(self, grads):
        """After autograd.grad inside the capture: every gradient into its slot (those that no tensor hook saw: a
        parameter autograd reported as unused gets zeros), then -- capturable transport only -- the remaining segments
        go and all of them are joined.  Returns the slot views in parameter order."""
        for h in self._hooks:
            h.remove()
        self._hooks, self._alias = [], {}
        out = []
        for i, g in enumerate(grads):
            if not self.got[i]:
                if g is None:
                    self._view(i).zero_()
                    self.got[i] = True
                    self.pending[self.seg_of[i]] -= 1
                else:
                    g = self._arrived(i, g)
            out.append(self._view(i))
        if self.capturable:
            self.reduce_all()
        else:
            from . import ops
            ops.flush_colsums()
            self._reset()
        return out

    def capture_abort(self):
        """The capture failed half-way: forget its hooks, aliases and recorded collectives."""
        for h in self._hooks:
            h.remove()
        self._hooks, self._alias, self.works = [], {}, []
        self._reset()

    def reduce_all(self):
        """All-reduce every segment that has not been launched yet and join (eagerly after a replay on a host-driven
        transport; inside the capture on RCCL)."""
        for k, done in enumerate(self.launched):
            if not done:
                self._launch(k)
        self._join()


# ================================================================================================================
# VF_REDUCER=xgmi: the hand-written exchange of SURVEY 8f rank 1 -- a one-shot all-reduce over IPC-mapped peer arenas
# fused with the Adam update (csrc/xgmi.hip).  Default stays the arena over RCCL above.
# ================================================================================================================
import ctypes  # noqa: E402


class _IpcBuffer:
    """Device memory a peer process can map (vf_xgmi_alloc / _export / _open).  torch sees it through the CUDA array
    interface: no copy, and the tensor keeps this object (hence the allocation) alive."""

    def __init__(self, nbytes):
        from . import _lib
        ptr = ctypes.c_void_p()
        _lib.call("vf_xgmi_alloc", ctypes.byref(ptr), int(nbytes))
        self.ptr, self.nbytes = int(ptr.value), int(nbytes)

    def handle(self):
        from . import _lib
        buf = ctypes.create_string_buffer(64)
        _lib.call("vf_xgmi_export", ctypes.c_void_p(self.ptr), buf)
        return buf.raw

    def floats(self, n, device):
        assert 4 * n <= self.nbytes
        self.__cuda_array_interface__ = dict(shape=(int(n),), typestr="<f4", data=(self.ptr, False), version=2)
        t = torch.as_tensor(self, device=device)
        assert t.data_ptr() == self.ptr and t.dtype == torch.float32
        return t

    def free(self):
        from . import _lib
        if self.ptr:
            _lib.call("vf_xgmi_free", ctypes.c_void_p(self.ptr))
            self.ptr = 0

    def __del__(self):
        try:
            self.free()
        except Exception:           # noqa: BLE001  (interpreter shutdown: the driver reclaims the memory with the process)
            pass


def _ipc_open(handle):
    from . import _lib
    ptr = ctypes.c_void_p()
    _lib.call("vf_xgmi_open", ctypes.create_string_buffer(handle, 64), ctypes.byref(ptr))
    return int(ptr.value)


class XgmiArena(GradArena):
    """GradArena whose exchange step is ours too: every rank's arena is mapped into every peer (hipIpc over xGMI on one
    node); per segment, on a side stream behind the backward pass: signal "my segment is complete" to all peers, wait for
    theirs, ONE kernel that sums the segment from all W arenas in rank order, writes the average to a local buffer and
    applies Adam to this rank's parameters (same sums in the same order everywhere: replicas stay bit-identical), signal
    "done reading".  Replaces the six RCCL all-reduces AND the optimizer launch (`owns_optimizer_step`): Trainer calls
    begin_step(opt) before the forward pass and skips opt.step().  p.grad shows the averaged gradients after finish(),
    as with DDP.  Eager launches only (no HIP-graph capture of the iteration in this mode).  A peer that never signals
    becomes a VFHipError one iteration later (bounded device-side wait), not a hung queue.

    Status: correctness-only.  Tested with two processes sharing one GPU (tests/test_gpu_two_rank.py: gradients and
    parameters bit-equal to the arena path over six iterations); it has never run across two devices."""
    owns_optimizer_step = True

    def __init__(self, module, world, segments=6, group=None):
        super().__init__(module, world, segments=segments, group=group)
        self.capturable = False
        self.avg = True                       # (the kernel averages; _join must not scale)
        self.rank = dist.get_rank(group)
        self.epoch = 0
        self.timeout_us = int(float(os.environ.get("VF_XGMI_TIMEOUT_S", "20")) * 1e6)
        self.comm = torch.cuda.Stream(device=self.params[0].device, priority=-1)
        self._mem = self._flagmem = None
        self._opened = []
        self._tables = None                   # (optimizer epoch, [(device rows, n, blocks)] per segment)
        self._hyper = None
        self._status_host = None
        self._status_ev = None

    # -- memory -------------------------------------------------------------------------------------------------
    def _alloc_flat(self, n):
        self._mem = _IpcBuffer(4 * n)
        return self._mem.floats(n, self.params[0].device)

    def _lay_out(self):
        super()._lay_out()
        dev, W, nseg = self.flat.device, self.world, len(self.seg_range)
        self._flagmem = _IpcBuffer(max(256, 4 * 2 * nseg * W))      # unsigned [ready k | done k][rank]
        handles = [None] * W
        dist.all_gather_object(handles, (self._mem.handle(), self._flagmem.handle()), group=self.group)
        bases, flags = [], []
        for r, (ha, hf) in enumerate(handles):
            if r == self.rank:
                bases.append(self._mem.ptr)
                flags.append(self._flagmem.ptr)
            else:
                a, f = _ipc_open(ha), _ipc_open(hf)
                self._opened += [a, f]
                bases.append(a)
                flags.append(f)
        self._bases = (ctypes.c_void_p * W)(*bases)
        self._flags = (ctypes.c_void_p * W)(*flags)
        self.gavg = torch.zeros_like(self.flat)                      # the averaged gradients (local)
        self._gviews = [self.gavg[o:o + p.numel()].view(p.shape) for p, o in zip(self.params, self.off)]
        self.status = torch.zeros(1, device=dev, dtype=torch.int32)
        self._status_host = torch.zeros(1, dtype=torch.int32).pin_memory()
        self._scal = torch.zeros(3, device=dev, dtype=torch.float32)
        self._events = [torch.cuda.Event() for _ in range(nseg + 1)]
        dist.barrier(group=self.group)                               # every rank has mapped every rank

    def close(self):
        """End of the run, called by EVERY rank: unmap the peers' memory, wait until every rank has done so (nobody may
        free memory a peer still has mapped), then free this rank's arena and flag block."""
        from . import _lib
        torch.cuda.synchronize()
        for ptr in self._opened:
            _lib.call("vf_xgmi_close", ctypes.c_void_p(ptr))
        self._opened = []
        for p in self.params:
            p.grad = None
        self.flat = self._gviews = self.gavg = None
        if self._mem is not None:
            dist.barrier(group=self.group)
            self._flagmem.free()
            # (the arena tensor may still be referenced by a caller's gradient views: its memory is returned to the driver
            # only when torch drops the last tensor over it -- _IpcBuffer.__del__)
            self._mem = self._flagmem = None

    # -- per iteration ------------------------------------------------------------------------------------------
    def begin_step(self, opt):
        """Before the forward pass: advance Adam's step count, put {lr, 1-b1^t, 1-b2^t} where the fused kernels read
        them and make sure every peer has finished reading last iteration's gradients out of this rank's arena."""
        self._check_status()
        ext, hyper = opt.external_begin()
        self._opt, self._ext, self._hyper = opt, ext, hyper
        self.epoch += 1
        if self.flat is not None:
            self._prologue()

    def _after_layout(self):
        self._prologue()

    def _prologue(self):
        from . import _lib
        lr, bc1, bc2 = self._hyper[:3]
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        _lib.call("vf_adam_set_scalars", ctypes.c_void_p(self._scal.data_ptr()), lr, bc1, bc2, st)
        nseg = len(self.seg_range)
        if self._signalled_before:
            # peers' "done reading" flags of the last exchange: in practice long set (their reduce kernels ran before
            # their own next forward pass)
            _lib.call("vf_xgmi_wait", ctypes.c_void_p(self._flagmem.ptr), self.world, nseg, 2 * nseg,
                      ctypes.c_uint(self._signalled_before), ctypes.c_void_p(self.status.data_ptr()), self.timeout_us, st)

    _signalled_before = 0        # epoch of the last exchange this rank took part in

    def _segment_tables(self):
        opt = self._opt
        if self._tables is not None and self._tables[0] == (opt.graph_epoch, id(opt)):
            return self._tables[1]
        dev = self.flat.device
        tabs = []
        for k in range(len(self.seg_range)):
            rows, first = [], 0
            for i in sorted((i for i in range(len(self.params)) if self.seg_of[i] == k), key=lambda i: self.off[i]):
                p = self.params[i]
                st = opt.state[p]
                rows.append([p.data_ptr(), self.base + 4 * self.off[i], st["exp_avg"].data_ptr(),
                             st["exp_avg_sq"].data_ptr(), p.numel(), first])
                first += (p.numel() + 1023) // 1024
            if k == len(self.seg_range) - 1:                          # the failure flag: averaged, no update
                rows.append([0, self.base + 4 * self.flag_off, 0, 0, 1, first])
                first += 1
            tabs.append((torch.tensor(rows, dtype=torch.int64).to(dev), len(rows), first))
        self._tables = ((opt.graph_epoch, id(opt)), tabs)
        return tabs

    def _launch(self, k):
        from . import _lib, ops
        ops.flush_colsums()
        assert k == self.next_seg, (k, self.next_seg)
        nseg = len(self.seg_range)
        if k == nseg - 1:
            self.flat[self.flag_off:self.flag_off + 1].fill_(self.flag_value)
        tab, n, blocks = self._segment_tables()[k]
        ev = self._events[k]
        ev.record(torch.cuda.current_stream())          # the segment's producers (and the table upload) are enqueued
        self.comm.wait_event(ev)
        st = ctypes.c_void_p(self.comm.cuda_stream)
        e = ctypes.c_uint(self.epoch & 0xFFFFFFFF)
        _, _, _, b1, b2, eps = self._hyper
        _lib.call("vf_xgmi_signal", self._flags, self.world, self.rank, k, e, st)
        _lib.call("vf_xgmi_wait", ctypes.c_void_p(self._flagmem.ptr), self.world, k, k + 1, e,
                  ctypes.c_void_p(self.status.data_ptr()), self.timeout_us, st)
        _lib.call("vf_xgmi_reduce_adam", ctypes.c_void_p(tab.data_ptr()), n, blocks, self._bases,
                  ctypes.c_void_p(self.base), ctypes.c_void_p(self.gavg.data_ptr()), self.world,
                  ctypes.c_void_p(self._scal.data_ptr()), float(b1), float(b2), float(eps), st)
        _lib.call("vf_xgmi_signal", self._flags, self.world, self.rank, nseg + k, e, st)
        self.works.append(k)
        self.launched[k] = True
        self.next_seg = k + 1

    def _join(self):
        if self.works:
            ev = self._events[-1]
            ev.record(self.comm)
            main = torch.cuda.current_stream()
            main.wait_event(ev)
            self.flag_acc.add_(self.gavg[self.flag_off:self.flag_off + 1])
            for p, g in zip(self.params, self._gviews):               # the averaged gradients, as DDP leaves them
                p.grad = g
            torch.autograd.graph.increment_version(self.params)       # updated through raw pointers
            self._signalled_before = self.epoch & 0xFFFFFFFF
            if self._status_ev is None:                               # read by a later begin_step, without blocking
                self._status_host.copy_(self.status, non_blocking=True)
                self._status_ev = torch.cuda.Event()
                self._status_ev.record(main)
        self.works = []
        self._reset()

    def _check_status(self):
        if self._status_ev is None or not self._status_ev.query():
            return
        self._status_ev = None
        code = int(self._status_host.item())
        if code:
            from . import _lib
            raise _lib.VFHipError(f"xgmi reducer: rank {self.rank} waited {self.timeout_us / 1e6:.0f} s for a peer's flag "
                                  f"(slot {code - 1}) -- a rank died or left the iteration; the step that followed used "
                                  "incomplete gradients")
