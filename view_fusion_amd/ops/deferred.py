"""Gradient destinations and the DEFERRED per-pass launches of the backward pass: the GroupNorm gamma / beta column sums
and the Winograd weight-gradient slab sums (one table-driven launch per flush instead of ~70 / ~65 small ones), the
gradient-arena slot lookup, and the descriptor tables a HIP-graph capture of the iteration records its launches on."""
import ctypes

import torch

from .. import _lib, reducer
from .state import st
from .core import _call, _launch, _ptr, _stream


# Per-(view, channel) map sums of a gradient tensor that a kernel already had in registers: the GroupNorm backward
# knows sum_hw(dx) in closed form, and dx is exactly the dY of the conv in front of it, whose bias / embedding-bias
# gradients are those sums; a residual 1x1 conv sees the same dY tensor as the 3x3 conv it is added to.  The sums
# travel ON the gradient tensor itself (a Python attribute; autograd hands the same tensor object from one backward
# node to the next), stamped with the tensor's version: no global table, nothing keyed on addresses, and a gradient
# that autograd had to re-materialise (an accumulation) simply does not carry them.  st.ROWSUM_FUSION = False disables.


def _rowsum_put(t, rowsum, colsum):
    if st.ROWSUM_FUSION:
        t._vf_sums = (t._version, rowsum, colsum)


def _rowsum_get(t):
    h = getattr(t, "_vf_sums", None) if st.ROWSUM_FUSION else None
    if h is None or h[0] != t._version:
        return None
    return h


def _gslot(p):
    """Parameter p's slot in the data-parallel gradient arena (reducer.py), or None: the backward kernels write a
    gradient there directly so that neither autograd nor the reducer has to copy it."""
    a = reducer.ACTIVE
    return a.slot(p) if (a is not None and p is not None) else None


def _gout(p, *shape, like):
    t = _gslot(p)
    return t if t is not None else torch.empty(*shape, device=like.device, dtype=torch.float32)


# GroupNorm weight / bias gradients = column sums over the views of the per-(view, channel) partials the backward kernel
# emits: ~70 launches of 5 us per backward pass.  They are deferred: every GroupNorm backward only registers its
# (partials, destination) pair, and ONE multi-tensor launch fills all destinations -- at the end of the backward pass (an
# autograd engine callback) in single-process training, or, with the data-parallel gradient arena, right before a
# segment's all-reduce is issued (reducer.GradArena calls flush_colsums(): ~6 launches per pass instead of 73).
# Deferral hands autograd a destination that is FILLED LATER, so it is only taken when nothing can read the gradient
# before the flush:
#   * both parameters' .grad is None (AccumulateGrad then adopts the tensor without reading it; zero_grad(set_to_none=
#     True), what Trainer.step does);
#   * no tensor hook / post-accumulate-grad hook sits on the parameters (the arena's own hooks are the exception -- the
#     post-accumulate hook of the eager step and the tensor hook on a leaf alias of the captured step: both flush before
#     they let a segment go);
#   * the parameter is not marked `_vf_no_defer`: torch's DistributedDataParallel copies a gradient into its bucket from
#     a hook on the AccumulateGrad NODE (invisible from the tensor) while the backward pass is still running, so Trainer
#     marks the parameters of a model it wraps in DDP (VF_REDUCER=ddp); st.COLSUM_DEFER = False switches deferral off
#     for the whole process (nothing in the package sets it).
# A backward pass that raises never runs its engine callbacks, and a re-entrant backward pass (torch.utils.checkpoint)
# starts a new graph task while the outer one still has entries pending: entries of another task are recognised by their
# graph-task id and FLUSHED by the next pass (filling a destination nobody will read is harmless; dropping one autograd
# still hands out is not).  A capture that aborts drops them explicitly (drop_pending_colsums).
# Round 5: the slab sums behind the Winograd weight-gradient kernels (65 launches of ~7 us per iteration) are deferred the
# same way -- every layer runs only its main kernel into a workspace of its own (vf_wino_wgrad_main) and registers a
# descriptor row; ONE vf_wino44_reduce_multi launch per flush fills dW (and the bias gradients that ride along).  Same
# conditions, same flush points, same graph-task bookkeeping as the GroupNorm sums.  VF_WRED_DEFER=0: tuning aid.
# The deferred layers' slab workspaces (~18 MiB each, ~1.15 GiB per backward pass at S = 96) are slices of ONE
# per-device arena that every pass -- eager or replayed, whatever its geometry -- fills from offset 0: passes are
# ordered on the stream and a pass's slabs are dead once its flush has run.  (Round 5 gave every layer a fresh tensor:
# inside a captured iteration that pinned 1.15 GiB per graph -- a ragged run holds up to B (N - 1) + 1 graphs.)
# The arena grows outside captures only (a new, larger tensor; slices already handed out keep the old one alive until
# their flush, and a graph keeps the arena it was captured with); a captured layer that does not fit runs the
# non-deferred launch instead.


def _wred_ws(device, need):
    """`need` floats of the slab arena for one layer of the running backward pass, or None (inside a capture, arena
    too small: the caller launches the non-deferred kernel pair)."""
    ent = st._WRED_ARENA.setdefault(device, [None, 0])
    n = (int(need) + 63) // 64 * 64
    cap = 0 if ent[0] is None else ent[0].numel()
    if ent[1] + n > cap:
        if torch.cuda.is_current_stream_capturing():
            return None
        # (the slices of the old arena stay valid: the pending entries reference them)
        ent[0] = torch.empty(max(2 * cap, ent[1] + n, 1 << 24), device=device, dtype=torch.float32)
        ent[1] = 0
    ws = ent[0][ent[1]:ent[1] + n]
    ent[1] += n
    return ws


def wred_arena_bytes(device=None):
    """Bytes of the slab arena(s) currently allocated (tools/long_run.py reports it)."""
    return sum(4 * e[0].numel() for d, e in st._WRED_ARENA.items() if e[0] is not None and (device is None or d == device))


_CS_RING = 16           # staging buffers in rotation (the gradient arena flushes once per segment: ~6 tables per pass)


# A captured training step (train.Trainer) launches its table-driven kernels on device tables whose CONTENTS are only
# needed when the graph is replayed: they are uploaded after the capture has ended, with ordinary copies -- no host-to-
# device copy node (and no pinned staging buffer to keep stable) inside the graph.  Such a table must NOT come from the
# capturing graph's memory pool: the pool hands a block that an earlier tensor of the same capture has released to a
# later one, so on every replay the earlier kernels would scribble over a table uploaded once.  begin_capture()
# therefore allocates the table BEFORE the capture starts.




def begin_capture(device, max_rows, max_conv_rows=0):
    st._CAPTURE_TABLE = [torch.empty(max(1, max_rows), 6, dtype=torch.int64, device=device), 0, [], []]
    st._CAPTURE_TABLE_W = [torch.empty(max(1, max_conv_rows), 9, dtype=torch.int64, device=device), 0, [], []] if max_conv_rows else None


def end_capture():
    """Upload the tables of the capture that just ended; returns what the graph's owner must keep referenced for as
    long as it replays the graph."""
    ct, st._CAPTURE_TABLE = st._CAPTURE_TABLE, None
    cw, st._CAPTURE_TABLE_W = st._CAPTURE_TABLE_W, None
    drop_pending_colsums()          # (only a capture that failed half-way leaves any)
    if ct is not None and ct[1]:
        ct[0][:ct[1]].copy_(torch.tensor(ct[2], dtype=torch.int64))
    if cw is not None and cw[1]:
        cw[0][:cw[1]].copy_(torch.tensor(cw[2], dtype=torch.int64))
    return ct, cw




def _wred_rows(pend):
    rows, first = [], 0
    for row, nblk, _ in pend:
        r = list(row)
        r[8] = (r[8] & ~0xFFFFFFFF) | first          # `first` = the int32 at byte 64 of the row
        rows.append(r)
        first += nblk
    return rows, first


def _flush_wred():
    pend, st._PENDING_WRED = st._PENDING_WRED, []
    for ent in st._WRED_ARENA.values():     # the next pass fills the arena from its start again (stream order)
        ent[1] = 0
    if not pend:
        return
    if torch.cuda.is_current_stream_capturing():
        cw = st._CAPTURE_TABLE_W
        rows, total = _wred_rows(pend)
        if cw is None or cw[1] + len(rows) > cw[0].shape[0]:
            raise _lib.VFHipError("deferred weight-gradient slab sums inside a stream capture need ops.begin_capture() "
                                  "with max_conv_rows >= the number of 3x3 layers (one backward pass per capture)")
        _launch("conv_wgrad", 0.0, "vf_wino44_reduce_multi", ctypes.c_void_p(cw[0].data_ptr() + 72 * cw[1]), len(rows),
                total, _stream())
        cw[1] += len(rows)
        cw[2] += rows
        cw[3].append(pend)
        return
    key = tuple(v for e in pend for v in e[0])
    dev = pend[0][2][0].device
    ent = st._WR_TABLE.setdefault(dev, {"ring": [], "next": 0})
    slot = None
    for r in ent["ring"]:          # the caching allocator cycles through a few address sets: reuse an uploaded table
        if r[3] == key:
            slot = r
            break
    if slot is None:
        rows, total = _wred_rows(pend)
        n = max(128, len(rows))
        if len(ent["ring"]) < _CS_RING:
            slot = [torch.empty(n, 9, dtype=torch.int64).pin_memory(), torch.empty(n, 9, dtype=torch.int64, device=dev),
                    torch.cuda.Event(), None, 0, 0]
            ent["ring"].append(slot)
        else:
            slot = ent["ring"][ent["next"] % _CS_RING]
            ent["next"] += 1
            if slot[0].shape[0] < n:
                slot[0], slot[1] = (torch.empty(n, 9, dtype=torch.int64).pin_memory(),
                                    torch.empty(n, 9, dtype=torch.int64, device=dev))
            slot[2].synchronize()
        slot[0][:len(rows)].copy_(torch.tensor(rows, dtype=torch.int64))
        slot[1][:len(rows)].copy_(slot[0][:len(rows)], non_blocking=True)
        slot[2].record()
        slot[3], slot[4], slot[5] = key, len(rows), total
    _launch("conv_wgrad", 0.0, "vf_wino44_reduce_multi", ctypes.c_void_p(slot[1].data_ptr()), slot[4], slot[5], _stream())
    st.keep_wred = pend             # workspaces / destinations stay referenced until the next flush


def _flush_colsums():
    _flush_wred()
    pend, st._PENDING_COLSUMS, st._PENDING_TASK = st._PENDING_COLSUMS, [], None
    if not pend:
        return
    if torch.cuda.is_current_stream_capturing():
        # (several flushes per capture with the gradient arena: each takes the next rows of the pre-allocated table)
        ct = st._CAPTURE_TABLE
        rows, first = [], 0
        for parts, dgb, batch, S, C in pend:
            rows.append([parts.data_ptr(), dgb.data_ptr(), S, C, batch, first])
            first += ((C + 63) // 64) * batch
        if ct is None or ct[1] + len(rows) > ct[0].shape[0]:
            raise _lib.VFHipError("deferred GroupNorm parameter sums inside a stream capture need ops.begin_capture() "
                                  "with room for every GroupNorm layer (one backward pass per capture)")
        _call("vf_colsum_multi", ctypes.c_void_p(ct[0].data_ptr() + 48 * ct[1]), len(rows), first, _stream())
        ct[1] += len(rows)
        ct[2] += rows
        ct[3].append(pend)
        return
    key = tuple(v for e in pend for v in (e[0].data_ptr(), e[1].data_ptr(), e[3], e[4]))
    dev = pend[0][0].device
    ent = st._CS_TABLE.setdefault(dev, {"ring": [], "next": 0})
    slot = None
    for r in ent["ring"]:          # the caching allocator cycles through a few address sets: reuse an uploaded table
        if r[3] == key:
            slot = r
            break
    if slot is None:
        rows, first = [], 0
        for parts, dgb, batch, S, C in pend:
            rows.append([parts.data_ptr(), dgb.data_ptr(), S, C, batch, first])
            first += ((C + 63) // 64) * batch
        n = max(128, len(rows))
        if len(ent["ring"]) < _CS_RING:
            slot = [torch.empty(n, 6, dtype=torch.int64).pin_memory(), torch.empty(n, 6, dtype=torch.int64, device=dev),
                    torch.cuda.Event(), None, 0, 0]
            ent["ring"].append(slot)
        else:
            slot = ent["ring"][ent["next"] % _CS_RING]
            ent["next"] += 1
            if slot[0].shape[0] < n:
                slot[0], slot[1] = (torch.empty(n, 6, dtype=torch.int64).pin_memory(),
                                    torch.empty(n, 6, dtype=torch.int64, device=dev))
            slot[2].synchronize()      # that buffer's last upload is at least _CS_RING backward passes old: no wait
        slot[0][:len(rows)].copy_(torch.tensor(rows, dtype=torch.int64))
        slot[1][:len(rows)].copy_(slot[0][:len(rows)], non_blocking=True)
        slot[2].record()
        slot[3], slot[4], slot[5] = key, len(rows), first
    _call("vf_colsum_multi", ctypes.c_void_p(slot[1].data_ptr()), slot[4], slot[5], _stream())
    st.keep_colsums = pend          # the partials / destinations stay referenced until the next flush


def _defer_ok(params):
    a = reducer.ACTIVE
    flushes = a is not None and getattr(a, "flushes_colsums", False)
    for p in params:
        if p is None or p.grad is not None or getattr(p, "_vf_no_defer", False):
            return False
        if getattr(p, "_backward_hooks", None) and not (flushes and id(p) in a._alias):
            return False        # a foreign tensor hook would read the gradient before the flush; the arena's own hook
                                # on a leaf alias (captured iteration) flushes before it lets the segment go
        if a is None and getattr(p, "_post_accumulate_grad_hooks", None):
            return False
    return a is None or flushes


def _colsum(parts, dgb, batch, S, C, params):
    """dgb[b][c] = sum_s parts[b][s][c], now or (see above) deferred to the flush of the running backward pass."""
    # (entries of another graph task: a backward pass that failed (its callback never ran) -- or the OUTER pass of a
    # re-entrant backward (torch.utils.checkpoint, autograd.grad inside a backward), whose destinations autograd will still
    # hand out.  Outside a capture filling them now is always right (the entries keep their tensors alive); inside a
    # capture the abort path has already dropped them (drop_pending_colsums).  _defer_begin does that.)
    if _defer_begin(params, st._CAPTURE_TABLE):
        st._PENDING_COLSUMS.append((parts, dgb, batch, S, C))
        return
    _call("vf_colsum", _ptr(parts), _ptr(dgb), batch, S, C, _stream())


def _defer_begin(params, capture_table):
    """Common entry of the two deferrals: True when a destination may be filled at the flush of the running backward
    pass; makes sure that flush is queued and that entries of another graph task are dealt with first."""
    task = torch._C._current_graph_task_id()
    if not (st.COLSUM_DEFER and task != -1 and _defer_ok(params)
            and (capture_table is not None or not torch.cuda.is_current_stream_capturing())):
        return False
    if (st._PENDING_COLSUMS or st._PENDING_WRED) and st._PENDING_TASK != task:
        if torch.cuda.is_current_stream_capturing():
            st._PENDING_COLSUMS.clear()
            st._PENDING_WRED.clear()
        else:
            _flush_colsums()
    if not st._PENDING_COLSUMS and not st._PENDING_WRED:
        st._PENDING_TASK = task
        torch.autograd.Variable._execution_engine.queue_callback(_flush_colsums)
    return True


def flush_colsums():
    """Fill the destinations registered so far (the gradient arena calls this before a segment's all-reduce)."""
    if st._PENDING_COLSUMS or st._PENDING_WRED:
        _flush_colsums()


def drop_pending_colsums():
    """Forget deferred sums of a backward pass that did not complete (capture failure paths)."""
    st._PENDING_COLSUMS.clear()
    st._PENDING_WRED.clear()
    st._PENDING_TASK = None
    for ent in st._WRED_ARENA.values():
        ent[1] = 0
