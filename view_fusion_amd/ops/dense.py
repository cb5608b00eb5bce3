"""Linear layers, the grouped FeatureWiseAffine launch, Swish, Dropout and the sin / cos embedding (reference
model/unet.py:27-32, 142-182)."""
import ctypes

import torch

from .. import _lib, reducer
from .state import st
from .core import _c, _call, _check, _ptr, _stream
from .deferred import _gout


# ---------------------------------------------------------------------------------------------
def _bgemm(A, B, C, bias, batch, M, N, K, sA, sB, sC, alpha=1.0, beta=0.0, offA=0, offB=0, offC=0):
    _call("vf_bgemm", _ptr(A, offA), _ptr(B, offB), _ptr(C, offC), _ptr(bias), batch, M, N, K, sA[0], sA[1],
              sA[2], sB[0], sB[1], sB[2], sC[0], sC[1], sC[2], alpha, beta, _stream(), flops=2.0 * batch * M * N * K)


class _LinearFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b):
        _check(x, w, b)
        S, I = x.shape
        O = w.shape[0]
        y = torch.empty(S, O, device=x.device, dtype=torch.float32)
        _bgemm(x, w, y, b, 1, S, O, I, (0, I, 1), (0, 1, I), (0, O, 1))
        ctx.save_for_backward(x, w)
        ctx.pw, ctx.pb = w, b
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dy = _c(dy)
        S, I = x.shape
        O = w.shape[0]
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            _bgemm(dy, w, dx, None, 1, S, I, O, (0, O, 1), (0, I, 1), (0, I, 1))
        if ctx.needs_input_grad[1]:
            dw = _gout(ctx.pw, O, I, like=x)
            _bgemm(dy, x, dw, None, 1, O, I, S, (0, 1, O), (0, I, 1), (0, I, 1))
        if ctx.needs_input_grad[2]:
            db = _gout(ctx.pb, O, like=x)
            _call("vf_colsum", _ptr(dy), _ptr(db), 1, S, O, _stream())
        return dx, dw, db


def linear(x, weight, bias):
    """(S,I) @ weight(O,I)^T + bias -> (S,O)."""
    return _LinearFn.apply(x, weight, bias)


# ---------------------------------------------------------------------------------------------
# All FeatureWiseAffine linears of the UNet (30 x Linear(K -> C_g) on the SAME embedding) as one grouped launch.


def _ta_desc(layers, S, device):
    key = (S, device, tuple(l.weight.data_ptr() for l in layers), tuple(l.bias.data_ptr() for l in layers))
    hit = st._TA_DESC.get(id(layers[0]))
    if hit is not None and hit[0] == key:
        return hit[1]
    rows, coff = [], 0
    for l in layers:
        C = l.weight.shape[0]
        rows.append([l.weight.data_ptr(), l.bias.data_ptr(), C, S * coff, coff])
        coff += C
    plan = (torch.tensor(rows, dtype=torch.int64).to(device), [r[2] for r in rows], [r[4] for r in rows], coff)
    st._TA_DESC[id(layers[0])] = (key, plan)
    return plan


class _TimeAffineFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, emb, layers, *params):          # params = (w_0, b_0, w_1, b_1, ...) of `layers`, for autograd
        _check(emb, *params)
        S, K = emb.shape
        desc, Cs, coffs, CT = _ta_desc(layers, S, emb.device)
        out = torch.empty(S * CT, device=emb.device, dtype=torch.float32)
        _call("vf_time_affine_fwd", ctypes.c_void_p(desc.data_ptr()), len(Cs), _ptr(emb), _ptr(out), S, K, CT,
                  _stream())
        ctx.save_for_backward(emb)
        ctx.plan = (desc, Cs, coffs, CT)
        ctx.params = params
        return tuple(out[S * o:S * (o + C)].view(S, C) for C, o in zip(Cs, coffs))

    @staticmethod
    def backward(ctx, *grads):
        (emb,) = ctx.saved_tensors
        desc, Cs, coffs, CT = ctx.plan
        S, K = emb.shape
        de = torch.cat([(g if g is not None else emb.new_zeros(S, C)).reshape(-1) for g, C in zip(grads, Cs)])
        slots = gdst = dw = db = None
        arena = reducer.ACTIVE
        if arena is not None:                       # every layer's dW / db straight into its arena slot
            slots = [arena.slot(p) for p in ctx.params]
            if any(t is None for t in slots):
                slots = None
            else:
                # (keyed on the slots, not on the parameter objects: a captured step runs on leaf aliases of the
                # parameters and must find the table the eager iterations before it uploaded -- no copy in a capture)
                key = (id(arena), arena.base, slots[0].data_ptr(), len(slots))
                hit = st._TA_GDST.get(key)
                if hit is None:
                    rows = [[slots[2 * g].data_ptr(), slots[2 * g + 1].data_ptr()] for g in range(len(Cs))]
                    hit = (key, torch.tensor(rows, dtype=torch.int64).to(emb.device))
                    if len(st._TA_GDST) > 8:
                        st._TA_GDST.clear()
                    st._TA_GDST[key] = hit
                gdst = hit[1]
        if slots is None:
            dw = torch.empty(CT, K, device=emb.device, dtype=torch.float32)
            db = torch.empty(CT, device=emb.device, dtype=torch.float32)
        demb = ws = None
        if ctx.needs_input_grad[0]:
            demb = torch.empty_like(emb)
            ws = torch.empty(_lib.load().vf_time_affine_ws_floats(S, K), device=emb.device, dtype=torch.float32)
        _call("vf_time_affine_bwd", ctypes.c_void_p(desc.data_ptr()), len(Cs), _ptr(emb), _ptr(de), _ptr(dw),
                  _ptr(db), ctypes.c_void_p(gdst.data_ptr()) if gdst is not None else None, _ptr(demb), _ptr(ws), S, K,
                  CT, _stream())
        out = [demb, None]
        if slots is not None:
            return tuple(out + slots)
        for C, o in zip(Cs, coffs):
            out += [dw[o:o + C], db[o:o + C]]
        return tuple(out)


def time_affine_all(emb, layers):
    """[Linear_g(emb) for g in layers] (each (S, C_g)) in one launch; `layers` = list of nn.Linear holders."""
    params = []
    for l in layers:
        params += [l.weight, l.bias]
    return _TimeAffineFn.apply(emb, layers, *params)


class _SwishFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        _check(x)
        y = torch.empty_like(x)
        _call("vf_swish_fwd", _ptr(x), _ptr(y), x.numel(), _stream())
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        dy = _c(dy)
        dx = torch.empty_like(x)
        _call("vf_swish_bwd", _ptr(x), _ptr(dy), _ptr(dx), x.numel(), _stream())
        return dx


def swish(x):
    return _SwishFn.apply(x)


class _DropoutFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, u, p):
        _check(x, u)
        y = torch.empty_like(x)
        _call("vf_dropout", _ptr(x), _ptr(u), _ptr(y), x.numel(), float(p), _stream())
        ctx.save_for_backward(u)
        ctx.p = float(p)
        return y

    @staticmethod
    def backward(ctx, dy):
        (u,) = ctx.saved_tensors
        dy = _c(dy)
        dx = torch.empty_like(dy)
        _call("vf_dropout", _ptr(dy), _ptr(u), _ptr(dx), dy.numel(), ctx.p, _stream())
        return dx, None, None


def dropout(x, p, u=None):
    """nn.Dropout(p) in training mode (reference Block, unet.py:207-216): x * (u >= p) / (1 - p).  u = uniform draws
    in [0,1) shaped like x (default: torch's device RNG; tests inject them)."""
    if u is None:
        u = torch.rand_like(x)
    return _DropoutFn.apply(x, _c(u), p)


def sincos_embedding(level, angle, dim):
    """(S,1),(S,1) -> (S,dim): [sin|cos](level*f) ++ [sin|cos](angle*f), dim/4 frequencies."""
    level = _c(level.detach().reshape(-1).float())
    angle = _c(angle.detach().reshape(-1).float())
    _check(level, angle)
    S = level.numel()
    out = torch.empty(S, dim, device=level.device, dtype=torch.float32)
    _call("vf_sincos_embed", _ptr(level), _ptr(angle), _ptr(out), S, dim, _stream())
    return out
