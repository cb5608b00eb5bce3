"""GroupNorm (+Swish) forward / backward autograd Functions (reference model/unet.py:211-212, 254, 180-182)."""
import torch

from .. import _lib, reducer
from .state import st
from .core import _c, _check, _launch, _ptr, _stream
from .deferred import _colsum, _rowsum_put


# ---------------------------------------------------------------------------------------------
def _gn_forward(x, gamma, beta, groups, silu):
    _check(x, gamma, beta)
    S, C, H, W = x.shape
    y = torch.empty_like(x)
    mean = torch.empty(S * groups, device=x.device, dtype=torch.float32)
    rstd = torch.empty_like(mean)
    _launch("gn_fwd", 0.0, "vf_gn_fwd", _ptr(x), _ptr(gamma), _ptr(beta), _ptr(y), _ptr(mean), _ptr(rstd), S, C, H * W,
            groups, 1e-5, int(silu), _stream(), nbytes=8.0 * x.numel())        # read x once, write y once
    return y, mean, rstd


def _gn_backward(ctx, dy, addend, addend2=None):
    x, gamma, beta, mean, rstd = ctx.saved_tensors
    dy = _c(dy)
    S, C, H, W = x.shape
    dx = torch.empty_like(x)
    parts = torch.empty(2, S, C, device=x.device, dtype=torch.float32)
    rowsum = None
    if addend is None and st.ROWSUM_FUSION and _lib.load().vf_gn_bwd_emits_rowsum(C, H * W, ctx.groups):
        rowsum = torch.empty(S, C, device=x.device, dtype=torch.float32)
    if addend is None and addend2 is not None:
        addend, addend2 = addend2, None
    _launch("gn_bwd", 0.0, "vf_gn_cat_bwd", _ptr(x), None, C, _ptr(gamma), _ptr(beta), _ptr(mean), _ptr(rstd), _ptr(dy),
            _ptr(addend), _ptr(addend2), _ptr(dx), None, _ptr(parts[0]), _ptr(parts[1]), _ptr(rowsum), S, C, H * W,
            ctx.groups, ctx.silu, _stream(),              # read x, dy (+ the fused residual / skip gradients), write dx
            nbytes=4.0 * x.numel() * (3 + (addend is not None) + (addend2 is not None)))
    if rowsum is not None:
        _rowsum_put(dx, rowsum, None)
    dgb = reducer.ACTIVE.slot_pair(*ctx.gb) if reducer.ACTIVE is not None else None
    if dgb is None:
        dgb = torch.empty(2, C, device=x.device, dtype=torch.float32)
    _colsum(parts, dgb, 2, S, C, ctx.gb)
    return dx, dgb[0], dgb[1]


class _GroupNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, groups, silu):
        y, mean, rstd = _gn_forward(x, gamma, beta, groups, silu)
        ctx.save_for_backward(x, gamma, beta, mean, rstd)
        ctx.groups, ctx.silu, ctx.gb = groups, int(silu), (gamma, beta)
        return y

    @staticmethod
    def backward(ctx, dy):
        return (*_gn_backward(ctx, dy, None), None, None)


class _GroupNormSkipFn(torch.autograd.Function):
    """(GN(x), x, x): the extra outputs are x itself for a residual consumer and for the decoder's skip
    connection, so that their gradients are added inside the GroupNorm backward kernel instead of by separate
    autograd adds."""

    @staticmethod
    def forward(ctx, x, gamma, beta, groups, silu):
        y, mean, rstd = _gn_forward(x, gamma, beta, groups, silu)
        ctx.save_for_backward(x, gamma, beta, mean, rstd)
        ctx.groups, ctx.silu, ctx.gb = groups, int(silu), (gamma, beta)
        ctx.set_materialize_grads(False)          # an unused handle must not cost a zero tensor + an add
        return y, x.view_as(x), x.view_as(x)

    @staticmethod
    def backward(ctx, dy, dskip, dtap):
        return (*_gn_backward(ctx, dy, None if dskip is None else _c(dskip), None if dtap is None else _c(dtap)),
                None, None)


class _GroupNormCatSkipFn(torch.autograd.Function):
    """GroupNorm over the channel concatenation [x1 | x2] that is never materialised (decoder skip connections,
    reference unet.py:134): returns (GN(cat), x1, x2); the gradients of the second consumers of x1 / x2 (the
    residual 1x1 conv) are added inside the backward kernel, which writes dx1 and dx2 separately."""

    @staticmethod
    def forward(ctx, x1, x2, gamma, beta, groups, silu):
        _check(x1, x2, gamma, beta)
        S, C1, H, W = x1.shape
        C = C1 + x2.shape[1]
        y = torch.empty(S, C, H, W, device=x1.device, dtype=torch.float32)
        mean = torch.empty(S * groups, device=x1.device, dtype=torch.float32)
        rstd = torch.empty_like(mean)
        _launch("gn_fwd", 0.0, "vf_gn_cat_fwd", _ptr(x1), _ptr(x2), C1, _ptr(gamma), _ptr(beta), _ptr(y), _ptr(mean),
                _ptr(rstd), S, C, H * W, groups, 1e-5, int(silu), _stream(), nbytes=8.0 * y.numel())
        ctx.save_for_backward(x1, x2, gamma, beta, mean, rstd)
        ctx.groups, ctx.silu, ctx.gb = groups, int(silu), (gamma, beta)
        return y, x1.view_as(x1), x2.view_as(x2)

    @staticmethod
    def backward(ctx, dy, d1, d2):
        x1, x2, gamma, beta, mean, rstd = ctx.saved_tensors
        dy = _c(dy)
        S, C1, H, W = x1.shape
        C = C1 + x2.shape[1]
        if (d1 is None) != (d2 is None):          # one second consumer only: give the other a zero gradient
            d1 = torch.zeros_like(x1) if d1 is None else d1
            d2 = torch.zeros_like(x2) if d2 is None else d2
        d1 = None if d1 is None else _c(d1)
        d2 = None if d2 is None else _c(d2)
        dx1, dx2 = torch.empty_like(x1), torch.empty_like(x2)
        parts = torch.empty(2, S, C, device=x1.device, dtype=torch.float32)
        _launch("gn_bwd", 0.0, "vf_gn_cat_bwd", _ptr(x1), _ptr(x2), C1, _ptr(gamma), _ptr(beta), _ptr(mean), _ptr(rstd),
                _ptr(dy), _ptr(d1), _ptr(d2), _ptr(dx1), _ptr(dx2), _ptr(parts[0]), _ptr(parts[1]), None, S, C, H * W,
                ctx.groups, ctx.silu, _stream(), nbytes=4.0 * dy.numel() * (3 + (d1 is not None)))
        dgb = reducer.ACTIVE.slot_pair(*ctx.gb) if reducer.ACTIVE is not None else None
        if dgb is None:
            dgb = torch.empty(2, C, device=x1.device, dtype=torch.float32)
        _colsum(parts, dgb, 2, S, C, ctx.gb)
        return dx1, dx2, dgb[0], dgb[1], None, None


def cat_fusable(C1, C, HW, groups):
    """The concat-free decoder path needs the single-pass GroupNorm backward and 64-aligned split points."""
    return C1 % 64 == 0 and bool(_lib.load().vf_gn_bwd_emits_rowsum(C, HW, groups))


def group_norm_cat_skip(x1, x2, weight, bias, groups, silu):
    """-> (GroupNorm(cat(x1, x2)), x1', x2') without building the concatenation; see _GroupNormCatSkipFn."""
    return _GroupNormCatSkipFn.apply(x1, x2, weight, bias, groups, silu)


def group_norm(x, weight, bias, groups, silu):
    """GroupNorm(groups, C, eps=1e-5) [+ x*sigmoid(x)] on (S,C,H,W)."""
    return _GroupNormFn.apply(x, weight, bias, groups, silu)


def group_norm_skip(x, weight, bias, groups, silu, tap=False):
    """-> (GroupNorm(x), x_for_the_residual_branch[, x_for_the_decoder_skip]); see _GroupNormSkipFn."""
    if not (torch.is_grad_enabled() and x.requires_grad):
        y = _GroupNormFn.apply(x, weight, bias, groups, silu)
        return (y, x, x) if tap else (y, x)
    out = _GroupNormSkipFn.apply(x, weight, bias, groups, silu)
    return out if tap else out[:2]
