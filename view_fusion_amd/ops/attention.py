"""Self-attention core (reference model/unet.py:258-277) and the channel concatenation fallback."""
import math

import torch

from .state import st
from .core import _c, _call, _check, _launch, _ptr, _stream
from .dense import _bgemm


# ---------------------------------------------------------------------------------------------
# VF_ATTN_DSCORE=0: tuning aid -- the attention backward's dP product + softmax backward as two launches (rounds 1-4)


class _AttentionFn(torch.autograd.Function):
    """softmax(Q^T K / sqrt(C)) applied to V for single-head spatial attention; qkv (S,3C,H,W)."""

    @staticmethod
    def forward(ctx, qkv, need_p):
        _check(qkv)
        S, C3, H, W = qkv.shape
        C, L = C3 // 3, H * W
        alpha = 1.0 / math.sqrt(C)
        out = torch.empty(S, C, H, W, device=qkv.device, dtype=torch.float32)
        if L in (64, 256) and C % 32 == 0:        # fused flash-style kernel, scores stay in registers
            P = torch.empty(S, L, L, device=qkv.device, dtype=torch.float32) if need_p else None
            _launch("attn_fwd", 4.0 * S * L * L * C, "vf_attention_fwd", _ptr(qkv), _ptr(out), _ptr(P), S, C, L, _stream(),
                    nbytes=4.0 * (qkv.numel() + out.numel() + (S * L * L if need_p else 0)))
        else:                                     # generic sizes: materialised scores
            P = torch.empty(S, L, L, device=qkv.device, dtype=torch.float32)
            _bgemm(qkv, qkv, P, None, S, L, L, C, (C3 * L, 1, L), (C3 * L, L, 1), (L * L, L, 1), alpha,
                   offA=0, offB=C * L)
            _call("vf_softmax_fwd", _ptr(P), _ptr(P), S * L, L, _stream())
            _bgemm(qkv, P, out, None, S, C, L, L, (C3 * L, L, 1), (L * L, 1, L), (C * L, L, 1), offA=2 * C * L)
        ctx.save_for_backward(qkv, P)
        return out

    @staticmethod
    def backward(ctx, dO):
        qkv, P = ctx.saved_tensors
        dO = _c(dO)
        S, C3, H, W = qkv.shape
        C, L = C3 // 3, H * W
        alpha = 1.0 / math.sqrt(C)
        dqkv = torch.empty_like(qkv)
        dS = torch.empty_like(P)
        st._KIND_OVERRIDE = "attn_bwd" if st.KERNEL_LOG is not None else None
        fused_dq = False
        if L == 256 and C % 32 == 0 and st.ATTN_DSCORE:
            # one launch (round 5): dS = P o (dP - rowsum(P o dP)) with dP[i][j] = sum_c dO[c][i] v[c][j] never written,
            # and dQ[c][i] = alpha sum_j k[c][j] dS[i][j] from the dS values still in registers
            _call("vf_attention_dscore", _ptr(qkv), _ptr(dO), _ptr(P), _ptr(dS), _ptr(dqkv), S, C, L, _stream(),
                  flops=4.0 * S * L * L * C)
            fused_dq = True
        else:
            # dP[i][j] = sum_c dO[c][i] v[c][j]
            _bgemm(dO, qkv, dS, None, S, L, L, C, (C * L, 1, L), (C3 * L, L, 1), (L * L, L, 1), offB=2 * C * L)
            _call("vf_softmax_bwd", _ptr(P), _ptr(dS), _ptr(dS), S * L, L, _stream())
        if fused_dq and C % 64 == 0 and st.ATTN_DVDK:
            # dV and dK (below) in one launch of a kernel written for these two products (round 5)
            _call("vf_attention_dvdk", _ptr(qkv), _ptr(dO), _ptr(P), _ptr(dS), _ptr(dqkv), S, C, L, _stream(),
                  flops=4.0 * S * L * L * C)
            st._KIND_OVERRIDE = None
            return dqkv, None
        # dV[c][j] = sum_i dO[c][i] P[i][j]
        _bgemm(dO, P, dqkv, None, S, C, L, L, (C * L, L, 1), (L * L, L, 1), (C3 * L, L, 1), offC=2 * C * L)
        if not fused_dq:
            # dQ[c][i] = alpha sum_j k[c][j] dS[i][j]
            _bgemm(qkv, dS, dqkv, None, S, C, L, L, (C3 * L, L, 1), (L * L, 1, L), (C3 * L, L, 1), alpha,
                   offA=C * L, offC=0)
        # dK[c][j] = alpha sum_i q[c][i] dS[i][j]
        _bgemm(qkv, dS, dqkv, None, S, C, L, L, (C3 * L, L, 1), (L * L, L, 1), (C3 * L, L, 1), alpha,
               offA=0, offC=C * L)
        st._KIND_OVERRIDE = None
        return dqkv, None


def attention(qkv):
    return _AttentionFn.apply(qkv, torch.is_grad_enabled() and qkv.requires_grad)


class _ConcatFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        _check(a, b)
        S, Ca, H, W = a.shape
        Cb = b.shape[1]
        out = torch.empty(S, Ca + Cb, H, W, device=a.device, dtype=torch.float32)
        _call("vf_concat_channels", _ptr(a), _ptr(b), _ptr(out), S, Ca * H * W, Cb * H * W, 0, _stream())
        ctx.shapes = (a.shape, b.shape)
        return out

    @staticmethod
    def backward(ctx, dout):
        dout = _c(dout)
        sa, sb = ctx.shapes
        da = torch.empty(sa, device=dout.device, dtype=torch.float32)
        db = torch.empty(sb, device=dout.device, dtype=torch.float32)
        _call("vf_concat_channels", _ptr(da), _ptr(db), _ptr(dout), sa[0], da[0].numel(), db[0].numel(), 1,
                  _stream())
        return da, db


def concat_channels(a, b):
    return _ConcatFn.apply(a, b)
