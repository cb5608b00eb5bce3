"""The ViewFusion ops around the UNet: view stacking + q_sample, compose / weighted-noise loss, the sampler tail, PSNR
(reference model/view_fusion.py:70-177, 229-300; utils/metrics.py:6-8)."""
import ctypes

import torch

from .state import st
from .core import _c, _call, _check, _ptr, _stream


# ---------------------------------------------------------------------------------------------
# ViewFusion glue


def view_offsets(view_count, device):
    """view_count (list / CPU tensor / device tensor) -> (off int32 [B+1] on device, S, maxV).

    A CPU-side view_count (what the harness and INTEGRATION.md hand over) needs no device sync.  A DEVICE tensor
    (what the reference's loops produce with `.to(device)`, experiment.py:277-279, 476-478) must be read back once,
    because S sizes every allocation -- the same one D2H the reference pays in `cumsum(view_count).tolist()`
    (view_fusion.py:95, 244); the result is remembered per tensor object and version, so a caller that drives
    `p_sample` / `p_mean_variance` step by step with the same device tensor syncs once, not once per step
    (`generate` resolves it once per call anyway).
    """
    if torch.is_tensor(view_count) and view_count.is_cuda:
        for ent in st._VC_CACHE:
            if ent[0] is view_count and ent[1] == view_count._version and ent[2][0].device == device:
                return ent[2]
        vc = view_count.detach().cpu().tolist()
    elif torch.is_tensor(view_count):
        vc = view_count.detach().tolist()
    else:
        vc = [int(v) for v in view_count]
    off = [0]
    for v in vc:
        if v < 1:
            raise ValueError("every sample needs at least one conditioning view")
        off.append(off[-1] + int(v))
    t = torch.tensor(off, dtype=torch.int32)
    if device.type == "cuda":
        t = t.pin_memory().to(device, non_blocking=True)
    out = (t, off[-1], max(vc))
    if torch.is_tensor(view_count) and view_count.is_cuda:
        st._VC_CACHE.insert(0, (view_count, view_count._version, out))
        del st._VC_CACHE[4:]
    return out


def gather_level(gammas, t, u=None):
    """level[b] = gammas[t[b]]  or the training draw (g[t]-g[t-1])*u + g[t-1]."""
    _check(gammas, u)
    t = _c(t.to(torch.int64))
    B = t.numel()
    level = torch.empty(B, device=gammas.device, dtype=torch.float32)
    _call("vf_gather_level", _ptr(gammas), ctypes.c_void_p(t.data_ptr()), _ptr(u), _ptr(level), B, _stream())
    return level


def stack_views(y_cond, y_t, noise, level, angle, off, S, x=None, copy_cond=True):
    """Ragged stacking (+ optional q_sample): -> x (S,Cc+3,H,W), level_s (S,1), angle_s (S,1); Cc = y_cond's
    channel count (3, or 6 for the `relative` configs)."""
    y_cond, y_t = _c(y_cond), _c(y_t)
    angle = _c(angle.reshape(-1).float())
    _check(y_cond, y_t, noise, level, angle)
    B, Nmax, Cc, H, W = y_cond.shape
    if y_t.shape[1] != 3:
        raise ValueError(f"the noisy target must have 3 channels, got {tuple(y_t.shape)}")
    if x is None:
        x = torch.empty(S, Cc + 3, H, W, device=y_cond.device, dtype=torch.float32)
    ls = torch.empty(S, 1, device=y_cond.device, dtype=torch.float32)
    as_ = torch.empty(S, 1, device=y_cond.device, dtype=torch.float32)
    _call("vf_stack_views", _ptr(y_cond), _ptr(y_t), _ptr(noise), _ptr(level), _ptr(angle),
              ctypes.c_void_p(off.data_ptr()), _ptr(x), _ptr(ls), _ptr(as_), B, Nmax, Cc, H * W, S, int(copy_cond),
              _stream())
    return x, ls, as_


class _ComposeLossFn(torch.autograd.Function):
    """MSE(target, compose(unet_out)) fused: softmax over each sample's views (or mean)."""

    @staticmethod
    def forward(ctx, out, target, off, B, weighting):
        _check(out, target)
        S, Cout, H, W = out.shape
        nh = torch.empty(B, 3, H, W, device=out.device, dtype=torch.float32)
        part = torch.empty(B * 64 + 1, device=out.device, dtype=torch.float32)
        loss = part[B * 64:]
        _call("vf_compose_fwd", _ptr(out), ctypes.c_void_p(off.data_ptr()), _ptr(target), _ptr(nh), None,
                  _ptr(part), _ptr(loss), B, Cout, H * W, 0, int(weighting), _stream())
        ctx.save_for_backward(out, target, nh, off)
        ctx.B, ctx.weighting = B, int(weighting)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, gloss):
        out, target, nh, off = ctx.saved_tensors
        S, Cout, H, W = out.shape
        gloss = _c(gloss.reshape(1).float())
        dout = torch.empty_like(out)
        _call("vf_compose_mse_bwd", _ptr(out), ctypes.c_void_p(off.data_ptr()), _ptr(target), _ptr(nh),
                  _ptr(gloss), _ptr(dout), ctx.B, Cout, H * W, ctx.weighting, _stream())
        return dout, None, None, None, None


def compose_mse_loss(unet_out, target_noise, off, B, weighting):
    return _ComposeLossFn.apply(unet_out, _c(target_noise), off, B, weighting)


def compose(unet_out, off, B, max_views, weighting, want_weights=True):
    """Inference compose: -> noise (B,3,H,W), weights (B,maxV,3,H,W) | None."""
    _check(unet_out)
    S, Cout, H, W = unet_out.shape
    nh = torch.empty(B, 3, H, W, device=unet_out.device, dtype=torch.float32)
    wts = None
    if weighting and want_weights:
        wts = torch.empty(B, max_views, 3, H, W, device=unet_out.device, dtype=torch.float32)
    _call("vf_compose_fwd", _ptr(unet_out), ctypes.c_void_p(off.data_ptr()), None, _ptr(nh), _ptr(wts), None,
              None, B, Cout, H * W, max_views, int(weighting), _stream())
    return nh, wts


def p_sample_tail(unet_out, off, y_t, z, t, sched, B, max_views, weighting, clip=True, want_weights=True,
                  want_mean=False, inplace=False):
    """Fused compose -> y0_hat -> clamp -> posterior mean -> + z*sigma.
    Returns (y_next, mean | None, weights | None)."""
    _check(unet_out, y_t, z)
    S, Cout, H, W = unet_out.shape
    t = _c(t.to(torch.int64))
    y_next = y_t if inplace else torch.empty_like(y_t)     # elementwise: safe to overwrite y_t
    mean = torch.empty_like(y_t) if want_mean else None
    wts = None
    if weighting and want_weights:
        wts = torch.empty(B, max_views, 3, H, W, device=y_t.device, dtype=torch.float32)
    _call("vf_p_sample_tail", _ptr(unet_out), ctypes.c_void_p(off.data_ptr()), _ptr(y_t), _ptr(z),
              ctypes.c_void_p(t.data_ptr()), _ptr(sched["sqrt_recip_gammas"]), _ptr(sched["sqrt_recipm1_gammas"]),
              _ptr(sched["posterior_log_variance_clipped"]), _ptr(sched["posterior_mean_coef1"]),
              _ptr(sched["posterior_mean_coef2"]), _ptr(y_next), _ptr(mean), _ptr(wts), B, Cout, H * W, max_views,
              int(weighting), int(clip), _stream())
    return y_next, mean, wts


def psnr(generated, target):
    """Per-image PSNR (B,) of (B,C,H,W) tensors in [0,1]."""
    generated, target = _c(generated), _c(target)
    _check(generated, target)
    B = generated.shape[0]
    out = torch.empty(B, device=generated.device, dtype=torch.float32)
    _call("vf_psnr", _ptr(generated), _ptr(target), _ptr(out), B, generated[0].numel(), _stream())
    return out
