"""The callers either side of the hot path that SURVEY §8(f) lists as "next" rows, restated for the
MI355X engine (no dataset / wandb / image-grid plumbing):

  * sampler drivers with the reference's real call shapes ... experiment.py:472-488 (extrapolate,
    7..23 conditioning views), :516-544 (autoregressive 24-step rollout, view_count 1 -> 24)
  * checkpoint wire format ................................. utils/checkpoint.py:31-72
  * eval reduction: PSNR + all_reduce(AVG) ................. utils/metrics.py:6-8, utils/dist.py:69-91
"""
import math
import os

import torch
import torch.distributed as dist


# ---- sampler drivers ---------------------------------------------------------------------------
@torch.no_grad()
def extrapolate(model, cond, angle, max_views=6, view_count=None, generator=None):
    """Generate with MORE views than the model was trained on (view_count ~ U[max_views+1, 24)).
    cond (B,23,3,H,W), angle (B,1) -> (generated_batch (B,1+k,3,H,W), logit_arr, weight_arr, view_count)."""
    B = cond.shape[0]
    if view_count is None:
        view_count = torch.randint(max_views + 1, 24, (B,), generator=generator)
    _, ret, logit_arr, weight_arr, _ = model(y_cond=cond, view_count=view_count, angle=angle, generate=True)
    return ret.clamp(0, 1), logit_arr, weight_arr, view_count


@torch.no_grad()
def autoregressive_rollout(model, first_view, steps=24):
    """Start from ONE view and synthesise the orbit view by view, feeding every sample back as an
    extra conditioning view (count = 1 .. steps; angle = 2*pi/24 * count).
    first_view (B,3,H,W) -> samples (B,steps,3,H,W)."""
    cond = first_view[:, None].contiguous()
    B = cond.shape[0]
    out = []
    for count in range(1, steps + 1):
        view_count = torch.full((B,), count)
        angle = torch.full((B, 1), 2 * math.pi / 24 * count, device=cond.device)
        *_, sample = model(y_cond=cond, view_count=view_count, angle=angle, generate=True)
        cond = torch.cat((cond, sample[:, None]), dim=1)
        out.append(sample)
    return torch.stack(out, dim=1)


# ---- checkpoint wire format --------------------------------------------------------------------
def save_checkpoint(path, model, optimizer, **extra):
    """{"model": state_dict, "optimizer": state_dict, + extra (it, t, run_id, ssim, psnr)} -- the
    reference's file layout, so either side can load the other's checkpoints."""
    out = dict(extra)
    out["model"] = model.state_dict()
    out["optimizer"] = optimizer.state_dict()
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    torch.save(out, path)


def load_checkpoint(path, model, optimizer=None, device=None):
    """Loads "model" (and "optimizer" if given and present); returns the remaining entries."""
    sd = torch.load(path, map_location=device, weights_only=False)
    model.load_state_dict(sd["model"])
    if optimizer is not None and "optimizer" in sd and sd["optimizer"].get("state"):
        optimizer.load_state_dict(sd["optimizer"])
    return {k: v for k, v in sd.items() if k not in ("model", "optimizer")}


# ---- eval reduction -----------------------------------------------------------------------------
def compute_psnr(generated, target):
    """20*log10(1/sqrt(mse)) per image; the per-image mean of squared differences is one HIP launch."""
    from . import ops
    return ops.psnr(generated, target)


def reduce_dict(d, average=True):
    """all_reduce every tensor of `d` across ranks (AVG by default); identity without a process group."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() < 2:
        return d
    op = dist.ReduceOp.AVG if average else dist.ReduceOp.SUM
    out = {}
    for k in sorted(d):
        v = d[k].clone()
        dist.all_reduce(v, op=op)
        out[k] = v
    return out
