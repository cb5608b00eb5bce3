"""The callers either side of the hot path that SURVEY §8(f) lists as "next" rows, restated for the
MI355X engine (no dataset / wandb / image-grid plumbing):

  * sampler drivers with the reference's real call shapes ... experiment.py:472-488 (extrapolate,
    7..23 conditioning views), :516-544 (autoregressive 24-step rollout, view_count 1 -> 24),
    :580-599 (weight animation: B=24 target angles x N=6 views in one call)
  * checkpoint wire format ................................. utils/checkpoint.py:31-72
  * eval reduction: PSNR + all_reduce(AVG) + barriers ...... utils/metrics.py:6-8, utils/dist.py:69-91,
    experiment.py:314-370
"""
import contextlib
import math
import os

import torch
import torch.distributed as dist


@contextlib.contextmanager
def _eval_mode(model):
    """The reference's eval / inference entry points call `self.model.eval()` first (experiment.py:316): Dropout (the
    only mode-dependent layer of the UNet) must be off while sampling.  The previous mode is restored on exit, so a
    Trainer that finds the model in training mode keeps skipping its (slow) Module.train() walk."""
    was = model.training
    if was:
        model.eval()
    try:
        yield
    finally:
        if was:
            model.train()


# ---- sampler drivers ---------------------------------------------------------------------------
# Each driver reaches the sampler through model(..., generate=True), like the reference's (DDP wraps forward), and
# takes the optional injected randomness of ViewFusion.forward (y_t = the start noise, z_seq[i] = the noise of
# reverse step i) so that its output can be compared with the CPU oracle.
@torch.no_grad()
def extrapolate(model, cond, angle, max_views=6, view_count=None, generator=None, **inject):
    """Generate with MORE views than the model was trained on (view_count ~ U[max_views+1, 24)),
    experiment.py:472-488.
    cond (B,23,3,H,W), angle (B,1) -> (generated_batch (B,1+k,3,H,W), logit_arr, weight_arr, view_count)."""
    B = cond.shape[0]
    if view_count is None:
        view_count = torch.randint(max_views + 1, 24, (B,), generator=generator)
    with _eval_mode(model):
        _, ret, logit_arr, weight_arr, _ = model(y_cond=cond, view_count=view_count, angle=angle, generate=True,
                                                 **inject)
    return ret.clamp(0, 1), logit_arr, weight_arr, view_count


@torch.no_grad()
def autoregressive_rollout(model, first_view, steps=24, y_t=None, z_seq=None):
    """Start from ONE view and synthesise the orbit view by view, feeding every sample back as an
    extra conditioning view (count = 1 .. steps; angle = 2*pi/24 * count), experiment.py:516-544.
    first_view (B,3,H,W) -> samples (B,steps,3,H,W).  y_t / z_seq: per-count lists of injected randomness."""
    cond = first_view[:, None].contiguous()
    B = cond.shape[0]
    out = []
    with _eval_mode(model):
        for count in range(1, steps + 1):
            view_count = torch.full((B,), count)
            angle = torch.full((B, 1), 2 * math.pi / 24 * count, device=cond.device)
            inject = dict(y_t=None if y_t is None else y_t[count - 1],
                          z_seq=None if z_seq is None else z_seq[count - 1])
            *_, sample = model(y_cond=cond, view_count=view_count, angle=angle, generate=True, **inject)
            cond = torch.cat((cond, sample[:, None]), dim=1)
            out.append(sample)
    return torch.stack(out, dim=1)


@torch.no_grad()
def orbit_frames(model, all_views, n=24, **inject):
    """The weight-animation ("GIF") call shape, experiment.py:580-599: ONE object's `all_views` (24,3,H,W); every one
    of the n target angles 2*pi/n*i is generated from the same 6 conditioning views (every 4th view), i.e. one
    generate() call at B = n, N = 6.
    -> (generated_batch (n,1+k,3,H,W) clamped to [0,1], logit_arr, weight_arr, cond_views (n,6,3,H,W), angles (n,1));
    the frame i of the animation shows weight_arr[i] next to cond_views[i] and generated_batch[i]."""
    assert all_views.shape[0] == 24 and n % 24 == 0
    dev = all_views.device
    angles = torch.tensor([2 * math.pi / n * i for i in range(n)], dtype=torch.float32, device=dev).unsqueeze(1)
    target = torch.repeat_interleave(all_views, n // 24, dim=0)
    cond_views = torch.stack([all_views[::4]] * target.shape[0], dim=0).contiguous()
    view_count = torch.full((target.shape[0],), cond_views.shape[1])
    with _eval_mode(model):
        _, ret, logit_arr, weight_arr, _ = model(y_cond=cond_views, view_count=view_count, angle=angles, generate=True,
                                                 **inject)
    return ret.clamp(0, 1), logit_arr, weight_arr, cond_views, angles


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
    """all_reduce every tensor of `d` across ranks (AVG by default), keys in sorted order as the reference
    (utils/dist.py:69-91); identity without a process group.  RCCL has ncclAvg; gloo only SUM, so there the mean is
    SUM / world."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() < 2:
        return d
    native_avg = dist.get_backend() == "nccl"
    out = {}
    for k in sorted(d):
        v = d[k].clone()
        if average and native_avg:
            dist.all_reduce(v, op=dist.ReduceOp.AVG)
        else:
            dist.all_reduce(v, op=dist.ReduceOp.SUM)
            if average:
                v /= dist.get_world_size()
        out[k] = v
    return out


def _barrier():
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()


@torch.no_grad()
def evaluate(model, batches, max_views=6, generator=None, extra_metrics=None, **inject):
    """The eval reduction of Experiment.eval (experiment.py:314-370): every rank generates its shard of the validation
    batches (view_count ~ U[1, max_views] per sample), PSNR per image on the GPU (utils/metrics.py:6-8), mean over the
    rank's images, barrier, all_reduce(AVG) of the scalars, barrier.  `batches`: iterable of dicts with target (B,3,H,W),
    cond (B,>=max_views,3,H,W), angle (B,1) and optionally view_count.  extra_metrics: {name: fn(generated, target) ->
    (B,)} for third-party metrics (the reference's SSIM).  Returns the reduced dict of 0-d tensors."""
    gen, gt = [], []
    with _eval_mode(model):                        # Experiment.eval: self.model.eval() (experiment.py:316)
        for b in batches:
            vc = b.get("view_count")
            if vc is None:
                vc = torch.randint(1, max_views + 1, (b["target"].shape[0],), generator=generator)
            *_, samples = model(y_cond=b["cond"], view_count=vc, angle=b["angle"], generate=True, **inject)
            gen.append(samples)
            gt.append(b["target"])
    _barrier()
    metrics = {"psnr": compute_psnr}
    metrics.update(extra_metrics or {})
    out = {k: torch.cat([fn(a, t) for a, t in zip(gen, gt)]).mean() for k, fn in metrics.items()}
    _barrier()
    out = reduce_dict(out)
    _barrier()
    return out
