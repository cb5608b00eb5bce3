"""Oracle: functional fp32 CPU restatement of the ViewFusion DDPM wrapper.

TEST INFRASTRUCTURE -- see oracle/__init__.py.  Follows (by behaviour)
/root/reference/model/view_fusion.py:

  * beta schedules (float64 numpy) ............. view_fusion.py:321-362
  * the six schedule buffers ................... view_fusion.py:35-68
  * q_sample ................................... view_fusion.py:162-164
  * ragged view stacking ....................... view_fusion.py:244-256 / 95-109
  * softmax-over-views compose / mean ablation . view_fusion.py:265-296 / 116-150
  * training loss .............................. view_fusion.py:229-300
  * posterior mean / variance, p_sample ........ view_fusion.py:70-84, 152-177
  * reverse loop ............................... view_fusion.py:179-214

All randomness is an explicit INPUT (t, u, noise, z) so that the HIP path and
the oracle can be driven with identical tensors.
"""
import math

import numpy as np
import torch

SCHEDULE_KEYS = ("gammas", "sqrt_recip_gammas", "sqrt_recipm1_gammas",
                 "posterior_log_variance_clipped", "posterior_mean_coef1",
                 "posterior_mean_coef2")


def beta_schedule(schedule, num_timesteps, linear_start=1e-6, linear_end=1e-2, cosine_s=8e-3):
    T = int(num_timesteps)
    if schedule == "linear":
        return np.linspace(linear_start, linear_end, T, dtype=np.float64)
    if schedule == "quad":
        return np.linspace(linear_start ** 0.5, linear_end ** 0.5, T, dtype=np.float64) ** 2
    if schedule in ("warmup10", "warmup50"):
        frac = 0.1 if schedule == "warmup10" else 0.5
        b = np.full(T, linear_end, dtype=np.float64)
        w = int(T * frac)
        b[:w] = np.linspace(linear_start, linear_end, w, dtype=np.float64)
        return b
    if schedule == "const":
        return np.full(T, linear_end, dtype=np.float64)
    if schedule == "jsd":
        return 1.0 / np.linspace(T, 1, T, dtype=np.float64)
    if schedule == "cosine":
        # the reference evaluates this branch with torch float64 ops
        s = torch.arange(T + 1, dtype=torch.float64) / T + cosine_s
        a = torch.cos(s / (1 + cosine_s) * math.pi / 2).pow(2)
        a = a / a[0]
        return (1 - a[1:] / a[:-1]).clamp(max=0.999).numpy()
    raise NotImplementedError(schedule)


def schedule_buffers(betas):
    """float64 betas (T,) -> dict of the six fp32 buffers."""
    betas = np.asarray(betas, dtype=np.float64)
    alphas = 1.0 - betas
    g = np.cumprod(alphas, axis=0)
    g_prev = np.append(1.0, g[:-1])
    var = betas * (1.0 - g_prev) / (1.0 - g)
    out = {
        "gammas": g,
        "sqrt_recip_gammas": np.sqrt(1.0 / g),
        "sqrt_recipm1_gammas": np.sqrt(1.0 / g - 1),
        "posterior_log_variance_clipped": np.log(np.maximum(var, 1e-20)),
        "posterior_mean_coef1": betas * np.sqrt(g_prev) / (1.0 - g),
        "posterior_mean_coef2": (1.0 - g_prev) * np.sqrt(alphas) / (1.0 - g),
    }
    return {k: torch.tensor(v, dtype=torch.float32) for k, v in out.items()}


def q_sample(y_0, gamma, noise):
    """gamma broadcastable (B,1,1,1)."""
    return gamma.sqrt() * y_0 + (1 - gamma).sqrt() * noise


def stack_views(y_cond, view_count, y_t, level, angle):
    """(B,Nmax,3,H,W),(B,) -> UNet input (S,6,H,W), angle (S,1), level (S,1)."""
    vc = [int(v) for v in view_count]
    cond = torch.cat([y_cond[i, :v] for i, v in enumerate(vc)], dim=0)
    rep = torch.tensor(vc, dtype=torch.long)
    x = torch.cat([cond, torch.repeat_interleave(y_t, rep, dim=0)], dim=1)
    return x, torch.repeat_interleave(angle, rep, dim=0), torch.repeat_interleave(level, rep, dim=0)


def compose(unet_out, view_count, weighting=True):
    """(S,6|3,H,W) -> composed noise (B,3,H,W), logits (S,3,H,W)|None,
    weights (B,maxV,3,H,W)|None (zero where a sample has fewer views)."""
    vc = [int(v) for v in view_count]
    B, vmax = len(vc), max(vc)
    eps = unet_out[:, :3]
    _, _, H, W = eps.shape
    off = np.concatenate([[0], np.cumsum(vc)])
    if not weighting:
        out = torch.stack([eps[off[b]:off[b + 1]].mean(dim=0) for b in range(B)])
        return out, None, None
    logits = unet_out[:, 3:]
    lpad = eps.new_full((B, vmax, 3, H, W), float("-inf"))
    epad = eps.new_zeros((B, vmax, 3, H, W))
    for b in range(B):
        lpad[b, :vc[b]] = logits[off[b]:off[b + 1]]
        epad[b, :vc[b]] = eps[off[b]:off[b + 1]]
    w = torch.softmax(lpad, dim=1)
    return (epad * w).sum(dim=1), logits, w


def train_loss(unet_fn, sched, y_cond, view_count, angle, y_0, t, u, noise, weighting=True):
    """Training forward with explicit randomness.

    t (B,) long in [1,T); u (B,1) uniform in [0,1); noise like y_0.
    unet_fn(x, angle_s, level_s) -> (S,Cout,H,W).  Returns scalar MSE loss.
    """
    g = sched["gammas"]
    g_lo = g[t - 1].reshape(-1, 1)
    g_hi = g[t].reshape(-1, 1)
    level = (g_hi - g_lo) * u + g_lo                                   # (B,1)
    y_noisy = q_sample(y_0, level.reshape(-1, 1, 1, 1), noise)
    x, ang_s, lvl_s = stack_views(y_cond, view_count, y_noisy, level, angle)
    out = unet_fn(x, ang_s, lvl_s)
    noise_hat, _, _ = compose(out, view_count, weighting)
    return torch.nn.functional.mse_loss(noise, noise_hat)


def p_mean_variance(unet_fn, sched, y_t, y_cond, view_count, angle, t, clip_denoised=True,
                    weighting=True):
    """t (B,) long.  Returns mean, log-variance (B,1,1,1), logits, weights."""
    level = sched["gammas"][t].reshape(-1, 1)
    x, ang_s, lvl_s = stack_views(y_cond, view_count, y_t, level, angle)
    out = unet_fn(x, ang_s, lvl_s)
    eps, logits, w = compose(out, view_count, weighting)
    pick = lambda k: sched[k][t].reshape(-1, 1, 1, 1)
    y0 = pick("sqrt_recip_gammas") * y_t - pick("sqrt_recipm1_gammas") * eps
    if clip_denoised:
        y0 = y0.clamp(-1.0, 1.0)
    mean = pick("posterior_mean_coef1") * y0 + pick("posterior_mean_coef2") * y_t
    return mean, pick("posterior_log_variance_clipped"), logits, w


def p_sample(unet_fn, sched, y_t, y_cond, view_count, angle, t, z, weighting=True):
    """One reverse step with injected gaussian z (ignored -> zero when all t == 0)."""
    mean, logvar, logits, w = p_mean_variance(unet_fn, sched, y_t, y_cond, view_count, angle, t,
                                              True, weighting)
    if not bool((t > 0).any()):
        z = torch.zeros_like(y_t)
    return mean + z * (0.5 * logvar).exp(), logits, w


def generate(unet_fn, sched, y_cond, view_count, angle, y_T, z_seq, sample_num=8, weighting=True):
    """Full reverse chain.  z_seq[i] is the noise used at step index i (i = T-1 .. 0).

    Returns the reference's 5-tuple (y_0, ret_arr, logit_arr, weight_arr, samples).
    """
    T = sched["gammas"].shape[0]
    assert T > sample_num
    every = T // sample_num
    B = y_cond.shape[0]
    y = y_T
    ret, logit_arr, weight_arr = [y], [], []
    for i in reversed(range(T)):
        t = torch.full((B,), i, dtype=torch.long)
        y, logits, w = p_sample(unet_fn, sched, y, y_cond, view_count, angle, t, z_seq[i], weighting)
        if i % every == 0:
            ret.append(y)
            logit_arr.append(logits)
            weight_arr.append(w)
    ret = torch.stack(ret, dim=1)
    if weighting:
        logit_arr = torch.stack(logit_arr, dim=1)
        weight_arr = torch.stack(weight_arr, dim=1)
    return y, ret, logit_arr, weight_arr, ret[:, -1]
