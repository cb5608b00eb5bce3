"""DDPM noise schedules (host side, float64 numpy -> six fp32 buffers).

Same formulas and buffer names as the reference (model/view_fusion.py:35-68, 321-362);
computed once per `set_new_noise_schedule`, never on the hot path.
"""
import math

import numpy as np
import torch

BUFFER_NAMES = ("gammas", "sqrt_recip_gammas", "sqrt_recipm1_gammas", "posterior_log_variance_clipped",
                "posterior_mean_coef1", "posterior_mean_coef2")


def make_beta_schedule(schedule, num_timesteps, linear_start=1e-6, linear_end=1e-2, cosine_s=8e-3):
    n = int(num_timesteps)

    def warm(frac):
        out = linear_end * np.ones(n, dtype=np.float64)
        k = int(n * frac)
        out[:k] = np.linspace(linear_start, linear_end, k, dtype=np.float64)
        return out

    table = {
        "linear": lambda: np.linspace(linear_start, linear_end, n, dtype=np.float64),
        "quad": lambda: np.linspace(linear_start ** 0.5, linear_end ** 0.5, n, dtype=np.float64) ** 2,
        "warmup10": lambda: warm(0.1),
        "warmup50": lambda: warm(0.5),
        "const": lambda: linear_end * np.ones(n, dtype=np.float64),
        "jsd": lambda: 1.0 / np.linspace(n, 1, n, dtype=np.float64),
    }
    if schedule in table:
        return table[schedule]()
    if schedule == "cosine":
        ts = torch.arange(n + 1, dtype=torch.float64) / n + cosine_s
        al = torch.cos(ts / (1 + cosine_s) * math.pi / 2).pow(2)
        al = al / al[0]
        return (1 - al[1:] / al[:-1]).clamp(max=0.999).numpy()
    raise NotImplementedError(schedule)


def schedule_tensors(betas, device):
    betas = np.asarray(betas, dtype=np.float64)
    alphas = 1.0 - betas
    gammas = np.cumprod(alphas, axis=0)
    prev = np.append(1.0, gammas[:-1])
    with np.errstate(divide="ignore", invalid="ignore"):
        var = betas * (1.0 - prev) / (1.0 - gammas)
        vals = (gammas, np.sqrt(1.0 / gammas), np.sqrt(1.0 / gammas - 1), np.log(np.maximum(var, 1e-20)),
                betas * np.sqrt(prev) / (1.0 - gammas), (1.0 - prev) * np.sqrt(alphas) / (1.0 - gammas))
    return {k: torch.tensor(v, dtype=torch.float32, device=device) for k, v in zip(BUFFER_NAMES, vals)}
