"""Timing of the reverse-diffusion sampler (BASELINE config C5): small UNet 64x64, T=1000 test
schedule, N in {1,6,12} conditioning views.  Used by bench.py and tools/bench_sampler.py."""
import time

import torch

from . import _lib, train

LAST_LAUNCHES_PER_STEP = None      # C-ABI launcher calls of one reverse step of the last timed configuration


def time_sampler(batch, views, steps=None, use_graph=True, device="cuda:0", model=None):
    """Returns dict(sampled_views_per_sec, ms_per_step, ...).  `steps`=None runs the full T=1000
    chain; otherwise `steps` reverse steps are timed and extrapolated linearly (the loop is strictly
    sequential with constant step cost)."""
    dev = torch.device(device)
    if model is None:
        model = train.build_model(device=device, phase="test")
    T = model.num_timesteps
    b = train.synthetic_batch(batch, views, 64, dev, seed=0)
    if steps is not None and steps < T:
        # shorten the chain but keep the real per-step work
        full = model.num_timesteps
        model.num_timesteps = steps
        try:
            return _run(model, b, batch, views, steps, full, use_graph)
        finally:
            model.num_timesteps = full
    return _run(model, b, batch, views, T, T, use_graph)


def _run(model, b, batch, views, steps, full_T, use_graph):
    global LAST_LAUNCHES_PER_STEP
    sample_num = min(8, steps - 1)
    with torch.no_grad():                      # one eager reverse step, counted (also packs the weights)
        y = torch.randn_like(b["y_0"])
        t = torch.full((batch,), steps - 1, device=y.device, dtype=torch.long)
        model.p_sample(y, b["y_cond"], b["view_count"], b["angle"], t)
        c0 = _lib.N_CALLS
        model.p_sample(y, b["y_cond"], b["view_count"], b["angle"], t)
        LAST_LAUNCHES_PER_STEP = _lib.N_CALLS - c0
    model.generate(b["y_cond"], b["view_count"], b["angle"], sample_num=sample_num, use_graph=use_graph)  # warm-up
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    model.generate(b["y_cond"], b["view_count"], b["angle"], sample_num=sample_num, use_graph=use_graph)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    per_step = dt / steps
    return dict(batch=batch, views=views, steps_timed=steps, T=full_T, graph=use_graph,
                ms_per_step=per_step * 1e3, sampled_views_per_sec=batch / (per_step * full_T),
                view_unet_evals_per_sec=batch * views / per_step)
