"""Test infrastructure: evaluate the CPU oracle sample-parallel in spawned worker processes.

Samples are independent through the whole path (GroupNorm / attention are per view, the compose softmax is per sample,
the training loss is the mean of per-sample MSEs), so a batch of B samples is B one-sample problems.  One PyTorch-CPU
process stops scaling at 16-32 threads on the GPU box's 128 cores; P workers x 16 threads use the machine and cut the
oracle legs of the full-size parity tests by 4-6x.  Workers are SPAWNED (the test process has initialised the GPU;
forking it is not safe) and only ever import `oracle/` -- never the product.

Every function returns exactly what the serial oracle call returns (tests/test_oracle_golden.py::test_pool_* checks that
on the tiny net on CPU).
"""
import multiprocessing as mp
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_POOL = None
_POOL_SHAPE = None


def plan(nsamples):
    """(workers, threads per worker) for this host."""
    ncpu = os.cpu_count() or 8
    phys = max(1, ncpu // 2) if ncpu > 16 else ncpu       # the GPU box reports SMT threads; this container does not
    workers = max(1, min(nsamples, phys // 16 if phys >= 32 else 2, 8))
    threads = max(1, min(16, phys // workers))
    return workers, threads


def _init(threads):
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    torch.set_num_threads(threads)


def pool(nsamples):
    global _POOL, _POOL_SHAPE
    shape = plan(nsamples)
    if _POOL is None or _POOL_SHAPE != shape:
        close()
        _POOL = mp.get_context("spawn").Pool(shape[0], initializer=_init, initargs=(shape[1],))
        _POOL_SHAPE = shape
    return _POOL


def close():
    global _POOL, _POOL_SHAPE
    if _POOL is not None:
        _POOL.terminate()
        _POOL.join()
    _POOL, _POOL_SHAPE = None, None


def _slices(B, parts):
    parts = min(parts, B)
    edges = [round(i * B / parts) for i in range(parts + 1)]
    return [slice(edges[i], edges[i + 1]) for i in range(parts) if edges[i + 1] > edges[i]]


def _np(sd):
    return {k: v.detach().cpu().numpy() for k, v in sd.items()}


# ---- workers (module-level: picklable) -------------------------------------------------------
def _w_generate(a):
    from oracle import unet_ref, view_fusion_ref as vfr
    sd = {k: torch.from_numpy(v) for k, v in a["sd"].items()}
    sched = vfr.schedule_buffers(vfr.beta_schedule(**a["sched_kw"]))
    fn = lambda x, ang, lvl: unet_ref.unet_forward(sd, a["hp"], x, ang, lvl)
    with torch.no_grad():
        y, ret, la, wa, smp = vfr.generate(fn, sched, a["y_cond"], a["vc"], a["angle"], a["y_T"], a["z_seq"],
                                           a["sample_num"], a["weighting"])
    return y, ret, la, wa, smp


def _w_chain(a):
    """p_sample for t = t_hi .. t_lo (inclusive, descending) of the schedule; returns the final y and the per-step y."""
    from oracle import unet_ref, view_fusion_ref as vfr
    sd = {k: torch.from_numpy(v) for k, v in a["sd"].items()}
    sched = vfr.schedule_buffers(vfr.beta_schedule(**a["sched_kw"]))
    fn = lambda x, ang, lvl: unet_ref.unet_forward(sd, a["hp"], x, ang, lvl)
    y, B, keep = a["y_T"], a["y_T"].shape[0], []
    with torch.no_grad():
        for n, i in enumerate(range(a["t_hi"], a["t_lo"] - 1, -1)):
            t = torch.full((B,), i, dtype=torch.long)
            y, _, w = vfr.p_sample(fn, sched, y, a["y_cond"], a["vc"], a["angle"], t, a["z_seq"][n], a["weighting"])
            if (n + 1) % a["keep_every"] == 0:
                keep.append(y)
    return y, torch.stack(keep, 0), w


def _w_train(a):
    from oracle import unet_ref, view_fusion_ref as vfr
    sd = {k: torch.from_numpy(v).requires_grad_(True) for k, v in a["sd"].items()}
    sched = vfr.schedule_buffers(vfr.beta_schedule(**a["sched_kw"]))
    fn = lambda x, ang, lvl: unet_ref.unet_forward(sd, a["hp"], x, ang, lvl)
    total, nb = 0.0, a["y_0"].shape[0]
    for lo in range(0, nb, a["chunk"]):
        sl = slice(lo, min(nb, lo + a["chunk"]))
        part = vfr.train_loss(fn, sched, a["y_cond"][sl], a["vc"][sl], a["angle"][sl], a["y_0"][sl], a["t"][sl],
                              a["u"][sl], a["noise"][sl], a["weighting"])
        part = part * ((sl.stop - sl.start) / a["B"])
        part.backward()
        total += float(part.item())
    return total, {k: (v.grad.numpy() if v.grad is not None else None) for k, v in sd.items()}


# ---- front ends ----------------------------------------------------------------------------
def generate(sd, hp, sched_kw, y_cond, vc, angle, y_T, z_seq, sample_num=8, weighting=True):
    """oracle.view_fusion_ref.generate over sample slices; same 5-tuple."""
    B = y_cond.shape[0]
    p = pool(B)
    sdn = _np(sd)
    jobs = [dict(sd=sdn, hp=hp, sched_kw=sched_kw, y_cond=y_cond[s], vc=vc[s], angle=angle[s], y_T=y_T[s],
                 z_seq=z_seq[:, s], sample_num=sample_num, weighting=weighting) for s in _slices(B, _POOL_SHAPE[0])]
    out = p.map(_w_generate, jobs)
    y = torch.cat([o[0] for o in out])
    ret = torch.cat([o[1] for o in out])
    smp = torch.cat([o[4] for o in out])
    if not weighting:
        return y, ret, [], [], smp
    maxv = int(vc.max())
    # logits are stacked over views (dim 0, sample-major); weights are padded to the slice's own max view count
    la = torch.cat([o[2] for o in out])
    wa = torch.cat([torch.nn.functional.pad(o[3], (0, 0, 0, 0, 0, 0, 0, maxv - o[3].shape[2])) for o in out])
    return y, ret, la, wa, smp


def chain(sd, hp, sched_kw, y_cond, vc, angle, y_T, z_seq, t_hi, t_lo, keep_every=1, weighting=True):
    """p_sample for t = t_hi .. t_lo on every sample; z_seq[n] is the noise of the n-th step taken.
    -> (y_final, y after every keep_every-th step (K,B,3,H,W), weights of the last step)."""
    B = y_cond.shape[0]
    p = pool(B)
    sdn = _np(sd)
    jobs = [dict(sd=sdn, hp=hp, sched_kw=sched_kw, y_cond=y_cond[s], vc=vc[s], angle=angle[s], y_T=y_T[s],
                 z_seq=z_seq[:, s], t_hi=t_hi, t_lo=t_lo, keep_every=keep_every, weighting=weighting)
            for s in _slices(B, _POOL_SHAPE[0])]
    out = p.map(_w_chain, jobs)
    maxv = int(vc.max())
    w = None
    if weighting:
        w = torch.cat([torch.nn.functional.pad(o[2], (0, 0, 0, 0, 0, 0, 0, maxv - o[2].shape[1])) for o in out])
    return torch.cat([o[0] for o in out]), torch.cat([o[1] for o in out], dim=1), w


def train(sd, hp, sched_kw, y_cond, vc, angle, y_0, t, u, noise, weighting=True, chunk=2):
    """Oracle training loss and parameter gradients of a large batch -> (loss, {name: grad tensor})."""
    B = y_0.shape[0]
    p = pool(B)
    sdn = _np(sd)
    jobs = [dict(sd=sdn, hp=hp, sched_kw=sched_kw, y_cond=y_cond[s], vc=vc[s], angle=angle[s], y_0=y_0[s], t=t[s],
                 u=u[s], noise=noise[s], weighting=weighting, chunk=chunk, B=B) for s in _slices(B, _POOL_SHAPE[0])]
    out = p.map(_w_train, jobs)
    loss = float(np.sum([o[0] for o in out]))
    grads = {}
    for k in sdn:
        parts = [o[1][k] for o in out if o[1][k] is not None]
        grads[k] = torch.from_numpy(np.sum(np.stack(parts, 0), 0, dtype=np.float64)) if parts else None
    return loss, grads
