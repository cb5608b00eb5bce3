"""SURVEY 8(e) on real kernels: two ranks (two processes sharing the one GPU of the box, gloo as the transport --
RCCL refuses two ranks on one device) train the HIP model through the gradient arena; the averaged gradients of
every iteration must equal single-process gradients on the concatenated batch, on the copy path (first
iteration) and on the zero-copy path (backward kernels writing into the all-reduce buffer while earlier
segments are already being reduced), and the replicas must stay bit-identical under Adam.  RCCL itself is
exercised with a world of one rank in test_gpu_model.py and by bench.py --gpus N on a multi-GPU node."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from conftest import TINY

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCHED = dict(schedule="linear", num_timesteps=2000, linear_start=1e-6, linear_end=1e-2)
B, N, HW = 12, 6, 16


def _batch(step, rank, ragged=True):
    g = torch.Generator().manual_seed(100 * step + rank)
    vc = torch.randint(1, N + 1, (B,), generator=g)
    if not ragged:                                # one geometry for every step: the graph variant replays it
        vc = torch.full((B,), N)
    return dict(y_0=torch.rand(B, 3, HW, HW, generator=g), y_cond=torch.rand(B, N, 3, HW, HW, generator=g),
                angle=2 * np.pi / 24 * torch.randint(0, 24, (B, 1), generator=g).float(),
                view_count=vc,
                noise=torch.randn(B, 3, HW, HW, generator=g), t=torch.randint(1, 2000, (B,), generator=g),
                u=torch.rand(B, 1, generator=g))


def _to(b, dev):
    return {k: (v if k == "view_count" else v.to(dev)) for k, v in b.items()}


def _model(dev):
    from view_fusion_amd import UNet, ViewFusion
    from view_fusion_amd.utils import deterministic_fill_
    net = UNet(**TINY)
    deterministic_fill_(net.state_dict())
    vf = ViewFusion(net.to(dev), {"train": SCHED}, True, True)
    vf.set_new_noise_schedule(device=dev, phase="train")
    return vf


def _run(tr, vf, batches, dev):
    grads = []
    for b in batches:
        b = _to(b, dev)
        extra = {k: b.pop(k) for k in ("noise", "t", "u")}
        if tr.arena is not None:
            tr.arena.copied = 0
        tr.step(b, **extra)
        grads.append([p.grad.detach().cpu().clone() for p in vf.parameters()])
    return grads


def _worker(rank, world, port, out, lr, kind, graph, steps, env=None):
    sys.path.insert(0, ROOT)
    env = dict(env or {})
    inject = env.pop("inject_capture_failure", None)     # "<rank>:<mode>": a Trainer attribute, not an environment knob
    leave_after = int(env.pop("leave_after", 0))         # this many iterations, then rank 1 stops taking part
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC (read when the HIP runtime initialises)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      VF_REDUCER=kind, **(env or {}))
    import torch.distributed as dist
    from view_fusion_amd import train
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    vf = _model(dev)
    tr = train.Trainer(vf, world=world, lr_warmup=1, graph=graph)
    tr.inject_capture_failure = inject
    assert (tr.arena is not None) == (kind in ("arena", "xgmi"))
    tr.it, tr.sched.peak_lr = 0, lr
    copied = []
    grads = []
    error = None
    for s in range(steps):
        if leave_after and rank == 1 and s >= leave_after:
            break
        try:
            grads += _run(tr, vf, [_batch(s, rank, ragged=not graph)], dev)
        except Exception as e:      # noqa: BLE001
            error = f"{type(e).__name__}: {e}"
            break
        copied.append(tr.arena.copied if tr.arena is not None else 0)
    torch.cuda.synchronize()
    sd = tr.opt.state_dict()["state"]
    out[rank] = dict(grads=grads, copied=copied, params=[p.detach().cpu().clone() for p in vf.parameters()],
                     graph_steps=tr.graph_steps, mode=tr.mode, demotions=tr.demotions, info=tr.dist_info(), error=error,
                     adam=[(float(v["step"]), v["exp_avg"].cpu(), v["exp_avg_sq"].cpu()) for v in sd.values()] if sd else [])
    dist.barrier()                  # (a rank's arena stays mapped in its peer until both are done)
    if hasattr(tr.arena, "close"):
        tr.arena.close()
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _spawn(lr, kind="arena", graph=False, steps=3, env=None, world=2):
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), out, lr, kind, graph, steps, env), nprocs=world, join=True)
    return tuple(out[r] for r in range(world))


def _check_against_global_batch(r0, r1, STEPS, ragged):
    from view_fusion_amd import train
    dev = torch.device("cuda:0")
    vf = _model(dev)
    tr = train.Trainer(vf, world=1, lr_warmup=1, graph=False)
    tr.it, tr.sched.peak_lr = 0, 0.0
    glob = []
    for s in range(STEPS):
        a, b = _batch(s, 0, ragged), _batch(s, 1, ragged)
        glob.append({k: torch.cat([a[k], b[k]]) for k in a})
    ref = _run(tr, vf, glob, dev)
    names = [k for k, _ in vf.named_parameters()]
    for s in range(STEPS):
        for k, g0, g1, gr in zip(names, r0["grads"][s], r1["grads"][s], ref[s]):
            assert torch.equal(g0, g1), (s, k)                       # both ranks hold the same average
            err = float((g0.double() - gr.double()).norm())
            assert err <= 2e-5 * float(gr.double().norm()) + 1e-7 * gr.numel() ** 0.5, (s, k, err)


# arena: eager launches / the iteration replayed as a HIP graph (gloo cannot be captured: the graph ends with the
# backward pass, the segment all-reduces and the Adam launch follow each replay -- train.Trainer, "split" mode);
# ddp: torch's DistributedDataParallel around the same HIP model (VF_REDUCER=ddp, the fallback of tools/scale_run.md)
@pytest.mark.parametrize("kind,graph,STEPS", [("arena", False, 3), ("arena", True, 6), ("ddp", False, 3)])
def test_two_rank_arena_gradients_match_the_global_batch(kind, graph, STEPS):
    from view_fusion_amd import train
    r0, r1 = _spawn(0.0, kind, graph, STEPS)      # lr 0: every iteration starts from the same parameters
    assert r0["copied"][1:] == [0] * (STEPS - 1) and r1["copied"][1:] == [0] * (STEPS - 1), (r0["copied"], r1["copied"])
    want = STEPS - 1 - train.Trainer.GRAPH_AFTER if graph else 0
    assert r0["graph_steps"] == r1["graph_steps"] == want, (r0["graph_steps"], r1["graph_steps"])
    assert r0["mode"] == r1["mode"] == ("split" if graph else "eager") and r0["info"]["world_size"] == 2
    _check_against_global_batch(r0, r1, STEPS, not graph)


def test_capture_failure_on_one_rank_steps_every_rank_down():
    """Rank 1's training-step capture fails (injected) while rank 0's succeeds.  Until the next agreement point the
    ranks run different launch modes -- rank 0 replays, rank 1 enqueues eagerly: same collectives in the same order --
    then BOTH step down to the eager mode together; the averaged gradients of every iteration, before, during and
    after, equal single-process gradients on the concatenated batch, and the replicas end bit-identical."""
    STEPS = 9
    r0, r1 = _spawn(0.0, "arena", True, STEPS, env=dict(inject_capture_failure="1:split"))
    # iterations it = 1..9: layout, two eager sightings, capture at it = 4 (rank 1 fails and raises the flag), rank 0
    # replays it = 4, 5; the flag is read at it = 6 (= 2 + 4): everybody eager from there on
    assert r0["graph_steps"] == 2 and r1["graph_steps"] == 0, (r0["graph_steps"], r1["graph_steps"])
    assert r0["mode"] == r1["mode"] == "eager" and r0["demotions"] == r1["demotions"] == 1
    _check_against_global_batch(r0, r1, STEPS, False)
    r0, r1 = _spawn(1e-4, "arena", True, STEPS, env=dict(inject_capture_failure="1:split"))
    for a, b in zip(r0["params"], r1["params"]):
        assert torch.equal(a, b)


def _skip_without_ipc(*results):
    """The xgmi reducer maps peer memory with HIP IPC (dmabuf mode on this driver).  A box where the three IPC memory calls
    themselves fail cannot run it: that is a property of the box, not of the reducer -- skip, loudly."""
    for r in results:
        e = r.get("error") or ""
        if any(k in e for k in ("vf_xgmi_export failed", "vf_xgmi_open failed", "vf_xgmi_alloc failed")):
            pytest.skip("HIP IPC is not available on this box: " + e)


def test_xgmi_reducer_matches_the_arena_bit_for_bit():
    """SURVEY 8f rank 1 (VF_REDUCER=xgmi): the one-shot all-reduce over IPC-mapped peer arenas fused with Adam
    (csrc/xgmi.hip), two processes sharing the GPU.  Over six iterations the averaged gradients every rank sees, the
    parameters and the Adam state (step counts, exp_avg, exp_avg_sq -- torch's layout) equal the gradient arena's
    (gloo all-reduce + the multi-tensor Adam launch) bit for bit; the replicas equal each other; no gradient is copied
    from the second iteration on; and with lr = 0 the averaged gradients equal single-process gradients on the
    concatenated batch."""
    STEPS = 6
    a0, a1 = _spawn(1e-4, "arena", False, STEPS)
    x0, x1 = _spawn(1e-4, "xgmi", False, STEPS)
    _skip_without_ipc(x0, x1)
    assert x0["error"] is None and x1["error"] is None, (x0["error"], x1["error"])
    assert x0["info"]["reducer"] == "xgmi" and x0["mode"] == x1["mode"] == "eager"
    assert x0["copied"][1:] == [0] * (STEPS - 1) and x1["copied"][1:] == [0] * (STEPS - 1), (x0["copied"], x1["copied"])
    for s in range(STEPS):
        for ga, g0, g1 in zip(a0["grads"][s], x0["grads"][s], x1["grads"][s]):
            assert torch.equal(g0, g1) and torch.equal(ga, g0), s
    for pa, p0, p1 in zip(a0["params"], x0["params"], x1["params"]):
        assert torch.equal(p0, p1) and torch.equal(pa, p0)
    assert len(x0["adam"]) == len(a0["adam"]) > 0
    for (ta, ma, va), (t0, m0, v0) in zip(a0["adam"], x0["adam"]):
        assert ta == t0 == STEPS and torch.equal(ma, m0) and torch.equal(va, v0)
    r0, r1 = _spawn(0.0, "xgmi", False, 3)
    _check_against_global_batch(r0, r1, 3, True)


def test_xgmi_reducer_three_ranks():
    """World 3 (three processes on the one GPU): the flag blocks, peer tables and the rank-order sum at a world that is
    neither 2 nor a power of two.  Replicas bit-identical after four updating iterations; with lr = 0 every rank's averaged
    gradients equal single-process gradients on the concatenated batch of all three shards."""
    from view_fusion_amd import train
    STEPS = 4
    rs = _spawn(1e-4, "xgmi", False, STEPS, world=3)
    _skip_without_ipc(*rs)
    assert all(r["error"] is None for r in rs) and rs[0]["info"]["world_size"] == 3
    for a, b, c in zip(rs[0]["params"], rs[1]["params"], rs[2]["params"]):
        assert torch.equal(a, b) and torch.equal(a, c)
    rs = _spawn(0.0, "xgmi", False, 2, world=3)
    dev = torch.device("cuda:0")
    vf = _model(dev)
    tr = train.Trainer(vf, world=1, lr_warmup=1, graph=False)
    tr.it, tr.sched.peak_lr = 0, 0.0
    glob = []
    for s in range(2):
        shards = [_batch(s, r, True) for r in range(3)]
        glob.append({k: torch.cat([sh[k] for sh in shards]) for k in shards[0]})
    ref = _run(tr, vf, glob, dev)
    for s in range(2):
        for g0, g1, g2, gr in zip(rs[0]["grads"][s], rs[1]["grads"][s], rs[2]["grads"][s], ref[s]):
            assert torch.equal(g0, g1) and torch.equal(g0, g2)
            err = float((g0.double() - gr.double()).norm())
            assert err <= 2e-5 * float(gr.double().norm()) + 1e-7 * gr.numel() ** 0.5, (s, err)


def test_xgmi_reducer_turns_a_missing_peer_into_an_error():
    """Rank 1 leaves after two iterations: rank 0's device-side waits give up after VF_XGMI_TIMEOUT_S and a later
    iteration raises VFHipError -- the queue is never left spinning."""
    r0, r1 = _spawn(1e-4, "xgmi", False, 12, env=dict(VF_XGMI_TIMEOUT_S="0.2", leave_after="2"))
    _skip_without_ipc(r0, r1)
    assert r1["error"] is None and len(r1["grads"]) == 2
    assert r0["error"] is not None and "VFHipError" in r0["error"] and "waited" in r0["error"], r0["error"]


@pytest.mark.parametrize("kind,graph,STEPS", [("arena", False, 3), ("arena", True, 6), ("ddp", False, 3)])
def test_two_rank_arena_replicas_stay_in_lock_step(kind, graph, STEPS):
    r0, r1 = _spawn(1e-4, kind, graph, STEPS)
    for a, b in zip(r0["params"], r1["params"]):
        assert torch.equal(a, b)
    moved = max(float((a - b).abs().max()) for a, b in zip(r0["params"], [p.detach().cpu() for p in _model(torch.device("cuda:0")).parameters()]))
    assert moved > 1e-5                           # and the parameters really were updated


def _eval_worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    from view_fusion_amd import drivers
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    vf = _eval_model(dev)
    res = drivers.evaluate(vf, [_eval_batch(rank, dev)], y_t=_eval_noise(rank)[0].to(dev), z_seq=_eval_noise(rank)[1].to(dev))
    out[rank] = float(res["psnr"])
    dist.destroy_process_group()


def _eval_model(dev):
    from view_fusion_amd import UNet, ViewFusion
    from view_fusion_amd.utils import deterministic_fill_
    net = UNet(**TINY)
    deterministic_fill_(net.state_dict())
    vf = ViewFusion(net.to(dev), {"train": dict(schedule="linear", num_timesteps=10, linear_start=1e-4, linear_end=0.09)})
    vf.set_new_noise_schedule(device=dev, phase="train")
    return vf


def _eval_batch(rank, dev):
    g = torch.Generator().manual_seed(500 + rank)
    return dict(target=torch.rand(3, 3, HW, HW, generator=g).to(dev), cond=torch.rand(3, 6, 3, HW, HW, generator=g).to(dev),
                angle=torch.rand(3, 1, generator=g).to(dev), view_count=torch.tensor([1 + rank, 6, 3]))


def _eval_noise(rank):
    g = torch.Generator().manual_seed(600 + rank)
    return torch.randn(3, 3, HW, HW, generator=g), torch.randn(10, 3, 3, HW, HW, generator=g)


def test_two_rank_eval_reduction():
    """SURVEY 8(f4) on the GPU: each rank generates its validation shard, PSNR per image with the HIP kernel, barriers
    and all_reduce(AVG) as Experiment.eval does (experiment.py:314-370) -> the mean over both shards on every rank."""
    from view_fusion_amd import drivers
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_eval_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    dev = torch.device("cuda:0")
    vf = _eval_model(dev)
    per_rank = []
    for r in (0, 1):
        res = drivers.evaluate(vf, [_eval_batch(r, dev)], y_t=_eval_noise(r)[0].to(dev), z_seq=_eval_noise(r)[1].to(dev))
        per_rank.append(float(res["psnr"]))
    want = 0.5 * (per_rank[0] + per_rank[1])
    assert abs(per_rank[0] - per_rank[1]) > 1e-3          # the shards really differ
    assert abs(out[0] - want) < 1e-4 and out[0] == out[1]
