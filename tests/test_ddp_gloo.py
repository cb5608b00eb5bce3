"""World-size-2 CPU (gloo) test of the data-parallel harness: the Trainer's reducer (the gradient
arena of reducer.py by default, torch DDP with VF_REDUCER=ddp), per-rank shards and the gradient
all-reduce (mean) give the same updates as one process on the concatenated batch, over several
iterations (the arena lays its slots out on the first one and overlaps segments from the second).  The HIP model cannot run on CPU (no fallback by design), so a small
stand-in module with the ViewFusion.forward signature takes its place; what is under test is
the harness (view_fusion_amd.train), which is device-agnostic.  RCCL itself only runs on the
GPU box (bench.py --gpus N)."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class StandIn(torch.nn.Module):
    """Same call signature as ViewFusion.forward; loss = per-sample MSE of a tiny conv net."""

    def __init__(self):
        super().__init__()
        torch.manual_seed(0)
        self.net = torch.nn.Sequential(torch.nn.Conv2d(3, 8, 3, padding=1), torch.nn.Tanh(),
                                       torch.nn.Conv2d(8, 3, 3, padding=1))

    def forward(self, y_cond, view_count, angle, y_0=None, noise=None, generate=False):
        pred = self.net(y_cond.mean(dim=1)) * angle.reshape(-1, 1, 1, 1).cos()
        return torch.nn.functional.mse_loss(pred, y_0)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


STEPS = 3


def _worker(rank, world, port, out, reducer_kind):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank),
                      WORLD_SIZE=str(world), VF_REDUCER=reducer_kind)
    torch.set_num_threads(1)
    from view_fusion_amd import train
    r, lr, w = train.init_distributed()
    assert (r, w) == (rank, world)
    tr = train.Trainer(StandIn(), world=w, local_rank=lr, lr_warmup=1)
    tr.it = 0                                   # lr(1) = peak
    assert (tr.arena is not None) == (reducer_kind == "arena")
    for it in range(STEPS):
        tr.step(train.synthetic_batch(4, 3, 8, "cpu", seed=10 * it + rank))
    if tr.arena is not None:                    # every gradient lives in its arena slot, segments cover the arena
        a = tr.arena
        assert all(p.grad.data_ptr() == a.base + 4 * a.off[i] for i, p in enumerate(a.params))
        assert a.seg_range[0][0] == 0 and a.seg_range[-1][1] == a.flat.numel()
        assert all(x[1] == y[0] for x, y in zip(a.seg_range, a.seg_range[1:]))
    out[rank] = [p.detach().clone() for p in tr.module.parameters()]
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize("reducer_kind", ["arena", "ddp"])
def test_two_rank_gloo_matches_single_process_on_the_global_batch(reducer_kind):
    sys.path.insert(0, ROOT)
    from view_fusion_amd import train
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), out, reducer_kind), nprocs=world, join=True)
    # replicas stay in lock-step
    for a, b in zip(out[0], out[1]):
        assert torch.equal(a, b)
    # single process on the concatenation of the two shards
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        os.environ.pop(k, None)
    tr = train.Trainer(StandIn(), world=1, lr_warmup=1)
    tr.it = 0
    for it in range(STEPS):
        shards = [train.synthetic_batch(4, 3, 8, "cpu", seed=10 * it + r) for r in range(world)]
        tr.step({k: torch.cat([s[k] for s in shards]) for k in shards[0]})
    for a, b in zip(out[0], tr.module.parameters()):
        assert torch.allclose(a, b.detach(), rtol=1e-5, atol=1e-7)


def _reduce_worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank),
                      WORLD_SIZE=str(world))
    import torch.distributed as dist
    from view_fusion_amd import drivers, train
    r, _, w = train.init_distributed()
    d = {"psnr": torch.tensor(20.0 + 4 * rank), "ssim": torch.tensor(0.5 + 0.25 * rank)}
    avg = drivers.reduce_dict(d)
    tot = drivers.reduce_dict(d, average=False)
    # the eval flow of Experiment.eval with its three barriers (experiment.py:347-366), metrics precomputed
    class Fixed(torch.nn.Module):
        def forward(self, y_cond, view_count, angle, generate=False):
            return (None, None, None, None, y_cond[:, 0] * (rank + 1))
    tgt = torch.ones(2, 3, 4, 4)
    ev = drivers.evaluate(Fixed(), [dict(target=tgt, cond=torch.full((2, 6, 3, 4, 4), 0.5), angle=torch.zeros(2, 1))],
                          extra_metrics={"mse": lambda a, b: ((a - b) ** 2).mean(dim=(1, 2, 3))})
    out[rank] = dict(avg={k: float(v) for k, v in avg.items()}, tot={k: float(v) for k, v in tot.items()},
                     untouched=float(d["psnr"]), mse=float(ev["mse"]))
    dist.destroy_process_group()


def test_eval_reduction_world_two():
    """reduce_dict (all_reduce AVG over the sorted keys, utils/dist.py:69-91) and the barrier-bracketed
    eval reduction at world size 2.  The PSNR kernel needs a GPU, so the CPU run injects a host metric."""
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_reduce_worker_patched, args=(2, _free_port(), out), nprocs=2, join=True)
    for r in (0, 1):
        assert out[r]["avg"] == {"psnr": 22.0, "ssim": 0.625} and out[r]["tot"] == {"psnr": 44.0, "ssim": 1.25}
        assert out[r]["untouched"] == 20.0 + 4 * r              # the input dict is not modified
        # rank 0 generates 0.5 (mse 0.25), rank 1 generates 1.0 (mse 0): mean over ranks
        assert abs(out[r]["mse"] - 0.125) < 1e-7


def _reduce_worker_patched(rank, world, port, out):
    sys.path.insert(0, ROOT)
    from view_fusion_amd import drivers
    drivers.compute_psnr = lambda a, b: ((a - b) ** 2).mean(dim=(1, 2, 3))   # host stand-in for the HIP PSNR kernel
    _reduce_worker(rank, world, port, out)
