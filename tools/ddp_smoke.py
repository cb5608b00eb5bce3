#!/usr/bin/env python3
"""Single-GPU check of the data-parallel exchange step over RCCL (world_size 1): the gradient arena
(reducer.py: gradients written into the communication buffer by the backward kernels, segment-wise
async all-reduce) against torch DDP and against plain single-process training.  Prints ms/step, the
number of gradients per step that still had to be copied into the arena, and a parameter checksum
(identical for all three: the same kernels run, only the destination of the gradients differs).

    python tools/ddp_smoke.py            # runs the three modes as child processes
"""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child(kind):
    import torch
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    from view_fusion_amd import train
    torch.cuda.set_device(0)
    eager = kind.endswith("-eager")
    split = kind.endswith("-split")            # the default launch mode of a real multi-rank run (round 5)
    if split:
        os.environ["VF_CAPTURE_COLLECTIVES"] = "0"
    kind = kind.replace("-eager", "").replace("-split", "")
    if kind != "none":
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ["VF_REDUCER"] = kind
        train.init_rccl_group(0, rank=0, world_size=1)
    model = train.build_model(device="cuda:0")
    tr = train.Trainer(model, world=2 if kind != "none" else 1, local_rank=0,      # world=2 only selects the reducer
                       graph=False if eager else None)
    batch = train.synthetic_batch(16, 6, 64, torch.device("cuda:0"))
    for _ in range(6):
        tr.step(batch)
    if tr.arena is not None:
        tr.arena.copied = 0
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 10
    for _ in range(n):
        loss = tr.step(batch)
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / n
    chk = sum(float(p.double().abs().sum()) for p in model.parameters())
    extra = ""
    if tr.arena is not None:
        a = tr.arena
        extra = f" copied/step {a.copied / n:.0f} of {len(a.params)} segments {[(hi - lo) * 4 >> 20 for lo, hi in a.seg_range]} MiB"
    kind = kind + ("-eager" if eager else "") + ("-split" if split else "")
    print(f"reducer={kind:11s} mode {tr.mode:8s} graph replays {tr.graph_steps:2d}  {ms:.2f} ms/step loss {loss.item():.6f} param-checksum {chk:.9e}{extra}", flush=True)
    if kind != "none":
        dist.destroy_process_group()


if __name__ == "__main__":
    if len(sys.argv) > 1:
        child(sys.argv[1])
    else:
        for kind in ("none-eager", "none", "ddp", "arena-eager", "arena-split", "arena"):
            subprocess.run([sys.executable, os.path.abspath(__file__), kind], check=False)
