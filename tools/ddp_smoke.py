#!/usr/bin/env python3
"""Single-GPU smoke test of the DDP/RCCL path (world_size 1): DDP wrapper + reducer hooks on top of
the custom autograd Functions, gradient buckets as views, fused Adam.  Prints ms/step."""
import os
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from view_fusion_amd import train  # noqa: E402

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
model = train.build_model(device="cuda:0")
tr = train.Trainer(model, world=2, local_rank=0)        # world=2 only selects the DDP branch
batch = train.synthetic_batch(16, 6, 64, torch.device("cuda:0"))
for _ in range(2):
    tr.step(batch)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5):
    loss = tr.step(batch)
torch.cuda.synchronize()
print(f"DDP(world=1) {1e3 * (time.perf_counter() - t0) / 5:.2f} ms/step loss {loss.item():.5f}")
dist.destroy_process_group()
