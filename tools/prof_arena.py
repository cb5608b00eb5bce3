"""torch-profiler view of one training step with the gradient arena on a world-1 RCCL group: which GPU ops the
reducer adds to the step."""
import os, sys
import torch
import torch.distributed as dist
from torch.profiler import profile, ProfilerActivity
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from view_fusion_amd import train
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
m = train.build_model(device="cuda:0")
tr = train.Trainer(m, world=2)
batch = train.synthetic_batch(16, 6, 64, torch.device("cuda:0"))
for _ in range(4):
    tr.step(batch)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    tr.step(batch)
    torch.cuda.synchronize()
rows = [(e.key, e.count, e.device_time_total if hasattr(e, "device_time_total") else e.cuda_time_total) for e in prof.key_averages()]
rows = [r for r in rows if r[2] > 0 and ("nccl" in r[0].lower() or "rccl" in r[0].lower() or "aten::" in r[0] or "Memcpy" in r[0] or "Memset" in r[0] or "c10d" in r[0])]
for k, n, t in sorted(rows, key=lambda r: -r[2])[:25]:
    print(f"{k[:70]:70s} {n:5d} {t:9.1f} us")
dist.destroy_process_group()
