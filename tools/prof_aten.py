"""Which aten ops (outside our kernels) run per training step: torch.profiler summary."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from view_fusion_amd import train
from torch.profiler import profile, ProfilerActivity
dev = "cuda:0"
model = train.build_model(device=dev, phase="train")
tr = train.Trainer(model)
batch = train.synthetic_batch(16, 6, 64, device=dev, seed=0)
for _ in range(3):
    tr.step(batch)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    tr.step(batch)
    torch.cuda.synchronize()
print(prof.key_averages(group_by_input_shape=False).table(sort_by="cuda_time_total", row_limit=40, max_name_column_width=60))
rows = [e for e in prof.key_averages(group_by_input_shape=True) if e.key.startswith("aten::")]
rows = [e for e in rows if e.device_time_total > 0]
rows.sort(key=lambda e: -e.device_time_total)
for e in rows[:60]:
    print(e.key, e.count, e.input_shapes[:3], round(e.device_time_total, 1))
