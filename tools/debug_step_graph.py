"""Diagnostic: capture parts of the training iteration into a HIP graph and replay them.  STAGE = fwd | bwd | adam."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from view_fusion_amd import train, ops
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from conftest import TINY
stage = os.environ.get("STAGE", "fwd")
dev = "cuda:0"
hp = TINY if os.environ.get("NET", "tiny") == "tiny" else None
hw = 16 if hp else 64
model = train.build_model(unet_params=hp, device=dev)
tr = train.Trainer(model, graph=False)
batch = train.synthetic_batch(2, 3, hw, device=dev, seed=0)
for _ in range(2):
    tr.step(batch)
torch.cuda.synchronize()
vc = torch.tensor(batch["view_count"].tolist(), dtype=torch.int64, device=dev)
ops.view_offsets(vc, torch.device(dev))
inputs = {k: batch[k].clone() for k in ("y_0", "y_cond", "angle")}
if os.environ.get("DRAWS") == "1":
    inputs.update(t=torch.tensor([5, 1500], device=dev), u=torch.rand(2, 1, device=dev), noise=torch.randn(2, 3, hw, hw, device=dev))
if os.environ.get("OTHER") == "1":
    import copy
    m2 = train.build_model(unet_params=hp, device=dev)
    t2 = train.Trainer(m2, graph=False)
    for _ in range(2):
        t2.step(batch)
tr.opt.zero_grad()
adam = tr.opt.graph_begin()
scal = torch.zeros(3, device=dev)
g = torch.cuda.CUDAGraph()
ops.st.COLSUM_DEFER = os.environ.get("DEFER", "1") == "1"
ops.begin_capture(torch.device(dev), 128)
with torch.cuda.graph(g, capture_error_mode="relaxed"):
    if stage == "fwd":
        with torch.no_grad():
            model.train()
            loss = model(view_count=vc, **inputs)
    else:
        loss = model(view_count=vc, **inputs)
        loss.backward()
        if stage == "adam":
            tr.opt.step_captured(adam, scal)
fix = ops.end_capture()
print("captured", stage, "table rows", fix[1], flush=True)
if stage == "adam":
    tr.opt.graph_end(adam)
for i in range(4):
    if stage == "adam":
        tr.opt.graph_tick(adam, scal)
    g.replay()
    torch.cuda.synchronize()
    print("replay", i, float(loss), flush=True)
