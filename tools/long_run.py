"""300 training iterations on one fixed ragged batch (warm-up LR schedule): the loss must fall and stay finite."""
import os, sys, time, torch
sys.path.insert(0, os.getcwd())
from view_fusion_amd import train
m = train.build_model(device="cuda:0")
tr = train.Trainer(m, lr_warmup=50)
batch = train.synthetic_batch(16, 6, 64, torch.device("cuda:0"), ragged=True)
losses = []
t0 = time.time()
for i in range(300):
    l = tr.step(batch)
    if i % 50 == 0 or i == 299:
        losses.append(round(float(l), 5))
print("losses", losses, "time", round(time.time() - t0, 1), "s", "finite", all(torch.isfinite(p).all().item() for p in m.parameters()))
