"""Long-run check of the training harness: ITERS iterations with a FRESH ragged batch geometry every iteration (the
reference draws view_count per sample and iteration, experiment.py:277-279) on fixed images, warm-up LR schedule, a few
sampler steps in between.  The loss must fall and stay finite, replayed and eager iterations must mix freely, and the
device memory in use must stop growing once every stacked-view count has its graph.
usage: long_run.py [ITERS] [graph 0|1]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from view_fusion_amd import train
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 600
graph = (sys.argv[2] if len(sys.argv) > 2 else "1") == "1"
dev = torch.device("cuda:0")
m = train.build_model(device="cuda:0")
tr = train.Trainer(m, lr_warmup=50, graph=graph)
B, N = 8, 3
batch = train.synthetic_batch(B, N, 64, dev, seed=1)
g = torch.Generator().manual_seed(5)
losses, mem, t0 = [], [], time.time()
for i in range(iters):
    batch["view_count"] = torch.randint(1, N + 1, (B,), generator=g)
    l = tr.step(batch)
    if i % 100 == 0 or i == iters - 1:
        torch.cuda.synchronize()
        losses.append(round(float(l), 5))
        mem.append(round(torch.cuda.memory_allocated() / 2 ** 30, 2))
    if i % 200 == 150:          # validation-style sampling between training iterations
        with torch.no_grad():
            y = torch.randn(2, 3, 64, 64, device=dev)
            m.p_sample(y, batch["y_cond"][:2], [N, N], batch["angle"][:2], torch.tensor([5, 5], device=dev))
torch.cuda.synchronize()
print(f"graph={graph} iters={iters} losses {losses} memory GiB {mem} reserved {torch.cuda.memory_reserved() / 2 ** 30:.1f} "
      f"graphs {len(tr._graphs)} replayed {tr.graph_steps} time {time.time() - t0:.1f}s "
      f"finite {all(torch.isfinite(p).all().item() for p in m.parameters())}")
