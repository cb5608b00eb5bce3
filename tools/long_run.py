"""Long-run check of the training harness: ITERS iterations with a FRESH ragged batch geometry every iteration (the
reference draws view_count per sample and iteration, experiment.py:277-279) on fixed images, warm-up LR schedule, a few
sampler steps in between.  The loss must fall and stay finite, replayed and eager iterations must mix freely, and the
device memory in use must stop growing once every stacked-view count has its graph.
usage: long_run.py [ITERS] [graph 0|1] [B] [N]      (round 6: B=16 N=6 = the reference's default ragged training batch)"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from view_fusion_amd import ops, train
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 600
graph = (sys.argv[2] if len(sys.argv) > 2 else "1") == "1"
dev = torch.device("cuda:0")
m = train.build_model(device="cuda:0")
tr = train.Trainer(m, lr_warmup=50, graph=graph)
B = int(sys.argv[3]) if len(sys.argv) > 3 else 8
N = int(sys.argv[4]) if len(sys.argv) > 4 else 3
batch = train.synthetic_batch(B, N, 64, dev, seed=1)
g = torch.Generator().manual_seed(5)
losses, mem, ngr, t0 = [], [], [], time.time()
for i in range(iters):
    batch["view_count"] = torch.randint(1, N + 1, (B,), generator=g)
    l = tr.step(batch)
    if i % 100 == 0 or i == iters - 1:
        torch.cuda.synchronize()
        losses.append(round(float(l), 5))
        mem.append(round(torch.cuda.memory_allocated() / 2 ** 30, 2))
        ngr.append(sum(1 for e in tr._graphs.values() if e.graph is not None))
    if i % 200 == 150:          # validation-style sampling between training iterations
        with torch.no_grad():
            y = torch.randn(2, 3, 64, 64, device=dev)
            m.p_sample(y, batch["y_cond"][:2], [N, N], batch["angle"][:2], torch.tensor([5, 5], device=dev))
torch.cuda.synchronize()
print(f"graph={graph} iters={iters} losses {losses} memory GiB {mem} reserved {torch.cuda.memory_reserved() / 2 ** 30:.1f} "
      f"graphs {len(tr._graphs)} (captured at the memory samples: {ngr}) replayed {tr.graph_steps} time {time.time() - t0:.1f}s "
      f"slab arena GiB {ops.wred_arena_bytes() / 2 ** 30:.2f} "
      f"finite {all(torch.isfinite(p).all().item() for p in m.parameters())}")
