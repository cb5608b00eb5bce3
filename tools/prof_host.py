"""Host-side (Python) cost of one training step: cProfile, GPU work not waited for inside the profiled region."""
import cProfile, os, pstats, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from view_fusion_amd import train
dev = "cuda:0"
model = train.build_model(device=dev, phase="train")
GRAPH = os.environ.get("VF_STEP_GRAPH") == "1"
tr = train.Trainer(model, graph=GRAPH)
print("whole-step HIP graph:", "on" if GRAPH else "off")
batch = train.synthetic_batch(16, 6, 64, device=dev, seed=0)
for _ in range(3):
    tr.step(batch)
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for _ in range(5):
    tr.step(batch)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host enqueue time per step {1e3 * (t1 - t0) / 5:.2f} ms; wall incl. drain {1e3 * (t2 - t0) / 5:.2f} ms")
single = []
for _ in range(5):                      # one step at a time from an EMPTY queue: no back-pressure from the launch queue
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tr.step(batch)
    single.append(time.perf_counter() - t0)
torch.cuda.synchronize()
print("host enqueue time of single steps from an empty queue, ms:", " ".join(f"{1e3 * t:.2f}" for t in single))
pr = cProfile.Profile()
pr.enable()
for _ in range(3):
    tr.step(batch)
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(28)
