"""Host-side (Python) cost of one training step: cProfile, GPU work not waited for inside the profiled region."""
import cProfile, os, pstats, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from view_fusion_amd import train
dev = "cuda:0"
ARENA = os.environ.get("VF_PROF_ARENA") == "1"      # the N > 1 launch path on one GPU: gradient arena over a world-1 RCCL group
if ARENA:
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29541")
    torch.cuda.set_device(0)
    train.init_rccl_group(0, rank=0, world_size=1)
model = train.build_model(device=dev, phase="train")
GRAPH = os.environ.get("VF_STEP_GRAPH", "1") == "1"
tr = train.Trainer(model, graph=GRAPH, world=2 if ARENA else 1)     # (world=2 only selects the reducer)
print("whole-step HIP graph:", "on" if GRAPH else "off", "| gradient arena over RCCL (world 1):", "on" if ARENA else "off")
batch = train.synthetic_batch(16, 6, 64, device=dev, seed=0)
for _ in range(6):           # (arena: layout iteration + two sightings + the capture)
    tr.step(batch)
torch.cuda.synchronize()
print("iterations replayed from a graph so far:", tr.graph_steps)
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
