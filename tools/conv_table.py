#!/usr/bin/env python3
"""Per-shape timing of the conv contraction kernels inside a real training step (GPU box).
Prints (kind, Cin, Cout, H, KS, mode): launches/step, avg us, TFLOP/s."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from view_fusion_amd import ops, train  # noqa: E402

dev = torch.device("cuda:0")
model = train.build_model(device="cuda:0")
tr = train.Trainer(model, graph=False)
import sys as _s
B = int(_s.argv[1]) if len(_s.argv) > 1 else 16
batch = train.synthetic_batch(B, 6, 64, dev)
for _ in range(2):
    tr.step(batch)
ops.st.KERNEL_LOG = []
steps = 2
for _ in range(steps):
    tr.step(batch)
torch.cuda.synchronize()
log, ops.st.KERNEL_LOG = ops.st.KERNEL_LOG, None
agg = {}
for kind, flops, e0, e1, tag, name, _nb in log:
    if tag is None:
        continue
    a = agg.setdefault((kind + ('/W44' if name.startswith('vf_wino44_conv') else '/W' if name.startswith('vf_wino') else '/D'),) + tag, [0.0, 0.0, 0])
    a[0] += flops; a[1] += e0.elapsed_time(e1) * 1e-3; a[2] += 1
tot = {}
for k, (f, t, n) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{k[0]:13s} Cin={k[1]:4d} Cout={k[2]:4d} H={k[3]:3d} KS={k[4]} m={k[5]}  n/step={n // steps:3d} "
          f"avg={t / n * 1e6:8.1f}us  tot/step={t / steps * 1e3:7.2f}ms  {f / t / 1e12:6.1f} TF")
    x = tot.setdefault(k[0], [0.0, 0.0]); x[0] += f; x[1] += t
for k, (f, t) in tot.items():
    print(f"TOTAL {k:11s} {t / steps * 1e3:7.2f} ms/step  {f / t / 1e12:6.1f} TF")
