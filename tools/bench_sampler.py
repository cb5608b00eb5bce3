#!/usr/bin/env python3
"""Sampler throughput table (GPU box): B x N grid, HIP graph vs eager launches."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from view_fusion_amd import sampling_bench, train  # noqa: E402

model = train.build_model(device="cuda:0", phase="test")
for B, N, steps in ((1, 1, 200), (1, 6, 200), (1, 12, 200), (16, 6, 40)):
    for graph in (True, False):
        r = sampling_bench.time_sampler(B, N, steps=steps, use_graph=graph, model=model)
        print(json.dumps(r), flush=True)
