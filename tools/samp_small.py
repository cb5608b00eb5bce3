#!/usr/bin/env python3
"""Sampler step time (graph replay) with the one-launch small-S conv kernel on / off and its grid-size threshold."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from view_fusion_amd import ops, sampling_bench, train  # noqa: E402
model = train.build_model(device="cuda:0", phase="test")
for B, N in ((1, 1), (1, 2), (1, 6), (1, 12), (2, 12)):
    row = []
    for small, mx in ((False, 0), (True, (512, 512)), (True, (1024, 512)), (True, (2048, 512)), (True, (4096, 512))):
        ops.SMALL_CONV, ops.SMALL_CONV_MAX_WGS = small, mx
        r = sampling_bench.time_sampler(B, N, steps=200, use_graph=True, model=model)
        row.append(f"{'off' if not small else mx}: {r['ms_per_step']:.3f}")
    print(f"B={B} N={N}  " + "  ".join(row), flush=True)
