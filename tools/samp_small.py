#!/usr/bin/env python3
"""Sampler step time (graph replay) against the policy knobs of the one-launch small-S conv kernels."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from view_fusion_amd import ops, sampling_bench, train  # noqa: E402
model = train.build_model(device="cuda:0", phase="test")
for B, N in ((1, 1), (1, 2), (1, 6)):
    row = []
    for small, cin3, s3 in ((False, 0, 0), (True, 256, 1), (True, 320, 1), (True, 384, 1), (True, 320, 2), (True, 1 << 30, 1)):
        ops.st.SMALL_CONV, ops.st.SMALL_CONV_MAX_CIN3, ops.st.SMALL_CONV_MAX_S3 = small, cin3, s3
        r = sampling_bench.time_sampler(B, N, steps=200, use_graph=True, model=model)
        row.append(f"{'off' if not small else (cin3, s3)}: {r['ms_per_step']:.3f}")
    print(f"B={B} N={N}  " + "  ".join(row), flush=True)
