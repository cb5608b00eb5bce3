#!/usr/bin/env python3
"""Forward-only timing of the sampler's conv shapes at a small view count (no grad), per launch incl. split-K reduce.
usage: small_conv.py [S]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from view_fusion_amd import ops  # noqa: E402

S = int(sys.argv[1]) if len(sys.argv) > 1 else 6
dev = torch.device("cuda:0")
shapes = [(64, 64, 64, 3), (128, 128, 32, 3), (192, 192, 16, 3), (384, 192, 16, 3), (320, 320, 8, 3), (640, 320, 8, 3),
          (192, 576, 16, 1), (192, 192, 16, 1), (384, 192, 16, 1), (512, 320, 8, 1)]
row = []
with torch.no_grad():
  for small in (False, True):
    ops.st.SMALL_CONV, ops.st.SMALL_CONV_MAX_WGS, ops.st.SMALL_CONV_MAX_CIN3, ops.st.SMALL_CONV_MAX_S3 = small, (1 << 30, 1 << 30), 1 << 30, 1 << 30
    row.append("| small" if small else "| split-K")
    for Cin, Cout, H, KS in shapes:
          layer = torch.nn.Conv2d(Cin, Cout, KS, padding=KS // 2).to(dev)
          x = torch.rand(S, Cin, H, H, device=dev)
          for _ in range(3):
              ops.conv2d(x, layer)
          e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
          g = torch.cuda.CUDAGraph()
          with torch.cuda.graph(g):
              for _ in range(20):
                  ops.conv2d(x, layer)
          g.replay()
          e0.record()
          g.replay()
          e1.record()
          torch.cuda.synchronize()
          us = e0.elapsed_time(e1) / 20 * 1e3
          row.append(f"{Cin}>{Cout}@{H}k{KS}:{us:.1f}")
print(f"S={S} KTARGET={os.environ.get('VF_CONV_KTARGET', '768')} WINO_MIN_FILL={os.environ.get('VF_WINO_MIN_FILL', '-')} | " + " ".join(row))
