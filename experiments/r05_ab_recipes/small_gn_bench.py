#!/usr/bin/env python3
"""What the launch-free GroupNorm costs the sampler's one-launch convs (S = 1): us per launch of vf_conv_small_gn
plain / + output statistics (integer atomics) / + GroupNorm applied on load / both, HIP events over 200 back-to-back
launches, next to the GroupNorm launch it replaces."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from view_fusion_amd import ops

dev = torch.device("cuda:0")
S = int(sys.argv[1]) if len(sys.argv) > 1 else 1


def t_us(fn, n=100):
    """GPU time per launch: n launches captured into one HIP graph (a dependent chain, as in the sampler's step) and
    replayed -- eager launches from Python are host-bound at ~10 us each and would hide everything below that."""
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        fn()
    torch.cuda.current_stream().wait_stream(side)
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


with torch.no_grad():
    for C, H, KS in ((64, 64, 3), (128, 32, 3), (192, 16, 3), (192, 16, 1), (320, 8, 1)):
        conv = torch.nn.Conv2d(C, C, KS, padding=KS // 2).to(dev)
        gn = torch.nn.GroupNorm(32, C).to(dev)
        x = torch.randn(S, C, H, H, device=dev)
        ops.STATS = ops.StatsArena(dev, S, channels=1 << 16)
        _, st = ops.conv2d(x, conv, want_stats=True)
        lazy = ops.LazyGN(x, st, gn, 32, True)

        def stats_run():
            ops.STATS.used = 0
            return ops.conv2d(x, conv, want_stats=True)

        def both_run():
            ops.STATS.used = 0
            return ops.conv2d(lazy, conv, want_stats=True)

        r = dict(plain=t_us(lambda: ops.conv2d(x, conv)), stats=t_us(stats_run), lazy_in=t_us(lambda: ops.conv2d(lazy, conv)),
                 both=t_us(both_run), gn_launch=t_us(lambda: ops.group_norm(x, gn.weight, gn.bias, 32, True)))
        ops.STATS = None
        print(f"S={S} C={C:4d} {H:2d}x{H:<2d} {KS}x{KS}: " + "  ".join(f"{k} {v:6.2f} us" for k, v in r.items()), flush=True)
