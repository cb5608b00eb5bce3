#!/usr/bin/env python3
"""BASELINE C2 as it is worded -- a 1000-iteration training run (B=16, N=6, small UNet 64x64) -- timed in blocks of 100
iterations, with the shader clock the driver reports sampled at every block boundary.  Answers whether the 20-step
burst that `bench.py` times by default survives 30 s of full-chip load (VERDICT r05 "what's weak" 7).

    python tools/sustained_run.py [--steps 1000] [--warmup 20] [--block 100] > profiles/r06_bench_n1_1000.json
"""
import argparse
import glob
import json
import os
import re
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from view_fusion_amd import train  # noqa: E402


def sclk_mhz():
    """Current shader clock from sysfs (the starred line of pp_dpm_sclk); None when the node is not readable."""
    for p in sorted(glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk")):
        try:
            txt = open(p).read()
        except OSError:
            continue
        m = re.search(r"(\d+)Mhz \*", txt)
        if m:
            return int(m.group(1))
    return None


def power_w():
    for p in sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/power1_average")):
        try:
            return int(open(p).read()) / 1e6
        except (OSError, ValueError):
            continue
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--block", type=int, default=100)
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--views", type=int, default=6)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    model = train.build_model(device=str(dev), seed=0)
    trainer = train.Trainer(model, graph=True)
    batch = train.synthetic_batch(args.batch, args.views, 64, dev, seed=0)
    S = int(batch["view_count"].sum())
    for _ in range(max(args.warmup, train.Trainer.GRAPH_AFTER + 1)):
        trainer.step(batch)
    torch.cuda.synchronize()
    blocks = []
    t_start = time.perf_counter()
    done = 0
    while done < args.steps:
        n = min(args.block, args.steps - done)
        c0, p0 = sclk_mhz(), power_w()
        t0 = time.perf_counter()
        for _ in range(n):
            loss = trainer.step(batch)
        # the clock / power are read while the queue is still full (before the sync)
        c1, p1 = sclk_mhz(), power_w()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        done += n
        blocks.append({"steps": n, "ms_per_step": dt / n * 1e3, "sclk_mhz_start": c0, "sclk_mhz_end": c1,
                       "power_w_start": p0, "power_w_end": p1, "loss": float(loss.item())})
    total = time.perf_counter() - t_start
    ms = [b["ms_per_step"] for b in blocks]
    print(json.dumps({
        "workload": "small UNet 64x64 (33.9M params), B=%d N=%d (S=%d), training iteration fwd+bwd+Adam, HIP-graph replay"
                    % (args.batch, args.views, S),
        "steps": args.steps, "warmup": args.warmup, "seconds": total, "ms_per_step": total / args.steps * 1e3,
        "view_steps_per_sec": S * args.steps / total, "first_block_ms": ms[0], "last_block_ms": ms[-1],
        "min_block_ms": min(ms), "max_block_ms": max(ms), "device": torch.cuda.get_device_name(dev), "blocks": blocks}))


if __name__ == "__main__":
    main()
