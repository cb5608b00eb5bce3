#!/usr/bin/env python3
"""Headline benchmark of the ViewFusion hot path on MI355X.

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one reference training iteration (experiment.py:286-293): zero_grad ->
ViewFusion.forward (q_sample, ragged stack, UNet, softmax compose, MSE) -> backward -> Adam.step,
on B=16 samples x N=6 views per GPU (S=96 stacked views), small UNet 64x64, fp32, synthetic
NMR-shaped tensors resident in HBM.  `value` = view denoise-steps/s over all ranks
(= S_per_rank * world * iterations/s); weak scaling (per-GPU work fixed).

Rank 0 prints ONE JSON line.  At --gpus 1 it also carries
  roofline     : the dominant kernel (conv_mfma_kernel, fwd + dgrad launches) timed with HIP events
                 on its launch stream over extra instrumented steps; algorithmic FLOPs / time
                 against the 157.3 TF fp32 matrix peak
  sampler      : sampled views/s of the T=1000 reverse loop (HIP-graph replay at small S)
  cpu_baseline : the CPU oracle (fresh PyTorch-CPU restatement of the reference, kind "port")
                 timed on a bounded sample (B=1, N=6 -> 6 views/iteration) on the host cores.
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from view_fusion_amd import ops, train  # noqa: E402

PEAK_FP32_MATRIX_TFLOPS = 157.3      # MI355X_MICROARCH.md chip table
PEAK_HBM_GBS = 8000.0


def cpu_baseline(hw, n_views, iters=2):
    """Oracle (oracle/, CPU fp32) training iteration incl. Adam on B=1 x N views."""
    from oracle import unet_ref, view_fusion_ref as vfr
    from view_fusion_amd import UNet
    hp = train.SMALL_UNET
    torch.manual_seed(0)
    sd = {k: v.clone().requires_grad_(True) for k, v in UNet(**hp).state_dict().items()}
    opt = torch.optim.Adam(list(sd.values()), lr=1e-4)
    sched = vfr.schedule_buffers(vfr.beta_schedule(**train.BETA_SCHEDULE["train"]))
    b = train.synthetic_batch(1, n_views, hw, "cpu", seed=0)
    fn = lambda x, a, l: unet_ref.unet_forward(sd, hp, x, a, l)
    g = torch.Generator().manual_seed(1)

    def one():
        t = torch.randint(1, 2000, (1,), generator=g)
        u, noise = torch.rand(1, 1, generator=g), torch.randn(1, 3, hw, hw, generator=g)
        opt.zero_grad()
        loss = vfr.train_loss(fn, sched, b["y_cond"], b["view_count"], b["angle"], b["y_0"], t, u, noise, True)
        loss.backward()
        opt.step()

    one()                                     # warm-up (oneDNN primitive creation)
    t0 = time.perf_counter()
    for _ in range(iters):
        one()
    dt = (time.perf_counter() - t0) / iters
    return dict(value=n_views / dt, unit="view denoise-steps/s", cores=torch.get_num_threads(), kind="port",
                sample=f"oracle train iteration (fwd+bwd+Adam), B=1 N={n_views} 64x64, {iters} timed iterations "
                       f"after 1 warm-up, {dt:.2f} s/iteration")


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes of this same bench
    command (profiles/r01_bench_pmc_traffic_kib_per_launch.json: FETCH_SIZE and WRITE_SIZE collected
    in separate --pmc runs; FETCH_SIZE doubled per MI355X_MICROARCH.md's gfx950 correction).  PMC
    counters cannot be read from inside the process, so this is the recorded, not a live, figure."""
    path = os.path.join(ROOT, "profiles", "r01_bench_pmc_traffic_kib_per_launch.json")
    try:
        tab = json.load(open(path))
        tot = n = 0.0
        for k in kernel:                      # launch-weighted mean over the kernels of the family
            d = tab[k]
            tot += (2 * d["FETCH_SIZE"] + d["WRITE_SIZE"]) * 1024.0 * d["launches"]
            n += d["launches"]
        return tot / n
    except (OSError, KeyError, ZeroDivisionError):
        return None


def roofline(trainer, batch, steps=2):
    """Per-launch HIP-event timing of the conv contraction kernels over `steps` extra iterations."""
    ops.KERNEL_LOG = []
    for _ in range(steps):
        trainer.step(batch)
    torch.cuda.synchronize()
    log, ops.KERNEL_LOG = ops.KERNEL_LOG, None
    agg = {}
    for kind, flops, e0, e1, _tag, name in log:
        a = agg.setdefault(kind, [0.0, 0.0, 0, 0.0])
        a[0] += flops
        a[1] += e0.elapsed_time(e1) * 1e-3
        a[2] += 1
        a[3] += flops / 2.25 if name.startswith("vf_wino") else flops     # multiplies the matrix cores execute
    f = agg["conv_fwd"][0] + agg["conv_dgrad"][0]
    s = agg["conv_fwd"][1] + agg["conv_dgrad"][1]
    n = agg["conv_fwd"][2] + agg["conv_dgrad"][2]
    x = agg["conv_fwd"][3] + agg["conv_dgrad"][3]
    out = dict(bound="mfma", kernel="conv forward + dgrad launches: wino_conv_kernel<LOGW,MODE> (fused Winograd F(2x2,3x3): "
                                   "every stride-1 3x3 layer) + conv_mfma_kernel<KS,LOGW,MODE,NPT> (1x1, stride 2)",
               achieved=f / s / 1e12, peak=PEAK_FP32_MATRIX_TFLOPS, unit="TFLOP/s",
               frac=f / s / 1e12 / PEAK_FP32_MATRIX_TFLOPS,
               note="achieved = ALGORITHMIC direct-convolution FLOPs (2*S*Cout*Cin*KS^2*H*W) / time; the Winograd "
                    "launches execute 2.25x fewer multiplies, so the fraction of the fp32 MFMA peak the matrix "
                    "cores actually sustain is mfma_executed_frac",
               mfma_executed_tflops=x / s / 1e12, mfma_executed_frac=x / s / 1e12 / PEAK_FP32_MATRIX_TFLOPS,
               traffic=pmc_traffic(("wino_conv_kernel", "conv_mfma_kernel")),
               traffic_unit="HBM bytes per launch (launch-weighted mean of the kernel family)",
               traffic_source="profiles/r01_bench_pmc_traffic_kib_per_launch.json (rocprofv3 --pmc, FETCH_SIZE and "
                              "WRITE_SIZE in separate passes, gfx950 corrections of MI355X_MICROARCH.md applied)",
               launches_per_step=n // steps,
               avg_launch_us=s / n * 1e6, algorithmic_gflop_per_launch=f / n / 1e9)
    out["other_kernels"] = {k: dict(achieved_tflops=v[0] / v[1] / 1e12, mfma_executed_tflops=v[3] / v[1] / 1e12,
                                    launches_per_step=v[2] // steps, avg_launch_us=v[1] / v[2] * 1e6)
                            for k, v in agg.items()}
    return out


def sampler_leg():
    """Second half of BASELINE's metric: sampled views/s of the reverse-diffusion loop (config C5:
    T=1000 test schedule, N conditioning views).  A bounded number of reverse steps is timed and
    extrapolated linearly to T=1000 (the loop is strictly sequential with constant step cost)."""
    from view_fusion_amd import sampling_bench
    model = train.build_model(device="cuda:0", phase="test")
    out = {"unit": "completed target views/s at T=1000", "schedule": "linear T=1000 1e-4..0.09"}
    for B, N, steps in ((1, 1, 100), (1, 6, 100), (1, 12, 100), (16, 6, 20)):
        r = sampling_bench.time_sampler(B, N, steps=steps, model=model, use_graph=None)
        out[f"B{B}_N{N}"] = dict(sampled_views_per_sec=r["sampled_views_per_sec"], ms_per_step=r["ms_per_step"],
                                 view_unet_evals_per_sec=r["view_unet_evals_per_sec"], steps_timed=steps)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=16, help="samples per GPU")
    ap.add_argument("--views", type=int, default=6)
    ap.add_argument("--ragged", action="store_true", help="view_count = randint(1, N+1) per sample (experiment.py:277-279)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-sampler", action="store_true")
    args = ap.parse_args()

    rank, local_rank, world = train.init_distributed()
    assert world == args.gpus, f"launched with WORLD_SIZE={world} but --gpus {args.gpus}"
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    model = train.build_model(device=str(dev), seed=0)
    trainer = train.Trainer(model, world=world, local_rank=local_rank)
    batch = train.synthetic_batch(args.batch, args.views, 64, dev, seed=rank, ragged=args.ragged)
    S = int(batch["view_count"].sum())

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        trainer.step(batch)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = trainer.step(batch)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    loss_val = float(loss.item())

    if rank == 0:
        ms = dt / args.steps * 1e3
        res = {
            "metric": "denoise_steps_per_sec", "value": S * world * args.steps / dt,
            "unit": "view denoise-steps/s (UNet fwd+bwd per stacked view, incl. compose/MSE/Adam)",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "small UNet 64x64 (33.9M params), B=%d/GPU N=%d (S=%d views/GPU), training "
                                   "iteration fwd+bwd+Adam, linear T=2000 schedule%s" % (args.batch, args.views, S,
                                                                                     ", ragged view_count" if args.ragged else ""),
                       "global_batch": args.batch * world, "views": args.views,
                       "parallelism": "dp%d" % world},
            "iters_per_sec": args.steps / dt, "loss": loss_val,
            "achieved_tflops_total": 62.98e9 * S * world * args.steps / dt / 1e12,
        }
        if world == 1 and not args.no_roofline:
            res["roofline"] = roofline(trainer, batch)
        if world == 1 and not args.no_sampler:
            res["sampler"] = sampler_leg()
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(64, args.views)
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
