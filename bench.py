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
  roofline     : the dominant kernel (wino_conv_kernel, forward + dgrad launches) timed with HIP events
                 on its launch stream over extra instrumented steps; `frac` = the multiplies the matrix
                 cores EXECUTE / time / the 157.3 TF fp32 matrix peak (<= 1); plus a per-kernel-family
                 table (both roofs) and the step's distance from its Winograd-adjusted compute floor
  sampler      : sampled views/s of the T=1000 reverse loop (HIP-graph replay at small S)
  cpu_baseline : the CPU oracle (fresh PyTorch-CPU restatement of the reference, kind "port")
                 timed on bounded samples (all cores and one thread; training and sampler).
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


# Algorithmic constants of one stacked view through the small UNet (measured on the reference):
GFLOP_PER_VIEW_TRAIN = 62.98           # 3 x 20.994 forward (conv3x3 19.19, conv1x1 1.45, attention 0.36)
GFLOP_3X3_S1_PER_VIEW_FWD = 19.1905 - 0.19     # stride-1 3x3 layers: what the Winograd kernels run
HBM_MB_PER_VIEW_TRAIN = 3 * 141.7      # ideal-fusion traffic


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(hw, n_views, threads_all=None):
    """The CPU oracle (oracle/: PyTorch-CPU fp32 restatement of the reference, pinned to it by tests/golden; kind
    "port") on the host cores, on BOUNDED samples of BASELINE.json's configurations, ~1 minute in all:
      calibration      : B=2 x N views, one iteration per thread count (PyTorch-CPU eager does not scale to all cores)
      C2 = `value`     : the bench workload itself, B=16 x N=6 (S=96), 1 warm-up + 2 timed iterations (fwd+bwd+Adam)
      C4               : B=8 x N=6 (S=48), 1 warm-up + 2 timed iterations
      C1 (in full)     : small UNet, B=2 N=2, one training iteration + the 10-step reverse chain
      one thread       : B=1 x N views, 1 warm-up + 1 timed iteration
      sampler (C5)     : B=1 x N views, 10 reverse steps of the T=1000 schedule (extrapolated linearly to T=1000)
    """
    from oracle import unet_ref, view_fusion_ref as vfr
    from view_fusion_amd import UNet
    hp = train.SMALL_UNET
    threads_all = threads_all or torch.get_num_threads()

    def train_leg(B, iters, threads, N=n_views, warm=True):
        torch.set_num_threads(threads)
        torch.manual_seed(0)
        sd = {k: v.clone().requires_grad_(True) for k, v in UNet(**hp).state_dict().items()}
        opt = torch.optim.Adam(list(sd.values()), lr=1e-4)
        sched = vfr.schedule_buffers(vfr.beta_schedule(**train.BETA_SCHEDULE["train"]))
        b = train.synthetic_batch(B, N, hw, "cpu", seed=0)
        fn = lambda x, a, l: unet_ref.unet_forward(sd, hp, x, a, l)
        g = torch.Generator().manual_seed(1)

        def one():
            t = torch.randint(1, 2000, (B,), generator=g)
            u, noise = torch.rand(B, 1, generator=g), torch.randn(B, 3, hw, hw, generator=g)
            opt.zero_grad()
            loss = vfr.train_loss(fn, sched, b["y_cond"], b["view_count"], b["angle"], b["y_0"], t, u, noise, True)
            loss.backward()
            opt.step()

        if warm:
            one()                                 # warm-up (oneDNN primitive creation)
        t0 = time.perf_counter()
        for _ in range(iters):
            one()
        dt = (time.perf_counter() - t0) / iters
        return B * N / dt, dt

    def sampler_leg_cpu(steps, threads, B=1, N=n_views, T=1000):
        torch.set_num_threads(threads)
        torch.manual_seed(0)
        sd = {k: v.clone() for k, v in UNet(**hp).state_dict().items()}
        kw = dict(train.BETA_SCHEDULE["test"])
        kw["num_timesteps"] = T
        sched = vfr.schedule_buffers(vfr.beta_schedule(**kw))
        b = train.synthetic_batch(B, N, hw, "cpu", seed=0)
        fn = lambda x, a, l: unet_ref.unet_forward(sd, hp, x, a, l)
        g = torch.Generator().manual_seed(2)
        y = torch.randn(B, 3, hw, hw, generator=g)
        with torch.no_grad():
            for i in range(steps + 1):
                if i == 1:
                    t0 = time.perf_counter()      # step 0 is the warm-up
                t = torch.full((B,), max(T - 1 - i, 0), dtype=torch.long)
                y, _, _ = vfr.p_sample(fn, sched, y, b["y_cond"], b["view_count"], b["angle"], t,
                                       torch.randn(B, 3, hw, hw, generator=g))
        return (time.perf_counter() - t0) / steps

    # PyTorch-CPU eager does not scale to every core of a large host on batches this small (on the 2 x 64-core box
    # 128 threads are SLOWER than one): calibrate the thread count on one iteration each and time the best one
    calib = {}
    for th in sorted({t for t in (8, 16, 32, 64, threads_all) if t <= threads_all}):
        calib[th] = train_leg(2, 1, th)[0]
    best = max(calib, key=calib.get)
    threads_max = threads_all
    # the bench workload itself is eight times larger than the calibration sample: let it also try two and four times the
    # calibrated thread count (one iteration each) before the timed pair
    c2_calib = {}
    for th in sorted({t for t in (best, 2 * best, 4 * best) if t <= threads_max}):
        c2_calib[th] = train_leg(16, 1, th, warm=False)[0]      # (12 s per iteration: the warm-up share is negligible)
    best = max(c2_calib, key=c2_calib.get)
    threads_all = best
    v_c2, dt_c2 = train_leg(16, 2, best)
    v_c4, dt_c4 = train_leg(8, 2, best)
    v_c1, dt_c1 = train_leg(2, 1, best, N=2)
    c1_chain = sampler_leg_cpu(10, best, B=2, N=2, T=10) * 10
    v_one, dt_one = train_leg(1, 1, 1)
    s_step = sampler_leg_cpu(10, threads_all)
    torch.set_num_threads(threads_max)
    return dict(value=v_c2, unit="view denoise-steps/s", cores=threads_all, kind="port",
                cpu_model=_cpu_model(), logical_cpus=os.cpu_count(), torch=torch.__version__,
                thread_calibration={str(k): v for k, v in calib.items()},
                thread_calibration_b16={str(k): v for k, v in c2_calib.items()},
                sample=f"C2, the bench workload itself: oracle training iteration (fwd+bwd+Adam), small UNet {hw}x{hw}, "
                       f"B=16 N={n_views} ({16 * n_views} views), 2 timed iterations after 1 warm-up, {dt_c2:.2f} s/iteration, "
                       f"{threads_all} threads (the fastest of the calibrated counts; the host offers {threads_max})",
                c4=dict(value=v_c4, cores=threads_all,
                        sample=f"C4: B=8 N={n_views} ({8 * n_views} views), 2 timed iterations after 1 warm-up, "
                               f"{dt_c4:.2f} s/iteration"),
                c1=dict(train_value=v_c1, train_s_per_iteration=dt_c1, chain_s=c1_chain, cores=threads_all,
                        sample="C1 in full: small UNet, B=2 N=2 64x64: one training iteration (after 1 warm-up) + the "
                               "10-step reverse chain (T=10 schedule, 10 p_sample calls after 1 warm-up call)"),
                one_thread=dict(value=v_one, cores=1, sample=f"B=1 ({n_views} views), 1 timed iteration after 1 warm-up, "
                                                             f"{dt_one:.2f} s/iteration"),
                sampler=dict(value=1.0 / (s_step * 1000), unit="sampled views/s at T=1000", cores=threads_all,
                             ms_per_step=s_step * 1e3,
                             sample=f"oracle p_sample, B=1 N={n_views}, 10 timed reverse steps after 1 warm-up, "
                                    "extrapolated linearly to T=1000"))


def pmc_launches():
    """Launch counts per kernel name in the committed PMC table (whole profiled run): a family's HBM bytes per launcher
    call = sum over its kernels of bytes per launch x that kernel's launches per launch of the family's main kernels."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench_pmc_traffic_kib_per_launch.json")), reverse=True):
        try:
            tab = json.load(open(path))
            return {k: float(d.get("launches", 0)) for k, d in tab.items() if not k.startswith("_")}
        except (OSError, ValueError, AttributeError):
            continue
    return {}


def pmc_traffic():
    """HBM bytes per launch per kernel from the committed rocprofv3 PMC passes of this same bench command
    (profiles/rNN_bench_pmc_traffic_kib_per_launch.json, newest round first: FETCH_SIZE and WRITE_SIZE collected in
    separate --pmc runs; FETCH_SIZE doubled per MI355X_MICROARCH.md's gfx950 correction).  PMC counters cannot be read
    from inside the process, so this is the recorded, not a live, figure; None when no table is committed."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench_pmc_traffic_kib_per_launch.json")), reverse=True):
        try:
            tab = json.load(open(path))
            meta = tab.pop("_meta", {})
            src = "profiles/" + os.path.basename(path)
            if meta.get("commit"):
                src += " (rocprofv3 --pmc passes of this bench command at commit %s)" % meta["commit"]
            return ({k: (2 * d["FETCH_SIZE"] + d["WRITE_SIZE"]) * 1024.0 for k, d in tab.items()
                     if "FETCH_SIZE" in d and "WRITE_SIZE" in d}, src)
        except (OSError, KeyError, ValueError):
            continue
    return {}, None


# kind (ops._launch) -> (label, binding roof).  Events bracket the C-ABI launcher, i.e. the kernel family it enqueues.
FAMILIES = {
    "wino_conv": ("wino44_conv_kernel<LOGW,MODE> (Winograd F(4x4,3x3): 64x64 / 32x32 maps) + wino_conv_kernel<LOGW,MODE> "
                  "(nested F(2,3)xF(4,3): 16x16 / 8x8 maps), + tail fix-ups: forward + dgrad of every stride-1 3x3 layer", "mfma"),
    "direct_conv": ("conv_mfma_kernel<KS,LOGW,MODE,NPT> / conv1x1_kernel<NCW>: forward + dgrad of the 1x1 / stride-2 layers", "mfma"),
    "wino_wgrad": ("wino44_wgrad_kernel (Winograd F(4x4,3x3), + slab-sum launch): weight gradient of the stride-1 3x3 layers", "mfma"),
    "direct_wgrad": ("conv1x1_wgrad_kernel / conv_wgrad_kernel (+ reduce): weight gradient of the 1x1 / stride-2 layers", "mfma"),
    "attn_fwd": ("attn_fwd_q32_kernel<false> (L=256) / attn_fwd_split_kernel<64> (L=64); kh / q16 kernels at other S", "mfma"),
    "attn_bwd": ("attention backward on the materialised P: attn_fwd_q32_kernel<true> (dS + dQ) + attn_bwd_dvdk_kernel at L=256, "
                 "bgemm_v2_kernel x4 + softmax_bwd_kernel at L=64", "mfma"),
    "bgemm": ("bgemm kernels outside attention (noise-level MLP linears)", "mfma"),
    "reduce": ("colsum / colsum_multi / rowsum / bias_grad / sumpool2 (parameter-gradient and pooling reductions)", "hbm"),
    "pack": ("pack_weights_multi / wino_pack_multi (weight re-layout, once per step)", "hbm"),
    "embed": ("time-embedding kernels (sincos, swish, grouped FeatureWiseAffine fwd/bwd)", "hbm"),
    "diffusion": ("stack_views (q_sample) / compose_fwd / compose_mse_bwd / gather_level", "hbm"),
    "misc": ("other launchers (concat fallback, dropout, ...)", "hbm"),
    "gn_fwd": ("gn_fwd_kernel (GroupNorm + Swish)", "hbm"),
    "gn_bwd": ("gn_bwd_fused_kernel (+ fused residual / skip gradient adds)", "hbm"),
    "adam": ("adam_multi_kernel", "hbm"),
}


# rocprofv3 kernel names whose PMC bytes make up a family's HBM traffic (profiles/rNN_bench_pmc_traffic_kib_per_launch.json):
# (kernels the launcher always runs, follow-up kernels it runs for some launches)
FAMILY_PMC = {
    "wino_conv": (("wino_conv_kernel", "wino44_conv_kernel"), ("wino_fixup_kernel", "wino44_fixup_kernel")),
    "wino_wgrad": (("wino44_wgrad_kernel",), ("wino44_reduce_kernel", "wino44_reduce_multi_kernel")),
    "direct_conv": (("conv_mfma_kernel", "conv1x1_kernel"), ()),
    "direct_wgrad": (("conv1x1_wgrad_kernel", "conv_wgrad_kernel"), ("wgrad_reduce_kernel",)),
    "attn_fwd": (("attn_fwd_q32_kernel<false>", "attn_fwd_kh_kernel", "attn_fwd_split_kernel"), ()),
    "attn_bwd": (("attn_fwd_q32_kernel<true>", "attn_bwd_dvdk_kernel", "bgemm_v2_kernel", "softmax_bwd_kernel"), ()),      # (one launcher call per kernel)
    "gn_fwd": (("gn_fwd_kernel",), ()), "gn_bwd": (("gn_bwd_fused_kernel",), ()), "adam": (("adam_multi_kernel",), ()),
}


# direct-convolution multiplies per multiply the Winograd kernels execute: the nested F(2,3) x F(4,3) forward / dgrad
# kernel (16x16 / 8x8 maps) 24 products per 2x4 outputs x 9 taps = 72 direct ones; F(4x4,3x3) -- forward / dgrad on the
# 64x64 / 32x32 maps and every weight gradient -- 36 products per 4x4 outputs x 9 taps = 144 direct ones
WINO_REDUCTION = {"wino_conv": 3.0, "wino_wgrad": 4.0}


def _reduction(fam, name):
    if fam == "wino_conv":
        return 4.0 if name.startswith("vf_wino44_conv") else 3.0
    return WINO_REDUCTION.get(fam, 1.0)


def _family(kind, name):
    if kind in ("conv_fwd", "conv_dgrad"):
        return "wino_conv" if name.startswith("vf_wino") else "direct_conv"
    if kind == "conv_wgrad":
        return "wino_wgrad" if name.startswith("vf_wino") else "direct_wgrad"
    return kind


def roofline(trainer, batch, S, ms_step, steps=3):
    """Per-launch HIP-event timing (events recorded on the launch stream) of the kernel families over `steps` extra
    iterations after the timed region.  The headline `frac` is the EXECUTED fp32-MFMA fraction of the dominant
    kernel, wino_conv_kernel: its launches' algorithmic direct-convolution FLOPs / 3 (the nested Winograd F(2,3)xF(4,3)
    multiplies 24 values per 2x4 outputs x 9 taps = 72 direct ones) / its own event time / 157.3 TF."""
    ops.st.KERNEL_LOG = []
    trainer.step(batch)              # untimed: the first eager iteration after graph replays (re-made gradient tensors,
    torch.cuda.synchronize()         # descriptor uploads) runs its first kernels 1-5 % slower than the following ones
    ops.st.KERNEL_LOG = []
    # the instrumented iterations run EAGERLY (a kernel log forces it) with two events around every launcher: their
    # own GPU time, bracketed by one more event pair per iteration, is what the family table must sum to -- not the
    # graph-replay step time of the headline (different launch regime: eager launches carry gaps)
    step_ev = []
    for _ in range(steps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        trainer.step(batch)
        e1.record()
        step_ev.append((e0, e1))
    torch.cuda.synchronize()
    eager_ms = sum(a.elapsed_time(b) for a, b in step_ev) / steps
    log, ops.st.KERNEL_LOG = ops.st.KERNEL_LOG, None
    agg = {}
    for kind, flops, e0, e1, _tag, name, nbytes in log:
        fam = _family(kind, name)
        a = agg.setdefault(fam, [0.0, 0.0, 0, 0.0, 0.0])
        a[0] += flops
        a[1] += e0.elapsed_time(e1) * 1e-3
        a[2] += 1
        a[3] += nbytes
        a[4] += flops / _reduction(fam, name)          # multiplies the matrix cores execute
    traffic, source = pmc_traffic()
    launches = pmc_launches()
    table = {}
    for fam, (f, s, n, nb, executed) in agg.items():
        label, bound = FAMILIES.get(fam, (fam, "hbm"))
        row = dict(kernel=label, bound=bound, launches_per_step=n // steps, ms_per_step=s / steps * 1e3,
                   avg_launch_us=s / n * 1e6)
        if f:
            row.update(direct_equivalent_tflops=f / s / 1e12, mfma_executed_tflops=executed / s / 1e12,
                       frac_of_fp32_mfma_peak=executed / s / 1e12 / PEAK_FP32_MATRIX_TFLOPS)
        if nb:
            row.update(algorithmic_gbs=nb / s / 1e9, frac_of_hbm_peak=nb / s / 1e9 / PEAK_HBM_GBS,
                       algorithmic_mb_per_launch=nb / n / 1e6)
        # both roofs per family: HBM bytes of the family's kernels from the committed PMC passes (bytes per
        # launch x this run's launches per step) over this run's event time
        mains, helpers = FAMILY_PMC.get(fam, ((), ()))
        n_main = sum(launches.get(k, 0.0) for k in mains)
        if n_main:
            per_call = sum(traffic.get(k, 0.0) * launches.get(k, 0.0) for k in mains + helpers) / n_main
            row.update(pmc_hbm_mb_per_launch=per_call / 1e6, pmc_gbs=per_call * n / s / 1e9,
                       frac_of_hbm_peak_pmc=per_call * n / s / 1e9 / PEAK_HBM_GBS)
        table[fam] = row
    # headline family: the Winograd forward/dgrad kernel when any launch took it, else the MFMA-bound family with the
    # largest time (small S: every 3x3 layer runs the direct kernel)
    head = "wino_conv" if "wino_conv" in agg else max((k for k in agg if agg[k][0] > 0), key=lambda k: agg[k][1])
    f, s, n, nb_head, executed = agg[head]
    head_traffic = table[head].get("pmc_hbm_mb_per_launch")
    head_traffic = head_traffic * 1e6 if head_traffic else None
    # floor of the whole step: every conv at the fp32 MFMA peak, the stride-1 3x3 layers at Winograd's multiply count
    # (forward + dgrad: nested F(2,3)xF(4,3); weight gradient: F(4x4,3x3))
    step_tflop_direct = GFLOP_PER_VIEW_TRAIN * S / 1e3
    g3 = GFLOP_3X3_S1_PER_VIEW_FWD * S / 1e3
    # (executed multiplies of THIS run's forward + dgrad launches: F(4x4) where it ran, nested elsewhere)
    fwd_dgrad_exec = agg["wino_conv"][4] / steps / 1e12 if "wino_conv" in agg else 2 * g3 / 3.0
    fwd_dgrad_direct = agg["wino_conv"][0] / steps / 1e12 if "wino_conv" in agg else 2 * g3
    step_tflop_wino = step_tflop_direct - (fwd_dgrad_direct - fwd_dgrad_exec) - g3 * (1 - 1 / WINO_REDUCTION["wino_wgrad"])
    floor_ms = step_tflop_wino / PEAK_FP32_MATRIX_TFLOPS * 1e3
    out = dict(bound="mfma", kernel=FAMILIES.get(head, (head,))[0], achieved=executed / s / 1e12, peak=PEAK_FP32_MATRIX_TFLOPS,
               unit="TFLOP/s", frac=executed / s / 1e12 / PEAK_FP32_MATRIX_TFLOPS,
               note="achieved = multiplies the matrix cores EXECUTE: algorithmic direct-convolution FLOPs "
                    "(2*S*Cout*Cin*9*H*W per launch) / 4 for the F(4x4,3x3) launches (64x64 / 32x32 maps), / 3 for the "
                    "nested F(2,3)xF(4,3) ones (16x16 / 8x8 maps), over the HIP-event time of all of them (forward + "
                    "dgrad); direct_equivalent_tflops prices the same launches at the direct count",
               direct_equivalent_tflops=f / s / 1e12, launches_per_step=n // steps, avg_launch_us=s / n * 1e6,
               algorithmic_gflop_per_launch=f / n / 1e9, executed_gflop_per_launch=executed / n / 1e9,
               traffic=head_traffic, traffic_unit="HBM bytes per launch (launch-weighted mean over the family's kernels, "
                                                  "fix-up launches included)", traffic_source=source,
               algorithmic_bytes_per_launch=nb_head / n if nb_head else None,
               traffic_ratio=(head_traffic / (nb_head / n) if nb_head and head_traffic else None),
               traffic_note="algorithmic bytes = input + output activations (+ residual) + the layer's weights, once "
                            "each; the PMC figure on top of that is: the 2-row halo re-read of every 2-row tile strip "
                            "(input rows fetched 2x on 64^2/32^2 maps when the halo misses L2), one more input read "
                            "per additional 64-channel co tile (Cout/64 workgroup columns stream the same x), the "
                            "transformed weight image U (24/9 = 2.67x the 3x3 weights, streamed by every workgroup: "
                            "L2 absorbs most, the rest is HBM/MALL fetch), and the K-split partial tiles of the tail "
                            "round (written raw and re-read by wino_fixup_kernel); per-shape passes (profiles/r05_hbm_per_shape.md) add one "
                            "more term for the F(4x4) kernel: its writes are 1.31x y because 40 VGPRs are spilled around (not "
                            "inside) every tile's chunk loop at the 256-register cap",
               step_floor_ms=dict(direct_fp32_mfma=step_tflop_direct / PEAK_FP32_MATRIX_TFLOPS * 1e3,
                                  winograd_adjusted=floor_ms,
                                  hbm=HBM_MB_PER_VIEW_TRAIN * S / 1e3 / PEAK_HBM_GBS * 1e3),
               step_frac_of_floor=floor_ms / ms_step, kernels=table,
               instrumented_eager_ms_per_step=eager_ms,
               unattributed_ms_per_step=eager_ms - sum(r["ms_per_step"] for r in table.values()),
               unattributed_note="GPU time of the instrumented EAGER iteration (one event pair around it) minus the sum "
                                 "of the per-launcher event pairs of every family: torch's own kernels (randn / rand / "
                                 "randint, a few copies) and the gaps between launches; the headline ms_per_step is the "
                                 "graph replay of the same launches")
    return out


GFLOP_PER_VIEW_FWD = 20.994            # one stacked view through the small UNet, forward
PARAM_BYTES = 33.9e6 * 4               # the weights every reverse step streams at least once


def sampler_leg():
    """Second half of BASELINE's metric: sampled views/s of the reverse-diffusion loop (config C5:
    T=1000 test schedule, N conditioning views).  A bounded number of reverse steps is timed and
    extrapolated linearly to T=1000 (the loop is strictly sequential with constant step cost).

    Roofline per configuration (one reverse step = one UNet forward over S = B*N stacked views + the fused tail):
      mfma : 20.994 GFLOP x S / ms_per_step / 157.3 TF (direct-convolution FLOPs; where the Winograd kernel runs -- the
             B=16 configuration -- the step executes fewer multiplies and `frac_of_winograd_floor` prices that)
      hbm  : (weights once + 141.7 MB ideal-fusion activation traffic per view) / ms_per_step / 8 TB/s
    At B=1 both fractions are small by construction: the step is a chain of ~260 short dependent kernels (graph replay),
    i.e. latency-bound; `launches_per_step` and `us_per_launch` say how short."""
    from view_fusion_amd import sampling_bench
    model = train.build_model(device="cuda:0", phase="test")
    out = {"unit": "completed target views/s at T=1000", "schedule": "linear T=1000 1e-4..0.09"}
    for B, N, steps in ((1, 1, 100), (1, 6, 100), (1, 12, 100), (16, 6, 20)):
        r = sampling_bench.time_sampler(B, N, steps=steps, model=model, use_graph=None)
        S, ms = B * N, r["ms_per_step"]
        gflop = GFLOP_PER_VIEW_FWD * S
        nbytes = PARAM_BYTES + 141.7e6 * S
        # the stride-1 3x3 layers run the nested Winograd kernel from S = 48 on (1/3 of the direct multiplies): the
        # fraction of the fp32 MFMA peak is priced on EXECUTED multiplies, so it cannot exceed 1
        wino = S >= 48
        gflop_exec = gflop - (GFLOP_3X3_S1_PER_VIEW_FWD * S * (1 - 1 / WINO_REDUCTION["wino_conv"]) if wino else 0.0)
        mfma_floor_ms = gflop_exec / PEAK_FP32_MATRIX_TFLOPS
        row = dict(sampled_views_per_sec=r["sampled_views_per_sec"], ms_per_step=ms,
                   view_unet_evals_per_sec=r["view_unet_evals_per_sec"], steps_timed=steps, graph_replay=bool(S <= 16),
                   roofline=dict(gflop_per_step=gflop, executed_gflop_per_step=gflop_exec,
                                 direct_equivalent_tflops=gflop / ms, achieved_tflops=gflop_exec / ms,
                                 mfma_floor_ms=mfma_floor_ms, frac_of_fp32_mfma_peak=mfma_floor_ms / ms,
                                 mbytes_per_step=nbytes / 1e6, hbm_floor_ms=nbytes / PEAK_HBM_GBS / 1e6,
                                 frac_of_hbm_peak=nbytes / PEAK_HBM_GBS / 1e6 / ms,
                                 weight_stream_floor_ms=PARAM_BYTES / PEAK_HBM_GBS / 1e6))
        if wino:
            row["roofline"]["frac_of_winograd_floor"] = mfma_floor_ms / ms
        n_launch = getattr(sampling_bench, "LAST_LAUNCHES_PER_STEP", None)
        if n_launch:
            row["roofline"].update(abi_launcher_calls_per_step=n_launch, us_per_launcher_call=ms * 1e3 / n_launch)
        out[f"B{B}_N{N}"] = row
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
    ap.add_argument("--graph", type=int, default=int(os.environ.get("VF_STEP_GRAPH", "1")),
                    help="1 (default): single-process runs replay the iteration as one HIP graph per batch geometry -- the "
                         "same launches on the same data, host enqueue 0.4 instead of 13 ms per step; 0: eager launches. "
                         "Multi-process runs with the gradient arena replay too (the segment all-reduces are part of "
                         "the graph on RCCL); VF_REDUCER=ddp stays eager.")
    args = ap.parse_args()

    rank, local_rank, world = train.init_distributed()
    assert world == args.gpus, f"launched with WORLD_SIZE={world} but --gpus {args.gpus}"
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    model = train.build_model(device=str(dev), seed=0)
    trainer = train.Trainer(model, world=world, local_rank=local_rank, graph=bool(args.graph))
    batch = train.synthetic_batch(args.batch, args.views, 64, dev, seed=rank, ragged=args.ragged)
    S = int(batch["view_count"].sum())

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # a graph is captured on the (GRAPH_AFTER+1)-th iteration of a geometry: keep the capture out of the timed region
    # (with the gradient arena one more: its first iteration lays the arena out and is not a sighting)
    need = train.Trainer.GRAPH_AFTER + 1 + (1 if trainer.arena is not None else 0)
    for _ in range(max(args.warmup, need) if trainer.use_graph else args.warmup):
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
    # what the job looked like from every rank: the driver can see that N processes on N devices took part, over which
    # backend / RCCL version, and which launch mode each ended in (train.Trainer.mode)
    info = trainer.dist_info()
    info["device"] = torch.cuda.get_device_name(dev) + f" #{local_rank}"
    ranks = [info]
    if world > 1:
        ranks = [None] * world
        dist.all_gather_object(ranks, info)

    if rank == 0:
        ms = dt / args.steps * 1e3
        res = {
            "metric": "denoise_steps_per_sec", "value": S * world * args.steps / dt,
            "unit": "view denoise-steps/s (UNet fwd+bwd per stacked view, incl. compose/MSE/Adam)",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            # VF_BF16X3=1 (experiment, default off): the 1x1 convolutions' forward / dgrad products run as bf16x3 split
            # products on the bf16 matrix path -- such a line is NOT the fp32-MFMA headline and says so here
            "dtype": "f32" if not ops.st.BF16X3 else "f32 (1x1 conv fwd+dgrad: bf16x3 split products on the bf16 MFMA path)",
            "data": "synthetic",
            "config": {"workload": "small UNet 64x64 (33.9M params), B=%d/GPU N=%d (S=%d views/GPU), training "
                                   "iteration fwd+bwd+Adam, linear T=2000 schedule%s" % (args.batch, args.views, S,
                                                                                     ", ragged view_count" if args.ragged else ""),
                       "global_batch": args.batch * world, "views": args.views,
                       "parallelism": "dp%d" % world,
                       "launch": {"graph": "HIP-graph replay of the whole iteration",
                                  "captured": "HIP-graph replay of the whole iteration, RCCL all-reduces inside the graph",
                                  "split": "HIP-graph replay of forward + backward, then six eager RCCL all-reduces + Adam",
                                  "eager": "eager launches"}[trainer.mode if trainer.graph_steps else "eager"]},
            "dist": {"world_size": dist.get_world_size() if world > 1 else 1, "backend": info.get("backend"),
                     "rccl_version": info.get("rccl_version"), "reducer": info["reducer"], "ranks": ranks},
            "iters_per_sec": args.steps / dt, "loss": loss_val,
            "achieved_tflops_total": 62.98e9 * S * world * args.steps / dt / 1e12,
        }
        # the extra legs must never cost the headline line: a failure is reported inside the JSON instead
        def leg(key, fn, *a):
            try:
                res[key] = fn(*a)
            except Exception as e:          # noqa: BLE001
                res[key] = {"error": f"{type(e).__name__}: {e}"}

        if world == 1 and not args.no_roofline:
            leg("roofline", roofline, trainer, batch, S, ms)
        if world == 1 and not args.no_sampler:
            leg("sampler", sampler_leg)
        if world == 1 and not args.no_cpu_baseline:
            leg("cpu_baseline", cpu_baseline, 64, args.views)
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
