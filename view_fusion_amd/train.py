"""Synthetic-data training / sampling harness for the ViewFusion hot path.

Mirrors the reference's iteration semantics (experiment.py:90-120 model/optimizer/DDP setup,
:265-293 the step: set LR -> zero_grad -> model(...) -> backward -> Adam.step) without its
dataset, wandb and checkpoint plumbing.  One process per GPU; gradients are averaged over RCCL
(backend "nccl" on ROCm) by the gradient arena of reducer.py -- the backward kernels write into the
communication buffer, one all-reduce per segment overlaps the rest of the backward pass -- or, with
VF_REDUCER=ddp, by torch's DistributedDataParallel as in the reference.
"""
import math
import os

import torch
import torch.distributed as dist
from torch.nn.parallel import DistributedDataParallel

from . import reducer
from .unet import UNet
from .view_fusion import ViewFusion

SMALL_UNET = dict(in_channel=6, out_channel=6, inner_channel=64, norm_groups=32, channel_mults=(1, 2, 3, 5),
                  attn_res=(16,), res_blocks=3, image_size=64)          # configs/small-v100.yaml:19-30
BETA_SCHEDULE = {
    "train": dict(schedule="linear", num_timesteps=2000, linear_start=1e-6, linear_end=1e-2),
    "test": dict(schedule="linear", num_timesteps=1000, linear_start=1e-4, linear_end=0.09),
}                                                                        # configs/small-v100.yaml:9-18


class LrScheduler:
    """Linear warm-up to `peak_lr` over `peak_it`, then peak * decay_rate**(dt/decay_it)
    (reference utils/schedulers.py, constants from experiment.py:112-117)."""

    def __init__(self, peak_lr=1e-4, peak_it=2500, decay_rate=0.16, decay_it=4000000):
        self.peak_lr, self.peak_it, self.decay_rate, self.decay_it = peak_lr, peak_it, decay_rate, decay_it

    def get_cur_lr(self, it):
        if it < self.peak_it:
            return self.peak_lr * (it / self.peak_it)
        return self.peak_lr * self.decay_rate ** ((it - self.peak_it) / self.decay_it)


def init_rccl_group(local_rank, rank=None, world_size=None):
    """RCCL ("nccl" on ROCm) process group for one rank per GPU.  RCCL's kernels run on HIGH-PRIORITY streams: the
    compute kernels of this engine (one 512-thread Winograd workgroup per CU, all of its LDS and registers) leave no
    room for a collective's workgroups to co-reside, so the segment all-reduces that the gradient arena issues during
    the backward pass can only run when a CU frees up between workgroup rounds -- with priority they get those CUs
    first instead of queueing behind the whole next conv launch.  Knobs left to the environment (RCCL defaults are
    kept): NCCL_ALGO / NCCL_PROTO / NCCL_MIN_NCHANNELS; the payload is 6 segments of ~22 MiB per step, so the
    bandwidth-optimal ring / direct algorithms apply, not the latency ones."""
    torch.cuda.set_device(local_rank)
    kw = {} if rank is None else dict(rank=rank, world_size=world_size)
    try:
        opts = dist.ProcessGroupNCCL.Options(is_high_priority_stream=True)
        dist.init_process_group(backend="nccl", pg_options=opts, device_id=torch.device("cuda", local_rank), **kw)
    except (AttributeError, TypeError):          # a torch build without the option object
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank), **kw)


def init_distributed():
    """(rank, local_rank, world).  torchrun env -> process group; else single process.

    Backend: RCCL ("nccl" on ROCm) over xGMI, one rank per GPU, as the reference (utils/dist.py:21).  Two knobs exist
    for exercising the multi-rank code path on a box with fewer GPUs than ranks (tests/test_gpu_bench_two_rank.py):
    VF_DIST_BACKEND=gloo selects the gloo transport (RCCL refuses two ranks on one device) and VF_SHARE_GPU=1 maps
    rank r to device r % device_count.  The process group is created BEFORE the first HIP call of the process."""
    if "WORLD_SIZE" not in os.environ or int(os.environ["WORLD_SIZE"]) <= 1:
        return 0, 0, 1
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    backend = os.environ.get("VF_DIST_BACKEND", "nccl" if torch.cuda.device_count() > 0 else "gloo")
    if os.environ.get("VF_SHARE_GPU") == "1" and torch.cuda.device_count() > 0:
        local_rank %= torch.cuda.device_count()
    if backend == "nccl":
        init_rccl_group(local_rank)
    else:
        dist.init_process_group(backend=backend)
        if torch.cuda.device_count() > 0:
            torch.cuda.set_device(local_rank)
    return dist.get_rank(), local_rank, dist.get_world_size()


def synthetic_batch(B, N, hw, device, seed=0, ragged=False):
    """NMR-shaped random batch (data/nmr_dataset.py:10-52): images in [0,1], angle = 2pi/24*k."""
    g = torch.Generator().manual_seed(seed)
    y_0 = torch.rand(B, 3, hw, hw, generator=g)
    y_cond = torch.rand(B, N, 3, hw, hw, generator=g)
    angle = 2 * math.pi / 24 * torch.randint(0, 24, (B, 1), generator=g).float()
    view_count = torch.randint(1, N + 1, (B,), generator=g) if ragged else torch.full((B,), N)
    return dict(y_0=y_0.to(device), y_cond=y_cond.to(device), angle=angle.to(device), view_count=view_count)


def build_model(unet_params=None, beta_schedule=None, device="cuda", phase="train", weighting=True, seed=0):
    torch.manual_seed(seed)
    net = UNet(**(unet_params or SMALL_UNET))
    vf = ViewFusion(net, beta_schedule or BETA_SCHEDULE, weighting, weighting).to(device)
    vf.set_new_noise_schedule(device=torch.device(device), phase=phase)
    return vf


class _StepGraph:
    """One captured training iteration (forward, backward, Adam) for one batch geometry."""
    __slots__ = ("graph", "inputs", "view_count", "off", "vc", "loss", "adam", "keep", "seen", "grads")

    def __init__(self):
        self.graph, self.seen = None, 0


class Trainer:
    """The reference's iteration (experiment.py:265-293) around `model`.

    graph (default: on; the VF_STEP_GRAPH=0 environment variable or graph=False turn it off): GPU runs replay the WHOLE
    iteration -- weight packs, forward, backward, the multi-tensor Adam launch: ~1000 kernels -- as one HIP graph per
    batch geometry instead of enqueueing it launch by launch.  The geometry is the tensor shapes plus the stacked-view
    count S = sum(view_count): which sample owns which views is the offsets table the kernels read from device memory,
    so it is a graph INPUT like the images, and a ragged run (experiment.py:277-279 draws view_count per sample and
    iteration) needs one graph per S it meets (at most B*(N-1)+1), not one per view_count vector.  A geometry is
    captured after it has run eagerly `GRAPH_AFTER` times (descriptor tables, packed-weight buffers and optimizer
    state then exist and are only re-used); up to `GRAPH_MAX` graphs share one memory pool (each keeps its own
    gradient tensors, 136 MB for the small UNet).  Eager as before: injected arguments other than the draws t / u /
    noise (which are graph inputs), a device-resident view_count (reading it back would be a sync per step), a
    kernel log, torch DDP as the reducer (VF_REDUCER=ddp).  world > 1 with the gradient arena has three launch modes
    (`self.mode`): "split" (the default: forward + backward replayed, the six segment all-reduces and the Adam launch
    issued eagerly after each replay), "captured" (VF_CAPTURE_COLLECTIVES=1: the collectives recorded into the graph on
    RCCL's stream, Adam included; the default only for a group of one) and "eager".  A rank whose capture fails raises
    the arena's failure flag, which rides the gradient all-reduce; at iteration numbers all ranks compute alike
    (`_agree`) every rank reads the accumulated flag and, if any rank failed, ALL step down one mode together
    (captured -> split -> eager) and drop their graphs -- never a lasting mix of modes.  The captured step is the same launches on the same data: with the same random draws its
    parameters match the eager step's bit for bit (tests/test_gpu_step_graph.py).  Learning rate and Adam bias
    corrections are read from device memory refreshed before each replay; torch's device RNG advances per replay
    exactly as it does per eager step.  The capture runs the forward on fresh leaf aliases of the parameters and takes
    the gradients with autograd.grad, so autograd state that callers keep alive (a loss with history) cannot pull
    another stream into it."""
    GRAPH_AFTER = 2
    GRAPH_MAX = 96
    AGREE_EVERY = 64            # multi-rank agreement: the failure flag is read at least this often (see _agree)
    inject_capture_failure = None   # set by tests (class or instance attribute), never read from the environment

    def __init__(self, model, world=1, local_rank=0, lr_warmup=2500, decay_it=4000000, bucket_cap_mb=32, graph=None):
        self.module = model
        self.model = model
        self.arena = None
        kind = os.environ.get("VF_REDUCER", "arena")
        if world > 1 and kind != "ddp":
            try:
                # "xgmi": the hand-written one-shot all-reduce over IPC-mapped peer arenas fused with Adam (correctness-
                # only so far: it has run with two processes on one GPU, never across devices); default: RCCL
                self.arena = reducer.ACTIVE = (reducer.XgmiArena if kind == "xgmi" else reducer.GradArena)(model, world)
            except (RuntimeError, dist.DistBackendError) as e:    # first collective of the job (parameter broadcast)
                import traceback
                traceback.print_exc()    # (anything else -- a programming error in the hook set-up -- propagates as is)
                reducer.ACTIVE = None
                raise SystemExit(
                    f"[view_fusion_amd] gradient-arena set-up failed at world={world}: {type(e).__name__}: {e}\n"
                    "  -> relaunch (a fresh process group, not a re-exec) with VF_REDUCER=ddp to use torch's "
                    "DistributedDataParallel for the gradient exchange (VF_REDUCER=xgmi: first back to the default, "
                    "the arena over RCCL); see tools/scale_run.md") from e
        elif world > 1:
            kw = dict(broadcast_buffers=False, gradient_as_bucket_view=True, bucket_cap_mb=bucket_cap_mb)
            if next(model.parameters()).is_cuda:
                kw.update(device_ids=[local_rank], output_device=local_rank)
            self.model = DistributedDataParallel(model, **kw)
            # DDP copies a gradient into its bucket from a hook on the AccumulateGrad node, during the backward pass:
            # a GroupNorm gradient whose column sum is deferred to the end of the pass would be read before it exists
            for p in model.parameters():        # (per parameter, not a process-wide switch: other models keep deferring)
                p._vf_no_defer = True
        # modules whose behaviour depends on train/eval mode (see step()): the blocks that carry a Dropout
        self._mode_modules = [m for m in model.modules() if getattr(m, "dropout", 0) and hasattr(m, "_drop")]
        self.sched = LrScheduler(peak_lr=1e-4, peak_it=lr_warmup, decay_it=decay_it, decay_rate=0.16)
        self._named = list(model.named_parameters())
        params = self._params = [p for _, p in self._named]
        if params[0].is_cuda:       # one multi-tensor HIP launch per step
            from .optim import FusedAdam
            self.opt = FusedAdam(params, lr=self.sched.get_cur_lr(0))
        else:                       # CPU harness tests (gloo) only
            self.opt = torch.optim.Adam(params, lr=self.sched.get_cur_lr(0))
        self.it = -1
        if graph is None:
            graph = os.environ.get("VF_STEP_GRAPH", "1") == "1"
        # (with torch DDP the iteration stays eager: its reducer is driven by autograd hooks on the host)
        self.use_graph = bool(graph) and params[0].is_cuda and (world == 1 or self.arena is not None) and \
            not getattr(self.arena, "owns_optimizer_step", False)     # (the xgmi reducer: eager launches only)
        self._graph_wanted = self.use_graph     # what the run was configured for (a local capture failure clears use_graph)
        self._graphs = {}           # geometry key -> _StepGraph
        self._pool = None           # the graphs' shared private memory pool
        self._scal = None           # device {lr, 1-b1^t, 1-b2^t}
        self._last_graph = None
        self._graph_epoch = None    # FusedAdam.graph_epoch the kept graphs were captured under (None: none captured)
        self.graph_steps = 0        # iterations that ran as a replay (diagnostics / tests)
        self.world = world
        self._check_base = self.GRAPH_AFTER     # multi-rank agreement: flag read at _check_base + 1, 4, 16, then every AGREE_EVERY
        self.demotions = 0

    @property
    def mode(self):
        """Launch mode of the iteration: "eager", "graph" (single process), "split" / "captured" (gradient arena)."""
        if not self.use_graph:
            return "eager"
        if self.arena is None:
            return "graph"
        return "captured" if self.arena.capturable else "split"

    def dist_info(self):
        """What a multi-rank run looks like from this rank (bench.py prints it)."""
        info = dict(world_size=self.world, reducer="none" if self.world == 1 else (("xgmi" if getattr(self.arena, "owns_optimizer_step", False) else "arena") if self.arena else "ddp"),
                    launch_mode=self.mode, graph_steps=self.graph_steps, mode_demotions=self.demotions)
        if dist.is_available() and dist.is_initialized():
            info.update(world_size=dist.get_world_size(), rank=dist.get_rank(), backend=dist.get_backend())
            if info["backend"] == "nccl":
                try:
                    info["rccl_version"] = ".".join(str(v) for v in torch.cuda.nccl.version())
                except Exception:           # noqa: BLE001
                    info["rccl_version"] = None
        if self.arena is not None:
            info["gradients_copied_last_step"] = self.arena.copied
        return info

    def _drop_graphs(self):
        self._graphs, self._last_graph, self._pool, self._graph_epoch = {}, None, None, None

    def _agree(self):
        """Multi-rank agreement on the launch mode (see the class doc).  Called at the top of every step; reads the
        arena's accumulated failure flag (one host sync) only at the iteration numbers below."""
        a = self.arena
        if a is None or a.flag_acc is None:
            return
        # d = 1, 4, 16 after the last change of mode (a capture that fails does so at a geometry's first capture, i.e.
        # early), then every AGREE_EVERY-th iteration for the rest of the run: a geometry first met at iteration 1000, or a
        # run whose `it` was restored from a checkpoint, is never more than AGREE_EVERY iterations away from the next
        # check (one host sync per 64 iterations).  Every rank computes the same iteration numbers.
        d = self.it - self._check_base
        if d < 1 or not (d in (1, 4, 16) or d % self.AGREE_EVERY == 0):
            return
        if float(a.flag_acc.item()) == 0.0:        # the same averaged value on every rank
            return
        was = "captured" if a.capturable and self._graph_wanted else ("split" if self._graph_wanted else "eager")
        if was == "captured":
            a.capturable = False
            self.use_graph = True
        else:
            self.use_graph = self._graph_wanted = False
        import sys
        print(f"[view_fusion_amd] rank {dist.get_rank() if dist.is_initialized() else 0}: a training-step capture failed "
              f"on some rank; all ranks step down from '{was}' to '{self.mode}' at iteration {self.it}", file=sys.stderr)
        self._drop_graphs()
        a.flag_value = 0.0
        a.flag_acc.zero_()
        self._check_base = self.it
        self.demotions += 1

    # -- whole-step HIP graph ---------------------------------------------------------------------------------------
    def _graph_key(self, batch, extra):
        from . import ops
        if not self.use_graph or ops.st.KERNEL_LOG is not None or (self.arena is not None and self.arena.flat is None):
            return None
        if any(k not in ("t", "u", "noise") for k in extra):       # injected draws are graph inputs, nothing else is
            return None
        vc = batch["view_count"]
        if torch.is_tensor(vc):
            if vc.is_cuda:          # reading it back would be a device sync per step
                return None
            vc = vc.tolist()
        ts = [("y_0", batch["y_0"]), ("y_cond", batch["y_cond"]), ("angle", batch["angle"])] + sorted(extra.items())
        if not all(torch.is_tensor(t) and t.is_cuda for _, t in ts):
            return None
        vc = tuple(int(v) for v in vc)
        if len(vc) != batch["y_0"].shape[0] or min(vc) < 1 or max(vc) > batch["y_cond"].shape[1]:
            return None                       # the eager path raises the reference's errors
        # the launch geometry depends on the stacked-view count S only: WHICH sample owns which views is the offsets
        # table the kernels read from device memory, a graph input like the images
        # (+ the addresses of what a captured step reads besides its inputs and is replaceable from outside: the noise
        # schedule buffer -- set_new_noise_schedule() makes new tensors -- and the parameter storage)
        g = getattr(self.module, "gammas", None)
        anchors = (g.data_ptr() if torch.is_tensor(g) else 0, self._params[0].data_ptr(), self._params[-1].data_ptr())
        return tuple((k, tuple(t.shape), t.dtype) for k, t in ts) + (anchors, sum(vc)), vc

    def _capture(self, e, key, vc, batch, extra):
        from . import ops
        dev = batch["y_0"].device
        arena = self.arena
        inj = self.inject_capture_failure    # tests only: "<rank>:<mode>" fails that rank's captures in that mode
        if inj and dist.is_initialized() and inj == f"{dist.get_rank()}:{self.mode}":
            raise RuntimeError("injected capture failure (Trainer.inject_capture_failure)")
        # With a host-driven transport (gloo) the graph ends with the backward pass; the exchange and the Adam launch
        # follow each replay eagerly.  On RCCL the segment all-reduces and Adam are part of the graph.
        split = arena is not None and not arena.capturable
        adam = None if split else self.opt.graph_begin()
        if adam is None and not split:
            return False
        if self._scal is None:
            self._scal = torch.zeros(3, device=dev, dtype=torch.float32)
        e.inputs = {k: {**batch, **extra}[k].detach().clone().contiguous() for k, _, _ in key[:-2]}
        e.view_count = torch.tensor(vc, dtype=torch.int64, device=dev)
        offs = ops.view_offsets(e.view_count, dev)   # resolved (one read-back) and remembered BEFORE the capture
        e.off, e.vc = offs[0], vc                    # the offsets table: rewritten when a replay's view_count differs
        tables = ops.prime_tables(getattr(self.module, "denoise_fn", None), key[-1], dev)
        self.opt.zero_grad()                           # the capture allocates the gradients in the graph's pool
        g = torch.cuda.CUDAGraph()
        kw = {} if self._pool is None else dict(pool=self._pool)
        ops.begin_capture(dev, sum(1 for m in self.module.modules() if isinstance(m, torch.nn.GroupNorm)),
                          sum(1 for m in self.module.modules() if isinstance(m, torch.nn.Conv2d)))
        failed = None
        try:
            with torch.cuda.graph(g, capture_error_mode="relaxed", **kw):
                try:        # an exception must not unwind through the capture: let it end, then report
                    # The captured forward runs on FRESH leaf tensors that alias the parameters' storage
                    # (functional_call), and its gradients are taken with autograd.grad: nothing passes through the
                    # parameters' own AccumulateGrad nodes.  Those remember the stream they were created on; if a
                    # caller keeps ANY tensor with autograd history of this model alive (a loss from an earlier
                    # forward), they are bound to the default stream, the backward pass forks the capture onto it, and
                    # hipStreamEndCapture takes the process down.
                    leaves = {n: p.detach().requires_grad_(True) for n, p in self._named}
                    if arena is not None:              # gradients are born in (or moved into) the all-reduce buffer
                        arena.capture_begin(list(leaves.values()))
                    loss = torch.func.functional_call(self.model, leaves, (), dict(view_count=e.view_count, **e.inputs))
                    grads = torch.autograd.grad(loss, list(leaves.values()), allow_unused=True)
                    if arena is not None:
                        grads = arena.capture_finish(grads)
                    for (_, p), gr in zip(self._named, grads):
                        p.grad = gr
                    if not split:
                        self.opt.step_captured(adam, self._scal)
                except Exception as err:      # noqa: BLE001
                    failed = err
        finally:
            fix = ops.end_capture()
        if failed is not None:
            if arena is not None:
                arena.capture_abort()
            raise failed
        if self._pool is None:
            self._pool = g.pool()
        e.adam = None if split else self.opt.graph_end(adam)
        self._graph_epoch = self.opt.graph_epoch if split else adam["epoch"]
        e.loss, e.graph = loss.detach(), g
        e.grads = list(grads)
        # what the captured launches address besides the graph's own pool
        e.keep = (fix, tables, offs, getattr(self.module, "gammas", None))
        return True

    def _graph_step(self, e, vc, batch, extra):
        if vc != e.vc:
            off = [0]
            for v in vc:
                off.append(off[-1] + v)
            e.off.copy_(torch.tensor(off, dtype=torch.int32).pin_memory(), non_blocking=True)
            e.vc = vc
        for k, dst in e.inputs.items():
            src = extra[k] if k in extra else batch[k]
            dst.copy_(src.reshape(dst.shape), non_blocking=True)
        if self._last_graph is not e:                  # .grad shows the gradients of the graph that ran last
            for p, gr in zip(self._params, e.grads):
                p.grad = gr
            self._last_graph = e
        if e.adam is None:                             # host-driven transport: exchange + Adam follow the replay
            e.graph.replay()
            self.arena.reduce_all()
            self.opt.step()
        else:
            self.opt.graph_tick(e.adam, self._scal)
            e.graph.replay()
        self.graph_steps += 1
        return e.loss.clone()

    def step(self, batch, **extra):
        """One reference iteration; returns the (device) loss tensor, no host sync."""
        self.it += 1
        lr = self.sched.get_cur_lr(self.it)
        for gparam in self.opt.param_groups:
            gparam["lr"] = lr
        # The reference calls model.train() every iteration (experiment.py:286); Module.train() walks all ~1400
        # submodules (2 ms), so it runs only when some module that HAS a mode-dependent layer is not in training mode.
        # The residual blocks' Dropout is the only such layer; the root flag alone would miss `vf.denoise_fn.eval()`.
        if not self.model.training or any(not m.training for m in self._mode_modules):
            self.model.train()
        self._agree()
        key, vc = self._graph_key(batch, extra) or (None, None)
        if key is not None and self._graph_epoch is not None and self.opt.graph_epoch != self._graph_epoch:
            # the optimizer state was replaced (load_state_dict, a changed parameter set): the captured steps still
            # address the old moment tensors -- drop them; every geometry is captured again after its eager sightings
            self._drop_graphs()
        if key is not None:
            e = self._graphs.get(key)
            if e is None and len(self._graphs) < self.GRAPH_MAX:
                e = self._graphs[key] = _StepGraph()
            if e is not None:
                if e.graph is None and e.seen >= self.GRAPH_AFTER:
                    try:
                        if not self._capture(e, key, vc, batch, extra):
                            self.use_graph = False
                            if self.arena is not None:     # this rank stays eager: tell the others (as a failure does)
                                self.arena.flag_value = 1.0
                    except Exception as err:           # leave the run on the eager path, loudly
                        import sys
                        print(f"[view_fusion_amd] training-step graph capture failed ({type(err).__name__}: {err}); "
                              "continuing eagerly" + ("" if self.arena is None else
                                                      " and telling the other ranks (failure flag)"), file=sys.stderr)
                        self.use_graph, e.graph = False, None
                        if self.arena is not None:     # rides the next gradient all-reduce; every rank steps down at
                            self.arena.flag_value = 1.0    # the next agreement point (_agree)
                        self.opt.zero_grad()
                if e.graph is not None:
                    return self._graph_step(e, vc, batch, extra)
                e.seen += 1
        self._last_graph = None
        self.opt.zero_grad()
        fused = self.arena is not None and getattr(self.arena, "owns_optimizer_step", False)
        if fused:                         # the exchange kernels apply Adam themselves (reducer.XgmiArena)
            self.arena.begin_step(self.opt)
        loss = self.model(y_0=batch["y_0"], y_cond=batch["y_cond"], view_count=batch["view_count"],
                          angle=batch["angle"], **extra)
        loss.backward()
        if self.arena is not None:
            self.arena.finish()
        if not fused:
            self.opt.step()
        # detached: a caller that keeps the returned loss should not keep this iteration's autograd graph (its saved
        # activations: several GB at B=16) alive with it
        return loss.detach()
