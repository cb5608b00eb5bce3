"""Packed-weight caches of the conv layers: direct, nested-Winograd, F(4x4) and the sampler's one-launch 3x3 format;
`pack_all` re-packs every layer of a network with one launch per format at the start of a training forward."""
import ctypes

import torch

from .. import _lib
from .state import st
from .core import _MODES, _call, _check, _ptr, _stream, _workspace
from .policy import _WINO_ABI, wino_kind


# ---------------------------------------------------------------------------------------------
def _conv_ws(device, S, Cin, Cout, H, W, KS):
    """Split-K workspace for small grids (sampler regime); (None, 0) when the grid fills the chip."""
    need = _lib.load().vf_conv_fwd_ws_floats(S, Cin, Cout, H, W, KS)
    if need <= 0:
        return None, 0
    ws = _workspace(device, need)
    return ws, ws.numel()


def _wino_ws(device, S, Cin, Cout, H, W, kind=1):
    """Room for the K-split tail tiles of a Winograd launch whose tile count does not divide the CUs."""
    need = getattr(_lib.load(), _WINO_ABI[kind][4])(S, Cin, Cout, H, W)
    if need <= 0:
        return None, 0
    ws = _workspace(device, need)
    return ws, ws.numel()


def _packed(layer, force):
    """Packed forward / dgrad weights of a conv layer.

    Inference: cached, keyed on the parameter's version counter (load_state_dict / copy_ / FusedAdam.step bump it).
    Training (`force`): re-packed on every forward -- fused optimizers (torch._fused_adam_) update
    parameters WITHOUT bumping `_version`, so the counter cannot be trusted across steps; for the same reason a
    training pack never becomes a cache hit for a later no-grad forward (its key stays None, see pack_all).
    """
    w = layer.weight
    cache = getattr(layer, "_vf_pack", None)
    key = (w._version, w.data_ptr(), w.device)
    if not force and cache is not None and cache[0] == key:
        return cache[1], cache[2]
    if force and cache is not None and getattr(layer, "_vf_pack_fresh", False):
        object.__setattr__(layer, "_vf_pack_fresh", False)       # packed by pack_all() for THIS forward
        return cache[1], cache[2]
    Cout, Cin, KS, _ = w.shape
    nf, nb = ctypes.c_long(), ctypes.c_long()
    _lib.call("vf_conv_pack_sizes", Cout, Cin, KS, ctypes.byref(nf), ctypes.byref(nb))
    if cache is not None and cache[1].numel() == nf.value and cache[1].device == w.device:
        wf, wb = cache[1], cache[2]
    else:
        wf = torch.empty(nf.value, device=w.device, dtype=torch.float32)
        wb = torch.empty(nb.value, device=w.device, dtype=torch.float32)
    wd = w.detach()
    _check(wd)
    _call("vf_conv_pack_weights", _ptr(wd), _ptr(wf), _ptr(wb), Cout, Cin, KS, _stream())
    object.__setattr__(layer, "_vf_pack", (None if force else key, wf, wb))   # training packs are never cache hits
    return wf, wb




def _packed_wino(layer, force, kind=1):
    """Winograd-transformed packed weights (forward / dgrad) of a 3x3 layer in the format of kernel `kind`
    (wino_kind); same caching rules as _packed."""
    attr, f_sizes, f_pack = _WINO_ABI[kind][:3]
    w = layer.weight
    cache = getattr(layer, attr, None)
    key = (w._version, w.data_ptr(), w.device)
    if not force and cache is not None and cache[0] == key:
        return cache[1], cache[2]
    if force and cache is not None and getattr(layer, attr + "_fresh", False):
        object.__setattr__(layer, attr + "_fresh", False)
        return cache[1], cache[2]
    Cout, Cin = w.shape[0], w.shape[1]
    nf, nb = ctypes.c_long(), ctypes.c_long()
    _lib.call(f_sizes, Cout, Cin, ctypes.byref(nf), ctypes.byref(nb))
    if cache is not None and cache[1].numel() == nf.value and cache[1].device == w.device:
        uf, ub = cache[1], cache[2]
    else:
        uf = torch.empty(nf.value, device=w.device, dtype=torch.float32)
        ub = torch.empty(nb.value, device=w.device, dtype=torch.float32)
    wd = w.detach()
    _check(wd)
    _call(f_pack, _ptr(wd), _ptr(uf), _ptr(ub), Cout, Cin, _stream())
    object.__setattr__(layer, attr, (None if force else key, uf, ub))
    return uf, ub


def pack_all(root, S=None):
    """Training forward: re-pack the weights of EVERY conv layer under `root` with one launch per
    format (device-side descriptor tables, rebuilt only if a parameter moved).  Layers annotated by
    the UNet with their output size (`_vf_geom` = (H, mode)) that will take the Winograd path at
    batch S get the transformed pack, all others the direct pack.  Each layer's fresh pack is
    consumed by its next training-mode conv2d call."""
    plan = getattr(root, "_vf_pack_plan", None)
    layers = plan[0] if plan is not None else [m for m in root.modules() if isinstance(m, torch.nn.Conv2d)]
    if not layers:
        return
    _check(layers[0].weight.detach())

    def kind_of(l):
        geom = getattr(l, "_vf_geom", None)
        if geom is None or S is None:
            return 0
        return wino_kind(S, l.weight.shape[1], l.weight.shape[0], geom[0], geom[0], l.weight.shape[2], _MODES[geom[1]],
                         train=True)

    key = tuple((l.weight.data_ptr(), kind_of(l)) for l in layers)
    if plan is None or plan[1] != key:
        dev = layers[0].weight.device
        rows, first = {0: [], 1: [], 2: []}, {0: 0, 1: 0, 2: 0}
        for l, (_, kind) in zip(layers, key):
            w = l.weight
            Cout, Cin, KS, _ = w.shape
            nf, nb = ctypes.c_long(), ctypes.c_long()
            if kind:
                _lib.call(_WINO_ABI[kind][1], Cout, Cin, ctypes.byref(nf), ctypes.byref(nb))
            else:
                _lib.call("vf_conv_pack_sizes", Cout, Cin, KS, ctypes.byref(nf), ctypes.byref(nb))
            pf = torch.empty(nf.value, device=dev, dtype=torch.float32)
            pb = torch.empty(nb.value, device=dev, dtype=torch.float32)
            nblk = (nf.value + nb.value + 255) // 256
            if kind:
                object.__setattr__(l, _WINO_ABI[kind][0], (None, pf, pb))
                rows[kind].append([w.data_ptr(), pf.data_ptr(), pb.data_ptr(), Cout, Cin, nf.value, nb.value, first[kind]])
            else:
                object.__setattr__(l, "_vf_pack", (None, pf, pb))
                rows[0].append([w.data_ptr(), pf.data_ptr(), pb.data_ptr(), Cout, Cin, KS, nf.value, nb.value, first[0]])
            first[kind] += nblk
        descs = tuple((torch.tensor(rows[k], dtype=torch.int64).to(dev) if rows[k] else None, len(rows[k]), first[k])
                      for k in (0, 1, 2))
        plan = (layers, key, descs)
        object.__setattr__(root, "_vf_pack_plan", plan)
    for k, fn in ((0, "vf_conv_pack_weights_multi"), (1, "vf_wino_pack_weights_multi"), (2, "vf_wino44_pack_weights_multi")):
        desc, n, blk = plan[2][k]
        if n:
            _call(fn, ctypes.c_void_p(desc.data_ptr()), n, blk, _stream())
    # A training pack is consumed once, through its `_fresh` flag, by this forward's conv2d call.  Its cache key stays
    # None (and the keys of the layer's OTHER formats are dropped too): an optimizer may update the weights without
    # touching `_version` (torch._fused_adam_), so after a training forward no cached pack of any format may be
    # trusted by a later no-grad forward (generate / p_sample after Trainer.step()).
    attrs = ("_vf_pack", "_vf_wpack", "_vf_w4pack")
    for l, (_, kind) in zip(layers, plan[1]):
        for k, attr in enumerate(attrs):
            c = getattr(l, attr, None)
            if c is not None and c[0] is not None:
                object.__setattr__(l, attr, (None, c[1], c[2]))
        object.__setattr__(l, attrs[kind] + "_fresh", True)
        c = getattr(l.weight, "_vf_small_pack", None)      # the sampler's one-launch 3x3 format: same rule (buffer kept)
        if c is not None and c[0] is not None:
            l.weight._vf_small_pack = (None, c[1])




def _packed_small(layer_or_weight):
    """3x3 weights in the load order of the one-launch kernel (vf_conv_small_pack), cached on the parameter and keyed on
    its version counter like _packed (inference only: the sampler's weights are static during a generate() call).
    None when packing is off or the holder is a bare tensor without a place for the cache."""
    w = layer_or_weight.weight if hasattr(layer_or_weight, "weight") else layer_or_weight
    if not st.SMALL_PACK or w.shape[2] != 3 or torch.is_grad_enabled():
        return None
    key = (w._version, w.data_ptr(), w.device)
    cache = getattr(w, "_vf_small_pack", None)
    if cache is not None and cache[0] == key:
        return cache[1]
    Cout, Cin = w.shape[0], w.shape[1]
    n = _lib.load().vf_conv_small_pack_floats(Cout, Cin)
    wp = cache[1] if (cache is not None and cache[1].numel() == n and cache[1].device == w.device) else \
        torch.empty(n, device=w.device, dtype=torch.float32)
    wd = w.detach()
    _check(wd)
    _call("vf_conv_small_pack", _ptr(wd), _ptr(wp), Cout, Cin, _stream())
    try:
        w._vf_small_pack = (key, wp)
    except AttributeError:
        pass
    return wp
