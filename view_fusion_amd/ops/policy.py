"""KERNEL-CHOICE POLICY, in one place: which kernel runs a conv layer at a given (S, Cin, Cout, H, W) -- direct /
nested Winograd / F(4x4) for forward + dgrad (`wino_kind`), the F(4x4) weight-gradient kernel (`use_winograd_wgrad`),
the sampler's one-launch convs (`use_small_conv`, `can_fold_residual`).  Thresholds live on `state.st`."""
import ctypes

import torch

from .. import _lib
from .state import st


# kernel kind (1 nested Winograd, 2 F(4x4)) -> (pack cache attribute, C-ABI names: pack sizes, pack, conv, workspace)
_WINO_ABI = {1: ("_vf_wpack", "vf_wino_pack_sizes", "vf_wino_pack_weights", "vf_wino_conv_fwd", "vf_wino_conv_ws_floats"),
             2: ("_vf_w4pack", "vf_wino44_pack_sizes", "vf_wino44_pack_weights", "vf_wino44_conv_fwd", "vf_wino44_conv_ws_floats")}

# Cost model behind the choice between the two Winograd forward / dgrad kernels (shader cycles, measured on S = 96 with
# tools/wino44_table.py / tools/wino44f_stamps.py, round 4): a workgroup tile costs (chunks + F) x CH cycles,
#   nested F(2,3)xF(4,3): CH = 3340 per 8-channel chunk of a 256-pixel tile,  F = 4.2 chunk-times of prologue + epilogue
#   F(4x4,3x3)          : CH = 5390 per chunk of a 512-pixel tile,            F = 4.7
# the tiles run in rounds of 256 (one workgroup per CU); K-split tail parts additionally write and re-read their raw
# partial outputs (64 / 128 KB per part, priced at 4 TB/s = 2000 bytes per cycle) and pay the fix-up launch.
_WINO_COST = {1: (3340.0, 4.2, 64 * 32 * 8), 2: (5390.0, 4.7, 64 * 32 * 16)}


def _wino_cycles(kind, S, Cin, Cout, H, W):
    lib = _lib.load()
    ch, F, part_floats = _WINO_COST[kind]
    tiles = ctypes.c_int(0)
    getattr(lib, "vf_wino_conv_fill_pct" if kind == 1 else "vf_wino44_conv_fill_pct")(S, Cin, Cout, H, W, ctypes.byref(tiles))
    T = tiles.value
    parts = getattr(lib, _WINO_ABI[kind][4])(S, Cin, Cout, H, W) // part_floats     # K-split tail parts (0: plain grid)
    nch = (Cin + 7) // 8
    if parts == 0:
        return -(-T // 256) * (nch + F) * ch
    ntail = T % 256
    split = max(1, parts // max(ntail, 1))
    cyc = (T // 256) * (nch + F) * ch + -(-parts // 256) * (-(-nch // split) + F) * ch
    return cyc + parts * part_floats * 4 * 2.5 / 2000.0 + 8000.0     # partials written + read (+ output), fix-up launch




def wino_kind(S, Cin, Cout, H, W, KS, m, train=None):
    """Cached front of _wino_kind (the decision costs up to a dozen host calls into the library; an eager iteration asks
    it twice per conv layer)."""
    if train is None:
        train = torch.is_grad_enabled()
    key = (S, Cin, Cout, H, W, KS, m, bool(train), st.WINOGRAD, st.WINOGRAD44, st.FORCE_WINOGRAD, st.FORCE_WINOGRAD44, st.WINO_MIN_TILES,
           st.WINO_MIN_FILL)
    k = st._WINO_KIND_CACHE.get(key)
    if k is None:
        if len(st._WINO_KIND_CACHE) > 4096:
            st._WINO_KIND_CACHE.clear()
        k = st._WINO_KIND_CACHE[key] = _wino_kind(S, Cin, Cout, H, W, KS, m, train)
    return k


def _wino_kind(S, Cin, Cout, H, W, KS, m, train):
    """Which kernel runs the forward AND the dgrad pass of a conv layer (they share one packed-weight format):
    0 direct (conv.hip), 1 nested Winograd F(2,3)xF(4,3) (winograd24.hip), 2 Winograd F(4x4,3x3) (winograd44f.hip).

    F(4x4) executes 25 % fewer multiplies than the nested kernel but its workgroup tile is 64 channels x 512 pixels:
    on the 32x32 / 64x64 maps it is taken when the cost model above prices forward + dgrad (`train`; forward alone
    otherwise; default: whether autograd is recording) below the nested kernel's -- a grid of 1.5 rounds with a short K (128 -> 128 at 32x32, S = 96) is the case
    it loses.  The nested kernel runs ONE 256-pixel workgroup per CU: taken when its tile count (after the K-split of
    the tail tiles) keeps >= 65 % of the CUs busy; small batches (sampler) stay on the direct kernel (+ split-K)."""
    if not st.WINOGRAD or KS != 3 or m not in (0, 2):
        return 0
    lib = _lib.load()
    nested = 0
    if lib.vf_wino_supported(H, W, m):
        if st.FORCE_WINOGRAD:
            nested = 1
        else:
            tiles = ctypes.c_int(0)
            fill = lib.vf_wino_conv_fill_pct(S, Cin, Cout, H, W, ctypes.byref(tiles))
            nested = 1 if (tiles.value >= st.WINO_MIN_TILES and fill >= st.WINO_MIN_FILL) else 0
    if st.WINOGRAD44 and lib.vf_wino44_supported(H, W, m):
        if st.FORCE_WINOGRAD44:
            return 2
        if nested and not st.FORCE_WINOGRAD:
            dirs = ((Cin, Cout), (Cout, Cin)) if train else ((Cin, Cout),)
            c1 = sum(_wino_cycles(1, S, ci, co, H, W) for ci, co in dirs)
            c2 = sum(_wino_cycles(2, S, ci, co, H, W) for ci, co in dirs)
            if c2 < c1:
                return 2
    return nested


def use_winograd(S, Cin, Cout, H, W, KS, m, train=None):
    return wino_kind(S, Cin, Cout, H, W, KS, m, train) != 0


def use_winograd_wgrad(S, Cin, Cout, H, W, KS, m):
    """Weight gradients split over (co, ci, tile range), so the grid fills the chip at any map size."""
    if not (st.WINOGRAD and st.WINOGRAD_WGRAD) or KS != 3 or not _lib.load().vf_wino_wgrad_supported(H, W, m):
        return False
    return st.FORCE_WINOGRAD or S * (H // 2) * (W // 2) >= st.WINO_WGRAD_MIN_TILES


# The sampler at few stacked views: one launch per conv layer (csrc/conv_small.hip, K split inside the workgroup)
# instead of split-K partials + a reduce launch.  Taken without autograd only.  Measured per layer against the split-K
# route (tools/small_conv.py): the 1x1 kernel wins 3 us per layer up to ~1000 workgroups of 32 channels x
# 16 pixels (S <= 3-6 on the 16x16 maps where the attention projections live); the 3x3 kernel wins 2-5 us per layer for
# ONE view and Cin <= 256 and loses from two views on (its workgroups are all fixed cost).


def use_small_conv(S, Cin, Cout, H, W, KS, m):
    """(only consulted with autograd off)"""
    if not st.SMALL_CONV or not _lib.load().vf_conv_small_supported(Cin, Cout, H, W, KS, m):
        return False
    wgs = S * ((Cout + 31) // 32) * (H * W // 16)
    if KS == 1:
        return wgs <= st.SMALL_CONV_MAX_WGS[0]
    return wgs <= st.SMALL_CONV_MAX_WGS[1] and Cin <= st.SMALL_CONV_MAX_CIN3 and S <= st.SMALL_CONV_MAX_S3


# Round 5, sampler at N >= 2: the Winograd conv's fix-up launch evaluates the GroupNorm behind the conv (vf_wino_conv_fwd_gn)


def can_fold_residual(S, C, H, W, res_layer):
    """Inference: may a residual block's last 3x3 conv (C -> C on an H x W map) take its residual 1x1 conv `res_layer`
    along as extra K (vf_conv_small_res: one launch instead of two)?  Only where that conv runs the one-launch kernel."""
    return (st.RES_FOLD and not torch.is_grad_enabled() and isinstance(res_layer, torch.nn.Conv2d)
            and res_layer.weight.shape[1] % 4 == 0 and use_small_conv(S, C, C, H, W, 3, 0))
