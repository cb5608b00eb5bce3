"""Convolution autograd Functions and the no-grad conv entry points (reference model/unet.py:42, 189, 198, 214, 238,
255, 256 with the FeatureWiseAffine / residual adds fused into the epilogue)."""
import ctypes

import torch

from .. import _lib, reducer
from .state import st
from .bf16x3 import _packed_b3, _use_b3
from .core import _MODES, _c, _call, _check, _launch, _ptr, _stream, _workspace
from .deferred import _defer_begin, _gout, _rowsum_get, _rowsum_put, _wred_ws
from .norm import group_norm
from .packing import _WINO_ABI, _conv_ws, _packed, _packed_small, _packed_wino, _wino_ws
from .policy import use_small_conv, use_winograd, use_winograd_wgrad, wino_kind


def _conv_small(x, x2, weight, bias, view_bias, residual, S, Cin, Cout, H, W, KS, m):
    wd = weight.detach()
    _check(wd)
    y = torch.empty(S, Cout, H, W, device=x.device, dtype=torch.float32)
    wp = _packed_small(weight) if KS == 3 else None
    if wp is not None:            # the general entry takes the packed copy
        _launch("conv_fwd", 2.0 * S * Cout * Cin * 9 * H * W, "vf_conv_small_gn", _ptr(x), None, 0, _ptr(wd), _ptr(bias),
                _ptr(view_bias), _ptr(residual), _ptr(y), S, Cin, Cout, H, W, 3, None, None, None, 0, 1e-5, 0, None, None,
                None, 0, 0, None, None, _ptr(wp), m, _stream(), tag=(Cin, Cout, H, KS, m))
        return y
    _launch("conv_fwd", 2.0 * S * Cout * Cin * KS * KS * H * W, "vf_conv_small", _ptr(x), _ptr(x2),
            x.shape[1] if x2 is not None else 0, _ptr(wd), _ptr(bias), _ptr(view_bias), _ptr(residual), _ptr(y), S, Cin,
            Cout, H, W, KS, m, _stream(), tag=(Cin, Cout, H, KS, m))
    return y


def _conv_small_res(x, layer, view_bias, res_layer, rx, rx2):
    S, Cin, H, W = x.shape
    Cout = layer.weight.shape[0]
    rC = res_layer.weight.shape[1]
    rC1 = rx.shape[1]
    assert rC1 + (rx2.shape[1] if rx2 is not None else 0) == rC and res_layer.weight.shape[0] == Cout
    wd, rwd = layer.weight.detach(), res_layer.weight.detach()
    _check(x, wd, rwd, layer.bias, view_bias, rx, rx2, res_layer.bias)
    y = torch.empty(S, Cout, H, W, device=x.device, dtype=torch.float32)
    wp = _packed_small(layer)
    if wp is not None:
        _launch("conv_fwd", 2.0 * S * Cout * (Cin * 9 + rC) * H * W, "vf_conv_small_gn", _ptr(x), None, 0, _ptr(wd),
                _ptr(layer.bias), _ptr(view_bias), None, _ptr(y), S, Cin, Cout, H, W, 3, None, None, None, 0, 1e-5, 0, None,
                _ptr(rx), _ptr(rx2), rC1, rC, _ptr(rwd), _ptr(res_layer.bias), _ptr(wp), 0, _stream(), tag=(Cin, Cout, H, 3, 0))
        return y
    _launch("conv_fwd", 2.0 * S * Cout * (Cin * 9 + rC) * H * W, "vf_conv_small_res", _ptr(x), _ptr(wd), _ptr(layer.bias),
            _ptr(view_bias), _ptr(y), S, Cin, Cout, H, W, _ptr(rx), _ptr(rx2), rC1, rC, _ptr(rwd), _ptr(res_layer.bias),
            _stream(), tag=(Cin, Cout, H, 3, 0))
    return y


class _Conv2dFn(torch.autograd.Function):
    """tap (stride-2 convs of the encoder): also return a handle on the INPUT x for the decoder's skip connection,
    whose gradient is then added in the dgrad kernel's epilogue instead of by an autograd add (as _GroupNormSkipFn
    does for the residual blocks)."""

    @staticmethod
    def forward(ctx, x, weight, bias, view_bias, residual, layer, mode, training, twin, tap=False):
        _check(x, bias, view_bias, residual)
        ctx.tap = tap
        if tap:
            ctx.set_materialize_grads(False)      # an unused handle must not cost a zero tensor
        S, Cin, Hi, Wi = x.shape
        Cout, _, KS, _ = weight.shape
        m = _MODES[mode]
        H, W = (Hi // 2, Wi // 2) if m == 1 else ((Hi * 2, Wi * 2) if m == 2 else (Hi, Wi))
        y = torch.empty(S, Cout, H, W, device=x.device, dtype=torch.float32)
        flops = 2.0 * S * Cout * Cin * KS * KS * H * W
        # algorithmic HBM bytes: input, output (+ residual) and the weights, each once
        nb = 4.0 * (x.numel() + y.numel() + weight.numel() + (residual.numel() if residual is not None else 0))
        wino = wino_kind(S, Cin, Cout, H, W, KS, m, train=bool(training))   # (grad mode is off inside Function.forward)
        ctx.b3 = _use_b3(KS, m, H * W)
        if ctx.b3:
            wf, wb = _packed_b3(layer, force=training)
            _launch("conv_fwd", flops, "vf_conv1x1_bf16x3", _ptr(x), None, 0, ctypes.c_void_p(wf.data_ptr()), _ptr(bias),
                    _ptr(view_bias), _ptr(residual), _ptr(y), None, 0, S, Cin, Cout, H * W, _stream(),
                    tag=(Cin, Cout, H, KS, m))
        elif wino:
            wf, wb = _packed_wino(layer, training, wino)
            ws, nws = _wino_ws(x.device, S, Cin, Cout, H, W, wino)
            _launch("conv_fwd", flops, _WINO_ABI[wino][3], _ptr(x), _ptr(wf), _ptr(bias), _ptr(view_bias),
                    _ptr(residual), _ptr(y), _ptr(ws), nws, S, Cin, Cout, H, W, m, _stream(),
                    tag=(Cin, Cout, H, KS, m), nbytes=nb)
        else:
            wf, wb = _packed(layer, force=training)
            ws, nws = _conv_ws(x.device, S, Cin, Cout, H, W, KS)
            _launch("conv_fwd", flops, "vf_conv_fwd", _ptr(x), _ptr(wf), _ptr(bias), _ptr(view_bias),
                    _ptr(residual), _ptr(y), _ptr(ws), nws, S, Cin, Cout, H, W, KS, m, _stream(),
                    tag=(Cin, Cout, H, KS, m))
        ctx.flops, ctx.tag = flops, (Cin, Cout, H, KS, m)
        ctx.save_for_backward(x)
        ctx.wino = wino
        ctx.wb, ctx.m, ctx.KS, ctx.Cout = wb, m, KS, Cout
        ctx.has = (bias is not None, view_bias is not None, residual is not None)
        ctx.pw, ctx.pb, ctx.twin = weight, bias, (twin.bias if twin is not None else None)
        return (y, x.view_as(x)) if tap else y

    @staticmethod
    def backward(ctx, dy, dtap=None):
        (x,) = ctx.saved_tensors
        dy = _c(dy)
        dtap = _c(dtap) if dtap is not None else None
        S, Cin, Hi, Wi = x.shape
        _, Cout, H, W = dy.shape
        KS, m = ctx.KS, ctx.m
        strm = _stream()
        dx = dw = db = dvb = dres = None
        if ctx.needs_input_grad[0] and ctx.wino:      # Winograd dgrad (dy and dx have the conv's output size)
            dfull = torch.empty(S, Cin, H, W, device=x.device, dtype=torch.float32)
            ws, nws = _wino_ws(x.device, S, Cout, Cin, H, W, ctx.wino)
            _launch("conv_dgrad", ctx.flops, _WINO_ABI[ctx.wino][3], _ptr(dy), _ptr(ctx.wb), None, None, None,
                    _ptr(dfull), _ptr(ws), nws, S, Cout, Cin, H, W, 0, strm, tag=ctx.tag,
                    nbytes=4.0 * (dy.numel() + dfull.numel() + ctx.pw.numel()))
            if m == 2:                                 # upsample + conv: 2x2 sum-pool back to the source size
                dx = torch.empty_like(x)
                _call("vf_sumpool2", _ptr(dfull), _ptr(dx), dx.numel(), Wi, strm)
            else:
                dx = dfull
        elif ctx.needs_input_grad[0] and ctx.b3:
            dx = torch.empty_like(x)
            _launch("conv_dgrad", ctx.flops, "vf_conv1x1_bf16x3", _ptr(dy), None, 0, ctypes.c_void_p(ctx.wb.data_ptr()), None,
                    None, None, _ptr(dx), None, 0, S, Cout, Cin, H * W, strm, tag=ctx.tag)
        elif ctx.needs_input_grad[0]:
            if m == 0:
                dx = torch.empty_like(x)
                ws, nws = _conv_ws(x.device, S, Cout, Cin, H, W, KS)
                _launch("conv_dgrad", ctx.flops, "vf_conv_fwd", _ptr(dy), _ptr(ctx.wb), None, None, None, _ptr(dx),
                        _ptr(ws), nws, S, Cout, Cin, H, W, KS, 0, strm, tag=ctx.tag)
            elif m == 1:      # stride-2 conv: sub-pixel transposed conv (each output parity gets its own taps)
                dx = torch.empty_like(x)
                _launch("conv_dgrad", ctx.flops, "vf_conv_fwd", _ptr(dy), _ptr(ctx.wb), None, None, _ptr(dtap), _ptr(dx),
                        None, 0, S, Cout, Cin, H, W, KS, 4, strm, tag=ctx.tag)
                dtap = None
            else:             # upsample + conv: dgrad at the upsampled size, then 2x2 sum-pool
                dup = torch.empty(S, Cin, H, W, device=x.device, dtype=torch.float32)
                _launch("conv_dgrad", ctx.flops, "vf_conv_fwd", _ptr(dy), _ptr(ctx.wb), None, None, None, _ptr(dup),
                        None, 0, S, Cout, Cin, H, W, KS, 0, strm, tag=ctx.tag)
                dx = torch.empty_like(x)
                _call("vf_sumpool2", _ptr(dup), _ptr(dx), dx.numel(), Wi, strm)
        hb, hv, hr = ctx.has
        want_b, want_v = hb and ctx.needs_input_grad[2], hv and ctx.needs_input_grad[3]
        arena = reducer.ACTIVE is not None
        hit = _rowsum_get(dy) if (want_b or want_v) else None
        if hit is not None:
            dvb, db = hit[1], (hit[2].view_as(hit[2]) if hit[2] is not None else None)   # (a fresh object: see _fresh)
        db2 = None            # this dY's channel sums for the residual 1x1 conv, in a tensor of its own
        if ctx.needs_input_grad[1] and use_winograd_wgrad(S, Cin, Cout, H, W, KS, m):
            need = _lib.load().vf_wino_wgrad_ws_floats(S, Cin, Cout, H, W)
            ws = _workspace(x.device, need)
            dw = _gout(ctx.pw, Cout, Cin, 3, 3, like=x)
            db_here = None
            if want_b and db is None:
                # the wgrad kernel reads every dY tile anyway: the bias gradient (sum over views and pixels) rides
                # along; with a residual branch its 1x1 conv has the same bias gradient and gets its own copy
                db = db_here = _gout(ctx.pb, Cout, like=x)
                if hr:
                    db2 = _gout(ctx.twin, Cout, like=x)
            owners = [ctx.pw] + ([ctx.pb] if db_here is not None else []) + \
                     ([ctx.twin] if (db2 is not None and ctx.twin is not None) else [])   # (identity residual: nobody owns db2)
            ws_own = _wred_ws(x.device, need) if (st.WRED_DEFER and _defer_begin(owners, st._CAPTURE_TABLE_W)) else None
            if ws_own is not None:
                # main kernel only, into this layer's slice of the slab arena; the slab sum joins the pass's one multi launch
                ws = ws_own
                row, nblk = (ctypes.c_longlong * 9)(), ctypes.c_int(0)
                _launch("conv_wgrad", ctx.flops, "vf_wino_wgrad_main", _ptr(x), _ptr(dy), _ptr(dw), _ptr(db_here), _ptr(db2),
                        _ptr(ws), ws.numel(), S, Cin, Cout, H, W, m, ctypes.cast(row, ctypes.c_void_p),
                        ctypes.cast(ctypes.pointer(nblk), ctypes.c_void_p), strm, tag=ctx.tag)
                # (x and dy are not kept: the main kernel has consumed them in stream order; the flush reads ws only)
                st._PENDING_WRED.append((list(row), nblk.value, (ws, dw, db_here, db2)))
                # AccumulateGrad adopts an incoming gradient only while nobody else references that tensor OBJECT; any
                # other reference (the entry above; a view's ._base) makes it clone the -- still unfilled -- tensor.  So
                # autograd gets fresh views (as the GroupNorm sums do with dgb[0] / dgb[1]); db2 reaches the residual conv
                # through _rowsum_put and is re-viewed there.
                dw = dw.view_as(dw)
                if db_here is not None:
                    db = db_here.view_as(db_here)
            else:
                _launch("conv_wgrad", ctx.flops, "vf_wino_wgrad", _ptr(x), _ptr(dy), _ptr(dw), _ptr(db_here), _ptr(db2),
                        _ptr(ws), ws.numel(), S, Cin, Cout, H, W, m, strm, tag=ctx.tag)
        elif ctx.needs_input_grad[1]:
            need = _lib.load().vf_conv_wgrad_ws_floats(S, Cin, Cout, H, W, KS)
            dw = _gout(ctx.pw, Cout, Cin, KS, KS, like=x)
            # (round 6) the direct / 1x1 kernels' slab sums join the pass's deferred multi launch too
            ws_own = _wred_ws(x.device, need) if (st.WRED_DEFER and st.WRED_DEFER_GENERIC
                                                 and _defer_begin([ctx.pw], st._CAPTURE_TABLE_W)) else None
            if ws_own is not None:
                row, nblk = (ctypes.c_longlong * 9)(), ctypes.c_int(0)
                _launch("conv_wgrad", ctx.flops, "vf_conv_wgrad_main", _ptr(x), _ptr(dy), _ptr(dw), _ptr(ws_own), ws_own.numel(),
                        S, Cin, Cout, H, W, KS, m, ctypes.cast(row, ctypes.c_void_p),
                        ctypes.cast(ctypes.pointer(nblk), ctypes.c_void_p), strm, tag=ctx.tag)
                st._PENDING_WRED.append((list(row), nblk.value, (ws_own, dw, None, None)))
                dw = dw.view_as(dw)                   # (a fresh object for autograd: see the Winograd branch above)
            else:
                ws = _workspace(x.device, need)
                _launch("conv_wgrad", ctx.flops, "vf_conv_wgrad", _ptr(x), _ptr(dy), _ptr(dw), _ptr(ws), ws.numel(), S,
                        Cin, Cout, H, W, KS, m, strm, tag=ctx.tag)
        if want_b or want_v:
            if (want_v and dvb is None) or (want_b and db is None and dvb is None):
                if Cout >= 192:                  # one launch, one workgroup per channel (enough channels to fill the chip)
                    db_new = _gout(ctx.pb, Cout, like=x) if (want_b and db is None) else None
                    dvb = torch.empty(S, Cout, device=x.device, dtype=torch.float32) if want_v else None
                    _call("vf_bias_grad", _ptr(dy), _ptr(db_new), _ptr(dvb), S, Cout, H * W, strm)
                    db = db_new if db_new is not None else db
                else:                            # few channels: wave-per-row partial sums, then the column sum
                    dvb = torch.empty(S, Cout, device=x.device, dtype=torch.float32)
                    _call("vf_rowsum", _ptr(dy), _ptr(dvb), S * Cout, H * W, strm)
            if want_b and db is None:
                db = _gout(ctx.pb, Cout, like=x)
                _call("vf_colsum", _ptr(dvb), _ptr(db), 1, S, Cout, strm)
            if hr and want_b:                    # the residual branch (1x1 conv) receives this very dY
                # a tensor living in the gradient arena is all-reduced in place as soon as its segment is complete:
                # it must never be handed to a second layer (which gets db2, or re-derives db from the row sums)
                _rowsum_put(dy, dvb, db2 if (db2 is not None or arena) else db)
            if not want_b:
                db = None
            if not want_v:
                dvb = None
        if hr and ctx.needs_input_grad[4]:
            dres = dy
        if dtap is not None:                          # (a path without the fused epilogue)
            dx = dtap if dx is None else dx + dtap
        return dx, dw, db, dvb, dres, None, None, None, None, None


class _Conv1x1CatFn(torch.autograd.Function):
    """1x1 conv (bias only) on the never-materialised channel concatenation [x1 | x2]: the residual conv of the
    decoder blocks (reference unet.py:134, 238)."""

    @staticmethod
    def forward(ctx, x1, x2, weight, bias, layer, training):
        _check(x1, x2, bias)
        S, C1, H, W = x1.shape
        Cout, Cin = weight.shape[0], weight.shape[1]
        y = torch.empty(S, Cout, H, W, device=x1.device, dtype=torch.float32)
        ctx.b3 = _use_b3(1, 0, H * W)
        if ctx.b3:
            wf, wb = _packed_b3(layer, force=training)
            _launch("conv_fwd", 2.0 * S * Cout * Cin * H * W, "vf_conv1x1_bf16x3", _ptr(x1), _ptr(x2), C1,
                    ctypes.c_void_p(wf.data_ptr()), _ptr(bias), None, None, _ptr(y), None, 0, S, Cin, Cout, H * W, _stream(),
                    tag=(Cin, Cout, H, 1, 0))
        else:
            wf, wb = _packed(layer, force=training)
            ws, nws = _conv_ws(x1.device, S, Cin, Cout, H, W, 1)
            _launch("conv_fwd", 2.0 * S * Cout * Cin * H * W, "vf_conv1x1_cat_fwd", _ptr(x1), _ptr(x2), C1, _ptr(wf),
                    _ptr(bias), _ptr(y), _ptr(ws), nws, S, Cin, Cout, H, W, _stream(), tag=(Cin, Cout, H, 1, 0))
        ctx.save_for_backward(x1, x2)
        ctx.wb, ctx.dims, ctx.has_bias = wb, (Cin, Cout), bias is not None
        ctx.pw, ctx.pb = weight, bias
        return y

    @staticmethod
    def backward(ctx, dy):
        x1, x2 = ctx.saved_tensors
        dy = _c(dy)
        S, C1, H, W = x1.shape
        Cin, Cout = ctx.dims
        strm = _stream()
        flops, tag = 2.0 * S * Cout * Cin * H * W, (Cin, Cout, H, 1, 0)
        dx1 = dx2 = dw = db = None
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[1]:
            dx1, dx2 = torch.empty_like(x1), torch.empty_like(x2)
            if ctx.b3:
                _launch("conv_dgrad", flops, "vf_conv1x1_bf16x3", _ptr(dy), None, 0, ctypes.c_void_p(ctx.wb.data_ptr()), None,
                        None, None, _ptr(dx1), _ptr(dx2), C1, S, Cout, Cin, H * W, strm, tag=tag)
            else:
                _launch("conv_dgrad", flops, "vf_conv1x1_cat_dgrad", _ptr(dy), _ptr(ctx.wb), _ptr(dx1), _ptr(dx2), C1, S,
                        Cin, Cout, H, W, strm, tag=tag)
        if ctx.needs_input_grad[2]:
            need = _lib.load().vf_conv_wgrad_ws_floats(S, Cin, Cout, H, W, 1)
            dw = _gout(ctx.pw, Cout, Cin, 1, 1, like=x1)
            ws_own = _wred_ws(x1.device, need) if (st.WRED_DEFER and st.WRED_DEFER_GENERIC
                                                  and _defer_begin([ctx.pw], st._CAPTURE_TABLE_W)) else None
            if ws_own is not None:                    # slab sum deferred to the pass's one multi launch (see _Conv2dFn)
                row, nblk = (ctypes.c_longlong * 9)(), ctypes.c_int(0)
                _launch("conv_wgrad", flops, "vf_conv1x1_cat_wgrad_main", _ptr(x1), _ptr(x2), C1, _ptr(dy), _ptr(dw),
                        _ptr(ws_own), ws_own.numel(), S, Cin, Cout, H, W, ctypes.cast(row, ctypes.c_void_p),
                        ctypes.cast(ctypes.pointer(nblk), ctypes.c_void_p), strm, tag=tag)
                st._PENDING_WRED.append((list(row), nblk.value, (ws_own, dw, None, None)))
                dw = dw.view_as(dw)
            else:
                ws = _workspace(x1.device, need)
                _launch("conv_wgrad", flops, "vf_conv1x1_cat_wgrad", _ptr(x1), _ptr(x2), C1, _ptr(dy), _ptr(dw), _ptr(ws),
                        ws.numel(), S, Cin, Cout, H, W, strm, tag=tag)
        if ctx.has_bias and ctx.needs_input_grad[3]:
            hit = _rowsum_get(dy)                     # the 3x3 conv this output is added to has summed this dY
            db = hit[2].view_as(hit[2]) if (hit is not None and hit[2] is not None) else None     # (a fresh object: see _fresh)
            if db is None:
                dvb = hit[1] if hit is not None else None
                if dvb is None:
                    dvb = torch.empty(S, Cout, device=x1.device, dtype=torch.float32)
                    _call("vf_rowsum", _ptr(dy), _ptr(dvb), S * Cout, H * W, strm)
                db = _gout(ctx.pb, Cout, like=x1)
                _call("vf_colsum", _ptr(dvb), _ptr(db), 1, S, Cout, strm)
        return dx1, dx2, dw, db, None, None


def conv1x1_cat(x1, x2, layer):
    """layer(cat(x1, x2)) for a 1x1 `layer` with bias, without building the concatenation."""
    if not torch.is_grad_enabled():
        S, C1, H, W = x1.shape
        Cout, Cin = layer.weight.shape[0], layer.weight.shape[1]
        if use_small_conv(S, Cin, Cout, H, W, 1, 0):
            _check(x1, x2, layer.bias)
            return _conv_small(x1, x2, layer.weight, layer.bias, None, None, S, Cin, Cout, H, W, 1, 0)
    training = torch.is_grad_enabled() and layer.weight.requires_grad
    return _Conv1x1CatFn.apply(x1, x2, layer.weight, layer.bias, layer, training)


def conv2d(x, layer, view_bias=None, residual=None, mode="same", twin=None, tap=False, res_fold=None):
    """3x3 (pad 1) or 1x1 convolution with the parameters of `layer` (an nn.Conv2d holder).
    tap=True: returns (y, x') with x' a handle on x for a second consumer (see _Conv2dFn).

    mode "same": stride 1; "down2": stride 2; "up2": nearest x2 upsample fused into the load.
    Epilogue adds bias[c] + view_bias[s,c] + residual.  twin: the 1x1 conv layer that produced `residual` (it has
    the same bias gradient, which this layer's weight-gradient kernel then writes for both).
    """
    if not torch.is_grad_enabled():
        S, Cin, Hi, Wi = x.shape
        Cout, _, KS, _ = layer.weight.shape
        m = _MODES[mode]
        H, W = (Hi // 2, Wi // 2) if m == 1 else ((Hi * 2, Wi * 2) if m == 2 else (Hi, Wi))
        if res_fold is not None:          # (res_layer, rx, rx2 | None): only offered where can_fold_residual() said so
            assert residual is None and not tap
            return _conv_small_res(x, layer, view_bias, *res_fold)
        if use_small_conv(S, Cin, Cout, H, W, KS, m):
            _check(x, layer.bias, view_bias, residual)
            y = _conv_small(x, None, layer.weight, layer.bias, view_bias, residual, S, Cin, Cout, H, W, KS, m)
            return (y, x) if tap else y
    assert res_fold is None
    training = torch.is_grad_enabled() and layer.weight.requires_grad
    if tap and torch.is_grad_enabled():
        return _Conv2dFn.apply(x, layer.weight, layer.bias, view_bias, residual, layer, mode, training, twin, True)
    y = _Conv2dFn.apply(x, layer.weight, layer.bias, view_bias, residual, layer, mode, training, twin)
    return (y, x) if tap else y


def conv2d_gn(x, layer, gn, groups, silu, view_bias=None, residual=None, mode="same", want_y=False, res_fold=None):
    """Inference only (no autograd): (y | None, a) with y = conv2d(x, layer, view_bias, residual, mode) and
    a = [Swish](GroupNorm(gn.weight, gn.bias, groups)(y)).  Where the conv runs split-K (small S: the sampler) the
    GroupNorm is evaluated by the conv's reduce launch -- one kernel instead of two, and y is only written if `want_y`;
    elsewhere (Winograd path, large grids) it is the two separate ops."""
    S, Cin, Hi, Wi = x.shape
    Cout, _, KS, _ = layer.weight.shape
    m = _MODES[mode]
    H, W = (Hi // 2, Wi // 2) if m == 1 else ((Hi * 2, Wi * 2) if m == 2 else (Hi, Wi))
    lib = _lib.load()
    fused = (not torch.is_grad_enabled() and not use_winograd(S, Cin, Cout, H, W, KS, m, False) and not _use_b3(KS, m, H * W)
             and not use_small_conv(S, Cin, Cout, H, W, KS, m)      # (one conv launch + the GroupNorm launch instead)
             and lib.vf_conv_fwd_ws_floats(S, Cin, Cout, H, W, KS) > 0)
    # Winograd route at a few views (the sampler at N = 2 ... 16): every tile of the launch is a K-split tail tile, and the
    # fix-up launch that sums the partials normalises too (vf_wino_conv_fwd_gn, round 5): conv + fix-up/GroupNorm instead
    # of conv + fix-up + GroupNorm
    if (st.WINO_GN_FUSION and not fused and not torch.is_grad_enabled() and res_fold is None
            and m in (0, 2) and KS == 3 and wino_kind(S, Cin, Cout, H, W, KS, m, False) == 1
            and lib.vf_wino_conv_gn_fusable(S, Cin, Cout, H, W, m, groups)):
        _check(x, layer.bias, view_bias, residual, gn.weight, gn.bias)
        y = torch.empty(S, Cout, H, W, device=x.device, dtype=torch.float32) if want_y else None
        a = torch.empty(S, Cout, H, W, device=x.device, dtype=torch.float32)
        wf, _ = _packed_wino(layer, False, 1)
        ws, nws = _wino_ws(x.device, S, Cin, Cout, H, W, 1)
        _launch("conv_fwd", 2.0 * S * Cout * Cin * 9 * H * W, "vf_wino_conv_fwd_gn", _ptr(x), _ptr(wf), _ptr(layer.bias),
                _ptr(view_bias), _ptr(residual), _ptr(y), int(want_y), _ptr(gn.weight), _ptr(gn.bias), _ptr(a), groups, 1e-5,
                int(silu), _ptr(ws), nws, S, Cin, Cout, H, W, m, _stream(), tag=(Cin, Cout, H, KS, m))
        return y, a
    if not fused:
        y = conv2d(x, layer, view_bias=view_bias, residual=residual, mode=mode, res_fold=res_fold)
        return y, group_norm(y, gn.weight, gn.bias, groups, silu)
    assert res_fold is None
    _check(x, layer.bias, view_bias, residual, gn.weight, gn.bias)
    y = torch.empty(S, Cout, H, W, device=x.device, dtype=torch.float32)
    a = torch.empty_like(y)
    stats = torch.empty(2 * S * groups, device=x.device, dtype=torch.float32)
    wf, _ = _packed(layer, force=False)
    ws, nws = _conv_ws(x.device, S, Cin, Cout, H, W, KS)
    _call("vf_conv_fwd_gn", _ptr(x), _ptr(wf), _ptr(layer.bias), _ptr(view_bias), _ptr(residual), _ptr(y), int(want_y),
              _ptr(gn.weight), _ptr(gn.bias), _ptr(a), _ptr(stats), groups, 1e-5, int(silu), _ptr(ws), nws, S, Cin, Cout,
              H, W, KS, m, _stream())
    return (y if want_y else None), a
