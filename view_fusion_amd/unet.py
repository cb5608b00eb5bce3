"""MI355X-native UNet denoiser: host-side mirror of the reference `UNet` interface.

Same constructor keywords, same `forward(x, angle, time)` and the same `state_dict` keys /
shapes / registration order as /root/reference/model/unet.py:8-138, so reference checkpoints
load unchanged.  The torch.nn layer objects below are PARAMETER HOLDERS only (they give the
reference's names and default initialisation); their own `forward` is never called.  All
arithmetic is dispatched to the hand-written gfx950 kernels through `view_fusion_amd.ops`
(C-ABI in include/vf_hip.h).  There is no CPU / eager fallback: without the HIP library or
with CPU tensors the forward raises.

Dataflow per residual block (reference unet.py:221-245):
    a1 = swish(GN(x)) ............ vf_gn_fwd (one read, one write)
    h  = conv3x3(a1) + b + Linear(t)[s,c] ........ vf_conv_fwd (epilogue: bias + per-view bias)
    a2 = swish(GN(h))
    y  = conv3x3(a2) + b + (x | conv1x1(x)) ...... vf_conv_fwd (epilogue: bias + residual)
"""
import torch
from torch import nn


class _Named(nn.Module):
    """Container that registers children under explicit (reference) names."""

    def __init__(self, **children):
        super().__init__()
        for name, mod in children.items():
            self.add_module(name, mod)

    def __getitem__(self, name):
        return self._modules[name]


def _norm_act_conv(cin, cout, groups):
    # reference Block: Sequential(GroupNorm, Swish, Identity|Dropout, Conv2d) -> slots "0" and "3"
    return _Named(block=_Named(**{"0": nn.GroupNorm(groups, cin), "3": nn.Conv2d(cin, cout, 3, padding=1)}))


class _ResBlock(nn.Module):
    def __init__(self, cin, cout, emb_dim, groups, dropout=0):
        super().__init__()
        self.noise_func = _Named(noise_func=_Named(**{"0": nn.Linear(emb_dim, cout)}))
        self.block1 = _norm_act_conv(cin, cout, groups)
        self.block2 = _norm_act_conv(cout, cout, groups)
        self.res_conv = nn.Conv2d(cin, cout, 1) if cin != cout else nn.Identity()
        self.groups = groups
        self.dropout = float(dropout)      # reference: nn.Dropout in block2 only (slot "2", no parameters), unet.py:236

    def _drop(self, a, draws):
        """block2's nn.Dropout between Swish and the conv: identity in eval mode, as in torch."""
        if self.dropout == 0 or not self.training:
            return a
        from . import ops
        return ops.dropout(a, self.dropout, next(draws) if draws is not None else None)

    def _forward_inference(self, x, e, skip, a_in, next_gn):
        """No-grad path (the sampler): the same arithmetic with as few launches as the sizes allow.
        a_in: GroupNorm1(x) if the producer of x already evaluated it (in its split-K reduce / Winograd fix-up launch,
        ops.conv2d_gn), else None.  next_gn = (GroupNorm holder, silu) of the consumer of this block's output, or None.
        -> (y, GroupNorm_next(y) | None).  The residual 1x1 conv rides in the block's last conv as extra K where that
        conv runs the one-launch kernel (ops.can_fold_residual)."""
        from . import ops
        b1, b2 = self.block1["block"], self.block2["block"]
        conv1, conv2 = b1["3"], b2["3"]
        S, _, H, W = x.shape
        Cout = conv2.weight.shape[0]
        if skip is not None:
            C1, C = x.shape[1], x.shape[1] + skip.shape[1]
            if not (isinstance(self.res_conv, nn.Conv2d) and ops.cat_fusable(C1, C, x.shape[2] * x.shape[3], self.groups)):
                x, skip = ops.concat_channels(x, skip), None
        fold = (self.res_conv, x, skip) if ops.can_fold_residual(S, Cout, H, W, self.res_conv) else None
        res = None
        if skip is not None:
            if fold is None:
                res = ops.conv1x1_cat(x, skip, self.res_conv)
            a, _, _ = ops.group_norm_cat_skip(x, skip, b1["0"].weight, b1["0"].bias, self.groups, silu=True)
        else:
            if isinstance(self.res_conv, nn.Identity):
                res = x
            elif fold is None:
                res = ops.conv2d(x, self.res_conv)
            a = a_in if a_in is not None else ops.group_norm(x, b1["0"].weight, b1["0"].bias, self.groups, silu=True)
        _, a2 = ops.conv2d_gn(a, conv1, b2["0"], self.groups, True, view_bias=e)
        if next_gn is None:
            return ops.conv2d(a2, conv2, residual=res, res_fold=fold), None
        return ops.conv2d_gn(a2, conv2, next_gn[0], self.groups, next_gn[1], residual=res, want_y=True, res_fold=fold)

    def forward(self, x, e, skip=None, tap=False, draws=None):
        """tap=True (encoder blocks): also return a handle on the INPUT x for the decoder's skip connection, whose
        gradient is then added inside this block's GroupNorm backward instead of by an autograd add.
        e = this block's FeatureWiseAffine Linear(emb), (S,Cout): all blocks' are computed in one grouped
        launch by UNet.forward.  skip (decoder blocks): the encoder feature map that the reference concatenates
        to x (unet.py:134); when the shapes allow, the concatenation is never built -- the first GroupNorm and the
        residual 1x1 conv read both tensors."""
        from . import ops
        b1, b2 = self.block1["block"], self.block2["block"]
        if skip is not None:
            C1, C = x.shape[1], x.shape[1] + skip.shape[1]
            if not (isinstance(self.res_conv, nn.Conv2d) and ops.cat_fusable(C1, C, x.shape[2] * x.shape[3], self.groups)):
                x, skip = ops.concat_channels(x, skip), None
        if skip is not None:
            a, x1, x2 = ops.group_norm_cat_skip(x, skip, b1["0"].weight, b1["0"].bias, self.groups, silu=True)
            h = ops.conv2d(a, b1["3"], view_bias=e)
            a = self._drop(ops.group_norm(h, b2["0"].weight, b2["0"].bias, self.groups, silu=True), draws)
            return ops.conv2d(a, b2["3"], residual=ops.conv1x1_cat(x1, x2, self.res_conv), twin=self.res_conv)
        # x feeds both the first GroupNorm and the residual branch: the GN op hands x back so that
        # the residual gradient is summed inside its backward kernel
        a, xs, xt = ops.group_norm_skip(x, b1["0"].weight, b1["0"].bias, self.groups, silu=True, tap=True)
        h = ops.conv2d(a, b1["3"], view_bias=e)
        a = self._drop(ops.group_norm(h, b2["0"].weight, b2["0"].bias, self.groups, silu=True), draws)
        twin = None if isinstance(self.res_conv, nn.Identity) else self.res_conv
        skip = xs if twin is None else ops.conv2d(xs, twin)
        y = ops.conv2d(a, b2["3"], residual=skip, twin=twin)
        return (y, xt) if tap else y


class _SelfAttention(nn.Module):
    def __init__(self, ch, groups):
        super().__init__()
        self.norm = nn.GroupNorm(groups, ch)
        self.qkv = nn.Conv2d(ch, ch * 3, 1, bias=False)
        self.out = nn.Conv2d(ch, ch, 1)
        self.groups = groups

    def forward(self, x, n=None):
        """n: GroupNorm(x) when the producer of x has already evaluated it (inference)."""
        from . import ops
        if n is None:
            n, xs = ops.group_norm_skip(x, self.norm.weight, self.norm.bias, self.groups, silu=False)
        else:
            xs = x
        qkv = ops.conv2d(n, self.qkv)                                               # (S,3C,H,W)
        o = ops.attention(qkv)                                                      # (S,C,H,W)
        return ops.conv2d(o, self.out, residual=xs)


class _ResAttnBlock(nn.Module):
    def __init__(self, cin, cout, emb_dim, groups, with_attn, dropout=0):
        super().__init__()
        self.with_attn = with_attn
        self.res_block = _ResBlock(cin, cout, emb_dim, groups, dropout)
        if with_attn:
            self.attn = _SelfAttention(cout, groups)

    def forward_inference(self, x, e, skip=None, a_in=None, next_gn=None):
        """No-grad path: -> (y, GroupNorm_next(y) | None); see _ResBlock._forward_inference."""
        if self.with_attn:
            y, n = self.res_block._forward_inference(x, e, skip, a_in, (self.attn.norm, False))
            return self.attn(y, n), None
        return self.res_block._forward_inference(x, e, skip, a_in, next_gn)

    def forward(self, x, e, skip=None, tap=False, draws=None):
        xt = None
        if tap:
            x, xt = self.res_block(x, e, skip, tap=True, draws=draws)
        else:
            x = self.res_block(x, e, skip, draws=draws)
        x = self.attn(x) if self.with_attn else x
        return (x, xt) if tap else x


class _Resample(nn.Module):
    def __init__(self, ch, up):
        super().__init__()
        self.up = up
        self.conv = nn.Conv2d(ch, ch, 3, padding=1) if up else nn.Conv2d(ch, ch, 3, 2, 1)

    def forward(self, x, tap=False):
        """tap (Downsample in the encoder): -> (y, x'), x' = the handle on x that the decoder's skip connection takes."""
        from . import ops
        return ops.conv2d(x, self.conv, mode="up2" if self.up else "down2", tap=tap)


class UNet(nn.Module):
    def __init__(self, in_channel=6, out_channel=3, inner_channel=32, norm_groups=32,
                 channel_mults=(1, 2, 4, 8, 8), attn_res=(8,), res_blocks=3, dropout=0,
                 with_noise_level_emb=True, image_size=128):
        super().__init__()
        if not with_noise_level_emb:
            # the reference's own constructor fails on this branch: it builds FeatureWiseAffine(None, ...) ->
            # nn.Linear(None, C) -> TypeError (unet.py:33-35, 52-58, 165); same error type here
            raise TypeError("with_noise_level_emb=False: the reference UNet cannot be constructed without the "
                            "noise-level embedding (nn.Linear(None, C) in FeatureWiseAffine, model/unet.py:165)")
        if not 0 <= dropout < 1:
            raise ValueError(f"dropout probability has to be in [0, 1), got {dropout}")
        self.inner_channel = inner_channel
        self.noise_level_mlp = _Named(**{"0": nn.Linear(inner_channel, inner_channel * 4),
                                         "2": nn.Linear(inner_channel * 4, inner_channel)})
        emb = inner_channel
        ch = inner_channel
        skips = [ch]
        res = image_size
        downs = [nn.Conv2d(in_channel, inner_channel, 3, padding=1)]
        last = len(channel_mults) - 1
        for lvl, mult in enumerate(channel_mults):
            width = inner_channel * mult
            for _ in range(res_blocks):
                downs.append(_ResAttnBlock(ch, width, emb, norm_groups, res in attn_res, dropout))
                ch = width
                skips.append(ch)
            if lvl != last:
                downs.append(_Resample(ch, up=False))
                skips.append(ch)
                res //= 2
        self.downs = nn.ModuleList(downs)
        self.mid = nn.ModuleList([_ResAttnBlock(ch, ch, emb, norm_groups, True, dropout),
                                  _ResAttnBlock(ch, ch, emb, norm_groups, False, dropout)])
        ups = []
        for lvl in reversed(range(len(channel_mults))):
            width = inner_channel * channel_mults[lvl]
            for _ in range(res_blocks + 1):
                ups.append(_ResAttnBlock(ch + skips.pop(), width, emb, norm_groups, res in attn_res, dropout))
                ch = width
            if lvl > 0:
                ups.append(_Resample(ch, up=True))
                res *= 2
        self.ups = nn.ModuleList(ups)
        self.final_conv = _norm_act_conv(ch, out_channel if out_channel is not None else in_channel,
                                         norm_groups)
        self.norm_groups = norm_groups
        self._annotate_geometry(image_size)

    def _annotate_geometry(self, image_size):
        """Record every conv layer's OUTPUT size and mode (`_vf_geom`) so that the per-step weight
        packing knows which layers take the Winograd format."""
        def mark(conv, res, mode="same"):
            object.__setattr__(conv, "_vf_geom", (res, mode))

        def mark_block(blk, res):
            rb = blk.res_block
            mark(rb.block1["block"]["3"], res)
            mark(rb.block2["block"]["3"], res)
            if isinstance(rb.res_conv, nn.Conv2d):
                mark(rb.res_conv, res)
            if blk.with_attn:
                mark(blk.attn.qkv, res)
                mark(blk.attn.out, res)

        res = image_size
        for layer in self.downs:
            if isinstance(layer, _ResAttnBlock):
                mark_block(layer, res)
            elif isinstance(layer, _Resample):
                res //= 2
                mark(layer.conv, res, "down2")
            else:
                mark(layer, res)
        for layer in self.mid:
            mark_block(layer, res)
        for layer in self.ups:
            if isinstance(layer, _ResAttnBlock):
                mark_block(layer, res)
            else:
                res *= 2
                mark(layer.conv, res, "up2")
        mark(self.final_conv["block"]["3"], res)

    def _affine_layers(self):
        lst = getattr(self, "_vf_affine", None)
        if lst is None:
            lst = [m.res_block.noise_func["noise_func"]["0"] for m in list(self.downs) + list(self.mid) + list(self.ups)
                   if isinstance(m, _ResAttnBlock)]
            object.__setattr__(self, "_vf_affine", lst)
        return lst

    def _has_dropout(self):
        return any(isinstance(m, _ResBlock) and m.dropout > 0 for m in self.modules())

    def _forward_inference(self, x, es):
        """The no-grad forward (sampler): same dataflow as below without the autograd handles; a residual block whose
        output goes straight into another residual block's first GroupNorm (or into the final one) lets its last conv
        evaluate that GroupNorm too."""
        from . import ops

        def gn1_of(layer):            # first GroupNorm of a residual block that consumes its input un-concatenated
            if not isinstance(layer, _ResAttnBlock):
                return None
            return (layer.res_block.block1["block"]["0"], True)

        feats, a_next = [], None
        downs = list(self.downs)
        for i, layer in enumerate(downs):
            if isinstance(layer, _ResAttnBlock):
                nxt = downs[i + 1] if i + 1 < len(downs) else self.mid[0]
                x, a_next = layer.forward_inference(x, next(es), a_in=a_next, next_gn=gn1_of(nxt))
            elif isinstance(layer, _Resample):
                x, a_next = layer(x), None
            else:
                x, a_next = ops.conv2d(x, layer), None
            feats.append(x)
        x, a_next = self.mid[0].forward_inference(x, next(es), a_in=a_next, next_gn=gn1_of(self.mid[1]))
        x, a_next = self.mid[1].forward_inference(x, next(es), a_in=a_next, next_gn=None)
        ups = list(self.ups)
        fc = self.final_conv["block"]
        for i, layer in enumerate(ups):
            if isinstance(layer, _ResAttnBlock):
                last = i + 1 == len(ups)
                x, a_next = layer.forward_inference(x, next(es), skip=feats.pop(), next_gn=(fc["0"], True) if last else None)
            else:
                x, a_next = layer(x), None
        a = a_next if a_next is not None else ops.group_norm(x, fc["0"].weight, fc["0"].bias, self.norm_groups, silu=True)
        return ops.conv2d(a, fc["3"])

    def forward(self, x, angle, time, dropout_u=None):
        """x (S,Cin,H,W), angle (S,1), time = noise level (S,1)  ->  (S,Cout,H,W).
        dropout_u (extra, default None = torch's device RNG): the uniform draws of the residual blocks' Dropout
        layers in execution order, for parity runs (only consulted when dropout > 0 and self.training)."""
        from . import ops
        draws = iter(dropout_u) if dropout_u is not None else None
        if torch.is_grad_enabled() and self.final_conv["block"]["3"].weight.requires_grad:
            ops.pack_all(self, x.shape[0])   # training: all 103 conv layers re-packed (one launch per format)
        mlp = self.noise_level_mlp
        inference = not torch.is_grad_enabled() and not (self.training and self._has_dropout())

        def embed():
            pe = ops.sincos_embedding(time, angle, self.inner_channel)                  # (S,inner)
            emb = ops.linear(pe, mlp["0"].weight, mlp["0"].bias)
            emb = ops.linear(ops.swish(emb), mlp["2"].weight, mlp["2"].bias)            # (S,inner)
            # FeatureWiseAffine of every residual block (unet.py:160-177) in one grouped launch
            return ops.time_affine_all(emb, self._affine_layers())

        if inference:                                   # (Dropout active: the general path below applies it)
            return self._forward_inference(x, iter(embed()))
        es = iter(embed())

        # feats[i] feeds the next encoder layer AND the decoder: where that next layer is a residual block or a
        # Downsample conv, the decoder takes that layer's handle on its input instead (see _ResBlock.forward, tap):
        # the two gradients of feats[i] are then summed inside that layer's backward kernel
        feats = []
        for layer in self.downs:
            if isinstance(layer, _ResAttnBlock):
                x, xin = layer(x, next(es), tap=True, draws=draws)
                if feats:
                    feats[-1] = xin
            elif isinstance(layer, _Resample):
                x, xin = layer(x, tap=True)
                if feats:
                    feats[-1] = xin
            else:
                x = ops.conv2d(x, layer)
            feats.append(x)
        first = True
        for layer in self.mid:
            if first:
                x, feats[-1] = layer(x, next(es), tap=True, draws=draws)
                first = False
            else:
                x = layer(x, next(es), draws=draws)
        for layer in self.ups:
            if isinstance(layer, _ResAttnBlock):
                x = layer(x, next(es), feats.pop(), draws=draws)
            else:
                x = layer(x)
        fc = self.final_conv["block"]
        a = ops.group_norm(x, fc["0"].weight, fc["0"].bias, self.norm_groups, silu=True)
        return ops.conv2d(a, fc["3"])
