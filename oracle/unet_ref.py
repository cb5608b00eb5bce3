"""Oracle: functional fp32 CPU restatement of the reference UNet denoiser.

TEST INFRASTRUCTURE -- see oracle/__init__.py.  Follows (by behaviour, not by
text) /root/reference/model/unet.py:

  * topology / channel bookkeeping ........ unet.py:9-112
  * forward order (embed, downs, mid, ups) . unet.py:114-138
  * sinusoidal encoding ................... unet.py:142-157
  * per-(sample,channel) embedding add .... unet.py:160-177
  * x*sigmoid(x) .......................... unet.py:180-182
  * nearest x2 + conv / stride-2 conv ..... unet.py:185-201
  * GN -> swish -> conv3x3 ................ unet.py:207-218
  * residual block ........................ unet.py:221-245
  * single-head spatial self-attention .... unet.py:248-277

The network is expressed as a flat list of steps computed from the
hyper-parameters and evaluated directly on a `state_dict` (same key names as
the reference), so it works with weights saved by the reference.
"""
import math

import torch
import torch.nn.functional as F


def unet_topology(in_channel=6, out_channel=3, inner_channel=32, norm_groups=32,
                  channel_mults=(1, 2, 4, 8, 8), attn_res=(8,), res_blocks=3,
                  image_size=128, **_unused):
    """Return the ordered list of layer descriptors for downs / mid / ups.

    Each descriptor is a dict with `kind` in {"stem","res","down","up"},
    its `state_dict` prefix, channel counts and whether attention follows.
    """
    downs, mid, ups = [], [], []
    ch = inner_channel
    skip = [ch]
    res = image_size
    downs.append(dict(kind="stem", key="downs.0", cin=in_channel, cout=ch))
    idx = 1
    nlev = len(channel_mults)
    for lvl, mult in enumerate(channel_mults):
        width = inner_channel * mult
        for _ in range(res_blocks):
            downs.append(dict(kind="res", key=f"downs.{idx}", cin=ch, cout=width,
                              attn=res in attn_res))
            idx += 1
            ch = width
            skip.append(ch)
        if lvl != nlev - 1:
            downs.append(dict(kind="down", key=f"downs.{idx}", cin=ch, cout=ch))
            idx += 1
            skip.append(ch)
            res //= 2
    mid.append(dict(kind="res", key="mid.0", cin=ch, cout=ch, attn=True))
    mid.append(dict(kind="res", key="mid.1", cin=ch, cout=ch, attn=False))
    idx = 0
    for lvl in reversed(range(nlev)):
        width = inner_channel * channel_mults[lvl]
        for _ in range(res_blocks + 1):
            ups.append(dict(kind="res", key=f"ups.{idx}", cin=ch + skip.pop(), cout=width,
                            attn=res in attn_res, skip=True))
            idx += 1
            ch = width
        if lvl >= 1:
            ups.append(dict(kind="up", key=f"ups.{idx}", cin=ch, cout=ch))
            idx += 1
            res *= 2
    head = dict(kind="head", key="final_conv", cin=ch,
                cout=out_channel if out_channel is not None else in_channel)
    return dict(downs=downs, mid=mid, ups=ups, head=head, groups=norm_groups,
                emb_dim=inner_channel)


def swish(x):
    return x * torch.sigmoid(x)


def sincos_encoding(level, dim):
    """level: (S,1) -> (S,1,dim); dim//2 frequencies exp(-ln(1e4)*k/(dim//2))."""
    half = dim // 2
    k = torch.arange(half, dtype=level.dtype, device=level.device) / half
    arg = level.unsqueeze(1) * torch.exp(-math.log(1e4) * k.unsqueeze(0))
    return torch.cat([torch.sin(arg), torch.cos(arg)], dim=-1)


def _gn_swish_conv(sd, key, x, groups, drop=None):
    h = F.group_norm(x, groups, sd[f"{key}.block.0.weight"], sd[f"{key}.block.0.bias"], eps=1e-5)
    h = swish(h)
    if drop is not None:                 # nn.Dropout(p), training mode, with explicit uniform draws u (unet.py:207-216)
        p, u = drop
        h = h * (u >= p).to(h.dtype) / (1.0 - p)
    return F.conv2d(h, sd[f"{key}.block.3.weight"], sd[f"{key}.block.3.bias"], padding=1)


def _res_block(sd, key, x, emb, groups, drop=None):
    rb = f"{key}.res_block"
    h = _gn_swish_conv(sd, f"{rb}.block1", x, groups)
    e = F.linear(emb, sd[f"{rb}.noise_func.noise_func.0.weight"],
                 sd[f"{rb}.noise_func.noise_func.0.bias"])          # (S,1,Cout)
    h = h + e.reshape(x.shape[0], -1, 1, 1)
    h = _gn_swish_conv(sd, f"{rb}.block2", h, groups, drop)          # the reference puts Dropout in block2 only
    if f"{rb}.res_conv.weight" in sd:
        x = F.conv2d(x, sd[f"{rb}.res_conv.weight"], sd[f"{rb}.res_conv.bias"])
    return h + x


def _self_attention(sd, key, x, groups):
    a = f"{key}.attn"
    S, C, H, W = x.shape
    n = F.group_norm(x, groups, sd[f"{a}.norm.weight"], sd[f"{a}.norm.bias"], eps=1e-5)
    qkv = F.conv2d(n, sd[f"{a}.qkv.weight"])                        # no bias
    q, k, v = qkv.reshape(S, 3, C, H * W).unbind(1)                 # each (S,C,L)
    score = torch.bmm(q.transpose(1, 2), k) / math.sqrt(C)          # (S,Lq,Lk)
    p = torch.softmax(score, dim=-1)
    o = torch.bmm(v, p.transpose(1, 2)).reshape(S, C, H, W)         # (S,C,Lq)
    o = F.conv2d(o, sd[f"{a}.out.weight"], sd[f"{a}.out.bias"])
    return o + x


def unet_forward(sd, hp, x, angle, level, dropout_u=None):
    """sd: state_dict of the UNet (no prefix); hp: hyper-parameter dict;
    x (S,Cin,H,W); angle (S,1); level (S,1)  ->  (S,Cout,H,W).
    dropout_u: None = eval mode (Dropout is the identity); else the uniform draws of the residual blocks'
    Dropout(p = hp["dropout"]) layers in execution order (training mode with explicit randomness)."""
    topo = unet_topology(**hp)
    p_drop = float(hp.get("dropout", 0) or 0)
    draws = iter(dropout_u) if (dropout_u is not None and p_drop > 0) else None
    g = topo["groups"]
    half = topo["emb_dim"] // 2
    emb = torch.cat([sincos_encoding(level, half), sincos_encoding(angle, half)], dim=-1)
    emb = F.linear(emb, sd["noise_level_mlp.0.weight"], sd["noise_level_mlp.0.bias"])
    emb = F.linear(swish(emb), sd["noise_level_mlp.2.weight"], sd["noise_level_mlp.2.bias"])

    def run(layer, x):
        kind, key = layer["kind"], layer["key"]
        if kind == "stem":
            return F.conv2d(x, sd[f"{key}.weight"], sd[f"{key}.bias"], padding=1)
        if kind == "down":
            return F.conv2d(x, sd[f"{key}.conv.weight"], sd[f"{key}.conv.bias"], stride=2, padding=1)
        if kind == "up":
            x = F.interpolate(x, scale_factor=2, mode="nearest")
            return F.conv2d(x, sd[f"{key}.conv.weight"], sd[f"{key}.conv.bias"], padding=1)
        x = _res_block(sd, key, x, emb, g, (p_drop, next(draws)) if draws is not None else None)
        if layer["attn"]:
            x = _self_attention(sd, key, x, g)
        return x

    feats = []
    for layer in topo["downs"]:
        x = run(layer, x)
        feats.append(x)
    for layer in topo["mid"]:
        x = run(layer, x)
    for layer in topo["ups"]:
        if layer.get("skip"):
            x = torch.cat((x, feats.pop()), dim=1)
        x = run(layer, x)
    return _gn_swish_conv(sd, "final_conv", x, g)
