"""CPU oracle for the Swin-T forward (BASELINE config 5) — TEST INFRASTRUCTURE ONLY.

The reference's Allen_data_Backbone/train.py:70-85 builds `SwinForImageClassification(SwinConfig(num_labels=5))`
from the `transformers` package (an un-vendored, un-pinned dependency of the reference). transformers 5.15.0 is
installed in the build container, so this functional fp32 torch-CPU restatement of
transformers/models/swin/modeling_swin.py (line numbers of 5.15.0 cited per function) is PINNED against that
package by oracle/make_golden_swin.py (max |diff| recorded in tests/golden/swin_*.npz). Nothing in the product
imports this file.
"""

import torch
import torch.nn.functional as F

DEFAULT_CFG = dict(image_size=224, patch_size=4, num_channels=3, embed_dim=96, depths=(2, 2, 6, 2),
                   num_heads=(3, 6, 12, 24), window_size=7, mlp_ratio=4.0, layer_norm_eps=1e-5, num_labels=5)


def relative_position_index(ws):
    """SwinRelativePositionBias._create_relative_position_index, modeling_swin.py:350-365."""
    coords = torch.stack(torch.meshgrid([torch.arange(ws), torch.arange(ws)], indexing="ij"))
    flat = torch.flatten(coords, 1)
    rel = (flat[:, :, None] - flat[:, None, :]).permute(1, 2, 0).contiguous()
    rel[:, :, 0] += ws - 1
    rel[:, :, 1] += ws - 1
    rel[:, :, 0] *= 2 * ws - 1
    return rel.sum(-1)


def window_partition(x, ws):
    """modeling_swin.py:486-495."""
    B, H, W, C = x.shape
    x = x.view(B, H // ws, ws, W // ws, ws, C)
    return x.transpose(2, 3).contiguous().view(-1, ws, ws, C)


def window_reverse(w, ws, H, W):
    """modeling_swin.py:498-505."""
    C = w.shape[-1]
    w = w.view(-1, H // ws, W // ws, ws, ws, C)
    return w.transpose(2, 3).contiguous().view(-1, H, W, C)


def shift_mask(H, W, ws, shift):
    """SwinLayer.get_attn_mask, modeling_swin.py:584-607 -> (nW, ws*ws, ws*ws) or None."""
    if shift <= 0:
        return None
    h, w = torch.arange(H), torch.arange(W)
    hr = (h >= H - ws).long() + (h >= H - shift).long()
    wr = (w >= W - ws).long() + (w >= W - shift).long()
    img = (hr[None, :, None, None] * 3 + wr[None, None, :, None]).to(torch.float32)
    mw = window_partition(img, ws).view(-1, ws * ws)
    m = mw.unsqueeze(1) - mw.unsqueeze(2)
    return m.masked_fill(m != 0, -100.0).masked_fill(m == 0, 0.0)


def swin_layer(sd, pre, x, H, W, heads, ws_cfg, shift_cfg, eps):
    """SwinLayer.forward, modeling_swin.py:529-574 (+ SwinAttention :418-468, SwinMLP :478-483)."""
    B, L, C = x.shape
    ws, shift = ws_cfg, shift_cfg
    if min(H, W) <= ws:  # set_shift_and_window_size :576-582
        ws, shift = min(H, W), 0
    shortcut = x
    y = F.layer_norm(x, (C,), sd[pre + "layernorm_before.weight"], sd[pre + "layernorm_before.bias"], eps)
    y = y.view(B, H, W, C)
    # maybe_pad (SwinLayer): zeros to the right / bottom AFTER layernorm_before, up to multiples of the window
    pad_r, pad_b = (ws - W % ws) % ws, (ws - H % ws) % ws
    Hp, Wp = H + pad_b, W + pad_r
    if pad_r or pad_b:
        y = F.pad(y, (0, 0, 0, pad_r, 0, pad_b))
    if shift > 0:
        y = torch.roll(y, shifts=(-shift, -shift), dims=(1, 2))
    win = window_partition(y, ws).view(-1, ws * ws, C)
    d = C // heads
    a = pre + "attention."
    q = F.linear(win, sd[a + "q_proj.weight"], sd[a + "q_proj.bias"]).view(-1, ws * ws, heads, d).transpose(1, 2)
    k = F.linear(win, sd[a + "k_proj.weight"], sd[a + "k_proj.bias"]).view(-1, ws * ws, heads, d).transpose(1, 2)
    v = F.linear(win, sd[a + "v_proj.weight"], sd[a + "v_proj.bias"]).view(-1, ws * ws, heads, d).transpose(1, 2)
    table = sd[a + "relative_position_bias.relative_position_bias_table"]
    bias = table[relative_position_index(ws).view(-1)].view(ws * ws, ws * ws, -1).permute(2, 0, 1).contiguous().unsqueeze(0)
    mask = shift_mask(Hp, Wp, ws, shift)  # get_attn_mask(height_pad, width_pad)
    if mask is not None:
        nW = mask.shape[0]
        m = mask.unsqueeze(1).unsqueeze(0).expand(win.shape[0] // nW, -1, -1, -1, -1).reshape(-1, 1, ws * ws, ws * ws)
        comb = bias + m
    else:
        comb = bias
    s = torch.matmul(q, k.transpose(2, 3)) * (d ** -0.5) + comb
    p = F.softmax(s, dim=-1, dtype=torch.float32)
    o = torch.matmul(p, v).transpose(1, 2).contiguous().reshape(-1, ws * ws, C)
    o = F.linear(o, sd[a + "o_proj.weight"], sd[a + "o_proj.bias"])
    o = window_reverse(o.view(-1, ws, ws, C), ws, Hp, Wp)
    if shift > 0:
        o = torch.roll(o, shifts=(shift, shift), dims=(1, 2))
    if pad_r or pad_b:  # the padded positions' outputs are dropped
        o = o[:, :H, :W, :].contiguous()
    x = shortcut + o.view(B, H * W, C)
    y = F.layer_norm(x, (C,), sd[pre + "layernorm_after.weight"], sd[pre + "layernorm_after.bias"], eps)
    y = F.linear(y, sd[pre + "mlp.fc1.weight"], sd[pre + "mlp.fc1.bias"])
    y = F.gelu(y)
    y = F.linear(y, sd[pre + "mlp.fc2.weight"], sd[pre + "mlp.fc2.bias"])
    return y + x


def patch_merging(sd, pre, x, H, W):
    """SwinPatchMerging.forward, modeling_swin.py:309-326 (maybe_pad: an odd H or W gets one row / column of zeros)."""
    B, L, C = x.shape
    x = x.view(B, H, W, C)
    if H % 2 or W % 2:
        x = F.pad(x, (0, 0, 0, W % 2, 0, H % 2))
    x = torch.cat([x[:, row::2, col::2, :] for col in range(2) for row in range(2)], dim=-1).view(B, -1, 4 * C)
    x = F.layer_norm(x, (4 * C,), sd[pre + "norm.weight"], sd[pre + "norm.bias"], 1e-5)
    return F.linear(x, sd[pre + "reduction.weight"])


@torch.no_grad()
def swin_forward(sd, cfg, pixel_values):
    """SwinForImageClassification.forward, modeling_swin.py:1029-1066 (SwinModel :849-901, SwinEncoder :778-822,
    SwinEmbeddings :219-244). Returns dict(logits, pooled, last_hidden_state, stage_out=[...])."""
    p, eps, ws = cfg["patch_size"], cfg["layer_norm_eps"], cfg["window_size"]
    e = "swin.embeddings."
    x = F.conv2d(pixel_values, sd[e + "patch_embeddings.projection.weight"], sd[e + "patch_embeddings.projection.bias"],
                 stride=p)
    H, W = x.shape[-2:]
    x = x.flatten(2).transpose(1, 2)
    x = F.layer_norm(x, (x.shape[-1],), sd[e + "norm.weight"], sd[e + "norm.bias"], 1e-5)
    stage_out = []
    for s, (depth, heads) in enumerate(zip(cfg["depths"], cfg["num_heads"])):
        for b in range(depth):
            x = swin_layer(sd, f"swin.encoder.layers.{s}.blocks.{b}.", x, H, W, heads, ws, 0 if b % 2 == 0 else ws // 2, eps)
        if s < len(cfg["depths"]) - 1:
            x = patch_merging(sd, f"swin.encoder.layers.{s}.downsample.", x, H, W)
            H, W = (H + 1) // 2, (W + 1) // 2
        stage_out.append(x)
    C = x.shape[-1]
    seq = F.layer_norm(x, (C,), sd["swin.layernorm.weight"], sd["swin.layernorm.bias"], eps)
    pooled = seq.transpose(1, 2).mean(-1)
    logits = F.linear(pooled, sd["classifier.weight"], sd["classifier.bias"])
    return dict(logits=logits, pooled=pooled, last_hidden_state=seq, stage_out=stage_out)
