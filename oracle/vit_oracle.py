"""ORACLE — TEST INFRASTRUCTURE ONLY.  Not part of the product path.

CPU restatement (fp32, plain torch-CPU functional ops, the reference's own op order) of the one hot
path this repository accelerates: the ViT forward of
    /root/reference/Self-supervised_segmentation/dino/vision_transformer.py
plus the few lines of its callers that define the output contract. Only `tests/`,
`__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` may import this file — as the
checker or the timed CPU baseline, never as something the product calls. The product
(`vit-ocm-wmsegmentation_amd/`) imports nothing from here and has no CPU fallback.

Parity is PINNED: `oracle/make_golden.py` imports the real reference module in the build container,
loads the same synthetic state_dict into it, asserts this restatement agrees to <= 1e-6 on every
output and writes `tests/golden/*.npz`; `tests/test_oracle_golden.py` re-checks the restatement
against those fixtures wherever the tests run (the reference itself never travels).

Everything operates on a flat `sd` = {state_dict key: fp32 tensor} (keys of SURVEY §8-b) and a
`cfg` dict(patch_size, num_heads, depth, eps, scale). Each function cites the reference lines
it restates (paths relative to Self-supervised_segmentation/).
"""
import math

import numpy as np
import torch
import torch.nn.functional as F


def make_cfg(sd, patch_size, num_heads, eps=1e-6, qk_scale=None):
    depth = 1 + max(int(k.split(".")[1]) for k in sd if k.startswith("blocks."))
    dim = sd["cls_token"].shape[-1]
    return dict(patch_size=patch_size, num_heads=num_heads, depth=depth, eps=eps,
                scale=qk_scale or (dim // num_heads) ** -0.5)


def patch_embed(sd, x, p):
    """dino/vision_transformer.py:129-132 — Conv2d(k=p, s=p) -> flatten(2) -> transpose(1, 2)."""
    y = F.conv2d(x, sd["patch_embed.proj.weight"], sd["patch_embed.proj.bias"], stride=p)
    return y.flatten(2).transpose(1, 2)


def interpolate_pos_encoding(sd, npatch, w, h, p):
    """dino/vision_transformer.py:176-196."""
    pos = sd["pos_embed"]
    n0 = pos.shape[1] - 1
    if npatch == n0 and w == h:
        return pos
    cls_pos, patch_pos = pos[:, 0], pos[:, 1:]
    dim = pos.shape[-1]
    w0, h0 = w // p + 0.1, h // p + 0.1
    side = int(math.sqrt(n0))
    patch_pos = F.interpolate(patch_pos.reshape(1, side, side, dim).permute(0, 3, 1, 2),
                              scale_factor=(w0 / math.sqrt(n0), h0 / math.sqrt(n0)), mode="bicubic")
    assert int(w0) == patch_pos.shape[-2] and int(h0) == patch_pos.shape[-1]
    patch_pos = patch_pos.permute(0, 2, 3, 1).view(1, -1, dim)
    return torch.cat((cls_pos.unsqueeze(0), patch_pos), dim=1)


def prepare_tokens(sd, cfg, x):
    """dino/vision_transformer.py:198-209 (pos_drop is p=0)."""
    B, _, w, h = x.shape
    t = patch_embed(sd, x, cfg["patch_size"])
    t = torch.cat((sd["cls_token"].expand(B, -1, -1), t), dim=1)
    return t + interpolate_pos_encoding(sd, t.shape[1] - 1, w, h, cfg["patch_size"])


def layer_norm(sd, prefix, x, eps):
    """nn.LayerNorm(D, eps=1e-6) — :98,102,158 with eps from :262-278."""
    return F.layer_norm(x, (x.shape[-1],), sd[prefix + ".weight"], sd[prefix + ".bias"], eps)


def attention(sd, cfg, i, x):
    """Attention.forward, dino/vision_transformer.py:78-90 -> (x, attn, qkv)."""
    B, N, Cd = x.shape
    H = cfg["num_heads"]
    pre = f"blocks.{i}.attn."
    qkv = F.linear(x, sd[pre + "qkv.weight"], sd.get(pre + "qkv.bias"))
    qkv = qkv.reshape(B, N, 3, H, Cd // H).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    attn = (q @ k.transpose(-2, -1)) * cfg["scale"]
    attn = attn.softmax(dim=-1)
    y = (attn @ v).transpose(1, 2).reshape(B, N, Cd)
    y = F.linear(y, sd[pre + "proj.weight"], sd[pre + "proj.bias"])
    return y, attn, qkv


def mlp(sd, i, x):
    """Mlp.forward, dino/vision_transformer.py:57-63 (exact-erf GELU, dropout p=0)."""
    pre = f"blocks.{i}.mlp."
    return F.linear(F.gelu(F.linear(x, sd[pre + "fc1.weight"], sd[pre + "fc1.bias"])), sd[pre + "fc2.weight"],
                    sd[pre + "fc2.bias"])


def block(sd, cfg, i, x, return_attention=False):
    """Block.forward, dino/vision_transformer.py:106-114."""
    y, attn, qkv = attention(sd, cfg, i, layer_norm(sd, f"blocks.{i}.norm1", x, cfg["eps"]))
    if return_attention:
        return attn
    x = x + y
    x = x + mlp(sd, i, layer_norm(sd, f"blocks.{i}.norm2", x, cfg["eps"]))
    return x, attn, qkv


@torch.no_grad()
def get_intermediate_feat(sd, cfg, x, n=1):
    """dino/vision_transformer.py:225-237 -> (feat, attns, qkvs) lists of length n."""
    x = prepare_tokens(sd, cfg, x)
    feat, attns, qkvs = [], [], []
    for i in range(cfg["depth"]):
        x, attn, qkv = block(sd, cfg, i, x)
        if cfg["depth"] - i <= n:
            feat.append(layer_norm(sd, "norm", x, cfg["eps"]))
            qkvs.append(qkv)
            attns.append(attn)
    return feat, attns, qkvs


@torch.no_grad()
def get_last_selfattention(sd, cfg, x):
    """dino/vision_transformer.py:239-246."""
    x = prepare_tokens(sd, cfg, x)
    for i in range(cfg["depth"]):
        if i < cfg["depth"] - 1:
            x = block(sd, cfg, i, x)[0]
        else:
            return block(sd, cfg, i, x, return_attention=True)


@torch.no_grad()
def forward_feats(sd, cfg, x):
    """dino/vision_transformer.py:218-223 (forward :211-216 is [:, 0] of this)."""
    x = prepare_tokens(sd, cfg, x)
    for i in range(cfg["depth"]):
        x = block(sd, cfg, i, x)[0]
    return layer_norm(sd, "norm", x, cfg["eps"])


@torch.no_grad()
def get_intermediate_layers(sd, cfg, x, n=1):
    """dino/vision_transformer.py:248-256."""
    x = prepare_tokens(sd, cfg, x)
    out = []
    for i in range(cfg["depth"]):
        x = block(sd, cfg, i, x)[0]
        if cfg["depth"] - i <= n:
            out.append(layer_norm(sd, "norm", x, cfg["eps"]))
    return out


def compute_attention(attentions, query, w_featmap, h_featmap, patch_size):
    """utils.py:229-235. Pinned: oracle/make_golden.py runs the reference's own function (extracted from the file,
    whose module import needs cv2/skimage) on the same inputs -> tests/golden/helpers.npz."""
    a = attentions[0]
    nh = a.shape[1]
    a = a[0, :, query, 1:].reshape(nh, -1)
    a = a.reshape(nh, w_featmap, h_featmap)
    a = F.interpolate(a.unsqueeze(0), scale_factor=patch_size, mode="nearest")[0].cpu().numpy()
    return a, nh


def region_query_index(py, px, patch_size, w_featmap):
    """analyse_attention.py:192-193."""
    return int(py // patch_size * w_featmap + px // patch_size)


def sliding_window_origins(height, width, stride):
    """sw_processing.py:151-163: `for y in range(0, height - stride*2, stride): for x in range(0,
    width - stride*2, stride)`, row-major; returns [(y, x), ...]. (The reference unpacks
    `height, width = image.size`, PIL's (W, H); it only runs on square images.)"""
    return [(y, x) for y in range(0, height - stride * 2, stride) for x in range(0, width - stride * 2, stride)]


def sliding_window_crops(image_chw, stride, window):
    """The crops of sw_processing.py:157-160 as a (T, C, window, window) tensor. PIL's Image.crop returns the
    array slice for in-bounds boxes and ZERO-fills whatever part of the box lies outside the image (windows
    of a slab whose side is not a multiple of the stride, or window > 3 * stride); pinned against PIL itself in
    tests/test_host_logic.py::test_out_of_bounds_windows_are_zero_filled_like_pil_crop."""
    _, H, W = image_chw.shape
    origins = sliding_window_origins(H, W, stride)
    pad_h = max(0, max(y for y, _ in origins) + window - H) if origins else 0
    pad_w = max(0, max(x for _, x in origins) + window - W) if origins else 0
    if pad_h or pad_w:
        image_chw = F.pad(image_chw, (0, pad_w, 0, pad_h))
    return torch.stack([image_chw[:, y:y + window, x:x + window] for y, x in origins])


def blend_overlap(a, b, axis):
    """sw_processing.py:136-149: linear ramp np.linspace(1, 0, n) over the overlap along `axis`
    (weights are float64 there; the sum is cast back to the crops' dtype on assignment)."""
    n = a.shape[axis]
    w = np.linspace(1, 0, n)
    w = w[:, None] if axis == 0 else w[None, :]
    return (a * w + b * (1 - w)).astype(a.dtype)


def concat_crops(crops, stride, window):
    """sw_processing.py:113-134: sequential left-to-right then top-to-bottom stitching of the n x n
    row-major window grid, each new crop blended over the `window - stride` overlap with the
    image accumulated so far."""
    n = int(np.sqrt(len(crops)))
    step = window - stride
    vertical = None
    for i in range(n):
        horizontal = crops[i * n]
        for j in range(1, n):
            right = crops[i * n + j]
            overlap = blend_overlap(horizontal[:, -step:], right[:, :-stride], axis=1)
            horizontal = np.concatenate((horizontal[:, :-step], overlap, right[:, -stride:]), axis=1)
        if i == 0:
            vertical = horizontal
        else:
            top = blend_overlap(vertical[-step:, :], horizontal[:-stride, :], axis=0)
            vertical = np.concatenate((vertical[:-step, :], top, horizontal[-stride:, :]), axis=0)
    return vertical


def tile_head_mean_maps(sd, cfg, tiles, patch_size):
    """The per-tile body of the serial loop sw_processing.py:235-245 at B=1 per call:
    get_intermediate_feat -> compute_attention(query 0) -> mean over heads. Returns
    (T, window, window) float32 numpy."""
    out = []
    for j in range(tiles.shape[0]):
        crop = tiles[j:j + 1]
        _, attns, _ = get_intermediate_feat(sd, cfg, crop, n=1)
        a, _ = compute_attention(attns, 0, crop.shape[-2] // patch_size, crop.shape[-1] // patch_size, patch_size)
        out.append(np.mean(a, axis=0))
    return np.stack(out)


def tile_postprocess(rows):
    """sw_processing.py:245,253-254 on (T, H, P) float32 CLS-row maps: np.mean over heads, per-window
    (v - min) / (max - min) * 255 — numpy float32 arithmetic, as the reference evaluates it."""
    rows = np.asarray(rows, dtype=np.float32)
    out = []
    for t in range(rows.shape[0]):
        avg = np.mean(rows[t], axis=0)
        avg = (avg - avg.min()) / (avg.max() - avg.min())
        out.append(avg * 255)
    return np.stack(out)


def bilinear_upsample(maps, scale):
    """cv2.resize(INTER_LINEAR) x scale (sw_processing.py:257) restated with torch's bilinear
    (align_corners=False = half-pixel centres + border replicate, cv2's geometry for up-scaling).
    cv2 is an un-vendored dependency that cannot be installed here: PARITY UNPINNED."""
    t = torch.from_numpy(np.asarray(maps, dtype=np.float32)).unsqueeze(1)
    return F.interpolate(t, scale_factor=scale, mode="bilinear", align_corners=False)[:, 0].numpy()


def otsu_level(img_u8):
    """cv2.threshold(..., THRESH_OTSU) level: OpenCV 4.6 getThreshVal_Otsu_8u restated (PARITY UNPINNED)."""
    hist = np.bincount(np.asarray(img_u8, dtype=np.uint8).ravel(), minlength=256).astype(np.float64)
    scale = 1.0 / hist.sum()
    mu = float((np.arange(256) * hist).sum() * scale)
    mu1 = q1 = max_sigma = 0.0
    level = 0
    eps = float(np.finfo(np.float32).eps)
    for i in range(256):
        p_i = hist[i] * scale
        mu1 *= q1
        q1 += p_i
        q2 = 1.0 - q1
        if min(q1, q2) < eps or max(q1, q2) > 1.0 - eps:
            continue
        mu1 = (mu1 + i * p_i) / q1
        mu2 = (mu - q1 * mu1) / q2
        sigma = q1 * q2 * (mu1 - mu2) ** 2
        if sigma > max_sigma:
            max_sigma, level = sigma, i
    return level


def heatmap_mask(heat):
    """threshold() heat-map branch, sw_processing.py:30-35,43,47-48,62."""
    heat = np.asarray(heat, dtype=np.float32)
    mn, mx = heat.min(), heat.max()
    a = heat if mx == mn else (heat - mn) / (mx - mn)
    img = (a * 255).astype(np.uint8)
    level = otsu_level(img)
    return img, np.where(img > level, 255, 0).astype(np.uint8), level


# ------------------------------------------------------------------------------------------------
# eval.py's per-image mask chain (eval.py:126-171 with the default --crop 1 --median_filter 1, and
# utils.py:55-115 threshold()). cv2 / torchvision / PIL arithmetic is restated: PARITY UNPINNED.
# ------------------------------------------------------------------------------------------------
def to_pil_gray_u8(img_chw):
    """transform(img.squeeze(0)).convert("L") (eval.py:122,166): torchvision ToPILImage on a float CHW
    tensor is pic.mul(255).byte(); PIL's RGB->L is (19595 R + 38470 G + 7471 B + 0x8000) >> 16."""
    a = (np.asarray(img_chw, dtype=np.float32) * np.float32(255)).astype(np.int32).astype(np.uint8).astype(np.uint32)
    if a.shape[0] == 1:
        return a[0].astype(np.uint8)
    return ((19595 * a[0] + 38470 * a[1] + 7471 * a[2] + 0x8000) >> 16).astype(np.uint8)


def eval_average_attention(cls_rows, hf, wf, patch_size):
    """eval.py:136-144,164-166 for one image: cls_rows (H, hf*wf) = attentions[0][0, :, 0, 1:] ->
    nearest x p -> np.mean over heads -> median_filter(size=1) (identity) -> cv2.resize down by p (returns
    the hf x wf block values) -> cv2.resize INTER_LINEAR up to the image size."""
    rows = np.asarray(cls_rows, dtype=np.float32)
    s = rows[0].copy()
    for h in range(1, rows.shape[0]):
        s = s + rows[h]
    avg = (s / np.float32(rows.shape[0])).astype(np.float32).reshape(1, hf, wf)
    return bilinear_upsample(avg, patch_size)[0]


def threshold_masks(img_u8, attention):
    """utils.py:61-115 threshold(img, attention, save=False) -> (th, th2, th3) plus the Otsu levels."""
    attention = np.asarray(attention, dtype=np.float32)
    mn, mx = attention.min(), attention.max()
    att = attention if mx == mn else (attention - mn) / (mx - mn)
    img = np.asarray(img_u8, dtype=np.uint8)
    alpha = 0.4
    att = (att * 255).astype(np.uint8)
    result = ((img / 2) * (1 - alpha) + (att / 2) * alpha).astype(np.uint8)
    l1, l2, l3 = otsu_level(result), otsu_level(img), otsu_level(att)
    th = np.where(result > l1, 255, 0).astype(np.uint8)
    th2 = (img > l2).astype(np.uint8) * 255
    th3 = np.where(att > l3, 255, 0).astype(np.uint8)
    return (th, th2, th3), (l1, l2, l3), result


# ------------------------------------------------------------------------------------------------
# model.py wrappers. Pinned against the reference's own classes, extracted from
# model.py by oracle/ref_extract.py (the module itself imports timm, absent here): oracle/make_golden.py.
# ------------------------------------------------------------------------------------------------
def encoder_fmap(sd, cfg, x, img_size, mask=None, mask_token=None):
    """VisionTransformerForSimMIM.forward (model.py:24-53) when `mask` is given, else
    VisionTransformerForFinetune.forward (model.py:121-139): (B, D, H, W) with H = W = int(L ** 0.5)."""
    p = cfg["patch_size"]
    t = patch_embed(sd, x, p)
    B, L, D = t.shape
    if mask is not None:
        tok = mask_token.expand(B, L, -1)
        w = mask.flatten(1).unsqueeze(-1).type_as(tok)
        t = t * (1 - w) + tok * w
    t = torch.cat((sd["cls_token"].expand(B, -1, -1), t), dim=1)
    if img_size != 224:
        t = t + interpolate_pos_encoding(sd, L, img_size, img_size, p)
    else:
        t = t + sd["pos_embed"]
    for i in range(cfg["depth"]):
        t = block(sd, cfg, i, t)[0]
    t = layer_norm(sd, "norm", t, cfg["eps"])[:, 1:]
    side = int(L ** 0.5)
    return t.permute(0, 2, 1).reshape(B, D, side, side)


def conv1x1_pixel_shuffle(z, weight, bias, stride):
    """nn.Sequential(Conv2d(D, s*s*c, 1), PixelShuffle(s)) of model.py:60-66,147-152."""
    return F.pixel_shuffle(F.conv2d(z, weight, bias), stride)


def two_layer_decoder(z, prm, stride, bn_eps=1e-5):
    """LinearProbing.two_layer_decoder in eval mode (model.py:154-166): Conv2d(3x3, pad 1) -> BatchNorm2d (running
    statistics) -> ReLU -> Conv2d(3x3, pad 1) -> PixelShuffle(stride). `prm`: synth.synth_two_layer_decoder_params."""
    y = F.conv2d(z, prm["0.weight"], prm["0.bias"], padding=1)
    y = F.batch_norm(y, prm["1.running_mean"], prm["1.running_var"], prm["1.weight"], prm["1.bias"], False, 0.0, bn_eps)
    y = F.relu(y)
    y = F.conv2d(y, prm["3.weight"], prm["3.bias"], padding=1)
    return F.pixel_shuffle(y, stride)


def mim_forward(sd, cfg, x, mask, img_size, mask_token, dec_w, dec_b, stride, patch_size=8, in_chans=3):
    """MIM.forward (model.py:68-74) -> (loss, x_rec, mask)."""
    z = encoder_fmap(sd, cfg, x, img_size, mask=mask, mask_token=mask_token)
    x_rec = conv1x1_pixel_shuffle(z, dec_w, dec_b, stride)
    m = mask.repeat_interleave(patch_size, 1).repeat_interleave(patch_size, 2).unsqueeze(1).contiguous()
    loss_recon = F.l1_loss(x, x_rec, reduction="none")
    loss = (loss_recon * m).sum() / (m.sum() + 1e-5) / in_chans
    return loss, x_rec, m
