"""Deterministic synthetic weights and OCM-like tiles.

No trained checkpoint exists offline (reference `checkpoints/model*.pth` are missing blobs, the
DINO URL fallback of eval.py:80-92 is a network fetch), so parity and throughput are measured
on synthetic parameters. Every tensor is a pure function of (seed, parameter name, shape) through
a counter-based generator, so this container, the GPU box and the golden-fixture script all
regenerate identical values without sharing RNG state with anything else.

Distributions follow the reference's initialisation (dino/vision_transformer.py:163-174):
Linear / cls / pos ~ N(0, 0.02^2) clamped to [-2, 2] (trunc_normal_ with absolute bounds, i.e.
effectively untruncated), conv weight/bias ~ U(+-1/sqrt(fan_in)) (PyTorch Conv2d default).
`variant` widens that so that every parameter matters in a parity test:
  "init"  : reference init exactly (LayerNorm = (1, 0), Linear biases = 0)
  "full"  : + random Linear biases N(0, 0.02^2), LayerNorm weight 1 + N(0, 0.1^2), bias N(0, 0.05^2)
  "sharp" : "full" with attn.qkv weights x4 (peaked attention; random-init attention is nearly
            uniform, max ~0.006, and hides precision bugs — SURVEY §7-1); attention max ~0.013
  "peaked": "full" with attn.qkv weights x8: attention max ~0.79, CLS-row max ~0.40 (ViT-S/16) —
            the precision stress set
  "qkv<g>": "full" with attn.qkv weights x g (e.g. "qkv6.5"): the gain that gives attention max ~0.8 depends on the
            embedding width and the token count (ViT-B/16 at 384^2: 6.5 -> 0.84; ViT-S/8 at 384^2: 10 -> 0.90), and
            x8 saturates ViT-B's softmax (max 1.0000: the fp32 reference itself is then only good to 5e-4)
"""
import zlib

import numpy as np
import torch

ARCHS = {
    # name: (embed_dim, depth, num_heads)  — vit_tiny/small/base factories (:259-279)
    "vit_tiny": (192, 12, 3),
    "vit_small": (384, 12, 6),
    "vit_base": (768, 12, 12),
}


def _rng(seed, name):
    return np.random.Generator(np.random.Philox(key=[int(seed) & 0xFFFFFFFF, zlib.crc32(name.encode())]))


def _normal(seed, name, shape, std):
    v = _rng(seed, name).standard_normal(size=shape, dtype=np.float64) * std
    return torch.from_numpy(np.clip(v, -2.0, 2.0).astype(np.float32))


def _uniform(seed, name, shape, bound):
    v = _rng(seed, name).uniform(-bound, bound, size=shape)
    return torch.from_numpy(v.astype(np.float32))


def param_shapes(embed_dim, depth, patch_size, in_chans=3, mlp_ratio=4.0, img_size=224):
    """state_dict key -> shape, identical to the reference module's."""
    D, M = embed_dim, int(embed_dim * mlp_ratio)
    n0 = (img_size // patch_size) ** 2 + 1
    shapes = {
        "cls_token": (1, 1, D),
        "pos_embed": (1, n0, D),
        "patch_embed.proj.weight": (D, in_chans, patch_size, patch_size),
        "patch_embed.proj.bias": (D,),
    }
    for i in range(depth):
        b = f"blocks.{i}."
        shapes.update({
            b + "norm1.weight": (D,), b + "norm1.bias": (D,),
            b + "attn.qkv.weight": (3 * D, D), b + "attn.qkv.bias": (3 * D,),
            b + "attn.proj.weight": (D, D), b + "attn.proj.bias": (D,),
            b + "norm2.weight": (D,), b + "norm2.bias": (D,),
            b + "mlp.fc1.weight": (M, D), b + "mlp.fc1.bias": (M,),
            b + "mlp.fc2.weight": (D, M), b + "mlp.fc2.bias": (D,),
        })
    shapes.update({"norm.weight": (D,), "norm.bias": (D,)})
    return shapes


def stress_variant(arch, patch):
    """The calibrated trained-like weight set of a geometry: the qkv gain that puts the attention maximum at 0.8-0.9 without
    saturating the softmax (x8 on ViT-S/16, x6.5 on ViT-B/16, x10 on ViT-S/8: tests/golden_cases.py)."""
    return {("vit_small", 16): "peaked", ("vit_base", 16): "qkv6.5", ("vit_small", 8): "qkv10"}.get((arch, patch), "peaked")


def qkv_gain_of(variant):
    """Factor applied to the attn.qkv weights by `variant` (raises on an unknown variant)."""
    if variant in ("init", "full"):
        return 1.0
    if variant in ("sharp", "peaked"):
        return 4.0 if variant == "sharp" else 8.0
    if variant.startswith("qkv"):
        try:
            return float(variant[3:])
        except ValueError:
            pass
    raise ValueError(f"unknown variant {variant!r}")


def synth_state_dict(embed_dim, depth, patch_size, *, seed=0, variant="full", in_chans=3, mlp_ratio=4.0,
                     img_size=224):
    gain = qkv_gain_of(variant)
    sd = {}
    for name, shape in param_shapes(embed_dim, depth, patch_size, in_chans, mlp_ratio, img_size).items():
        if name.startswith("patch_embed.proj"):
            fan_in = in_chans * patch_size * patch_size
            t = _uniform(seed, name, shape, 1.0 / np.sqrt(fan_in))
        elif ".norm" in name or name.startswith("norm."):
            if variant == "init":
                t = torch.ones(shape) if name.endswith("weight") else torch.zeros(shape)
            elif name.endswith("weight"):
                t = 1.0 + _normal(seed, name, shape, 0.1)
            else:
                t = _normal(seed, name, shape, 0.05)
        elif name.endswith(".bias"):
            t = torch.zeros(shape) if variant == "init" else _normal(seed, name, shape, 0.02)
        else:
            t = _normal(seed, name, shape, 0.02)
            if gain != 1.0 and name.endswith("attn.qkv.weight"):
                t = t * gain
        sd[name] = t
    return sd


def synth_arch_state_dict(arch, patch_size, **kw):
    D, L, _ = ARCHS[arch]
    return synth_state_dict(D, L, patch_size, **kw)


def synth_tiles(batch, height, width=None, *, seed=1234, channels=3):
    """Grayscale OCM-like tiles replicated to RGB, fp32 in [0, 0.3) (real tiles have
    R == G == B, mean ~0.145; ToTensor only, no mean/std normalisation — data.py:291-299)."""
    width = height if width is None else width
    g = torch.Generator().manual_seed(seed)
    x1 = torch.rand(batch, 1, height, width, generator=g) * 0.3
    return x1.expand(-1, channels, -1, -1).contiguous() if channels > 1 else x1


def synth_wrapper_params(embed_dim, stride, out_mult, *, seed=0):
    """Extra parameters of the model.py wrappers: the SimMIM mask token (1,1,D) and the
    1x1-conv decoder Conv2d(D, stride^2 * out_mult, 1) (out_mult = 3 for MIM, 1 for LinearProbing)."""
    O = stride * stride * out_mult
    return {
        "mask_token": _normal(seed, "mask_token", (1, 1, embed_dim), 0.02),
        "decoder.weight": _uniform(seed, f"decoder{out_mult}.weight", (O, embed_dim, 1, 1), 1.0 / np.sqrt(embed_dim)),
        "decoder.bias": _uniform(seed, f"decoder{out_mult}.bias", (O,), 1.0 / np.sqrt(embed_dim)),
    }


def synth_two_layer_decoder_params(embed_dim, stride, *, seed=0):
    """Parameters of LinearProbing.two_layer_decoder (model.py:154-166), keyed by their nn.Sequential state_dict names:
    0 = Conv2d(D, 4 s^2, 3, padding=1), 1 = BatchNorm2d(4 s^2) (eval-mode statistics), 3 = Conv2d(4 s^2, s^2, 3, padding=1)."""
    mid, out = 4 * stride * stride, stride * stride
    return {
        "0.weight": _uniform(seed, "dec2.0.weight", (mid, embed_dim, 3, 3), 1.0 / np.sqrt(9 * embed_dim)),
        "0.bias": _uniform(seed, "dec2.0.bias", (mid,), 1.0 / np.sqrt(9 * embed_dim)),
        "1.weight": 1.0 + _uniform(seed, "dec2.1.weight", (mid,), 0.3),
        "1.bias": _uniform(seed, "dec2.1.bias", (mid,), 0.2),
        "1.running_mean": _uniform(seed, "dec2.1.mean", (mid,), 0.3),
        "1.running_var": 1.0 + _uniform(seed, "dec2.1.var", (mid,), 0.5),
        "3.weight": _uniform(seed, "dec2.3.weight", (out, mid, 3, 3), 1.0 / np.sqrt(9 * mid)),
        "3.bias": _uniform(seed, "dec2.3.bias", (out,), 1.0 / np.sqrt(9 * mid)),
    }


def synth_patch_mask(batch, side, *, seed=7, ratio=0.6):
    """SimMIM-style 0/1 patch mask (B, side, side) int64 with about `ratio` of the patches masked."""
    v = _rng(seed, "patch_mask").uniform(0, 1, size=(batch, side, side))
    return torch.from_numpy((v < ratio).astype(np.int64))


# ---- Swin-T: HF `SwinForImageClassification` state_dict keys and shapes ----
SWIN_TINY = dict(image_size=224, patch_size=4, num_channels=3, embed_dim=96, depths=(2, 2, 6, 2),
                 num_heads=(3, 6, 12, 24), window_size=7, mlp_ratio=4.0, layer_norm_eps=1e-5, num_labels=5)


def swin_param_shapes(cfg):
    C0, p, ws = cfg["embed_dim"], cfg["patch_size"], cfg["window_size"]
    shapes = {
        "swin.embeddings.patch_embeddings.projection.weight": (C0, cfg["num_channels"], p, p),
        "swin.embeddings.patch_embeddings.projection.bias": (C0,),
        "swin.embeddings.norm.weight": (C0,), "swin.embeddings.norm.bias": (C0,),
    }
    ns = len(cfg["depths"])
    for s, (depth, heads) in enumerate(zip(cfg["depths"], cfg["num_heads"])):
        C = C0 * 2 ** s
        M = int(cfg["mlp_ratio"] * C)
        for b in range(depth):
            pre = f"swin.encoder.layers.{s}.blocks.{b}."
            for nm in ("q_proj", "k_proj", "v_proj", "o_proj"):
                shapes[pre + f"attention.{nm}.weight"] = (C, C)
                shapes[pre + f"attention.{nm}.bias"] = (C,)
            shapes[pre + "attention.relative_position_bias.relative_position_bias_table"] = ((2 * ws - 1) ** 2, heads)
            for nm in ("layernorm_before", "layernorm_after"):
                shapes[pre + nm + ".weight"] = (C,)
                shapes[pre + nm + ".bias"] = (C,)
            shapes[pre + "mlp.fc1.weight"] = (M, C)
            shapes[pre + "mlp.fc1.bias"] = (M,)
            shapes[pre + "mlp.fc2.weight"] = (C, M)
            shapes[pre + "mlp.fc2.bias"] = (C,)
        if s < ns - 1:
            pre = f"swin.encoder.layers.{s}.downsample."
            shapes[pre + "reduction.weight"] = (2 * C, 4 * C)
            shapes[pre + "norm.weight"] = (4 * C,)
            shapes[pre + "norm.bias"] = (4 * C,)
    Cl = C0 * 2 ** (ns - 1)
    shapes.update({"swin.layernorm.weight": (Cl,), "swin.layernorm.bias": (Cl,),
                   "classifier.weight": (cfg["num_labels"], Cl), "classifier.bias": (cfg["num_labels"],)})
    return shapes


def synth_swin_state_dict(cfg, *, seed=0, qk_gain=1.0):
    """Deterministic Swin weights: Linear / table ~ N(0, 0.02^2) (HF _init_weights), random biases N(0, 0.02^2),
    LayerNorm weight 1 + N(0, 0.1^2), bias N(0, 0.05^2), conv U(+-1/sqrt(fan_in)). `qk_gain` scales the q/k
    projections and the relative-position table (peaked window attention for the parity tests)."""
    sd = {}
    for name, shape in swin_param_shapes(cfg).items():
        if "projection" in name:
            fan_in = cfg["num_channels"] * cfg["patch_size"] ** 2
            t = _uniform(seed, name, shape, 1.0 / np.sqrt(fan_in))
        elif "norm" in name:
            t = 1.0 + _normal(seed, name, shape, 0.1) if name.endswith("weight") else _normal(seed, name, shape, 0.05)
        elif name.endswith(".bias"):
            t = _normal(seed, name, shape, 0.02)
        else:
            t = _normal(seed, name, shape, 0.02)
            if "q_proj.weight" in name or "k_proj.weight" in name:
                t = t * qk_gain
            if "relative_position_bias_table" in name:
                t = t * (25.0 * qk_gain)
        sd[name] = t
    return sd
