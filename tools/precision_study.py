#!/usr/bin/env python3
"""CPU precision studies behind DESIGN.md §3.9 / §5 (diagnostic; uses the oracle's state dicts, never part of the product).

    python tools/precision_study.py conditioning   # fp32 vs float64 reference, split-bf16 emulation, per qkv gain (ViT-B/16 384^2)
    python tools/precision_study.py fp16           # which contractions could run in single fp16 beside split-bf16 (ViT-S/16 peaked)
    python tools/precision_study.py fold           # LayerNorm folded into its consumer GEMM (un-normalised split operands + row sums)
    python tools/precision_study.py saturated      # which contraction carries the split-bf16 error on the saturated ViT-B set

Every contraction is emulated with its operands rounded the way a mode rounds them (tools/emulate_precision.py) and float64
accumulation; "fp64" runs the whole forward in float64. Findings recorded in DESIGN.md:
  * ViT-B with the x8 qkv gain ("peaked") saturates the softmax (max 1.0000): the fp32 reference itself is 5.5e-4 from float64,
    split-bf16 6e-3 — conditioning; gains 6.5 (ViT-B/16 384^2) and 10 (ViT-S/8 384^2) give attention max 0.84 / 0.90 (the golden sets);
  * single fp16 for the MLP GEMMs beside split-bf16 elsewhere: 4.6e-4 on ViT-S/16 peaked but 1.5e-3 on ViT-B at 0.84 — not adopted;
  * the folded LayerNorm keeps the error (1.3e-4 / 3.7e-4 / 2.9e-4 on the three stress sets); a row mean of 25 sigma costs 6x."""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tools.emulate_precision as E  # noqa: E402
from oracle import vit_oracle as O  # noqa: E402
from vit_ocm_wmsegmentation_amd import synth  # noqa: E402

_base_mm = E.mm


def _mm(a, b, mode):
    if mode == "fp64":
        return a.double() @ b.double()
    if mode == "h1":  # both operands rounded to fp16
        return (a.float().half().double() @ b.float().half().double()).float()
    return _base_mm(a, b, mode)


E.mm = _mm


def _setup(arch, patch, size, gain, iseed, B=1, shift=0.0):
    D, L, H = synth.ARCHS[arch]
    sd = synth.synth_state_dict(D, L, patch, seed=0, variant="full", img_size=224)
    for k in sd:
        if k.endswith("attn.qkv.weight"):
            sd[k] = sd[k] * gain
    if shift:  # stress: a common offset in the residual stream (|mu| >> sigma)
        sd["cls_token"] = sd["cls_token"] + shift
        sd["pos_embed"] = sd["pos_embed"] + shift
    return sd, O.make_cfg(sd, patch, H), synth.synth_tiles(B, size, size, seed=iseed)


def study(arch, patch, size, gain, iseed, B=1, modes=("x3",), ops=None):
    sd, cfg, x = _setup(arch, patch, size, gain, iseed, B)
    ref32 = E.forward_attn(sd, cfg, x, "fp32")
    ref64 = E.forward_attn({k: v.double() for k, v in sd.items()}, cfg, x.double(), "fp64")
    print(f"{arch}/{patch} {size} gain {gain}: attention max {ref32.max():.4f}, cls-row max {ref32[:, :, 0, 1:].max():.4f}, "
          f"fp32 vs float64 {(ref32.double() - ref64).abs().max():.3e}", flush=True)
    for m in modes:
        a = E.forward_attn(sd, cfg, x, m)
        print(f"   {m}: vs fp32 {(a - ref32).abs().max():.3e}  vs float64 {(a.double() - ref64).abs().max():.3e}", flush=True)
    for name, mo in (ops or {}).items():
        a = E.forward_attn(sd, cfg, x, "x3", mo)
        print(f"   x3 + {name}: vs fp32 {(a - ref32).abs().max():.3e}", flush=True)


def saturated(arch="vit_base", patch=16, size=384, gain=8.0, iseed=99):
    """Which contraction carries the split-bf16 error on the SATURATED softmax (`vitb16_384_saturated`: attention max 1.0000)?
    Everything in float64 except ONE contraction class (or one layer's instance of it) with split-bf16 operands."""
    sd, cfg, x = _setup(arch, patch, size, gain, iseed)
    sd64 = {k: v.double() for k, v in sd.items()}
    L = cfg["depth"]
    ref64 = E.forward_attn(sd64, cfg, x.double(), "fp64")
    ref32 = E.forward_attn(sd, cfg, x, "fp32")
    allx3 = E.forward_attn(sd, cfg, x, "x3")
    print(f"{arch}/{patch} {size} gain {gain}: attention max {ref32.max():.4f}; fp32 reference vs float64 "
          f"{(ref32.double() - ref64).abs().max():.3e}; all contractions split-bf16 vs float64 {(allx3.double() - ref64).abs().max():.3e} "
          f"(vs the fp32 reference {(allx3 - ref32).abs().max():.3e})", flush=True)

    def one(label, overrides):
        a = E.forward_attn(sd64, cfg, x.double(), "fp64", overrides)
        print(f"   float64 except {label:44s}: {(a.double() - ref64).abs().max():.3e}", flush=True)
    for op in ("patch", "qkv", "qk", "pv", "proj", "fc1", "fc2"):
        one(f"{op} (all layers) in split-bf16", {op: "x3"})
    last = L - 1
    one("qkv of the LAST block only", {("qkv", last): "x3"})
    one("Q.K^T of the LAST block only", {("qk", last): "x3"})
    one("qkv + Q.K^T of the LAST block", {("qkv", last): "x3", ("qk", last): "x3"})
    every = {op: "x3" for op in ("patch", "qkv", "qk", "pv", "proj", "fc1", "fc2")}
    one("everything BUT the last block's qkv + Q.K^T", {**every, ("qkv", last): "fp64", ("qk", last): "fp64"})
    one("everything BUT the last block's Q.K^T", {**every, ("qk", last): "fp64"})
    for lay in (0, L // 2, last - 1):
        one(f"all contractions of block {lay} only", {(op, lay): "x3" for op in ("qkv", "qk", "pv", "proj", "fc1", "fc2")})


def forward_folded(sd, cfg, x):
    """out = rstd * (split(x) . split(W * gamma)^T - mu * c) + d for attn.qkv and mlp.fc1; everything else as the x3 mode."""
    p, H, eps = cfg["patch_size"], cfg["num_heads"], cfg["eps"]
    B, D = x.shape[0], sd["cls_token"].shape[-1]
    cols = F.unfold(x, p, stride=p).transpose(1, 2)
    t = _mm(cols, sd["patch_embed.proj.weight"].reshape(D, -1).t(), "x3") + sd["patch_embed.proj.bias"]
    t = torch.cat((sd["cls_token"].expand(B, -1, -1), t), dim=1)
    x = t + O.interpolate_pos_encoding(sd, t.shape[1] - 1, x.shape[2], x.shape[3], p)

    def folded(x, W, bias, g, b):
        Wp = (W * g[None, :]).float()
        c, d = Wp.double().sum(1).float(), (W.double() @ b.double()).float() + bias
        mu = x.sum(-1, keepdim=True) / x.shape[-1]
        var = (x * x).sum(-1, keepdim=True) / x.shape[-1] - mu * mu  # one-pass fp32, as the device does
        return (_mm(x, Wp.t(), "x3") - mu * c) / torch.sqrt(var + eps) + d
    L = cfg["depth"]
    for i in range(L):
        pre = f"blocks.{i}."
        qkv = folded(x, sd[pre + "attn.qkv.weight"], sd[pre + "attn.qkv.bias"], sd[pre + "norm1.weight"], sd[pre + "norm1.bias"])
        N = x.shape[1]
        q, k, v = qkv.reshape(B, N, 3, H, D // H).permute(2, 0, 3, 1, 4)
        attn = (_mm(q, k.transpose(-2, -1), "x3") * cfg["scale"]).softmax(-1)
        if i == L - 1:
            return attn
        y = _mm(attn, v, "x3").transpose(1, 2).reshape(B, N, D)
        x = x + _mm(y, sd[pre + "attn.proj.weight"].t(), "x3") + sd[pre + "attn.proj.bias"]
        hdn = F.gelu(folded(x, sd[pre + "mlp.fc1.weight"], sd[pre + "mlp.fc1.bias"], sd[pre + "norm2.weight"], sd[pre + "norm2.bias"]))
        x = x + _mm(hdn, sd[pre + "mlp.fc2.weight"].t(), "x3") + sd[pre + "mlp.fc2.bias"]


def fold(arch, patch, size, gain, iseed, B=1, shift=0.0):
    sd, cfg, x = _setup(arch, patch, size, gain, iseed, B, shift)
    ref = E.forward_attn(sd, cfg, x, "fp32")
    a, f = E.forward_attn(sd, cfg, x, "x3"), forward_folded(sd, cfg, x)
    print(f"{arch}/{patch} {size} gain {gain} offset {shift}: attention max {ref.max():.3f}; LayerNorm then x3 GEMM "
          f"{float((a - ref).abs().max()):.3e}; folded {float((f - ref).abs().max()):.3e}", flush=True)


def main():
    torch.set_num_threads(8)
    which = sys.argv[1] if len(sys.argv) > 1 else "conditioning"
    if which == "conditioning":
        for g in (8.0, 7.0, 6.5, 6.0, 5.0):
            study("vit_base", 16, 384, g, 99, modes=("x3", "bf16"))
        for g in (8.0, 9.5, 10.0, 10.5):
            study("vit_small", 8, 384, g, 4321)
    elif which == "fp16":
        ops = {"fc2 in fp16": {"fc2": "h1"}, "fc1 in fp16": {"fc1": "h1"}, "fc1 + fc2 in fp16": {"fc1": "h1", "fc2": "h1"},
               "proj in fp16": {"proj": "h1"}, "P.V in fp16": {"pv": "h1"}, "patch in fp16": {"patch": "h1"},
               "qkv in fp16": {"qkv": "h1"}, "Q.K^T in fp16": {"qk": "h1"}}
        study("vit_small", 16, 224, 8.0, 1234, B=2, modes=("x3", "h1", "bf16"), ops=ops)
        study("vit_base", 16, 384, 6.5, 99, ops={"fc1 + fc2 in fp16": {"fc1": "h1", "fc2": "h1"}})
    elif which == "saturated":
        saturated()
        saturated("vit_base", 16, 384, 6.5, 99)  # the calibrated (unsaturated) set of the same geometry, for scale
    elif which == "fold":
        fold("vit_small", 16, 224, 8.0, 1234, B=2)
        fold("vit_small", 16, 224, 8.0, 1234, B=2, shift=0.5)
        fold("vit_base", 16, 384, 6.5, 99)
        fold("vit_small", 8, 384, 10.0, 4321)
    else:
        raise SystemExit(__doc__)


if __name__ == "__main__":
    main()
