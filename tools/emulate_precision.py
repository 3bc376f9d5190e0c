"""CPU emulation of the contraction-operand roundings of the precision modes (diagnostic; uses the oracle's
state dicts, never part of the product). For a golden case it runs the forward with every MFMA operand rounded the
way the mode rounds it and prints the attention-map L-inf against the fp32 oracle:
  bf16 : both operands rounded to bf16                                   (OCM_PREC_BF16)
  x3   : x = hi + lo (two bf16), product = hi*hi + hi*lo + lo*hi         (OCM_PREC_BF16X3, split-bf16)
  x2w  : activations hi only, weights hi + lo
Usage: python tools/emulate_precision.py [case ...]"""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import vit_oracle as O  # noqa: E402
from tests.helpers import CASES, case_dims, case_inputs, case_state_dict  # noqa: E402


def bf(x):
    return x.to(torch.bfloat16).to(torch.float32)


def split(x):
    hi = bf(x)
    return hi, bf(x - hi)


def mm(a, b, mode):
    """a @ b with the mode's operand rounding, fp64 accumulation (the MFMA accumulates in fp32; the difference is
    far below the effects studied here)."""
    a, b = a.double(), b.double()
    if mode == "fp32":
        return (a @ b).float()
    ah, al = split(a.float())
    bh, bl = split(b.float())
    ah, al, bh, bl = ah.double(), al.double(), bh.double(), bl.double()
    if mode == "bf16":
        return (ah @ bh).float()
    if mode == "x3":
        return (ah @ bh + ah @ bl + al @ bh).float()
    if mode == "x2w":  # second operand (weights / K / V) split, first hi only
        return (ah @ bh + ah @ bl).float()
    raise ValueError(mode)


def forward_attn(sd, cfg, x, mode, modes_by_op=None):
    # per-contraction overrides: {"qk": mode} for every layer, {("qk", 11): mode} for one layer (wins over the class entry)
    layer = [None]
    mo = lambda op: (modes_by_op or {}).get((op, layer[0]), (modes_by_op or {}).get(op, mode))  # noqa: E731
    p, H, eps = cfg["patch_size"], cfg["num_heads"], cfg["eps"]
    B = x.shape[0]
    D = sd["cls_token"].shape[-1]
    w = sd["patch_embed.proj.weight"].reshape(D, -1)
    cols = F.unfold(x, p, stride=p).transpose(1, 2)  # (B, P, C*p*p)
    t = mm(cols, w.t(), mo("patch")) + sd["patch_embed.proj.bias"]
    t = torch.cat((sd["cls_token"].expand(B, -1, -1), t), dim=1)
    t = t + O.interpolate_pos_encoding(sd, t.shape[1] - 1, x.shape[2], x.shape[3], p)
    x = t
    L = cfg["depth"]
    for i in range(L):
        pre = f"blocks.{i}."
        layer[0] = i
        xn = O.layer_norm(sd, pre + "norm1", x, eps)
        qkv = mm(xn, sd[pre + "attn.qkv.weight"].t(), mo("qkv")) + sd[pre + "attn.qkv.bias"]
        N = x.shape[1]
        qkv = qkv.reshape(B, N, 3, H, D // H).permute(2, 0, 3, 1, 4)
        q, k, v = qkv[0], qkv[1], qkv[2]
        s = mm(q, k.transpose(-2, -1), mo("qk")) * cfg["scale"]
        attn = s.softmax(-1)
        if i == L - 1:
            return attn
        y = mm(attn, v, mo("pv")).transpose(1, 2).reshape(B, N, D)
        y = mm(y, sd[pre + "attn.proj.weight"].t(), mo("proj")) + sd[pre + "attn.proj.bias"]
        x = x + y
        xn = O.layer_norm(sd, pre + "norm2", x, eps)
        hdn = F.gelu(mm(xn, sd[pre + "mlp.fc1.weight"].t(), mo("fc1")) + sd[pre + "mlp.fc1.bias"])
        x = x + mm(hdn, sd[pre + "mlp.fc2.weight"].t(), mo("fc2")) + sd[pre + "mlp.fc2.bias"]


def main():
    names = sys.argv[1:] or ["vits16_sharp", "vits16_peaked"]
    torch.set_num_threads(8)
    for name in names:
        case = CASES[name]
        sd = case_state_dict(case)
        cfg = O.make_cfg(sd, case["patch"], case_dims(case)[2])
        x = case_inputs(case)[0]
        ref = forward_attn(sd, cfg, x, "fp32")
        print(f"{name}: attention max {ref.max():.3f}")
        for mode in ("bf16", "x3", "x2w"):
            a = forward_attn(sd, cfg, x, mode)
            print(f"  {mode:5s} L_inf = {(a - ref).abs().max():.3e}")
        for op in ("patch", "qkv", "qk", "pv", "proj", "fc1", "fc2"):
            a = forward_attn(sd, cfg, x, "x3", {op: "bf16"})
            print(f"  x3 with {op:5s} in bf16: L_inf = {(a - ref).abs().max():.3e}")


if __name__ == "__main__":
    main()
