"""Swin-T (BASELINE config 5). CPU: the oracle reproduces the fixtures written from the installed
transformers package, the mirror keeps its state_dict keys. GPU: the product against the fixtures and the stand-alone
(shifted-)window attention against the oracle's window arithmetic."""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle import swin_oracle as SO
from tests.golden_cases import SWIN_CASES
from tests.helpers import load_golden
from vit_ocm_wmsegmentation_amd import _lib, synth
from vit_ocm_wmsegmentation_amd import swin as SW


def _case(name):
    c = SWIN_CASES[name]
    cfg = dict(synth.SWIN_TINY, **c.get("cfg", {}))
    sd = synth.synth_swin_state_dict(cfg, seed=c["seed"], qk_gain=c["qk_gain"])
    x = synth.synth_tiles(c["batch"], cfg["image_size"], seed=c["seed"] + 50)
    return c, cfg, sd, x


@pytest.mark.parametrize("name", ["small56", "pad120"])
def test_oracle_reproduces_transformers_fixture(name):
    c, cfg, sd, x = _case(name)
    g = load_golden("swin_" + name)
    assert float(g["oracle_vs_transformers_maxabs"]) <= 2e-5
    o = SO.swin_forward(sd, cfg, x)
    assert np.abs(o["logits"].numpy() - g["logits"]).max() <= 2e-5
    assert np.abs(o["pooled"].numpy() - g["pooled"]).max() <= 2e-5
    for s, t in enumerate(o["stage_out"]):
        assert np.abs(t[:, :4, :32].numpy() - g[f"stage{s}_head"]).max() <= 2e-5


def test_mirror_keeps_transformers_state_dict_keys():
    cfg = SW.SwinConfig(num_labels=5)
    m = SW.SwinForImageClassification(cfg)
    keys = set(m.state_dict())
    assert keys == set(synth.swin_param_shapes(synth.SWIN_TINY))
    assert "swin.encoder.layers.2.blocks.5.attention.relative_position_bias.relative_position_bias_table" in keys
    sd = synth.synth_swin_state_dict(synth.SWIN_TINY, seed=3)
    msg = m.load_state_dict(sd, strict=True)
    assert not msg.missing_keys and not msg.unexpected_keys
    assert torch.equal(m.state_dict()["classifier.weight"], sd["classifier.weight"])
    assert cfg.hidden_size == 768 and m.num_labels == 5
    with pytest.raises(RuntimeError, match="HIP"):
        m(pixel_values=torch.zeros(1, 3, 224, 224))
    for mode in ("bf16", "fp32", "bf16x3"):  # the split-bf16 mode exists for Swin since round 3
        assert m.set_precision(mode) is m
    with pytest.raises(ValueError):
        m.set_precision("fp16")


def test_swin_create_rejects_unbuilt_geometries(lib):
    def create(**kw):
        base = dict(image_size=224, patch_size=4, num_channels=3, embed_dim=96, num_stages=4, window_size=7, num_labels=5,
                    mlp_ratio=4.0, ln_eps=1e-5, precision=0, reserved=0)
        base.update(kw)
        heads = base.pop("heads", (3, 6, 12, 24))
        cfg = _lib.OcmSwinConfig(**base)
        for i in range(4):
            cfg.depths[i], cfg.num_heads[i] = (2, 2, 6, 2)[i], heads[i]
        h = C.c_void_p(0)
        return lib.ocm_swin_create(C.byref(cfg), C.byref(h))

    # (image_size 96: the third stage's 6 x 6 grid is smaller than the window, which transformers itself cannot run; 202 is not
    # a multiple of the patch size. Grids that are not multiples of the window, or odd, are padded since round 4: 200 is built)
    for bad in (dict(patch_size=8), dict(window_size=8), dict(image_size=96), dict(image_size=202), dict(heads=(4, 6, 12, 24)),
                dict(embed_dim=100), dict(num_labels=0), dict(precision=3), dict(precision=-1)):
        assert create(**bad) == _lib.OCM_EINVAL, bad


# ------------------------------------------------------------------------------------------------
def _window_attention_oracle(qkv, B, H, W, heads, ws, shift, table):
    """(shifted-)window attention of SwinLayer.forward on already projected q|k|v tokens."""
    C_ = heads * 32
    t = qkv.view(B, H, W, 3 * C_)
    if shift:
        t = torch.roll(t, shifts=(-shift, -shift), dims=(1, 2))
    win = SO.window_partition(t, ws).view(-1, ws * ws, 3, heads, 32)
    q, k, v = (win[:, :, i].transpose(1, 2) for i in range(3))
    bias = table[SO.relative_position_index(ws).view(-1)].view(ws * ws, ws * ws, -1).permute(2, 0, 1).unsqueeze(0)
    s = q @ k.transpose(2, 3) * 32 ** -0.5 + bias
    m = SO.shift_mask(H, W, ws, shift)
    if m is not None:
        nW = m.shape[0]
        s = s + m.unsqueeze(1).unsqueeze(0).expand(win.shape[0] // nW, -1, -1, -1, -1).reshape(-1, 1, ws * ws, ws * ws)
    o = (torch.softmax(s, -1) @ v).transpose(1, 2).reshape(-1, ws, ws, C_)
    o = SO.window_reverse(o, ws, H, W)
    if shift:
        o = torch.roll(o, shifts=(shift, shift), dims=(1, 2))
    return o.reshape(B * H * W, C_)


@pytest.mark.gpu
@pytest.mark.parametrize("precision,tol", [("fp32", 2e-5), ("bf16", 2e-2), ("bf16x3", 1e-4)])
@pytest.mark.parametrize("H,W,ws,shift,heads", [(14, 14, 7, 0, 3), (14, 14, 7, 3, 3), (28, 14, 7, 3, 6), (7, 7, 7, 0, 12),
                                              (8, 12, 4, 2, 2)])
def test_window_attention_op(lib, dev, precision, tol, H, W, ws, shift, heads):
    from vit_ocm_wmsegmentation_amd.engine import from_split, to_operand
    B, C_ = 2, heads * 32
    g = torch.Generator().manual_seed(H * 100 + W + shift)
    qkv = torch.randn(B * H * W, 3 * C_, generator=g)
    qkv[:, :2 * C_] *= 1.5
    table = torch.randn((2 * ws - 1) ** 2, heads, generator=g)
    if precision == "bf16x3":  # split-bf16 pairs: 4 bytes per element, groups of 32 as [32 x hi | 32 x lo]
        qd = to_operand(qkv.to(dev), _lib.OCM_PREC_BF16X3)
        want = _window_attention_oracle(from_split(qd).cpu(), B, H, W, heads, ws, shift, table)
        ctx = torch.full((B * H * W, C_), -1, dtype=torch.int32, device=dev)  # 0xFFFF pairs: NaN + NaN
    else:
        e = torch.float32 if precision == "fp32" else torch.bfloat16
        qd = qkv.to(e).to(dev)
        want = _window_attention_oracle(qd.float().cpu(), B, H, W, heads, ws, shift, table)
        ctx = torch.full((B * H * W, C_), float("nan"), dtype=e, device=dev)
    scratch = torch.empty(heads * (4096 + ws ** 4), dtype=torch.float32, device=dev)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    _lib.check(lib.ocm_op_swin_window_attention(_lib.PRECISIONS[precision], C.c_void_p(qd.data_ptr()), 3 * C_,
                                                C.c_void_p(ctx.data_ptr()), C_, C.c_void_p(table.to(dev).data_ptr()),
                                                C.c_void_p(scratch.data_ptr()), B, H, W, ws, shift, heads, st))
    got = (from_split(ctx) if precision == "bf16x3" else ctx.float()).cpu()
    assert torch.isfinite(got).all()
    err = (got - want).abs().max().item()
    print(f"GPUTEST swin window attention {precision} {H}x{W} ws{ws} shift{shift}: max|d| = {err:.2e}")
    assert err <= tol


@pytest.mark.gpu
@pytest.mark.parametrize("precision,tol", [("fp32", 2e-4), ("bf16", 6e-2), ("bf16x3", 1e-3)])
@pytest.mark.parametrize("name", sorted(SWIN_CASES))
def test_swin_matches_transformers_fixture(dev, name, precision, tol):
    """SwinForImageClassification on the HIP path against the outputs of the installed transformers model
    (fixtures). fp32 mode: round-off of 12-24 layers; bf16 mode: logits are O(1), tolerance 6e-2 abs; split-bf16
    mode (pairs hi + lo on the bf16 MFMA, three instructions per product): 1e-3 abs, the bar of the ViT path."""
    c, cfg, sd, x = _case(name)
    g = load_golden("swin_" + name)
    hf = SW.SwinConfig(image_size=cfg["image_size"], depths=cfg["depths"], num_heads=cfg["num_heads"], num_labels=cfg["num_labels"])
    model = SW.SwinForImageClassification(hf)
    assert not model.load_state_dict(sd, strict=True).missing_keys
    model = model.to(dev).eval().set_precision(precision)
    out = model(pixel_values=x.to(dev), output_hidden_states=True)
    print(f"GPUTEST swin {name} {precision}: logits max|d| = {np.abs(out.logits.cpu().numpy() - g['logits']).max():.2e}, "
          f"pooled {np.abs(out.pooler_output.cpu().numpy() - g['pooled']).max():.2e}, hidden "
          f"{np.abs(out.last_hidden_state[:, :8, :64].cpu().numpy() - g['last_hidden_head']).max():.2e}")
    assert np.abs(out.logits.cpu().numpy() - g["logits"]).max() <= tol
    assert np.abs(out.pooler_output.cpu().numpy() - g["pooled"]).max() <= tol
    assert np.abs(out.last_hidden_state[:, :8, :64].cpu().numpy() - g["last_hidden_head"]).max() <= 4 * tol
    assert np.array_equal(out.logits.argmax(-1).cpu().numpy(), g["logits"].argmax(-1))


@pytest.mark.gpu
@pytest.mark.parametrize("T,Cn", [(3136, 96), (1000, 96), (33, 96), (700, 128)])
def test_swin_fused_mlp_op(lib, dev, T, Cn):
    """ocm_op_swin_mlp (LayerNorm + fc1 + GELU + fc2 + residual in one kernel, split-bf16) against float64 torch on the
    same parameters (modeling_swin.py:668-672); token counts with full, partial and single workgroups."""
    from vit_ocm_wmsegmentation_amd.engine import to_operand
    g = torch.Generator().manual_seed(T + Cn)
    x = torch.randn(T, Cn, generator=g) * 2 + 0.3
    gam, bet = torch.randn(Cn, generator=g) * 0.2 + 1, torch.randn(Cn, generator=g) * 0.1
    w1, b1 = torch.randn(4 * Cn, Cn, generator=g) * 0.08, torch.randn(4 * Cn, generator=g) * 0.1
    w2, b2 = torch.randn(Cn, 4 * Cn, generator=g) * 0.05, torch.randn(Cn, generator=g) * 0.1
    xd = x.double()
    h = torch.nn.functional.gelu(torch.nn.functional.layer_norm(xd, (Cn,), gam.double(), bet.double(), 1e-5) @ w1.double().t()
                                 + b1.double())
    want = xd + h @ w2.double().t() + b2.double()
    xg = x.to(dev)
    w1s, w2s = to_operand(w1.to(dev), _lib.OCM_PREC_BF16X3), to_operand(w2.to(dev), _lib.OCM_PREC_BF16X3)
    dv = [t.to(dev) for t in (gam, bet, b1, b2)]
    p = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    _lib.check(lib.ocm_op_swin_mlp(_lib.OCM_PREC_BF16X3, p(xg), p(dv[0]), p(dv[1]), p(w1s), p(dv[2]), p(w2s), p(dv[3]), T, Cn,
                                   4 * Cn, 1e-5, st))
    err = (xg.cpu().double() - want).abs().max().item()
    print(f"GPUTEST swin fused mlp T={T} C={Cn}: max|d| = {err:.2e}")
    assert err <= 5e-5
    assert lib.ocm_op_swin_mlp(_lib.OCM_PREC_BF16, p(xg), p(dv[0]), p(dv[1]), p(w1s), p(dv[2]), p(w2s), p(dv[3]), T, Cn, 4 * Cn,
                               1e-5, st) == _lib.OCM_EINVAL
    assert lib.ocm_op_swin_mlp(_lib.OCM_PREC_BF16X3, p(xg), p(dv[0]), p(dv[1]), p(w1s), p(dv[2]), p(w2s), p(dv[3]), T, 192, 768,
                               1e-5, st) == _lib.OCM_EINVAL


@pytest.mark.gpu
@pytest.mark.parametrize("T,Cn", [(3136, 96), (1000, 96), (33, 96), (700, 128), (1500, 192)])
def test_swin_fused_lnqkv_op(lib, dev, T, Cn):
    """ocm_op_swin_lnqkv (layernorm_before + the q | k | v projection in one kernel, split-bf16) against float64 torch
    (modeling_swin.py:641, :430-432); token counts with full, partial and single workgroups; rows past T stay untouched."""
    from vit_ocm_wmsegmentation_amd.engine import from_split, to_operand
    g = torch.Generator().manual_seed(7 * T + Cn)
    x = torch.randn(T, Cn, generator=g) * 2 + 0.3
    gam, bet = torch.randn(Cn, generator=g) * 0.2 + 1, torch.randn(Cn, generator=g) * 0.1
    w, b = torch.randn(3 * Cn, Cn, generator=g) * 0.08, torch.randn(3 * Cn, generator=g) * 0.1
    want = torch.nn.functional.layer_norm(x.double(), (Cn,), gam.double(), bet.double(), 1e-5) @ w.double().t() + b.double()
    ws = to_operand(w.to(dev), _lib.OCM_PREC_BF16X3)
    dv = [t.to(dev) for t in (x, gam, bet, b)]
    qkv = torch.full((T + 5, 3 * Cn), -1, dtype=torch.int32, device=dev)
    p = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    _lib.check(lib.ocm_op_swin_lnqkv(_lib.OCM_PREC_BF16X3, p(dv[0]), p(dv[1]), p(dv[2]), p(ws), p(dv[3]), p(qkv), T, Cn, 1e-5, st))
    err = (from_split(qkv[:T]).cpu().double() - want).abs().max().item()
    print(f"GPUTEST swin fused ln+qkv T={T} C={Cn}: max|d| = {err:.2e}")
    assert err <= 5e-5
    assert (qkv[T:] == -1).all()
    assert lib.ocm_op_swin_lnqkv(_lib.OCM_PREC_FP32, p(dv[0]), p(dv[1]), p(dv[2]), p(ws), p(dv[3]), p(qkv), T, Cn, 1e-5,
                                 st) == _lib.OCM_EINVAL
    assert lib.ocm_op_swin_lnqkv(_lib.OCM_PREC_BF16X3, p(dv[0]), p(dv[1]), p(dv[2]), p(ws), p(dv[3]), p(qkv), T, 384, 1e-5,
                                 st) == _lib.OCM_EINVAL


def _attention_half_oracle(x, gam, bet, wqkv, bqkv, wo, bo, table, B, H, W, heads, ws, shift):
    """x + o_proj(window attention of LayerNorm(x)) in float64: the first half of the oracle's swin_layer
    (modeling_swin.py:641-666) on explicit tensors, wqkv rows q | k | v."""
    F = torch.nn.functional
    Cn, A = heads * 32, ws * ws
    xd = x.double().view(B, H * W, Cn)
    y = F.layer_norm(xd, (Cn,), gam.double(), bet.double(), 1e-5).view(B, H, W, Cn)
    if shift:
        y = torch.roll(y, shifts=(-shift, -shift), dims=(1, 2))
    win = SO.window_partition(y, ws).view(-1, A, Cn)
    q, k, v = [t.reshape(-1, A, heads, 32).transpose(1, 2) for t in (win @ wqkv.double().t() + bqkv.double()).split(Cn, -1)]
    bias = table.double()[SO.relative_position_index(ws).view(-1)].view(A, A, -1).permute(2, 0, 1).unsqueeze(0)
    mask = SO.shift_mask(H, W, ws, shift)
    if mask is not None:
        nW = mask.shape[0]
        bias = bias + mask.double().unsqueeze(1).unsqueeze(0).expand(win.shape[0] // nW, -1, -1, -1, -1).reshape(-1, 1, A, A)
    p = torch.softmax(q @ k.transpose(2, 3) * 32 ** -0.5 + bias, -1)
    o = (p @ v).transpose(1, 2).reshape(-1, A, Cn) @ wo.double().t() + bo.double()
    o = SO.window_reverse(o.view(-1, ws, ws, Cn), ws, H, W)
    if shift:
        o = torch.roll(o, shifts=(shift, shift), dims=(1, 2))
    return xd + o.reshape(B, H * W, Cn)


@pytest.mark.gpu
@pytest.mark.parametrize("B,H,W,ws,shift,heads", [(2, 14, 14, 7, 0, 3), (2, 14, 14, 7, 3, 3), (3, 7, 7, 7, 0, 3), (1, 12, 8, 4, 2, 3),
                                                  (2, 56, 56, 7, 3, 3), (2, 14, 14, 7, 3, 6), (3, 7, 7, 7, 0, 6), (1, 8, 12, 4, 2, 6),
                                                  (2, 28, 28, 7, 0, 6), (2, 14, 14, 7, 3, 4), (3, 12, 8, 4, 0, 4)])
def test_swin_fused_attention_half_op(lib, dev, B, H, W, ws, shift, heads):
    """ocm_op_swin_attn_block (layernorm_before + q | k | v + (shifted-)window attention + o_proj + residual in one kernel,
    split-bf16; 3 heads: all of it in one kernel, 6 heads: up to the context, then the o_proj GEMM) against float64 torch
    (modeling_swin.py:641-666): plain and shifted windows, window counts that leave idle wavefront pairs, a window side other
    than 7, and the stage-0 / stage-1 grids of Swin-T."""
    from vit_ocm_wmsegmentation_amd.engine import to_operand
    Cn = heads * 32
    g = torch.Generator().manual_seed(B * 1000 + H * 10 + shift)
    x = torch.randn(B * H * W, Cn, generator=g) * 1.5 + 0.2
    gam, bet = torch.randn(Cn, generator=g) * 0.2 + 1, torch.randn(Cn, generator=g) * 0.1
    wqkv, bqkv = torch.randn(3 * Cn, Cn, generator=g) * 0.12, torch.randn(3 * Cn, generator=g) * 0.1
    wo, bo = torch.randn(Cn, Cn, generator=g) * 0.1, torch.randn(Cn, generator=g) * 0.1
    table = torch.randn((2 * ws - 1) ** 2, heads, generator=g)
    want = _attention_half_oracle(x, gam, bet, wqkv, bqkv, wo, bo, table, B, H, W, heads, ws, shift).view(-1, Cn)
    xg = x.to(dev)
    wq_s, wo_s = to_operand(wqkv.to(dev), _lib.OCM_PREC_BF16X3), to_operand(wo.to(dev), _lib.OCM_PREC_BF16X3)
    dv = [t.to(dev) for t in (gam, bet, bqkv, bo, table)]
    scratch = torch.empty(heads * 4096 + B * H * W * Cn, dtype=torch.float32, device=dev)
    p = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    _lib.check(lib.ocm_op_swin_attn_block(_lib.OCM_PREC_BF16X3, p(xg), p(dv[0]), p(dv[1]), p(wq_s), p(dv[2]), p(wo_s), p(dv[3]),
                                          p(dv[4]), p(scratch), B, H, W, ws, shift, heads, 1e-5, st))
    got = xg.cpu().double()
    assert torch.isfinite(got).all()
    err = (got - want).abs().max().item()
    print(f"GPUTEST swin fused attention half B={B} {H}x{W} ws{ws} shift{shift} heads{heads}: max|d| = {err:.2e}")
    assert err <= 1.2e-4 * heads / 3  # sums over C = 32 * heads terms
    assert lib.ocm_op_swin_attn_block(_lib.OCM_PREC_BF16, p(xg), p(dv[0]), p(dv[1]), p(wq_s), p(dv[2]), p(wo_s), p(dv[3]), p(dv[4]),
                                      p(scratch), B, H, W, ws, shift, heads, 1e-5, st) == _lib.OCM_EINVAL
    assert lib.ocm_op_swin_attn_block(_lib.OCM_PREC_BF16X3, p(xg), p(dv[0]), p(dv[1]), p(wq_s), p(dv[2]), p(wo_s), p(dv[3]),
                                      p(dv[4]), p(scratch), B, H, W, ws, shift, 12, 1e-5, st) == _lib.OCM_EINVAL


@pytest.mark.gpu
@pytest.mark.parametrize("heads,H", [(3, 56), (6, 28)])
def test_swin_fused_attention_half_is_deterministic(lib, dev, heads, H):
    """The fused attention half hands K / V images and weight chunks between wavefronts through LDS behind counted waits and
    barriers: a misplaced wait shows as run-to-run differences long before it shows against a reference. Sixteen images of
    the stage's grid (thousands of workgroups in flight), shifted windows, five runs from the same input: identical bits."""
    from vit_ocm_wmsegmentation_amd.engine import to_operand
    B, ws, shift, Cn = 16, 7, 3, heads * 32
    g = torch.Generator().manual_seed(heads)
    x = (torch.randn(B * H * H, Cn, generator=g) * 1.5).to(dev)
    gam, bet = (torch.randn(Cn, generator=g) * 0.2 + 1).to(dev), (torch.randn(Cn, generator=g) * 0.1).to(dev)
    wq = to_operand((torch.randn(3 * Cn, Cn, generator=g) * 0.12).to(dev), _lib.OCM_PREC_BF16X3)
    wo = to_operand((torch.randn(Cn, Cn, generator=g) * 0.1).to(dev), _lib.OCM_PREC_BF16X3)
    bq, bo = (torch.randn(3 * Cn, generator=g) * 0.1).to(dev), (torch.randn(Cn, generator=g) * 0.1).to(dev)
    table = torch.randn((2 * ws - 1) ** 2, heads, generator=g).to(dev)
    scratch = torch.empty(heads * 4096 + B * H * H * Cn, dtype=torch.float32, device=dev)
    p = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    outs = []
    for _ in range(5):
        xg = x.clone()
        _lib.check(lib.ocm_op_swin_attn_block(_lib.OCM_PREC_BF16X3, p(xg), p(gam), p(bet), p(wq), p(bq), p(wo), p(bo), p(table),
                                              p(scratch), B, H, H, ws, shift, heads, 1e-5, st))
        outs.append(xg)
    assert torch.isfinite(outs[0]).all() and not torch.equal(outs[0], x)
    for o in outs[1:]:
        assert torch.equal(o, outs[0])


@pytest.mark.gpu
def test_swin_fused_mlp_agrees_with_three_launches(lib, dev):
    """OCM_SWIN_OPT_FUSE_MLP on / off (fused MLP and fused LayerNorm + qkv kernels of the narrow stage against LayerNorm kernels
    and GEMMs): the same model, logits and hidden states agree to fp32 rounding."""
    c, cfg, sd, x = _case("tiny224")
    hf = SW.SwinConfig(image_size=cfg["image_size"], depths=cfg["depths"], num_heads=cfg["num_heads"], num_labels=cfg["num_labels"])
    model = SW.SwinForImageClassification(hf)
    model.load_state_dict(sd, strict=True)
    model = model.to(dev).eval().set_precision("bf16x3")
    a = model(pixel_values=x.to(dev), output_hidden_states=True)
    _lib.check(lib.ocm_swin_set_option(model._engine["h"], _lib.OCM_SWIN_OPT_FUSE_MLP, 0))
    b = model(pixel_values=x.to(dev), output_hidden_states=True)
    assert lib.ocm_swin_set_option(model._engine["h"], 7, 0) == _lib.OCM_EINVAL
    d = (a.logits - b.logits).abs().max().item()
    dh = (a.last_hidden_state - b.last_hidden_state).abs().max().item()
    print(f"GPUTEST swin fused vs unfused mlp: logits {d:.2e}, hidden {dh:.2e}")
    assert d <= 2e-5 and dh <= 2e-4 and d + dh > 0  # the two paths really differ (and only in rounding)


@pytest.mark.gpu
def test_swin_single_channel_batch_one_vs_oracle(dev):
    """Edge geometry: one grayscale plane (num_channels = 1, K = 16 patch values) and a batch of one, against the
    (transformers-pinned) oracle run on the same synthetic weights."""
    cfg = dict(synth.SWIN_TINY, image_size=56, depths=(2, 2), num_heads=(3, 6), num_channels=1, num_labels=3)
    sd = synth.synth_swin_state_dict(cfg, seed=31, qk_gain=4.0)
    x = synth.synth_tiles(1, 56, seed=77, channels=1)
    want = SO.swin_forward(sd, cfg, x)
    hf = SW.SwinConfig(image_size=56, num_channels=1, depths=cfg["depths"], num_heads=cfg["num_heads"], num_labels=3)
    model = SW.SwinForImageClassification(hf)
    assert not model.load_state_dict(sd, strict=True).missing_keys
    model = model.to(dev).eval().set_precision("fp32")
    out = model(pixel_values=x.to(dev), output_hidden_states=True)
    assert tuple(out.logits.shape) == (1, 3)
    assert (out.logits.cpu() - want["logits"]).abs().max().item() <= 2e-4
    assert (out.last_hidden_state.cpu() - want["last_hidden_state"]).abs().max().item() <= 1e-3
    with pytest.raises(ValueError):
        model(pixel_values=torch.zeros(1, 3, 56, 56, device=dev))


@pytest.mark.gpu
def test_swin_embed_dim_128_vs_oracle(dev):
    """embed_dim 128 (the Swin-B family's channel counts: 4 / 8 heads): stage 0 runs the fused attention half up to the context
    (C = 128) and the fused MLP, stage 1 (C = 256) the unfused chain; split-bf16 and fp32 against the transformers-pinned oracle
    on the same synthetic weights, with shifted windows in both stages."""
    cfg = dict(synth.SWIN_TINY, image_size=112, embed_dim=128, depths=(2, 2), num_heads=(4, 8), num_labels=4)
    sd = synth.synth_swin_state_dict(cfg, seed=41, qk_gain=4.0)
    x = synth.synth_tiles(2, 112, seed=78)
    want = SO.swin_forward(sd, cfg, x)
    hf = SW.SwinConfig(image_size=112, embed_dim=128, depths=cfg["depths"], num_heads=cfg["num_heads"], num_labels=4)
    model = SW.SwinForImageClassification(hf)
    assert not model.load_state_dict(sd, strict=True).missing_keys
    for prec, tol in (("bf16x3", 1e-3), ("fp32", 2e-4)):
        out = model.to(dev).eval().set_precision(prec)(pixel_values=x.to(dev), output_hidden_states=True)
        d = (out.logits.cpu() - want["logits"]).abs().max().item()
        dh = (out.last_hidden_state.cpu() - want["last_hidden_state"]).abs().max().item()
        print(f"GPUTEST swin embed_dim 128 {prec}: logits {d:.2e}, hidden {dh:.2e}")
        assert d <= tol and dh <= 5 * tol
