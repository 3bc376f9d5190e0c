"""Parity of the HIP path (through the nn.Module surface -> C ABI -> gfx950 kernels) against
(1) the golden vectors captured from the real reference and (2) the CPU oracle on the same seeded
inputs. Needs a real MI355X.

Tolerances (default precision = "bf16x3", split-bf16 operands)
  attention probabilities : 1e-3 absolute — the bar BASELINE.json's north_star states — on EVERY golden weight
                            set whose attention is not saturated: the stress sets of all three BASELINE geometries
                            included (ViT-S/16 "peaked" 0.79, ViT-B/16 384^2 "sharp" 0.84, ViT-S/8 384^2 "peaked" 0.90;
                            measured 1-4e-4 there, <= 1e-5 elsewhere). The saturated ViT-B set (attention max 1.0000,
                            where the fp32 reference itself is 5.5e-4 from float64) has its own stated bounds.
  indices                 : bit-exact (token <-> patch mapping, nearest upsample, window origins)
  feat / qkv / tokens     : bounded relative to the tensor's own scale (stated per assert)
The single-bf16 mode ("bf16", the fastest) is held to the same 1e-3 on the init / full / sharp sets; it is not
claimed on the peaked set (rounding operands to 8 bits is amplified layer by layer to 4-8e-2 there, which is why
it is not the default), and the fp32 mode to fp32 round-off.
"""
import numpy as np
import pytest
import torch

from oracle import vit_oracle as O
from tests.golden_cases import SATURATED, STRESS
from tests.helpers import CASES, build_module, case_dims, case_inputs, case_state_dict, load_golden
import vit_ocm_wmsegmentation_amd.dino.vision_transformer as vits
from vit_ocm_wmsegmentation_amd import synth
from vit_ocm_wmsegmentation_amd import utils as amd_utils
from vit_ocm_wmsegmentation_amd.sw_processing import SlidingWindowAttention

pytestmark = pytest.mark.gpu

ATTN_TOL = 1e-3


def _attn_tol(name, mode="bf16x3"):
    """1e-3 everywhere but on the saturated set (golden_cases.SATURATED): there the softmax turns an operand rounding
    of 2^-17 into ~1e-2 (emulated on the CPU: 6e-3; measured on the device: DESIGN.md §5), and even exact-fp32 MFMA
    arithmetic in another summation order than the CPU's moves the result by ~5e-4."""
    if name in SATURATED:
        return 3e-2 if mode == "bf16x3" else 3e-3
    return ATTN_TOL


def _rel(a, b):
    """max |a-b| relative to the reference tensor's max magnitude."""
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


@pytest.mark.parametrize("name", list(CASES))
def test_golden_parity(dev, name):
    case, gold = CASES[name], load_golden(name)
    model = build_module(case, dev)
    for idx, x in enumerate(case_inputs(case)):
        pfx = f"in{idx}_"
        n = case["n"]
        feat, attns, qkvs = model.get_intermediate_feat(x.to(dev), n)
        assert len(feat) == len(attns) == len(qkvs) == n
        a = attns[-1]
        B, H, N, _ = a.shape
        assert a.dtype == torch.float32 and a.is_contiguous() and qkvs[-1].shape == (3, B, H, N, 64)
        cls_rows = a[:, :, 0, 1:].cpu().numpy()
        e_cls = np.abs(cls_rows - gold[pfx + "cls_rows"]).max()
        e_mid = np.abs(a[:, :, N // 2, :].cpu().numpy() - gold[pfx + "mid_rows"]).max()
        rs = a.sum(-1)
        print(f"\n[{name}/{idx}] attn L_inf: cls-row {e_cls:.2e} mid-row {e_mid:.2e} (attn max {float(gold[pfx + 'attn_max']):.3f}); "
              f"feat rel {_rel(feat[-1][:, :4, :16].cpu(), gold[pfx + 'feat_head']):.2e} "
              f"qkv rel {_rel(qkvs[-1][:, :, :, :3, :8].cpu(), gold[pfx + 'qkv_head']):.2e}")
        tol = _attn_tol(name)
        assert e_cls <= tol and e_mid <= tol
        assert float((rs - 1).abs().max()) < 1e-4  # rows are probabilities
        # the head-mean argmax of the HIP map must point at a (near-)maximal reference value
        hm_ref = gold[pfx + "head_mean"]
        am = a[:, :, 0, 1:].mean(1).argmax(-1).cpu().numpy()
        for b in range(B):
            assert hm_ref[b, am[b]] >= hm_ref[b].max() - 2 * tol
        # split-bf16 operands (2^-17): round-off level on the well-conditioned sets, amplified on the stress sets
        ftol = 2e-2 if name in SATURATED else 2e-3 if name in STRESS else 2e-4
        assert _rel(feat[-1][:, :4, :16].cpu(), gold[pfx + "feat_head"]) < ftol
        assert _rel(qkvs[-1][:, :, :, :3, :8].cpu(), gold[pfx + "qkv_head"]) < ftol
        assert abs(float(feat[-1].double().abs().sum()) / float(gold[pfx + "feat_abssum"]) - 1) < 1e-4
        tokens = model.prepare_tokens(x.to(dev))
        assert _rel(tokens[:, :3, :16].cpu(), gold[pfx + "tokens_head"]) < 2e-5
        assert abs(float(tokens.double().abs().sum()) / float(gold[pfx + "tokens_abssum"]) - 1) < 1e-5
        if case.get("full"):
            for j in range(n):
                assert np.abs(attns[j].cpu().numpy() - gold[pfx + f"attn{j}"]).max() <= 1e-5
                assert _rel(feat[j].cpu(), gold[pfx + f"feat{j}"]) < 2e-4
                assert _rel(qkvs[j].cpu(), gold[pfx + f"qkv{j}"]) < 2e-4
            assert _rel(tokens.cpu(), gold[pfx + "tokens"]) < 2e-5
        assert torch.equal(model.get_last_selfattention(x.to(dev)), a)


@pytest.mark.parametrize("name", [n for n in CASES if n not in STRESS])
def test_golden_parity_bf16_mode(dev, name):
    """OCM_PREC_BF16 (single bf16 operands, the fastest mode): 1e-3 on the init / full / sharp weight sets. The
    peaked set is outside what this mode claims (module docstring) and is not run here."""
    case, gold = CASES[name], load_golden(name)
    model = build_module(case, dev).set_precision("bf16")
    for idx, x in enumerate(case_inputs(case)):
        pfx = f"in{idx}_"
        n = case["n"]
        feat, attns, qkvs = model.get_intermediate_feat(x.to(dev), n)
        a = attns[-1]
        N = a.shape[-1]
        e_cls = np.abs(a[:, :, 0, 1:].cpu().numpy() - gold[pfx + "cls_rows"]).max()
        e_mid = np.abs(a[:, :, N // 2, :].cpu().numpy() - gold[pfx + "mid_rows"]).max()
        print(f"\n[bf16 {name}/{idx}] attn L_inf: cls-row {e_cls:.2e} mid-row {e_mid:.2e}")
        assert e_cls <= ATTN_TOL and e_mid <= ATTN_TOL
        assert float((a.sum(-1) - 1).abs().max()) < 1e-4
        # bf16 GEMM operands: a few 1e-3 of the tensor scale per layer, accumulated over the depth
        assert _rel(feat[-1][:, :4, :16].cpu(), gold[pfx + "feat_head"]) < 4e-2
        assert _rel(qkvs[-1][:, :, :, :3, :8].cpu(), gold[pfx + "qkv_head"]) < 4e-2
        tokens = model.prepare_tokens(x.to(dev))
        assert _rel(tokens[:, :3, :16].cpu(), gold[pfx + "tokens_head"]) < 5e-3  # one bf16 GEMM, K <= 768
        assert torch.equal(model.get_last_selfattention(x.to(dev)), a)


@pytest.mark.parametrize("name", list(CASES))
def test_golden_parity_fp32_mode(dev, name):
    """OCM_PREC_FP32 (exact-fp32 MFMA, fp32 operands end to end): every golden case — the "peaked" stress
    set included — inside the north star's 1e-3, in fact at fp32 round-off level."""
    case, gold = CASES[name], load_golden(name)
    model = build_module(case, dev).set_precision("fp32")
    for idx, x in enumerate(case_inputs(case)):
        pfx = f"in{idx}_"
        feat, attns, qkvs = model.get_intermediate_feat(x.to(dev), case["n"])
        a = attns[-1]
        N = a.shape[-1]
        e_cls = np.abs(a[:, :, 0, 1:].cpu().numpy() - gold[pfx + "cls_rows"]).max()
        e_mid = np.abs(a[:, :, N // 2, :].cpu().numpy() - gold[pfx + "mid_rows"]).max()
        r_feat = _rel(feat[-1][:, :4, :16].cpu(), gold[pfx + "feat_head"])
        r_qkv = _rel(qkvs[-1][:, :, :, :3, :8].cpu(), gold[pfx + "qkv_head"])
        print(f"\n[fp32 {name}/{idx}] attn L_inf: cls-row {e_cls:.2e} mid-row {e_mid:.2e}; feat rel {r_feat:.2e} qkv rel {r_qkv:.2e}")
        # fp32 round-off; on the stress sets the summation order of the MFMA vs the CPU's BLAS shows (fp32 vs float64 on
        # the CPU: 1e-5 .. 4e-5 there, 5.5e-4 on the saturated set)
        tol = _attn_tol(name, "fp32") if name in SATURATED else 2e-4 if name in STRESS else 2e-5
        assert e_cls <= tol and e_mid <= tol
        assert np.array_equal(a[:, :, 0, 1:].mean(1).argmax(-1).cpu().numpy(), gold[pfx + "argmax"]) or "init" in name \
            or "full" in name  # near-uniform maps may tie; the peaked / sharp maxima must match exactly
        ftol = 2e-3 if name in SATURATED else 2e-4
        assert r_feat < ftol and r_qkv < ftol
        assert torch.equal(model.get_last_selfattention(x.to(dev)), a)


@pytest.mark.parametrize("name", ["tiny_p8", "vits16_sharp", "vits16_peaked"])
def test_entry_points_agree_with_oracle(dev, name):
    """Every method of the module surface against the oracle on the same inputs (full tensors)."""
    case = CASES[name]
    model = build_module(case, dev)
    sd = case_state_dict(case)
    cfg = O.make_cfg(sd, case["patch"], case_dims(case)[2])
    x = case_inputs(case)[0]
    xg = x.to(dev)
    feat, attns, qkvs = model.get_intermediate_feat(xg, 2)
    ofeat, oattn, oqkv = O.get_intermediate_feat(sd, cfg, x, 2)
    for j in range(2):
        e = float((attns[j].cpu() - oattn[j]).abs().max())
        print(f"\n[{name}] block -{2 - j}: attn L_inf {e:.2e} feat rel {_rel(feat[j].cpu(), ofeat[j]):.2e}")
        assert e <= _attn_tol(name)
        ftol = 2e-3 if "peaked" in name else 2e-4
        assert _rel(feat[j].cpu(), ofeat[j]) < ftol and _rel(qkvs[j].cpu(), oqkv[j]) < ftol
    # get_last_selfattention == attns[-1] (bit-for-bit in the reference: SURVEY §0-3; here the same
    # kernels run on the same operands, so it is bit-exact too)
    last = model.get_last_selfattention(xg)
    assert torch.equal(last, attns[-1])
    ff = model.forward_feats(xg)
    assert torch.equal(model(xg), ff[:, 0]) and torch.equal(ff, feat[-1])
    layers = model.get_intermediate_layers(xg, 2)
    assert torch.equal(layers[0], feat[0]) and torch.equal(layers[1], feat[1])
    # CLS-row fast path == the corresponding rows of the full matrix (fp32 dot vs MFMA: tiny diff)
    qr = torch.tensor([0, 5], dtype=torch.int32, device=dev)
    rows = model.get_last_attention_rows(xg, qr)
    assert rows.shape == (x.shape[0], cfg["num_heads"], 2, last.shape[-1] - 1)
    assert float((rows - last[:, :, [0, 5], 1:]).abs().max()) < 2e-5
    # compute_attention (utils.py:229-235): same numbers, same index map, via the HIP gather kernel
    wf, hf = x.shape[-2] // case["patch"], x.shape[-1] // case["patch"]
    for query in (0, 3):
        got, nh = amd_utils.compute_attention(attns[-1:], query, wf, hf, case["patch"])
        ref, nh2 = O.compute_attention([attns[-1].cpu()], query, wf, hf, case["patch"])
        assert nh == nh2 and np.array_equal(got, ref)  # bit-exact: pure gather


def test_block_level_surface(dev):
    """model.py:45-46,132-133 call blk(x) / norm(x) / patch_embed(x) directly."""
    case = CASES["tiny_p8"]
    model = build_module(case, dev)
    sd = case_state_dict(case)
    cfg = O.make_cfg(sd, case["patch"], 2)
    x = case_inputs(case)[0]
    tok = O.prepare_tokens(sd, cfg, x)
    ref_x, ref_attn, ref_qkv = O.block(sd, cfg, 0, tok)
    got_x = model.blocks[0](tok.to(dev))
    assert _rel(got_x.cpu(), ref_x) < 1e-2
    got_x2, got_attn, got_qkv = model.blocks[0](tok.to(dev), return_qkv=True)
    assert torch.equal(got_x, got_x2)
    assert float((got_attn.cpu() - ref_attn).abs().max()) <= ATTN_TOL and _rel(got_qkv.cpu(), ref_qkv) < 1e-2
    assert torch.equal(model.blocks[0](tok.to(dev), return_attention=True), got_attn)
    assert _rel(model.norm(got_x).cpu(), O.layer_norm(sd, "norm", got_x.cpu(), 1e-6)) < 1e-5
    pe = model.patch_embed(x.to(dev))
    assert _rel(pe.cpu(), O.patch_embed(sd, x, 8)) < 5e-3
    # the free-standing sub-modules (Attention / Mlp outside a VisionTransformer, dino/vision_transformer.py:47-90) run the
    # stand-alone operators in their own `precision`: split-bf16 by default (1e-4-grade), single bf16 / fp32 on request
    blk = model.blocks[1]
    xin1, xin2 = blk.norm1(got_x), blk.norm2(got_x)
    ry, rattn, rqkv = O.attention(sd, cfg, 1, O.layer_norm(sd, "blocks.1.norm1", got_x.cpu(), 1e-6))
    rm = O.mlp(sd, 1, O.layer_norm(sd, "blocks.1.norm2", got_x.cpu(), 1e-6))
    for prec, tol in (("bf16x3", 1e-4), ("fp32", 1e-4), ("bf16", 2e-2)):
        blk.attn.precision = blk.mlp.precision = prec
        y, attn, qkv = blk.attn(xin1)
        assert _rel(y.cpu(), ry) < tol and _rel(qkv.cpu(), rqkv) < tol, prec
        assert float((attn.cpu() - rattn).abs().max()) <= (ATTN_TOL if prec == "bf16" else 1e-5), prec
        assert _rel(blk.mlp(xin2).cpu(), rm) < tol, prec
    del blk.attn.precision, blk.mlp.precision


@pytest.mark.parametrize("dim,heads", [(256, 8), (384, 3), (192, 4), (128, 1)])
def test_free_standing_attention_accepts_any_head_width(dev, dim, heads):
    """The reference's Attention (dino/vision_transformer.py:66-90) takes any dim / num_heads; the free-standing mirror runs
    64-wide heads (and 128-wide ones in split-bf16 precision) on the MFMA kernels and every other width — 32, 48, 128 here — on
    the generic fp32 attention kernel. Against the reference's arithmetic in float64 on the same parameters."""
    from vit_ocm_wmsegmentation_amd.dino.vision_transformer import Attention
    g = torch.Generator().manual_seed(dim + heads)
    B, N, hd = 2, 37, dim // heads
    m = Attention(dim, num_heads=heads, qkv_bias=True)
    with torch.no_grad():
        m.qkv.weight.copy_(torch.randn(3 * dim, dim, generator=g) * 0.08)
        m.qkv.bias.copy_(torch.randn(3 * dim, generator=g) * 0.1)
        m.proj.weight.copy_(torch.randn(dim, dim, generator=g) * 0.06)
        m.proj.bias.copy_(torch.randn(dim, generator=g) * 0.1)
    x = torch.randn(B, N, dim, generator=g)
    xd = x.double()
    qkv = (xd @ m.qkv.weight.double().t() + m.qkv.bias.double()).reshape(B, N, 3, heads, hd).permute(2, 0, 3, 1, 4)
    att = torch.softmax(qkv[0] @ qkv[1].transpose(-2, -1) * hd ** -0.5, -1)
    want = (att @ qkv[2]).transpose(1, 2).reshape(B, N, dim) @ m.proj.weight.double().t() + m.proj.bias.double()
    m = m.to(dev).eval()
    for prec, tol in (("bf16x3", 2e-4), ("fp32", 2e-5), ("bf16", 8e-2)):  # single bf16: 8 significant bits
        m.precision = prec
        y, a, q3 = m(x.to(dev))
        assert y.shape == (B, N, dim) and a.shape == (B, heads, N, N) and q3.shape == (3, B, heads, N, hd)
        ey, ea = (y.cpu().double() - want).abs().max().item(), (a.cpu().double() - att).abs().max().item()
        eq = (q3.cpu().double() - qkv).abs().max().item()
        print(f"GPUTEST free-standing Attention dim {dim} heads {heads} {prec}: y {ey:.2e} attn {ea:.2e} qkv {eq:.2e}")
        assert ey <= tol and ea <= tol and eq <= tol, prec
    with pytest.raises(ValueError):
        Attention(100, num_heads=5).to(dev)(torch.zeros(1, 4, 100, device=dev))  # head width 20: not a multiple of 8


def test_input_variants_and_state_refresh(dev):
    case = CASES["tiny_p8"]
    model = build_module(case, dev)
    x = case_inputs(case)[0].to(dev)
    base = model.get_last_selfattention(x)
    # non-contiguous batch view, float64 input, channel-expanded grayscale view
    big = torch.zeros(4, 3, 32, 32, device=dev)
    big[::2] = x
    assert torch.equal(model.get_last_selfattention(big[::2]), base)
    assert torch.equal(model.get_last_selfattention(x.double()), base)
    gray = x[:, :1].expand(-1, 3, -1, -1)  # stride_c == 0
    assert torch.equal(model.get_last_selfattention(gray), base)
    # grayscale fold: one plane + W.sum(dim=1)
    model.enable_grayscale_fold(True)
    folded = model.get_last_selfattention(x)
    folded1 = model.get_last_selfattention(x[:, :1].contiguous())
    assert torch.equal(folded, folded1) and float((folded - base).abs().max()) < ATTN_TOL
    model.enable_grayscale_fold(False)
    # parameters changed through load_state_dict are picked up by the engine
    sd2 = synth.synth_state_dict(128, 2, 8, seed=99, variant="full", img_size=32)
    model.load_state_dict(sd2)
    cfg = O.make_cfg(sd2, 8, 2)
    ref = O.get_last_selfattention(sd2, cfg, x.cpu())
    assert float((model.get_last_selfattention(x).cpu() - ref).abs().max()) <= ATTN_TOL
    with pytest.raises(ValueError):
        model.get_last_selfattention(torch.zeros(1, 3, 30, 32, device=dev))  # not a multiple of p
    with pytest.raises(RuntimeError, match="HIP"):
        model.get_last_selfattention(x.cpu())  # no CPU fallback


def test_sliding_window_single_rank(dev):
    """The batched, origin-gathered sweep == the reference's serial B=1 crop loop
    (sw_processing.py:151-163, 235-245) computed by the oracle on materialised crops."""
    case = CASES["tiny_p8"]
    model = build_module(case, dev)
    sd = case_state_dict(case)
    cfg = O.make_cfg(sd, 8, 2)
    window, stride, size = 96, 32, 160  # 3 x 3 windows, window = 3 * stride as in the reference
    slab = synth.synth_tiles(1, size, seed=5)[0]
    crops = O.sliding_window_crops(slab, stride, window)
    assert crops.shape[0] == 9
    ref = O.tile_head_mean_maps(sd, cfg, crops, 8)  # (9, 96, 96): nearest-upsampled head means
    sweep = SlidingWindowAttention(model, window=window, stride=stride, batch_tiles=4)
    maps = sweep(slab.to(dev))  # (9, H, 1, 12, 12)
    assert maps.shape == (9, 2, 1, 12, 12)
    got = maps[:, :, 0].mean(1)  # head mean, (9, 12, 12)
    ref_small = torch.from_numpy(ref)[:, ::8, ::8]  # undo the nearest x8 upsample
    assert float((got.cpu() - ref_small).abs().max()) <= ATTN_TOL
    # and via the full-matrix route of the module on materialised crops
    full = model.get_last_selfattention(crops.to(dev))[:, :, 0, 1:].reshape(9, 2, 12, 12)
    assert float((maps[:, :, 0] - full).abs().max()) < 2e-5


@pytest.mark.parametrize("name", ["tiny_p8", "vits16_peaked"])
@pytest.mark.parametrize("precision", ["bf16x3", "bf16", "fp32"])
def test_fused_gemm_layernorm_is_bit_identical(dev, lib, name, precision):
    """proj / fc2 + residual + the following LayerNorm in one kernel (engine default for T >= 8192 rows, forced here
    through the per-handle option OCM_OPT_FUSE_LN) against the separate GEMM and LayerNorm launches: same accumulation
    order, same two-pass statistics on the same fp32 values -> identical bits in every output."""
    case = CASES[name]
    model = build_module(case, dev).set_precision(precision)
    x = case_inputs(case)[0].to(dev)
    eng = model._engine(dev)
    try:
        eng.set_fold_layernorm(False)  # split-bf16 engines fold the LayerNorm into its consumer by default: not this test's subject
        eng.set_fuse_layernorm("never")
        f0, a0, q0 = model.get_intermediate_feat(x, 2)
        l0 = model.get_last_selfattention(x)
        eng.set_fuse_layernorm("always")
        f1, a1, q1 = model.get_intermediate_feat(x, 2)
        l1 = model.get_last_selfattention(x)
    finally:
        eng.set_fuse_layernorm("auto")
        eng.set_fold_layernorm(True)
    with pytest.raises(ValueError):
        from vit_ocm_wmsegmentation_amd import _lib as L
        L.check(lib.ocm_vit_set_option(eng._h, 7, 0))  # unknown option
    assert not hasattr(lib, "ocm_debug_knob") or "OCM_VIT_LIB" in __import__("os").environ  # product build: no knobs
    for u, v in zip(f0 + a0 + q0 + [l0], f1 + a1 + q1 + [l1]):
        assert torch.equal(u, v)


@pytest.mark.parametrize("name", ["tiny_p8", "vits16_peaked"])
def test_folded_layernorm_matches_layernorm_kernels(dev, name):
    """Split-bf16 default: every LayerNorm is folded into the GEMM that consumes it (un-normalised split operands + row
    sums, rstd * (acc - mu c) + d in the epilogue; Block.forward dino/vision_transformer.py:107,111). Against the same
    engine with LayerNorm kernels: the same arithmetic up to operand rounding (2^-17), and — no atomics in the row sums —
    the same bits from call to call."""
    case = CASES[name]
    model = build_module(case, dev)
    x = case_inputs(case)[0].to(dev)
    eng = model._engine(dev)
    f1, a1, q1 = model.get_intermediate_feat(x, 2)
    f1b, a1b, q1b = model.get_intermediate_feat(x, 2)
    for u, v in zip(f1 + a1 + q1, f1b + a1b + q1b):
        assert torch.equal(u, v)
    # the third call would replay the capture of the folded launch sequence if the replay key ignored the option
    # (ADVICE r3): the setters bump Engine.option_epoch, which is part of the key — and the two paths must differ in
    # at least one bit, or this test compares folded against folded
    n_graphs = len(model.__dict__.get("_auto_graphs", {}))
    try:
        eng.set_fold_layernorm(False)
        f0, a0, q0 = model.get_intermediate_feat(x, 2)
        assert len(model.__dict__.get("_auto_graphs", {})) == n_graphs  # first sighting of the new key: launch by launch
    finally:
        eng.set_fold_layernorm(True)
    assert sum(int((u != v).sum()) for u, v in zip(f0 + a0 + q0, f1 + a1 + q1)) > 0, "LayerNorm kernels and the fold gave the same bits"
    tol = 5e-4 if name in STRESS else 1e-6
    for u, v in zip(a0, a1):
        assert float((u - v).abs().max()) <= tol
    for u, v in zip(f0 + q0, f1 + q1):
        assert _rel(u.cpu(), v.cpu()) < (2e-3 if name in STRESS else 1e-4)
    # a LayerNorm parameter changed in place is picked up (the folded weights are rebuilt before the next forward)
    with torch.no_grad():
        model.blocks[0].norm1.weight.mul_(1.25)
        model.blocks[1].norm2.bias.add_(0.1)
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    cfg = O.make_cfg(sd, case["patch"], case_dims(case)[2])
    ref = O.get_last_selfattention(sd, cfg, x.cpu())
    assert float((model.get_last_selfattention(x).cpu() - ref).abs().max()) <= ATTN_TOL


@pytest.mark.parametrize("size,stride,window", [(200, 32, 96), (160, 32, 128), (130, 32, 96)])
def test_sliding_window_past_the_slab_edge(dev, size, stride, window):
    """Slabs whose side is not a multiple of the stride / windows wider than 3 strides: the last windows reach past the
    image and the reference's PIL crop zero-fills them (sw_processing.py:157-160). The device gather must see the same
    pixels (it used to read past the slab)."""
    case = CASES["tiny_p8"]
    model = build_module(case, dev)
    sd = case_state_dict(case)
    cfg = O.make_cfg(sd, 8, 2)
    slab = synth.synth_tiles(1, size, seed=7)[0]
    crops = O.sliding_window_crops(slab, stride, window)
    sweep = SlidingWindowAttention(model, window=window, stride=stride, batch_tiles=5)
    maps = sweep(slab.to(dev))
    hf = window // 8
    assert maps.shape == (crops.shape[0], 2, 1, hf, hf)
    ref = O.get_last_selfattention(sd, cfg, crops)[:, :, 0, 1:].reshape(-1, 2, hf, hf)
    assert float((maps[:, :, 0].cpu() - ref).abs().max()) <= 1e-5


@pytest.mark.gpu
def test_hip_graph_replay_matches_plain_launches(dev):
    """OCM_USE_GRAPH: the one-tile-per-call loop of the reference's scripts replays a cached hipGraph; results are
    bit-identical to plain launches, a pointer / shape change re-captures, and weights can change under the graph."""
    case = CASES["tiny_p8"]
    model = build_module(case, dev)
    model.auto_graph = False  # the engine-level cache on its own (the module's automatic replay has its own test below)
    eng = model._engine(dev)
    x1 = case_inputs(case)[0][:1].to(dev)
    x2 = (x1 * 0.5 + 0.1).contiguous()
    eng.hip_graph = False
    want1 = model.get_last_selfattention(x1).clone()
    want2 = model.get_last_selfattention(x2).clone()
    eng.hip_graph = True
    r0, c0 = eng.graph_stats()
    outs = []
    for i in range(6):  # same input tensor, outputs freed between calls: pointers repeat -> replays
        a = model.get_last_selfattention(x1)
        outs.append(a.clone())
        del a
    for o in outs:
        assert torch.equal(o, want1)
    r1, c1 = eng.graph_stats()
    assert c1 - c0 >= 1 and r1 - r0 >= 1, (r0, c0, r1, c1)  # the caching allocator hands the same blocks back
    assert torch.equal(model.get_last_selfattention(x2), want2)  # new input pointer: re-capture, same numbers
    feat, attn, qkv = model.get_intermediate_feat(x1, n=1)  # other flags / outputs: re-capture
    assert torch.equal(attn[0], want1)
    # parameters updated in place are picked up by the replayed graph (the graph holds pointers, not values)
    with torch.no_grad():
        model.blocks[0].attn.qkv.weight.mul_(1.5)
    changed = model.get_last_selfattention(x1).clone()
    eng2 = model._engine(dev)
    eng2.hip_graph = False
    assert torch.equal(model.get_last_selfattention(x1), changed)
    assert not torch.equal(changed, want1)


def test_key_split_attention_agrees_across_batch_sizes(dev):
    """One ViT-S/8 window of 384^2 per call cuts the key range of its attention into four slices per workgroup (two at B = 2,
    three at 120 workgroups) and merges them; five windows per call run unsplit. The same tiles give the same maps either way
    (summation order differs: fp32 rounding), and the returned matrix equals get_intermediate_feat's bit for bit."""
    from vit_ocm_wmsegmentation_amd import synth
    import vit_ocm_wmsegmentation_amd.dino.vision_transformer as vits
    x = synth.synth_tiles(5, 384, seed=77).to(dev)
    # "full": well-conditioned weights, the two orders agree to fp32 rounding; "qkv10": the trained-like stress set (attention max
    # 0.9), where a last-bit difference in a block's context (P is split into pairs relative to a slice's own running maximum)
    # is amplified like any other rounding — both results are 1e-4 from the reference, 5e-5 from each other
    for variant, tol in (("full", 2e-6), ("qkv10", 2e-4)):
        model = vits.vit_small(patch_size=8, num_classes=0)
        model.load_state_dict(synth.synth_arch_state_dict("vit_small", 8, seed=0, variant=variant))
        model = model.eval().to(dev)
        model.auto_graph = False
        ref = model.get_last_selfattention(x)  # 300 workgroups per attention launch: no key split
        ref_feat = model.get_intermediate_feat(x, 1)[0][0]
        for lo, hi in ((0, 1), (1, 3), (4, 5)):  # 1 tile: four slices; 2 tiles: two slices
            got = model.get_last_selfattention(x[lo:hi])
            err = float((got - ref[lo:hi]).abs().max())
            print(f"GPUTEST key-split attention {variant} tiles {lo}:{hi}: max|d| vs the unsplit batch = {err:.2e}")
            assert err < tol, (variant, lo, hi)
            feat, attn, _ = model.get_intermediate_feat(x[lo:hi], 1)
            assert torch.equal(attn[0], got)
            assert float((feat[0] - ref_feat[lo:hi]).abs().max()) < 1e-3 * float(ref_feat.abs().max())


def test_one_tile_calls_replay_a_graph_automatically(dev):
    """Plain one-tile calls (the reference's loops, eval.py:126-171) are replayed as a HIP graph by the module itself:
    same bits as the launch-by-launch path, new inputs followed, a parameter update or precision change re-captured,
    larger batches left alone."""
    case = CASES["vits16_peaked"]
    model = build_module(case, dev)
    x = case_inputs(case)[0][:1].to(dev)
    x2 = torch.flip(x, dims=(-1,)).contiguous()
    model.auto_graph = False
    want = [model.get_last_selfattention(x), model.get_last_selfattention(x2)]
    feat_want = model.get_intermediate_feat(x, 2)
    rows_want = model.get_last_attention_rows(x, torch.tensor([0, 5], dtype=torch.int32, device=dev))
    model.auto_graph = True
    assert torch.equal(model.get_last_selfattention(x), want[0])  # first sighting of this (shape, outputs): launch by launch
    assert len(model.__dict__.get("_auto_graphs", {})) == 0
    for _ in range(3):  # second sighting captures, then replays
        assert torch.equal(model.get_last_selfattention(x), want[0])
        assert torch.equal(model.get_last_selfattention(x2), want[1])
    assert len(model._auto_graphs) == 1
    for _ in range(3):
        got = model.get_intermediate_feat(x, 2)
    assert len(model._auto_graphs) == 2
    for a, b in zip(feat_want, got):
        for u, v in zip(a, b):
            assert torch.equal(u, v)
    qr = torch.tensor([0, 5], dtype=torch.int32, device=dev)
    for _ in range(3):
        assert torch.equal(model.get_last_attention_rows(x, qr), rows_want)
    n_graphs = len(model._auto_graphs)
    for _ in range(3):  # a new index tensor per call is never captured
        assert torch.equal(model.get_last_attention_rows(x, torch.tensor([0, 5], dtype=torch.int32, device=dev)), rows_want)
    assert len(model._auto_graphs) == n_graphs
    # returned tensors are copies: a later call does not overwrite them
    a = model.get_last_selfattention(x)
    b = model.get_last_selfattention(x2)
    assert torch.equal(a, want[0]) and torch.equal(b, want[1]) and a.data_ptr() != b.data_ptr()
    # in-place parameter update: the next call sees it
    with torch.no_grad():
        model.blocks[0].attn.qkv.weight.mul_(1.5)
    changed = model.get_last_selfattention(x)
    model.auto_graph = False
    assert torch.equal(changed, model.get_last_selfattention(x)) and not torch.equal(changed, want[0])
    model.auto_graph = True
    # another precision: another engine, another capture
    a32 = model.set_precision("fp32").get_last_selfattention(x)
    model.auto_graph = False
    assert torch.equal(a32, model.get_last_selfattention(x))
    model.auto_graph = True
    # batches beyond AUTO_GRAPH_TOKENS rows run launch by launch
    n_before = len(model._auto_graphs)
    model.get_last_selfattention(torch.cat([x] * 6))
    assert len(model._auto_graphs) == n_before


@pytest.mark.parametrize("method,kwargs", [("get_last_selfattention", {}), ("get_intermediate_feat", {"n": 2}),
                                           ("get_last_attention_rows", {}), ("forward", {})])
def test_graphed_call_replays_the_same_kernels(dev, method, kwargs):
    """model.graphed(method): HIP-graph replay of a one-tile call (what the reference's loops issue) returns bit for
    bit what the plain call returns, follows new inputs, and re-captures on a precision or shape change."""
    def same(a, b):
        if isinstance(a, torch.Tensor):
            return torch.equal(a, b)
        return len(a) == len(b) and all(same(x, y) for x, y in zip(a, b))

    model = vits.vit_small(patch_size=16, num_classes=0)
    model.load_state_dict(synth.synth_arch_state_dict("vit_small", 16, seed=3, variant="sharp"))
    model = model.eval().to(dev)
    run = model.graphed(method, **kwargs)
    x1, x2 = synth.synth_tiles(1, 224, seed=11).to(dev), synth.synth_tiles(1, 224, seed=12).to(dev)
    for x in (x1, x2, x1):
        assert same(run(x), getattr(model, method)(x, **kwargs))
    first = run(x1)
    run(x2)  # copies are returned: an earlier result does not change under a later call
    assert same(first, getattr(model, method)(x1, **kwargs))
    model.set_precision("bf16")
    assert same(run(x2), getattr(model, method)(x2, **kwargs))
    model.set_precision("bf16x3")
    x3 = synth.synth_tiles(2, 96, seed=13).to(dev)  # another batch and tile size: re-captured
    assert same(run(x3), getattr(model, method)(x3, **kwargs))
    with pytest.raises(AttributeError):
        model.graphed("no_such_method")


@pytest.mark.gpu
@pytest.mark.parametrize("ratio", [0.0, 3.0, 30.0])
def test_folded_layernorm_on_rows_with_a_large_mean(dev, ratio):
    """the folded LayerNorm multiplies UN-normalised rows, and every golden / stress fixture has a zero-mean residual
    stream. Here the stream gets a common offset of `ratio` standard deviations (cls_token and pos_embed shifted: Block.forward
    dino/vision_transformer.py:107,111 sees x = tokens + offset) and the LayerNorm parameters a wide dynamic range (gamma
    0.22 .. 4.5, beta +-1: attention max 0.43 on the qkv x4 weight set; wider ranges saturate the softmax to one-hot rows and test
    conditioning instead). LayerNorm does not see the offset, but the split-bf16 rounding of the operand does: before round 4's
    row centring (csrc/launch.h: the pairs and sums a producer writes are those of x minus the row's mean at the previous
    LayerNorm site) this test measured folded-vs-oracle 5e-5 / 9e-4 / 5e-3 at 0 / 3 / 30 sigma against 8e-5 / 7e-5 / 1e-4 for the
    LayerNorm kernels; with it 8e-5 / 7e-5 / 1.3e-4. The folded engine, the same engine with LayerNorm kernels and the fp32 CPU
    oracle must agree on the attention maps at every offset."""
    case = CASES["vits16_sharp"]
    sd = case_state_dict(case)
    cfg = O.make_cfg(sd, case["patch"], case_dims(case)[2])
    x = case_inputs(case)[0][:2]
    sigma = float(O.prepare_tokens(sd, cfg, x).std())
    g = torch.Generator().manual_seed(5)
    sd = {k: v.clone() for k, v in sd.items()}
    sd["cls_token"] += ratio * sigma
    sd["pos_embed"] += ratio * sigma
    for k in sd:
        if ".norm1.weight" in k or ".norm2.weight" in k:
            sd[k] = torch.exp(torch.empty_like(sd[k]).uniform_(-1.5, 1.5, generator=g))
        elif ".norm1.bias" in k or ".norm2.bias" in k:
            sd[k] = torch.empty_like(sd[k]).uniform_(-1.0, 1.0, generator=g)
    model = vits.vit_small(patch_size=case["patch"], num_classes=0)
    model.load_state_dict(sd)
    model = model.eval().to(dev)
    model.auto_graph = False
    ref = O.get_last_selfattention(sd, cfg, x)
    eng = model._engine(dev)
    folded = model.get_last_selfattention(x.to(dev)).cpu()
    one = model.get_last_selfattention(x[:1].to(dev)).cpu()  # the one-tile dispatch: split-K finish as the producer
    try:
        eng.set_fold_layernorm(False)
        kernels = model.get_last_selfattention(x.to(dev)).cpu()
    finally:
        eng.set_fold_layernorm(True)
    e_f, e_k = float((folded - ref).abs().max()), float((kernels - ref).abs().max())
    e_1 = float((one - ref[:1]).abs().max())
    print(f"\n[fold, row mean = {ratio:g} sigma] attention max {float(ref.max()):.3f}: folded vs oracle {e_f:.2e} (one tile per call "
          f"{e_1:.2e}), LayerNorm kernels vs oracle {e_k:.2e}, folded vs kernels {float((folded - kernels).abs().max()):.2e}")
    assert e_k <= 5e-4 and e_f <= 5e-4 and e_1 <= 5e-4
    assert torch.isfinite(folded).all() and float((folded.sum(-1) - 1).abs().max()) < 1e-4
