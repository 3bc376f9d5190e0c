"""Kernel-level parity of the OCM_PREC_BF16X3 operators (split-bf16 pairs, three bf16 MFMAs per product) against
float64 references: a split operand carries ~17 bits, the dropped lo*lo term is 2^-18 relative, so results must sit at
a few 1e-6 of the data scale — 250x inside what a single bf16 operand (2^-9) could reach. Runs on a real MI355X only."""
import ctypes as C
import math

import pytest
import torch
import torch.nn.functional as F

from vit_ocm_wmsegmentation_amd import _lib
from vit_ocm_wmsegmentation_amd.engine import from_split, to_operand

pytestmark = pytest.mark.gpu
X3 = _lib.OCM_PREC_BF16X3


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _s():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ok(lib, rc):
    assert rc == 0, lib.ocm_last_error().decode()


def _rand(shape, dev, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(dev)


def test_split_roundtrip_and_layout(lib, dev):
    x = _rand((37, 96), dev, 1, 3.0)
    xs = to_operand(x, X3)
    assert xs.dtype == torch.int32 and xs.shape == x.shape
    back = from_split(xs)
    assert ((back - x).abs() <= x.abs() * 2 ** -16).all()  # hi + lo carries >= 16 bits
    # layout: every 32 elements are [32 x bf16 hi | 32 x bf16 lo]
    raw = xs.view(torch.bfloat16).reshape(37, 3, 2, 32)
    hi = x.to(torch.bfloat16).reshape(37, 3, 32)
    assert torch.equal(raw[:, :, 0], hi)
    lo = (x - hi.reshape(37, 96).float()).to(torch.bfloat16).reshape(37, 3, 32)
    assert torch.equal(raw[:, :, 1], lo)
    with pytest.raises(ValueError):
        to_operand(x[:, :40], X3)


@pytest.mark.parametrize("rows,dim", [(1000, 384), (7, 192), (513, 768), (65, 160)])
def test_layernorm_split_out(lib, dev, rows, dim):
    x = _rand((rows, dim), dev, 1, 3.0) + 0.5
    g = _rand((dim,), dev, 2) * 0.1 + 1
    b = _rand((dim,), dev, 3) * 0.1
    ref = F.layer_norm(x.double(), (dim,), g.double(), b.double(), 1e-6)
    ys = torch.empty((rows, dim), dtype=torch.int32, device=dev)
    _ok(lib, lib.ocm_op_layernorm(_p(x), _p(g), _p(b), _p(ys), _lib.OCM_LN_SPLIT, rows, dim, 1e-6, _s()))
    err = (from_split(ys).double() - ref).abs()
    assert (err <= ref.abs() * 2 ** -16 + 2e-5).all(), err.max().item()  # fp32 LayerNorm, then 2^-17 pairs


# The shapes reach every split-bf16 GEMM the BASELINE configurations dispatch (gemm_kernels.h::launch_linear_epi):
#   (12608, 1536, 384)  config 2 fc1: 1188 tiles, N >= 1024 -> gemm_dma_kernel<128x128, eight waves>, two workgroups per CU; with an
#                       activation output (epilogues 2 / 3) 948 tiles of 160 x 128 since round 4 (GemmCfg::HALF: a 16-row half tile
#                       per wave row) — (12609, ...) leaves ONE valid row in the last block's half band, (20000, 1024, 384) is
#                       125 full blocks of 160 rows
#   (12608, 384, *)     config 2 proj / fc2: gemm_dma_kernel<128x192> on a three-stage ring
#   (32768, 1024, 768)  config 3: 512 tiles of 256x256 -> gemm_dma_kernel<256x256>, banded epilogue
#   (24576, 384, 1536)  config 4 fc2 (48 K steps) -> gemm_dma_kernel<128x192> as well since round 3
#   (16384, 512, 384)   512 tiles of 128x128, N < 1024 and not a multiple of 192 -> gemm_dma_kernel<128x128, four waves>
#                       (config 4's fc1 takes it at 48 k rows: tests/test_bench_configs_gpu.py)
#   (1000 / 333 / 70 / 64 rows): the one-tile-per-call DMA kernels (64x128 on eight waves, 64x64 four-stage) and tails
#   (6000, 288 / 96, 96), (5000, 576 / 192, *): Swin-T's narrow stages in split-bf16 mode -> gemm_dma_kernel<128x96> / <128x192>
@pytest.mark.parametrize("M,N,K", [(1000, 384, 384), (12608, 1536, 384), (333, 384, 1536), (70, 96, 192),
                                   (12608, 384, 384), (64, 192, 64), (32768, 1024, 768), (24576, 384, 1536),
                                   (6000, 288, 96), (6000, 96, 384), (5000, 576, 192), (5000, 192, 768), (16384, 512, 384),
                                   (12609, 1536, 384), (20000, 1024, 384)])
@pytest.mark.parametrize("epi", [0, 1, 2, 3])
def test_linear_x3(lib, dev, M, N, K, epi):
    a = _rand((M, K), dev, 40)
    w = _rand((N, K), dev, 41, 0.05)
    bias = _rand((N,), dev, 42, 0.1)
    resid = _rand((M, N), dev, 43)
    ref = a.double() @ w.double().t() + bias.double()
    if epi == 1:
        ref = ref + resid.double()
    if epi == 2:
        ref = F.gelu(ref)
    act_out = epi in (2, 3)
    out = torch.full((M, N), float("nan"), dtype=torch.float32, device=dev)
    if act_out:
        out = out.view(torch.int32)
    if epi == 1:
        out.copy_(resid)
    a_s, w_s = to_operand(a, X3), to_operand(w, X3)  # named: the operands must outlive the call
    _ok(lib, lib.ocm_op_linear(X3, _p(a_s), _p(w_s), _p(bias), _p(out) if epi == 1 else None, _p(out), M, N, K, epi, _s()))
    got = from_split(out) if act_out else out
    # operands carry 2^-17, fp32 accumulation over K terms of |a||w| ~ 0.05
    assert (got.double() - ref).abs().max().item() < 3e-5 * max(1.0, math.sqrt(K) / 8)


# (64, 197, 6): config 2, 891 tiles of 128x128 -> qkv_dma_kernel<Cfg128x128q> (8 waves); (26, 577, 12): ViT-B rows enough
# for big_tiles_pay -> qkv_dma_kernel<Cfg256x256> (config 3's kernel); the others: 64x128 DMA (M <= 1024) and tails
@pytest.mark.parametrize("B,N,H", [(3, 197, 6), (2, 17, 2), (1, 577, 12), (5, 50, 3), (64, 197, 6), (26, 577, 12)])
def test_qkv_proj_x3(lib, dev, B, N, H):
    D = H * 64
    a, w, bias = _rand((B * N, D), dev, 50), _rand((3 * D, D), dev, 51, 0.05), _rand((3 * D,), dev, 52, 0.1)
    npad = lib.ocm_n_pad_prec(X3, N)
    assert npad % 32 == 0 and npad >= N
    q = torch.zeros((B * H, npad, 64), dtype=torch.int32, device=dev)
    k = torch.zeros_like(q)
    vt = torch.zeros((B * H, 64, npad), dtype=torch.int32, device=dev)
    qkv32 = torch.empty((3, B, H, N, 64), device=dev)
    a_s, w_s = to_operand(a, X3), to_operand(w, X3)
    _ok(lib, lib.ocm_op_qkv_proj(X3, _p(a_s), _p(w_s), _p(bias), _p(q), _p(k), _p(vt), _p(qkv32), B, N, H, _s()))
    ref = (a.double() @ w.double().t() + bias.double()).reshape(B, N, 3, H, 64).permute(2, 0, 3, 1, 4)
    tol = 3e-5 * max(1.0, math.sqrt(D) / 8)
    assert (qkv32.double() - ref).abs().max().item() < tol
    assert (from_split(q)[:, :N].double() - ref[0].reshape(B * H, N, 64)).abs().max().item() < tol
    assert (from_split(k)[:, :N].double() - ref[1].reshape(B * H, N, 64)).abs().max().item() < tol
    assert (from_split(vt)[:, :, :N].double() - ref[2].reshape(B * H, N, 64).transpose(1, 2)).abs().max().item() < tol
    assert (q[:, N:] == 0).all()  # padding rows are never written
    # V^T padding columns share 128-byte groups with valid keys but are never written either
    assert (from_split(vt)[:, :, N:] == 0).all()


@pytest.mark.parametrize("B,N,H", [(2, 197, 6), (1, 17, 2), (1, 577, 3), (1, 64, 1), (1, 65, 1), (1, 2305, 2),
                                   (1, 5, 1), (1, 32, 2), (1, 33, 1), (1, 1024, 1), (1, 1025, 1),  # one tile, no padding,
                                   # one key into the second tile, the 4-wave / 8-wave switch
                                   (64, 197, 6),   # config 2's launch: 768 workgroups through xcd_remap2, EVERY (b, h) compared
                                   (3, 2305, 6)])  # config 4's shape per window, several images: batch strides of the 8-wave kernel
@pytest.mark.parametrize("sharp", [1.0, 3.0])
def test_attention_x3(lib, dev, B, N, H, sharp):
    g = torch.Generator().manual_seed(60)
    q = (torch.randn((B * H, N, 64), generator=g) * sharp).to(dev)
    k = (torch.randn((B * H, N, 64), generator=g) * sharp).to(dev)
    v = torch.randn((B * H, N, 64), generator=g).to(dev)
    npad = lib.ocm_n_pad_prec(X3, N)
    nan = float("nan")  # padding may hold anything, NaN included
    qp = torch.full((B * H, npad, 64), nan, device=dev)
    kp = torch.full((B * H, npad, 64), nan, device=dev)
    vp = torch.full((B * H, 64, npad), nan, device=dev)
    qp[:, :N], kp[:, :N], vp[:, :, :N] = q, k, v.transpose(1, 2)
    qs, ks, vs = to_operand(qp, X3), to_operand(kp, X3), to_operand(vp, X3)
    scale = 0.125
    s = (q.double() @ k.double().transpose(1, 2)) * scale
    pref = s.softmax(-1)
    oref = (pref @ v.double()).reshape(B, H, N, 64).permute(0, 2, 1, 3).reshape(B, N, H * 64)
    ctx = torch.full((B, N, H * 64), nan, device=dev).view(torch.int32)
    lse = torch.empty((B * H, N), device=dev)
    _ok(lib, lib.ocm_op_attention(X3, _p(qs), _p(ks), _p(vs), _p(ctx), _p(lse), B, N, H, scale, _s()))
    # a score is a sum of 64 products of 2^-17-accurate operands of size ~sharp: its absolute error, and with it the
    # relative error of every probability, grows with sharp^2 (1e-5 .. 1e-4 here; single bf16: 256x that)
    es = sharp * sharp
    assert (lse.double() - torch.logsumexp(s, -1) / math.log(2.0)).abs().max().item() < 1e-4 * es
    assert (from_split(ctx).double() - oref).abs().max().item() < 4e-5 * es
    attn = torch.full((B, H, N, N), nan, device=dev)
    _ok(lib, lib.ocm_op_attention_probs(X3, _p(qs), _p(ks), _p(lse), _p(attn), B, N, H, scale, _s()))
    assert (attn.reshape(B * H, N, N).double() - pref).abs().max().item() < 2e-5 * es
    assert (attn.sum(-1) - 1).abs().max().item() < 1e-4
    lse2 = torch.empty_like(lse)
    _ok(lib, lib.ocm_op_attention(X3, _p(qs), _p(ks), _p(vs), None, _p(lse2), B, N, H, scale, _s()))
    assert torch.equal(lse, lse2)
    rows_idx = torch.tensor([0, N - 1, N // 2], dtype=torch.int32, device=dev)
    rows = torch.empty((B, H, 3, N - 1), device=dev)
    _ok(lib, lib.ocm_op_attention_rows(X3, _p(qs), _p(ks), _p(rows_idx), 3, _p(rows), B, N, H, scale, _s()))
    assert (rows.double() - pref.reshape(B, H, N, N)[:, :, rows_idx.long(), 1:]).abs().max().item() < 2e-5 * es


@pytest.mark.parametrize("B,N,H,spike_at", [(2, 197, 3, 100), (1, 2305, 2, 1500), (2, 577, 2, 40), (1, 197, 1, 0)])
def test_attention_x3_deferred_max_is_exact_when_the_maximum_jumps(lib, dev, B, N, H, spike_at):
    """The flash kernels move a row's reference point only when some row of the wavefront found a score more than 8 (log2
    units) above its own (kernels_attn.hip: defer_max_update; cdna_hip_programming.md T13 and its rule 26). The rare branch —
    a rescale after many deferred tiles — needs an input that forces it: scores that stay within the threshold for every tile
    (|q|, |k| ~ 0.5: the maximum creeps, nothing rescales) except ONE key whose row is a large multiple of a query direction, so
    that at its tile the maximum of most rows jumps by tens of units and everything accumulated against the stale reference
    (O, l) must be rescaled exactly once. Against float64 on the FULL tensor: context, log-sum-exp and probabilities."""
    g = torch.Generator().manual_seed(61)
    q = (torch.randn((B * H, N, 64), generator=g) * 0.5)
    k = (torch.randn((B * H, N, 64), generator=g) * 0.5)
    v = torch.randn((B * H, N, 64), generator=g)
    k[:, spike_at] = 6.0 * q[:, N // 2] + 3.0 * q[:, 1]  # scores of ~100 against those queries, large against their neighbours
    q, k, v = q.to(dev), k.to(dev), v.to(dev)
    npad = lib.ocm_n_pad_prec(X3, N)
    nan = float("nan")
    qp = torch.full((B * H, npad, 64), nan, device=dev)
    kp = torch.full((B * H, npad, 64), nan, device=dev)
    vp = torch.full((B * H, 64, npad), nan, device=dev)
    qp[:, :N], kp[:, :N], vp[:, :, :N] = q, k, v.transpose(1, 2)
    qs, ks, vs = to_operand(qp, X3), to_operand(kp, X3), to_operand(vp, X3)
    scale = 0.125
    s = (q.double() @ k.double().transpose(1, 2)) * scale
    assert float((s.max(-1).values - s.median(-1).values).max()) * math.log2(math.e) > 16  # the jump is far past the threshold
    pref = s.softmax(-1)
    oref = (pref @ v.double()).reshape(B, H, N, 64).permute(0, 2, 1, 3).reshape(B, N, H * 64)
    ctx = torch.full((B, N, H * 64), nan, device=dev).view(torch.int32)
    lse = torch.empty((B * H, N), device=dev)
    _ok(lib, lib.ocm_op_attention(X3, _p(qs), _p(ks), _p(vs), _p(ctx), _p(lse), B, N, H, scale, _s()))
    es = float(s.abs().max()) / 8  # operand rounding is relative: the score error scales with the largest score
    assert (lse.double() - torch.logsumexp(s, -1) / math.log(2.0)).abs().max().item() < 2e-4 * max(1.0, es)
    got = from_split(ctx).double()
    assert torch.isfinite(got).all()
    assert (got - oref).abs().max().item() < 1e-4 * max(1.0, es)
    attn = torch.full((B, H, N, N), nan, device=dev)
    _ok(lib, lib.ocm_op_attention_probs(X3, _p(qs), _p(ks), _p(lse), _p(attn), B, N, H, scale, _s()))
    assert (attn.reshape(B * H, N, N).double() - pref).abs().max().item() < 1e-4 * max(1.0, es)
    assert (attn.sum(-1) - 1).abs().max().item() < 1e-4


@pytest.mark.parametrize("B,N,H", [(2, 65, 3), (1, 197, 3), (3, 300, 2), (1, 32, 1)])
@pytest.mark.parametrize("sharp", [1.0, 2.0])
def test_attention_x3_128_wide_heads(lib, dev, B, N, H, sharp):
    """The encoder the reference's build_model() constructs (model.py:93-103) has 3 heads of 128 channels: qkv projection,
    attention and probabilities on the split-bf16 MFMA kernels templated on the head width, against float64 torch."""
    HD, D = 128, H * 128
    a, w, bias = _rand((B * N, D), dev, 70), _rand((3 * D, D), dev, 71, 0.05 * sharp), _rand((3 * D,), dev, 72, 0.1)
    npad = lib.ocm_n_pad_prec(X3, N)
    q = torch.full((B * H, npad, HD), -1, dtype=torch.int32, device=dev)  # 0xFFFF pairs: NaN wherever nothing is written
    k = torch.full_like(q, -1)
    vt = torch.full((B * H, HD, npad), -1, dtype=torch.int32, device=dev)
    qkv32 = torch.empty((3, B, H, N, HD), device=dev)
    a_s, w_s = to_operand(a, X3), to_operand(w, X3)
    _ok(lib, lib.ocm_op_qkv_proj_hd(X3, _p(a_s), _p(w_s), _p(bias), _p(q), _p(k), _p(vt), _p(qkv32), B, N, H, HD, _s()))
    ref = (a.double() @ w.double().t() + bias.double()).reshape(B, N, 3, H, HD).permute(2, 0, 3, 1, 4)
    tol = 3e-5 * max(1.0, math.sqrt(D) / 8) * sharp
    assert (qkv32.double() - ref).abs().max().item() < tol
    assert (from_split(q)[:, :N].double() - ref[0].reshape(B * H, N, HD)).abs().max().item() < tol
    assert (from_split(k)[:, :N].double() - ref[1].reshape(B * H, N, HD)).abs().max().item() < tol
    assert (from_split(vt)[:, :, :N].double() - ref[2].reshape(B * H, N, HD).transpose(1, 2)).abs().max().item() < tol
    # attention on what the projection wrote (padding rows / columns still NaN pairs: they must not reach a result),
    # against float64 attention of exactly those operand values
    qd, kd = from_split(q)[:, :N].double(), from_split(k)[:, :N].double()
    vd = from_split(vt)[:, :, :N].double().transpose(1, 2)
    scale = HD ** -0.5
    s = (qd @ kd.transpose(1, 2)) * scale
    pref = s.softmax(-1)
    oref = (pref @ vd).reshape(B, H, N, HD).permute(0, 2, 1, 3).reshape(B, N, D)
    ctx = torch.full((B, N, D), -1, dtype=torch.int32, device=dev)
    lse = torch.empty((B * H, N), device=dev)
    _ok(lib, lib.ocm_op_attention_hd(X3, _p(q), _p(k), _p(vt), _p(ctx), _p(lse), B, N, H, HD, scale, _s()))
    # as in test_attention_x3: a score is a sum of products of 2^-17-accurate operands, so its absolute error (and the relative
    # error of every probability) grows with the operands' magnitude: |q| |k| ~ 1 .. 4 here
    es = max(1.0, float(qd.std() * kd.std())) * max(1.0, float(vd.abs().max()) / 4)
    assert (lse.double() - torch.logsumexp(s, -1) / math.log(2.0)).abs().max().item() < 1e-4 * es
    got = from_split(ctx).double()
    assert torch.isfinite(got).all()
    assert (got - oref).abs().max().item() < 4e-5 * es
    attn = torch.full((B, H, N, N), float("nan"), device=dev)
    _ok(lib, lib.ocm_op_attention_probs_hd(X3, _p(q), _p(k), _p(lse), _p(attn), B, N, H, HD, scale, _s()))
    assert (attn.reshape(B * H, N, N).double() - pref).abs().max().item() < 2e-5 * es
    lse2 = torch.empty_like(lse)
    _ok(lib, lib.ocm_op_attention_hd(X3, _p(q), _p(k), _p(vt), None, _p(lse2), B, N, H, HD, scale, _s()))
    assert torch.equal(lse, lse2)
    # other widths keep the generic fp32 kernel inside an engine handle; the operator says so
    assert lib.ocm_op_attention_hd(X3, _p(q), _p(k), _p(vt), _p(ctx), _p(lse), B, N, H, 96, scale, _s()) == _lib.OCM_EINVAL


@pytest.mark.parametrize("B,N,H", [(2, 65, 3), (1, 197, 3), (2, 300, 2), (1, 32, 1)])
@pytest.mark.parametrize("prec,tq,to", [("fp32", 1e-5, 2e-5), ("bf16", 2e-2, 3e-2)])
def test_attention_128_wide_heads_fp32_and_bf16(lib, dev, B, N, H, prec, tq, to):
    """128-wide heads in the other two precisions (round 4: the streaming kernels templated on the head width — exact-fp32 MFMA
    on 128 KiB of LDS, single bf16): qkv projection, attention, probabilities and selected rows against float64 torch on the
    operand values the projection wrote; NaN-poisoned padding must not reach a result."""
    P = _lib.PRECISIONS[prec]
    dt = torch.float32 if prec == "fp32" else torch.bfloat16
    HD, D = 128, H * 128
    a, w, bias = _rand((B * N, D), dev, 80), _rand((3 * D, D), dev, 81, 0.05), _rand((3 * D,), dev, 82, 0.1)
    npad = lib.ocm_n_pad_prec(P, N)
    q = torch.full((B * H, npad, HD), float("nan"), dtype=dt, device=dev)
    k = torch.full_like(q, float("nan"))
    vt = torch.full((B * H, HD, npad), float("nan"), dtype=dt, device=dev)
    qkv32 = torch.empty((3, B, H, N, HD), device=dev)
    a_s, w_s = to_operand(a, P), to_operand(w, P)
    _ok(lib, lib.ocm_op_qkv_proj_hd(P, _p(a_s), _p(w_s), _p(bias), _p(q), _p(k), _p(vt), _p(qkv32), B, N, H, HD, _s()))
    ref = (a.double() @ w.double().t() + bias.double()).reshape(B, N, 3, H, HD).permute(2, 0, 3, 1, 4)
    assert (qkv32.double() - ref).abs().max().item() < tq * 3
    assert (q[:, :N].double() - ref[0].reshape(B * H, N, HD)).abs().max().item() < tq * 3
    qd, kd = q[:, :N].double(), k[:, :N].double()
    vd = vt[:, :, :N].double().transpose(1, 2)
    scale = HD ** -0.5
    sc = (qd @ kd.transpose(1, 2)) * scale
    pref = sc.softmax(-1)
    oref = (pref @ vd).reshape(B, H, N, HD).permute(0, 2, 1, 3).reshape(B, N, D)
    ctx = torch.full((B, N, D), float("nan"), dtype=dt, device=dev)
    lse = torch.empty((B * H, N), device=dev)
    _ok(lib, lib.ocm_op_attention_hd(P, _p(q), _p(k), _p(vt), _p(ctx), _p(lse), B, N, H, HD, scale, _s()))
    got = ctx.double()
    assert torch.isfinite(got).all()
    assert (lse.double() - torch.logsumexp(sc, -1) / math.log(2.0)).abs().max().item() < (1e-4 if prec == "fp32" else 1e-5)
    assert (got - oref).abs().max().item() < to
    attn = torch.full((B, H, N, N), float("nan"), device=dev)
    _ok(lib, lib.ocm_op_attention_probs_hd(P, _p(q), _p(k), _p(lse), _p(attn), B, N, H, HD, scale, _s()))
    assert (attn.reshape(B * H, N, N).double() - pref).abs().max().item() < 2e-5
    lse2 = torch.empty_like(lse)
    _ok(lib, lib.ocm_op_attention_hd(P, _p(q), _p(k), _p(vt), None, _p(lse2), B, N, H, HD, scale, _s()))
    assert torch.equal(lse, lse2)


@pytest.mark.parametrize("prec", ["bf16", "fp32", "bf16x3"])
@pytest.mark.parametrize("M,D,K", [(1000, 384, 384), (12608, 384, 1536), (70, 128, 192), (333, 256, 512), (64, 256, 64)])
def test_linear_resid_ln_equals_linear_then_layernorm(lib, dev, prec, M, D, K):
    """ocm_op_linear_resid_ln (one kernel: GEMM + bias + residual, then LayerNorm of the rows it has just produced) is
    bit for bit ocm_op_linear(RESID_F32) followed by ocm_op_layernorm, in every precision mode, and sits where the
    float64 reference says (Block.forward, dino/vision_transformer.py:107-111)."""
    pc = _lib.PRECISIONS[prec]
    assert lib.ocm_linear_resid_ln_supported(D) == 1 and lib.ocm_linear_resid_ln_supported(96) == 0
    assert lib.ocm_linear_resid_ln_supported(512) == 0  # its split-bf16 tile would spill: not offered
    a, w = _rand((M, K), dev, 70), _rand((D, K), dev, 71, 0.05)
    bias, resid = _rand((D,), dev, 72, 0.1), _rand((M, D), dev, 73)
    gamma, beta = _rand((D,), dev, 74) * 0.1 + 1, _rand((D,), dev, 75) * 0.1
    a_s, w_s = to_operand(a, pc), to_operand(w, pc)
    out_kind = {"bf16": _lib.OCM_LN_BF16, "fp32": _lib.OCM_LN_F32, "bf16x3": _lib.OCM_LN_SPLIT}[prec]
    act = {"bf16": torch.bfloat16, "fp32": torch.float32, "bf16x3": torch.int32}[prec]
    # two kernels
    x2 = resid.clone()
    _ok(lib, lib.ocm_op_linear(pc, _p(a_s), _p(w_s), _p(bias), _p(x2), _p(x2), M, D, K, _lib.OCM_EPI_BIAS_RESID_F32, _s()))
    xn2 = torch.empty((M, D), dtype=act, device=dev)
    _ok(lib, lib.ocm_op_layernorm(_p(x2), _p(gamma), _p(beta), _p(xn2), out_kind, M, D, 1e-6, _s()))
    # one kernel (resid aliases x, as the engine calls it)
    x1 = resid.clone()
    xn1 = torch.empty((M, D), dtype=act, device=dev)
    _ok(lib, lib.ocm_op_linear_resid_ln(pc, _p(a_s), _p(w_s), _p(bias), _p(x1), _p(x1), _p(gamma), _p(beta), _p(xn1),
                                        M, D, K, 1e-6, _s()))
    torch.cuda.synchronize()
    assert torch.equal(x1, x2)
    assert torch.equal(xn1.view(torch.int32) if act != torch.bfloat16 else xn1.view(torch.int16),
                       xn2.view(torch.int32) if act != torch.bfloat16 else xn2.view(torch.int16))
    ref = resid.double() + a.double() @ w.double().t() + bias.double()
    tol = {"bf16": 2e-2, "fp32": 2e-4, "bf16x3": 3e-5}[prec] * max(1.0, math.sqrt(K) / 8)
    assert (x1.double() - ref).abs().max().item() < tol
    with pytest.raises(AssertionError):
        _ok(lib, lib.ocm_op_linear_resid_ln(pc, _p(a_s), _p(w_s), _p(bias), _p(x1), _p(x1), _p(gamma), _p(beta), _p(xn1),
                                            M, 96, K, 1e-6, _s()))
