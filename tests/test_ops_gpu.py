"""Kernel-level parity (K1..K14 of SURVEY §2.4): every stand-alone operator of the C ABI
against a plain PyTorch fp32 reference of the same op evaluated on the same (bf16-rounded)
operands. Runs on a real MI355X only."""
import ctypes as C
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _s():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ok(lib, rc):
    assert rc == 0, lib.ocm_last_error().decode()


def _rand(shape, dev, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(dev)


@pytest.mark.parametrize("rows,dim", [(1000, 384), (7, 192), (513, 768)])
def test_layernorm(lib, dev, rows, dim):
    x = _rand((rows, dim), dev, 1, 3.0) + 0.5
    g = _rand((dim,), dev, 2) * 0.1 + 1
    b = _rand((dim,), dev, 3) * 0.1
    ref = F.layer_norm(x, (dim,), g, b, 1e-6)
    y32 = torch.empty_like(x)
    _ok(lib, lib.ocm_op_layernorm(_p(x), _p(g), _p(b), _p(y32), 0, rows, dim, 1e-6, _s()))
    assert (y32 - ref).abs().max().item() < 2e-5  # fp32 tolerance
    y16 = torch.empty((rows, dim), dtype=torch.bfloat16, device=dev)
    _ok(lib, lib.ocm_op_layernorm(_p(x), _p(g), _p(b), _p(y16), 1, rows, dim, 1e-6, _s()))
    # bf16 output: within one bf16 ulp (2^-8 relative) of the fp32 reference
    assert ((y16.float() - ref).abs() <= ref.abs() * 2 ** -8 + 1e-6).all()


def test_cast_bf16(lib, dev):
    x = _rand((100003,), dev, 5)
    y = torch.empty(x.shape, dtype=torch.bfloat16, device=dev)
    _ok(lib, lib.ocm_op_cast_bf16(_p(x), _p(y), x.numel(), _s()))
    assert torch.equal(y, x.to(torch.bfloat16))  # bit-exact RNE


@pytest.mark.parametrize("M,N,K", [(1000, 384, 384), (12608, 1536, 384), (333, 384, 1536), (70, 96, 192),
                                   (12608, 384, 384), (64, 192, 64),
                                   (3000, 288, 128), (2100, 96, 128), (500, 192, 192),  # Swin: N tails, K of 2 / 3 steps
                                   (32768, 1024, 768)])  # 512 tiles of 256x256: the 8-wave tile + banded epilogue
@pytest.mark.parametrize("epi", [0, 1, 2, 3])
def test_linear(lib, dev, M, N, K, epi):
    a = _rand((M, K), dev, 10).to(torch.bfloat16)
    w = _rand((N, K), dev, 11, 0.05).to(torch.bfloat16)
    bias = _rand((N,), dev, 12, 0.1)
    resid = _rand((M, N), dev, 13)
    ref = a.float() @ w.float().t() + bias
    if epi == 1:
        ref = ref + resid
    if epi == 2:
        ref = F.gelu(ref)
    out_bf16 = epi in (2, 3)
    out = torch.full((M, N), float("nan"), dtype=torch.bfloat16 if out_bf16 else torch.float32, device=dev)
    if epi == 1:
        out.copy_(resid)  # in-place residual, as the engine uses it
    _ok(lib, lib.ocm_op_linear(0, _p(a), _p(w), _p(bias), _p(out) if epi == 1 else None, _p(out), M, N, K, epi, _s()))
    err = (out.float() - ref).abs()
    if out_bf16:
        assert (err <= ref.abs() * 2 ** -8 + 2e-4).all(), err.max().item()
    else:
        # fp32 accumulate of exact bf16 products: only summation order differs
        assert err.max().item() < 1e-4 * max(1.0, math.sqrt(K) / 8), err.max().item()


def _qkv_inputs(dev, B, N, H):
    D = H * 64
    a = _rand((B * N, D), dev, 20).to(torch.bfloat16)
    w = _rand((3 * D, D), dev, 21, 0.05).to(torch.bfloat16)
    bias = _rand((3 * D,), dev, 22, 0.1)
    return a, w, bias


@pytest.mark.parametrize("B,N,H", [(3, 197, 6), (2, 17, 2), (1, 577, 12), (5, 50, 3),
                                   (26, 577, 12)])  # 531 tiles of 256x256: big-tile q/k and transposed V^T epilogues
def test_qkv_proj(lib, dev, B, N, H):
    D = H * 64
    a, w, bias = _qkv_inputs(dev, B, N, H)
    npad = lib.ocm_n_pad(N)
    q = torch.zeros((B * H, npad, 64), dtype=torch.bfloat16, device=dev)
    k = torch.zeros_like(q)
    vt = torch.zeros((B * H, 64, npad), dtype=torch.bfloat16, device=dev)
    qkv32 = torch.empty((3, B, H, N, 64), dtype=torch.float32, device=dev)
    _ok(lib, lib.ocm_op_qkv_proj(0, _p(a), _p(w), _p(bias), _p(q), _p(k), _p(vt), _p(qkv32), B, N, H, _s()))
    # reference: Attention.forward :80
    ref = (a.float() @ w.float().t() + bias).reshape(B, N, 3, H, 64).permute(2, 0, 3, 1, 4)
    assert (qkv32 - ref).abs().max().item() < 2e-4
    rq = ref[0].reshape(B * H, N, 64)
    assert ((q[:, :N].float() - rq).abs() <= rq.abs() * 2 ** -8 + 1e-5).all()
    rk = ref[1].reshape(B * H, N, 64)
    assert ((k[:, :N].float() - rk).abs() <= rk.abs() * 2 ** -8 + 1e-5).all()
    rv = ref[2].reshape(B * H, N, 64).transpose(1, 2)
    assert ((vt[:, :, :N].float() - rv).abs() <= rv.abs() * 2 ** -8 + 1e-5).all()
    assert (q[:, N:] == 0).all() and (vt[:, :, N:] == 0).all()  # padding is never written


def _attn_inputs(dev, B, N, H, sharp=1.0):
    g = torch.Generator().manual_seed(30)
    q = (torch.randn((B * H, N, 64), generator=g) * sharp).to(torch.bfloat16)
    k = (torch.randn((B * H, N, 64), generator=g) * sharp).to(torch.bfloat16)
    v = torch.randn((B * H, N, 64), generator=g).to(torch.bfloat16)
    return q.to(dev), k.to(dev), v.to(dev)


def _pack(lib, q, k, v, poison):
    BH, N, _ = q.shape
    npad = lib.ocm_n_pad(N)
    fill = float("nan") if poison else 0.0  # padding may hold anything, NaN included
    qp = torch.full((BH, npad, 64), fill, dtype=torch.bfloat16, device=q.device)
    kp = torch.full((BH, npad, 64), fill, dtype=torch.bfloat16, device=q.device)
    vp = torch.full((BH, 64, npad), fill, dtype=torch.bfloat16, device=q.device)
    qp[:, :N] = q
    kp[:, :N] = k
    vp[:, :, :N] = v.transpose(1, 2)
    return qp, kp, vp


@pytest.mark.parametrize("B,N,H", [(2, 197, 6), (1, 17, 2), (1, 577, 3), (1, 64, 1), (1, 65, 1), (1, 2305, 2)])
@pytest.mark.parametrize("sharp", [1.0, 3.0])
def test_attention(lib, dev, B, N, H, sharp):
    q, k, v = _attn_inputs(dev, B, N, H, sharp)
    qp, kp, vp = _pack(lib, q, k, v, poison=True)
    scale = 0.125
    s = (q.float() @ k.float().transpose(1, 2)) * scale
    pref = s.softmax(-1)
    oref = (pref @ v.float()).reshape(B, H, N, 64).permute(0, 2, 1, 3).reshape(B, N, H * 64)
    ctx = torch.full((B, N, H * 64), float("nan"), dtype=torch.bfloat16, device=dev)
    lse = torch.empty((B * H, N), dtype=torch.float32, device=dev)
    _ok(lib, lib.ocm_op_attention(0, _p(qp), _p(kp), _p(vp), _p(ctx), _p(lse), B, N, H, scale, _s()))
    lse_ref = torch.logsumexp(s, -1) / math.log(2.0)
    assert (lse - lse_ref).abs().max().item() < 2e-4
    # P is rounded to bf16 before P·V and the output is bf16: ~2^-8 relative of |v|max-scaled rows
    assert (ctx.float() - oref).abs().max().item() < 0.03
    # probabilities from lse
    attn = torch.full((B, H, N, N), float("nan"), dtype=torch.float32, device=dev)
    _ok(lib, lib.ocm_op_attention_probs(0, _p(qp), _p(kp), _p(lse), _p(attn), B, N, H, scale, _s()))
    assert (attn.reshape(B * H, N, N) - pref).abs().max().item() < 2e-5
    assert (attn.sum(-1) - 1).abs().max().item() < 1e-4
    # stats-only variant (no ctx) gives the same lse
    lse2 = torch.empty_like(lse)
    _ok(lib, lib.ocm_op_attention(0, _p(qp), _p(kp), _p(vp), None, _p(lse2), B, N, H, scale, _s()))
    assert torch.equal(lse, lse2)
    # selected rows, CLS column dropped (utils.py:232)
    rows_idx = torch.tensor([0, N - 1, N // 2], dtype=torch.int32, device=dev)
    rows = torch.empty((B, H, 3, N - 1), dtype=torch.float32, device=dev)
    _ok(lib, lib.ocm_op_attention_rows(0, _p(qp), _p(kp), _p(rows_idx), 3, _p(rows), B, N, H, scale, _s()))
    ref_rows = pref.reshape(B, H, N, N)[:, :, rows_idx.long(), 1:]
    assert (rows - ref_rows).abs().max().item() < 2e-5


def test_attention_map_matches_reference_golden(lib, dev):
    """ocm_op_attention_map against the output of the reference's own compute_attention (helpers.npz)."""
    from tests.helpers import load_golden
    gold = load_golden("helpers")
    attn = torch.from_numpy(gold["ca_attn"]).to(dev)
    for query in (0, 9):
        maps = torch.empty((3, 5 * 8, 7 * 8), device=dev)
        _ok(lib, lib.ocm_op_attention_map(_p(attn), _p(maps), 0, 3, 36, query, 5, 7, 8, _s()))
        assert np.array_equal(maps.cpu().numpy(), gold[f"ca_maps_q{query}"])


def test_attention_map(lib, dev):
    B, H, hf, wf, p = 2, 3, 5, 7, 8
    N = hf * wf + 1
    attn = torch.rand((B, H, N, N), device=dev)
    for b, query in [(0, 0), (1, 9)]:
        maps = torch.empty((H, hf * p, wf * p), device=dev)
        _ok(lib, lib.ocm_op_attention_map(_p(attn), _p(maps), b, H, N, query, hf, wf, p, _s()))
        # compute_attention, utils.py:229-235
        ref = attn[b, :, query, 1:].reshape(H, hf, wf)
        ref = F.interpolate(ref.unsqueeze(0), scale_factor=p, mode="nearest")[0]
        assert torch.equal(maps, ref)  # pure index math: bit-exact


# ---------------------------------------------------------------------------------------------
# OCM_PREC_FP32: the same operators on fp32 operands (v_mfma_f32_32x32x2_f32 = exact fp32 products):
# results must agree with a float64 reference to fp32 round-off.
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("M,N,K", [(1000, 384, 384), (333, 384, 1536), (70, 96, 192), (64, 192, 64)])
@pytest.mark.parametrize("epi", [0, 1, 2, 3])
def test_linear_fp32(lib, dev, M, N, K, epi):
    a = _rand((M, K), dev, 40)
    w = _rand((N, K), dev, 41, 0.05)
    bias = _rand((N,), dev, 42, 0.1)
    resid = _rand((M, N), dev, 43)
    ref = a.double() @ w.double().t() + bias.double()
    if epi == 1:
        ref = ref + resid.double()
    if epi == 2:
        ref = F.gelu(ref)
    out = torch.full((M, N), float("nan"), dtype=torch.float32, device=dev)
    if epi == 1:
        out.copy_(resid)
    _ok(lib, lib.ocm_op_linear(1, _p(a), _p(w), _p(bias), _p(out) if epi == 1 else None, _p(out), M, N, K, epi, _s()))
    assert (out.double() - ref).abs().max().item() < 2e-5 * max(1.0, math.sqrt(K) / 8)


@pytest.mark.parametrize("B,N,H", [(3, 197, 6), (2, 17, 2), (5, 50, 3)])
def test_qkv_proj_fp32(lib, dev, B, N, H):
    D = H * 64
    a, w, bias = _rand((B * N, D), dev, 50), _rand((3 * D, D), dev, 51, 0.05), _rand((3 * D,), dev, 52, 0.1)
    npad = lib.ocm_n_pad(N)
    q = torch.zeros((B * H, npad, 64), device=dev)
    k = torch.zeros_like(q)
    vt = torch.zeros((B * H, 64, npad), device=dev)
    qkv32 = torch.empty((3, B, H, N, 64), device=dev)
    _ok(lib, lib.ocm_op_qkv_proj(1, _p(a), _p(w), _p(bias), _p(q), _p(k), _p(vt), _p(qkv32), B, N, H, _s()))
    ref = (a.double() @ w.double().t() + bias.double()).reshape(B, N, 3, H, 64).permute(2, 0, 3, 1, 4)
    assert (qkv32.double() - ref).abs().max().item() < 2e-5
    assert (q[:, :N].double() - ref[0].reshape(B * H, N, 64)).abs().max().item() < 2e-5
    assert (k[:, :N].double() - ref[1].reshape(B * H, N, 64)).abs().max().item() < 2e-5
    assert (vt[:, :, :N].double() - ref[2].reshape(B * H, N, 64).transpose(1, 2)).abs().max().item() < 2e-5
    assert (q[:, N:] == 0).all() and (vt[:, :, N:] == 0).all()


@pytest.mark.parametrize("B,N,H", [(2, 197, 6), (1, 17, 2), (1, 577, 3), (1, 65, 1)])
@pytest.mark.parametrize("sharp", [1.0, 3.0])
def test_attention_fp32(lib, dev, B, N, H, sharp):
    g = torch.Generator().manual_seed(60)
    q = (torch.randn((B * H, N, 64), generator=g) * sharp).to(dev)
    k = (torch.randn((B * H, N, 64), generator=g) * sharp).to(dev)
    v = torch.randn((B * H, N, 64), generator=g).to(dev)
    npad = lib.ocm_n_pad(N)
    qp = torch.full((B * H, npad, 64), float("nan"), device=dev)
    kp = torch.full((B * H, npad, 64), float("nan"), device=dev)
    vp = torch.full((B * H, 64, npad), float("nan"), device=dev)
    qp[:, :N], kp[:, :N], vp[:, :, :N] = q, k, v.transpose(1, 2)
    scale = 0.125
    s = (q.double() @ k.double().transpose(1, 2)) * scale
    pref = s.softmax(-1)
    oref = (pref @ v.double()).reshape(B, H, N, 64).permute(0, 2, 1, 3).reshape(B, N, H * 64)
    ctx = torch.full((B, N, H * 64), float("nan"), device=dev)
    lse = torch.empty((B * H, N), device=dev)
    _ok(lib, lib.ocm_op_attention(1, _p(qp), _p(kp), _p(vp), _p(ctx), _p(lse), B, N, H, scale, _s()))
    assert (lse.double() - torch.logsumexp(s, -1) / math.log(2.0)).abs().max().item() < 1e-4
    assert (ctx.double() - oref).abs().max().item() < 2e-5
    attn = torch.full((B, H, N, N), float("nan"), device=dev)
    _ok(lib, lib.ocm_op_attention_probs(1, _p(qp), _p(kp), _p(lse), _p(attn), B, N, H, scale, _s()))
    assert (attn.reshape(B * H, N, N).double() - pref).abs().max().item() < 2e-5
    rows_idx = torch.tensor([0, N - 1], dtype=torch.int32, device=dev)
    rows = torch.empty((B, H, 2, N - 1), device=dev)
    _ok(lib, lib.ocm_op_attention_rows(1, _p(qp), _p(kp), _p(rows_idx), 2, _p(rows), B, N, H, scale, _s()))
    assert (rows.double() - pref.reshape(B, H, N, N)[:, :, rows_idx.long(), 1:]).abs().max().item() < 2e-5
