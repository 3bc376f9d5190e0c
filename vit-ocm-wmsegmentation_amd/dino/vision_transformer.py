"""Drop-in module surface of the reference's `dino/vision_transformer.py`, executed by the
gfx950 HIP engine (libocm_vit.so) instead of ATen.

What is kept: the factories `vit_tiny/vit_small/vit_base(patch_size, **kw)`, the
class names, constructor keywords, attribute names (`patch_embed.proj`, `cls_token`, `pos_embed`,
`pos_drop`, `blocks`, `norm`, `head`, `embed_dim`, `num_features`), the state_dict keys and the
method signatures / return conventions of `VisionTransformer`
(reference dino/vision_transformer.py:135-256), so eval.py:60-77,136, sw_processing.py:181-239,
analyse_attention.py:55-116 and the subclasses in model.py:11-53,110-139 bind unchanged.

What is different: the nn.Linear / nn.Conv2d / nn.LayerNorm children are parameter containers.
No forward below calls their ATen kernels; every method packs device pointers and calls the C ABI
(include/ocm_vit.h). Inputs must be on a HIP device — there is deliberately no CPU fallback
(the CPU statement of this arithmetic is the test oracle under oracle/).
"""
import math
import threading
import warnings
from functools import partial

import torch
import torch.nn as nn

from .. import _lib
from ..engine import Engine, _p, _require_hip, _stream, to_operand
from .utils import trunc_normal_


def _f32c(t):
    return t.detach().to(torch.float32).contiguous()


class _OperandCache:
    """Operand copy (engine.to_operand: bf16 / fp32 / split-bf16 pairs) of an fp32 matrix parameter for the stand-alone
    operator paths, rebuilt when the parameter or the precision changes."""

    def __init__(self):
        self.key, self.val = None, None

    def get(self, param, prec):
        key = (param.data_ptr(), param._version, param.device, prec)
        if key != self.key:
            self.key, self.val = key, to_operand(_f32c(param), prec)
        return self.val


_ACT_DTYPE = {_lib.OCM_PREC_BF16: torch.bfloat16, _lib.OCM_PREC_FP32: torch.float32, _lib.OCM_PREC_BF16X3: torch.int32}


def _module_prec(mod):
    """OCM_PREC_* of a free-standing sub-module: its `precision` attribute ("bf16x3" unless changed)."""
    name = getattr(mod, "precision", _lib.DEFAULT_PRECISION)
    if name not in _lib.PRECISIONS:
        raise ValueError(f"precision must be one of {sorted(_lib.PRECISIONS)}, got {name!r}")
    return _lib.PRECISIONS[name]


def _bias_or_zeros(linear):
    if linear.bias is not None:
        return _f32c(linear.bias)
    return torch.zeros(linear.out_features, dtype=torch.float32, device=linear.weight.device)


class LayerNorm(nn.LayerNorm):
    """nn.LayerNorm parameter container whose forward runs ocm_op_layernorm (fp32 out)."""

    def forward(self, x):
        _require_hip(x, "LayerNorm input")
        x32 = _f32c(x)
        y = torch.empty_like(x32)
        dim = x32.shape[-1]
        with torch.cuda.device(x32.device):
            _lib.check(_lib.load().ocm_op_layernorm(_p(x32), _p(_f32c(self.weight)), _p(_f32c(self.bias)), _p(y), 0,
                                                    x32.numel() // dim, dim, float(self.eps), _stream()))
        return y


class Mlp(nn.Module):
    """fc2(GELU_erf(fc1(x))) — reference Mlp (:47-63). Dropout is p=0 in every reference config."""

    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, drop=0.):
        super().__init__()
        self.fc1 = nn.Linear(in_features, hidden_features or in_features)
        self.act = act_layer()
        self.fc2 = nn.Linear(hidden_features or in_features, out_features or in_features)
        self.drop = nn.Dropout(drop)
        self._w = (_OperandCache(), _OperandCache())

    def forward(self, x):
        """Free-standing use (outside a VisionTransformer): the stand-alone operators in this module's `precision`
        ("bf16x3" by default, like the engine; "bf16" / "fp32")."""
        _require_hip(x, "Mlp input")
        lib, pc = _lib.load(), _module_prec(self)
        shape = x.shape
        x32 = _f32c(x).reshape(-1, shape[-1])
        rows, hid, out_f = x32.shape[0], self.fc1.out_features, self.fc2.out_features
        with torch.cuda.device(x32.device):
            a = to_operand(x32, pc)
            h = torch.empty((rows, hid), dtype=_ACT_DTYPE[pc], device=x32.device)
            _lib.check(lib.ocm_op_linear(pc, _p(a), _p(self._w[0].get(self.fc1.weight, pc)), _p(_bias_or_zeros(self.fc1)), None,
                                         _p(h), rows, hid, shape[-1], _lib.OCM_EPI_BIAS_GELU_BF16, _stream()))
            y = torch.empty((rows, out_f), dtype=torch.float32, device=x32.device)
            _lib.check(lib.ocm_op_linear(pc, _p(h), _p(self._w[1].get(self.fc2.weight, pc)), _p(_bias_or_zeros(self.fc2)), None,
                                         _p(y), rows, out_f, hid, _lib.OCM_EPI_BIAS_F32, _stream()))
        return y.reshape(*shape[:-1], out_f)


class Attention(nn.Module):
    """Multi-head self-attention returning (x, attn, qkv) — reference Attention (:66-90)."""

    def __init__(self, dim, num_heads=8, qkv_bias=False, qk_scale=None, attn_drop=0., proj_drop=0.):
        super().__init__()
        self.num_heads = num_heads
        self.scale = qk_scale or (dim // num_heads) ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.attn_drop = nn.Dropout(attn_drop)
        self.proj = nn.Linear(dim, dim)
        self.proj_drop = nn.Dropout(proj_drop)
        self._w = (_OperandCache(), _OperandCache())

    def forward(self, x, return_qkv=False):
        """Free-standing use (outside a VisionTransformer): the stand-alone operators in this module's `precision`
        ("bf16x3" by default, like the engine; "bf16" / "fp32"). The V^T padding columns are zeroed here: the operators
        never write them and they are multiplied by exact zeros."""
        _require_hip(x, "Attention input")
        lib, pc = _lib.load(), _module_prec(self)
        B, N, Cd = x.shape
        H = self.num_heads
        if Cd % H or (Cd // H) % 8:
            raise ValueError(f"dim {Cd} / num_heads {H}: the HIP kernels need a head width that is a multiple of 8")
        hd = Cd // H
        # 64- and 128-wide heads (model.py:93-103) run the MFMA attention kernels in every precision; any other width (the
        # reference accepts every dim / num_heads, :66-90) the generic fp32 kernel on the fp32 qkv tensor
        mfma = hd in (64, 128)
        x32 = _f32c(x).reshape(B * N, Cd)
        dev, npad, adt = x32.device, lib.ocm_n_pad_prec(pc, N), _ACT_DTYPE[pc]
        with torch.cuda.device(dev):
            a = to_operand(x32, pc)
            qkv = torch.empty((3, B, H, N, hd), dtype=torch.float32, device=dev)
            ctx = torch.empty((B * N, Cd), dtype=adt, device=dev)
            attn = torch.empty((B, H, N, N), dtype=torch.float32, device=dev)
            wq, bq = _p(self._w[0].get(self.qkv.weight, pc)), _p(_bias_or_zeros(self.qkv))
            if mfma:
                q = torch.empty((B * H, npad, hd), dtype=adt, device=dev)
                k = torch.empty_like(q)
                vt = torch.zeros((B * H, hd, npad), dtype=adt, device=dev)
                _lib.check(lib.ocm_op_qkv_proj_hd(pc, _p(a), wq, bq, _p(q), _p(k), _p(vt), _p(qkv), B, N, H, hd, _stream()))
                lse = torch.empty((B * H, N), dtype=torch.float32, device=dev)
                _lib.check(lib.ocm_op_attention_hd(pc, _p(q), _p(k), _p(vt), _p(ctx), _p(lse), B, N, H, hd, float(self.scale),
                                                   _stream()))
                _lib.check(lib.ocm_op_attention_probs_hd(pc, _p(q), _p(k), _p(lse), _p(attn), B, N, H, hd, float(self.scale),
                                                         _stream()))
            else:
                _lib.check(lib.ocm_op_qkv_proj_hd(pc, _p(a), wq, bq, None, None, None, _p(qkv), B, N, H, hd, _stream()))
                _lib.check(lib.ocm_op_attention_generic(pc, _p(qkv), _p(ctx), _p(attn), B, N, H, hd, float(self.scale), _stream()))
            y = torch.empty((B * N, Cd), dtype=torch.float32, device=dev)
            _lib.check(lib.ocm_op_linear(pc, _p(ctx), _p(self._w[1].get(self.proj.weight, pc)), _p(_bias_or_zeros(self.proj)),
                                         None, _p(y), B * N, Cd, Cd, _lib.OCM_EPI_BIAS_F32, _stream()))
        return y.reshape(B, N, Cd), attn, qkv


class Block(nn.Module):
    """Pre-norm transformer block — reference Block (:94-114)."""

    def __init__(self, dim, num_heads, mlp_ratio=4., qkv_bias=False, qk_scale=None, drop=0., attn_drop=0.,
                 drop_path=0., act_layer=nn.GELU, norm_layer=nn.LayerNorm):
        super().__init__()
        self.norm1 = _make_norm(norm_layer, dim)
        self.attn = Attention(dim, num_heads=num_heads, qkv_bias=qkv_bias, qk_scale=qk_scale, attn_drop=attn_drop,
                              proj_drop=drop)
        self.drop_path = nn.Identity()
        self._drop_rates = (drop, attn_drop, drop_path)
        self.norm2 = _make_norm(norm_layer, dim)
        self.mlp = Mlp(in_features=dim, hidden_features=int(dim * mlp_ratio), act_layer=act_layer, drop=drop)

    def forward(self, x, return_attention=False, return_qkv=False):
        owner = self.__dict__.get("_owner")
        if owner is not None:  # the block of a VisionTransformer: one fused engine call
            return owner[0]._block_forward(self.__dict__["_index"], x, return_attention, return_qkv)
        # free-standing block: same kernels through the stand-alone operators
        y, attn, qkv = self.attn(self.norm1(x))
        if return_attention:
            return attn
        x = _f32c(x) + y
        x = x + self.mlp(self.norm2(x))
        return (x, attn, qkv) if return_qkv else x


class PatchEmbed(nn.Module):
    """Image to patch embedding — reference PatchEmbed (:117-132): Conv2d(k=p, s=p), row-major
    flatten, token-major output (B, P, D)."""

    def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=768):
        super().__init__()
        self.img_size = img_size
        self.patch_size = patch_size
        self.num_patches = (img_size // patch_size) * (img_size // patch_size)
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size)

    def forward(self, x):
        owner = self.__dict__.get("_owner")
        if owner is None:
            raise RuntimeError("PatchEmbed runs through its VisionTransformer's HIP engine; a free-standing "
                               "PatchEmbed has no engine to call")
        return owner[0]._patch_embed_only(x)


def _make_norm(norm_layer, dim):
    probe = norm_layer(dim)
    if not isinstance(probe, nn.LayerNorm) or not probe.elementwise_affine:
        raise NotImplementedError("the HIP engine implements affine nn.LayerNorm only")
    return LayerNorm(dim, eps=probe.eps)


class VisionTransformer(nn.Module):
    """Vision Transformer — reference VisionTransformer (:135-256)."""

    def __init__(self, img_size=[224], patch_size=16, in_chans=3, num_classes=0, embed_dim=768, depth=12,
                 num_heads=12, mlp_ratio=4., qkv_bias=False, qk_scale=None, drop_rate=0., attn_drop_rate=0.,
                 drop_path_rate=0., norm_layer=nn.LayerNorm, **kwargs):
        super().__init__()
        self.num_features = self.embed_dim = embed_dim
        self.patch_embed = PatchEmbed(img_size=img_size[0], patch_size=patch_size, in_chans=in_chans,
                                      embed_dim=embed_dim)
        n0 = self.patch_embed.num_patches + 1
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.pos_embed = nn.Parameter(torch.zeros(1, n0, embed_dim))
        self.pos_drop = nn.Dropout(p=drop_rate)
        rates = torch.linspace(0, drop_path_rate, depth).tolist()
        self.blocks = nn.ModuleList([
            Block(dim=embed_dim, num_heads=num_heads, mlp_ratio=mlp_ratio, qkv_bias=qkv_bias, qk_scale=qk_scale,
                  drop=drop_rate, attn_drop=attn_drop_rate, drop_path=rates[i], norm_layer=norm_layer)
            for i in range(depth)])
        self.norm = _make_norm(norm_layer, embed_dim)
        self.head = nn.Linear(embed_dim, num_classes) if num_classes > 0 else nn.Identity()

        trunc_normal_(self.pos_embed, std=.02)
        trunc_normal_(self.cls_token, std=.02)
        self.apply(self._init_weights)

        # engine-side state (never part of the state_dict)
        self._hyper = dict(patch_size=patch_size, in_chans=in_chans, embed_dim=embed_dim, depth=depth,
                           num_heads=num_heads, mlp_hidden=int(embed_dim * mlp_ratio), ln_eps=float(self.norm.eps),
                           qk_scale=float(qk_scale or (embed_dim // num_heads) ** -0.5))
        self._drop_any = max(drop_rate, attn_drop_rate, drop_path_rate) > 0
        self.__dict__["_engines"] = {}
        self.__dict__["_pos_cache"] = {}
        self.__dict__["_gray_fold"] = False
        self.__dict__["_precision"] = _lib.DEFAULT_PRECISION
        for i, blk in enumerate(self.blocks):
            blk.__dict__["_owner"] = (self,)
            blk.__dict__["_index"] = i
        self.patch_embed.__dict__["_owner"] = (self,)

    def _init_weights(self, m):
        # reference :167-174 — Linear: trunc-normal(.02) weight, zero bias; LayerNorm: (1, 0)
        if isinstance(m, nn.Linear):
            trunc_normal_(m.weight, std=.02)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.LayerNorm):
            nn.init.constant_(m.bias, 0)
            nn.init.constant_(m.weight, 1.0)

    def __getstate__(self):
        st = self.__dict__.copy()
        st["_engines"], st["_pos_cache"] = {}, {}
        st.pop("_auto_graphs", None)
        st.pop("_param_slots", None)
        st.pop("_auto_seen", None)
        st.pop("_auto_lock", None)  # locks do not pickle / deepcopy
        return st

    # ---- engine management ------------------------------------------------------------------
    def enable_grayscale_fold(self, on=True):
        """OCM tiles are grayscale replicated to RGB (data.py:27,68). With the fold on, the patch
        embedding reads ONE plane and uses W.sum(dim=1): pass (B,1,H,W) tiles, or
        (B,3,H,W) tiles whose planes are identical (only plane 0 is read)."""
        if self.patch_embed.proj.in_channels != 3 and on:
            raise ValueError("grayscale fold applies to a 3-channel patch embedding")
        self.__dict__["_gray_fold"] = bool(on)
        self.__dict__["_engines"] = {}
        return self

    def set_precision(self, precision):
        """Arithmetic of the contraction kernels (residual stream, LayerNorm, softmax and accumulators
        are fp32 in all three):
          "bf16x3" (default) split-bf16: every operand is a pair hi + lo of bf16 numbers and every product three
                           bf16 MFMAs (hi*hi + hi*lo + lo*hi). Attention maps within 1e-3 of the fp32 reference on
                           every golden weight set whose softmax is not saturated: the trained-like sets of
                           ViT-S/16, ViT-B/16 and ViT-S/8 (attention max 0.79 .. 0.96) included. On a saturated
                           softmax (attention max 1.0000) fp32 arithmetic itself is 5e-4 from float64 and this
                           mode 2-5e-3.
          "bf16"           single bf16 MFMA operands — the fastest path; within 1e-3 on well-conditioned weights
                           (init / full / sharp sets) but NOT on peaked, trained-like attention (4-8e-2 there):
                           rounding operands to 8 mantissa bits is amplified layer by layer
          "fp32"           fp32 operands on v_mfma_f32_32x32x2_f32 (exact fp32 products, 1/16 the matrix
                           rate): maps at fp32 round-off level"""
        if precision not in _lib.PRECISIONS:
            raise ValueError(f"precision must be one of {sorted(_lib.PRECISIONS)}, got {precision!r}")
        if precision != self._precision:
            self.__dict__["_precision"] = precision
            self.__dict__["_engines"] = {}
        return self

    def _named_engine_params(self):
        """(name, parameter) of everything the engine holds a copy of. The (owning module, key) pairs are resolved once —
        walking the module tree with named_parameters() costs more than a one-tile forward's launches — and the parameters
        are then read from the modules' own dicts, so a replaced Parameter object is still seen. A replaced SUB-MODULE
        (`blk.attn.qkv = nn.Linear(...)`, a swapped Block, a LoRA / quantisation wrapper) is seen too: the cache keeps the
        (parent._modules, child name, child) edges of the tree it was built from and is rebuilt when one of them no longer
        holds (174 identity checks for ViT-S, ~10 us). A parameter set to None afterwards (`lin.bias = None`) drops out."""
        cached = self.__dict__.get("_param_slots")
        if cached is not None and not all(d.get(n) is c for d, n, c in cached[1]):
            cached = None
        if cached is None:
            skip = ("pos_embed", "head.")
            slots = []
            for n, _ in self.named_parameters():
                if n.startswith(skip):
                    continue
                prefix, _, key = n.rpartition(".")
                slots.append((self.get_submodule(prefix) if prefix else self, key, n))
            edges = [(mod._modules, cn, child) for _, mod in self.named_modules() for cn, child in mod._modules.items()]
            cached = self.__dict__["_param_slots"] = (slots, edges)
        return [(n, p) for mod, key, n in cached[0] if (p := mod._parameters.get(key)) is not None]

    def _engine(self, device):
        if self.training and self._drop_any:
            raise NotImplementedError("dropout / drop-path > 0 in training mode is outside the inference hot path")
        named = self._named_engine_params()
        sig = tuple((p.data_ptr(), p._version) for _, p in named)
        ent = self._engines.get(device)
        if ent is not None and ent[1] == sig:
            return ent[0]
        for _, p in named:
            if p.device != device:
                raise RuntimeError(f"parameters are on {p.device} but the input is on {device}; call model.to(device)")
        hy = dict(self._hyper)
        if self._gray_fold:
            hy["in_chans"] = 1
        eng = ent[0] if ent is not None else Engine(device=device, precision=_lib.PRECISIONS[self._precision], **hy)
        have, keep = set(), []
        for name, p in named:
            eng.set_param(name, p, keep)
            have.add(name)
        zeros = {}
        for i, blk in enumerate(self.blocks):  # qkv_bias=False etc.: the engine always adds a bias
            for sub, lin in (("attn.qkv", blk.attn.qkv), ("attn.proj", blk.attn.proj), ("mlp.fc1", blk.mlp.fc1),
                             ("mlp.fc2", blk.mlp.fc2)):
                key = f"blocks.{i}.{sub}.bias"
                if key not in have:
                    z = zeros.setdefault(lin.out_features, torch.zeros(lin.out_features, device=device))
                    eng.set_param(key, z, keep)
        if self.patch_embed.proj.bias is None:
            eng.set_param("patch_embed.proj.bias", torch.zeros(self.embed_dim, device=device), keep)
        eng.flush(keep)  # one synchronisation per load, not one per parameter
        self._engines[device] = (eng, sig)
        self.__dict__["_engine_epoch"] = self.__dict__.get("_engine_epoch", 0) + 1  # captured launch sequences are stale now
        return eng

    def _pos_for(self, npatch, w, h, device):
        """(N, D) fp32 positional table for a w x h pixel tile, on `device`, cached per shape."""
        key = (npatch, w, h, device, self.pos_embed.data_ptr(), self.pos_embed._version)
        pos = self._pos_cache.get(key)
        if pos is None:
            pos = self._interpolated_pos(npatch, w, h)[0].to(device=device, dtype=torch.float32).contiguous()
            self.__dict__["_pos_cache"] = {key: pos}
        return pos

    def _interpolated_pos(self, npatch, w, h):
        """reference interpolate_pos_encoding (:176-196), evaluated once per tile shape on the host
        (it does not depend on pixel values): identity for the native square grid, otherwise bicubic
        resampling of the sqrt(N0) x sqrt(N0) grid with the +0.1 scale-factor trick."""
        pos = self.pos_embed.detach().float().cpu()
        n0 = pos.shape[1] - 1
        if npatch == n0 and w == h:
            return pos
        dim = pos.shape[-1]
        side = int(math.sqrt(n0))
        w0 = w // self.patch_embed.patch_size + 0.1
        h0 = h // self.patch_embed.patch_size + 0.1
        grid = pos[:, 1:].reshape(1, side, side, dim).permute(0, 3, 1, 2)
        grid = nn.functional.interpolate(grid, scale_factor=(w0 / math.sqrt(n0), h0 / math.sqrt(n0)), mode="bicubic")
        assert int(w0) == grid.shape[-2] and int(h0) == grid.shape[-1]
        grid = grid.permute(0, 2, 3, 1).reshape(1, -1, dim)
        return torch.cat((pos[:, :1], grid), dim=1)

    def _check_input(self, x):
        _require_hip(x, "input")
        if x.dim() != 4:
            raise ValueError(f"expected (B, C, H, W) input, got {tuple(x.shape)}")
        p = self.patch_embed.patch_size
        if x.shape[-2] % p or x.shape[-1] % p:
            raise ValueError(f"input {tuple(x.shape[-2:])} is not a multiple of patch_size {p}")
        x = x.detach()
        if x.dtype != torch.float32:
            x = x.float()
        if self._gray_fold and x.shape[1] == 3:
            x = x[:, :1]
        return x

    # One tile per call — what the reference's loops issue (eval.py:126-171, sw_processing.py:235-258) — is bound by the 60-odd
    # kernel launches of a forward rather than by the kernels. Such calls (eval mode, at most AUTO_GRAPH_TOKENS token rows) are
    # therefore replayed as a HIP graph without the caller asking: the second call of a given (shape, outputs) captures the
    # launch sequence on fixed buffers, later calls copy the tile in, replay with one launch and return copies. Same kernels,
    # same order, same bits; every call still compares the parameters' (address, version) signature, so a load_state_dict
    # or an in-place update re-captures. `model.auto_graph = False` switches it off.
    AUTO_GRAPH_TOKENS = 1024
    AUTO_GRAPH_ENTRIES = 8
    auto_graph = True

    def _run(self, x, **kw):
        x = self._check_input(x)
        eng = self._engine(x.device)
        w, h = x.shape[-2], x.shape[-1]
        npatch = (w // eng.p) * (h // eng.p)
        if (self.auto_graph and not self.training and not self.__dict__.get("_graph_suspended")
                and x.shape[0] * (npatch + 1) <= self.AUTO_GRAPH_TOKENS and not torch.cuda.is_current_stream_capturing()):
            out = self._run_auto_graph(x, eng, kw)
            if out is not None:
                return out
        return eng.forward(x, self._pos_for(npatch, w, h, x.device), **kw)

    def _run_auto_graph(self, x, eng, kw):
        """The replayed result, or None when this call should run launch by launch: a (shape, outputs, query_rows tensor)
        combination is captured the SECOND time it is seen — a loop over differently sized tiles, or one that builds a new
        query_rows tensor per call, never pays for warm-ups and captures it would not reuse.
        The static input / output buffers of a capture belong to the (thread, stream) that made it (both are part of the
        key), and capture + replay run under one lock per model, so two threads or streams driving the same model neither
        share buffers nor see each other's `_graph_suspended` window."""
        qr = kw.get("query_rows")
        cur = torch.cuda.current_stream(x.device)
        key = (tuple(x.shape), x.device, id(eng), self.__dict__.get("_engine_epoch", 0), self._precision, self._gray_fold,
               eng.option_epoch,  # set_fold_layernorm / set_fuse_layernorm change the launch sequence a capture froze
               threading.get_ident(), cur.cuda_stream, self.pos_embed.data_ptr(),
               self.pos_embed._version, kw.get("flags"), kw.get("n_last"), None if qr is None else (id(qr), qr._version))
        lock = self.__dict__.get("_auto_lock")
        if lock is None:
            lock = self.__dict__.setdefault("_auto_lock", threading.RLock())
        with lock:
            cache = self.__dict__.setdefault("_auto_graphs", {})
            ent = cache.get(key)
            if ent is None:
                seen = self.__dict__.setdefault("_auto_seen", {})
                if key not in seen:
                    if len(seen) >= 64:
                        seen.clear()
                    seen[key] = qr  # (keeps the index tensor alive, so its id cannot be reused by another one)
                    return None
                del seen[key]
                for k in [k for k in cache if k[2:4] != key[2:4]]:  # captures of engines / parameter uploads that are gone
                    del cache[k]
                while len(cache) >= self.AUTO_GRAPH_ENTRIES:
                    del cache[next(iter(cache))]
                with torch.inference_mode(False):  # a normal tensor: a later copy_ outside inference mode stays legal
                    xs = torch.empty(x.shape, dtype=x.dtype, device=x.device)
                xs.copy_(x)
                self.__dict__["_graph_suspended"] = True
                try:
                    side = torch.cuda.Stream(device=x.device)
                    side.wait_stream(cur)
                    with torch.cuda.stream(side):  # warm-up off the capturing stream: workspace, LDS opt-ins
                        for _ in range(2):         # (an engine error here is the caller's to see: not caught)
                            self._run(xs, **kw)
                    cur.wait_stream(side)
                    graph = torch.cuda.CUDAGraph()
                    try:
                        # thread-local capture: HIP calls of the caller's other threads (loaders, pinned copies) stay legal
                        with torch.cuda.graph(graph, capture_error_mode="thread_local"):
                            out = self._run(xs, **kw)
                    except RuntimeError as exc:  # a capture this process cannot make: run launch by launch from now on
                        warnings.warn(f"automatic HIP-graph replay switched off for this model (capture failed: {exc})",
                                      RuntimeWarning, stacklevel=3)
                        self.auto_graph = False
                        return None
                finally:
                    self.__dict__["_graph_suspended"] = False
                # everything the captured launches point at stays alive with the capture
                ent = cache[key] = (graph, xs, out, (eng, list(eng._ws.values()), dict(self._pos_cache), qr))
            graph, xs, out, _ = ent
            xs.copy_(x)
            graph.replay()
            return _tree_map(torch.clone, out)

    # ---- reference methods ------------------------------------------------------------------
    def interpolate_pos_encoding(self, x, w, h):
        return self._interpolated_pos(x.shape[1] - 1, w, h).to(x.device)

    def prepare_tokens(self, x):
        x = self._check_input(x)
        eng = self._engine(x.device)
        w, h = x.shape[-2], x.shape[-1]
        return eng.prepare_tokens(x, self._pos_for((w // eng.p) * (h // eng.p), w, h, x.device))

    def _patch_embed_only(self, x):
        x = self._check_input(x)
        eng = self._engine(x.device)
        n = eng.n_tokens(x.shape[-2], x.shape[-1])
        zero_pos = torch.zeros((n, self.embed_dim), dtype=torch.float32, device=x.device)
        return eng.prepare_tokens(x, zero_pos)[:, 1:]

    def _block_forward(self, index, x, return_attention, return_qkv):
        _require_hip(x, "block input")
        eng = self._engine(x.device)
        xo, attn, qkv = eng.block_forward(index, x, want_attn=return_attention or return_qkv, want_qkv=return_qkv,
                                          attn_only=return_attention)
        if return_attention:
            return attn
        return (xo, attn, qkv) if return_qkv else xo

    def forward(self, x):
        return self._run(x, flags=_lib.OCM_OUT_FEAT)["feat"][0][:, 0]

    def forward_feats(self, x):
        return self._run(x, flags=_lib.OCM_OUT_FEAT)["feat"][0]

    def get_intermediate_feat(self, x, n=1):
        n = max(1, min(int(n), len(self.blocks)))
        out = self._run(x, flags=_lib.OCM_OUT_FEAT | _lib.OCM_OUT_ATTN | _lib.OCM_OUT_QKV, n_last=n)
        return list(out["feat"].unbind(0)), list(out["attn"].unbind(0)), list(out["qkv"].unbind(0))

    def get_last_selfattention(self, x):
        return self._run(x, flags=_lib.OCM_OUT_ATTN | _lib.OCM_LAST_ATTN_ONLY)["attn"][0]

    def get_intermediate_layers(self, x, n=1):
        n = max(1, min(int(n), len(self.blocks)))
        return list(self._run(x, flags=_lib.OCM_OUT_FEAT, n_last=n)["feat"].unbind(0))

    # ---- MI355X-native extensions ----------------------------------------------------------------
    def graphed(self, method="get_last_selfattention", **kwargs):
        """`method` as a HIP-graph replay for repeated same-shaped calls (one tile per call: -16 % latency):
        `run = model.graphed("get_last_selfattention"); attn = run(img)`. See GraphedCall."""
        return GraphedCall(self, method, kwargs)

    def get_last_attention_rows(self, x, query_rows=None):
        """attentions[0][:, :, query, 1:] of the last block for the given token indices (default: CLS)
        without materialising the (B,H,N,N) matrix: (B, H, n_rows, N-1) fp32."""
        return self._run(x, flags=_lib.OCM_OUT_ROWS | _lib.OCM_LAST_ATTN_ONLY, query_rows=query_rows)["rows"]


def _tree_map(fn, obj):
    if isinstance(obj, torch.Tensor):
        return fn(obj)
    if isinstance(obj, (list, tuple)):
        return type(obj)(_tree_map(fn, o) for o in obj)
    if isinstance(obj, dict):
        return {k: _tree_map(fn, v) for k, v in obj.items()}
    return obj


class GraphedCall:
    """One method of a VisionTransformer replayed as a HIP graph (`model.graphed("get_last_selfattention")`).

    The reference's callers run one tile per call (eval.py:126-171, sw_processing.py:235-258, analyse_attention.py):
    61 kernel launches of 5-25 us each, so a call is bound by launch overhead and inter-kernel gaps rather than by the
    kernels. The first call with a given input shape warms the engine up, captures the method's launches on the current
    stream into a graph whose input and outputs are fixed buffers, and every later call copies the tile in, replays
    the graph with ONE launch and returns copies of the outputs (`clone=False`: the fixed buffers themselves, valid
    until the next call). ViT-S/16, 224^2, B = 1, split-bf16: 0.99 -> 0.83 ms per synchronised call; results are bit
    for bit those of the plain call (same kernels, same order). A new shape, device or precision re-captures, and so
    does a parameter update once a plain call (or `reset()`) has seen it. Keyword arguments are captured by value / by tensor identity and must not change between calls."""

    def __init__(self, model, method, kwargs):
        if not callable(getattr(model, method, None)):
            raise AttributeError(f"{type(model).__name__} has no method {method!r}")
        self.model, self.method, self.kwargs = model, method, dict(kwargs)
        self._cap = None

    def _key(self, x):
        # cheap on purpose (this runs on every call): the engine entry is replaced whenever a plain call finds changed
        # parameters, so its identity stands for the weights; reset() forces a re-capture after an in-place update
        m = self.model
        return (tuple(x.shape), x.dtype, x.device, m.__dict__.get("_engine_epoch", 0), m._precision, m._gray_fold,
                m.pos_embed.data_ptr(), m.pos_embed._version)

    def reset(self):
        """Drop the capture (the next call re-captures): call after updating parameters in place."""
        self._cap = None

    def _capture(self, x, key):
        m, fn = self.model, getattr(self.model, self.method)
        xs = x.detach().clone()
        cur = torch.cuda.current_stream(x.device)
        side = torch.cuda.Stream(device=x.device)
        side.wait_stream(cur)
        m.__dict__["_graph_suspended"] = True  # the plain launch sequence, not the model's own automatic replay
        try:
            with torch.cuda.stream(side):  # warm-up off the capturing stream: engine build, workspace, LDS opt-ins
                for _ in range(2):
                    fn(xs, **self.kwargs)
            cur.wait_stream(side)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                out = fn(xs, **self.kwargs)
        finally:
            m.__dict__["_graph_suspended"] = False
        eng = m._engine(x.device)
        # everything the captured launches point at stays alive with the capture, whatever the engine caches next
        keep = (eng, list(eng._ws.values()), dict(m._pos_cache))
        self._cap = (self._key(x), graph, xs, out, keep)  # the warm-up may have (re)built the engine entry

    @torch.no_grad()
    def __call__(self, x, clone=True):
        _require_hip(x, "input")
        key = self._key(x)
        if self._cap is None or self._cap[0] != key:
            self._capture(x, key)
        _, graph, xs, out, _ = self._cap
        xs.copy_(x)
        graph.replay()
        return _tree_map(torch.clone, out) if clone else out


def vit_tiny(patch_size=16, **kwargs):
    return VisionTransformer(patch_size=patch_size, embed_dim=192, depth=12, num_heads=3, mlp_ratio=4, qkv_bias=True,
                             norm_layer=partial(nn.LayerNorm, eps=1e-6), **kwargs)


def vit_small(patch_size=16, **kwargs):
    return VisionTransformer(patch_size=patch_size, embed_dim=384, depth=12, num_heads=6, mlp_ratio=4, qkv_bias=True,
                             norm_layer=partial(nn.LayerNorm, eps=1e-6), **kwargs)


def vit_base(patch_size=16, **kwargs):
    return VisionTransformer(patch_size=patch_size, embed_dim=768, depth=12, num_heads=12, mlp_ratio=4, qkv_bias=True,
                             norm_layer=partial(nn.LayerNorm, eps=1e-6), **kwargs)
