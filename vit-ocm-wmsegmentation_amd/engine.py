"""Plumbing between torch-ROCm tensors and the C ABI (include/ocm_vit.h).

torch is used for what the boundary needs and nothing else: device memory for inputs,
outputs and workspace, the current HIP stream, and `torch.distributed`. All arithmetic
of the hot path happens inside libocm_vit.so.
"""
import ctypes as C
import os

import torch

from . import _lib
from ._lib import (OCM_LAST_ATTN_ONLY, OCM_OUT_ATTN, OCM_OUT_FEAT, OCM_OUT_FMAP, OCM_OUT_QKV, OCM_OUT_ROWS, OCM_USE_GRAPH,
                   OCM_OUT_TOKENS, OcmVitConfig, OcmVitIO, check)


def _require_hip(t, what):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise RuntimeError(
            f"{what} must live on a HIP device (got {getattr(t, 'device', type(t))}). This path runs only "
            "through the gfx950 HIP extension; there is no CPU fallback.")


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def to_operand(x32, precision):
    """fp32 HIP tensor -> the contraction kernels' operand type for `precision` (an _lib.OCM_PREC_* value):
    bf16 tensor, the fp32 tensor itself, or split-bf16 pairs (an int32 tensor of the same shape: 4 bytes per
    element, every 32 consecutive elements of the last axis stored as [32 x hi | 32 x lo])."""
    _require_hip(x32, "operand")
    x32 = x32.detach().to(torch.float32).contiguous()
    if precision == _lib.OCM_PREC_FP32:
        return x32
    lib = _lib.load()
    with torch.cuda.device(x32.device):
        if precision == _lib.OCM_PREC_BF16:
            out = torch.empty(x32.shape, dtype=torch.bfloat16, device=x32.device)
            check(lib.ocm_op_cast_bf16(_p(x32), _p(out), x32.numel(), _stream()))
        else:
            if x32.shape[-1] % 32:
                raise ValueError(f"split-bf16 operands need a last axis that is a multiple of 32, got {tuple(x32.shape)}")
            out = torch.empty(x32.shape, dtype=torch.int32, device=x32.device)
            check(lib.ocm_op_cast_split(_p(x32), _p(out), x32.numel(), _stream()))
    return out


def from_split(xs):
    """split-bf16 pairs (int32 tensor as made by to_operand) -> fp32 values hi + lo."""
    out = torch.empty(xs.shape, dtype=torch.float32, device=xs.device)
    with torch.cuda.device(xs.device):
        check(_lib.load().ocm_op_merge_split(_p(xs), _p(out), xs.numel(), _stream()))
    return out


class Engine:
    """Owns one ocm_vit_t handle (packed bf16/fp32 parameter copies in HBM) for one device."""

    # hip_graph: False (default) / True / "auto" (graph replay when a forward has at most this many tokens).
    # Measured on MI355X at B = 1 (ViT-S/16, 224^2): 0.79 ms per call with or without the graph — the ~83 kernels
    # of a forward are bound by the GPU-side kernel boundaries (about 7 us each), not by host launch time — so
    # the replay path is opt-in: it pays only where the host is the slow side.
    GRAPH_AUTO_TOKENS = 8192

    def __init__(self, *, patch_size, in_chans, embed_dim, depth, num_heads, mlp_hidden, ln_eps, qk_scale,
                 device, precision=_lib.PRECISIONS[_lib.DEFAULT_PRECISION]):
        self.lib = _lib.load()
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("Engine needs a HIP device (torch device type 'cuda' on ROCm); no CPU fallback.")
        self.cfg = OcmVitConfig(patch_size, in_chans, embed_dim, depth, num_heads, mlp_hidden, ln_eps, qk_scale,
                                precision, 0)
        self.D, self.H, self.L, self.p = embed_dim, num_heads, depth, patch_size
        self.hd = embed_dim // num_heads
        self._h = C.c_void_p(0)
        with torch.cuda.device(self.device):
            check(self.lib.ocm_vit_create(C.byref(self.cfg), C.byref(self._h)))
        self._ws = {}
        self.option_epoch = 0  # bumped by every set_* option: captured launch sequences (module-level replay) are stale then
        self.hip_graph = {"1": True, "auto": "auto"}.get(os.environ.get("OCM_HIP_GRAPH", "0"), False)
        if os.environ.get("OCM_FUSE_LN"):  # "auto" / "never" / "always" (A/B runs; the default is "auto")
            self.set_fuse_layernorm(os.environ["OCM_FUSE_LN"])
        if os.environ.get("OCM_FOLD_LN"):  # "0" switches the folded LayerNorm off (A/B runs)
            self.set_fold_layernorm({"0": False, "2": "always"}.get(os.environ["OCM_FOLD_LN"], True))

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self.lib.ocm_vit_destroy(self._h)
            self._h = C.c_void_p(0)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- parameters -------------------------------------------------------------------------
    def set_param(self, name, tensor, keep=None):
        """Upload one state_dict entry (reference key name, reference layout). The source may be a temporary copy that
        must outlive the asynchronous cast kernel: pass a list as `keep` to collect such temporaries and synchronise ONCE
        after the last upload (`Engine.flush(keep)`), otherwise the stream is synchronised here."""
        t = tensor.detach().to(device=self.device, dtype=torch.float32).contiguous()
        with torch.cuda.device(self.device):
            check(self.lib.ocm_vit_set_param(self._h, name.encode(), _p(t), t.numel(), _stream()))
            if keep is None:
                torch.cuda.current_stream().synchronize()
            else:
                keep.append(t)

    def flush(self, keep):
        """Wait for the uploads whose sources `keep` holds, then let them go."""
        with torch.cuda.device(self.device):
            torch.cuda.current_stream().synchronize()
        keep.clear()

    def n_tokens(self, tile_h, tile_w):
        return (tile_h // self.p) * (tile_w // self.p) + 1

    def workspace(self, batch, n):
        key = (batch, n)
        ws = self._ws.get(key)
        if ws is None:
            nbytes = self.lib.ocm_vit_workspace_bytes(self._h, batch, n)
            ws = torch.empty(nbytes + 256, dtype=torch.uint8, device=self.device)
            self._ws = {key: ws}  # keep only the latest shape resident
        off = (-ws.data_ptr()) % 256
        return ws.data_ptr() + off, ws.numel() - off

    # ---- forward ----------------------------------------------------------------------------
    def _io(self, image, strides, origins, batch, tile_h, tile_w, pos):
        io = OcmVitIO()
        io.image = image.data_ptr()
        io.img_stride_b, io.img_stride_c, io.img_stride_y = strides
        io.tile_origins = origins.data_ptr() if origins is not None else None
        io.batch, io.tile_h, io.tile_w = batch, tile_h, tile_w
        io.pos_embed = pos.data_ptr()
        io.stream = torch.cuda.current_stream().cuda_stream
        return io

    def forward_tiles(self, image, strides, origins, batch, tile_h, tile_w, pos, *, flags, n_last=1,
                      query_rows=None, patch_mask=None):
        """One ocm_vit_forward. `image` is an fp32 HIP tensor holding the planes, `strides`
        = (batch, channel, row) element strides, `origins` an int32 (B,2) HIP tensor or None.
        Returns a dict with the requested outputs (fresh tensors)."""
        _require_hip(image, "image")
        _require_hip(pos, "pos_embed")
        if image.dtype != torch.float32:
            raise ValueError(f"image must be float32 (got {image.dtype})")
        n = self.n_tokens(tile_h, tile_w)
        if tuple(pos.shape) != (n, self.D) or pos.dtype != torch.float32 or not pos.is_contiguous():
            raise AssertionError(f"pos_embed must be a contiguous float32 ({n},{self.D}) tensor, got {tuple(pos.shape)}")
        B, D, H = batch, self.D, self.H
        out = {}
        with torch.cuda.device(self.device):
            io = self._io(image, strides, origins, B, tile_h, tile_w, pos)
            io.flags, io.n_last = flags, n_last
            kw = dict(dtype=torch.float32, device=self.device)
            if flags & OCM_OUT_FEAT:
                out["feat"] = torch.empty((n_last, B, n, D), **kw)
                io.out_feat = out["feat"].data_ptr()
            if flags & OCM_OUT_ATTN:
                out["attn"] = torch.empty((n_last, B, H, n, n), **kw)
                io.out_attn = out["attn"].data_ptr()
            if flags & OCM_OUT_QKV:
                out["qkv"] = torch.empty((n_last, 3, B, H, n, self.hd), **kw)
                io.out_qkv = out["qkv"].data_ptr()
            if flags & OCM_OUT_TOKENS:
                out["tokens"] = torch.empty((B, n, D), **kw)
                io.out_tokens = out["tokens"].data_ptr()
            if flags & OCM_OUT_ROWS:
                if query_rows is None:  # the CLS row: one resident index tensor per engine, not a fill kernel per forward
                    query_rows = self.__dict__.get("_cls_row")
                    if query_rows is None:
                        query_rows = self.__dict__["_cls_row"] = torch.zeros(1, dtype=torch.int32, device=self.device)
                _require_hip(query_rows, "query_rows")
                query_rows = query_rows.to(torch.int32).contiguous()
                out["rows"] = torch.empty((B, H, query_rows.numel(), n - 1), **kw)
                io.query_rows, io.n_rows = query_rows.data_ptr(), query_rows.numel()
                io.out_rows = out["rows"].data_ptr()
            if flags & OCM_OUT_FMAP:
                out["fmap"] = torch.empty((B, D, tile_h // self.p, tile_w // self.p), **kw)
                io.out_fmap = out["fmap"].data_ptr()
            if patch_mask is not None:
                _require_hip(patch_mask, "mask")
                patch_mask = patch_mask.reshape(B, -1).to(torch.float32).contiguous()
                if patch_mask.shape[1] != n - 1:
                    raise ValueError(f"mask has {patch_mask.shape[1]} entries per image, expected {n - 1}")
                io.patch_mask = patch_mask.data_ptr()
            io.workspace, io.workspace_bytes = self.workspace(B, n)
            # launch-bound regime (the reference's one-tile-per-call loops): replay a cached hipGraph
            if self.hip_graph is True or (self.hip_graph == "auto" and B * n <= self.GRAPH_AUTO_TOKENS):
                io.flags = flags | OCM_USE_GRAPH
            check(self.lib.ocm_vit_forward(self._h, C.byref(io)))
        return out

    def set_fuse_layernorm(self, mode):
        """Per-handle dispatch option OCM_OPT_FUSE_LN: "auto" (default: where it is faster), "never", "always" (whenever the
        embedding width has the full-row GEMM + LayerNorm kernel). All three give bit-identical results."""
        value = {"auto": 0, "never": 1, "always": 2}[mode]
        check(self.lib.ocm_vit_set_option(self._h, _lib.OCM_OPT_FUSE_LN, value))
        self.option_epoch += 1

    def set_fold_layernorm(self, on):
        """Per-handle option OCM_OPT_FOLD_LN (split-bf16 engines): True / "auto" (default) hands the residual stream to the
        qkv / fc1 GEMMs un-normalised and finishes the LayerNorm in their epilogues wherever that is faster (forwards of
        fewer than ~22 k token rows at D = 384); "always" does so in every forward; False runs LayerNorm kernels (or the
        fused GEMM + LayerNorm kernels, see set_fuse_layernorm)."""
        value = 2 if on == "always" else 0 if on in (True, "auto") else 1
        check(self.lib.ocm_vit_set_option(self._h, _lib.OCM_OPT_FOLD_LN, value))
        self.option_epoch += 1

    def graph_stats(self):
        """(replays, captures) of the hipGraph path."""
        a, b = C.c_uint64(0), C.c_uint64(0)
        check(self.lib.ocm_vit_graph_stats(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def forward(self, x, pos, **kw):
        """x: (B, C, Hpx, Wpx) fp32 HIP tensor (any strides with unit x stride)."""
        _require_hip(x, "input")
        if x.dim() != 4:
            raise ValueError(f"expected a (B,C,H,W) tensor, got {tuple(x.shape)}")
        if x.stride(3) != 1:
            x = x.contiguous()
        B, Cc, Hpx, Wpx = x.shape
        if Cc != self.cfg.in_chans:
            raise ValueError(f"input has {Cc} channels, engine was built for {self.cfg.in_chans}")
        return self.forward_tiles(x, (x.stride(0), x.stride(1), x.stride(2)), None, B, Hpx, Wpx, pos, **kw)

    def prepare_tokens(self, x, pos):
        _require_hip(x, "input")
        if x.stride(3) != 1:
            x = x.contiguous()
        B, Cc, Hpx, Wpx = x.shape
        n = self.n_tokens(Hpx, Wpx)
        with torch.cuda.device(self.device):
            io = self._io(x, (x.stride(0), x.stride(1), x.stride(2)), None, B, Hpx, Wpx, pos)
            out = torch.empty((B, n, self.D), dtype=torch.float32, device=self.device)
            check(self.lib.ocm_vit_prepare_tokens(self._h, C.byref(io), _p(out)))
        return out

    def block_forward(self, index, x, *, want_attn=False, want_qkv=False, attn_only=False):
        """Block.forward on x (B,N,D) fp32; returns (x_out | None, attn | None, qkv | None)."""
        _require_hip(x, "x")
        B, n, D = x.shape
        xo = x.detach().to(torch.float32).contiguous().clone()
        flags = (OCM_OUT_ATTN if want_attn else 0) | (OCM_OUT_QKV if want_qkv else 0) | \
                (OCM_LAST_ATTN_ONLY if attn_only else 0)
        kw = dict(dtype=torch.float32, device=self.device)
        attn = torch.empty((B, self.H, n, n), **kw) if want_attn else None
        qkv = torch.empty((3, B, self.H, n, self.hd), **kw) if want_qkv else None
        with torch.cuda.device(self.device):
            ws, wsb = self.workspace(B, n)
            check(self.lib.ocm_vit_block_forward(self._h, index, _p(xo), B, n, flags, _p(attn), _p(qkv), ws, wsb,
                                                 _stream()))
        return (None if attn_only else xo), attn, qkv

    def final_norm(self, x):
        _require_hip(x, "x")
        xc = x.detach().to(torch.float32).contiguous()
        y = torch.empty_like(xc)
        with torch.cuda.device(self.device):
            check(self.lib.ocm_vit_final_norm(self._h, _p(xc), _p(y), xc.numel() // self.D, _stream()))
        return y
