"""Sliding-window attention maps over a large OCM slab, sharded across the GPUs of a node.

Reference: sw_processing.py — `sliding_window` (:151-163) cuts 384-px windows at stride 128 and a
serial Python loop (:235-258) pushes them through the model one at a time (B=1) on one GPU.
Here the slab stays resident in HBM, the windows are NOT materialised (the patch-embedding kernel
gathers each window straight from the slab through per-tile origins), every rank runs batched
forwards over its contiguous block of windows, only the CLS-row maps (B, H, hf*wf) are produced
(never the (H, N, N) matrices: 127 MB per window at N=2305), and ONE all-gather (RCCL over xGMI
on GPUs, gloo in the CPU tests) leaves every rank with all T maps in the reference's row-major
window order.  Windows are independent, so there is no other collective on the data path.
"""
import ctypes as C

import numpy as np
import torch
import torch.distributed as dist

from . import _lib


def window_count(size, stride=128):
    """len(range(0, size - 2*stride, stride)) — windows per axis (sw_processing.py:156-157)."""
    return int(_lib.load().ocm_sw_count(int(size), int(stride)))


def sliding_window_origins(height, width, stride=128):
    """(T, 2) int32 array of (y0, x0), row-major — the crop boxes of sw_processing.py:151-163."""
    lib = _lib.load()
    cap = max(1, window_count(height, stride) * window_count(width, stride))
    buf = (C.c_int32 * (2 * cap))()
    n = lib.ocm_sw_origins(int(height), int(width), int(stride), buf, cap)
    if n < 0:
        raise ValueError(f"ocm_sw_origins failed with code {-n}")
    return np.ctypeslib.as_array(buf)[: 2 * n].reshape(n, 2).copy()


def shard_range(n_tiles, world, rank):
    """Contiguous block partition: returns (begin, end, share) with share =
    ceil(n_tiles / world) the padded per-rank count of the equal-size collective."""
    b, e = C.c_int32(), C.c_int32()
    share = _lib.load().ocm_sw_shard(int(n_tiles), int(world), int(rank), C.byref(b), C.byref(e))
    if share < 0:
        raise ValueError(f"bad shard request n_tiles={n_tiles} world={world} rank={rank}")
    return b.value, e.value, share


def gather_tile_maps(local, n_tiles, group=None):
    """All-gather the per-rank maps. `local` is (share, ...) with rows past this rank's tile count
    zero-padded; returns (n_tiles, ...) in global tile order on every rank. Single-process (no
    initialised process group) returns local[:n_tiles]."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return local[:n_tiles]
    world = dist.get_world_size(group)
    out = torch.empty((world * local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, local.contiguous(), group=group)
    return out[:n_tiles]  # rank r's block starts at r*share; only the tail of the last blocks is padding


class SlidingWindowAttention:
    """CLS-row attention maps of every window of a slab.

    model   : vit_ocm_wmsegmentation_amd.dino.vision_transformer.VisionTransformer on a HIP device
    window  : window side in pixels (reference: 384);  stride: 128
    batch_tiles : windows per forward on each rank (the reference uses 1)
    """

    def __init__(self, model, window=384, stride=128, batch_tiles=16, group=None):
        self.model, self.window, self.stride, self.batch_tiles, self.group = model, window, stride, batch_tiles, group

    @torch.no_grad()
    def __call__(self, slab, query_rows=None):
        """slab: (C, H, W) or (1, C, H, W) fp32 HIP tensor. Returns (T, heads, n_rows, hf, wf) fp32
        on every rank, T = windows in row-major order, hf = wf = window // patch."""
        if slab.dim() == 4:
            slab = slab[0]
        if slab.stride(2) != 1:
            slab = slab.contiguous()
        m, dev = self.model, slab.device
        eng = m._engine(dev)
        p, Hh = eng.p, eng.H
        if m._gray_fold and slab.shape[0] == 3:
            slab = slab[:1]
        origins = sliding_window_origins(slab.shape[1], slab.shape[2], self.stride)
        T = origins.shape[0]
        world = dist.get_world_size(self.group) if dist.is_available() and dist.is_initialized() else 1
        rank = dist.get_rank(self.group) if world > 1 else 0
        begin, end, share = shard_range(T, world, rank)
        hf = wf = self.window // p
        nq = 1 if query_rows is None else int(query_rows.numel())
        local = torch.zeros((share, Hh, nq, hf * wf), dtype=torch.float32, device=dev)
        pos = m._pos_for(hf * wf, self.window, self.window, dev)
        dev_origins = torch.from_numpy(origins[begin:end]).to(dev)
        strides = (0, slab.stride(0), slab.stride(1))
        for s in range(0, end - begin, self.batch_tiles):
            nb = min(self.batch_tiles, end - begin - s)
            out = eng.forward_tiles(slab, strides, dev_origins[s:s + nb].contiguous(), nb, self.window, self.window, pos,
                                    flags=_lib.OCM_OUT_ROWS | _lib.OCM_LAST_ATTN_ONLY, query_rows=query_rows)
            local[s:s + nb] = out["rows"]
        maps = gather_tile_maps(local, T, self.group)
        return maps.reshape(T, Hh, nq, hf, wf)
