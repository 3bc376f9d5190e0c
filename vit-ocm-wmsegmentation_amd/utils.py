"""Output contract of the attention map: the hot-path part of the reference's utils.py and the
region-query index of analyse_attention.py, on device.

  compute_attention   reference utils.py:229-235
  region_query_index  reference analyse_attention.py:192-195, 234-236
"""
import ctypes as C

import torch

from . import _lib
from .engine import _p, _require_hip, _stream


def compute_attention(attentions, query, w_featmap, h_featmap, patch_size):
    """attentions: list whose first entry is the (B, H, N, N) fp32 attention of the last block
    (what get_intermediate_feat returns). Returns (maps, nh) with maps a numpy array
    (nh, w_featmap*p, h_featmap*p): row `query` of batch element 0 with the CLS column dropped,
    reshaped (w_featmap, h_featmap) row-major and nearest-neighbour upsampled by patch_size.
    The gather + upsample is one HIP kernel (ocm_op_attention_map); the D2H copy is the only
    device->host transfer of the inference path, as in the reference."""
    attn = attentions[0]
    _require_hip(attn, "attentions[0]")
    if attn.dim() != 4 or attn.dtype != torch.float32:
        raise ValueError(f"expected a float32 (B, H, N, N) tensor, got {attn.dtype} {tuple(attn.shape)}")
    attn = attn.contiguous()
    nh, n = attn.shape[1], attn.shape[2]
    query = int(query)
    maps = torch.empty((nh, w_featmap * patch_size, h_featmap * patch_size), dtype=torch.float32, device=attn.device)
    with torch.cuda.device(attn.device):
        _lib.check(_lib.load().ocm_op_attention_map(_p(attn), _p(maps), 0, nh, n, query, w_featmap, h_featmap,
                                                    patch_size, _stream()))
    return maps.cpu().numpy(), nh


def region_query_index(py, px, patch_size, w_featmap):
    """Token index (CLS excluded, i.e. the value passed as `query` minus nothing: the reference
    indexes attentions[..., query, 1:] with this number directly) of the patch containing pixel
    (py, px): int(py // p * w_featmap + px // p)  — analyse_attention.py:192. Pure integer math."""
    return int(py // patch_size * w_featmap + px // patch_size)


def grid_query_index(i, j, w_featmap, rate):
    """analyse_attention.py:234: query of grid cell (i, j) at sub-sampling `rate`."""
    return int(i * w_featmap * rate + j * rate)
