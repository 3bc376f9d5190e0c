"""Output contract of the attention map: the hot-path part of the reference's utils.py and the
region-query index of analyse_attention.py, on device.

  compute_attention   reference utils.py:229-235
  region_query_index  reference analyse_attention.py:192-195, 234-236
  threshold           reference utils.py:55-115 (the three Otsu masks of eval.py's "ours" / "otsu" /
                      "heatmap_threshold" methods), on device
"""
import ctypes as C

import numpy as np
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


def _otsu_from_hist(hist_dev, count):
    """256-bin device histogram -> OpenCV-style Otsu level (a 256-step scalar loop: done on the host)."""
    h = hist_dev.cpu().numpy().astype(np.uint64)
    level = int(_lib.load().ocm_otsu_threshold(h.ctypes.data_as(C.POINTER(C.c_uint64)), int(count)))
    if level < 0:
        raise ValueError("ocm_otsu_threshold failed")
    return level


def skimage_otsu_from_hist(hist_dev):
    """skimage.filters.threshold_otsu (0.19.3, sw_processing.py:57) from a 256-bin device histogram of a uint8 image:
    integer bins over [min, max], float64 class weights / means, first maximum of the between-class variance
    (a 256-step scalar computation: on the host, like the reference). scikit-image is not installable here:
    restated from its source, parity unpinned."""
    h = hist_dev.cpu().numpy().astype(np.float64)
    nz = np.nonzero(h)[0]
    if nz.size == 0:
        raise ValueError("empty histogram")
    lo, hi = int(nz[0]), int(nz[-1])
    if lo == hi:
        return lo
    counts, centers = h[lo:hi + 1], np.arange(lo, hi + 1)
    weight1 = np.cumsum(counts)
    weight2 = np.cumsum(counts[::-1])[::-1]
    mean1 = np.cumsum(counts * centers) / weight1
    mean2 = (np.cumsum((counts * centers)[::-1]) / weight2[::-1])[::-1]
    variance12 = weight1[:-1] * weight2[1:] * (mean1[:-1] - mean2[1:]) ** 2
    return int(centers[int(np.argmax(variance12))])


def histogram_u8(img_u8):
    """256-bin int64 histogram of a uint8 HIP tensor (ocm_op_histogram_u8)."""
    _require_hip(img_u8, "img")
    img_u8 = img_u8.contiguous()
    hist = torch.empty(256, dtype=torch.int64, device=img_u8.device)
    with torch.cuda.device(img_u8.device):
        _lib.check(_lib.load().ocm_op_histogram_u8(_p(img_u8), img_u8.numel(), _p(hist), _stream()))
    return hist


def image_to_gray_u8(img):
    """transform(img.squeeze(0)).convert("L") of eval.py:166 on device: (C,H,W) or (1,C,H,W) float tensor in
    [0,1] with C in {1,3} -> ((H,W) uint8 tensor, 256-bin int64 histogram)."""
    _require_hip(img, "img")
    if img.dim() == 4:
        img = img[0]
    if img.dim() != 3 or img.shape[0] not in (1, 3) or img.dtype != torch.float32:
        raise ValueError(f"expected a float32 (C,H,W) tensor with C in (1,3), got {img.dtype} {tuple(img.shape)}")
    if img.stride(2) != 1 or img.stride(1) != img.shape[2]:
        img = img.contiguous()
    out = torch.empty(img.shape[1:], dtype=torch.uint8, device=img.device)
    hist = torch.empty(256, dtype=torch.int64, device=img.device)
    with torch.cuda.device(img.device):
        _lib.check(_lib.load().ocm_op_image_to_gray_u8(_p(img), int(img.stride(0)), int(img.shape[0]), out.numel(),
                                                       _p(out), _p(hist), _stream()))
    return out, hist


def threshold(img, attention, output_directory="", save=False, name=None, as_numpy=True, return_levels=False):
    """utils.py:61-115 on device. img: the float (C,H,W) image tensor (what the reference turns into a PIL "L"
    image first) or an (H,W) uint8 tensor; attention: (H,W) float32 heat map. Returns (th, th2, th3):
    Otsu mask of the 0.6/0.4 image/attention blend, of the image, and of the attention — numpy uint8 arrays as
    the reference returns them (as_numpy=False keeps them on the device). Saving figures (save=True) is
    the reference's matplotlib side effect and is not part of this path."""
    if save:
        raise NotImplementedError("threshold(save=True) writes figures with matplotlib in the reference; not on this path")
    _require_hip(attention, "attention")
    if attention.dtype != torch.float32 or attention.dim() != 2:
        raise ValueError(f"expected a float32 (H,W) attention map, got {attention.dtype} {tuple(attention.shape)}")
    attention = attention.contiguous()
    lib, dev, n = _lib.load(), attention.device, attention.numel()
    if img.dtype == torch.uint8:
        _require_hip(img, "img")
        img_u8 = img.contiguous()
        hist_img = histogram_u8(img_u8)
    else:
        img_u8, hist_img = image_to_gray_u8(img)
    if tuple(img_u8.shape) != tuple(attention.shape):
        raise ValueError(f"image {tuple(img_u8.shape)} and attention {tuple(attention.shape)} differ in size")
    att_u8 = torch.empty(attention.shape, dtype=torch.uint8, device=dev)
    res_u8 = torch.empty(attention.shape, dtype=torch.uint8, device=dev)
    masks = torch.empty((3,) + tuple(attention.shape), dtype=torch.uint8, device=dev)
    scratch = torch.empty(2048, dtype=torch.uint8, device=dev)
    hist_att = torch.empty(256, dtype=torch.int64, device=dev)
    hist_res = torch.empty(256, dtype=torch.int64, device=dev)
    alpha = 0.4
    with torch.cuda.device(dev):
        st = _stream()
        _lib.check(lib.ocm_op_normalize_u8(_p(attention), n, _p(scratch), _p(att_u8), _p(hist_att), st))
        _lib.check(lib.ocm_op_blend_u8(_p(img_u8), _p(att_u8), n, alpha, 1 - alpha, _p(res_u8), _p(hist_res), st))
        levels = (_otsu_from_hist(hist_res, n), _otsu_from_hist(hist_img, n), _otsu_from_hist(hist_att, n))
        for k, src in enumerate((res_u8, img_u8, att_u8)):
            _lib.check(lib.ocm_op_threshold_u8(_p(src), _p(masks[k]), n, levels[k], st))
    out = tuple(m.cpu().numpy() for m in masks) if as_numpy else tuple(masks)
    return (out + (levels,)) if return_levels else out
