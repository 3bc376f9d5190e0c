"""The per-image body of the reference's eval.py `validate` loop (eval.py:126-177) as a batched device pipeline:
ViT forward -> CLS-row attention of the last block -> head mean -> [median filter] -> [tiling of the crops] -> down /
up resize to the image size -> threshold() masks. The reference runs it one image (and, with --crop 4 / 16, one
crop) at a time at B = 1 with the maps going through numpy / scipy / cv2 on the host; here every image and crop of the
batch goes through ONE forward and only the final masks leave the GPU.
"""
import torch

from . import _lib
from .engine import _p, _stream
from .utils import threshold

METHODS = {"ours": 0, "otsu": 1, "heatmap_threshold": 2}  # index into threshold()'s (th, th2, th3)


def _head_mean_small(model, tiles, median_filter):
    """(T,C,s,s) tiles -> (T, s/p, s/p) fp32: compute_attention(query 0) -> np.mean over heads -> scipy
    median_filter(size) -> cv2.resize down by p (eval.py:136-144,169). The median runs on the nearest-upsampled
    (T,s,s) map like the reference's; for size 1 (the default) that chain is the identity on the s/p x s/p map."""
    rows = model.get_last_attention_rows(tiles)  # (T, heads, 1, hf*wf) = attentions[0][:, :, 0, 1:]
    T, Hh, nr, P = rows.shape
    p = model.patch_embed.patch_size
    hf = wf = tiles.shape[-1] // p
    lib = _lib.load()
    small = torch.empty((T, hf, wf), dtype=torch.float32, device=rows.device)
    with torch.cuda.device(rows.device):
        _lib.check(lib.ocm_op_head_mean(_p(rows), _p(small), T, Hh, nr, P, _stream()))
        k = int(median_filter)
        if k != 1:
            up = torch.empty((T, hf * p, wf * p), dtype=torch.float32, device=rows.device)  # nearest x p: index replication
            _lib.check(lib.ocm_op_nearest_upsample(_p(small), _p(up), T, hf, wf, p, _stream()))
            filt = torch.empty_like(up)
            _lib.check(lib.ocm_op_median_filter(_p(up), _p(filt), T, hf * p, wf * p, k, _stream()))
            _lib.check(lib.ocm_op_downscale_centre(_p(filt), _p(small), T, hf * p, wf * p, p, _stream()))
    return small


def _upsample(small, p):
    T, h, w = small.shape
    big = torch.empty((T, h * p, w * p), dtype=torch.float32, device=small.device)
    with torch.cuda.device(small.device):
        _lib.check(_lib.load().ocm_op_bilinear_upsample(_p(small.contiguous()), _p(big), T, h, w, p, _stream()))
    return big


@torch.no_grad()
def average_attention_maps(model, images, median_filter=1):
    """images: (B,C,S,S) float32 on the HIP device — or (B,crops,C,s,s) for the reference's --crop 4 / 16 datasets
    (eval.py:146-167: every crop through the model, the maps tiled by utils.concat_crops). Returns (B,S,S) fp32:
    eval.py:136-171 per image — compute_attention(query=0) -> np.mean over heads -> median_filter -> [tile] -> resize
    down by p -> cv2 INTER_LINEAR up to (S,S)."""
    p = model.patch_embed.patch_size
    if images.dim() == 5:
        B, n, Cc, s, s2 = images.shape
        r = int(round(n ** 0.5))
        if r * r != n or s != s2:
            raise ValueError("the crop path tiles a square grid of square crops (utils.concat_crops)")
        small = _head_mean_small(model, images.reshape(B * n, Cc, s, s), median_filter)  # ONE forward for all crops
        hs = small.shape[-1]
        small = small.reshape(B, r, r, hs, hs).permute(0, 1, 3, 2, 4).reshape(B, r * hs, r * hs)  # concat_crops
        return _upsample(small, p)
    if images.dim() != 4 or images.shape[-1] != images.shape[-2]:
        raise ValueError("eval's resize to (img.shape[-1], img.shape[-1]) assumes square images")
    return _upsample(_head_mean_small(model, images, median_filter), p)


def tile_crops_image(images):
    """eval.py:160-166: img = concat_crops(images[i, :, 0, :, :]) replicated to three planes — the (B,1,S,S) image the
    crop path thresholds (plane 0 of every crop, tiled)."""
    B, n, _, s, _ = images.shape
    r = int(round(n ** 0.5))
    return images[:, :, 0].reshape(B, r, r, s, s).permute(0, 1, 3, 2, 4).reshape(B, 1, r * s, r * s).contiguous()


@torch.no_grad()
def segment_images(model, images, method="ours", median_filter=1, as_numpy=False):
    """The mask eval.py scores for `method` in {"ours", "otsu", "heatmap_threshold"} for every image of a batch
    ((B,C,S,S), or (B,crops,C,s,s) for --crop 4 / 16), with --median_filter `median_filter`.
    Returns (masks (B,S,S) uint8 in {0,255}, average_attentions (B,S,S) fp32)."""
    if method not in METHODS:
        raise ValueError(f"method {method!r} is not on this path (k-means / chan-vese stay on the host in the reference)")
    if not 1 <= int(median_filter) <= 15:
        raise ValueError("median_filter must be in 1..15")
    maps = average_attention_maps(model, images, median_filter)
    gray_src = tile_crops_image(images) if images.dim() == 5 else images
    masks = torch.empty(maps.shape, dtype=torch.uint8, device=maps.device)
    for b in range(images.shape[0]):
        masks[b] = threshold(gray_src[b], maps[b], as_numpy=False)[METHODS[method]]
    return (masks.cpu().numpy(), maps.cpu().numpy()) if as_numpy else (masks, maps)
