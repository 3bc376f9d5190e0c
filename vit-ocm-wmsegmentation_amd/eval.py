"""The per-image body of the reference's eval.py `validate` loop (eval.py:126-177, default --crop 1,
--median_filter 1) as a batched device pipeline: ViT forward -> CLS-row attention of the last block ->
head mean -> bilinear upsample to the image size -> threshold() masks. The reference runs it one image
at a time with the maps going through numpy / cv2 on the host; here only the final masks leave the GPU.
"""
import torch

from . import _lib
from .engine import _p, _stream
from .utils import threshold

METHODS = {"ours": 0, "otsu": 1, "heatmap_threshold": 2}  # index into threshold()'s (th, th2, th3)


@torch.no_grad()
def average_attention_maps(model, images):
    """images: (B,C,S,S) float32 on the HIP device. Returns (B,S,S) fp32: eval.py:136-166 per image —
    compute_attention(query=0) -> np.mean over heads -> resize down by p -> cv2 INTER_LINEAR up to (S,S)."""
    if images.dim() != 4 or images.shape[-1] != images.shape[-2]:
        raise ValueError("eval's resize to (img.shape[-1], img.shape[-1]) assumes square images")
    rows = model.get_last_attention_rows(images)  # (B, heads, 1, hf*wf) = attentions[0][:, :, 0, 1:]
    B, Hh, nr, P = rows.shape
    p = model.patch_embed.patch_size
    hf = wf = images.shape[-1] // p
    lib = _lib.load()
    small = torch.empty((B, hf, wf), dtype=torch.float32, device=rows.device)
    big = torch.empty((B, hf * p, wf * p), dtype=torch.float32, device=rows.device)
    with torch.cuda.device(rows.device):
        _lib.check(lib.ocm_op_head_mean(_p(rows), _p(small), B, Hh, nr, P, _stream()))
        _lib.check(lib.ocm_op_bilinear_upsample(_p(small), _p(big), B, hf, wf, p, _stream()))
    return big


@torch.no_grad()
def segment_images(model, images, method="ours", median_filter=1, as_numpy=False):
    """The mask eval.py scores for `method` in {"ours", "otsu", "heatmap_threshold"} for every image of a
    batch. Returns (masks (B,S,S) uint8 in {0,255}, average_attentions (B,S,S) fp32)."""
    if method not in METHODS:
        raise ValueError(f"method {method!r} is not on this path (k-means / chan-vese stay on the host in the reference)")
    if int(median_filter) != 1:
        raise ValueError("only the reference's default --median_filter 1 (identity) is on this path")
    maps = average_attention_maps(model, images)
    masks = torch.empty(maps.shape, dtype=torch.uint8, device=maps.device)
    for b in range(images.shape[0]):
        masks[b] = threshold(images[b], maps[b], as_numpy=False)[METHODS[method]]
    return (masks.cpu().numpy(), maps.cpu().numpy()) if as_numpy else (masks, maps)
