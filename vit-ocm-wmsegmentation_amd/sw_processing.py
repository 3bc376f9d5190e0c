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


def _dev_call(t):
    return torch.cuda.device(t.device)


def postprocess_windows(rows, hf, wf, patch_size):
    """rows: (T, heads, n_rows, hf*wf) CLS-row maps on a HIP device. Per window, as the tile loop of
    sw_processing.py:245,253-257 does on the CPU: head mean -> (v - min) / (max - min) * 255 -> cv2.resize DOWN BY 8
    of the nearest-upsampled (x patch_size) map -> cv2.resize INTER_LINEAR UP BY 8 (the reference hard-codes the 8s).
    For patch 8 the down-scaled map is the hf x wf map itself; for patch 16 every down-scaled pixel samples inside
    one 16 x 16 block, so it is the hf x wf map nearest-upsampled by 2. Returns (T, hf*p, wf*p) fp32."""
    if patch_size % 8:
        raise ValueError(f"the reference's //8 then *8 resize needs a patch size that is a multiple of 8, got {patch_size}")
    rep, scale = patch_size // 8, 8
    _require = rows.is_cuda and rows.dtype == torch.float32
    if not _require:
        raise RuntimeError("postprocess_windows needs a float32 tensor on a HIP device (no CPU fallback)")
    rows = rows.contiguous()
    T, H, nr, P = rows.shape
    if P != hf * wf:
        raise ValueError(f"rows have {P} pixels, expected {hf}x{wf}")
    lib = _lib.load()
    small = torch.empty((T, hf, wf), dtype=torch.float32, device=rows.device)
    big = torch.empty((T, hf * rep * scale, wf * rep * scale), dtype=torch.float32, device=rows.device)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    with _dev_call(rows):
        _lib.check(lib.ocm_op_tile_postprocess(C.c_void_p(rows.data_ptr()), C.c_void_p(small.data_ptr()), T, H, nr, P, st))
        if rep > 1:  # pure index replication (the block values the //8 resize lands on)
            blocks = torch.empty((T, hf * rep, wf * rep), dtype=torch.float32, device=rows.device)
            _lib.check(lib.ocm_op_nearest_upsample(C.c_void_p(small.data_ptr()), C.c_void_p(blocks.data_ptr()), T, hf, wf, rep, st))
            small = blocks
        _lib.check(lib.ocm_op_bilinear_upsample(C.c_void_p(small.data_ptr()), C.c_void_p(big.data_ptr()), T, hf * rep,
                                                wf * rep, scale, st))
    return big


def stitch_windows(crops, stride, window):
    """concat_crops (sw_processing.py:113-149) on device: crops (n*n, window, window) fp32 in row-major
    window order -> (S, S), S = window + (n-1)*stride. Bit-exact with the reference's sequential stitcher."""
    if not (crops.is_cuda and crops.dtype == torch.float32):
        raise RuntimeError("stitch_windows needs a float32 tensor on a HIP device (no CPU fallback)")
    crops = crops.contiguous()
    n = int(round(crops.shape[0] ** 0.5))
    if n * n != crops.shape[0] or crops.shape[1] != window or crops.shape[2] != window:
        raise ValueError(f"expected (n*n, {window}, {window}) crops, got {tuple(crops.shape)}")
    S = window + (n - 1) * stride
    ramp = torch.from_numpy(np.linspace(1, 0, window - stride)).to(crops.device)  # float64, as the reference builds it
    out = torch.empty((S, S), dtype=torch.float32, device=crops.device)
    with _dev_call(crops):
        _lib.check(_lib.load().ocm_op_stitch(C.c_void_p(crops.data_ptr()), C.c_void_p(out.data_ptr()),
                                             C.c_void_p(ramp.data_ptr()), n, window, stride,
                                             C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    return out


def stitched_gray_image(slab, stride, window):
    """sw_processing.py:224-227 on device: concat_crops of the uint8 RGB windows (float64 blends truncated into the
    uint8 overlap at every fold) -> .convert("L"). slab: (C,H,W) float32 HIP tensor in [0,1] (the ToTensor image, C in
    {1,3}). Returns ((S,S) uint8 image, 256-bin int64 histogram), S = window + (n-1)*stride. Bit-exact against the
    reference's own functions (tests/golden/helpers.npz)."""
    if not (slab.is_cuda and slab.dtype == torch.float32 and slab.dim() == 3 and slab.shape[0] in (1, 3)):
        raise RuntimeError("stitched_gray_image needs a float32 (C,H,W) tensor, C in (1,3), on a HIP device")
    if slab.stride(2) != 1:
        slab = slab.contiguous()
    n = window_count(slab.shape[1], stride)
    if n <= 0 or n != window_count(slab.shape[2], stride):
        raise ValueError("concat_crops stitches square window grids")
    S = window + (n - 1) * stride
    ramp = torch.from_numpy(np.linspace(1, 0, window - stride)).to(slab.device)
    out = torch.empty((S, S), dtype=torch.uint8, device=slab.device)
    hist = torch.empty(256, dtype=torch.int64, device=slab.device)
    with _dev_call(slab):
        _lib.check(_lib.load().ocm_op_stitch_image_u8(
            C.c_void_p(slab.data_ptr()), int(slab.stride(0)), int(slab.stride(1)), int(slab.shape[0]), int(slab.shape[1]),
            int(slab.shape[2]), C.c_void_p(out.data_ptr()), C.c_void_p(ramp.data_ptr()), n, window, stride,
            C.c_void_p(hist.data_ptr()), C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    return out, hist


def threshold(img_u8, heat, hist_img=None, as_numpy=False):
    """threshold() of sw_processing.py:37-81 (save=False) on device: img_u8 (S,S) uint8 "L" image, heat (S,S) float32
    stitched attention. Returns dict(th, th2, th3, result, attention, levels):
      result = (img * attention / max(attention)).astype(uint8), attention = min_max_normalize(heat)     :42-46
      th  = cv2 Otsu mask of result (:53), th2 = img > skimage Otsu level of img (:55-58), th3 = cv2 Otsu mask of
      attention * 255 (:60). The Otsu levels are 256-step scalar loops over device histograms, evaluated on the host."""
    from .utils import _otsu_from_hist, histogram_u8, skimage_otsu_from_hist
    if not (heat.is_cuda and heat.dtype == torch.float32 and img_u8.is_cuda and img_u8.dtype == torch.uint8):
        raise RuntimeError("threshold needs a uint8 image and a float32 heat map on a HIP device (no CPU fallback)")
    if tuple(img_u8.shape) != tuple(heat.shape):
        raise ValueError(f"image {tuple(img_u8.shape)} and heat map {tuple(heat.shape)} differ in size")
    heat, img_u8 = heat.contiguous(), img_u8.contiguous()
    lib, dev, n = _lib.load(), heat.device, heat.numel()
    if hist_img is None:
        hist_img = histogram_u8(img_u8)
    res = torch.empty(heat.shape, dtype=torch.uint8, device=dev)
    att = torch.empty(heat.shape, dtype=torch.uint8, device=dev)
    masks = torch.empty((3,) + tuple(heat.shape), dtype=torch.uint8, device=dev)
    scratch = torch.empty(2048, dtype=torch.uint8, device=dev)
    h_res = torch.empty(256, dtype=torch.int64, device=dev)
    h_att = torch.empty(256, dtype=torch.int64, device=dev)
    vp = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
    with _dev_call(heat):
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        _lib.check(lib.ocm_op_weighted_u8(vp(heat), vp(img_u8), n, vp(scratch), vp(res), vp(att), vp(h_res), vp(h_att), st))
        levels = (_otsu_from_hist(h_res, n), skimage_otsu_from_hist(hist_img), _otsu_from_hist(h_att, n))
        for k, src in enumerate((res, img_u8, att)):
            _lib.check(lib.ocm_op_threshold_u8(vp(src), vp(masks[k]), n, levels[k], st))
    out = dict(th=masks[0], th2=masks[1], th3=masks[2], result=res, attention=att, levels=levels)
    if as_numpy:
        out = {k: (v.cpu().numpy() if torch.is_tensor(v) else v) for k, v in out.items()}
    return out


def otsu_heatmap_mask(heat):
    """The heat-map branch of threshold() (sw_processing.py:43-48,62): min_max_normalize -> *255 -> uint8 ->
    cv2.threshold(THRESH_BINARY + THRESH_OTSU). Returns (uint8 image, uint8 mask in {0,255}, level)."""
    if not (heat.is_cuda and heat.dtype == torch.float32):
        raise RuntimeError("otsu_heatmap_mask needs a float32 tensor on a HIP device (no CPU fallback)")
    heat = heat.contiguous()
    lib = _lib.load()
    dev, n = heat.device, heat.numel()
    img = torch.empty(heat.shape, dtype=torch.uint8, device=dev)
    mask = torch.empty(heat.shape, dtype=torch.uint8, device=dev)
    scratch = torch.empty(2048, dtype=torch.uint8, device=dev)
    hist = torch.empty(256, dtype=torch.int64, device=dev)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    with _dev_call(heat):
        _lib.check(lib.ocm_op_normalize_u8(C.c_void_p(heat.data_ptr()), n, C.c_void_p(scratch.data_ptr()),
                                           C.c_void_p(img.data_ptr()), C.c_void_p(hist.data_ptr()), st))
        h = hist.cpu().numpy().astype(np.uint64)  # 2 KiB D2H: the level is a 256-step scalar loop
        level = int(lib.ocm_otsu_threshold(h.ctypes.data_as(C.POINTER(C.c_uint64)), n))
        if level < 0:
            raise ValueError("ocm_otsu_threshold failed")
        _lib.check(lib.ocm_op_threshold_u8(C.c_void_p(img.data_ptr()), C.c_void_p(mask.data_ptr()), n, level, st))
    return img, mask, level


class SlidingWindowAttention:
    """CLS-row attention maps of every window of a slab.

    model   : vit_ocm_wmsegmentation_amd.dino.vision_transformer.VisionTransformer on a HIP device
    window  : window side in pixels (reference: 384);  stride: 128
    batch_tiles : windows per forward on each rank (the reference uses 1), or "auto": at most `max_batch`, sized so that
                  the token rows of a forward fill whole rounds of the GPU's CUs (auto_batch_plan)
    """

    def __init__(self, model, window=384, stride=128, batch_tiles="auto", group=None, max_batch=24):
        self.model, self.window, self.stride, self.batch_tiles, self.group = model, window, stride, batch_tiles, group
        self.max_batch = max_batch

    # ---- the two device-touching steps, overridable (tests/test_sw_gloo.py drives __call__ on CPU through them) ----
    def _geometry(self, device):
        """(patch size, heads) of the model and the (N, D) positional table of one window on `device`."""
        m = self.model
        eng = m._engine(device)
        hf = self.window // eng.p
        return eng.p, eng.H, m._pos_for(hf * hf, self.window, self.window, device)

    def _forward_batch(self, slab, origins_dev, nb, pos, query_rows):
        """CLS-row (or `query_rows`) maps of `nb` windows of the resident slab: (nb, heads, n_rows, hf*wf) fp32. The
        windows are gathered in place through their origins (no crops are materialised)."""
        eng = self.model._engine(slab.device)
        out = eng.forward_tiles(slab, (0, slab.stride(0), slab.stride(1)), origins_dev, nb, self.window, self.window, pos,
                                flags=_lib.OCM_OUT_ROWS | _lib.OCM_LAST_ATTN_ONLY, query_rows=query_rows)
        return out["rows"]

    @staticmethod
    def batch_plan(count, batch_tiles):
        """Split `count` windows into ceil(count / batch_tiles) batches whose sizes differ by at most one (113 windows
        at 16 per forward -> 15 + 7 x 14 rather than 7 x 16 + a lone B = 1 forward that costs 4x its share)."""
        if count <= 0:
            return []
        nb = -(-count // max(1, batch_tiles))
        base, extra = divmod(count, nb)
        return [base + (1 if i < extra else 0) for i in range(nb)]

    @staticmethod
    def auto_batch_plan(count, n_tokens, cus, max_batch=24, dim=384):
        """Balanced batches (sizes differ by at most one, none above `max_batch`) whose count is chosen for the kernels
        rather than fixed. A forward is charged the rows its two MLP GEMMs (equal FLOPs, together the largest share of a
        forward) really occupy the chip for, priced by the tiles the engine dispatches in split-bf16 precision
        (csrc/gemm_kernels.h, launch_linear_epi):
          mlp.fc1 (N = 4 dim): 128 x 128 tiles, two workgroups per CU -> whole rounds of ceil(rows / 128) * (4 dim / 128)
                               tiles over 2 * CUs slots;
          mlp.fc2 (N = dim):   128 x 192 tiles, one workgroup per CU, when dim is one or two such tiles wide and either the
                               128 x 128 grid has fewer than 512 tiles or the 192-wide grid fills its last round >= 10 %
                               better; otherwise 128 x 128 tiles, two per CU.
        The plan with the lowest charge wins (ties: fewer forwards). Measured on one MI355X (split-bf16, 900 windows of 2305
        tokens): 16 .. 32 windows per forward all land within 522 .. 536 ms per sweep; a plan that spills fc1 into a
        nearly empty extra round (22 windows: 9.1 rounds) costs 3-7 %.
        (Round 4: from 16 384 rows mlp.fc1 runs 256 x 256 tiles, one per CU. The charge still prices the 128 x 128 grid on
        purpose: priced on the 256-row grid the model prefers 50 forwards of 18 windows, which MEASURES 533.1 ms per sweep
        against 528.9 ms for the 43 forwards of 21 this pricing picks — 16, 21 and 24 windows per forward are within 0.6 % of
        each other, the per-forward overhead decides.)"""
        if count <= 0:
            return []
        cus = max(1, cus)

        def rounds(tiles, slots):
            return -(-tiles // slots)

        def charge(b):
            rows = b * n_tokens
            rb = -(-rows // 128)
            cols1 = max(1, 4 * dim // 128)
            fc1 = rounds(rb * cols1, 2 * cus) * (2 * cus) * 128 / cols1  # rows the launch occupies the chip for
            t128 = rb * -(-dim // 128)
            fc2 = rounds(t128, 2 * cus) * (2 * cus) * 128 / -(-dim // 128)
            if dim % 192 == 0 and dim // 192 <= 2 and rows >= 4096:
                t192 = rb * (dim // 192)
                e192 = t192 / (cus * rounds(t192, cus))
                e128 = t128 / (2 * cus * rounds(t128, 2 * cus))
                if t128 < 512 or e192 > 1.1 * e128:
                    fc2 = rounds(t192, cus) * cus * 128 / (dim // 192)
            return 0.5 * (fc1 + fc2)

        best = None
        nb_min = -(-count // max(1, max_batch))
        for nb in range(nb_min, min(count, nb_min + max(2, nb_min // 2)) + 1):  # up to 1.5x the fewest forwards
            base, extra = divmod(count, nb)
            plan = [base + (1 if i < extra else 0) for i in range(nb)]
            cost = sum(charge(b) + 1500 for b in plan)  # + launch overhead of a forward, in rows (~0.3 ms)
            if best is None or (cost, nb) < best[0]:
                best = ((cost, nb), plan)
        return best[1]

    def _plan(self, count, n_tokens, device):
        if self.batch_tiles == "auto":
            cus = torch.cuda.get_device_properties(device).multi_processor_count if device.type == "cuda" else 256
            return self.auto_batch_plan(count, n_tokens, cus, self.max_batch, self._geometry(device)[2].shape[1])
        return self.batch_plan(count, self.batch_tiles)

    @torch.no_grad()
    def __call__(self, slab, query_rows=None):
        """slab: (C, H, W) or (1, C, H, W) fp32 HIP tensor. Returns (T, heads, n_rows, hf, wf) fp32
        on every rank, T = windows in row-major order, hf = wf = window // patch."""
        if slab.dim() == 4:
            slab = slab[0]
        if slab.stride(2) != 1:
            slab = slab.contiguous()
        m, dev = self.model, slab.device
        p, Hh, pos = self._geometry(dev)
        if getattr(m, "_gray_fold", False) and slab.shape[0] == 3:
            slab = slab[:1]
        origins = sliding_window_origins(slab.shape[1], slab.shape[2], self.stride)
        T = origins.shape[0]
        if T == 0:
            raise ValueError(f"a {slab.shape[1]}x{slab.shape[2]} slab has no windows at stride {self.stride} "
                             "(range(0, size - 2*stride, stride) is empty)")
        if self.stride % 4:
            raise ValueError("the window gather needs 16-byte aligned rows: stride must be a multiple of 4 pixels")
        # Windows may reach past the slab (side not a multiple of the stride, or window > 3*stride): the reference's
        # PIL crop zero-fills that part (sw_processing.py:157-160). The gather kernel reads origin + window unguarded,
        # so the slab is zero-padded up to the farthest window edge, rows kept 16-byte aligned.
        need_h = int(origins[:, 0].max()) + self.window
        need_w = int(origins[:, 1].max()) + self.window
        pad_h, pad_w = max(0, need_h - slab.shape[1]), max(0, need_w - slab.shape[2])
        pad_w += (-(slab.shape[2] + pad_w)) % 4
        if pad_h or pad_w or slab.shape[2] % 4:
            slab = torch.nn.functional.pad(slab, (0, pad_w, 0, pad_h)).contiguous()
        elif slab.stride(1) % 4 or slab.stride(0) % 4 or slab.data_ptr() % 16:
            # a view whose rows are unit-stride but do not START on 16-byte boundaries (big[:, :, 1:4097], odd row
            # pitch): the gather's float4 loads need aligned rows, so take an aligned copy
            slab = slab.contiguous()
        world = dist.get_world_size(self.group) if dist.is_available() and dist.is_initialized() else 1
        rank = dist.get_rank(self.group) if world > 1 else 0
        begin, end, share = shard_range(T, world, rank)
        hf = wf = self.window // p
        nq = 1 if query_rows is None else int(query_rows.numel())
        local = torch.zeros((share, Hh, nq, hf * wf), dtype=torch.float32, device=dev)
        dev_origins = torch.from_numpy(origins[begin:end]).to(dev)
        s = 0
        for nb in self._plan(end - begin, hf * wf + 1, dev):
            local[s:s + nb] = self._forward_batch(slab, dev_origins[s:s + nb].contiguous(), nb, pos, query_rows)
            s += nb
        maps = gather_tile_maps(local, T, self.group)
        return maps.reshape(T, Hh, nq, hf, wf)

    @torch.no_grad()
    def segment(self, slab):
        """The whole of sw_processing.py:223-262 on device: windows -> CLS-row maps (sharded, all-gathered)
        -> per-window head mean / min-max / down-up resize -> overlap-blended stitch of the maps AND of the uint8
        image -> threshold(): the three masks the reference returns. Returns dict(
          maps, heat (S,S) fp32 stitched attention, gray (S,S) uint8 stitched "L" image,
          th (Otsu of image x attention), th2 (skimage Otsu of the image), th3 = mask (Otsu of the heat map),
          result, image (= attention * 255 as uint8), level (of th3), levels (all three))."""
        if slab.dim() == 4:
            slab = slab[0]
        maps = self(slab)
        T, Hh, _, hf, wf = maps.shape
        n = int(round(T ** 0.5))
        if n * n != T:
            raise ValueError("segment() stitches square window grids (as concat_crops does)")
        up = postprocess_windows(maps.reshape(T, Hh, 1, hf * wf), hf, wf, self.window // hf)
        heat = stitch_windows(up, self.stride, self.window)
        gray, hist_gray = stitched_gray_image(slab, self.stride, self.window)
        t = threshold(gray, heat, hist_img=hist_gray)
        return dict(maps=maps, heat=heat, gray=gray, th=t["th"], th2=t["th2"], th3=t["th3"], mask=t["th3"],
                    result=t["result"], image=t["attention"], level=t["levels"][2], levels=t["levels"])
