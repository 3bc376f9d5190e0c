"""Sliding-window post-processing on device against the oracle and, for the
stitcher, against outputs of the reference's own concat_crops (tests/golden/helpers.npz)."""
import numpy as np
import pytest
import torch

from oracle import vit_oracle as O
from tests.helpers import CASES, build_module, case_state_dict, load_golden
from vit_ocm_wmsegmentation_amd import sw_processing as sw
from vit_ocm_wmsegmentation_amd import synth

pytestmark = pytest.mark.gpu


def test_stitcher_bit_exact_vs_reference_golden(dev):
    gold = load_golden("helpers")
    for n in (3, 2):
        rng = np.random.default_rng(int(gold[f"stitch_{n}_seed"]))
        crops = np.stack([rng.random((384, 384), dtype=np.float32) * 255 for _ in range(n * n)])
        got = sw.stitch_windows(torch.from_numpy(crops).to(dev), 128, 384).cpu().numpy()
        assert got.shape == gold[f"stitch_{n}"].shape and np.array_equal(got, gold[f"stitch_{n}"])  # bit-exact


@pytest.mark.parametrize("n,window,stride", [(1, 96, 32), (4, 96, 32), (5, 48, 16), (7, 384, 128)])
def test_stitcher_other_geometries_vs_oracle(dev, n, window, stride):
    rng = np.random.default_rng(n)
    crops = (rng.random((n * n, window, window), dtype=np.float32) * 255)
    got = sw.stitch_windows(torch.from_numpy(crops).to(dev), stride, window).cpu().numpy()
    assert np.array_equal(got, O.concat_crops(crops, stride, window))


def test_tile_postprocess_and_upsample(dev):
    g = torch.Generator().manual_seed(3)
    rows = torch.rand((5, 6, 1, 48 * 48), generator=g) * 0.01
    ref_small = O.tile_postprocess(rows[:, :, 0].numpy())  # (5, 2304) numpy float32 arithmetic
    up = sw.postprocess_windows(rows.to(dev), 48, 48, 8).cpu().numpy()
    assert up.shape == (5, 384, 384)
    # the block centres of the bilinear map are convex combinations dominated by the source pixel; compare the
    # whole map against the restated cv2 geometry (parity unpinned: cv2 absent) to fp32 round-off of 255-range data
    ref_up = O.bilinear_upsample(ref_small.reshape(5, 48, 48), 8)
    assert np.abs(up - ref_up).max() < 2e-4
    # exactness of the pre-resize stage: the min-max map itself, bit for bit (ocm_op_tile_postprocess)
    import ctypes as C
    from vit_ocm_wmsegmentation_amd import _lib
    small = torch.empty((5, 48 * 48), device=dev)
    rd = rows.to(dev)
    _lib.check(_lib.load().ocm_op_tile_postprocess(C.c_void_p(rd.data_ptr()), C.c_void_p(small.data_ptr()), 5, 6, 1, 48 * 48,
                                                   C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    assert np.array_equal(small.cpu().numpy(), ref_small)
    # patch 16: the reference's hard-coded //8 then *8 lands on the nearest-x2 replication of the 24 x 24 map
    rows16 = torch.rand((2, 6, 1, 24 * 24), generator=g) * 0.01
    up16 = sw.postprocess_windows(rows16.to(dev), 24, 24, 16).cpu().numpy()
    small16 = O.tile_postprocess(rows16[:, :, 0].numpy()).reshape(2, 24, 24)
    full16 = np.repeat(np.repeat(small16, 16, 1), 16, 2)  # compute_attention's nearest x16
    ref16 = O.bilinear_upsample(O.cv2_downscale(full16, 8), 8)
    assert up16.shape == (2, 384, 384) and np.abs(up16 - ref16).max() < 2e-4
    with pytest.raises(ValueError):
        sw.postprocess_windows(rows16.to(dev), 24, 24, 12)


def test_heatmap_otsu_mask(dev):
    rng = np.random.default_rng(9)
    heat = np.concatenate([rng.normal(60, 12, 300000), rng.normal(170, 20, 200000)]).astype(np.float32)
    rng.shuffle(heat)
    heat = heat.reshape(500, 1000)
    img, mask, level = sw.otsu_heatmap_mask(torch.from_numpy(heat).to(dev))
    rimg, rmask, rlevel = O.heatmap_mask(heat)
    assert level == rlevel and np.array_equal(img.cpu().numpy(), rimg) and np.array_equal(mask.cpu().numpy(), rmask)
    # independent check of the level: brute-force maximiser of the between-class variance
    hist = np.bincount(rimg.ravel(), minlength=256).astype(np.float64)
    p = hist / hist.sum()
    best, arg = -1.0, 0
    for t in range(255):
        q1, q2 = p[:t + 1].sum(), p[t + 1:].sum()
        if q1 < 1e-7 or q2 < 1e-7:
            continue
        m1 = (np.arange(t + 1) * p[:t + 1]).sum() / q1
        m2 = (np.arange(t + 1, 256) * p[t + 1:]).sum() / q2
        s = q1 * q2 * (m1 - m2) ** 2
        if s > best:
            best, arg = s, t
    assert abs(arg - level) <= 1 and 90 < level < 140


def test_segment_pipeline_vs_oracle(dev):
    """sw_processing.py:223-262 end to end on a small slab with the tiny model: device pipeline vs the oracle
    pipeline fed with the ORACLE's attention (so the comparison includes the bf16 forward error)."""
    case = CASES["tiny_p8"]
    model = build_module(case, dev)
    sd = case_state_dict(case)
    cfg = O.make_cfg(sd, 8, 2)
    window, stride, size = 96, 32, 160  # 3 x 3 windows
    slab = synth.synth_tiles(1, size, seed=5)[0]
    sweep = sw.SlidingWindowAttention(model, window=window, stride=stride, batch_tiles=4)
    out = sweep.segment(slab.to(dev))
    assert out["heat"].shape == (size, size) and out["mask"].shape == (size, size)
    crops = O.sliding_window_crops(slab, stride, window)
    rows = []
    for j in range(crops.shape[0]):
        attn = O.get_last_selfattention(sd, cfg, crops[j:j + 1])
        rows.append(attn[0, :, 0, 1:].numpy())
    ref_small = O.tile_postprocess(np.stack(rows)).reshape(9, 12, 12)
    ref_heat = O.concat_crops(O.bilinear_upsample(ref_small, 8), stride, window)
    # min-max normalisation divides by the (small) dynamic range of near-uniform maps (random-init maps span
    # ~1e-3), amplifying the forward's error: on the 0..255 scale the default split-bf16 mode stays within a tenth
    # of a grey level (single bf16: within 3 grey levels)
    assert np.abs(out["heat"].cpu().numpy() - ref_heat).max() < 0.1
    rimg, rmask, rlevel = O.heatmap_mask(out["heat"].cpu().numpy())  # same heat -> identical mask
    assert out["level"] == rlevel and np.array_equal(out["mask"].cpu().numpy(), rmask)
    # fp32 mode: the heat map itself agrees to round-off
    out32 = sw.SlidingWindowAttention(model.set_precision("fp32"), window=window, stride=stride, batch_tiles=4).segment(
        slab.to(dev))
    assert np.abs(out32["heat"].cpu().numpy() - ref_heat).max() < 2e-2


# ---- eval.py's mask chain: utils.threshold() and the batched per-image pipeline ----
def _smooth_field(rng, size, blobs=6):
    yy, xx = np.mgrid[0:size, 0:size].astype(np.float32)
    f = np.zeros((size, size), np.float32)
    for _ in range(blobs):
        cy, cx, s = rng.uniform(0, size), rng.uniform(0, size), rng.uniform(size / 10, size / 3)
        f += rng.uniform(0.2, 1.0) * np.exp(-((yy - cy) ** 2 + (xx - cx) ** 2) / (2 * s * s))
    return f


@pytest.mark.gpu
@pytest.mark.parametrize("chans", [1, 3])
def test_image_to_gray_u8_matches_pil_restatement(dev, chans):
    from vit_ocm_wmsegmentation_amd.utils import image_to_gray_u8
    rng = np.random.default_rng(3)
    img = rng.uniform(0, 1, size=(chans, 96, 80)).astype(np.float32)
    got, hist = image_to_gray_u8(torch.from_numpy(img).to(dev))
    want = O.to_pil_gray_u8(img)
    assert np.array_equal(got.cpu().numpy(), want)
    assert np.array_equal(hist.cpu().numpy(), np.bincount(want.ravel(), minlength=256))


@pytest.mark.gpu
@pytest.mark.parametrize("flat", [False, True])
def test_threshold_masks_bit_exact_vs_oracle(dev, flat):
    """threshold() of utils.py:61-115 on device against its numpy restatement: all three masks and levels."""
    from vit_ocm_wmsegmentation_amd.utils import threshold
    rng = np.random.default_rng(11)
    S = 192
    gray = np.clip(_smooth_field(rng, S) * 0.4 + rng.uniform(0, 0.05, (S, S)).astype(np.float32), 0, 1).astype(np.float32)
    img = np.repeat(gray[None], 3, 0)
    att = np.full((S, S), 0.25, np.float32) if flat else (_smooth_field(rng, S) * 0.01).astype(np.float32)
    (th, th2, th3), levels, _ = O.threshold_masks(O.to_pil_gray_u8(img), att)
    g1, g2, g3, glev = threshold(torch.from_numpy(img).to(dev), torch.from_numpy(att).to(dev), return_levels=True)
    assert tuple(glev) == tuple(levels)
    assert np.array_equal(g1, th) and np.array_equal(g2, th2) and np.array_equal(g3, th3)
    assert g1.dtype == np.uint8 and set(np.unique(g1)) <= {0, 255}


@pytest.mark.gpu
def test_threshold_rejects_cpu_and_save(dev):
    from vit_ocm_wmsegmentation_amd.utils import threshold
    a = torch.zeros(8, 8)
    with pytest.raises(RuntimeError):
        threshold(torch.zeros(3, 8, 8), a)
    with pytest.raises(NotImplementedError):
        threshold(torch.zeros(3, 8, 8, device=dev), a.to(dev), save=True)


@pytest.mark.gpu
@pytest.mark.parametrize("precision", ["fp32", "bf16x3", "bf16"])
def test_segment_images_matches_oracle_chain(dev, precision):
    """eval.py:126-171 (method ours / otsu / heatmap_threshold) for a batch of tiles against the oracle chain
    run image by image on the CPU, as the reference's loop does."""
    from vit_ocm_wmsegmentation_amd.eval import average_attention_maps, segment_images
    case = CASES["vits16_sharp"]
    model = build_module(case, dev)
    model.set_precision(precision)
    sd = case_state_dict(case)
    p, S, B = case["patch"], 224, 3
    rng = np.random.default_rng(5)
    gray = np.stack([np.clip(_smooth_field(rng, S) * 0.3, 0, 1) for _ in range(B)]).astype(np.float32)
    x = torch.from_numpy(np.repeat(gray[:, None], 3, 1).copy())
    cfg = O.make_cfg(sd, p, 6)
    attn = O.get_last_selfattention(sd, cfg, x).numpy()
    hf = wf = S // p
    maps = average_attention_maps(model, x.to(dev)).cpu().numpy()
    tol = {"fp32": 2e-6, "bf16x3": 4e-6, "bf16": 2e-4}[precision]
    for method, k in (("ours", 0), ("otsu", 1), ("heatmap_threshold", 2)):
        masks, _ = segment_images(model, x.to(dev), method=method, as_numpy=True)
        for b in range(B):
            want_map = O.eval_average_attention(attn[b, :, 0, 1:], hf, wf, p)
            assert np.abs(maps[b] - want_map).max() <= tol
            want = O.threshold_masks(O.to_pil_gray_u8(x[b].numpy()), want_map)[0][k]
            # uint8 truncation and the Otsu level quantise the map: allow a sliver of boundary pixels to flip
            frac = np.mean(masks[b] != want)
            # pixels whose value sits within the map's error of the Otsu level flip: a sliver in the fp32 and the
            # default split-bf16 modes; in single-bf16 mode the 2e-4 map error is a few percent of the dynamic range
            # of these near-uniform synthetic maps (measured: 3.2 % of the pixels)
            assert frac <= (0.0 if k == 1 else (6e-2 if precision == "bf16" else 2e-3)), (method, b, frac)
    with pytest.raises(ValueError):
        segment_images(model, x.to(dev), method="k-means")


# ---- verdict r1 #8: the rest of the post-processing chain (median filter, crops, sw-variant th / th2) ----
def test_median_filter_bit_exact_vs_scipy_fixture(dev, lib):
    """ocm_op_median_filter against outputs of the real scipy.ndimage.median_filter (tests/golden/median.npz)."""
    import ctypes as C
    from tests.golden_cases import MEDIAN_SIZES as SIZES, median_inputs as inputs
    gold = load_golden("median")
    x = torch.from_numpy(inputs(int(gold["seed"]))).to(dev)
    for k in SIZES:
        out = torch.empty_like(x)
        assert lib.ocm_op_median_filter(C.c_void_p(x.data_ptr()), C.c_void_p(out.data_ptr()), x.shape[0], x.shape[1], x.shape[2], k,
                                        C.c_void_p(torch.cuda.current_stream().cuda_stream)) == 0
        assert np.array_equal(out.cpu().numpy(), gold[f"size{k}"]), k


def test_stitched_gray_image_bit_exact_vs_reference_fixture(dev):
    """sw_processing.py:224-227: concat_crops of the uint8 RGB windows + .convert("L"), against the output of the
    reference's own sliding_window / concat_crops / blend functions and PIL (helpers.npz)."""
    gold = load_golden("helpers")
    rng = np.random.default_rng(int(gold["stitch_u8_seed"]))
    img = rng.integers(0, 256, (160, 160), dtype=np.uint8)
    slab = torch.from_numpy(img.astype(np.float32) / 255.0)  # ToTensor
    for planes in (1, 3):
        s = slab[None].expand(planes, -1, -1).contiguous().to(dev)
        gray, hist = sw.stitched_gray_image(s, 32, 96)
        assert np.array_equal(gray.cpu().numpy(), gold["stitch_u8_gray"])
        assert np.array_equal(hist.cpu().numpy(), np.bincount(gold["stitch_u8_gray"].ravel(), minlength=256))
    # (for windows cut from ONE image the truncating blend v*w + v*(1-w) lands back on v: the stitched image equals the
    # source here; the fold / truncation itself is pinned on independent windows in test_oracle_golden.py)
    # a slab whose windows reach past the edge (PIL crop zero fill) against the oracle
    img2 = rng.integers(0, 256, (200, 200), dtype=np.uint8)
    g2, _ = sw.stitched_gray_image(torch.from_numpy(img2.astype(np.float32) / 255.0)[None].to(dev), 32, 96)
    assert np.array_equal(g2.cpu().numpy(), O.stitched_gray_image(img2, 32, 96))


def test_sw_threshold_three_masks_vs_oracle(dev):
    """threshold() of sw_processing.py:37-81: th (cv2 Otsu of image x attention), th2 (skimage Otsu of the image), th3."""
    rng = np.random.default_rng(4)
    S = 256
    img = np.clip(np.concatenate([rng.normal(70, 15, S * S // 2), rng.normal(180, 25, S * S // 2)]), 0, 255).astype(np.uint8)
    rng.shuffle(img)
    img = img.reshape(S, S)
    heat = (_smooth_field(rng, S) * 200).astype(np.float32)
    (th, th2, th3), levels, result = O.sw_threshold_masks(img, heat)
    got = sw.threshold(torch.from_numpy(img).to(dev), torch.from_numpy(heat).to(dev), as_numpy=True)
    assert tuple(got["levels"]) == tuple(levels)
    assert np.array_equal(got["result"], result)
    assert np.array_equal(got["th"], th) and np.array_equal(got["th2"], th2) and np.array_equal(got["th3"], th3)
    assert 100 < levels[1] < 150  # the bimodal image's valley
    # a flat heat map: min_max_normalize returns it unchanged (:32-33)
    flat = np.full((S, S), 0.25, np.float32)
    (_, _, _), lv, res = O.sw_threshold_masks(img, flat)
    g2 = sw.threshold(torch.from_numpy(img).to(dev), torch.from_numpy(flat).to(dev), as_numpy=True)
    assert np.array_equal(g2["result"], res) and tuple(g2["levels"]) == tuple(lv)


def test_segment_returns_reference_masks(dev):
    """SlidingWindowAttention.segment == the oracle's chain fed with the device heat map: stitched image, th, th2, th3."""
    case = CASES["tiny_p8"]
    model = build_module(case, dev)
    window, stride, size = 96, 32, 160
    rng = np.random.default_rng(11)
    img = np.clip(_smooth_field(rng, size) * 140 + rng.uniform(0, 40, (size, size)), 0, 255).astype(np.uint8)
    slab = torch.from_numpy(img.astype(np.float32) / 255.0)[None].expand(3, -1, -1).contiguous()
    out = sw.SlidingWindowAttention(model, window=window, stride=stride, batch_tiles=4).segment(slab.to(dev))
    gray = O.stitched_gray_image(img, stride, window)
    assert np.array_equal(out["gray"].cpu().numpy(), gray)
    (th, th2, th3), levels, result = O.sw_threshold_masks(gray, out["heat"].cpu().numpy())
    assert tuple(out["levels"]) == tuple(levels)
    for k, want in (("th", th), ("th2", th2), ("th3", th3), ("mask", th3), ("result", result)):
        assert np.array_equal(out[k].cpu().numpy(), want), k


@pytest.mark.parametrize("median", [3, 13])
def test_segment_images_median_filter_and_crops(dev, median):
    """eval.py --median_filter k and --crop 4: one batched forward for all crops, then the reference's per-crop median,
    utils.concat_crops tiling and resize chain (oracle, crop by crop at B = 1 like the reference's loop)."""
    from vit_ocm_wmsegmentation_amd.eval import average_attention_maps, segment_images
    case = CASES["tiny_p8"]
    model = build_module(case, dev)
    sd = case_state_dict(case)
    cfg = O.make_cfg(sd, 8, 2)
    p, s, B = 8, 48, 2
    rng = np.random.default_rng(6)
    crops = torch.from_numpy(np.stack([np.clip(_smooth_field(rng, s) * 0.3, 0, 1) for _ in range(B * 4)]).astype(np.float32))
    crops = crops.reshape(B, 4, 1, s, s).expand(-1, -1, 3, -1, -1).contiguous()
    hf = s // p
    maps = average_attention_maps(model, crops.to(dev), median_filter=median).cpu().numpy()
    single = average_attention_maps(model, crops[:, 0].to(dev), median_filter=median).cpu().numpy()
    for b in range(B):
        rows = [O.get_last_selfattention(sd, cfg, crops[b, j:j + 1])[0, :, 0, 1:].numpy() for j in range(4)]
        want = O.eval_crops_average_attention(rows, hf, hf, p, median)
        assert maps[b].shape == (2 * s, 2 * s) and np.abs(maps[b] - want).max() <= 1e-6
        assert np.abs(single[b] - O.eval_average_attention(rows[0], hf, hf, p, median)).max() <= 1e-6
    masks, _ = segment_images(model, crops.to(dev), method="ours", median_filter=median, as_numpy=True)
    from vit_ocm_wmsegmentation_amd.eval import tile_crops_image
    gray = tile_crops_image(crops)
    for b in range(B):
        want = O.threshold_masks(O.to_pil_gray_u8(gray[b].numpy()), maps[b])[0][0]
        assert np.mean(masks[b] != want) == 0.0  # same map in, same mask out
    # a 13 x 13 window holds more foreign than own-block pixels and really changes the map; a 3 x 3 window around the
    # block-centre pixels the down-scale samples lies inside one constant block: the identity
    base = average_attention_maps(model, crops.to(dev), median_filter=1).cpu().numpy()
    assert (np.abs(maps - base).max() > 0) == (median == 13)


@pytest.mark.parametrize("T,h,w,rep", [(3, 14, 14, 16), (2, 48, 48, 8), (1, 5, 7, 3), (4, 6, 6, 1)])
def test_nearest_upsample_is_index_replication(dev, lib, T, h, w, rep):
    """ocm_op_nearest_upsample: dst[t][y][x] = src[t][y // rep][x // rep] — bit for bit what np.repeat /
    F.interpolate(mode="nearest") with an integer factor give (utils.py:233 compute_attention, the //8 *8 block values of
    sw_processing.py:255-257, the patch mask of model.py:71)."""
    import ctypes as C
    from vit_ocm_wmsegmentation_amd import _lib
    g = torch.Generator().manual_seed(3)
    src = torch.randn((T, h, w), generator=g).to(dev)
    dst = torch.full((T, h * rep, w * rep), float("nan"), device=dev)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    _lib.check(lib.ocm_op_nearest_upsample(C.c_void_p(src.data_ptr()), C.c_void_p(dst.data_ptr()), T, h, w, rep, st))
    want = src.repeat_interleave(rep, 1).repeat_interleave(rep, 2)
    assert torch.equal(dst, want)
    assert lib.ocm_op_nearest_upsample(C.c_void_p(src.data_ptr()), C.c_void_p(dst.data_ptr()), T, h, w, 0, st) == _lib.OCM_EINVAL
