"""The oracle (oracle/vit_oracle.py, the CPU restatement) against the golden vectors that
oracle/make_golden.py captured from the REAL reference module. Runs anywhere (CPU only)."""
import numpy as np
import pytest
import torch

from oracle import vit_oracle as O
from tests.golden_cases import SATURATED
from tests.helpers import CASES, case_dims, case_inputs, case_state_dict, load_golden

# fp32 CPU arithmetic re-run on possibly different host cores / thread counts: summation order of
# MKL / oneDNN kernels may differ, so allow a few ulps of the O(1) activations.
TOL = 2e-5


@pytest.mark.parametrize("name", list(CASES))
def test_oracle_matches_reference_golden(name):
    case, gold = CASES[name], load_golden(name)
    assert float(gold["oracle_vs_reference_maxabs"]) <= 1e-6  # pinned when the fixture was made
    sd = case_state_dict(case)
    cfg = O.make_cfg(sd, case["patch"], case_dims(case)[2])
    for idx, x in enumerate(case_inputs(case)):
        pfx = f"in{idx}_"
        n = case["n"]
        feat, attns, qkvs = O.get_intermediate_feat(sd, cfg, x, n)
        a = attns[-1]
        # the saturated set is ill-conditioned: another summation order of the same fp32 arithmetic (other core count /
        # BLAS blocking) moves it by up to ~5e-4 (fp32 vs float64: tools/precision_study.py)
        atol = 2e-3 if name in SATURATED else TOL
        np.testing.assert_allclose(a[:, :, 0, 1:].numpy(), gold[pfx + "cls_rows"], rtol=0, atol=atol)
        np.testing.assert_allclose(a[:, :, a.shape[-1] // 2, :].numpy(), gold[pfx + "mid_rows"], rtol=0, atol=atol)
        assert np.array_equal(a[:, :, 0, 1:].mean(1).argmax(-1).numpy(), gold[pfx + "argmax"])  # indices: exact
        np.testing.assert_allclose(feat[-1][:, :4, :16].numpy(), gold[pfx + "feat_head"], rtol=0, atol=10 * TOL)
        assert abs(float(feat[-1].double().abs().sum()) / float(gold[pfx + "feat_abssum"]) - 1) < 1e-5
        np.testing.assert_allclose(qkvs[-1][:, :, :, :3, :8].numpy(), gold[pfx + "qkv_head"], rtol=0, atol=10 * TOL)
        tokens = O.prepare_tokens(sd, cfg, x)
        np.testing.assert_allclose(tokens[:, :3, :16].numpy(), gold[pfx + "tokens_head"], rtol=0, atol=TOL)
        assert abs(float(tokens.double().abs().sum()) / float(gold[pfx + "tokens_abssum"]) - 1) < 1e-6
        if name in ("tiny_p8", "vits16_init"):  # the cheap cases also pin the other entry points
            assert torch.equal(O.get_last_selfattention(sd, cfg, x), a)
            np.testing.assert_allclose(O.forward_feats(sd, cfg, x)[:, 0, :32].numpy(), gold[pfx + "cls_out"],
                                       rtol=0, atol=10 * TOL)
        if case.get("full"):
            for j in range(n):
                np.testing.assert_allclose(feat[j].numpy(), gold[pfx + f"feat{j}"], rtol=0, atol=10 * TOL)
                np.testing.assert_allclose(attns[j].numpy(), gold[pfx + f"attn{j}"], rtol=0, atol=TOL)
                np.testing.assert_allclose(qkvs[j].numpy(), gold[pfx + f"qkv{j}"], rtol=0, atol=10 * TOL)


def test_compute_attention_contract():
    """utils.py:229-235: batch 0, one query row, CLS column dropped, (w,h) row-major, nearest x p."""
    B, H, wf, hf, p = 2, 3, 4, 5, 8
    N = wf * hf + 1
    attn = torch.arange(B * H * N * N, dtype=torch.float32).reshape(B, H, N, N)
    maps, nh = O.compute_attention([attn], 7, wf, hf, p)
    assert nh == H and maps.shape == (H, wf * p, hf * p)
    for head in range(H):
        for ty in range(wf):
            for tx in range(hf):
                blockv = maps[head, ty * p:(ty + 1) * p, tx * p:(tx + 1) * p]
                assert (blockv == attn[0, head, 7, 1 + ty * hf + tx].item()).all()
    assert O.region_query_index(37, 90, 16, 14) == 37 // 16 * 14 + 90 // 16


def test_sliding_window_origins_restatement():
    """sw_processing.py:151-163 on the sizes the reference / BASELINE use."""
    o = O.sliding_window_origins(1152, 1152, 128)
    assert len(o) == 49 and o[0] == (0, 0) and o[1] == (0, 128) and o[7] == (128, 0) and o[-1] == (768, 768)
    o = O.sliding_window_origins(4096, 4096, 128)
    assert len(o) == 900 and o[-1] == (3712, 3712) and o[-1][0] + 384 == 4096
    assert O.sliding_window_origins(256, 256, 128) == []  # size - 2*stride <= 0: no window
    assert O.sliding_window_origins(384, 384, 128) == [(0, 0)]


def test_helpers_match_reference_golden():
    """compute_attention / sliding_window / concat_crops: the oracle against outputs of the REFERENCE's own
    functions (tests/golden/helpers.npz, written by oracle/make_golden.py::run_helpers)."""
    gold = load_golden("helpers")
    attn = torch.from_numpy(gold["ca_attn"])
    for query in (0, 9):
        maps, nh = O.compute_attention([attn], query, 5, 7, 8)
        assert nh == 3 and np.array_equal(maps, gold[f"ca_maps_q{query}"])
    for size in (640, 1152):
        assert [tuple(o) for o in gold[f"sw_origins_{size}"].tolist()] == O.sliding_window_origins(size, size, 128)
    for n in (3, 2):
        rng = np.random.default_rng(int(gold[f"stitch_{n}_seed"]))
        crops = np.stack([rng.random((384, 384), dtype=np.float32) * 255 for _ in range(n * n)])
        assert np.array_equal(O.concat_crops(crops, 128, 384), gold[f"stitch_{n}"])


def test_oracle_threshold_chain_properties():
    """utils.py:61-115 restatement: PIL grey conversion is the identity on R=G=B planes, a flat attention map
    leaves min_max_normalize inactive, and Otsu separates a clean bimodal image between its two modes."""
    rng = np.random.default_rng(0)
    g = rng.uniform(0, 1, (1, 40, 40)).astype(np.float32)
    u1 = O.to_pil_gray_u8(g)
    assert np.array_equal(u1, O.to_pil_gray_u8(np.repeat(g, 3, 0)))
    assert np.array_equal(u1, (g[0] * np.float32(255)).astype(np.uint8))
    img = np.where(np.arange(1600).reshape(40, 40) % 3 == 0, 200, 40).astype(np.uint8)
    (th, th2, th3), (l1, l2, l3), result = O.threshold_masks(img, np.full((40, 40), 0.5, np.float32))
    assert 40 <= l2 < 200 and np.array_equal(th2, np.where(img > l2, 255, 0))
    assert np.array_equal(result, ((img / 2) * 0.6 + (127 / 2) * 0.4).astype(np.uint8))  # flat map: 0.5*255 -> 127
    assert set(np.unique(th3)) <= {0, 255}


def test_median_filter_restatement_equals_scipy_fixture():
    """oracle.median_filter against outputs of the real scipy.ndimage.median_filter (eval.py:144,158) — fixture written by
    oracle/make_golden_median.py; when scipy is importable where the tests run, against scipy itself too."""
    import numpy as np
    from oracle import vit_oracle as O
    from tests.golden_cases import MEDIAN_SIZES as SIZES, median_inputs as inputs
    from tests.helpers import load_golden
    gold = load_golden("median")
    x = inputs(int(gold["seed"]))
    for k in SIZES:
        assert np.array_equal(O.median_filter(x, k), gold[f"size{k}"]), k
    try:
        from scipy.ndimage import median_filter
    except ImportError:
        return
    for k in (2, 3, 6):
        assert np.array_equal(O.median_filter(x[:1], k)[0], median_filter(x[0], size=k))


def test_uint8_stitcher_restatement_equals_reference_fixture():
    """oracle.stitched_gray_image / concat_crops on uint8 RGB windows against the reference's own sliding_window +
    concat_crops + PIL convert (helpers.npz: stitch_u8_*)."""
    import numpy as np
    from oracle import vit_oracle as O
    from tests.helpers import load_golden
    gold = load_golden("helpers")
    rng = np.random.default_rng(int(gold["stitch_u8_seed"]))
    img = rng.integers(0, 256, (160, 160), dtype=np.uint8)
    assert np.array_equal(O.stitched_gray_image(img, 32, 96), gold["stitch_u8_gray"])
    rnd = [rng.integers(0, 256, (96, 96, 3), dtype=np.uint8) for _ in range(9)]
    assert np.array_equal(O.pil_rgb_to_l(O.concat_crops(rnd, 32, 96)), gold["stitch_u8_random"])


def test_post_chain_restatements_are_self_consistent():
    """cv2 / skimage steps are restated (parity unpinned): properties that must hold whatever the library version."""
    import numpy as np
    from oracle import vit_oracle as O
    rng = np.random.default_rng(2)
    small = rng.random((3, 6, 5), dtype=np.float32)
    for p in (8, 16):
        up = np.repeat(np.repeat(small, p, 1), p, 2)
        assert np.array_equal(O.cv2_downscale(up, p), small)      # block-constant map: the centre average is the value
        assert np.array_equal(O.cv2_downscale(np.repeat(np.repeat(small, 16, 1), 16, 2), 8), np.repeat(np.repeat(small, 2, 1), 2, 2))
    img = np.concatenate([rng.integers(40, 90, 5000), rng.integers(150, 220, 5000)]).astype(np.uint8).reshape(100, 100)
    lvl = O.skimage_otsu_level(img)
    assert 89 <= lvl < 150 and abs(lvl - O.otsu_level(img)) <= 1  # both maximise the same between-class variance
    assert O.skimage_otsu_level(np.full((4, 4), 7, np.uint8)) == 7
    tiles = [np.full((2, 3), i, np.float32) for i in range(4)]
    assert np.array_equal(O.plain_concat_crops(tiles), np.array([[0, 0, 0, 1, 1, 1]] * 2 + [[2, 2, 2, 3, 3, 3]] * 2, np.float32))
