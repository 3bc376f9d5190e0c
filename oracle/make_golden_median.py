"""Pins oracle.vit_oracle.median_filter (and with it the device kernel ocm_op_median_filter) against the real
scipy.ndimage.median_filter the reference calls (eval.py:144,158). scipy IS importable in the build container, so this
post-processing step is pinned rather than restated: run `python oracle/make_golden_median.py` there; the fixture
tests/golden/median.npz holds scipy's outputs only (inputs are regenerated from the seed)."""
import os
import sys

import numpy as np
from scipy.ndimage import median_filter

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import vit_oracle as O  # noqa: E402

SIZES = (2, 3, 4, 5, 7)


def inputs(seed=17):
    rng = np.random.default_rng(seed)
    smooth = rng.random((2, 6, 5), dtype=np.float32)
    up = np.repeat(np.repeat(smooth, 4, axis=1), 4, axis=2)  # block-constant like a nearest-upsampled attention map
    noisy = rng.random((2, 24, 20), dtype=np.float32)
    noisy[0, 3:9, 2:7] = 0.5  # ties
    return np.concatenate([up, noisy], 0)


def main():
    x = inputs()
    out = {"seed": np.int64(17)}
    import scipy
    out["scipy_version"] = np.array(scipy.__version__)
    for k in SIZES:
        ref = np.stack([median_filter(x[t], size=k) for t in range(x.shape[0])])
        assert np.array_equal(ref, O.median_filter(x, k)), f"oracle median_filter != scipy at size {k}"
        out[f"size{k}"] = ref
    path = os.path.join(ROOT, "tests", "golden", "median.npz")
    np.savez_compressed(path, **out)
    print(f"median_filter sizes {SIZES} pinned against scipy {scipy.__version__} -> {os.path.relpath(path, ROOT)}")


if __name__ == "__main__":
    main()
