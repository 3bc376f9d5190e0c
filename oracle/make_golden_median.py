"""Pins oracle.vit_oracle.median_filter (and with it the device kernel ocm_op_median_filter) against the real
scipy.ndimage.median_filter the reference calls (eval.py:144,158). scipy IS importable in the build container, so this
post-processing step is pinned rather than restated: run `python oracle/make_golden_median.py` there; the fixture
tests/golden/median.npz holds scipy's outputs only (inputs are regenerated from the seed)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import vit_oracle as O  # noqa: E402
from tests.golden_cases import MEDIAN_SIZES as SIZES, median_inputs as inputs  # noqa: E402  (scipy-free, shared with the tests)


def main():
    from scipy.ndimage import median_filter  # only the fixture writer needs scipy
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
