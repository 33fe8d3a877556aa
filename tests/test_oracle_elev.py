"""CPU suite: the elevation-mapping restatement (oracle/elev_oracle.cpp, the checker of row N3) pinned to the reference's
own source: Mapping/src/elevation_mapping_periodical/elevation_mapping/cuda/gpu_process.cu compiled for the host by
oracle/Makefile (threads in gid order, stand-ins for the CUDA runtime and the few Eigen operations it uses)."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from elev_session import session  # noqa: E402


@pytest.mark.parametrize("L", [60, 61])
def test_restatement_equals_the_reference_source_run_on_the_host(oracle, L):
    if oracle.ref_lib("elev") is None:
        pytest.skip("oracle/_ref/libref_elev.so not built (no reference tree at build time)")
    want = session(oracle.RefElevMap(L, 0.2), np.random.default_rng(3), 5, L)
    got = session(oracle.ElevMap(L, 0.2), np.random.default_rng(3), 5, L)
    assert len(got) == len(want)
    worst = {}
    for (kg, g), (kw, w) in zip(got, want):
        assert kg == kw
        if kg in ("move", "frame"):
            for a, b in zip(g, w):
                np.testing.assert_array_equal(a, b)
        elif kg == "optmove":
            np.testing.assert_array_equal(g, w)
        elif kg == "points":
            for k in ("map_index", "x", "y", "z", "x_ts", "y_ts", "z_ts"):
                np.testing.assert_array_equal(g[k], w[k])
            assert (g["map_index"] >= 0).sum() > 500
            np.testing.assert_allclose(g["var"], w["var"], rtol=1e-6, atol=1e-12)
            worst["var"] = max(worst.get("var", 0), float(np.abs(g["var"] - w["var"]).max()))
        elif kg == "feature":
            for k in ("colorR", "colorG", "colorB"):
                np.testing.assert_array_equal(g[k], w[k])
            for k in ("elevation", "var", "intensity"):
                np.testing.assert_allclose(g[k], w[k], rtol=2e-6, atol=2e-6, err_msg=k)
                worst[k] = max(worst.get(k, 0), float(np.abs(g[k] - w[k]).max()))
            seen = w["elevation"] != -10      # empty cells: the reference returns before writing rough / slope / traver, its
            assert seen.sum() > 300           # output there is whatever cudaMalloc handed out (gpu_process.cu:577-578,1267-1269)
            for k in ("rough", "slope", "traver"):
                np.testing.assert_allclose(g[k][seen], w[k][seen], rtol=2e-6, atol=2e-6, err_msg=k)
                worst[k] = max(worst.get(k, 0), float(np.abs(g[k][seen] - w[k][seen]).max()))
        else:
            for i, (a, b) in enumerate(zip(g, w)):
                np.testing.assert_allclose(a, b, rtol=2e-6, atol=2e-6, err_msg=f"layer {i}")
    print("largest differences:", worst)
