"""GPU parity for the elevation-mapping row N3: a multi-frame session (move, process, fuse, variance update,
features, ray tracing, loop-closure shifts) replayed on the HIP library, on the sequential restatement and on the
reference's own gpu_process.cu built for the host (oracle/_ref/libref_elev.so; tests/test_oracle_elev.py pins the
restatement to it on the CPU)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    import torch
    assert torch.cuda.is_available()
    from mr_slam_amd import _lib
    _lib.load()
    return "cuda:0"


from elev_session import session as _session  # noqa: E402


@pytest.mark.parametrize("L", [60, 61])
def test_session_matches_sequential_restatement(dev, oracle, L):
    from mr_slam_amd import elevation
    got = _session(elevation.ElevationMap(L, 0.2), np.random.default_rng(3), 5, L)
    want = _session(oracle.ElevMap(L, 0.2), np.random.default_rng(3), 5, L)
    assert len(got) == len(want)
    seen_cells = 0
    for (kg, g), (kw, w) in zip(got, want):
        assert kg == kw
        if kg in ("move", "frame"):
            for a, b in zip(g, w):
                np.testing.assert_allclose(a, b, rtol=0, atol=1e-6)
        elif kg == "optmove":
            np.testing.assert_allclose(g, w, atol=1e-6)
        elif kg == "points":
            np.testing.assert_array_equal(g["map_index"], w["map_index"])
            assert (g["map_index"] >= 0).sum() > 500
            for k in ("x", "y", "z", "x_ts", "y_ts", "z_ts"):
                np.testing.assert_array_equal(g[k], w[k])
            np.testing.assert_allclose(g["var"], w["var"], rtol=1e-6, atol=1e-12)
        elif kg == "feature":
            for k in ("elevation", "var", "intensity"):
                np.testing.assert_allclose(g[k], w[k], rtol=2e-6, atol=1e-7)
            for k in ("colorR", "colorG", "colorB"):
                np.testing.assert_array_equal(g[k], w[k])
            np.testing.assert_allclose(g["rough"], w["rough"], rtol=1e-4, atol=1e-5)
            ok = np.abs(g["slope"] - w["slope"]) < 2e-3        # float Jacobi with libm vs device sin/cos
            assert ok.mean() > 0.995
            np.testing.assert_allclose(g["traver"][ok], w["traver"][ok], rtol=0, atol=5e-3)
            seen_cells = max(seen_cells, int((g["elevation"] != -10).sum()))
        else:
            for a, b in zip(g, w):
                same_empty = (a == -10) == (b == -10)
                assert same_empty.mean() > 0.999
                keep = same_empty & (a != -10)
                np.testing.assert_allclose(a[keep], b[keep], rtol=2e-6, atol=5e-3)
    assert seen_cells > 300


def test_session_matches_the_reference_source_built_for_the_host(dev, oracle):
    """Same session, HIP library vs the reference's gpu_process.cu itself (host build, threads in gid order)."""
    if oracle.ref_lib("elev") is None:
        pytest.skip("oracle/_ref/libref_elev.so not built")
    from mr_slam_amd import elevation
    L = 60
    got = _session(elevation.ElevationMap(L, 0.2), np.random.default_rng(3), 5, L)
    want = _session(oracle.RefElevMap(L, 0.2), np.random.default_rng(3), 5, L)
    for (kg, g), (kw, w) in zip(got, want):
        assert kg == kw
        if kg == "points":
            for k in ("map_index", "x", "y", "z", "x_ts", "y_ts", "z_ts"):
                np.testing.assert_array_equal(g[k], w[k])
            np.testing.assert_allclose(g["var"], w["var"], rtol=1e-6, atol=1e-12)
        elif kg == "feature":
            seen = w["elevation"] != -10                       # elsewhere the reference's rough / slope / traver are uninitialised
            np.testing.assert_array_equal(g["elevation"] != -10, seen)
            for k in ("elevation", "var", "intensity"):
                np.testing.assert_allclose(g[k], w[k], rtol=2e-6, atol=1e-7)
            for k in ("colorR", "colorG", "colorB"):
                np.testing.assert_array_equal(g[k], w[k])
            np.testing.assert_allclose(g["rough"][seen], w["rough"][seen], rtol=1e-4, atol=1e-5)
            ok = np.abs(g["slope"] - w["slope"])[seen] < 2e-3
            assert ok.mean() > 0.995
        elif kg in ("move", "frame"):
            for a, b in zip(g, w):
                np.testing.assert_allclose(a, b, rtol=0, atol=1e-6)
