"""GPU parity for the elevation-mapping row N3: a multi-frame session (move, process, fuse, variance update,
features, ray tracing, loop-closure shifts) replayed on the HIP library and on the sequential restatement."""
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


def _frame_points(rng, n, pose_xy):
    """A terrain-like cloud in the sensor frame: points behind the robot (y < -1) survive the reference's filter."""
    x = rng.uniform(-6, 6, n).astype(np.float32)
    y = rng.uniform(-7, 2, n).astype(np.float32)
    z = (0.15 * np.sin(0.8 * (x + pose_xy[0])) + 0.1 * np.cos(1.1 * (y + pose_xy[1])) - 0.6 + rng.normal(0, 0.02, n)).astype(np.float32)
    bump = (np.abs(x - 2) < 0.4) & (np.abs(y + 4) < 0.4)
    z[bump] += 0.8
    return x, y, z


def _session(m, rng, frames, L):
    out = []
    pose = np.array([0.0, 0.0, 0.9], np.float32)
    for k in range(frames):
        pose[:2] += rng.uniform(-0.5, 0.7, 2).astype(np.float32)
        out.append(("move", m.move(pose)))
        x, y, z = _frame_points(rng, 6000, pose)
        yaw = 0.1 * k
        T = np.eye(4, dtype=np.float32)
        T[:2, :2] = [[np.cos(yaw), -np.sin(yaw)], [np.sin(yaw), np.cos(yaw)]]
        T[:3, 3] = [pose[0], pose[1], 0.9]
        rv = np.diag([1e-4, 1e-4, 4e-4]).astype(np.float32)
        res = m.process_points(x, y, z, T, -2.0, 3.0, 0.02, 0.003, 0.01, [0.0, 0.0, 1.0], rv, np.eye(3), [0.0, 0.0, 1.0],
                               [[0, -0.2, 0.1], [0.2, 0, -0.05], [-0.1, 0.05, 0]])
        out.append(("points", res))
        n = x.size
        cr = rng.integers(0, 256, n); cg = rng.integers(0, 256, n); cb = rng.integers(0, 256, n)
        inten = rng.uniform(0, 1, n).astype(np.float32)
        m.fuse(res["map_index"], cr, cg, cb, inten, res["z_ts"], res["var"])
        m.mapvar_update(1e-4)
        out.append(("feature", m.map_feature()))
        m.raytracing()
        out.append(("layers", [m.layer(w) for w in range(5)]))
        if k == 2:
            out.append(("optmove", m.map_optmove(pose[:2] + 0.33, 0.05)))
        if k == 3:
            m.map_closeloop(pose[:2] - 0.41, -0.02)
            out.append(("frame", m.frame()))
    return out


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
