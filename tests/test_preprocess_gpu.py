"""GPU parity for the pre-processing row N2 vs numpy restatements of open3d voxel_down_sample
(main_RING.py:257-259) and load_pc_infer (util.py:91-112)."""
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


def _np_voxel_down_sample(p, vs):
    p = np.asarray(p, np.float64)
    idx = np.floor((p - (p.min(0) - vs * 0.5)) / vs).astype(np.int64)
    order = np.lexsort((idx[:, 2], idx[:, 1], idx[:, 0]))
    idx, ps = idx[order], p[order]
    head = np.ones(len(p), bool)
    head[1:] = (idx[1:] != idx[:-1]).any(1)
    seg = np.cumsum(head) - 1
    out = np.zeros((seg[-1] + 1, 3))
    np.add.at(out, seg, ps)
    return out / np.bincount(seg)[:, None]


def _np_load_pc_infer(pc):
    pc = np.array(pc, dtype=np.float32)
    keep = (np.abs(pc[:, 0]) < 70.) & (np.abs(pc[:, 1]) < 70.) & (pc[:, 2] < 30.) & (pc[:, 2] > 0.)
    h = pc[keep][:, :3].copy()
    h[:, 0] = h[:, 0] / 70.
    h[:, 1] = h[:, 1] / 70.
    h[:, 2] = h[:, 2] / 30.
    return h


@pytest.mark.parametrize("dtype", ["float32", "float64"])
def test_voxel_down_sample(dev, dtype):
    import torch
    from mr_slam_amd import preprocess, synth
    pts = synth.lidar_scan(70, 60000, metric=True).astype(dtype)
    got = preprocess.voxel_down_sample(torch.from_numpy(pts).to(dev), 0.2).cpu().numpy()
    want = _np_voxel_down_sample(pts, 0.2)
    assert got.shape == want.shape and 2000 < got.shape[0] < pts.shape[0]
    np.testing.assert_allclose(got, want, rtol=0, atol=1e-12)


@pytest.mark.parametrize("dtype", ["float32", "float64"])
def test_voxel_down_sample_batch_is_the_per_scan_filter(dev, dtype):
    """mrs_voxel_downsample_batch (hash grid, fixed-point sums, first-occurrence order) == the numpy restatement of open3d's filter per scan:
    the same voxels, centroids within 1e-12 m; an empty scan inside the batch; run to run the same bits in the same order."""
    import torch
    from mr_slam_amd import preprocess, synth
    rng = np.random.default_rng(11)
    scans = [synth.lidar_scan(60, 40000, metric=True), np.zeros((0, 3), np.float32), synth.lidar_scan(61, 25000, metric=True) * 1.7,
             rng.uniform(-3, 3, (5000, 3)).astype(np.float32), np.repeat(rng.normal(size=(7, 3)), 300, axis=0).astype(np.float32)]
    pts = np.concatenate([np.concatenate([s, np.full((s.shape[0], 1), 0.5, s.dtype)], 1) for s in scans]).astype(dtype)   # [N, 4]: x, y, z, intensity
    offs = np.concatenate([[0], np.cumsum([s.shape[0] for s in scans])]).astype(np.int64)
    t = torch.from_numpy(pts).to(dev)
    out, o = preprocess.voxel_down_sample_batch(t, offs, 0.2)
    out2, o2 = preprocess.voxel_down_sample_batch(t, offs, 0.2)
    assert torch.equal(out, out2) and torch.equal(o, o2)
    o = o.cpu().numpy()
    assert o[0] == 0 and (np.diff(o) >= 0).all()
    out = out.cpu().numpy()
    for b, sc in enumerate(scans):
        got = out[o[b]:o[b + 1]]
        if sc.shape[0] == 0:
            assert got.shape[0] == 0
            continue
        want = _np_voxel_down_sample(pts[offs[b]:offs[b + 1], :3], 0.2)
        assert got.shape == want.shape, (b, got.shape, want.shape)
        gs = got[np.lexsort((got[:, 2], got[:, 1], got[:, 0]))]
        ws = want[np.lexsort((want[:, 2], want[:, 1], want[:, 0]))]
        np.testing.assert_allclose(gs, ws, rtol=0, atol=1e-12)
        # first-occurrence order: the first output voxel holds the scan's first point
        p0 = pts[offs[b], :3].astype(np.float64)
        assert np.abs(got[0] - p0).max() <= 0.2 + 1e-9
    single = preprocess.voxel_down_sample(t[offs[0]:offs[1]], 0.2).cpu().numpy()            # the sort-based single-scan call: same set
    g0 = out[o[0]:o[1]]
    np.testing.assert_allclose(g0[np.lexsort((g0[:, 2], g0[:, 1], g0[:, 0]))], single[np.lexsort((single[:, 2], single[:, 1], single[:, 0]))], rtol=0, atol=1e-12)


def test_voxel_down_sample_batch_crowded_voxel(dev):
    """200 000 copies of one point (0.9 of a voxel from the voxel's lower corner in every axis) + a corner marker: the fixed-point sums of the
    hash grid must not leave 63 bits (at the 2^46 scale of scans up to 2^17 points they would)"""
    import torch
    from mr_slam_amd import preprocess
    p = np.array([0.18, 0.18, 0.18])
    pts = np.concatenate([np.full((1, 3), -0.1), np.tile(p, (200000, 1)), np.array([[3.0, 3.0, 3.0]])]).astype(np.float64)
    out, o = preprocess.voxel_down_sample_batch(torch.from_numpy(pts).to(dev), np.array([0, pts.shape[0]], np.int64), 0.2)
    out = out.cpu().numpy()
    want = _np_voxel_down_sample(pts, 0.2)
    assert out.shape == want.shape == (3, 3)
    np.testing.assert_allclose(out[np.lexsort((out[:, 2], out[:, 1], out[:, 0]))], want[np.lexsort((want[:, 2], want[:, 1], want[:, 0]))], rtol=0, atol=1e-11)


def test_load_pc_infer_batch_feeds_bev(dev, oracle):
    import torch
    from mr_slam_amd import bev, preprocess, synth
    rng = np.random.default_rng(5)
    raws = []
    for i, n in enumerate((30000, 0, 12345)):
        p = synth.lidar_scan(80 + i, max(n, 1), metric=True)[:n].astype(np.float64)
        p[:, 2] -= rng.uniform(0, 0.5)                    # push some points below z = 0
        raws.append(np.concatenate([p, rng.uniform(size=(n, 1))], 1))   # x, y, z, intensity
    offs = np.cumsum([0] + [r.shape[0] for r in raws]).astype(np.int64)
    pts = torch.from_numpy(np.concatenate(raws)).to(dev)
    xyz, doffs = preprocess.load_pc_infer_batch(pts, offs)
    h_offs = doffs.cpu().numpy()
    img = bev.cart_bev(xyz, doffs, 1, 1, 120, 120, 1).cpu().numpy()
    for b, r in enumerate(raws):
        want = _np_load_pc_infer(r)
        assert h_offs[b + 1] - h_offs[b] == want.shape[0]
        nb = want.shape[0]
        got = xyz[3 * h_offs[b]: 3 * h_offs[b] + 3 * nb].cpu().numpy().reshape(3, nb).T
        np.testing.assert_array_equal(got, want)
        soa = np.ascontiguousarray(want.T).reshape(-1)
        np.testing.assert_array_equal(img[b].reshape(-1), oracle.bev_cart(soa, 1, 1, 120, 120, 1).reshape(-1, 3)[:, 2])


def test_wire_format():
    from mr_slam_amd import preprocess as P
    id0, id1 = P.loop_ids(0, 41, 2, 7)
    assert id0 == (97 << 56) + 42 and id1 == (99 << 56) + 8
    assert chr(id0 >> 56) == "a" and (id0 & ((1 << 56) - 1)) == 42
    assert P.loopinfo_line(0, 41, 2, 7, (1.0, 2.0, 3.0), (0.0, 0.0, 0.0, 1.0)) == "0 41 2 7 1.0 2.0 3.0 0.0 0.0 0.0 1.0"


def test_approximate_voxel_grid_is_the_sequential_pcl_filter(oracle):
    """Row G1: pygicp.downsample = pcl::ApproximateVoxelGrid.  The GPU form (stable sort by hash bucket, runs, flush-time
    sort) reproduces the sequential 512-entry history filter bit for bit -- values AND output order -- on lidar-ordered,
    shuffled, tiny and heavily colliding inputs, float32 and float64 sources."""
    import torch
    from mr_slam_amd import preprocess, synth
    from mr_slam_amd.compat import pygicp
    rng = np.random.default_rng(0)
    lidar = synth.lidar_scan(3, 120000, metric=True)
    cases = [(lidar, 0.2), (lidar[rng.permutation(lidar.shape[0])], 0.2), (lidar[:5], 0.2), (lidar[:1], 0.5),
             (rng.normal(0, 30, (50000, 3)).astype(np.float32), 1.0),          # > 512 live voxels: constant evictions
             (np.repeat(lidar[:300], 7, 0), 0.05), (rng.uniform(-0.01, 0.01, (4000, 3)).astype(np.float32), 0.2)]
    for pts, leaf in cases:
        want = oracle.approx_voxel_grid(pts, leaf)
        for dt in (torch.float32, torch.float64):
            got = preprocess.approx_voxel_grid(torch.from_numpy(np.ascontiguousarray(pts)).to(dt).cuda(), leaf).cpu().numpy()
            assert got.shape == want.shape
            np.testing.assert_array_equal(got, want)
    got = pygicp.downsample(lidar.astype(np.float64), 0.2)                   # the drop-in's default is upstream's filter
    np.testing.assert_array_equal(got, oracle.approx_voxel_grid(lidar, 0.2))
    exact = pygicp.downsample(lidar.astype(np.float64), 0.2, approximate=False)
    assert exact.shape[0] == _np_voxel_down_sample(lidar, 0.2).shape[0] and 0.6 * got.shape[0] < exact.shape[0] < 1.1 * got.shape[0]
