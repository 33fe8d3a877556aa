"""CPU suite for oracle/voxel_oracle.c (row G1, pcl::ApproximateVoxelGrid restated; parity unpinned: PCL is not in the
reference tree).  Checked against an independent numpy formulation of the same filter (its closed form: runs of equal
voxels inside one hash bucket, flushed at the next run's first point) and against the properties the filter has by
construction."""
import numpy as np


def _closed_form(p, leaf):
    p = np.asarray(p, np.float32)
    inv = np.float32(1) / np.float32(leaf)
    v = np.floor(p * inv).astype(np.int64)
    h = (v[:, 0] * 7171 + v[:, 1] * 3079 + v[:, 2] * 4231) & 511
    order = np.argsort(h, kind="stable")
    sh, sv = h[order], v[order]
    head = np.ones(len(p), bool)
    head[1:] = (sh[1:] != sh[:-1]) | (sv[1:] != sv[:-1]).any(1)
    starts = np.flatnonzero(head)
    ends = np.r_[starts[1:], len(p)]
    cent, keys = [], []
    for s, e in zip(starts, ends):
        acc = np.zeros(3, np.float32)
        for j in order[s:e]:
            acc = acc + p[j]
        cent.append(acc / np.float32(e - s))
        keys.append(order[e] if e < len(p) and sh[e] == sh[s] else len(p) + sh[s])
    return np.asarray(cent)[np.argsort(keys)].astype(np.float64)


def test_approx_voxel_grid_matches_closed_form_and_properties(oracle):
    from mr_slam_amd import synth
    rng = np.random.default_rng(1)
    lidar = synth.lidar_scan(4, 20000, metric=True)
    for pts, leaf in ((lidar, 0.2), (lidar[rng.permutation(20000)], 0.2), (rng.normal(0, 20, (8000, 3)).astype(np.float32), 1.0),
                      (lidar[:3], 0.2)):
        out = oracle.approx_voxel_grid(pts, leaf)
        np.testing.assert_array_equal(out, _closed_form(pts, leaf))
        n_vox = np.unique(np.floor(pts.astype(np.float32) * (np.float32(1) / np.float32(leaf))), axis=0).shape[0]
        assert n_vox <= out.shape[0] <= pts.shape[0]          # an evicted voxel met again is emitted again: >= exact count
        # every output point is the mean of points of ONE voxel, so it lies inside (the closure of) a voxel that holds input
        vin = {tuple(v) for v in np.floor(pts.astype(np.float32) * (np.float32(1) / np.float32(leaf))).astype(np.int64)}
        vout = np.floor(out / leaf + 1e-4).astype(np.int64), np.floor(out / leaf - 1e-4).astype(np.int64)
        assert all(tuple(a) in vin or tuple(b) in vin for a, b in zip(*vout))
    # fewer than 512 distinct voxels and no hash collision between them: the filter is exact
    grid = np.stack(np.meshgrid(np.arange(6), np.arange(6), np.arange(6), indexing="ij"), -1).reshape(-1, 3).astype(np.float32)
    h = (grid[:, 0].astype(int) * 7171 + grid[:, 1].astype(int) * 3079 + grid[:, 2].astype(int) * 4231) & 511
    keep = grid[np.unique(h, return_index=True)[1]]
    pts = (np.repeat(keep, 5, 0) + rng.uniform(0.1, 0.9, (keep.shape[0] * 5, 3))).astype(np.float32)
    out = oracle.approx_voxel_grid(pts[rng.permutation(pts.shape[0])], 1.0)
    assert out.shape[0] == keep.shape[0]
