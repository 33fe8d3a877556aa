"""BASELINE configs[0] on the GPU: the reference's one real scan (NCLT, disco_ros/test.bin, committed as
tests/golden/nclt_scan.npz with the reference CPU rasteriser's output) through every stage of the path."""
import os

import numpy as np
import pytest
from scipy.spatial.transform import Rotation as Rot

from test_oracle_bev import NCLT_LAYOUTS, nclt_bytes

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    import torch
    assert torch.cuda.is_available()
    from mr_slam_amd import _lib
    _lib.load()
    return "cuda:0"


@pytest.fixture(scope="module")
def scan(golden_dir):
    g = np.load(os.path.join(golden_dir, "nclt_scan.npz"))
    from mr_slam_amd import preprocess
    hits = preprocess.load_lidar_file_nclt(nclt_bytes(g))
    xyz, _, _ = preprocess.decode_nclt(nclt_bytes(g))
    return g, hits, xyz


def test_polar_bev_bit_exact_vs_reference_output(dev, scan):
    import torch
    from mr_slam_amd import bev
    from mr_slam_amd.compat import gputransform
    g, hits, _ = scan
    soa = hits.transpose().flatten().astype(np.float32)
    t = torch.from_numpy(soa).to(dev)
    offs = torch.tensor([0, hits.shape[0]], dtype=torch.int64, device=dev)
    for (R, S, H) in NCLT_LAYOUTS:
        tag = f"{R}x{S}x{H}"
        r, s, h = (x.cpu().numpy() for x in bev.polar_indices(t, 1, 1, R, S, H))
        np.testing.assert_array_equal(r, g[f"ring_{tag}"])
        np.testing.assert_array_equal(s, g[f"sector_{tag}"])
        np.testing.assert_array_equal(h, g[f"height_{tag}"])
        occ = bev.polar_bev(t, offs, 1, 1, R, S, H).cpu().numpy().reshape(-1)
        np.testing.assert_array_equal(np.flatnonzero(occ).astype(np.int32), g[f"occupied_{tag}"])
        # the drop-in module, called like load_pc_file_infer does (loading_pointclouds.py:78-83)
        tr = gputransform.GPUTransformer(soa, hits.shape[0], 1, 1, R, S, H, 1)
        tr.transform()
        out = tr.retreive().reshape(-1, 3)[..., 2]
        np.testing.assert_array_equal(np.flatnonzero(out).astype(np.int32), g[f"occupied_{tag}"])


def test_ring_yaw_recovered_on_real_scan(dev, scan, oracle):
    """RING descriptor of the scan vs the same scan yawed by 33 degrees = 11 angle bins of 3 degrees... the
    sinogram grid has 119 steps per turn (util.py:191 linspace includes both ends), so expect |shift| in {10, 11}."""
    import torch
    from mr_slam_amd import ring, bev
    _, hits, _ = scan
    yaw = np.deg2rad(33.0)
    Rz = np.array([[np.cos(yaw), -np.sin(yaw), 0], [np.sin(yaw), np.cos(yaw), 0], [0, 0, 1]])
    clouds = [hits.astype(np.float32), (hits @ Rz.T).astype(np.float32)]
    xyz, offs = bev.pack_scans(clouds, dev)
    img = bev.cart_bev(xyz, offs, 1, 1, 120, 120, 1)
    want = np.stack([oracle.bev_cart(np.ascontiguousarray(c.T).reshape(-1), 1, 1, 120, 120, 1).reshape(-1, 3)[:, 2].reshape(120, 120)
                     for c in clouds])
    np.testing.assert_array_equal(img.cpu().numpy().reshape(2, 120, 120), want)
    plan = ring.ring_plan(0)
    _, sino = plan.forward(img.reshape(2, 120, 120).contiguous(), raw=False, normalized=True)
    ang = np.linspace(0, 2 * np.pi, 120).astype(np.float32)
    from oracle import corr_oracle as K
    want_s = np.stack([K.ring_normalize(torch.from_numpy(oracle.radon_parallel(w, ang, 120, 1.0))[None]).numpy()[0] for w in want])
    np.testing.assert_allclose(sino.cpu().numpy().reshape(2, 120, 120), want_s, rtol=0, atol=2e-5)
    dist, shift = ring.corr_pairs(sino[:1].reshape(1, 1, 120, 120), sino[1:].reshape(1, 1, 120, 120))
    assert abs(int(shift[0])) in (10, 11)
    assert float(dist[0]) < 0.6


def test_gicp_on_real_scan_within_north_star_tolerance(dev, scan, oracle):
    """the scan vs a copy moved by yaw 30 deg, t = (2, 1, 0), sigma = 2 cm; the initial
    guess plays the RING estimate (main_RING.py:96-103).  HIP vs restatement within 1e-4 m / 1e-4 rad."""
    from mr_slam_amd import gicp
    _, _, xyz = scan
    keep = (np.abs(xyz[:, 0]) < 70) & (np.abs(xyz[:, 1]) < 70) & ~((np.abs(xyz[:, 0]) < 2) & (np.abs(xyz[:, 1]) < 2))
    src = xyz[keep].astype(np.float32)
    rng = np.random.default_rng(7)
    R = Rot.from_euler("z", 30.0, degrees=True).as_matrix()
    t = np.array([2.0, 1.0, 0.0])
    tgt = (src.astype(np.float64) @ R.T + t + rng.normal(0, 0.02, src.shape)).astype(np.float32)
    T0 = np.eye(4)
    T0[:3, :3] = Rot.from_euler("z", 28.0, degrees=True).as_matrix()
    T0[:3, 3] = [1.7, 1.2, 0.05]
    b = gicp.GicpBatch(1)
    b.set_params(k_correspondences=15, max_correspondence_distance=5.0, max_iterations=64)
    b.set_sources([src]); b.set_targets([tgt])
    T, conv, iters = b.align(T0[None])
    g = oracle.Gicp(k=15, max_corr=5.0, max_iter=64)
    g.set_source(src); g.set_target(tgt)
    Tw, cw, _, _ = g.align(T0)
    assert conv[0] and cw
    dt = np.linalg.norm(T[0][:3, 3] - Tw[:3, 3])
    dr = np.linalg.norm(Rot.from_matrix(T[0][:3, :3] @ Tw[:3, :3].T).as_rotvec())
    assert dt < 1e-4 and dr < 1e-4, (dt, dr)
    Ttrue = np.eye(4); Ttrue[:3, :3] = R; Ttrue[:3, 3] = t
    assert np.linalg.norm(T[0][:3, 3] - t) < 0.05
    assert np.linalg.norm(Rot.from_matrix(T[0][:3, :3] @ R.T).as_rotvec()) < 2e-3
