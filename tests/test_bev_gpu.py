"""GPU parity suite for the BEV rasterisers: HIP (through the C ABI) vs the CPU oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from mr_slam_amd import _lib
    _lib.load()  # fail loudly if the HIP library is missing
    return "cuda:0"


def _float_neighbours(v):
    v = np.asarray(v, np.float32)
    return np.concatenate([np.nextafter(v, np.float32(-np.inf)), v, np.nextafter(v, np.float32(np.inf))])


def _adversarial_cloud(rng, n=60000):
    """Points sitting exactly on / one ulp around bin edges of every axis, zeros, axis-aligned
    points, out-of-range and clamped coordinates."""
    xyz = rng.uniform(-1, 1, size=(3, n)).astype(np.float32)
    xyz[2] *= 0.999
    k = 0
    edges120 = _float_neighbours(np.arange(0, 121) * np.float32(2.0 / 120) - 1)      # cartesian edges
    m = edges120.size
    xyz[0, k:k + m] = edges120; k += m
    xyz[1, k:k + m] = edges120; k += m
    edges_h = _float_neighbours(np.arange(0, 21) * np.float32(0.1) - 1)
    m = edges_h.size
    xyz[2, k:k + m] = np.clip(edges_h, -0.99999, 0.99999); k += m
    # sector edges: angles at multiples of 3 degrees, a few radii
    for rad in (0.05, 0.3, 0.77):
        ang = np.deg2rad(np.arange(0, 360, 3, dtype=np.float64))
        for dx in (-1, 0, 1):
            x = (rad * np.cos(ang)).astype(np.float32)
            y = (rad * np.sin(ang)).astype(np.float32)
            x = np.nextafter(x, np.float32(np.inf)) if dx > 0 else (np.nextafter(x, np.float32(-np.inf)) if dx < 0 else x)
            m = x.size
            xyz[0, k:k + m] = x; xyz[1, k:k + m] = y; k += m
    # ring edges: radii at multiples of 1/40 and 1/120 along random directions
    for R in (40, 120):
        rr = _float_neighbours(np.arange(1, R + 1, dtype=np.float32) / np.float32(R))
        th = rng.uniform(0, 2 * np.pi, size=rr.size)
        m = rr.size
        xyz[0, k:k + m] = (rr * np.cos(th)).astype(np.float32)
        xyz[1, k:k + m] = (rr * np.sin(th)).astype(np.float32); k += m
    # exact zeros, axes, diagonals, tiny and denormal values
    special = np.array([0.0, -0.0, 1e-4, -1e-4, 1e-38, -1e-38, 1e-45, 0.5, -0.5, 0.9999, -0.9999], np.float32)
    for a in special:
        for b in special:
            xyz[0, k] = a; xyz[1, k] = b; xyz[2, k] = rng.uniform(-0.9, 0.9); k += 1
    assert k < n
    return xyz


def _cmp_polar_indices(dev, oracle, soa, R, S, H, ml=1, mh=1):
    import torch
    from mr_slam_amd import bev
    t = torch.from_numpy(soa).to(dev)
    r, s, h = (x.cpu().numpy() for x in bev.polar_indices(t, ml, mh, R, S, H))
    er, es, eh, _ = oracle.bev_polar_indices(soa, ml, mh, R, S, H)
    np.testing.assert_array_equal(r, er)
    np.testing.assert_array_equal(s, es)
    np.testing.assert_array_equal(h, eh)


@pytest.mark.parametrize("layout", [(40, 120, 20), (120, 120, 1), (7, 33, 3)])
def test_polar_indices_bit_exact(dev, oracle, layout):
    rng = np.random.default_rng(11)
    R, S, H = layout
    _cmp_polar_indices(dev, oracle, _adversarial_cloud(rng).reshape(-1), R, S, H)
    # unnormalised metric cloud, larger extents (ring clamp, heights out of range -> aliasing/drop)
    big = (rng.normal(0, 1, size=(3, 40000)) * np.array([[30.0], [30.0], [3.0]])).astype(np.float32)
    _cmp_polar_indices(dev, oracle, big.reshape(-1), R, S, H, ml=50, mh=5)


def test_polar_indices_golden(dev, oracle, golden_dir):
    import os
    import torch
    from mr_slam_amd import bev
    for name in ("1", "2"):
        g = np.load(os.path.join(golden_dir, f"bev_polar_{name}.npz"))
        t = torch.from_numpy(g["xyz_soa"]).to(dev)
        for (R, S, H) in ((40, 120, 20), (120, 120, 1)):
            tag = f"{R}x{S}x{H}"
            r, s, h = (x.cpu().numpy() for x in bev.polar_indices(t, 1, 1, R, S, H))
            np.testing.assert_array_equal(r, g[f"ring_{tag}"])
            np.testing.assert_array_equal(s, g[f"sector_{tag}"])
            np.testing.assert_array_equal(h, g[f"height_{tag}"])
            offs = torch.tensor([0, t.numel() // 3], dtype=torch.int64, device=dev)
            occ = bev.polar_bev(t, offs, 1, 1, R, S, H).cpu().numpy().reshape(-1)
            np.testing.assert_array_equal(np.flatnonzero(occ).astype(np.int32), g[f"occupied_{tag}"])


def test_polar_special_values_dropped_like_oracle(dev, oracle):
    v = np.array([np.nan, np.inf, -np.inf, 1e30, -1e30, 3e9, 1e-45, 0.3], np.float32)
    xs, ys, zs = np.meshgrid(v, v, v, indexing="ij")
    soa = np.concatenate([xs.reshape(-1), ys.reshape(-1), zs.reshape(-1)])
    _cmp_polar_indices(dev, oracle, soa, 40, 120, 20)


@pytest.mark.parametrize("layout", [(120, 120, 1), (120, 120, 5), (64, 96, 2), (33, 17, 1)])
def test_cart_indices_bit_exact(dev, oracle, layout):
    import torch
    from mr_slam_amd import bev
    rng = np.random.default_rng(12)
    NX, NY, H = layout
    xyz = _adversarial_cloud(rng)
    xyz[:, -5000:] *= 1.3  # exercise the +-0.9999 clamps
    ex = _float_neighbours(np.arange(0, NX + 1) * np.float32(2.0 / NX) - 1)
    xyz[0, 20000:20000 + ex.size] = ex
    soa = xyz.reshape(-1)
    t = torch.from_numpy(soa).to(dev)
    a, b, c = (x.cpu().numpy() for x in bev.cart_indices(t, 1, 1, NX, NY, H))
    ea, eb, ec, _ = oracle.bev_cart_indices(soa, 1, 1, NX, NY, H)
    np.testing.assert_array_equal(a, ea)
    np.testing.assert_array_equal(b, eb)
    np.testing.assert_array_equal(c, ec)


def _scans(rng, sizes):
    from mr_slam_amd import synth
    out = []
    for i, n in enumerate(sizes):
        if n == 0:
            out.append(np.zeros((0, 3), np.float32))
        elif i % 2:
            out.append(synth.uniform_scan(100 + i, n))
        else:
            out.append(synth.lidar_scan(100 + i, n))
    return out


def test_batched_compact_matches_oracle_ragged(dev, oracle):
    """Ragged batch incl. an empty scan and sizes that break 16-byte plane alignment."""
    from mr_slam_amd import bev, synth
    rng = np.random.default_rng(1)
    scans = _scans(rng, [4096, 0, 12345, 20000, 1, 7777])
    xyz, offs = bev.pack_scans(scans, dev)
    cart = bev.cart_bev(xyz, offs, 1, 1, 120, 120, 1).cpu().numpy()
    pol = bev.polar_bev(xyz, offs, 1, 1, 40, 120, 20).cpu().numpy()
    pol1 = bev.polar_bev(xyz, offs, 1, 1, 120, 120, 1).cpu().numpy()
    for b, s in enumerate(scans):
        soa = synth.to_soa(s)
        np.testing.assert_array_equal(cart[b].reshape(-1), oracle.bev_cart(soa, 1, 1, 120, 120, 1).reshape(-1, 3)[:, 2])
        np.testing.assert_array_equal(pol[b].reshape(-1), oracle.bev_polar(soa, 1, 1, 40, 120, 20).reshape(-1, 3)[:, 2])
        np.testing.assert_array_equal(pol1[b].reshape(-1), oracle.bev_polar(soa, 1, 1, 120, 120, 1).reshape(-1, 3)[:, 2])


@pytest.mark.parametrize("cfg", [(40, 120, 20, 1), (40, 120, 4, 3), (120, 120, 1, 2)])
def test_polar_reference_layout_bit_exact(dev, oracle, cfg):
    from mr_slam_amd import bev, synth
    from mr_slam_amd._lib import OUT_REFERENCE
    R, S, H, K = cfg
    scans = _scans(np.random.default_rng(2), [30000, 9999])
    xyz, offs = bev.pack_scans(scans, dev)
    got = bev.polar_bev(xyz, offs, 1, 1, R, S, H, enough_large=K, layout=OUT_REFERENCE).cpu().numpy()
    for b, s in enumerate(scans):
        np.testing.assert_array_equal(got[b], oracle.bev_polar(synth.to_soa(s), 1, 1, R, S, H, K))


@pytest.mark.parametrize("cfg", [(120, 120, 1), (120, 120, 4), (50, 70, 3)])
def test_cart_reference_layout_bit_exact(dev, oracle, cfg):
    """Order-dependent semantics of manager.cu:64-72, incl. num_height > 1 (one running max
    per column while values land in the point's own layer)."""
    from mr_slam_amd import bev, synth
    from mr_slam_amd._lib import OUT_REFERENCE
    NX, NY, H = cfg
    scans = _scans(np.random.default_rng(3), [30000, 20001])
    scans[1][:, 2] = np.random.default_rng(9).uniform(-1, 1, size=scans[1].shape[0]).astype(np.float32)
    xyz, offs = bev.pack_scans(scans, dev)
    got = bev.cart_bev(xyz, offs, 1, 1, NX, NY, H, layout=OUT_REFERENCE).cpu().numpy()
    comp = bev.cart_bev(xyz, offs, 1, 1, NX, NY, H).cpu().numpy()
    for b, s in enumerate(scans):
        want = oracle.bev_cart(synth.to_soa(s), 1, 1, NX, NY, H)
        np.testing.assert_array_equal(got[b], want)
        np.testing.assert_array_equal(comp[b].reshape(-1), want.reshape(-1, 3)[:, 2])


def test_feat_bev_matches_oracle(dev, oracle):
    import torch
    from mr_slam_amd import bev
    from mr_slam_amd._lib import OUT_REFERENCE
    rng = np.random.default_rng(4)
    F = 9
    sizes = [15000, 8191]
    planes = [rng.uniform(-1, 1, size=(F, n)).astype(np.float32) for n in sizes]
    pts, offs = bev.pack_scans([p.reshape(-1) for p in planes], dev, planes=F)
    ref = bev.feat_bev(pts, offs, F, 1, 1, 120, 120, 1, layout=OUT_REFERENCE).cpu().numpy()
    comp = bev.feat_bev(pts, offs, F, 1, 1, 120, 120, 1).cpu().numpy()
    for b, p in enumerate(planes):
        want = oracle.bev_feat(p.reshape(-1), F, 1, 1, 120, 120, 1)
        np.testing.assert_array_equal(ref[b], want)
        np.testing.assert_array_equal(comp[b].reshape(F - 3, -1), want.reshape(-1, F)[:, 3:].T)


def test_compat_modules_reference_calling_convention(dev, oracle):
    """The drop-in modules take host numpy buffers exactly like gputransform.pyx / wrapper.pyx."""
    from mr_slam_amd import synth
    from mr_slam_amd.compat import gputransform, voxelocc, voxelfeat
    s = synth.lidar_scan(5, 25000)
    soa = synth.to_soa(s)
    t = voxelocc.GPUTransformer(soa, s.shape[0], 1, 1, 120, 120, 1, 1)
    t.transform()
    np.testing.assert_array_equal(t.retreive(), oracle.bev_cart(soa, 1, 1, 120, 120, 1))
    g = gputransform.GPUTransformer(soa, s.shape[0], 1, 1, 40, 120, 20, 1)
    g.transform()
    np.testing.assert_array_equal(g.retreive(), oracle.bev_polar(soa, 1, 1, 40, 120, 20, 1))
    F = 9
    pts = np.concatenate([soa, np.random.default_rng(6).uniform(0, 1, size=(F - 3) * s.shape[0]).astype(np.float32)])
    f = voxelfeat.GPUTransformer(pts, s.shape[0], 1, 1, 120, 120, 1, F)
    f.transform()
    np.testing.assert_array_equal(f.retreive(), oracle.bev_feat(pts, F, 1, 1, 120, 120, 1))
    with pytest.raises(ValueError):
        voxelocc.GPUTransformer(soa.astype(np.float64), s.shape[0], 1, 1, 120, 120, 1, 1)


def test_full_size_properties(dev):
    """BASELINE size (120k points x 64 scans): size-independent properties -- idempotence,
    permutation invariance of the COMPACT maps, occupancy == (index histogram > 0)."""
    import torch
    from mr_slam_amd import bev, synth
    base = [synth.lidar_scan(s) for s in range(2)]
    rng = np.random.default_rng(8)
    scans = [base[i % 2] for i in range(64)]
    xyz, offs = bev.pack_scans(scans, dev)
    a = bev.cart_bev(xyz, offs, 1, 1, 120, 120, 1)
    b = bev.cart_bev(xyz, offs, 1, 1, 120, 120, 1)
    assert torch.equal(a, b) and torch.equal(a[0], a[2]) and torch.equal(a[1], a[63])
    perm = rng.permutation(base[0].shape[0])
    x2, o2 = bev.pack_scans([base[0][perm]], dev)
    assert torch.equal(bev.cart_bev(x2, o2, 1, 1, 120, 120, 1)[0], a[0])
    p = bev.polar_bev(xyz, offs, 1, 1, 40, 120, 20)
    assert torch.equal(bev.polar_bev(x2, o2, 1, 1, 40, 120, 20)[0], p[0])
    soa = torch.from_numpy(synth.to_soa(base[0])).to(dev)
    r, s, h = bev.polar_indices(soa, 1, 1, 40, 120, 20)
    lin = (s.long() + r.long() * 120 + h.long() * 4800)
    hist = torch.bincount(lin, minlength=96000)[:96000]
    assert torch.equal((hist > 0).float(), p[0].reshape(-1))


# ---- HIP vs the REFERENCE's own generate_bev_* code (rows A3-A5 pinned by reference-run output) ------------------
def _golden_clouds(golden_dir):
    import os
    yield "testbin", np.load(os.path.join(golden_dir, "bev_polar_1.npz"))["xyz_soa"].reshape(3, -1).T
    yield "nclt", np.load(os.path.join(golden_dir, "nclt_scan.npz"))["hits"].astype(np.float32)


def test_cart_bev_matches_reference_golden(dev, golden_dir):
    """tests/golden/bev_cart_ref.npz = output of the reference's kernel.cu + manager.cu (voxelocc) run on the host."""
    import os, zlib
    from mr_slam_amd import bev
    from mr_slam_amd._lib import OUT_REFERENCE
    g = np.load(os.path.join(golden_dir, "bev_cart_ref.npz"))
    clouds = list(_golden_clouds(golden_dir))
    xyz, offs = bev.pack_scans([c.astype(np.float32) for _, c in clouds], dev)
    for (NX, NY, H) in ((120, 120, 1), (40, 120, 20)):
        ref = bev.cart_bev(xyz, offs, 1, 1, NX, NY, H, layout=OUT_REFERENCE).cpu().numpy()
        comp = bev.cart_bev(xyz, offs, 1, 1, NX, NY, H).cpu().numpy()
        for b, (name, _) in enumerate(clouds):
            tag = f"{name}_{NX}x{NY}x{H}"
            assert zlib.crc32(ref[b].tobytes()) == int(g[f"crc_{tag}"][0])
            want = np.zeros(NX * NY * H, np.float32)
            want[g[f"occ_{tag}"]] = g[f"z_{tag}"]
            np.testing.assert_array_equal(comp[b].reshape(-1), want)


def test_cart_and_feat_bev_match_reference_build(dev, oracle):
    """oracle/_ref/libref_cart.so / libref_feat.so (built from the reference sources, travel with the snapshot) on the
    bench's own synthetic scans and on stress clouds."""
    from mr_slam_amd import bev, synth
    from mr_slam_amd._lib import OUT_REFERENCE
    if oracle.ref_lib("cart") is None or oracle.ref_lib("feat") is None:
        pytest.skip("oracle/_ref not built")
    rng = np.random.default_rng(21)
    stress = rng.uniform(-1.3, 1.3, size=(40000, 3)).astype(np.float32)
    stress[:100, 0] = 0.0; stress[50:150, 1] = 0.0; stress[100:200, 2] = 0.0
    stress[np.abs(stress) == 1.0] = 0.5
    scans = [synth.lidar_scan(40, 120000), stress, synth.uniform_scan(41, 33333)]
    xyz, offs = bev.pack_scans(scans, dev)
    for (NX, NY, H) in ((120, 120, 1), (64, 50, 3)):
        got = bev.cart_bev(xyz, offs, 1, 1, NX, NY, H, layout=OUT_REFERENCE).cpu().numpy()
        comp = bev.cart_bev(xyz, offs, 1, 1, NX, NY, H).cpu().numpy()
        for b, s in enumerate(scans):
            want = oracle.ref_bev_cart(synth.to_soa(s), 1, 1, NX, NY, H)
            np.testing.assert_array_equal(got[b], want)
            np.testing.assert_array_equal(comp[b].reshape(-1), want.reshape(-1, 3)[:, 2])
    F = 9
    planes = [np.concatenate([synth.to_soa(s).reshape(3, -1), rng.uniform(-1, 1, size=(F - 3, s.shape[0])).astype(np.float32)])
              for s in scans[1:]]
    pts, foffs = bev.pack_scans([p.reshape(-1) for p in planes], dev, planes=F)
    ref = bev.feat_bev(pts, foffs, F, 1, 1, 120, 120, 1, layout=OUT_REFERENCE).cpu().numpy()
    for b, p in enumerate(planes):
        np.testing.assert_array_equal(ref[b], oracle.ref_bev_feat(p.reshape(-1), F, 1, 1, 120, 120, 1))


def test_compact_maps_drop_special_values_like_the_oracle(dev, oracle):
    """NaN / inf / huge coordinates through the COMPACT (LDS) kernels of both rasterisers: same cells as the restatement
    (which drops what the reference handles with undefined behaviour)."""
    from mr_slam_amd import bev
    v = np.array([np.nan, np.inf, -np.inf, 1e30, -1e30, 3e9, 1e-45, 0.3, -0.7, 0.0], np.float32)
    xs, ys, zs = np.meshgrid(v, v, np.array([np.nan, np.inf, 0.5, -0.5, 0.25, 0.0], np.float32), indexing="ij")
    pts = np.stack([xs.reshape(-1), ys.reshape(-1), zs.reshape(-1)], 1)
    rng = np.random.default_rng(3)
    filler = rng.uniform(-1, 1, size=(5000, 3)).astype(np.float32)
    cloud = np.concatenate([pts, filler])[rng.permutation(pts.shape[0] + 5000)]
    xyz, offs = bev.pack_scans([cloud, cloud[:4096]], dev)      # second scan: 16-byte aligned planes (vector path)
    cart = bev.cart_bev(xyz, offs, 1, 1, 120, 120, 1).cpu().numpy()
    pol = bev.polar_bev(xyz, offs, 1, 1, 40, 120, 20).cpu().numpy()
    pol1 = bev.polar_bev(xyz, offs, 1, 1, 120, 120, 1).cpu().numpy()
    for b, c in enumerate((cloud, cloud[:4096])):
        soa = np.ascontiguousarray(c.T).reshape(-1)
        np.testing.assert_array_equal(cart[b].reshape(-1), oracle.bev_cart(soa, 1, 1, 120, 120, 1).reshape(-1, 3)[:, 2])
        np.testing.assert_array_equal(pol[b].reshape(-1), oracle.bev_polar(soa, 1, 1, 40, 120, 20).reshape(-1, 3)[:, 2])
        np.testing.assert_array_equal(pol1[b].reshape(-1), oracle.bev_polar(soa, 1, 1, 120, 120, 1).reshape(-1, 3)[:, 2])


def test_feat_bev_height_layers_follow_the_reference_order_semantics(dev, oracle):
    """num_height > 1: kernel.cu:151-158 keeps ONE running maximum per column while writing into the point's own layer, so
    the cell of a layer keeps the last point of that layer that raised its column's maximum.  (The reference's manager
    allocates a single layer for that kernel -- manager.cu:31-32 -- so its own run overflows; the restatement applies the
    kernel statement sequentially on a full-size map.)  HIP (sorted columns, one thread per column) == restatement,
    REFERENCE and COMPACT layouts, ragged batch."""
    from mr_slam_amd import bev
    from mr_slam_amd._lib import OUT_REFERENCE
    rng = np.random.default_rng(31)
    for (NX, NY, H, F) in ((40, 30, 4, 9), (120, 120, 3, 6)):
        planes = [rng.uniform(-1.1, 1.1, size=(F, n)).astype(np.float32) for n in (30000, 1, 12345)]
        for p in planes:
            p[np.abs(p) == 1.0] = 0.5
        pts, offs = bev.pack_scans([p.reshape(-1) for p in planes], dev, planes=F)
        ref = bev.feat_bev(pts, offs, F, 1, 1, NX, NY, H, layout=OUT_REFERENCE).cpu().numpy()
        comp = bev.feat_bev(pts, offs, F, 1, 1, NX, NY, H).cpu().numpy()
        for b, p in enumerate(planes):
            want = oracle.bev_feat(p.reshape(-1), F, 1, 1, NX, NY, H)
            np.testing.assert_array_equal(ref[b], want)
            np.testing.assert_array_equal(comp[b].reshape(F - 3, -1), want.reshape(-1, F)[:, 3:].T)
