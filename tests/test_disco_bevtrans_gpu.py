"""GPU parity suite for the rocFFT-backed rows: DiSCO descriptor / phase correlation (D1, D2) and the
RING++ BEV translation search (C4) vs the torch-CPU restatements (unpinned in the reference)."""
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


@pytest.mark.parametrize("H", [20, 1])
def test_disco_descriptor_matches_restatement(dev, oracle, H):
    """disco_ros/main.py:94-125 + DiSCO.py:315-334 on real rasterised scans (H = 20 nominal, H = 1 as
    the ROS node ends up running because of main.py:498)."""
    import torch
    from mr_slam_amd import bev, disco, synth
    from oracle import corr_oracle as K
    scans = [synth.lidar_scan(40 + i, 30000) for i in range(3)]
    xyz, offs = bev.pack_scans(scans, dev)
    sig, spec = disco.disco_descriptors(xyz, offs, 40, 120, H)
    occ = np.stack([oracle.bev_polar(synth.to_soa(s), 1, 1, 40, 120, H).reshape(-1, 3)[:, 2].reshape(H, 40, 120) for s in scans])
    wsig, wspec = K.disco_forward(occ)
    assert tuple(sig.shape) == (3, 1024) and tuple(spec.shape) == (3, 1, 40, 120)
    scale = np.abs(wspec.numpy()).max()
    assert np.abs(spec.cpu().numpy() - wspec.numpy()).max() < 2e-6 * scale + 1e-5
    np.testing.assert_allclose(sig.cpu().numpy(), wsig, rtol=1e-4, atol=2e-5 * scale)


def test_phase_corr_matches_restatement(dev):
    import torch
    from mr_slam_amd import disco
    from oracle import corr_oracle as K
    rng = np.random.default_rng(1)
    base = (rng.uniform(size=(1, 20, 40, 120)) > 0.93).astype(np.float32)
    shifts = [0, 7, -31, 60]
    bevs = np.concatenate([np.roll(base, k, axis=3) for k in shifts])
    _, spec = disco.disco_from_bev(torch.from_numpy(bevs).to(dev))
    a = spec[:1].expand(len(shifts), -1, -1, -1).contiguous()
    yaw, corr = disco.phase_corr(a, spec, want_corr=True)
    yaw = yaw.cpu().numpy()
    _, wspec = K.disco_forward(bevs)
    for i, k in enumerate(shifts):
        wy, wc = K.phase_corr(wspec[:1], wspec[i:i + 1])
        assert yaw[i] == wy
        np.testing.assert_allclose(corr[i].cpu().numpy(), wc[0], rtol=1e-4, atol=1e-4 * wc.max())
        assert (yaw[i] - 60 + k) % 120 == 0           # (bin - 60)/120*360 deg is the yaw (main.py:291)


def test_rotate_bev_matches_restatement(dev):
    import torch
    from mr_slam_amd import ring
    from oracle import corr_oracle as K
    rng = np.random.default_rng(2)
    img = rng.uniform(size=(6, 120, 120)).astype(np.float32)
    for ang in (0.0, np.pi / 2, 0.3, -1.1, np.pi):
        got = ring.rotate_bev(torch.from_numpy(img).to(dev), ang).cpu().numpy()
        want = K.rotate_nearest(img, ang * 180.0 / np.pi).numpy()
        # every output pixel equal, except where the source coordinate sits on a texel boundary (x.5 within float rounding)
        # and nearest-neighbour rounding may go either way: shown per mismatching pixel in float64
        bad = np.argwhere((got != want).any(0))
        for (yy, xx) in bad:
            px, py = xx - 59.5, yy - 59.5
            near = []
            for rot in (ang, -ang):                   # the source pixel of (yy, xx) under the inverse map (either handedness)
                sx = np.cos(rot) * px + np.sin(rot) * py + 59.5
                sy = -np.sin(rot) * px + np.cos(rot) * py + 59.5
                near.append(min(abs(sx - np.floor(sx) - 0.5), abs(sy - np.floor(sy) - 0.5)))
            assert min(near) < 2e-4, (ang, yy, xx, near)
        assert len(bad) < 0.002 * 120 * 120
    np.testing.assert_array_equal(ring.rotate_bev(torch.from_numpy(img).to(dev), 0.0).cpu().numpy(), img)


def test_solve_translation_bev_matches_restatement(dev):
    import torch
    from mr_slam_amd import ring
    from oracle import corr_oracle as K
    rng = np.random.default_rng(3)
    a = np.abs(rng.normal(size=(6, 120, 120))).astype(np.float32)
    a[a < 1.2] = 0
    pairs_b = [np.roll(np.roll(a, 9, axis=1), -14, axis=2), np.roll(a, 3, axis=2), a.copy()]
    A = torch.from_numpy(np.stack([a] * 3)).to(dev)
    B = torch.from_numpy(np.stack(pairs_b)).to(dev)
    y, x, neg, corr = ring.solve_translation_bev(A, B, want_corr=True)
    for i, b in enumerate(pairs_b):
        wy, wx, wneg, wcorr = K.solve_translation_bev(a, b)
        np.testing.assert_allclose(corr[i].cpu().numpy(), wcorr, rtol=2e-4, atol=2e-4 * wcorr.max())
        assert (y[i], x[i]) == (wy, wx)
        assert abs(neg[i] - wneg) < 2e-4 * abs(wneg)
    y1, x1, n1 = ring.solve_translation_bev(A[0], B[0])
    assert (y1, x1) == (y[0], x[0])


def test_mapping_side_calc_rel_ori_and_signature_search(dev, oracle):
    """Row N4: calcRelOri literal (global_manager.cpp:2719-2762) and the kd-tree's job (1-NN over signatures)."""
    import torch
    from mr_slam_amd import disco
    rng = np.random.default_rng(7)
    base = (rng.uniform(size=(1, 20, 40, 120)) > 0.93).astype(np.float32)
    bevs = np.concatenate([np.roll(base, k, axis=3) for k in (0, 9, -20)])
    sig, spec = disco.disco_from_bev(torch.from_numpy(bevs).to(dev))
    a = spec[:1].expand(3, -1, -1, -1).contiguous()
    got = disco.calc_rel_ori(a, spec).cpu().numpy()
    A = a.cpu().numpy()[:, 0]; B = spec.cpu().numpy()[:, 0]
    for i in range(3):
        ra, ia, rb, ib = A[i].real, A[i].imag, B[i].real, B[i].imag
        cross = (ra * rb + ia * ib).astype(np.float64) + 1j * (ra * ib + rb * ia).astype(np.float64)
        real = (np.fft.ifft2(cross) * cross.size).real.astype(np.float32)
        assert got[i] == float(int(np.argmax(real)) % 120) * 3.0
        if oracle.ref_lib("relori") is not None:      # the reference's own calcRelOri (host build with FFTW / Eigen stand-ins)
            assert got[i] == oracle.ref_calc_rel_ori(A[i], B[i])
    # signature search: noisy copies of database entries must find their originals (exact brute force)
    dbn = rng.normal(size=(500, 1024)).astype(np.float32)
    pick = np.array([17, 499, 0, 256])
    qn = dbn[pick] + 0.05 * rng.normal(size=(4, 1024)).astype(np.float32)
    idx, d2 = disco.signature_search(torch.from_numpy(qn).to(dev), torch.from_numpy(dbn).to(dev))
    np.testing.assert_array_equal(idx.cpu().numpy(), pick)
    want = ((qn[:, None, :].astype(np.float64) - dbn[None]) ** 2).sum(-1).min(1)
    np.testing.assert_allclose(d2.cpu().numpy(), want, rtol=1e-5)
    # ragged sizes (tiles of 64 x 64 x 16 inside): odd dimension, database / query counts off the tile grid
    dbr = rng.normal(size=(777, 37)).astype(np.float32)
    qr = rng.normal(size=(130, 37)).astype(np.float32)
    idx, d2 = disco.signature_search(torch.from_numpy(qr).to(dev), torch.from_numpy(dbr).to(dev))
    full = ((qr[:, None, :].astype(np.float64) - dbr[None]) ** 2).sum(-1)
    np.testing.assert_array_equal(idx.cpu().numpy(), full.argmin(1))
    np.testing.assert_allclose(d2.cpu().numpy(), full.min(1), rtol=1e-5)
    # rotation invariance of the DiSCO signature: all three rotated copies are (near-)zero distance apart
    _, d2r = disco.signature_search(sig, sig[:1].contiguous())
    assert float(d2r.max()) < 1e-3 * float((sig ** 2).sum(1).max())


def test_signature_knn_equals_the_reference_kdtree(dev, oracle):
    """Row N4, the Mapping side's candidate query: the k = 10 nearest DiSCO signatures, ascending, vs the reference's own
    kd-tree built in place (oracle/_ref/libref_kdtree.so) and vs float64 brute force."""
    import torch
    from mr_slam_amd import disco
    rng = np.random.default_rng(11)
    for n, dim, k in ((3000, 1024, 10), (777, 37, 10), (100, 1024, 32), (5, 64, 10)):
        db = rng.normal(size=(n, dim)).astype(np.float32)
        q = np.stack([db[rng.integers(n)] + 0.05 * rng.normal(size=dim).astype(np.float32) for _ in range(3)]
                     + [rng.normal(size=dim).astype(np.float32)])
        idx, d2 = disco.signature_knn(torch.from_numpy(q).to(dev), torch.from_numpy(db).to(dev), k)
        idx, d2 = idx.cpu().numpy(), d2.cpu().numpy()
        full = ((q[:, None, :].astype(np.float64) - db[None]) ** 2).sum(-1)
        m = min(k, n)
        order = np.argsort(full, axis=1, kind="stable")[:, :m]
        np.testing.assert_array_equal(idx[:, :m], order)
        np.testing.assert_allclose(d2[:, :m], np.take_along_axis(full, order, 1), rtol=2e-5)
        assert (idx[:, m:] == -1).all() and np.isinf(d2[:, m:]).all()
        one_i, one_d = disco.signature_search(torch.from_numpy(q).to(dev), torch.from_numpy(db).to(dev))
        assert np.array_equal(one_i.cpu().numpy(), idx[:, 0]) and np.array_equal(one_d.cpu().numpy(), d2[:, 0])   # same tile kernel
        if oracle.ref_lib("kdtree") is not None:
            for i in range(q.shape[0]):
                ri, rd = oracle.ref_kdtree_knn(db, q[i], k)
                np.testing.assert_array_equal(idx[i, :len(ri)], ri)
                np.testing.assert_allclose(np.sqrt(d2[i, :len(ri)]), rd, rtol=2e-5)


def test_signature_knn_in_query_chunks(dev):
    """More queries than one distance-matrix chunk holds (2^28 floats): the chunked calls give what single-query calls give."""
    import torch
    from mr_slam_amd import disco
    g = torch.Generator(device=dev).manual_seed(3)
    db = torch.randn((70000, 32), device=dev, generator=g)
    q = torch.randn((4000, 32), device=dev, generator=g)            # 4000 x 70000 > 2^28: two chunks of 3834 + 166 queries
    idx, d2 = disco.signature_knn(q, db, 5)
    for i in (0, 3833, 3834, 3999):
        one_i, one_d = disco.signature_knn(q[i:i + 1], db, 5)
        assert torch.equal(one_i[0], idx[i]) and torch.equal(one_d[0], d2[i])
    full = torch.cdist(q[3990:].double(), db.double()) ** 2
    assert torch.equal(torch.topk(full, 5, dim=1, largest=False).indices.int(), idx[3990:])

