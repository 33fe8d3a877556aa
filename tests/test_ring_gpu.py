"""GPU parity suite: Radon, RING/RING++ descriptors, rotation correlation, translation."""
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


ANG = np.linspace(0, 2 * np.pi, 120).astype(np.float32)


def _bev_images(n, seed=0):
    """Sparse positive max-z style images like the Cartesian BEV produces."""
    rng = np.random.default_rng(seed)
    img = rng.uniform(0, 0.2, size=(n, 120, 120)).astype(np.float32)
    img[rng.uniform(size=img.shape) > 0.15] = 0
    return img


def test_radon_matches_oracle_bit_exact(dev, oracle):
    import torch
    from mr_slam_amd import ring
    img = _bev_images(5)
    img[3] = 0                                    # empty image
    img[4] = np.random.default_rng(9).normal(size=(120, 120)).astype(np.float32)  # dense, signed
    plan = ring.RadonPlan(120, ANG, 1.0, 120, 120)
    sino, _ = plan.forward(torch.from_numpy(img).to(dev))
    want = oracle.radon_parallel(img, ANG, 120, 1.0)
    got = sino.cpu().numpy()
    np.testing.assert_allclose(got, want, rtol=1e-6, atol=1e-6)
    assert np.array_equal(got, want), "expected bit-identical results (shared op order)"


# the last two do not fit the LDS: zero-bordered copy in global memory, same sample loop
@pytest.mark.parametrize("cfg", [(128, 128, 79, 2.0), (128, 128, 243, 0.5), (64, 100, 90, 1.3), (120, 120, 120, 1.0),
                                 (256, 256, 256, 1.0), (200, 333, 301, 1.5)])
def test_radon_other_geometries(dev, oracle, cfg):
    import torch
    from mr_slam_amd.compat import torch_radon
    H, W, det, spacing = cfg
    rng = np.random.default_rng(3)
    x = rng.uniform(0, 1, size=(2, 3, H, W)).astype(np.float32)
    ang = np.linspace(0, np.pi, 64, endpoint=False).astype(np.float32)
    radon = torch_radon.ParallelBeam(det, ang, spacing)
    y = radon.forward(torch.from_numpy(x).to(dev))
    assert tuple(y.shape) == (2, 3, 64, det)        # tests/test_torch.py:22-42 shape contract
    want = oracle.radon_parallel(x.reshape(-1, H, W), ang, det, spacing).reshape(2, 3, 64, det)
    np.testing.assert_allclose(y.cpu().numpy(), want, rtol=1e-6, atol=1e-6)
    with pytest.raises(RuntimeError):
        radon.forward(torch.from_numpy(x))          # CPU tensor rejected like pytorch.cpp:16-20


def test_radon_golden_and_analytic_bound(dev, golden_dir):
    """HIP sinogram vs the reference's analytic sinogram (fixture), bound of
    torch-radon/tests/test_parallel_beam.py:70."""
    import os
    import torch
    from mr_slam_amd import ring
    g = np.load(os.path.join(golden_dir, "radon_ring120.npz"))
    plan = ring.RadonPlan(120, g["angles"], 1.0, 120, 120)
    s, _ = plan.forward(torch.from_numpy(g["image"][None]).to(dev))
    s = s.cpu().numpy()[0]
    np.testing.assert_allclose(s, g["sino_oracle"], rtol=1e-6, atol=1e-6)
    err = np.linalg.norm(g["sino_analytic"] - s) / (np.linalg.norm(g["sino_analytic"]) + 1e-6)
    assert err < 2e-3 * (512 / 120) * (512 / 120)


def test_generate_ring_matches_restatement(dev, oracle):
    """generate_RING (util.py:174-200): BEV -> Radon -> normalise -> FFT over the angle axis."""
    from mr_slam_amd import ring, synth
    from oracle import corr_oracle as K
    pc = synth.lidar_scan(21, 30000)
    bev_img, sino, tiring = ring.generate_RING(pc)
    want_bev = oracle.bev_cart(synth.to_soa(pc), 1, 1, 120, 120, 1).reshape(-1, 3)[:, 2].reshape(1, 120, 120)
    np.testing.assert_array_equal(bev_img, want_bev)
    want_sino = oracle.radon_parallel(want_bev, ANG, 120, 1.0)
    np.testing.assert_allclose(sino.numpy(), want_sino, rtol=1e-6, atol=1e-6)
    want_t = K.tiring_from_sinogram(want_sino).numpy()
    scale = np.abs(want_t).max()
    assert np.abs(tiring.numpy() - want_t).max() < 2e-5 * scale


def test_normalize_and_row_fft(dev):
    import torch
    from mr_slam_amd import ring
    from oracle import corr_oracle as K
    rng = np.random.default_rng(5)
    x = np.abs(rng.normal(size=(2, 6, 120, 120))).astype(np.float32)
    t = torch.from_numpy(x).to(dev)
    n = ring.normalize(t).cpu().numpy()
    for b in range(2):
        np.testing.assert_allclose(n[b], K.ring_normalize(x[b]).numpy(), rtol=2e-5, atol=2e-6)
    m = ring.forward_row_fft(t).cpu().numpy()
    want, _ = K.forward_row_fft(x)
    np.testing.assert_allclose(m, want.numpy(), rtol=1e-4, atol=2e-5)


def _ring_db(n, seed):
    """n normalised sinograms (device-independent numpy) built from random sparse images."""
    from oracle import pyoracle as O
    from oracle import corr_oracle as K
    sino = O.radon_parallel(_bev_images(n, seed), ANG, 120, 1.0)
    return np.stack([K.ring_normalize(s[None]).numpy() for s in sino])   # [n,1,120,120]


def test_corr_sweep_matches_fast_corr_restatement(dev):
    import torch
    from mr_slam_amd import ring
    from oracle import corr_oracle as K
    db = _ring_db(6, 1)
    rng = np.random.default_rng(2)
    q = np.stack([np.roll(db[1], 17, axis=1) + 0.01 * rng.normal(size=db[1].shape).astype(np.float32),
                  db[4], np.roll(db[5], -33, axis=1)]).astype(np.float32)
    q = np.stack([K.ring_normalize(v).numpy() for v in q])
    dist, ang, corr = ring.corr_sweep(torch.from_numpy(q).to(dev), torch.from_numpy(db).to(dev), want_corr=True)
    dist, ang, corr = dist.cpu().numpy(), ang.cpu().numpy(), corr.cpu().numpy()
    for i in range(q.shape[0]):
        a = torch.fft.fft2(torch.from_numpy(q[i]), dim=-2, norm="ortho")
        for j in range(db.shape[0]):
            b = torch.fft.fft2(torch.from_numpy(db[j]), dim=-2, norm="ortho")
            d, g, c = K.fast_corr(a, b)
            np.testing.assert_allclose(corr[i, j], c, rtol=1e-4, atol=1e-3)
            assert abs(dist[i, j] - float(d)) < 1e-5
            top2 = np.sort(c)[-2:]
            if top2[1] - top2[0] > 1e-3 * top2[1]:      # unambiguous maximum
                assert ang[i, j] == g
    assert ang[0, 1] == -17 and ang[1, 4] == 0 and ang[2, 5] == 33   # query rolled by +k <=> angle -k
    assert dist[1, 4] <= dist.min() + 1e-6


def test_fast_corr_literal_spectra(dev):
    """The drop-in fast_corr(a, b) on complex TIRING tensors, incl. non-Hermitian input."""
    import torch
    from mr_slam_amd import ring
    from oracle import corr_oracle as K
    db = _ring_db(2, 4)
    a = torch.fft.fft2(torch.from_numpy(db[0]), dim=-2, norm="ortho")
    b = torch.fft.fft2(torch.from_numpy(np.roll(db[0], 9, axis=1)), dim=-2, norm="ortho")
    for (u, v) in ((a, b), (b, a), (a, a)):
        d, g, c = ring.fast_corr(u, v, want_corr=True)
        wd, wg, wc = K.fast_corr(u, v)
        np.testing.assert_allclose(c, wc, rtol=1e-4, atol=1e-3)
        assert g == wg and abs(d - float(wd)) < 1e-5
    rng = np.random.default_rng(6)
    z1 = torch.from_numpy((rng.normal(size=(2, 120, 120)) + 1j * rng.normal(size=(2, 120, 120))).astype(np.complex64))
    z2 = torch.from_numpy((rng.normal(size=(2, 120, 120)) + 1j * rng.normal(size=(2, 120, 120))).astype(np.complex64))
    d, g, c = ring.fast_corr(z1, z2, want_corr=True)
    wd, wg, wc = K.fast_corr(z1, z2)
    np.testing.assert_allclose(c, wc, rtol=2e-4, atol=2e-3)


def test_fast_corr_ringplusplus(dev):
    from mr_slam_amd import ring
    from oracle import corr_oracle as K
    rng = np.random.default_rng(7)
    a = np.abs(rng.normal(size=(6, 120, 120))).astype(np.float32)
    b = (np.roll(a, 23, axis=1) + 0.02 * np.abs(rng.normal(size=a.shape))).astype(np.float32)
    d, g = ring.fast_corr_RINGplusplus(a, b)
    wd, wg, _ = K.fast_corr_ringplusplus(a, b)
    assert g == wg and abs(d - float(wd)) < 2e-5


def test_solve_translation(dev):
    from mr_slam_amd import ring
    from oracle import corr_oracle as K
    from oracle import pyoracle as O
    img = _bev_images(1, 8)[0]
    shifted = np.roll(np.roll(img, 5, axis=0), -3, axis=1)
    q = O.radon_parallel(img, ANG, 120, 1.0)[None]
    p = O.radon_parallel(shifted, ANG, 120, 1.0)[None]
    x, y, err, sh = ring.solve_translation(q, p, 0.3, want_shifts=True, least_squares=True)
    wx, wy, werr, wsh = K.solve_translation(q, p, 0.3, literal=False)
    # integer row shifts: equal, or -- where the two FFT implementations pick different maxima -- the float64 circular
    # correlation of that row has the same value at both positions to fp32 precision (a genuine tie)
    for i in np.flatnonzero(sh != wsh):
        c = np.abs(np.fft.ifft(np.fft.fft(q[0, i].astype(np.float64)) * np.conj(np.fft.fft(p[0, i].astype(np.float64)))))
        c = np.fft.fftshift(c)
        a, b = c[int(60 - sh[i])], c[int(60 - wsh[i])]
        assert abs(a - b) <= 2e-6 * c.max(), (i, sh[i], wsh[i], a, b)
    assert (sh == wsh).mean() > 0.9
    # the 120 x 2 least-squares solve, unconditionally: on the GPU's own shifts against a float64 pseudo-inverse
    ang = np.linspace(0, 2 * np.pi, 120).astype(np.float32).astype(np.float64) + 0.3
    Amat = np.stack([np.cos(ang), np.sin(ang)], 1)
    sol = np.linalg.pinv(Amat) @ sh.astype(np.float64)
    res = np.linalg.norm(Amat @ sol - sh)
    assert abs(x[0] - sol[0]) < 1e-3 and abs(y[0] - sol[1]) < 1e-3 and abs(err - res) < 1e-2 * max(res, 1.0)
    if (sh == wsh).all():
        assert abs(x[0] - wx.item()) < 1e-3 and abs(y[0] - wy.item()) < 1e-3 and abs(err - werr.item()) < 1e-2


def test_full_size_pipeline_properties(dev):
    """BASELINE size: 120k-point scans -> descriptors; rotating a scan by k*3 degrees about z
    rolls its sinogram, so the sweep must report that rotation and dist(self) ~ minimal."""
    import torch
    from mr_slam_amd import bev, ring, synth
    base = synth.lidar_scan(31, metric=True)
    scans = []
    for k in (0, 10, 33):
        th = np.deg2rad(3.0 * k)
        R = np.array([[np.cos(th), -np.sin(th), 0], [np.sin(th), np.cos(th), 0], [0, 0, 1]], np.float32)
        scans.append(synth.preprocess(base @ R.T))
    scans.append(synth.lidar_scan(77))
    xyz, offs = bev.pack_scans(scans, dev)
    _, sino, norm = ring.ring_descriptors(xyz, offs)
    assert torch.isfinite(norm).all()
    d, a = ring.corr_sweep(norm[:1, None], norm[:, None])
    d, a = d.cpu().numpy()[0], a.cpu().numpy()[0]
    assert d[0] < d[1] < d[3] and d[2] < d[3]
    assert a[0] == 0 and abs(abs(a[1]) - 10) <= 1 and abs(abs(a[2]) - 33) <= 1
    # determinism
    d2, a2 = ring.corr_sweep(norm[:1, None], norm[:, None])
    assert np.array_equal(d2.cpu().numpy()[0], d) and np.array_equal(a2.cpu().numpy()[0], a)


def test_fft_domain_sweep_matches_reference_and_direct_kernel(dev):
    """Half-spectrum database + in-register FFT correlation vs fast_corr (util.py:362-374) and vs the
    direct sinogram-domain kernel."""
    import torch
    from mr_slam_amd import ring
    from oracle import corr_oracle as K
    db = _ring_db(7, 11)
    q = np.stack([np.roll(db[2], 21, axis=1), db[5], np.roll(db[6], -7, axis=1)])
    tq, tdb = torch.from_numpy(q).to(dev), torch.from_numpy(db).to(dev)
    sq, sdb = ring.half_spectrum(tq[:, 0]), ring.half_spectrum(tdb[:, 0])
    # the half spectrum is the first 61 rows of the reference's TIRING
    want_spec = torch.fft.fft2(torch.from_numpy(db[:, 0]), dim=-2, norm="ortho")[:, :61].numpy()
    assert np.abs(sdb.cpu().numpy() - want_spec).max() < 2e-5 * np.abs(want_spec).max()
    dist, ang, corr = ring.corr_sweep_fft(sq, sdb, want_corr=True)
    d2, a2, c2 = ring.corr_sweep(tq, tdb, want_corr=True)
    np.testing.assert_allclose(corr.cpu().numpy(), c2.cpu().numpy(), rtol=1e-4, atol=1e-3)
    assert np.abs(dist.cpu().numpy() - d2.cpu().numpy()).max() < 1e-5
    dist, ang = dist.cpu().numpy(), ang.cpu().numpy()
    for i in range(q.shape[0]):
        a = torch.fft.fft2(torch.from_numpy(q[i]), dim=-2, norm="ortho")
        for j in range(db.shape[0]):
            b = torch.fft.fft2(torch.from_numpy(db[j]), dim=-2, norm="ortho")
            wd, wa, wc = K.fast_corr(a, b)
            assert abs(dist[i, j] - float(wd)) < 1e-5
            top2 = np.sort(wc)[-2:]
            if top2[1] - top2[0] > 1e-3 * top2[1]:
                assert ang[i, j] == wa
    assert ang[0, 2] == -21 and ang[1, 5] == 0 and ang[2, 6] == 7
    dp, ap = ring.corr_pairs_fft(sq, sdb[[2, 5, 6]])
    assert np.array_equal(ap.cpu().numpy(), [-21, 0, 7])
    assert np.abs(dp.cpu().numpy() - dist[[0, 1, 2], [2, 5, 6]]).max() < 1e-6


def test_fp16_replica_format(dev):
    """Multi-GPU exchange format: fp16 copies of the half spectra are exactly the rounded fp32 ones, and a
    sweep over replicas agrees with the fp32 sweep (dist within 2e-3, same angle where the peak is clear)."""
    import torch
    from mr_slam_amd import ring
    db = _ring_db(9, 21)
    q = np.stack([np.roll(db[2], 21, axis=1), db[5], np.roll(db[8], -40, axis=1)])
    tq, tdb = torch.from_numpy(q).to(dev), torch.from_numpy(db).to(dev)
    sq = ring.half_spectrum(tq[:, 0])
    sdb, sdb16 = ring.half_spectrum_f16(tdb[:, 0])
    assert sdb16.dtype == torch.float16 and sdb16.shape == (9, 61, 120, 2)
    assert torch.equal(sdb, ring.half_spectrum(tdb[:, 0]))
    assert torch.equal(sdb16, torch.view_as_real(sdb).to(torch.float16))       # round-to-nearest-even
    only16 = ring.half_spectrum_f16(tdb[:, 0], want_f32=False)
    assert only16[0] is None and torch.equal(only16[1], sdb16)
    d32, a32, c32 = ring.corr_sweep_fft(sq, sdb, want_corr=True)
    d16, a16, c16 = ring.corr_sweep_fft(sq, sdb16, want_corr=True)
    assert float((d16 - d32).abs().max()) < 2e-3
    np.testing.assert_allclose(c16.cpu().numpy(), c32.cpu().numpy(), rtol=2e-3, atol=0.2)
    c = c32.cpu().numpy()
    top2 = np.sort(c, axis=-1)[..., -2:]
    clear = (top2[..., 1] - top2[..., 0]) > 1e-2 * top2[..., 1]
    assert clear.sum() >= 3
    assert np.array_equal(a16.cpu().numpy()[clear], a32.cpu().numpy()[clear])
    assert a16[0, 2] == -21 and a16[1, 5] == 0 and a16[2, 8] == 40


def test_ringpp_fft_domain_matches_restatement(dev):
    """RING++ (6 channels): FFT-domain sweep / pairs on [C][61][120] half spectra vs fast_corr_RINGplusplus
    (util.py:337-358) and vs the direct sinogram-domain kernel."""
    import torch
    from mr_slam_amd import ring
    from oracle import corr_oracle as K
    rng = np.random.default_rng(31)
    C = 6
    raw = rng.uniform(0, 1, size=(5, C, 120, 120)).astype(np.float32)
    raw *= (rng.uniform(size=(5, C, 120, 120)) < 0.3)
    db = raw.copy()
    q = np.stack([np.roll(db[1], 13, axis=1), db[3], np.roll(db[4], -50, axis=1)])
    tq, tdb = torch.from_numpy(q).to(dev), torch.from_numpy(db).to(dev)
    nq, ndb = ring.normalize(tq), ring.normalize(tdb)              # one mean/std over all channels (util.py:339-340)
    sq, sdb = ring.half_spectrum(nq), ring.half_spectrum(ndb)
    assert sq.shape == (3, C, 61, 120)
    dist, ang, corr = ring.corr_sweep_fft(sq, sdb, want_corr=True)
    d2, a2, c2 = ring.corr_sweep(nq, ndb, want_corr=True)
    np.testing.assert_allclose(corr.cpu().numpy(), c2.cpu().numpy(), rtol=1e-4, atol=5e-3)
    assert np.abs(dist.cpu().numpy() - d2.cpu().numpy()).max() < 1e-5
    dist, ang = dist.cpu().numpy(), ang.cpu().numpy()
    for i in range(3):
        for j in range(5):
            wd, wa, wc = K.fast_corr_ringplusplus(q[i], db[j])
            assert abs(dist[i, j] - float(wd)) < 1e-5
            top2 = np.sort(wc)[-2:]
            if top2[1] - top2[0] > 1e-3 * top2[1]:
                assert ang[i, j] == wa
    assert ang[0, 1] == -13 and ang[1, 3] == 0 and ang[2, 4] == 50
    dp, ap = ring.corr_pairs_fft(sq, sdb[[1, 3, 4]])
    assert np.array_equal(ap.cpu().numpy(), [-13, 0, 50])
    assert np.abs(dp.cpu().numpy() - dist[[0, 1, 2], [1, 3, 4]]).max() < 1e-6
    d1, a1 = ring.fast_corr_RINGplusplus(q[0], db[1], device=dev)     # the drop-in, now on the FFT-domain kernel
    assert abs(float(d1) - float(K.fast_corr_ringplusplus(q[0], db[1])[0])) < 1e-5 and int(a1) == -13
    # throughput note (not asserted): see bench.py sweep leg


def test_bench_sized_batch_is_deterministic_and_matches_oracle_samples(dev, oracle):
    """The bench.py step at its real size (512 scans x 120k points): two runs are bitwise identical, sampled
    entries are bit-exact with the restatements (BEV, sinogram), every self-pair scores the same distance with
    angle 0, and pairing each scan with its neighbour is symmetric in distance and antisymmetric in angle."""
    import torch
    from mr_slam_amd import bev, ring, synth
    B = 512
    base = [synth.lidar_scan(s) for s in range(8)]
    xyz, offs = bev.pack_scans([base[i % 8] for i in range(B)], dev)
    plan = ring.ring_plan(0)

    def run():
        img = bev.cart_bev(xyz, offs, 1, 1, 120, 120, 1)
        raw, norm = plan.forward(img.view(B, 120, 120), raw=True, normalized=True)
        spec = ring.half_spectrum(norm)
        d, a = ring.corr_pairs_fft(spec, spec.roll(1, 0).contiguous())
        return img, raw, norm, spec, d, a

    r1, r2 = run(), run()
    for x, y in zip(r1, r2):
        assert torch.equal(x, y)
    img, raw, norm, spec, d, a = r1
    ang = np.linspace(0, 2 * np.pi, 120).astype(np.float32)
    for i in (0, 3, 257, 511):
        want = oracle.bev_cart(synth.to_soa(base[i % 8]), 1, 1, 120, 120, 1).reshape(-1, 3)[:, 2].reshape(120, 120)
        np.testing.assert_array_equal(img[i].cpu().numpy().reshape(120, 120), want)
        np.testing.assert_array_equal(raw[i].cpu().numpy(), oracle.radon_parallel(want, ang, 120, 1.0))
    ds, as_ = ring.corr_pairs_fft(spec, spec)
    assert int(as_.abs().max()) == 0
    # scans repeat with period 8, so entry i and i + 8 are the same scan: identical results
    assert torch.equal(d[8:], d[:-8]) and torch.equal(a[8:], a[:-8])
    d_rev, a_rev = ring.corr_pairs_fft(spec.roll(1, 0).contiguous(), spec)
    assert float((d - d_rev).abs().max()) < 1e-6
    clear = (d < 0.9 * float(d.max()))        # a clear peak: the reversed pair must report the opposite rotation
    assert int(((a + a_rev) % 120)[clear].abs().max() if clear.any() else 0) == 0


def test_more_pairs_than_one_grid_dimension(dev):
    """70 000 pairs in one call (HIP limits grid.y to 65 535): the entry point chunks; every pair still gets its result."""
    import torch
    from mr_slam_amd import ring
    g = torch.Generator(device=dev).manual_seed(3)
    base = ring.half_spectrum(ring.normalize(torch.randn((64, 120, 120), device=dev, generator=g)))
    P = 70000
    idx = torch.arange(P, device=dev) % 64
    a = base[idx].contiguous()
    b = base[(idx + 1) % 64].contiguous()
    d, ang = ring.corr_pairs_fft(a, b)
    d64, a64 = ring.corr_pairs_fft(base, base.roll(-1, 0).contiguous())
    assert torch.equal(d, d64[idx]) and torch.equal(ang, a64[idx])
    # direct (sinogram-domain) kernel: same limit, same chunking
    sino = ring.normalize(torch.randn((8, 1, 60, 40), device=dev, generator=g))
    i2 = torch.arange(P, device=dev) % 8
    d2, a2 = ring.corr_pairs(sino[i2].contiguous(), sino[(i2 + 3) % 8].contiguous())
    d8, a8 = ring.corr_pairs(sino, sino.roll(-3, 0).contiguous())
    assert torch.equal(d2, d8[i2]) and torch.equal(a2, a8[i2])


def test_fused_spectrum_and_pair_correlation_is_bitwise_the_two_step_path(dev):
    import torch
    from mr_slam_amd import ring
    g = torch.Generator(device=dev).manual_seed(5)
    norm = ring.normalize(torch.randn((37, 120, 120), device=dev, generator=g))
    cand = ring.half_spectrum(ring.normalize(torch.randn((37, 120, 120), device=dev, generator=g)))
    cand[3] = ring.half_spectrum(norm[3:4].roll(25, 1))[0]
    spec, spec16, d, a = ring.spectrum_corr_pairs(norm, cand, want_f16=True)
    want_spec, want16 = ring.half_spectrum_f16(norm)
    wd, wa = ring.corr_pairs_fft(want_spec, cand)
    assert torch.equal(spec, want_spec) and torch.equal(spec16, want16)
    assert torch.equal(d, wd) and torch.equal(a, wa) and int(a[3]) == 25
    none, only16, d2, a2 = ring.spectrum_corr_pairs(norm, cand, want_f32=False, want_f16=True)
    assert none is None and torch.equal(only16, want16) and torch.equal(d2, wd) and torch.equal(a2, wa)


def test_blank_scan_is_reported_not_silently_nan(dev):
    """A blank / fully cropped scan gives a constant sinogram (std == 0).  The reference's fn.normalize raises ValueError
    (util.py:197); the kernels write a finite all-zero descriptor and count the event, the host mirror raises."""
    import torch
    from mr_slam_amd import ring
    plan = ring.ring_plan(0)
    plan.degenerate_count()
    imgs = torch.zeros((5, 120, 120), device=dev)
    imgs[1, 40:60, 30:90] = 0.5
    imgs[3, 10, 10] = 1.0
    sino, norm = plan.forward(imgs, raw=True, normalized=True)
    assert plan.degenerate_count() == 3                      # images 0, 2, 4
    assert plan.degenerate_count() == 0                      # reset by the read
    assert torch.isfinite(norm).all() and (norm[0] == 0).all() and (norm[2] == 0).all() and norm[1].std() > 0.99
    one, n1 = plan.forward(imgs[:1], raw=True, normalized=True)   # single image -> the one-image kernel: same behaviour
    assert plan.degenerate_count() == 1 and (n1 == 0).all()
    far = np.full((100, 3), 5.0, np.float32); far[:, 2] = -1.0   # nothing with z > 0 inside the grid: blank BEV
    with pytest.raises(ValueError):
        ring.generate_RING(far, dev)
    x = ring.normalize(torch.ones((2, 1, 120, 120), device=dev))
    assert (x == 0).all()


def test_two_images_per_workgroup_equal_one_image_per_workgroup(dev, oracle):
    """The three LDS kernels give the same bits, raw and normalised: k_radon2 (batches > 128: pairs of images interleaved in
    the LDS; even and odd batch), k_radon_split + k_normalize (batches <= 128: one image over 15 workgroups), and the checker."""
    import torch
    from mr_slam_amd import ring
    rng = np.random.default_rng(5)
    imgs = (rng.random((131, 120, 120)) * (rng.random((131, 120, 120)) < 0.3)).astype(np.float32)
    t = torch.from_numpy(imgs).to(dev)
    plan = ring.ring_plan(0)
    s_odd, n_odd = plan.forward(t, raw=True, normalized=True)               # 131 images: k_radon2, last workgroup half empty
    s_even, n_even = plan.forward(t[:130], raw=True, normalized=True)       # 130 images: k_radon2
    assert torch.equal(s_even, s_odd[:130]) and torch.equal(n_even, n_odd[:130])
    ang = np.linspace(0, 2 * np.pi, 120).astype(np.float32)
    np.testing.assert_array_equal(s_odd[:9].cpu().numpy(), oracle.radon_parallel(imgs[:9], ang, 120, 1.0))
    for lo, hi in ((0, 1), (1, 8), (8, 131 - 3), (130, 131)):             # 1, 7, 120 and 1 images: the split path
        s, n = plan.forward(t[lo:hi], raw=True, normalized=True)
        assert torch.equal(s, s_odd[lo:hi]) and torch.equal(n, n_odd[lo:hi])
    only_norm = plan.forward(t[3:4], raw=False, normalized=True)[1]
    assert torch.equal(only_norm, n_odd[3:4])


def test_spectrum_corr_pairs_db_indexes_the_database_and_flags_missing_rows(dev):
    """Candidates picked by row out of a database (exact entries or fp16 replicas) == the pairwise kernel on the gathered
    rows, bit for bit (fp32 DB) / within the replica bound (fp16 DB); an index outside the database reports no match."""
    import torch
    from mr_slam_amd import ring
    rng = np.random.default_rng(11)
    imgs = torch.from_numpy((rng.random((40, 120, 120)) * (rng.random((40, 120, 120)) < 0.3)).astype(np.float32)).to(dev)
    _, norm = ring.ring_plan(0).forward(imgs, raw=False, normalized=True)
    db = ring.half_spectrum(norm[:32])
    q = norm[32:]
    idx = torch.tensor([5, 31, 0, 17, 17, 2, 99, -1], dtype=torch.int32, device=dev)
    slot = torch.zeros((8, 61, 120), dtype=torch.complex64, device=dev)
    spec, _, d, a = ring.spectrum_corr_pairs_db(q, db, idx, spec_out=slot)
    assert spec.data_ptr() == slot.data_ptr() and torch.equal(slot, ring.half_spectrum(q))
    ok = torch.tensor([True] * 6 + [False] * 2, device=dev)
    wd, wa = ring.corr_pairs_fft(ring.half_spectrum(q)[ok], db[idx[ok].long()])
    assert torch.equal(d[ok], wd) and torch.equal(a[ok], wa)
    assert torch.isinf(d[~ok]).all() and (a[~ok] == 0).all()
    db16 = torch.view_as_real(db).to(torch.float16)
    _, s16, d16, a16 = ring.spectrum_corr_pairs_db(q, db16, idx, want_f16=True)
    assert (d16[ok] - wd).abs().max() < 2e-3 and torch.isinf(d16[~ok]).all()
    assert torch.equal(s16, torch.view_as_real(ring.half_spectrum(q)).to(torch.float16))


def test_multi_round_sweeps_equal_the_pairwise_kernel_bit_for_bit(dev):
    """Databases large enough that every sweep kernel runs several (ragged) rounds per workgroup: the one-query and
    several-query RING sweeps, the fp16-replica sweep with the next candidate prefetched (1-2 queries) and without (3), and
    the channel-outer RING++ sweep must all give exactly what the one-candidate-per-workgroup pairwise kernel gives."""
    import torch
    from mr_slam_amd import ring
    g = torch.Generator(device=dev).manual_seed(5)
    n_db = 2 * 2 * 256 * 2 + 77                      # > 2 rounds of 2 x (2 workgroups per CU) x 256 CUs, ragged tail
    sino = torch.rand((n_db, 120, 120), device=dev, generator=g) * (torch.rand((n_db, 120, 120), device=dev, generator=g) < 0.3)
    sdb, sdb16 = ring.half_spectrum_f16(ring.normalize(sino[:, None])[:, 0])
    sq = sdb[[3, n_db - 1, 1000]].contiguous()
    pick = torch.tensor([0, 1, 511, 512, 1023, 1024, 1500, 2047, 2048, n_db - 2, n_db - 1], device=dev)
    for nq in (1, 3):
        d, a = ring.corr_sweep_fft(sq[:nq], sdb)
        for qi in range(nq):
            wd, wa = ring.corr_pairs_fft(sq[qi:qi + 1].expand(len(pick), -1, -1).contiguous(), sdb[pick].contiguous())
            assert torch.equal(d[qi, pick], wd) and torch.equal(a[qi, pick], wa)
        assert int(torch.argmin(d[0])) == 3 and int(a[0, 3]) == 0      # the query is entry 3 itself
    d16_1, a16_1 = ring.corr_sweep_fft(sq[:1], sdb16)              # prefetching kernel
    d16_3, a16_3 = ring.corr_sweep_fft(sq, sdb16)                  # plain kernel
    assert torch.equal(d16_1[0], d16_3[0]) and torch.equal(a16_1[0], a16_3[0])
    assert float((d16_3 - ring.corr_sweep_fft(sq, sdb)[0]).abs().max()) < 2e-3
    # RING++: 6 channels, > 1 round of 8 candidates per slot
    C, n_pp = 6, 2 * 256 * 2 * 8 + 333
    spp = sdb.new_empty((n_pp, C, 61, 120))
    idx = torch.randint(0, n_db, (n_pp, C), device=dev, generator=g)
    spp.copy_(sdb[idx.reshape(-1)].reshape(n_pp, C, 61, 120))       # channel planes drawn from the RING entries
    qpp = spp[[5, n_pp - 1]].contiguous()
    pick = torch.tensor([0, 5, 15, 16, 4095, 4096, 8191, 8192, n_pp - 1], device=dev)
    for nq in (1, 2):
        d, a = ring.corr_sweep_fft(qpp[:nq], spp)
        for qi in range(nq):
            wd, wa = ring.corr_pairs_fft(qpp[qi:qi + 1].expand(len(pick), -1, -1, -1).contiguous(), spp[pick].contiguous())
            assert torch.equal(d[qi, pick], wd) and torch.equal(a[qi, pick], wa)
    assert int(torch.argmin(d[0])) == 5 and int(torch.argmin(d[1])) == n_pp - 1


def test_blocked_sweeps_equal_single_sweeps_bit_for_bit(dev):
    """mrs_ring_corr_fft_sweep_blocks: several (query, database slice) sweeps of one descriptor pool in ONE launch (bench.py: the 16 per-launch
    sweeps of a group) give exactly what one mrs_ring_corr_fft_sweep per query gives -- overlapping, adjacent and repeated slices, a
    query that lies inside its own slice, slices that end at the pool's last entry."""
    import torch
    from mr_slam_amd import ring
    g = torch.Generator(device=dev).manual_seed(9)
    n_pool, n_db = 3000, 700
    sino = torch.rand((n_pool, 120, 120), device=dev, generator=g) * (torch.rand((n_pool, 120, 120), device=dev, generator=g) < 0.3)
    pool = ring.half_spectrum(ring.normalize(sino[:, None])[:, 0]).contiguous()
    qrow = torch.tensor([5, 2999, 100, 100, 1234, 0, 2300], dtype=torch.int64, device=dev)
    first = torch.tensor([0, 2300, 50, 0, 1234, 2300, 1600], dtype=torch.int64, device=dev)
    d, a = ring.corr_sweep_fft_blocks(pool, qrow, first, n_db, check=True)
    assert d.shape == (7, n_db)
    for i in range(qrow.numel()):
        wd, wa = ring.corr_sweep_fft(pool[int(qrow[i]):int(qrow[i]) + 1], pool[int(first[i]):int(first[i]) + n_db])
        assert torch.equal(d[i], wd[0]) and torch.equal(a[i], wa[0]), i
    assert int(torch.argmin(d[0])) == 5 and int(torch.argmin(d[2])) == 50 and int(torch.argmin(d[4])) == 0     # the query itself where its slice holds it


def test_several_queries_per_dma_sweep_equal_single_query_sweeps_bit_for_bit(dev):
    """Round 6: mrs_ring_corr_fft_sweep_tiled_q (and the row-layout mrs_ring_corr_fft_sweep[_mc] with 2 .. 32 queries) run the one-query LDS-DMA
    pipeline for Q queries in one launch, the queries' workgroups grouped per XCD.  Every (query, entry) must carry the bits of the
    one-query sweep and of the register-staged k_ring_corr_fft (MRS_SWEEP_MQ_VARIANT=0 is only reachable under MRS_DEV, so the pairwise
    kernel is the independent reference here): Q = 2, 4, 5 (groups with idle workgroups), 33 (two launches), databases smaller than one
    round of a workgroup and ragged ones; RING++ with Q = 3 and 9 (two launches)."""
    import torch
    from mr_slam_amd import ring
    g = torch.Generator(device=dev).manual_seed(21)
    n_pool = 2500
    sino = torch.rand((n_pool, 120, 120), device=dev, generator=g) * (torch.rand((n_pool, 120, 120), device=dev, generator=g) < 0.3)
    pool = ring.half_spectrum(ring.normalize(sino[:, None])[:, 0]).contiguous()
    for n_db in (2500, 2049, 61, 5):
        db = pool[:n_db].contiguous()
        tiled = ring.spec_to_tiled(db)
        for nq in (2, 4, 5, 33):
            qrows = torch.randint(0, n_pool, (nq,), device=dev, generator=g)
            q = pool[qrows].contiguous()
            d, a = ring.corr_sweep_fft_tiled_q(q, tiled)
            assert d.shape == (nq, n_db)
            dr, ar = ring.corr_sweep_fft(q, db)                     # row layout, same pipeline for nq <= 32
            assert torch.equal(d, dr) and torch.equal(a, ar), (n_db, nq)
            for qi in (0, nq - 1):
                d1, a1 = ring.corr_sweep_fft_tiled(q[qi], tiled)    # the one-query form
                assert torch.equal(d[qi], d1) and torch.equal(a[qi], a1), (n_db, nq, qi)
            pick = torch.randint(0, n_db, (min(n_db, 64),), device=dev, generator=g)
            wd, wa = ring.corr_pairs_fft(q[:1].expand(len(pick), -1, -1).contiguous(), db[pick].contiguous())
            assert torch.equal(d[0, pick], wd) and torch.equal(a[0, pick], wa)
    # RING++: 6 channels
    C = 6
    for n_pp in (700, 9):
        idx = torch.randint(0, n_pool, (n_pp, C), device=dev, generator=g)
        spp = pool[idx.reshape(-1)].reshape(n_pp, C, 61, 120).contiguous()
        tiled = ring.spec_to_tiled(spp)
        for nq in (3, 9):
            q = spp[torch.randint(0, n_pp, (nq,), device=dev, generator=g)].contiguous()
            d, a = ring.corr_sweep_fft_tiled_q(q, tiled)
            for qi in range(nq):
                d1, a1 = ring.corr_sweep_fft_tiled(q[qi], tiled)
                assert torch.equal(d[qi], d1) and torch.equal(a[qi], a1), (n_pp, nq, qi)
            pick = torch.randint(0, n_pp, (min(n_pp, 32),), device=dev, generator=g)
            wd, wa = ring.corr_pairs_fft(q[1:2].expand(len(pick), -1, -1, -1).contiguous(), spp[pick].contiguous())
            assert torch.equal(d[1, pick], wd) and torch.equal(a[1, pick], wa)
