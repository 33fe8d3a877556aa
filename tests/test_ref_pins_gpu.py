"""GPU suite: the HIP path (through mr_slam_amd's host mirrors of the reference functions) against output of the
REFERENCE's own Python run in the build container (tests/golden/ref_corr.npz; generator
tests/golden/make_golden_ref_corr.py).  Rows R2, C1-C4, D1-D2, N1 of SURVEY.md section 8(a).
Tolerances: integer outputs (angles, shifts, bins, cell indices) exact; floating point within the stated bound."""
import os

import numpy as np
import pytest
import torch

from test_ref_pins import G, mk, clouds, _loop_AB, _pp_bev, _f  # noqa: F401  (fixtures + helpers shared with the CPU suite)

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module", autouse=True)
def _lib_loaded():
    assert torch.cuda.is_available()
    from mr_slam_amd import _lib
    _lib.load()


def test_generate_RING_matches_reference(G, clouds):
    """util.generate_RING: pc_bev exact, pc_RING within the Radon deviation bound of DESIGN.md (fp32 vs the checker:
    bit-exact), pc_TIRING 2e-5 absolute (ortho FFT of a normalised sinogram, |values| <= ~60)."""
    from mr_slam_amd import ring
    for name, pc in zip("AB", clouds[:2]):
        bev, RING, TIRING = ring.generate_RING(pc, DEV)
        occ = np.flatnonzero(bev.reshape(-1))
        np.testing.assert_array_equal(occ, G[f"ring_bev_occ_{name}"])
        np.testing.assert_array_equal(bev.reshape(-1)[occ], G[f"ring_bev_val_{name}"])
        np.testing.assert_array_equal(RING.numpy(), G[f"ring_RING_{name}"])
        np.testing.assert_allclose(TIRING.numpy(), G[f"ring_TIRING_{name}"], rtol=0, atol=2e-5)


def test_fast_corr_matches_reference(G, clouds):
    from mr_slam_amd import ring
    T = {n: torch.from_numpy(G[f"ring_TIRING_{n}"]) for n in "AB"}
    T["C"] = ring.generate_RING(clouds[2], DEV)[2]
    for a, b in (("A", "B"), ("A", "C"), ("B", "C"), ("A", "A"), ("B", "A")):
        dist, angle = ring.fast_corr(T[a], T[b], DEV)                      # literal spectra entry (mrs_ring_corr_spectra)
        wd, wa = G[f"fast_corr_{a}{b}"]
        assert int(angle) == int(wa) and abs(float(dist) - wd) < 1e-5, (a, b, dist, angle, wd, wa)
    dist, angle = ring.fast_corr(torch.cat([T["A"], T["C"]]), torch.cat([T["B"], T["C"]]), DEV)
    wd, wa = G["fast_corr_2ch"]
    assert int(angle) == int(wa) and abs(float(dist) - wd) < 1e-5          # C = 2: the reference has no channel factor
    # the hot path (half-spectrum database + FFT-domain kernel) gives the same answers
    norm = {n: ring.normalize(torch.from_numpy(G[f"ring_RING_{n}"]).to(DEV)) for n in "AB"}
    hs = {n: ring.half_spectrum(norm[n]) for n in "AB"}
    for a, b in (("A", "B"), ("B", "A"), ("A", "A")):
        dist, ang = ring.corr_pairs_fft(hs[a], hs[b])
        wd, wa = G[f"fast_corr_{a}{b}"]
        assert int(ang[0]) == int(wa) and abs(float(dist[0]) - wd) < 1e-5
    d, ang = ring.corr_sweep_fft(hs["A"], torch.cat([hs["B"], hs["A"]]))
    assert [int(v) for v in ang[0]] == [int(G["fast_corr_AB"][1]), int(G["fast_corr_AA"][1])]


def test_loop_sequence_and_solve_translation_match_reference(G):
    """main_RING.py:147-178 replayed on the product: fast_corr -> row shifts -> solve_translation.  The DEFAULT call returns the
    reference-run numbers (its method='svd' call, util.py:415); least_squares=True its 'pinv' branch."""
    from mr_slam_amd import ring
    RA, RBs, RBe, rad, rad_e = _loop_AB(G)
    for i, (pos, r, tag) in enumerate(((RBs, rad, ""), (RBe, rad_e, "_extra"))):
        x, y, err, sh = ring.solve_translation(RA, pos, r, DEV, want_shifts=True)
        np.testing.assert_array_equal(sh, G["solve_translation_b"][i])                  # every one of the 120 row shifts
        np.testing.assert_allclose([_f(x), _f(y), _f(err)], G[f"solve_translation_AB{tag}"], rtol=2e-4, atol=2e-4)
        x, y, err = ring.solve_translation(RA, pos, r, DEV)
        np.testing.assert_allclose([_f(x), _f(y), _f(err)], G[f"solve_translation_AB{tag}"], rtol=2e-4, atol=2e-4)
        x, y, err = ring.solve_translation(RA, pos, r, DEV, least_squares=True)
        np.testing.assert_allclose([_f(x), _f(y)], G[f"solve_translation_AB{tag}_pinv"], rtol=1e-4, atol=1e-4)


def test_ringplusplus_correlation_matches_reference(G, mk):
    from mr_slam_amd import ring
    RING = torch.from_numpy(G["pp_RING_A"]).to(DEV); TIRING = G["pp_TIRING_A"]
    out = ring.forward_row_fft(RING).cpu().numpy()
    np.testing.assert_allclose(out, TIRING, rtol=0, atol=1e-5 * np.abs(TIRING).max())
    Tb = mk.rolled_noisy(TIRING, 23, seed=5)
    for tag, (a, b) in (("AA", (TIRING, TIRING)), ("Arolled", (TIRING, Tb)), ("rolledA", (Tb, TIRING))):
        dist, angle = ring.fast_corr_RINGplusplus(a, b, DEV)
        wd, wa = G[f"fast_corr_pp_{tag}"]
        assert int(angle) == int(wa) and abs(float(dist) - wd) < 1e-5, (tag, dist, angle, wd, wa)


def test_solve_translation_bev_matches_reference(G, mk):
    from mr_slam_amd import ring
    bev = _pp_bev(G)
    for tag, (dy, dx) in (("m7p11", (-7, 11)), ("p3m20", (3, -20)), ("zero", (0, 0))):
        a = torch.from_numpy(mk.shifted_bev(bev, dy, dx)).to(DEV); b = torch.from_numpy(bev).to(DEV)
        y, x, neg = ring.solve_translation_bev(a, b)
        wy, wx, wneg = G[f"solve_translation_bev_{tag}"]
        assert (int(y), int(x)) == (int(wy), int(wx)) and abs(float(neg) - wneg) < 1e-3 * abs(wneg)


def test_generate_RINGplusplus_matches_reference(G, clouds):
    """The whole RING++ front end (kNN k = 30 + eigenvalues + 13 features + feature BEV + Radon + row FFT) against the
    reference's generate_RINGplusplus (CPU kd-tree / eigvalsh + its kernels).  Feature maps are per-cell maxima of
    float32 features of slightly different eigen-solvers: compared on the cells' values with a relative bound, and
    end to end through the descriptor distance."""
    from mr_slam_amd import ring
    fb, sino, tiring = ring.generate_RINGplusplus(clouds[0], DEV)
    want_bev = _pp_bev(G)
    got = fb.cpu().numpy()
    assert np.array_equal(got != 0, want_bev != 0)                           # same occupied cells in all 6 channels
    nz = want_bev != 0
    rel = np.abs(got[nz] - want_bev[nz]) / np.maximum(np.abs(want_bev[nz]), 1e-3)
    assert np.percentile(rel, 99) < 5e-3 and np.median(rel) < 1e-5
    want_T = G["pp_TIRING_A"]
    assert np.abs(tiring.numpy() - want_T).max() < 5e-3 * np.abs(want_T).max()
    dist, angle = ring.fast_corr_RINGplusplus(tiring.numpy(), want_T, DEV)
    assert int(angle) == 0 and abs(float(dist) - G["fast_corr_pp_AA"][0]) < 2e-4


def test_disco_matches_reference(G, clouds):
    from mr_slam_amd import bev, disco
    xyz, offs = bev.pack_scans(list(clouds[:2]), DEV)
    sig, spec = disco.disco_descriptors(xyz, offs, 40, 120, 20)
    for i, name in enumerate("AB"):
        np.testing.assert_allclose(sig[i].cpu().numpy(), G[f"disco_sig_{name}"], rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(spec[i:i + 1].cpu().numpy(), G[f"disco_fft_{name}"], rtol=0, atol=1e-5)
    ref = {n: torch.from_numpy(G[f"disco_fft_{n}"]).to(DEV) for n in "AB"}
    for a, b in (("A", "B"), ("B", "A"), ("A", "A")):
        yaw, corr = disco.phase_corr(ref[a], ref[b], want_corr=True)
        assert int(yaw[0]) == int(G[f"phase_corr_{a}{b}"][0]) and abs(float(corr.max()) - G[f"phase_corr_{a}{b}"][1]) < 1e-4


def test_point_features_match_reference(G, clouds):
    """N1 kernels vs the reference: kNN sets (kd-tree) equal up to exact distance ties, eigenvalues 1e-4 relative,
    features against the reference kernel's host run on identical neighbours."""
    from mr_slam_amd import pointfeat
    A = clouds[0]
    sel = G["pf_sel"]
    pts = torch.from_numpy(A).to(DEV)
    out = pointfeat.point_features(pts, np.array([0, A.shape[0]], np.int64), 30, want=("knn", "eigens", "features"))
    knn = out["knn"].cpu().numpy()[sel].astype(np.int64)
    want = G["pf_knn"].astype(np.int64)
    d = lambda ii: np.sort(((A[sel][:, None, :].astype(np.float64) - A[ii].astype(np.float64)) ** 2).sum(-1), 1)
    np.testing.assert_allclose(d(knn), d(want), rtol=0, atol=1e-12)       # different members only at equal distance
    same = (np.sort(knn, 1) == np.sort(want, 1)).all(1)
    assert same.mean() > 0.98
    eig = out["eigens"].cpu().numpy()[sel]
    np.testing.assert_allclose(eig[same], G["pf_eigens"][same], rtol=2e-4, atol=1e-9)
    feats = out["features"].cpu().numpy()[sel]
    ref = G["pf_features_kernel"]
    ok = same & np.isfinite(ref).all(1) & np.isfinite(feats).all(1)
    assert ok.mean() > 0.9
    # C O L E P S A X D S2 L2 dZ vZ; D = 3k / (4 pi prod) amplifies the eigenvalue difference threefold
    cols = [0, 1, 2, 3, 4, 5, 6, 7, 9, 10, 11, 12]
    np.testing.assert_allclose(feats[ok][:, cols], ref[ok][:, cols], rtol=2e-3, atol=2e-5)
    np.testing.assert_allclose(feats[ok][:, 8], ref[ok][:, 8], rtol=1e-2)
