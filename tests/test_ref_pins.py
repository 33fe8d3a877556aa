"""CPU suite: pins the Python restatements (oracle/corr_oracle.py, oracle/pointfeat_oracle.py) to output of the
REFERENCE's own code run in the build container (tests/golden/ref_corr.npz, made by tests/golden/make_golden_ref_corr.py
from RING_ros/util.py, disco_ros/models/DiSCO.py, disco_ros/main.py:phase_corr and the numpy twin of the feature
kernel).  SURVEY.md section 8(a) rows R2, C1-C4, D1-D2, N1."""
import importlib.util
import os

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))


def _f(v):
    return float(np.ravel(v)[0])


def _mk():
    spec = importlib.util.spec_from_file_location("make_golden_ref_corr", os.path.join(HERE, "golden", "make_golden_ref_corr.py"))
    m = importlib.util.module_from_spec(spec)
    import sys
    sys.path.insert(0, os.path.join(HERE, "golden"))
    try:
        spec.loader.exec_module(m)
    finally:
        sys.path.remove(os.path.join(HERE, "golden"))
    return m


@pytest.fixture(scope="module")
def G():
    return np.load(os.path.join(HERE, "golden", "ref_corr.npz"))


@pytest.fixture(scope="module")
def mk():
    return _mk()


@pytest.fixture(scope="module")
def clouds(mk, G):
    A, B, C = mk.inputs()
    assert A.shape[0] == int(G["n_A"][0]) and B.shape[0] == int(G["n_B"][0])
    assert B.astype(np.float64).sum() == float(G["sum_B"][0])         # the generator's inputs, bit for bit
    return A, B, C


def _ring_chain(oracle, pc):
    """generate_RING's front half on the checkers: Cartesian BEV -> Radon (util.py:177-195)."""
    soa = np.ascontiguousarray(pc[:, :3].T).reshape(-1)
    bev = oracle.bev_cart(soa, 1, 1, 120, 120, 1).reshape(-1, 3)[:, 2].reshape(1, 120, 120)
    ang = np.linspace(0, 2 * np.pi, 120).astype(np.float32)
    return bev, oracle.radon_parallel(bev, ang, 120, 1.0)


def test_tiring_matches_reference_generate_RING(oracle, G, clouds):
    from oracle import corr_oracle as K
    for name, pc in zip("AB", clouds[:2]):
        bev, sino = _ring_chain(oracle, pc)
        occ = np.flatnonzero(bev.reshape(-1))
        np.testing.assert_array_equal(occ, G[f"ring_bev_occ_{name}"])
        np.testing.assert_array_equal(bev.reshape(-1)[occ], G[f"ring_bev_val_{name}"])
        np.testing.assert_array_equal(sino, G[f"ring_RING_{name}"])
        t = K.tiring_from_sinogram(sino).numpy()
        np.testing.assert_allclose(t, G[f"ring_TIRING_{name}"], rtol=0, atol=2e-5)


def test_fast_corr_matches_reference(oracle, G, clouds):
    from oracle import corr_oracle as K
    T = {n: torch.from_numpy(G[f"ring_TIRING_{n}"]) for n in "AB"}
    T["C"] = K.tiring_from_sinogram(_ring_chain(oracle, clouds[2])[1])
    for a, b in (("A", "B"), ("A", "C"), ("B", "C"), ("A", "A"), ("B", "A")):
        dist, angle, _ = K.fast_corr(T[a], T[b])
        wd, wa = G[f"fast_corr_{a}{b}"]
        assert angle == int(wa) and abs(float(dist) - wd) < 2e-6, (a, b, dist, angle, wd, wa)
    dist, angle, _ = K.fast_corr(torch.cat([T["A"], T["C"]]), torch.cat([T["B"], T["C"]]))
    wd, wa = G["fast_corr_2ch"]
    assert angle == int(wa) and abs(float(dist) - wd) < 2e-6      # C = 2: no channel factor in the denominator (util.py:369)


def _loop_AB(G):
    """main_RING.py:147-178 inputs as the generator built them."""
    angle, row_shift, row_shift_extra, rad, rad_extra = G["loop_AB"]
    RA = torch.from_numpy(G["ring_RING_A"]); RB = torch.from_numpy(G["ring_RING_B"])
    return RA, torch.roll(RB, int(row_shift), dims=1), torch.roll(RB, int(row_shift_extra), dims=1), float(rad), float(rad_extra)


def test_solve_translation_matches_reference_literal_and_pinv(G):
    from oracle import corr_oracle as K
    RA, RBs, RBe, rad, rad_e = _loop_AB(G)
    for i, (pos, r, tag) in enumerate(((RBs, rad, ""), (RBe, rad_e, "_extra"))):
        x, y, err, sh = K.solve_translation(RA, pos, r)            # default = the reference's call (method='svd')
        np.testing.assert_array_equal(sh, G["solve_translation_b"][i])                        # the 120 integer row shifts
        want = G[f"solve_translation_AB{tag}"]
        # the literal v.t() product of util.py:488-506 (CPU LAPACK): same library here -> same result
        np.testing.assert_allclose([_f(x), _f(y), _f(err)], want, rtol=2e-4, atol=2e-4)
        x, y, err, _ = K.solve_translation(RA, pos, r, literal=False)
        np.testing.assert_allclose([_f(x), _f(y)], G[f"solve_translation_AB{tag}_pinv"], rtol=1e-4, atol=1e-4)
    # for the record: the reference's result is its own least-squares solution turned by an orthogonal matrix
    # (same norm, larger residual); the drop-in returns the reference's numbers by default
    lit, pinv = G["solve_translation_AB"][:2], G["solve_translation_AB_pinv"]
    assert abs(np.linalg.norm(lit) - np.linalg.norm(pinv)) < 1e-3 and np.linalg.norm(lit - pinv) > 1.0


def test_ringplusplus_functions_match_reference(G, mk):
    from oracle import corr_oracle as K
    RING = G["pp_RING_A"]; TIRING = G["pp_TIRING_A"]
    out, _ = K.forward_row_fft(RING)
    np.testing.assert_allclose(out.numpy(), TIRING, rtol=0, atol=1e-5 * np.abs(TIRING).max())
    Tb = mk.rolled_noisy(TIRING, 23, seed=5)
    for tag, (a, b) in (("AA", (TIRING, TIRING)), ("Arolled", (TIRING, Tb)), ("rolledA", (Tb, TIRING))):
        dist, angle, _ = K.fast_corr_ringplusplus(a, b)
        wd, wa = G[f"fast_corr_pp_{tag}"]
        assert angle == int(wa) and abs(float(dist) - wd) < 2e-6, (tag, dist, angle)


def _pp_bev(G):
    bev = np.zeros(6 * 120 * 120, np.float32)
    bev[G["pp_bev_nz_A"]] = G["pp_bev_val_A"]
    return bev.reshape(6, 120, 120)


def test_solve_translation_bev_matches_reference(G, mk):
    from oracle import corr_oracle as K
    bev = _pp_bev(G)
    for tag, (dy, dx) in (("m7p11", (-7, 11)), ("p3m20", (3, -20)), ("zero", (0, 0))):
        y, x, neg, _ = K.solve_translation_bev(mk.shifted_bev(bev, dy, dx), bev)
        wy, wx, wneg = G[f"solve_translation_bev_{tag}"]
        assert (y, x) == (int(wy), int(wx)) and abs(neg - wneg) < 1e-3 * abs(wneg)


def test_disco_matches_reference(oracle, G, clouds):
    from oracle import corr_oracle as K
    spec = {}
    for name, pc in zip("AB", clouds[:2]):
        soa = np.ascontiguousarray(pc.T).reshape(-1)
        bev = oracle.bev_polar(soa, 1, 1, 40, 120, 20, 1).reshape(-1, 3)[:, 2].reshape(1, 20, 40, 120)
        sig, sp = K.disco_forward(bev)
        np.testing.assert_allclose(sig[0], G[f"disco_sig_{name}"], rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(sp.numpy(), G[f"disco_fft_{name}"], rtol=0, atol=1e-5)
        spec[name] = torch.from_numpy(G[f"disco_fft_{name}"])
    for a, b in (("A", "B"), ("B", "A"), ("A", "A")):
        yaw, corr = K.phase_corr(spec[a], spec[b])
        assert yaw == int(G[f"phase_corr_{a}{b}"][0]) and abs(corr.max() - G[f"phase_corr_{a}{b}"][1]) < 1e-4


def test_point_features_match_reference(G, clouds):
    """N1: util.build_neighbors_NN (sklearn kd-tree + torch eigvalsh), the reference feature kernel run on the host
    (kernel.cu:16-104) and its numpy twin (test.py:68-98)."""
    from oracle import pointfeat_oracle as PF
    A = clouds[0]
    sel = G["pf_sel"]
    idx = PF.knn_indices(A, 30)
    want_idx = G["pf_knn"].astype(np.int64)
    same = (np.sort(idx[sel], 1) == np.sort(want_idx, 1)).all(1)
    # where the sets differ the k-th distances tie exactly (kd-tree order of equidistant points)
    d = lambda ii: np.sort(((A[sel][:, None, :].astype(np.float64) - A[ii].astype(np.float64)) ** 2).sum(-1), 1)
    np.testing.assert_allclose(d(idx[sel]), d(want_idx), rtol=0, atol=1e-12)
    assert same.mean() > 0.98
    eig = PF.covariation_eigenvalue(A, idx)
    np.testing.assert_allclose(eig[sel][same], G["pf_eigens"][same], rtol=1e-4, atol=1e-9)
    feats = PF.calculate_features(A, want_idx_full(idx, sel, want_idx), eig_full(eig, sel, G["pf_eigens"]))[sel]
    ref = G["pf_features_kernel"]
    ok = np.isfinite(ref).all(1) & np.isfinite(feats).all(1)
    assert ok.mean() > 0.95
    np.testing.assert_allclose(feats[ok], ref[ok], rtol=2e-4, atol=1e-6)
    # the numpy twin agrees with the kernel to float32 rounding where its own (np.linalg.eig) eigenvalues agree
    tw = G["pf_features_numpy_twin"]
    close = np.isfinite(tw).all(1) & ok & (np.abs(G["pf_twin_eigens"] - G["pf_eigens"]).max(1) < 1e-7)
    assert close.mean() > 0.5
    # twin order: C O L E P S A X D S2 L2 dZ vZ -- D (density) amplifies eigenvalue noise (1 / product): looser
    cols = [0, 1, 2, 3, 4, 5, 6, 7, 9, 10, 11, 12]
    np.testing.assert_allclose(ref[close][:, cols], tw[close][:, cols], rtol=5e-3, atol=1e-5)


def want_idx_full(idx, sel, want_sel):
    out = idx.copy()
    out[sel] = want_sel
    return out


def eig_full(eig, sel, want_sel):
    out = eig.copy()
    out[sel] = want_sel
    return out
