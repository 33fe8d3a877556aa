"""CPU suite: the GICP restatement (parity unpinned in the reference, see oracle/gicp_oracle.cpp)
against (1) an independent float64 numpy/scipy formulation of the same equations and (2) known
SE(3) perturbations of synthetic clouds."""
import numpy as np
import pytest
from scipy.spatial import cKDTree
from scipy.spatial.transform import Rotation as Rot


def _pair(seed, n, rotvec=(0.02, -0.03, 0.08), t=(0.6, -0.4, 0.1), noise=0.01):
    from mr_slam_amd import synth
    rng = np.random.default_rng(seed)
    base = synth.lidar_scan(seed, n, metric=True).astype(np.float64)
    R = Rot.from_rotvec(rotvec).as_matrix()
    src = (base + rng.normal(0, noise, base.shape)).astype(np.float32)
    tgt = (base @ R.T + np.asarray(t) + rng.normal(0, noise, base.shape)).astype(np.float32)
    T = np.eye(4); T[:3, :3] = R; T[:3, 3] = t
    return src, tgt, T


def _np_covariances(pts, k):
    p = pts.astype(np.float64)
    _, idx = cKDTree(pts).query(pts, k=k)
    nb = p[idx]
    d = nb - nb.mean(1, keepdims=True)
    cov = np.einsum("nki,nkj->nij", d, d) / k
    w, v = np.linalg.eigh(cov)
    n = v[:, :, 0]
    return np.eye(3)[None] - 0.999 * n[:, :, None] * n[:, None, :], idx


def _np_linearize(src, tgt, cov_s, cov_t, T, max_corr):
    Tf = T.astype(np.float32)
    q = (src @ Tf[:3, :3].T + Tf[:3, 3]).astype(np.float32)
    d, j = cKDTree(tgt).query(q, k=1)
    ok = d.astype(np.float32) ** 2 < max_corr ** 2
    R = T[:3, :3]
    RCR = cov_t[j] + R[None] @ cov_s @ R.T[None]
    M = np.linalg.inv(RCR)
    ta = src.astype(np.float64) @ R.T + T[:3, 3]
    e = tgt[j].astype(np.float64) - ta
    J = np.zeros((src.shape[0], 3, 6))
    J[:, 0, 1], J[:, 0, 2] = -ta[:, 2], ta[:, 1]
    J[:, 1, 0], J[:, 1, 2] = ta[:, 2], -ta[:, 0]
    J[:, 2, 0], J[:, 2, 1] = -ta[:, 1], ta[:, 0]
    J[:, :, 3:] = -np.eye(3)[None]
    J, M, e = J[ok], M[ok], e[ok]
    H = np.einsum("nai,nab,nbj->ij", J, M, J)
    b = np.einsum("nai,nab,nb->i", J, M, e)
    err = np.einsum("na,nab,nb->", e, M, e)
    return err, H, b, np.where(ok, j, -1)


def test_covariances_and_linearisation_match_numpy(oracle):
    src, tgt, Ttrue = _pair(1, 6000)
    g = oracle.Gicp(k=20, max_corr=5.0)
    g.set_source(src); g.set_target(tgt)
    cs, ct = g.covariances(0), g.covariances(1)
    ns, idx = _np_covariances(src, 20)
    nt, _ = _np_covariances(tgt, 20)
    np.testing.assert_array_equal(np.sort(oracle.knn(src, 20), 1), np.sort(idx, 1))
    assert np.abs(cs - ns).max() < 1e-6 and np.abs(ct - nt).max() < 1e-6
    T = Ttrue.copy(); T[:3, 3] += [0.2, -0.1, 0.05]
    e, H, b, corr = g.linearize(T)
    ne, nH, nb, ncorr = _np_linearize(src, tgt, ns, nt, T, 5.0)
    assert (corr == ncorr).mean() > 0.999
    assert abs(e - ne) < 1e-3 * abs(ne)
    np.testing.assert_allclose(H, nH, rtol=1e-3, atol=1e-3 * np.abs(nH).max())
    np.testing.assert_allclose(b, nb, rtol=1e-3, atol=1e-3 * np.abs(nb).max())


@pytest.mark.parametrize("seed,rotvec,t", [(2, (0.02, -0.03, 0.08), (0.6, -0.4, 0.1)),
                                           (3, (0.0, 0.0, -0.05), (-0.8, 0.3, 0.0)),
                                           (4, (0.01, 0.01, 0.0), (0.1, 0.1, -0.05))])
def test_recovers_known_transform(oracle, seed, rotvec, t):
    src, tgt, Ttrue = _pair(seed, 12000, rotvec, t)
    g = oracle.Gicp(k=20, max_corr=5.0)
    g.set_source(src); g.set_target(tgt)
    T, conv, its, trials = g.align()
    assert conv and its <= 12
    dt = np.linalg.norm(T[:3, 3] - Ttrue[:3, 3])
    dr = np.linalg.norm(Rot.from_matrix(T[:3, :3] @ Ttrue[:3, :3].T).as_rotvec())
    assert dt < 2e-3 and dr < 3e-4          # the 1 cm noise floor, not an implementation bound
    assert g.fitness(T, 1.0) < 2e-3


def test_se3_exp_matches_scipy(oracle):
    from scipy.linalg import expm
    rng = np.random.default_rng(0)
    for a in [np.zeros(6), np.array([1e-7, 0, 0, 1, 2, 3.0]), rng.normal(size=6), 0.01 * rng.normal(size=6)]:
        w, v = a[:3], a[3:]
        X = np.zeros((4, 4))
        X[:3, :3] = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
        X[:3, 3] = v
        np.testing.assert_allclose(oracle.se3_exp(a), expm(X), atol=1e-12)


def test_mapping_side_parameters(oracle):
    """global_manager.cpp:2437-2442: k = 15, transEps 1e-3, maxCorrDist 100."""
    src, tgt, Ttrue = _pair(5, 8000)
    g = oracle.Gicp(k=15, max_corr=100.0, max_iter=50, trans_eps=1e-3)
    g.set_source(src); g.set_target(tgt)
    T, conv, its, _ = g.align(np.eye(4))
    assert conv
    assert np.linalg.norm(T[:3, 3] - Ttrue[:3, 3]) < 3e-3
