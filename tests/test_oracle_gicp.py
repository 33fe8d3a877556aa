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


def _np_align_upstream(src, tgt, k, max_corr, max_iter, rot_eps, trans_eps, guess, oracle):
    """LsqRegistration::computeTransformation / step_lm as upstream writes it, in float64 numpy: ONE neighbour search
    per outer iteration (linearize); every LM trial (compute_error) re-evaluates the residuals of the cached
    correspondences with the cached Mahalanobis matrices."""
    cov_s, _ = _np_covariances(src, k)
    cov_t, _ = _np_covariances(tgt, k)
    tree = cKDTree(tgt)
    x0 = guess.astype(np.float32).astype(np.float64)
    lam, searches, its, conv = -1.0, 0, 0, False
    for it in range(max_iter):
        Tf = x0.astype(np.float32)
        q = (src @ Tf[:3, :3].T + Tf[:3, 3]).astype(np.float32)
        d, j = tree.query(q, k=1); searches += 1
        ok = d.astype(np.float32) ** 2 < max_corr ** 2
        R = x0[:3, :3]
        M = np.linalg.inv(cov_t[j] + R[None] @ cov_s @ R.T[None])[ok]
        a = src.astype(np.float64)[ok]; bpt = tgt[j].astype(np.float64)[ok]

        def err_at(T):
            e = bpt - (a @ T[:3, :3].T + T[:3, 3])
            return np.einsum("na,nab,nb->", e, M, e), e

        y0, e = err_at(x0)
        ta = a @ x0[:3, :3].T + x0[:3, 3]
        J = np.zeros((a.shape[0], 3, 6))
        J[:, 0, 1], J[:, 0, 2] = -ta[:, 2], ta[:, 1]
        J[:, 1, 0], J[:, 1, 2] = ta[:, 2], -ta[:, 0]
        J[:, 2, 0], J[:, 2, 1] = -ta[:, 1], ta[:, 0]
        J[:, :, 3:] = -np.eye(3)[None]
        H = np.einsum("nai,nab,nbj->ij", J, M, J)
        b = np.einsum("nai,nab,nb->i", J, M, e)
        if lam < 0:
            lam = 1e-9 * np.abs(np.diag(H)).max()
        nu, stepped, delta = 2.0, False, None
        for _ in range(10):
            dvec = np.linalg.solve(H + lam * np.eye(6), -b)
            delta = oracle.se3_exp(dvec)
            xi = delta @ x0
            yi, _ = err_at(xi)
            rho = (y0 - yi) / dvec.dot(lam * dvec - b)
            small = max((10 * np.abs(delta[:3, :3] - np.eye(3)) / rot_eps).max(), (10 * np.abs(delta[:3, 3]) / trans_eps).max()) < 1
            if rho < 0:
                if small:
                    stepped = True
                    break
                lam *= nu; nu *= 2
                continue
            x0 = xi
            lam *= max(1.0 / 3.0, 1 - (2 * rho - 1) ** 3)
            stepped = True
            break
        if not stepped:
            break
        its = it + 1
        conv = max((10 * np.abs(delta[:3, :3] - np.eye(3)) / rot_eps).max(), (10 * np.abs(delta[:3, 3]) / trans_eps).max()) < 1
        if conv:
            break
    return x0.astype(np.float32).astype(np.float64), conv, its, searches


def test_lm_follows_upstream_compute_error_semantics(oracle):
    """The restatement searches once per outer iteration and evaluates LM trials on the cached correspondences:
    an independent numpy transcription of upstream's step_lm takes the same number of iterations and lands on
    the same transform."""
    src, tgt, _ = _pair(6, 5000)
    g = oracle.Gicp(k=20, max_corr=5.0)
    g.set_source(src); g.set_target(tgt)
    T, conv, its, trials = g.align()
    assert g.nn_passes == its                      # linearize is the only place that searches
    nT, nconv, nits, nsearch = _np_align_upstream(src, tgt, 20, 5.0, 64, 2e-3, 5e-4, np.eye(4), oracle)
    assert conv and nconv and its == nits and nsearch == nits
    assert np.abs(T - nT).max() < 1e-5


def test_convergence_test_has_upstreams_factor_ten(oracle):
    """is_converged scales |R - I| / rotation_epsilon and |t| / transformation_epsilon by 10 (upstream
    lsq_registration_impl.hpp): with the factor removed the loop stops no later."""
    src, tgt, _ = _pair(7, 5000)
    its = {}
    for f in (10.0, 1.0):
        g = oracle.Gicp(k=20, max_corr=5.0, conv_factor=f)
        g.set_source(src); g.set_target(tgt)
        _, conv, its[f], _ = g.align()
        assert conv
    assert its[1.0] <= its[10.0]


def test_vgicp_voxel_convention_and_cached_correspondences(oracle):
    """Row G7: voxel coordinate = floor(x / res - 0.5) (upstream calc_voxel_coord), one correspondence update per
    outer iteration, near-solution start converges to the GICP answer's neighbourhood."""
    src, tgt, Ttrue = _pair(8, 6000)
    g = oracle.Gicp(k=20, max_corr=1e300, max_iter=50, trans_eps=1e-3)
    g.set_voxel(0.5, 1)
    g.set_source(src); g.set_target(tgt)
    guess = Ttrue.copy(); guess[:3, 3] += [0.05, -0.03, 0.01]
    T, conv, its, _ = g.align(guess)
    assert conv and g.nn_passes == its
    assert np.linalg.norm(T[:3, 3] - Ttrue[:3, 3]) < 1e-2
    # a target that is one point per voxel cell centre: the point at (0.75, 0.75, 0.75) has coordinate 1 at res 0.5
    # under floor(x/res - 0.5) (it would be 1 under floor(x/res) too) while (0.6, 0.6, 0.6) has coordinate 0 (1 under
    # plain floor): a source point at 0.6 must pair with the voxel that also holds 0.3, not the one holding 0.8
    tg = np.array([[0.3, 0.3, 0.3], [0.6, 0.6, 0.6], [0.8, 0.8, 0.8]] * 8, np.float32) + np.random.default_rng(0).normal(0, 1e-3, (24, 3)).astype(np.float32)
    g2 = oracle.Gicp(k=5, max_corr=1e300)
    g2.set_voxel(0.5, 1)
    g2.set_source(np.array([[0.62, 0.62, 0.62]] * 6, np.float32) + np.random.default_rng(1).normal(0, 1e-3, (6, 3)).astype(np.float32))
    g2.set_target(tg)
    e, H, b, _ = g2.linearize(np.eye(4))
    # the voxel [0.25, 0.75)^3 holds the 0.3 and 0.6 families (mean 0.45): residual mean - x = -0.17 per axis, so
    # b = sum J^T M e points along -(-1) * M e: the translation part of b is positive
    assert e > 0 and (b[3:] > 0).all()
