"""GPU parity suite for the batched GICP (HIP through the C ABI) vs the CPU restatement.
Tolerance from BASELINE.json north_star: transforms within 1e-4 m / 1e-4 rad."""
import numpy as np
import pytest
from scipy.spatial.transform import Rotation as Rot

pytestmark = pytest.mark.gpu

TOL_T = 1e-4   # metres
TOL_R = 1e-4   # radians


@pytest.fixture(scope="module")
def dev():
    import torch
    assert torch.cuda.is_available()
    from mr_slam_amd import _lib
    _lib.load()
    return "cuda:0"


def _pair(seed, n, rotvec=(0.02, -0.03, 0.08), t=(0.6, -0.4, 0.1), noise=0.01):
    from mr_slam_amd import synth
    rng = np.random.default_rng(seed)
    base = synth.lidar_scan(seed, n, metric=True).astype(np.float64)
    R = Rot.from_rotvec(rotvec).as_matrix()
    src = (base + rng.normal(0, noise, base.shape)).astype(np.float32)
    tgt = (base @ R.T + np.asarray(t) + rng.normal(0, noise, base.shape)).astype(np.float32)
    T = np.eye(4); T[:3, :3] = R; T[:3, 3] = t
    return src, tgt, T


def _pose_err(A, B):
    dt = np.linalg.norm(A[:3, 3] - B[:3, 3])
    dr = np.linalg.norm(Rot.from_matrix(A[:3, :3] @ B[:3, :3].T).as_rotvec())
    return dt, dr


def test_knn_and_covariances_match_oracle(dev, oracle):
    from mr_slam_amd import gicp
    src, tgt, _ = _pair(1, 7000)
    for k in (20, 15):
        b = gicp.GicpBatch(2)
        b.set_params(k_correspondences=k)
        b.set_sources([src, tgt[:3001]])
        knn = b.compute_covariances(0, want_knn=True).cpu().numpy()
        want = np.concatenate([oracle.knn(src, k), oracle.knn(tgt[:3001], k)])
        same = (np.sort(knn, 1) == np.sort(want, 1)).all(1)
        # exactness: wherever the index sets differ the DISTANCES are identical (exact float ties), i.e. the multiset of
        # the k smallest squared distances (the search's own float operation chain) agrees for every point
        I = np.eye(4)
        for cloud, sl in ((src, slice(0, src.shape[0])), (tgt[:3001], slice(src.shape[0], None))):
            dg = np.stack([oracle.pair_d2(cloud, I, cloud, knn[sl][:, j]) for j in range(k)], 1)
            dw = np.stack([oracle.pair_d2(cloud, I, cloud, want[sl][:, j]) for j in range(k)], 1)
            assert np.array_equal(np.sort(dg, 1), np.sort(dw, 1))
            assert (np.diff(dg, axis=1) >= 0).all()            # ascending distance, as the header promises
        assert same.mean() > 0.99                                # ties are rare on noisy clouds
        g = oracle.Gicp(k=k)
        g.set_source(src); g.set_target(tgt[:3001])
        want_cov = np.concatenate([g.covariances(0), g.covariances(1)])
        got = b.covariances(0)
        # identical neighbour sets -> identical covariances (to rounding of the 3x3 eigenvector); a point whose set
        # differs by an exact tie legitimately has another (equally valid) covariance
        assert np.abs(got - want_cov).reshape(-1, 9).max(1)[same].max() < 1e-9


def test_linearize_matches_oracle(dev, oracle):
    from mr_slam_amd import gicp
    src, tgt, Ttrue = _pair(2, 9000)
    T = Ttrue.copy(); T[:3, 3] += [0.25, -0.1, 0.05]
    for max_corr in (5.0, 0.5):
        b = gicp.GicpBatch(1)
        b.set_params(max_correspondence_distance=max_corr)
        b.set_sources([src]); b.set_targets([tgt])
        e, H, bb, corr = b.linearize(T[None], want_corr=True)
        g = oracle.Gicp(k=20, max_corr=max_corr)
        g.set_source(src); g.set_target(tgt)
        we, wH, wb, wcorr = g.linearize(T)
        # exact search: different indices only where the two candidates are at exactly the same float distance
        assert np.array_equal(oracle.pair_d2(src, T, tgt, corr), oracle.pair_d2(src, T, tgt, wcorr))
        assert np.array_equal(corr >= 0, wcorr >= 0)
        assert (corr == wcorr).mean() > 0.999
        assert (wcorr >= 0).sum() > 100
        assert abs(e[0] - we) < 2e-3 * abs(we)
        np.testing.assert_allclose(H[0], wH, rtol=2e-3, atol=2e-3 * np.abs(wH).max())
        np.testing.assert_allclose(bb[0], wb, rtol=2e-3, atol=2e-3 * np.abs(wb).max())
        if (corr == wcorr).all():
            assert abs(e[0] - we) < 1e-9 * abs(we)
            np.testing.assert_allclose(H[0], wH, rtol=1e-9, atol=1e-9 * np.abs(wH).max())


def test_align_batch_within_north_star_tolerance(dev, oracle):
    """Ragged batch of three pairs; every transform within 1e-4 m / 1e-4 rad of the restatement."""
    from mr_slam_amd import gicp
    cfgs = [(3, 9000, (0.02, -0.03, 0.08), (0.6, -0.4, 0.1)),
            (4, 6001, (0.0, 0.0, -0.05), (-0.8, 0.3, 0.0)),
            (5, 12000, (0.01, 0.01, 0.0), (0.1, 0.1, -0.05))]
    pairs = [_pair(s, n, r, t) for (s, n, r, t) in cfgs]
    b = gicp.GicpBatch(3)
    b.set_params(max_correspondence_distance=5.0)
    b.set_sources([p[0] for p in pairs]); b.set_targets([p[1] for p in pairs])
    guess = np.stack([np.eye(4)] * 3)
    guess[1, :3, 3] = [-0.5, 0.2, 0.0]
    T, conv, its = b.align(guess)
    fit = b.fitness(T, 1.0)
    for i, (src, tgt, Ttrue) in enumerate(pairs):
        g = oracle.Gicp(k=20, max_corr=5.0)
        g.set_source(src); g.set_target(tgt)
        wT, wconv, wits, _ = g.align(guess[i])
        dt, dr = _pose_err(T[i], wT)
        assert dt < TOL_T and dr < TOL_R, (i, dt, dr)
        assert conv[i] == wconv and abs(int(its[i]) - wits) <= 1
        assert abs(fit[i] - g.fitness(wT, 1.0)) < 1e-5
        gt, gr = _pose_err(T[i], Ttrue)
        assert gt < 3e-3 and gr < 5e-4


def test_mapping_side_configuration(dev, oracle):
    """global_manager.cpp:2437-2442: k = 15, transEps 1e-3, maxIter 50, maxCorrDist 100."""
    from mr_slam_amd import gicp
    src, tgt, _ = _pair(6, 8000)
    b = gicp.GicpBatch(1)
    b.set_params(k_correspondences=15, max_correspondence_distance=100.0, max_iterations=50,
                 transformation_epsilon=1e-3)
    b.set_sources([src]); b.set_targets([tgt])
    T, conv, its = b.align()
    g = oracle.Gicp(k=15, max_corr=100.0, max_iter=50, trans_eps=1e-3)
    g.set_source(src); g.set_target(tgt)
    wT, wconv, wits, _ = g.align()
    dt, dr = _pose_err(T[0], wT)
    assert dt < TOL_T and dr < TOL_R and conv[0] == wconv
    # upstream semantics: linearize is the only step that searches -> one NN pass per outer iteration, none per LM trial
    assert b.nn_passes == its[0] == wits == g.nn_passes


def test_pygicp_drop_in(dev, oracle):
    """The call sequence of main_RING.py:81-104."""
    from mr_slam_amd.compat import pygicp
    src, tgt, Ttrue = _pair(7, 30000, (0.0, 0.0, 0.1), (1.0, 0.5, 0.0))
    source = pygicp.downsample(src.astype(np.float64), 0.2)
    target = pygicp.downsample(tgt.astype(np.float64), 0.2)
    assert 1000 < source.shape[0] < src.shape[0]
    gicp = pygicp.FastGICP()
    gicp.set_input_target(target)
    gicp.set_input_source(source)
    gicp.set_num_threads(4)
    gicp.set_max_correspondence_distance(5.0)
    T = gicp.align(initial_guess=np.eye(4))
    fitness = gicp.get_fitness_score(1.0)
    assert np.array_equal(T, gicp.get_final_transformation())
    g = oracle.Gicp(k=20, max_corr=5.0)
    g.set_source(source); g.set_target(target)
    wT, _, _, _ = g.align()
    dt, dr = _pose_err(T, wT)
    assert dt < TOL_T and dr < TOL_R
    assert abs(fitness - g.fitness(wT, 1.0)) < 1e-5
    assert _pose_err(T, Ttrue)[0] < 0.05


def test_windowed_lm_schedule_equals_the_per_tick_schedule_bit_for_bit(dev):
    """Round 6: batches of <= 8 pairs (one registration at a time is how the nodes call fast_gicp, main_RING.py:81-104,
    global_manager.cpp:2016-2021) run the LM schedule in windows of 4 ticks with no host round trip in between and compute the two clouds'
    covariances side by side on two streams.  Same kernels on the same device-side state: transforms, iteration counts, convergence flags,
    Hessians and the number of nearest-neighbour passes must equal the per-tick schedule (development switch MRS_GICP_WINDOW=0) bit for bit --
    one pair, three pairs that converge at different ticks, forced iterations, and a window that is not a divisor of the tick count."""
    import os
    from mr_slam_amd import gicp
    pairs = [_pair(21 + i, 9000 + 2500 * i, (0.01 * i, -0.02, 0.05 + 0.02 * i), (0.4 + 0.2 * i, -0.3, 0.05)) for i in range(3)]

    def run(n, env, **prm):
        old = {k: os.environ.get(k) for k in ("MRS_DEV", "MRS_GICP_WINDOW", "MRS_GICP_SERIAL_COV")}
        os.environ.update(env)
        try:
            b = gicp.GicpBatch(n)
            b.set_params(max_correspondence_distance=5.0, **prm)
            b.set_sources([p[0] for p in pairs[:n]]); b.set_targets([p[1] for p in pairs[:n]])
            T, conv, its = b.align()
            return T.copy(), conv.copy(), its.copy(), b.nn_passes, b.hessian.copy()
        finally:
            for k, v in old.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
    per_tick = {"MRS_DEV": "1", "MRS_GICP_WINDOW": "0", "MRS_GICP_SERIAL_COV": "1"}
    for n, prm in ((1, {}), (3, {}), (2, {"force_iterations": 5}), (1, {"k_correspondences": 15})):
        want = run(n, per_tick, **prm)
        for env in ({}, {"MRS_DEV": "1", "MRS_GICP_WINDOW": "3"}, {"MRS_DEV": "1", "MRS_GICP_WINDOW": "16"}):
            got = run(n, env, **prm)
            assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1]) and np.array_equal(got[2], want[2]), (n, prm, env)
            assert got[3] == want[3], (got[3], want[3])
            if want[4] is not None:
                assert np.array_equal(got[4], want[4])
        if not prm:
            assert want[1].all() and _pose_err(want[0][0], pairs[0][2])[0] < 0.05


def test_force_iterations_and_failure_modes(dev):
    from mr_slam_amd import gicp, _lib
    src, tgt, _ = _pair(8, 5000)
    b = gicp.GicpBatch(1)
    b.set_params(force_iterations=7)
    b.set_sources([src]); b.set_targets([tgt])
    T, conv, its = b.align()
    assert its[0] == 7 and not conv[0]
    with pytest.raises(_lib.MrsError):
        b.set_params(k_correspondences=100)
    with pytest.raises(_lib.MrsError):
        gicp.GicpBatch(1).align()               # no clouds set


def test_full_size_pair(dev, oracle):
    """BASELINE size (120k x 120k): recovers a known transform to the noise floor, the fitness improves, and the
    result is within the north_star tolerance of the restatement at full size too."""
    from mr_slam_amd import gicp
    src, tgt, Ttrue = _pair(9, 120000, (0.01, -0.02, 0.05), (0.5, -0.3, 0.05), noise=0.02)
    b = gicp.GicpBatch(1)
    b.set_params(max_correspondence_distance=5.0)
    b.set_sources([src]); b.set_targets([tgt])
    f0 = b.fitness(np.eye(4)[None], 1.0)[0]
    T, conv, its = b.align()
    f1 = b.fitness(T, 1.0)[0]
    dt, dr = _pose_err(T[0], Ttrue)
    assert conv[0] and dt < 5e-3 and dr < 5e-4 and f1 < f0
    g = oracle.Gicp(k=20, max_corr=5.0)
    g.set_source(src); g.set_target(tgt)
    wT, wconv, wits, _ = g.align()
    dt, dr = _pose_err(T[0], wT)
    assert wconv and dt < TOL_T and dr < TOL_R, (dt, dr, its, wits)
    assert abs(f1 - g.fitness(wT, 1.0)) < 1e-6


@pytest.mark.parametrize("res,nb", [(0.5, 1), (0.5, 7), (1.0, 27)])
def test_vgicp_matches_restatement(dev, oracle, res, nb):
    """Row G7 (FastVGICP / FastVGICPCuda as configured at global_manager.cpp:2445-2455): voxelised target,
    DIRECT1/7/27 correspondences; started near the solution like ICPCheck does."""
    from mr_slam_amd import gicp
    src, tgt, Ttrue = _pair(11, 15000)
    guess = Ttrue.copy()
    guess[:3, 3] += [0.08, -0.05, 0.02]
    guess[:3, :3] = Rot.from_rotvec([0, 0, 0.01]).as_matrix() @ guess[:3, :3]
    b = gicp.GicpBatch(1)
    b.set_params(k_correspondences=15, max_iterations=50, transformation_epsilon=1e-3, voxel_resolution=res, voxel_neighbors=nb)
    b.set_sources([src]); b.set_targets([tgt])
    e, H, bb, _ = b.linearize(guess[None])
    g = oracle.Gicp(k=15, max_corr=1e300, max_iter=50, trans_eps=1e-3)
    g.set_voxel(res, nb)
    g.set_source(src); g.set_target(tgt)
    we, wH, wb, _ = g.linearize(guess)
    assert abs(e[0] - we) < 1e-6 * abs(we)
    np.testing.assert_allclose(H[0], wH, rtol=1e-6, atol=1e-6 * np.abs(wH).max())
    np.testing.assert_allclose(bb[0], wb, rtol=1e-6, atol=1e-6 * np.abs(wb).max())
    T, conv, its = b.align(guess[None])
    wT, wconv, wits, _ = g.align(guess)
    dt, dr = _pose_err(T[0], wT)
    assert dt < TOL_T and dr < TOL_R and conv[0] == wconv
    assert _pose_err(T[0], Ttrue)[0] < 1e-2


def test_degenerate_clouds_terminate_with_finite_results(dev):
    """Inputs the reference never guards against (fewer points than k, one target point, identical / collinear /
    duplicated points, NaNs, no correspondence in range): every call returns, transforms stay finite, and an
    empty correspondence set scores like PCL's getFitnessScore (max double)."""
    from mr_slam_amd import gicp
    rng = np.random.default_rng(0)
    plane = np.c_[rng.uniform(-5, 5, (3000, 2)), np.zeros(3000)]
    line = np.c_[np.linspace(0, 10, 400), np.zeros(400), np.zeros(400)]
    cases = [
        (rng.normal(size=(5, 3)), rng.normal(size=(500, 3)), {}),
        (rng.normal(size=(500, 3)), rng.normal(size=(1, 3)), {}),
        (np.ones((300, 3)), np.ones((300, 3)), {}),
        (line, line + [0.1, 0, 0], {}),
        (plane, plane + [0.05, 0.02, 0], {}),
        (rng.normal(size=(500, 3)), rng.normal(size=(500, 3)) + 1000, {"max_correspondence_distance": 1.0}),
        (np.r_[rng.normal(size=(400, 3)), [[np.nan, 0, 0]]], rng.normal(size=(400, 3)), {}),
        (np.repeat(rng.normal(size=(50, 3)), 20, 0), np.repeat(rng.normal(size=(50, 3)), 20, 0), {}),
    ]
    for i, (src, tgt, kw) in enumerate(cases):
        b = gicp.GicpBatch(1)
        b.set_params(**kw)
        b.set_sources([src.astype(np.float32)]); b.set_targets([tgt.astype(np.float32)])
        T, conv, its = b.align()
        assert np.isfinite(T).all(), i
        f = b.fitness(T, 1.0)[0]
        assert f >= 0 and not np.isnan(f), i
        if i == 5:
            assert not conv[0] and f > 1e300
    # fewer points than k: every point's neighbour list is exactly the cloud (itself first), padded with -1 -- nothing from beyond the cloud
    tiny = rng.normal(size=(5, 3)).astype(np.float32)
    b = gicp.GicpBatch(2)
    b.set_params(k_correspondences=15)
    b.set_sources([tiny, rng.normal(size=(700, 3)).astype(np.float32)])
    knn = b.compute_covariances(0, want_knn=True).cpu().numpy()[:5]
    assert (knn[:, 0] == np.arange(5)).all() and (np.sort(knn[:, :5], 1) == np.arange(5)).all() and (knn[:, 5:] == -1).all()


def test_cloud_beyond_the_ordered_tile_window(dev, oracle):
    """More than 512 tiles of 1024 points per cloud: the tiles past the distance-ordered window are still visited
    (neighbours stay exact)."""
    from mr_slam_amd import gicp, synth
    rng = np.random.default_rng(5)
    base = np.concatenate([synth.lidar_scan(60 + i, 120000, metric=True) + [0.0, 0.0, 0.001 * i] for i in range(5)]).astype(np.float64)
    base = base[rng.permutation(base.shape[0])[:560000]]
    R = Rot.from_rotvec([0.004, -0.003, 0.01]).as_matrix()
    src = (base + rng.normal(0, 0.01, base.shape)).astype(np.float32)
    tgt = (base @ R.T + [0.15, -0.1, 0.02] + rng.normal(0, 0.01, base.shape)).astype(np.float32)
    b = gicp.GicpBatch(1)
    b.set_params(max_correspondence_distance=2.0)
    b.set_sources([src]); b.set_targets([tgt])
    T = np.eye(4)
    e, H, bb, corr = b.linearize(T[None], want_corr=True)
    g = oracle.Gicp(k=20, max_corr=2.0)
    g.set_source(src); g.set_target(tgt)
    we, wH, wb, wcorr = g.linearize(T)
    assert np.array_equal(oracle.pair_d2(src, T, tgt, corr), oracle.pair_d2(src, T, tgt, wcorr))   # exact up to true ties
    assert (corr == wcorr).mean() > 0.999 and (wcorr >= 0).mean() > 0.9
    assert abs(e[0] - we) < 2e-3 * abs(we)


def test_timed_protocol_batch_matches_restatement(dev, oracle):
    """The protocol bench.py times (BASELINE configs[2]): 256 pairs x 120k x 120k points in ONE batch, force_iterations = 20, k = 15,
    max_correspondence_distance 5.0.  Eight pairs sampled out of that batch against the restatement run with the same forced
    iteration count: within the north_star tolerance, and every pair did exactly 20 iterations with one NN pass each."""
    import bench
    from mr_slam_amd import gicp
    n_pairs, iters = 256, 20
    srcs, tgts = bench._gicp_pairs(n_pairs, 0)
    b = gicp.GicpBatch(n_pairs)
    b.set_params(k_correspondences=15, max_correspondence_distance=5.0, force_iterations=iters)
    b.set_sources(srcs); b.set_targets(tgts)
    T, conv, its = b.align()
    assert (its == iters).all() and b.nn_passes == iters
    rng = np.random.default_rng(4)
    worst = (0.0, 0.0)
    for i in sorted(rng.choice(n_pairs, 8, replace=False)):
        g = oracle.Gicp(k=15, max_corr=5.0, threads=16)
        g.set_source(srcs[i]); g.set_target(tgts[i])
        wT, _, wits, _ = g.align(np.eye(4), force_iters=iters)
        dt, dr = _pose_err(T[i], wT)
        assert wits == iters and dt < TOL_T and dr < TOL_R, (i, dt, dr)
        worst = (max(worst[0], dt), max(worst[1], dr))
    print("timed-protocol GICP parity, worst of 8 pairs: %.2e m, %.2e rad" % worst)


def _clouds_for_search_tests():
    from mr_slam_amd import synth
    rng = np.random.default_rng(11)
    lidar = synth.lidar_scan(7, 30000, metric=True)
    uni = rng.uniform(-20, 20, size=(6000, 3)).astype(np.float32)
    lattice = np.stack(np.meshgrid(np.arange(24), np.arange(24), np.arange(12), indexing="ij"), -1).reshape(-1, 3).astype(np.float32) * 0.25
    lattice = lattice[rng.permutation(lattice.shape[0])]
    # 60 copies of one point right next to a handful of distinct points: more exact ties at the k-th distance than any list has room for
    cluster = np.concatenate([np.repeat(np.array([[1.0, 2.0, 0.5]], np.float32), 60, 0), rng.normal(0, 0.3, (300, 3)).astype(np.float32) + [1.0, 2.0, 0.5],
                              np.repeat(np.array([[1.05, 2.0, 0.5]], np.float32), 45, 0)])
    cluster = cluster[rng.permutation(cluster.shape[0])]
    same = np.repeat(np.array([[3.0, -1.0, 2.0]], np.float32), 200, 0)           # one cell of the finest level holds everything
    return {"lidar": lidar, "uniform": uni, "lattice": lattice, "cluster": cluster, "same": same, "seventeen": uni[:17], "one": uni[:1]}


def _knn_d2(oracle, cloud, knn):
    I = np.eye(4)
    return np.stack([oracle.pair_d2(cloud, I, cloud, knn[:, j]) for j in range(knn.shape[1])], 1)


@pytest.mark.parametrize("k", [15, 20, 30])
def test_knn_exact_on_adversarial_clouds(dev, oracle, k):
    """Both k-NN kernels (round 4: octree-cell leaves, per-query culling; round 3: the default) against the restatement's exact kd-tree
    search: same multiset of float distances, ascending, every index valid and distinct -- on a lidar scan, uniform points, a lattice
    (every distance tied many times over), clusters of duplicates larger than k and clouds smaller than a leaf / than k."""
    from mr_slam_amd import gicp
    for name, cloud in _clouds_for_search_tests().items():
        n = cloud.shape[0]
        res = {}
        for core in (3, 0):          # 3: the round-4 k-NN kernel (octree-cell leaves), 0: the round-3 one (the default for k-NN)
            b = gicp.GicpBatch(1)
            b.set_search(core)
            b.set_params(k_correspondences=k)
            b.set_sources([cloud])
            res[core] = b.compute_covariances(0, want_knn=True).cpu().numpy()
            cov = b.covariances(0)
            assert np.isfinite(cov).all(), (name, core)
        knn = res[3]
        kk = min(k, n)
        assert (knn[:, kk:] == -1).all() and (knn[:, :kk] >= 0).all() and (knn[:, :kk] < n).all(), name
        assert all(len(set(r[:kk])) == kk for r in knn[:: max(1, n // 500)]), name        # no point twice
        d = _knn_d2(oracle, cloud, knn[:, :kk])
        want = _knn_d2(oracle, cloud, oracle.knn(cloud, kk))                               # the restatement's exact kd-tree search
        assert np.array_equal(d, np.sort(want, 1)), name                                   # same multiset of float distances, ascending
        # both kernels are tie-safe (a cluster of more than k duplicates must neither overflow the candidate list nor push a
        # closer point out): the round-3 kernel returns the same distances on every cloud, the duplicate clusters included
        d0 = _knn_d2(oracle, cloud, res[0][:, :kk])
        assert np.array_equal(d0, d), name
        assert (res[0][:, kk:] == -1).all() and all(len(set(r[:kk])) == kk for r in res[0][:: max(1, n // 500)]), name


@pytest.mark.parametrize("k", [15, 30])
def test_knn_second_pass_forms_agree(dev, k):
    """Pass 2 of the k-NN selection revisits the minis pass 1 noted (wave-private list, 64 entries; a wave that noted more walks the hierarchy
    again).  Both forms must return the same neighbour INDICES on every cloud: the development switch MRS_KNN_REC=0 forces the walk."""
    import os
    from mr_slam_amd import gicp
    old = {v: os.environ.get(v) for v in ("MRS_DEV", "MRS_KNN_REC")}
    try:
        for name, cloud in _clouds_for_search_tests().items():
            res = []
            for rec in ("1", "0"):
                os.environ["MRS_DEV"] = "1"; os.environ["MRS_KNN_REC"] = rec
                b = gicp.GicpBatch(1)
                b.set_params(k_correspondences=k)
                b.set_sources([cloud])
                res.append(b.compute_covariances(0, want_knn=True).cpu().numpy())
            assert np.array_equal(res[0], res[1]), name
    finally:
        for v, x in old.items():
            if x is None:
                os.environ.pop(v, None)
            else:
                os.environ[v] = x
        b = gicp.GicpBatch(1); b.set_sources([_clouds_for_search_tests()["seventeen"]]); b.compute_covariances(0)      # back to the default form


def test_correspondences_agree_between_search_cores(dev, oracle):
    """k_nn_scan_g (round 4) and k_nn_scan (round 3) on the same pairs and poses: the same squared distance for every source point
    (indices may differ only at exact ties), cold and warm-started, with and without a correspondence threshold."""
    from mr_slam_amd import gicp
    src, tgt, Ttrue = _pair(21, 40000)
    poses = [np.eye(4), Ttrue]
    for max_corr in (5.0, 0.3, 1e300):
        got = {}
        for core in (1, 0):
            b = gicp.GicpBatch(2)
            b.set_search(core)
            b.set_params(max_correspondence_distance=max_corr)
            b.set_sources([src, src[:777]]); b.set_targets([tgt, tgt[:5000]])
            out = []
            for T in poses:                      # the second call is warm-started from the first one's neighbours
                e, H, bb, corr = b.linearize(np.stack([T, T]), want_corr=True)
                out.append((e, corr))
            got[core] = out
        for (e1, c1), (e0, c0), T in zip(got[1], got[0], poses):
            for sl, s, t in ((slice(0, src.shape[0]), src, tgt), (slice(src.shape[0], None), src[:777], tgt[:5000])):
                a = oracle.pair_d2(s, T, t, c1[sl]); bq = oracle.pair_d2(s, T, t, c0[sl])
                assert np.array_equal(a, bq)
                assert np.array_equal(c1[sl] >= 0, c0[sl] >= 0)
            assert (c1 == c0).mean() > 0.999
            np.testing.assert_allclose(e1, e0, rtol=1e-6)


def test_align_identical_between_search_cores(dev):
    """whole alignments (cold start, LM trials, warm passes): every search setting gives the same transforms -- the round-3 core (0),
    the round-4 core with certified neighbours (1, the default), without certificates (2), with the round-4 kernel for the cold pass (3)"""
    from mr_slam_amd import gicp
    pairs = [_pair(31 + i, 15000 + 1000 * i) for i in range(3)]
    out, frac = {}, {}
    for core in (1, 0, 2, 3):
        b = gicp.GicpBatch(3)
        b.set_search(core)
        b.set_params(k_correspondences=15, max_correspondence_distance=5.0)
        b.set_sources([p[0] for p in pairs]); b.set_targets([p[1] for p in pairs])
        out[core] = b.align()
        frac[core] = b.searched_fraction
    for core in (1, 2, 3):
        assert np.array_equal(out[core][0], out[0][0]) and (out[core][2] == out[0][2]).all() and (out[core][1] == out[0][1]).all(), core
    assert frac[0] == 1.0 and frac[2] == 1.0 and frac[1] < 0.9 and frac[3] < 0.9, frac


def test_certified_passes_are_exact(dev, oracle):
    """Certificates (k_nn_certify): with forced iterations long past convergence nearly every pass is certified, and the correspondences
    the certified passes leave behind are the exact nearest neighbours at the final pose: a fresh full search at that pose returns the same
    squared distances for every source point, with and without a correspondence threshold, for a pose that keeps moving (second align from
    a perturbed guess) and for clouds with duplicated points (exact ties can never be certified)."""
    from mr_slam_amd import gicp
    src, tgt, Ttrue = _pair(41, 30000)
    dup = np.concatenate([tgt, tgt[:3000]])                      # 3000 exact duplicates in the target
    for target, max_corr in ((tgt, 5.0), (tgt, 0.25), (dup, 5.0)):
        res = {}
        for core in (1, 2):
            b = gicp.GicpBatch(1)
            b.set_search(core)
            b.set_params(k_correspondences=15, max_correspondence_distance=max_corr, force_iterations=14)
            b.set_sources([src]); b.set_targets([target])
            T, _, its = b.align()
            f1 = b.searched_fraction
            g = np.eye(4); g[:3, 3] = [0.05, -0.04, 0.02]
            T2, _, _ = b.align((g @ T[0])[None])                  # seeds warm, certificates start over
            res[core] = (T, T2, f1, b.searched_fraction)
            # the correspondences a certified run leaves behind == a fresh exact search at the same pose
            if core == 1:
                b2 = gicp.GicpBatch(1)
                b2.set_search(0)
                b2.set_params(k_correspondences=15, max_correspondence_distance=max_corr)
                b2.set_sources([src]); b2.set_targets([target])
                for pose in (T[0], T2[0]):
                    _, _, _, c_new = b.linearize(pose[None], want_corr=True)
                    _, _, _, c_old = b2.linearize(pose[None], want_corr=True)
                    assert np.array_equal(oracle.pair_d2(src, pose, target, c_new), oracle.pair_d2(src, pose, target, c_old))
        assert np.array_equal(res[1][0], res[2][0]) and np.array_equal(res[1][1], res[2][1])
        assert res[2][2] == 1.0 and res[1][2] < 0.5, res[1][2:]   # most (point, pass) pairs of 14 forced iterations were certified


def test_submap_store_pairs_equal_per_pair_clouds_bit_for_bit(dev):
    """A new scan against several stored candidates (main_RING.py:81-104; global_manager.cpp:2016-2021): clouds kept once in a store batch
    (Morton order, boxes, hierarchy, covariances) and COPIED into the pairs give the transforms, iteration counts and fitness of handing every
    pair its own copies of the points."""
    from mr_slam_amd import gicp, synth
    rng = np.random.default_rng(11)
    scans = [synth.lidar_scan(700 + i, 9000 + 500 * i, metric=True) for i in range(4)]          # 4 stored submaps of different sizes
    new = (scans[1] @ np.array([[np.cos(0.03), -np.sin(0.03), 0], [np.sin(0.03), np.cos(0.03), 0], [0, 0, 1]]).T + np.array([0.2, -0.1, 0.02])
           + rng.normal(0, 0.01, scans[1].shape)).astype(np.float32)
    pairs_src = [4, 4, 4, 4, 2]                                                                   # the new scan against all four + one stored pair
    pairs_tgt = [0, 1, 2, 3, 3]
    store = gicp.GicpBatch(5)
    store.set_params(k_correspondences=15, max_correspondence_distance=5.0)
    store.set_targets(scans + [new])
    store.compute_covariances(1)
    a = gicp.GicpBatch(5)
    a.set_params(k_correspondences=15, max_correspondence_distance=5.0)
    a.set_sources_from(store, pairs_src)
    a.set_targets_from(store, pairs_tgt)
    Ta, ca, ia = a.align()
    fa = a.fitness(Ta, 1.0)
    b = gicp.GicpBatch(5)
    b.set_params(k_correspondences=15, max_correspondence_distance=5.0)
    allc = scans + [new]
    b.set_sources([allc[i] for i in pairs_src])
    b.set_targets([allc[i] for i in pairs_tgt])
    Tb, cb, ib = b.align()
    fb = b.fitness(Tb, 1.0)
    assert np.array_equal(Ta, Tb) and np.array_equal(ca, cb) and np.array_equal(ia, ib) and np.array_equal(fa, fb)
    assert np.array_equal(a.covariances(0), b.covariances(0)) and np.array_equal(a.covariances(1), b.covariances(1))
    assert ca[1] and np.abs(Ta[1][:3, 3] - np.array([-0.2, 0.1, -0.02])).max() < 0.08           # pair 1 is the true match
    # the pairs change, the store stays: a second round against other candidates re-uses everything
    a.set_targets_from(store, [3, 2, 1, 0, 0])
    T2, _, _ = a.align()
    b.set_targets([allc[i] for i in [3, 2, 1, 0, 0]])
    T3, _, _ = b.align()
    assert np.array_equal(T2, T3)
