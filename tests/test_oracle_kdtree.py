"""CPU suite, row N4: the brute-force checker of the signature search pinned to the reference's own kd-tree
(Mapping/src/global_manager/src/kdtree.cpp compiled in place -> oracle/_ref/libref_kdtree.so), driven like
GlobalManager::detectLoopClosure does (global_manager.cpp:1002-1007: k = NUM_CANDIDATES_FROM_TREE = 10)."""
import numpy as np
import pytest


def brute_knn(db, q, k):
    d2 = ((db.astype(np.float64) - q.astype(np.float64)) ** 2).sum(1)
    order = np.argsort(d2, kind="stable")[:k]
    return order, np.sqrt(d2[order])


@pytest.mark.parametrize("n,dim,k,seed", [(3000, 1024, 10, 0), (257, 37, 10, 1), (6, 8, 10, 2), (1, 16, 3, 3)])
def test_reference_kdtree_is_exact_k_nearest_ascending(oracle, n, dim, k, seed):
    if oracle.ref_lib("kdtree") is None:
        pytest.skip("oracle/_ref/libref_kdtree.so not built (no reference tree at build time)")
    rng = np.random.default_rng(seed)
    db = rng.normal(size=(n, dim)).astype(np.float32)
    for trial in range(3):
        q = (db[rng.integers(n)] + 0.05 * rng.normal(size=dim)).astype(np.float32) if trial < 2 else rng.normal(size=dim).astype(np.float32)
        idx, dist = oracle.ref_kdtree_knn(db, q, k)
        want_i, want_d = brute_knn(db, q, k)
        assert len(idx) == min(k, n)
        np.testing.assert_array_equal(idx, want_i)
        np.testing.assert_allclose(dist, want_d, rtol=2e-5)      # float32 running sum inside the tree vs float64 here
        assert np.all(np.diff(dist) >= 0)


def test_numpy_literal_of_calc_rel_ori_equals_the_reference_function(oracle):
    """The numpy statement the GPU test checks mrs_disco_rel_ori_literal against (non-conjugate cross term, no normalisation,
    unshifted argmax, x 3 degrees) vs GlobalManager::calcRelOri itself (global_manager.cpp:2719-2762 built with FFTW / Eigen
    stand-ins: oracle/_ref/libref_relori.so)."""
    if oracle.ref_lib("relori") is None:
        pytest.skip("oracle/_ref/libref_relori.so not built (no reference tree at build time)")
    rng = np.random.default_rng(4)
    base = (rng.uniform(size=(40, 120)) > 0.9).astype(np.float32)
    spec = lambda img: np.fft.fft2(img).astype(np.complex64)
    seen = set()
    for shift in (0, 7, -31, 59, 60, 119):
        A, B = spec(base), spec(np.roll(base, shift, axis=1) + 0.05 * rng.uniform(size=base.shape).astype(np.float32))
        ra, ia, rb, ib = A.real, A.imag, B.real, B.imag
        cross = (ra * rb + ia * ib).astype(np.float64) + 1j * (ra * ib + rb * ia).astype(np.float64)
        real = (np.fft.ifft2(cross) * cross.size).real.astype(np.float32)
        want = float(int(np.argmax(real)) % 120) * 3.0
        got = oracle.ref_calc_rel_ori(A, B)
        assert got == want, (shift, got, want)
        seen.add(got)
    assert len(seen) >= 3
