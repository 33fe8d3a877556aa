"""mr_slam_amd/csrc/eig3.hpp -- the closed-form symmetric 3x3 eigen-solver of the covariance tails (k_cov_from_knn, k_feat_from_knn) -- compiled
for the HOST with g++ from the very header the kernels include, against LAPACK (numpy) and against the cyclic Jacobi of the oracle
(oracle/gicp_oracle.cpp:136, through Gicp.covariances).  No GPU needed: the device build compiles the same text."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "tests", "cpp", "build", "libeig3_host.so")


def _lib():
    os.makedirs(os.path.dirname(SO), exist_ok=True)
    src = os.path.join(ROOT, "tests", "cpp", "eig3_host.cpp")
    hdr = os.path.join(ROOT, "mr_slam_amd", "csrc", "eig3.hpp")
    if not os.path.exists(SO) or os.path.getmtime(SO) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        r = subprocess.run(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-shared", "-fPIC", src, "-o", SO], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
    return C.CDLL(SO)


def _vec(A):
    A = np.ascontiguousarray(A, np.float64).reshape(-1, 9)
    out = np.empty((A.shape[0], 3))
    _lib().eig3_smallest_vec(A.ctypes.data_as(C.c_void_p), A.shape[0], out.ctypes.data_as(C.c_void_p))
    return out


def _vals(A):
    A = np.ascontiguousarray(A, np.float64).reshape(-1, 9)
    out = np.empty((A.shape[0], 3))
    _lib().eig3_values_desc(A.ctypes.data_as(C.c_void_p), A.shape[0], out.ctypes.data_as(C.c_void_p))
    return out


def _random_spd(rng, n, spread):
    Q, _ = np.linalg.qr(rng.normal(size=(n, 3, 3)))
    w = np.sort(10.0 ** rng.uniform(-spread, 0, size=(n, 3)), axis=1)
    return np.einsum("nij,nj,nkj->nik", Q, w, Q), Q, w


def test_smallest_eigenvector_matches_lapack():
    rng = np.random.default_rng(0)
    for spread in (1, 4, 8):
        A, _, _ = _random_spd(rng, 20000, spread)
        A *= 10.0 ** rng.uniform(-6, 6, size=(A.shape[0], 1, 1))          # any overall scale
        v = _vec(A)
        w, V = np.linalg.eigh(A)
        assert np.abs(np.linalg.norm(v, axis=1) - 1).max() < 1e-14
        gap = (w[:, 1] - w[:, 0]) / w[:, 2]
        err = np.linalg.norm(np.cross(v, V[:, :, 0]), axis=1)              # sine of the angle: sign-free
        # backward-stable accuracy: angle error ~ eps ||A|| / gap (LAPACK's own vector carries the same bound)
        assert (err * gap).max() < 2e-14, spread
        # eigenvalues: all three to ~eps ||A|| (the pair that may meet comes from the 2x2 stage, not from the trigonometric form)
        assert (np.abs(_vals(A) - w[:, ::-1]).max(1) / w[:, 2]).max() < 1e-14


def test_rank_deficient_and_degenerate_inputs():
    # exactly collinear neighbours: rank 1 -> any unit vector orthogonal to the line
    d = np.array([1.0, 2.0, -0.5]); d /= np.linalg.norm(d)
    A = np.outer(d, d) * 3.7
    v = _vec(A[None])[0]
    assert abs(np.linalg.norm(v) - 1) < 1e-14 and abs(v @ d) < 1e-12
    # zero matrix (all neighbours coincide), multiples of the identity: (1, 0, 0) like the Jacobi this replaces
    assert np.array_equal(_vec(np.zeros((1, 3, 3)))[0], [1, 0, 0])
    assert np.array_equal(_vec(np.eye(3)[None] * 5.0)[0], [1, 0, 0])
    # diagonal, smallest in the middle
    assert np.allclose(np.abs(_vec(np.diag([3.0, 1.0, 2.0])[None])[0]), [0, 1, 0], atol=1e-15)
    # exact plane z = 0 (smallest eigenvalue exactly 0, well separated)
    rng = np.random.default_rng(1)
    P = np.c_[rng.normal(size=(20, 2)), np.zeros(20)]
    P -= P.mean(0)
    v = _vec((P.T @ P / 20)[None])[0]
    assert np.allclose(np.abs(v), [0, 0, 1], atol=1e-14)
    # double LARGE eigenvalue (disc): the normal is still well defined
    A = np.diag([2.0, 2.0, 1e-6])
    Q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
    v = _vec((Q @ A @ Q.T)[None])[0]
    assert np.linalg.norm(np.cross(v, Q[:, 2])) < 1e-14
    # eigenvalues of the same inputs
    assert np.allclose(_vals(np.diag([3.0, 1.0, 2.0])[None])[0], [3, 2, 1], atol=1e-15)
    assert np.array_equal(_vals(np.zeros((1, 3, 3)))[0], [0, 0, 0])
    assert np.all(np.isfinite(_vec(np.full((1, 3, 3), 1e-300)))) and np.all(np.isfinite(_vec(np.full((1, 3, 3), 1e300))))


def test_plane_regularised_covariances_match_the_oracle_jacobi():
    """The quantity the GICP tail stores: C = I - 0.999 n n^T of a noisy lidar scan's k = 15 neighbourhoods (ring-line neighbourhoods have two
    small eigenvalues 1e-5 of the largest apart) against the oracle's Jacobi, at the tolerance of tests/test_gicp_gpu.py (1e-9; measured 4e-13)."""
    import sys
    sys.path.insert(0, ROOT)
    from oracle import pyoracle as O
    from mr_slam_amd import synth
    rng = np.random.default_rng(5)
    pts = (synth.lidar_scan(500, 40000, metric=True).astype(np.float64) + rng.normal(0, 0.02, (40000, 3))).astype(np.float32)
    k = 15
    g = O.Gicp(k=k); g.set_source(pts); g.set_target(pts[:64])
    want = g.covariances(0)
    P = pts.astype(np.float64)[O.knn(pts, k)]
    D = P - P.mean(1, keepdims=True)
    A = np.einsum("nki,nkj->nij", D, D) / k
    n = _vec(A)
    got = np.eye(3)[None] - 0.999 * n[:, :, None] * n[:, None, :]
    assert np.abs(got - want).max() < 1e-10
