"""The reference's BEV boundary is a set of COMPILED extension modules (Cython: generate_bev_cython_binary/wrapper.pyx,
generate_bev_pointfeat_cython/wrapper.pyx, multi-layer-polar-*/cython/gputransform.pyx).  bindings/cython/ holds the same
three modules -- same module, class and method names, same buffer-typed constructor signatures -- compiled by the image's
Cython against libmrslam_hip.so.  CPU: they build, import and validate arguments; GPU: retreive() / get_features() equal
the checkers."""
import importlib
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILT = os.path.join(ROOT, "bindings", "cython", "_built")


@pytest.fixture(scope="module")
def mods():
    import glob
    if len(glob.glob(os.path.join(BUILT, "*.so"))) < 3:
        import __graft_entry__
        if not os.path.exists(os.path.join(ROOT, "mr_slam_amd", "libmrslam_hip.so")):
            __graft_entry__.build()
        else:
            __graft_entry__.build_bindings()
    saved = {n: sys.modules.pop(n, None) for n in ("voxelocc", "gputransform", "voxelfeat")}
    sys.path.insert(0, BUILT)
    try:
        out = {n: importlib.import_module(n) for n in ("voxelocc", "gputransform", "voxelfeat")}
    finally:
        sys.path.remove(BUILT)
        for n, m in saved.items():
            sys.modules.pop(n, None)
            if m is not None:
                sys.modules[n] = m
    return out


def test_modules_are_compiled_extensions_with_the_reference_surface(mods):
    for name, m in mods.items():
        assert m.__file__.endswith(".so") and os.path.dirname(m.__file__) == BUILT
        assert hasattr(m, "GPUTransformer")
        for meth in ("transform", "retreive"):
            assert callable(getattr(m.GPUTransformer, meth))
    assert callable(mods["voxelfeat"].GPUFeatureExtractor.get_features)
    # links the C-ABI library, nothing CUDA
    out = subprocess.run(["ldd", mods["voxelocc"].__file__], capture_output=True, text=True).stdout
    assert "libmrslam_hip.so" in out and "cuda" not in out.lower()


def test_buffer_typed_constructors_reject_what_the_reference_rejects(mods):
    soa = np.zeros(30, np.float32)
    with pytest.raises(ValueError):                       # Cython buffer check: "Buffer dtype mismatch"
        mods["voxelocc"].GPUTransformer(soa.astype(np.float64), 10, 1, 1, 120, 120, 1, 1)
    with pytest.raises(ValueError):                       # ndim=1, mode="c"
        mods["gputransform"].GPUTransformer(np.zeros((3, 10), np.float32), 10, 1, 1, 40, 120, 20, 1)
    with pytest.raises(TypeError):                        # `not None`
        mods["voxelfeat"].GPUTransformer(None, 10, 1, 1, 120, 120, 1, 9)
    with pytest.raises(ValueError):
        mods["voxelocc"].GPUTransformer(soa, 11, 1, 1, 120, 120, 1, 1)     # shorter than 3 * size
    mods["voxelocc"].GPUTransformer(soa, 10, 1, 1, 120, 120, 1, 1)         # constructing needs no GPU


@pytest.mark.gpu
def test_compiled_modules_match_the_checkers(mods, oracle):
    from mr_slam_amd import synth
    s = synth.lidar_scan(5, 25000)
    soa = synth.to_soa(s)
    n = s.shape[0]
    t = mods["voxelocc"].GPUTransformer(soa, n, 1, 1, 120, 120, 1, 1)
    t.transform()
    got = t.retreive()
    np.testing.assert_array_equal(got, oracle.bev_cart(soa, 1, 1, 120, 120, 1))
    if oracle.ref_lib("cart") is not None:
        np.testing.assert_array_equal(got, oracle.ref_bev_cart(soa, 1, 1, 120, 120, 1))      # the reference's own code
    g = mods["gputransform"].GPUTransformer(soa, n, 1, 1, 40, 120, 20, 1)
    g.transform()
    np.testing.assert_array_equal(g.retreive(), oracle.ref_bev_polar(soa, 1, 1, 40, 120, 20, 1) if oracle.ref_polar() is not None
                                  else oracle.bev_polar(soa, 1, 1, 40, 120, 20, 1))
    F = 9
    pts = np.concatenate([soa, np.random.default_rng(6).uniform(0, 1, size=(F - 3) * n).astype(np.float32)])
    f = mods["voxelfeat"].GPUTransformer(pts, n, 1, 1, 120, 120, 1, F)
    f.transform()
    np.testing.assert_array_equal(f.retreive(), oracle.bev_feat(pts, F, 1, 1, 120, 120, 1))
    # GPUFeatureExtractor on the reference's own neighbour / eigenvalue inputs (util.py:204-218)
    from oracle import pointfeat_oracle as PF
    p = s[:4000]
    idx = PF.knn_indices(p, 30)
    eig = PF.covariation_eigenvalue(p, idx)
    x = mods["voxelfeat"].GPUFeatureExtractor(p.reshape(-1).copy(), p.shape[0], 13, 30, idx.reshape(-1).astype(np.int32), eig.reshape(-1).copy())
    feats = x.get_features().reshape(-1, 13)
    want = oracle.ref_point_features(p, idx, eig) if oracle.ref_lib("feat") is not None else PF.calculate_features(p, idx, eig)
    ok = np.isfinite(want).all(1)
    np.testing.assert_allclose(feats[ok], want[ok], rtol=2e-4, atol=1e-6)
