"""torch_radon_cuda: the compiled (pybind11) backend module of the reference's torch-radon package
(LoopDetection/torch-radon/src/pytorch.cpp:175-260), rebuilt over the C ABI (bindings/pybind/torch_radon_cuda.cpp).
CPU: it builds, exposes the reference's names, and the reference's OWN Python package torch_radon imports and
constructs a ParallelBeam on top of it (where /root/reference exists).  GPU: forward() equals the checker."""
import importlib
import os
import sys
import types

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILT = os.path.join(ROOT, "bindings", "pybind", "_built")
REF_PKG = "/root/reference/LoopDetection/torch-radon"


@pytest.fixture(scope="module")
def backend():
    import glob
    if not glob.glob(os.path.join(BUILT, "torch_radon_cuda*.so")):
        import __graft_entry__
        if not os.path.exists(os.path.join(ROOT, "mr_slam_amd", "libmrslam_hip.so")):
            __graft_entry__.build()
        else:
            __graft_entry__.build_bindings()
    sys.path.insert(0, BUILT)
    try:
        return importlib.import_module("torch_radon_cuda")
    finally:
        sys.path.remove(BUILT)


def test_backend_module_surface(backend):
    assert backend.__file__.endswith(".so")
    for n in ("forward", "backward", "add_noise", "symbolic_forward", "symbolic_discretize", "rfft", "irfft", "set_log_level",
              "TextureCache", "FFTCache", "RadonNoiseGenerator", "VolumeCfg", "ProjectionCfg", "ExecCfg"):
        assert hasattr(backend, n), n                                          # pytorch.cpp:175-260
    v = backend.VolumeCfg(0, 120, 100, 0.0, 0.0, 0.0, 1.0, 1.0, 1.0, False)
    assert (v.height, v.width, v.is_3d) == (120, 100, False)
    p = backend.ProjectionCfg(120, 1.0)
    p.n_angles = 7
    q = p.copy()
    assert p.is_2d() and q.n_angles == 7 and q.det_count_u == 120 and p.projection_type == 0
    assert not backend.ProjectionCfg(64, 1.0, 64, 1.0, 100.0, 100.0, 0.0, 0.0, 2).is_2d()
    backend.ExecCfg(16, 16, 1, 4); backend.TextureCache(8).free(); backend.FFTCache(8).free()
    with pytest.raises(NotImplementedError):
        backend.backward(None, None)


@pytest.mark.skipif(not os.path.isdir(REF_PKG), reason="/root/reference not present")
def test_reference_python_package_runs_on_the_backend(backend):
    """The reference's torch_radon package, unmodified, imports `torch_radon_cuda` = this module, builds its
    Projection / Volume2D / TextureCache objects and reaches backend.forward (which refuses a CPU tensor like
    pytorch.cpp:16-20).  Only `turtle` (volumes.py:1, a stray import that needs tkinter) is stubbed."""
    import torch
    saved = {k: sys.modules.get(k) for k in ("turtle", "torch_radon", "torch_radon_cuda")}
    for k in list(sys.modules):
        if k == "torch_radon" or k.startswith("torch_radon."):
            saved.setdefault(k, sys.modules[k]); del sys.modules[k]
    t = types.ModuleType("turtle"); t.width = None
    sys.modules["turtle"] = t
    sys.modules["torch_radon_cuda"] = backend
    sys.path.insert(0, REF_PKG)
    try:
        tr = importlib.import_module("torch_radon")
        assert tr.__file__.startswith(REF_PKG) and tr.cuda_backend.forward is backend.forward
        pb = tr.ParallelBeam(120, np.linspace(0, 2 * np.pi, 120).astype(np.float32))
        assert isinstance(pb.projection.cfg, backend.ProjectionCfg) and isinstance(pb.tex_cache, backend.TextureCache)
        with pytest.raises(RuntimeError, match="CUDA tensor"):
            pb.forward(torch.zeros(1, 120, 120))
    finally:
        sys.path.remove(REF_PKG)
        for k in list(sys.modules):
            if k == "torch_radon" or k.startswith("torch_radon.") or k == "turtle":
                del sys.modules[k]
        for k, m in saved.items():
            if m is not None:
                sys.modules[k] = m


@pytest.mark.gpu
def test_backend_forward_matches_checker(backend, oracle):
    import torch
    rng = np.random.default_rng(2)
    imgs = (rng.random((5, 120, 120)) * (rng.random((5, 120, 120)) < 0.3)).astype(np.float32)
    ang = np.linspace(0, 2 * np.pi, 120).astype(np.float32)
    x = torch.from_numpy(imgs).cuda()
    cache = backend.TextureCache(8)
    y = backend.forward(x, torch.from_numpy(ang).cuda(), cache, backend.VolumeCfg(0, 120, 120, 0.0, 0.0, 0.0, 1.0, 1.0, 1.0, False),
                        backend.ProjectionCfg(120, 1.0), backend.ExecCfg(16, 16, 1, 4))
    assert tuple(y.shape) == (5, 120, 120) and y.is_cuda
    np.testing.assert_array_equal(y.cpu().numpy(), oracle.radon_parallel(imgs, ang, 120, 1.0))
    # another geometry through the same cache, non-square image, explicit stream
    img2 = rng.random((3, 64, 96)).astype(np.float32)
    ang2 = np.linspace(0, np.pi, 45, endpoint=False).astype(np.float32)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        y2 = backend.forward(torch.from_numpy(img2).cuda(), torch.from_numpy(ang2).cuda(), cache,
                             backend.VolumeCfg(0, 64, 96, 0.0, 0.0, 0.0, 1.0, 1.0, 1.0, False), backend.ProjectionCfg(110, 1.0), backend.ExecCfg(16, 16, 1, 4))
    s.synchronize()
    np.testing.assert_array_equal(y2.cpu().numpy(), oracle.radon_parallel(img2, ang2, 110, 1.0))
    with pytest.raises(RuntimeError):
        backend.forward(x.cpu(), torch.from_numpy(ang).cuda(), cache, backend.VolumeCfg(0, 120, 120, 0, 0, 0, 1, 1, 1, False),
                        backend.ProjectionCfg(120, 1.0), backend.ExecCfg(16, 16, 1, 4))
