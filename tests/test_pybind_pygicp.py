"""pygicp: the compiled (pybind11) module with the part of fast_gicp's Python binding the LoopDetection nodes call
(main_RING.py:81-104), built over the C ABI (bindings/pybind/pygicp.cpp).  CPU: it builds and exposes the names; GPU: the
reference's own `fast_gicp(source, target, ...)` helper body runs on it and agrees with the Python mirror and the checker."""
import glob
import importlib
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILT = os.path.join(ROOT, "bindings", "pybind", "_built")


@pytest.fixture(scope="module")
def pygicp():
    if not glob.glob(os.path.join(BUILT, "pygicp*.so")):
        import __graft_entry__
        if not os.path.exists(os.path.join(ROOT, "mr_slam_amd", "libmrslam_hip.so")):
            __graft_entry__.build()
        else:
            __graft_entry__.build_bindings()
    import torch  # noqa: F401  -- before the compiled module, like every reference node (torch wheels carry a HIP runtime)
    saved = sys.modules.pop("pygicp", None)
    sys.path.insert(0, BUILT)
    try:
        return importlib.import_module("pygicp")
    finally:
        sys.path.remove(BUILT)
        if saved is not None:
            sys.modules["pygicp"] = saved
        else:
            sys.modules.pop("pygicp", None)


def test_module_surface(pygicp):
    assert pygicp.__file__.endswith(".so")
    for n in ("downsample", "align_points", "FastGICP"):
        assert hasattr(pygicp, n), n
    for n in ("set_input_target", "set_input_source", "set_num_threads", "set_max_correspondence_distance",
              "set_correspondence_randomness", "align", "get_final_transformation", "get_fitness_score", "has_converged"):
        assert hasattr(pygicp.FastGICP, n), n                       # main_RING.py:88-102
    with pytest.raises(ValueError):
        pygicp.downsample(np.zeros((5, 2)), 0.2)


@pytest.mark.gpu
def test_reference_helper_body_on_the_compiled_module(pygicp, oracle):
    """The body of the reference's fast_gicp() helper (main_RING.py:81-104), statement for statement, on the compiled module."""
    from scipy.spatial.transform import Rotation as Rot
    from mr_slam_amd.compat import pygicp as mirror
    rng = np.random.default_rng(8)
    ground = np.c_[rng.uniform(-20, 20, (30000, 2)), 0.05 * rng.normal(size=30000)]
    wall = np.c_[rng.uniform(-20, 20, 15000), np.full(15000, 8.0) + 0.05 * rng.normal(size=15000), rng.uniform(0, 4, 15000)]
    box = np.c_[rng.uniform(3, 6, 8000), rng.uniform(-7, -4, 8000), rng.uniform(0, 2, 8000)]
    target = np.concatenate([ground, wall, box])
    Ttrue = np.eye(4); Ttrue[:3, :3] = Rot.from_rotvec([0, 0, 0.04]).as_matrix(); Ttrue[:3, 3] = [0.4, -0.3, 0.02]
    source = (target[rng.permutation(len(target))[:40000]] - Ttrue[:3, 3]) @ Ttrue[:3, :3] + 0.01 * rng.normal(size=(40000, 3))

    def fast_gicp(mod, source, target, max_correspondence_distance=1.0, init_pose=np.eye(4)):
        source = mod.downsample(source, 0.2)
        target = mod.downsample(target, 0.2)
        gicp = mod.FastGICP()
        gicp.set_input_target(target)
        gicp.set_input_source(source)
        gicp.set_num_threads(4)
        gicp.set_max_correspondence_distance(max_correspondence_distance)
        T_matrix = gicp.align(initial_guess=init_pose)
        fitness = gicp.get_fitness_score(1.0)
        T_matrix = gicp.get_final_transformation()
        return fitness, T_matrix, source, target

    fit, T, s_ds, t_ds = fast_gicp(pygicp, source, target, 5.0)
    fit_m, T_m, s_m, t_m = fast_gicp(mirror, source, target, 5.0)
    np.testing.assert_array_equal(s_ds, s_m); np.testing.assert_array_equal(t_ds, t_m)      # same kernel behind both
    np.testing.assert_array_equal(s_ds, oracle.approx_voxel_grid(source, 0.2))              # = the sequential filter, bit for bit
    np.testing.assert_allclose(T, T_m, atol=1e-12)
    assert abs(fit - fit_m) < 1e-12 and fit < 0.05
    assert np.linalg.norm(T[:3, 3] - Ttrue[:3, 3]) < 0.03 and np.abs(T[:3, :3] - Ttrue[:3, :3]).max() < 5e-3
    T2 = pygicp.align_points(target, source, downsample_resolution=0.2, max_correspondence_distance=5.0, k_correspondences=20)
    np.testing.assert_allclose(T2, T, atol=1e-12)
