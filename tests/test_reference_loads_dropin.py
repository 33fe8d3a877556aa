"""Boundary test: the UNMODIFIED reference modules import and bind against the drop-in packages.

`mr_slam_amd.compat.install()` registers voxelocc / voxelfeat / gputransform / torch_radon / pygicp; then the reference's
own RING_ros/util.py and disco_ros/models/DiSCO.py are imported from /root/reference (skipped where that tree is absent,
i.e. on the GPU box).  Import only -- no compute, so it runs without a GPU: it proves every name the reference takes
from the boundary modules exists with the reference's call signature."""
import inspect
import os
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import ref_import  # noqa: E402

needs_ref = pytest.mark.skipif(not ref_import.available(), reason="/root/reference not present")


@needs_ref
def test_reference_util_imports_against_the_dropin():
    with ref_import.reference_modules("dropin") as ref:
        u = ref.util                                           # runs util.py:1-24 incl. `from torch_radon import Radon, ParallelBeam, RadonFanbeam`
        import mr_slam_amd.compat.torch_radon as tr
        import mr_slam_amd.compat.voxelocc as vo
        import mr_slam_amd.compat.voxelfeat as vf
        assert u.ParallelBeam is tr.ParallelBeam and u.Radon is tr.Radon and u.RadonFanbeam is tr.RadonFanbeam
        assert u.voxelocc is vo and u.voxelfeat is vf
        # the constructor calls of util.py:180,192,218,228,242 bind
        inspect.signature(vo.GPUTransformer.__init__).bind(None, None, 1, 1, 1, 120, 120, 1, 1)
        inspect.signature(vf.GPUFeatureExtractor.__init__).bind(None, None, 1, 13, 30, None, None)
        inspect.signature(vf.GPUTransformer.__init__).bind(None, None, 1, 1, 1, 120, 120, 1, 9)
        inspect.signature(tr.ParallelBeam.__init__).bind(None, 120, None)
        with pytest.raises(NotImplementedError):
            u.RadonFanbeam(120, [0.0], 100.0)                  # importable, not on the path


@needs_ref
def test_reference_disco_model_imports_against_the_dropin():
    with ref_import.reference_modules("dropin") as ref:
        d = ref.disco                                          # DiSCO.py:13 `import gputransform`
        import mr_slam_amd.compat.gputransform as gt
        assert d.gputransform is gt
        inspect.signature(gt.GPUTransformer.__init__).bind(None, None, 1, 1, 1, 40, 120, 20, 1)   # disco_ros/main.py:118


@needs_ref
def test_pygicp_surface_used_by_the_nodes():
    """main_RING.py:81-104 / disco_ros/main.py:174-197 / main_SC.py:108-131."""
    from mr_slam_amd.compat import pygicp
    for name in ("downsample", "FastGICP"):
        assert hasattr(pygicp, name)
    for m in ("set_input_target", "set_input_source", "set_num_threads", "set_max_correspondence_distance", "align",
              "get_fitness_score", "get_final_transformation"):
        assert callable(getattr(pygicp.FastGICP, m))


def test_torch_radon_package_surface():
    """Names exported by torch_radon/__init__.py:3-9."""
    from mr_slam_amd.compat import torch_radon as tr
    for n in ("Volume2D", "Volume3D", "FanBeam", "ParallelBeam", "ConeBeam", "Radon", "RadonFanbeam"):
        assert hasattr(tr, n)
    import numpy as np
    pb = tr.ParallelBeam(120, np.linspace(0, 2 * np.pi, 120).astype(np.float32))
    assert pb.det_count == 120 and pb.det_spacing == 1.0 and tuple(pb.angles.shape) == (120,)
    import torch
    with pytest.raises(RuntimeError):
        pb.forward(torch.zeros(1, 120, 120))                   # CPU tensor: same refusal as pytorch.cpp:16-20
    with pytest.raises(NotImplementedError):
        pb.backward(None)
