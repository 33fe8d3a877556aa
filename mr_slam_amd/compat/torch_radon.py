"""Drop-in for the part of `torch_radon` MR_SLAM uses: ParallelBeam(...).forward(x)
(torch_radon/radon.py:62-87,139-167; call sites RING_ros/util.py:192-195,241-245).

Every name the reference's own modules import from the package resolves (`from torch_radon import Radon,
ParallelBeam, RadonFanbeam`: RING_ros/util.py:15, pr_methods/RadonSinogram.py:8; the package exports
torch_radon/__init__.py:3-9): the parallel-beam forward projection is implemented on the HIP path, everything MR_SLAM
never calls (back-projection, filtering, fan / cone beam, 3-D volumes: SURVEY.md section 2 row 5) raises
NotImplementedError when USED, not when imported."""
import warnings

import numpy as np
import torch

from .. import ring as _ring

__version__ = "2.0"   # torch_radon/__init__.py:12


class Volume2D:
    """torch_radon/volumes.py:5-33.  Only the default uniform volume (centre 0, voxel size 1) is supported."""

    def __init__(self, center=(0.0, 0.0), voxel_size=(1.0, 1.0)):
        self.height = -1
        self.width = -1
        self.center = center
        self.voxel_size = voxel_size

    def num_dimensions(self):
        return 2

    def has_size(self):
        return self.height > 0 and self.width > 0

    def set_size(self, height, width):
        self.height = height
        self.width = width

    def _is_default(self):
        try:
            return tuple(float(v) for v in self.center) == (0.0, 0.0) and tuple(float(v) for v in self.voxel_size) == (1.0, 1.0)
        except TypeError:
            return False


def _not_on_path(what):
    raise NotImplementedError(f"torch_radon.{what} is not on MR_SLAM's loop-closure path and is not provided by the "
                              "MI355X drop-in (only ParallelBeam.forward is)")


class ParallelBeam:
    """ParallelBeam(det_count, angles, det_spacing=1.0, volume=None) -- radon.py:159-167."""

    def __init__(self, det_count, angles, det_spacing=1.0, volume=None):
        if volume is None:
            volume = Volume2D()
        if not isinstance(volume, Volume2D) or not volume._is_default():
            raise NotImplementedError("only the default uniform Volume2D() is supported")
        if isinstance(angles, tuple) and len(angles) == 3:      # radon.py:29-32
            angles = np.linspace(angles[0], angles[1], angles[2], endpoint=False)
        if not isinstance(angles, torch.Tensor):                 # radon.py:35-36
            angles = torch.FloatTensor(np.asarray(angles, dtype=np.float32))
        self.angles = angles
        self.volume = volume
        self.det_count = int(det_count)
        self.det_spacing = float(det_spacing)
        self._angles_host = angles.detach().cpu().numpy().astype(np.float32)
        self._plans = {}

    def forward(self, x, angles=None, exec_cfg=None):
        if angles is not None:
            raise NotImplementedError("per-call angles are not supported; build another ParallelBeam")
        if not x.is_cuda:                                        # pytorch.cpp:16-20 (TORCH_CHECK)
            raise RuntimeError("Input tensor must be on a GPU device")
        x = x.contiguous().float()
        H, W = x.shape[-2:]
        self.volume.set_size(H, W)                               # radon.py:76-77
        key = (x.device.index or 0, H, W)
        if key not in self._plans:
            self._plans[key] = _ring.RadonPlan(self.det_count, self._angles_host, self.det_spacing, H, W, key[0])
        lead = x.shape[:-2]
        sino, _ = self._plans[key].forward(x.reshape(-1, H, W))
        return sino.reshape(*lead, self._angles_host.size, self.det_count)

    def backward(self, *a, **k):
        _not_on_path("ParallelBeam.backward")

    backprojection = backward

    def filter_sinogram(self, *a, **k):
        _not_on_path("ParallelBeam.filter_sinogram")


class Radon(ParallelBeam):
    """Deprecated constructor Radon(resolution, angles, det_count=-1, det_spacing=1.0, clip_to_circle=False)
    (radon.py:253-269), mapped onto the parallel-beam geometry it describes."""

    def __init__(self, resolution, angles, det_count=-1, det_spacing=1.0, clip_to_circle=False):
        warnings.warn("Radon() class is deprecated, use ParallelBeam instead", DeprecationWarning)
        if det_count <= 0:
            det_count = resolution
        super().__init__(det_count, angles, det_spacing)


class RadonFanbeam:
    def __init__(self, *a, **k):
        _not_on_path("RadonFanbeam")


class FanBeam:
    def __init__(self, *a, **k):
        _not_on_path("FanBeam")


class ConeBeam:
    def __init__(self, *a, **k):
        _not_on_path("ConeBeam")


class Volume3D:
    def __init__(self, *a, **k):
        _not_on_path("Volume3D")
