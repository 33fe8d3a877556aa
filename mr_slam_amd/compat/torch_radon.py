"""Drop-in for the part of `torch_radon` MR_SLAM uses: ParallelBeam(...).forward(x)
(torch_radon/radon.py:62-87,139-167; call sites RING_ros/util.py:192-195,241-245).
Back-projection, fan/cone beam, noise, shearlets, solvers and filtering are CT-only features
MR_SLAM never calls and are not provided."""
import numpy as np
import torch

from .. import ring as _ring


class ParallelBeam:
    def __init__(self, det_count, angles, det_spacing=1.0, volume=None):
        if volume is not None:
            raise NotImplementedError("only the default uniform Volume2D() is supported")
        if isinstance(angles, tuple) and len(angles) == 3:      # radon.py:29-32
            angles = np.linspace(angles[0], angles[1], angles[2], endpoint=False)
        if isinstance(angles, torch.Tensor):
            angles = angles.detach().cpu().numpy()
        self.angles = np.asarray(angles, dtype=np.float32)
        self.det_count = int(det_count)
        self.det_spacing = float(det_spacing)
        self._plans = {}

    def forward(self, x):
        if not x.is_cuda:                                        # pytorch.cpp:16-20 (TORCH_CHECK)
            raise RuntimeError("Input tensor must be on a GPU device")
        x = x.contiguous().float()
        H, W = x.shape[-2:]
        key = (x.device.index or 0, H, W)
        if key not in self._plans:
            self._plans[key] = _ring.RadonPlan(self.det_count, self.angles, self.det_spacing, H, W, key[0])
        lead = x.shape[:-2]
        sino, _ = self._plans[key].forward(x.reshape(-1, H, W))
        return sino.reshape(*lead, self.angles.size, self.det_count)


Radon = ParallelBeam
