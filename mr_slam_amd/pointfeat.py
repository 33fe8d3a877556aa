"""RING++ point-feature front-end over the C ABI: exact kNN,
neighbourhood eigenvalues and the 13 hand-crafted features of generate_RINGplusplus
(RING_ros/util.py:204-228), batched and device resident."""
import ctypes as C

import numpy as np
import torch

from . import _lib

K_NUM = 30     # RING_ros/config.py:4, util.py:211


def point_features(points, offsets, k=K_NUM, want=("features",)):
    """points: float32 device tensor [N, s>=3] (packed clouds), offsets: int64 [B+1] (host array).
    want: any of "knn", "eigens", "features", "planes".  Returns a dict of device tensors in the
    caller's point order; "planes" is the flat [9*N] channel-major input of bev.feat_bev (F = 9)."""
    assert points.is_cuda and points.dtype == torch.float32
    d = points.device.index or 0
    points = points.contiguous()
    offs = np.ascontiguousarray(offsets, dtype=np.int64)
    N = int(offs[-1])
    out = {}
    if "knn" in want:
        out["knn"] = torch.empty((N, k), dtype=torch.int32, device=points.device)
    if "eigens" in want:
        out["eigens"] = torch.empty((N, 5), dtype=torch.float32, device=points.device)
    if "features" in want:
        out["features"] = torch.empty((N, 13), dtype=torch.float32, device=points.device)
    if "planes" in want:
        out["planes"] = torch.empty(9 * N, dtype=torch.float32, device=points.device)
    g = lambda n: _lib.ptr(out[n]) if n in out else None
    _lib.check(_lib.load().mrs_pointfeat_batch(_lib.ctx(d), _lib.ptr(points), int(points.shape[1]), _lib.ptr(offs),
                                               offs.size - 1, int(k), g("knn"), g("eigens"), g("features"), g("planes"),
                                               _lib.current_stream(d)))
    return out
