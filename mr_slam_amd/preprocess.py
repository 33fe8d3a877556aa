"""Scan pre-processing on the GPU (row N2): voxel down-sampling and the load_pc_infer crop/scale,
plus the wire-format helpers of the loop messages (host side)."""
import ctypes as C

import numpy as np
import torch

from . import _lib


def voxel_down_sample(points, voxel_size):
    """open3d voxel_down_sample (main_RING.py:257-259).  points: device tensor [n, s>=3] float32/float64.
    Returns a float64 device tensor [m,3] of voxel centroids (sorted by voxel index)."""
    assert points.is_cuda and points.dtype in (torch.float32, torch.float64)
    d = points.device.index or 0
    p = points.contiguous()
    n = p.shape[0]
    out = torch.empty((n, 3), dtype=torch.float64, device=p.device)
    cnt = C.c_int32(0)
    _lib.check(_lib.load().mrs_voxel_downsample(_lib.ctx(d), _lib.ptr(p), int(p.dtype == torch.float64), int(p.shape[1]),
                                                n, C.c_double(voxel_size), _lib.ptr(out), C.byref(cnt), _lib.current_stream(d)))
    return out[: cnt.value]


def voxel_down_sample_batch(points, raw_offsets, voxel_size):
    """voxel_down_sample for a batch of clouds in one set of launches (hash grid; each scan's voxels in order of first occurrence).
    points: device tensor [N, s>=3] float32/float64, raw_offsets: host int64 [B+1] starting at 0.
    Returns (centroids float64 device [M,3], offsets int64 device [B+1])."""
    assert points.is_cuda and points.dtype in (torch.float32, torch.float64)
    d = points.device.index or 0
    p = points.contiguous()
    h_off = np.ascontiguousarray(raw_offsets, dtype=np.int64)
    d_off = torch.from_numpy(h_off).to(p.device)
    B = h_off.size - 1
    out = torch.empty((int(h_off[-1]), 3), dtype=torch.float64, device=p.device)
    offs = torch.empty(B + 1, dtype=torch.int64, device=p.device)
    _lib.check(_lib.load().mrs_voxel_downsample_batch(_lib.ctx(d), _lib.ptr(p), int(p.dtype == torch.float64), int(p.shape[1]), _lib.ptr(d_off),
                                                      _lib.ptr(h_off), B, C.c_double(voxel_size), _lib.ptr(out), _lib.ptr(offs),
                                                      _lib.current_stream(d)))
    return out[: int(offs[-1])], offs


def approx_voxel_grid(points, leaf_size):
    """pcl::ApproximateVoxelGrid as pygicp.downsample(points, leaf) applies it (main_RING.py:84-85; row G1): points device
    tensor [n, s>=3] float32/float64 -> float64 device tensor [m,3] (float-precision centroids in the filter's flush
    order, bit-identical to the sequential filter)."""
    assert points.is_cuda and points.dtype in (torch.float32, torch.float64)
    d = points.device.index or 0
    p = points.contiguous()
    n = p.shape[0]
    out = torch.empty((n, 3), dtype=torch.float64, device=p.device)
    cnt = C.c_int32(0)
    _lib.check(_lib.load().mrs_voxel_downsample_approx(_lib.ctx(d), _lib.ptr(p), int(p.dtype == torch.float64), int(p.shape[1]),
                                                       n, C.c_double(leaf_size), _lib.ptr(out), C.byref(cnt), _lib.current_stream(d)))
    return out[: cnt.value]


def load_pc_infer_batch(points, raw_offsets):
    """util.py:91-112 for a batch: points device tensor [N, s>=3] (float32/float64) of raw clouds,
    raw_offsets host int64 [B+1].  Returns (xyz_soa float32 device [3*N] (upper bound, ragged SoA),
    offsets int64 device [B+1]) ready for bev.cart_bev / polar_bev."""
    assert points.is_cuda and points.dtype in (torch.float32, torch.float64)
    d = points.device.index or 0
    p = points.contiguous()
    h_off = np.ascontiguousarray(raw_offsets, dtype=np.int64)
    d_off = torch.from_numpy(h_off).to(p.device)
    B = h_off.size - 1
    out = torch.empty(3 * max(1, int(h_off[-1])), dtype=torch.float32, device=p.device)
    offs = torch.empty(B + 1, dtype=torch.int64, device=p.device)
    _lib.check(_lib.load().mrs_crop_scale_batch(_lib.ctx(d), _lib.ptr(p), int(p.dtype == torch.float64), int(p.shape[1]),
                                                _lib.ptr(d_off), _lib.ptr(h_off), B, _lib.ptr(out), _lib.ptr(offs),
                                                _lib.current_stream(d)))
    return out, offs


# ---- wire format of the loop messages (dislam_msgs/Loop.msg, main_RING.py:221-233, util.py:253-260) ----
def robotid_to_key(robotid):
    """util.py:253-260: chr('a' + robotid) in the top byte of a 64-bit key."""
    return (97 + int(robotid)) << 56


def loop_ids(robot_cur, idx_cur, robot_cand, idx_cand):
    """Loop.id0 / Loop.id1 (main_RING.py:221-222)."""
    return robotid_to_key(robot_cur) + idx_cur + 1, robotid_to_key(robot_cand) + idx_cand + 1


def loopinfo_line(robot_cur, idx_cur, robot_cand, idx_cand, position, quaternion_xyzw):
    """One line of ./loopinfo.txt (main_RING.py:229-233)."""
    vals = [robot_cur, idx_cur, robot_cand, idx_cand, *position, *quaternion_xyzw]
    return " ".join(str(v) for v in vals)


# ---- NCLT velodyne_sync record format (disco_ros/loading_pointclouds.py:27-68) ----
def decode_nclt(buf):
    """Raw NCLT hits: 8-byte records <HHHBB (x, y, z in 5 mm steps offset -100 m, intensity, laser).
    Returns (xyz float64 [n,3] metres, intensity uint8 [n], laser uint8 [n])."""
    rec = np.frombuffer(buf, dtype=np.dtype([("x", "<u2"), ("y", "<u2"), ("z", "<u2"), ("i", "u1"), ("l", "u1")]))
    xyz = np.stack([rec["x"], rec["y"], rec["z"]], axis=1).astype(np.float64) * 0.005 + (-100.0)
    return xyz, rec["i"].copy(), rec["l"].copy()


def load_lidar_file_nclt(path_or_bytes):
    """loading_pointclouds.py:38-68: crop |x|,|y| < 70, -20 < z < -2, drop the 5 m box around the sensor,
    scale by 70/70/20 and flip z.  Returns float64 [n,3] in acquisition order."""
    if isinstance(path_or_bytes, (bytes, bytearray, memoryview)):
        xyz, _, _ = decode_nclt(path_or_bytes)
    else:
        with open(path_or_bytes, "rb") as f:
            xyz, _, _ = decode_nclt(f.read())
    x, y, z = xyz[:, 0], xyz[:, 1], xyz[:, 2]
    keep = (np.abs(x) < 70.0) & (z > -20.0) & (z < -2.0) & (np.abs(y) < 70.0) & ~((np.abs(x) < 5.0) & (np.abs(y) < 5.0))
    hits = np.stack([x[keep] / 70.0, y[keep] / 70.0, z[keep] / 20.0], axis=1)
    hits[:, 2] = -hits[:, 2]
    return hits
