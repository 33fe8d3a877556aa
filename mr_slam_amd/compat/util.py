"""`util`-named mirrors of the functions the LoopDetection nodes take from RING_ros/util.py with `from util import *`
(main_RING.py:8, main_RINGplusplus.py:8): registered as module `util` by `mr_slam_amd.compat.install(node=True)`, so a node can run
without the reference's util.py (and its matplotlib / sklearn / skimage / torchvision imports).  Same names, argument order and return
values; the work runs on the GPU through the C ABI (mr_slam_amd/ring.py, preprocess.py, node.py)."""
import numpy as np
import torch

from .. import preprocess as _pre
from ..ring import (fast_corr, fast_corr_RINGplusplus, forward_row_fft, generate_RING, generate_RINGplusplus,   # noqa: F401
                    rotate_bev, solve_translation, solve_translation_bev)

device = torch.device("cuda:0")                                    # util.py:23 (there is no CPU path behind these names)


def euler2rot(roll, pitch, yaw):
    """util.py:51-75: R = Rz(yaw) Ry(pitch) Rx(roll)"""
    cr, sr, cp, sp, cy, sy = np.cos(roll), np.sin(roll), np.cos(pitch), np.sin(pitch), np.cos(yaw), np.sin(yaw)
    Rx = np.array([[1, 0, 0], [0, cr, -sr], [0, sr, cr]])
    Ry = np.array([[cp, 0, sp], [0, 1, 0], [-sp, 0, cp]])
    Rz = np.array([[cy, -sy, 0], [sy, cy, 0], [0, 0, 1]])
    return Rz @ Ry @ Rx


def getSE3(x, y, yaw):
    """util.py:78-82: planar pose as a homogeneous 4 x 4"""
    T = np.eye(4)
    T[:3, :3] = euler2rot(0, 0, yaw)
    T[0, 3], T[1, 3] = x, y
    return T


def calculate_row_shift(shift, num_ring=120):
    """util.py:378-385 (the node passes cfg.num_ring = 120 implicitly)"""
    return -shift if shift < num_ring // 2 else shift - num_ring


def load_pc_infer(pc):
    """util.py:91-112: crop to |x|, |y| < 70 m, 0 < z < 30 m and scale to the unit box; Nx3 float32"""
    pc = np.array(pc, dtype=np.float32)
    keep = (np.abs(pc[..., 0]) < 70.) & (np.abs(pc[..., 1]) < 70.) & (pc[..., 2] < 30.) & (pc[..., 2] > 0.)
    hits = pc[keep]
    hits[..., 0] = hits[..., 0] / 70.
    hits[..., 1] = hits[..., 1] / 70.
    hits[..., 2] = hits[..., 2] / 30.
    return hits


def robotid_to_key(robotid):
    """util.py:253-260 (prints like the reference)"""
    print("robotid: ", robotid, " outkey: ", 97 + robotid)
    return _pre.robotid_to_key(robotid)
