"""mr_slam_amd -- MI355X-native loop-closure hot path of MR_SLAM.

Host-side mirror of the reference's extension-module interfaces over one C ABI
(include/mrslam_hip.h -> libmrslam_hip.so, hand-written HIP for gfx950).
Drop-in module names live in mr_slam_amd.compat (gputransform, voxelocc, voxelfeat,
torch_radon, pygicp); see INTEGRATION.md.
"""
from ._lib import MrsError, load, LIB_PATH  # noqa: F401

__all__ = ["MrsError", "load", "LIB_PATH"]
