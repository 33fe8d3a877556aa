"""Drop-in replacements for the reference's native extension modules.

`mr_slam_amd.compat.install()` registers them in sys.modules under the reference's names so
that `import voxelocc`, `import gputransform`, `import voxelfeat`, `import torch_radon`,
`import pygicp` in the unmodified LoopDetection nodes resolve to the HIP implementation.  `install(node=True)` adds a `util` module with the
names the nodes import from RING_ros/util.py.
"""
import importlib
import sys

_NAMES = ("gputransform", "voxelocc", "voxelfeat", "torch_radon", "pygicp")


def install(names=_NAMES, node=False):
    """node=True also registers `util` (the names the nodes pull in with `from util import *`: mr_slam_amd/compat/util.py); the candidate loop
    itself is replaced by `mr_slam_amd.node.bind_detect_loop_icp` (INTEGRATION.md 1a')."""
    for n in tuple(names) + (("util",) if node else ()):
        try:
            sys.modules[n] = importlib.import_module("mr_slam_amd.compat." + n)
        except ModuleNotFoundError:
            pass
