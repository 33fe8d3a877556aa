"""Drop-in replacements for the reference's native extension modules.

`mr_slam_amd.compat.install()` registers them in sys.modules under the reference's names so
that `import voxelocc`, `import gputransform`, `import voxelfeat`, `import torch_radon`,
`import pygicp` in the unmodified LoopDetection nodes resolve to the HIP implementation.
"""
import importlib
import sys

_NAMES = ("gputransform", "voxelocc", "voxelfeat", "torch_radon", "pygicp")


def install(names=_NAMES):
    for n in names:
        try:
            sys.modules[n] = importlib.import_module("mr_slam_amd.compat." + n)
        except ModuleNotFoundError:
            pass
