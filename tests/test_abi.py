"""CPU suite: the C-ABI library loads and exports every symbol include/mrslam_hip.h declares."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "mrslam_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(mrs_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from mr_slam_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    lib = C.CDLL(_lib.LIB_PATH)
    names = _declared_symbols()
    assert len(names) >= 10
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing
    assert lib.mrs_abi_version() == 1


def test_no_cpu_fallback_without_gpu():
    """Without a GPU the product path must fail loudly (MRS_ERR_NO_DEVICE), never compute."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from mr_slam_amd import _lib
    with pytest.raises(_lib.MrsError):
        _lib.ctx(0)
    import numpy as np
    from mr_slam_amd.compat import voxelocc
    t = voxelocc.GPUTransformer(np.zeros(30, np.float32), 10, 1, 1, 120, 120, 1, 1)
    with pytest.raises(_lib.MrsError):
        t.retreive()
    from mr_slam_amd import ring
    xyz, offs = torch.zeros(30), torch.tensor([0, 10])
    for fused in (False, True):       # host tensors are rejected, not computed on
        with pytest.raises(_lib.MrsError):
            ring.ring_descriptors(xyz, offs, fused=fused)


def test_product_never_imports_oracle():
    """oracle/ is test infrastructure: no module of the package may reference it."""
    pkg = os.path.join(ROOT, "mr_slam_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h")):
                txt = open(os.path.join(dp, f)).read()
                assert "pyoracle" not in txt and "liboracle" not in txt and "import oracle" not in txt, f
