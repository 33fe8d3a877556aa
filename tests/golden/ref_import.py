"""Imports the REFERENCE's own Python modules (RING_ros/util.py, disco_ros/models/DiSCO.py, functions of
disco_ros/main.py and generate_bev_pointfeat_cython/test.py) in this container, where none of their native or
third-party dependencies exist.  TEST INFRASTRUCTURE ONLY: used by the golden-vector generators under tests/golden/
and by tests/test_reference_loads_dropin.py.  /root/reference is read, never copied.

What is replaced, and by what:
  * voxelocc / voxelfeat / gputransform / torch_radon  -- the hot-path boundary itself.  `backend="oracle"` registers
    stand-ins with the reference modules' exact constructor / method signatures that run the CPU checkers of
    oracle/ (the reference-built oracle/_ref libraries where they exist), so that the reference's generate_RING /
    generate_RINGplusplus run end to end on the CPU; `backend="dropin"` registers mr_slam_amd.compat (the product)
    instead, which is what a user of the reference would do.
  * torchvision.transforms.functional.normalize -- torchvision is not installed; normalize(t, mean, std) is
    (t - mean) / std and raises ValueError when std == 0 (torchvision/transforms/_functional_tensor.py: normalize).
    `rotate` is NOT provided (it is torchvision's own code, not the reference's): calling it raises.
  * skimage.morphology, knn_cuda -- imported by the reference, never called on the paths used here: empty modules.
"""
import ast
import importlib
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
# the reference tree where it exists (this container); on a GPU box the few Python files staged by tools/stage_reference_py.py
# (tests/_refpy/, git-ignored scratch, same relative layout) so that the reference's own functions can run through the drop-in there
REF = os.environ.get("MRSLAM_REFERENCE", "/root/reference")
if not os.path.isdir(os.path.join(REF, "LoopDetection")) and os.path.isdir(os.path.join(ROOT, "tests", "_refpy", "LoopDetection")):
    REF = os.path.join(ROOT, "tests", "_refpy")
RING_ROS = os.path.join(REF, "LoopDetection", "src", "RING_ros")
DISCO_ROS = os.path.join(REF, "LoopDetection", "src", "disco_ros")
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def available():
    return os.path.isdir(RING_ROS)


def _normalize(tensor, mean, std, inplace=False):
    import torch
    mean = torch.as_tensor(mean, dtype=tensor.dtype, device=tensor.device)
    std = torch.as_tensor(std, dtype=tensor.dtype, device=tensor.device)
    if (std == 0).any():
        raise ValueError(f"std evaluated to zero after conversion to {tensor.dtype}, leading to division by zero.")
    return (tensor - mean) / std


def _rotate(*a, **k):
    raise NotImplementedError("torchvision.transforms.functional.rotate is not available in this container")


def _third_party_stubs():
    tv = types.ModuleType("torchvision")
    tvt = types.ModuleType("torchvision.transforms")
    tvf = types.ModuleType("torchvision.transforms.functional")
    tvu = types.ModuleType("torchvision.utils")
    tvf.normalize = _normalize
    tvf.rotate = _rotate
    tvt.functional = tvf
    tvt.ToPILImage = lambda *a, **k: None
    tv.transforms = tvt
    tv.utils = tvu
    sk = types.ModuleType("skimage")
    sk.morphology = types.ModuleType("skimage.morphology")
    sk.transform = types.ModuleType("skimage.transform")
    sk.transform.radon = None
    return {"torchvision": tv, "torchvision.transforms": tvt, "torchvision.transforms.functional": tvf,
            "torchvision.utils": tvu, "skimage": sk, "skimage.morphology": sk.morphology,
            "skimage.transform": sk.transform, "knn_cuda": types.ModuleType("knn_cuda")}


# ---------------------------------------------------------------- oracle-backed stand-ins for the boundary modules
def _oracle_modules():
    import torch
    from oracle import pyoracle as O

    class _T:
        """GPUTransformer(point, size, max_length, max_height, n0, n1, num_height, last) + transform() + retreive()
        (wrapper.pyx / gputransform.pyx signatures)."""
        def __init__(self, point, size, max_length, max_height, n0, n1, num_height, last):
            assert point.dtype == np.float32 and point.ndim == 1 and point.flags["C_CONTIGUOUS"]
            self.a = (point, int(size), int(max_length), int(max_height), int(n0), int(n1), int(num_height), int(last))

        def transform(self):
            pass

    class CartT(_T):
        def retreive(self):
            p, n, ml, mh, nx, ny, nh, el = self.a
            f = O.ref_bev_cart if O.ref_lib("cart") is not None else (lambda *a: O.bev_cart(*a[:-1]))
            return f(p[:3 * n], ml, mh, nx, ny, nh, el)

    class PolarT(_T):
        def retreive(self):
            p, n, ml, mh, r, s, h, el = self.a
            f = O.ref_bev_polar if O.ref_polar() is not None else O.bev_polar
            return f(p[:3 * n], ml, mh, r, s, h, el)

    class FeatT(_T):
        def retreive(self):
            p, n, ml, mh, nx, ny, nh, F = self.a
            f = O.ref_bev_feat if O.ref_lib("feat") is not None else O.bev_feat
            return f(p[:F * n], F, ml, mh, nx, ny, nh)

    class FeatX:
        def __init__(self, point, size, featsize, k, neighbors_indices, eigens):
            self.a = (point, int(size), int(featsize), int(k), neighbors_indices, eigens)

        def get_features(self):
            p, n, F, k, nb, eg = self.a
            assert F == 13
            if O.ref_lib("feat") is not None:
                return O.ref_point_features(np.asarray(p, np.float32).reshape(n, 3), np.asarray(nb, np.int32).reshape(n, k),
                                            np.asarray(eg, np.float32).reshape(n, 5)).reshape(-1)
            from oracle import pointfeat_oracle as PF
            return PF.calculate_features(np.asarray(p, np.float32).reshape(n, 3), np.asarray(nb).reshape(n, k),
                                         np.asarray(eg, np.float32).reshape(n, 5)).reshape(-1)

    class ParallelBeam:
        """torch_radon.ParallelBeam(det_count, angles, det_spacing=1.0, volume=None).forward(x) on the CPU checker
        (oracle/radon_oracle.c; pinned to the reference's analytic test, tests/test_oracle_radon_corr.py)."""
        def __init__(self, det_count, angles, det_spacing=1.0, volume=None):
            self.det, self.sp = int(det_count), float(det_spacing)
            self.ang = np.asarray(angles.detach().cpu().numpy() if isinstance(angles, torch.Tensor) else angles, np.float32)

        def forward(self, x):
            a = x.detach().cpu().numpy().astype(np.float32)
            lead = a.shape[:-2]
            s = O.radon_parallel(a.reshape((-1,) + a.shape[-2:]), self.ang, self.det, self.sp)
            return torch.from_numpy(s.reshape(lead + s.shape[-2:])).to(x.device)

    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        return m

    def _absent(*a, **k):
        raise NotImplementedError("not used by MR_SLAM")

    return {"voxelocc": mod("voxelocc", GPUTransformer=CartT),
            "gputransform": mod("gputransform", GPUTransformer=PolarT),
            "voxelfeat": mod("voxelfeat", GPUTransformer=FeatT, GPUFeatureExtractor=FeatX),
            "torch_radon": mod("torch_radon", ParallelBeam=ParallelBeam, Radon=ParallelBeam, RadonFanbeam=_absent)}


class reference_modules:
    """Context manager: `with reference_modules("oracle") as ref: ref.util.fast_corr(...)`.
    Restores sys.modules / sys.path afterwards so the stand-ins never leak into other tests."""

    def __init__(self, backend="oracle"):
        assert backend in ("oracle", "dropin")
        self.backend = backend

    def __enter__(self):
        if not available():
            raise RuntimeError("reference tree not present")
        self._saved_modules = dict(sys.modules)
        self._saved_path = list(sys.path)
        for n in ("util", "config", "DiSCO", "models", "models.DiSCO"):
            sys.modules.pop(n, None)
        sys.modules.update(_third_party_stubs())
        if self.backend == "oracle":
            sys.modules.update(_oracle_modules())
        else:
            from mr_slam_amd import compat
            for n in ("voxelocc", "voxelfeat", "gputransform", "torch_radon", "pygicp"):
                sys.modules.pop(n, None)
            compat.install()
        return self

    _OURS = ("util", "config", "DiSCO", "models", "voxelocc", "voxelfeat", "gputransform", "torch_radon", "pygicp",
             "torchvision", "torchvision.transforms", "torchvision.transforms.functional", "torchvision.utils",
             "skimage", "skimage.morphology", "skimage.transform", "knn_cuda")

    def __exit__(self, *exc):
        # drop only what this context registered or the reference pulled in under these names (never torch's own
        # late-imported submodules), then put back whatever was there before
        for k in self._OURS:
            sys.modules.pop(k, None)
            if k in self._saved_modules:
                sys.modules[k] = self._saved_modules[k]
        sys.path[:] = self._saved_path
        return False

    # -- the reference modules themselves, imported from where they lie
    @property
    def util(self):
        """LoopDetection/src/RING_ros/util.py (with its config.py)."""
        sys.modules.pop("config", None)
        sys.path.insert(0, RING_ROS)
        try:
            return importlib.import_module("util")
        finally:
            sys.path.remove(RING_ROS)

    @property
    def disco(self):
        """LoopDetection/src/disco_ros/models/DiSCO.py (with disco_ros/config.py)."""
        for n in ("config", "util"):
            sys.modules.pop(n, None)
        sys.path.insert(0, os.path.join(DISCO_ROS, "models"))
        sys.path.insert(0, DISCO_ROS)
        try:
            return importlib.import_module("DiSCO")
        finally:
            sys.path.remove(DISCO_ROS)
            sys.path.remove(os.path.join(DISCO_ROS, "models"))

    @staticmethod
    def functions_of(path, names, namespace):
        """exec only the named top-level functions of a reference file that cannot be imported as a whole (rospy,
        CUDA-only imports): disco_ros/main.py:phase_corr, generate_bev_pointfeat_cython/test.py:calculate_features."""
        tree = ast.parse(open(path).read(), filename=path)
        keep = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in names]
        assert len(keep) == len(names), (path, names)
        code = compile(ast.Module(body=keep, type_ignores=[]), path, "exec")
        ns = dict(namespace)
        exec(code, ns)
        return ns
