"""Shared body of the three reference-style GPUTransformer classes (host numpy in / out)."""
import ctypes as C

import numpy as np

from .. import _lib


class HostTransformer:
    """Reference calling convention: construct with a C-contiguous float32 host array, call
    transform() (kept for API parity: the rasteriser is fused, so it only validates), then
    retreive() -> zero-initialised float32 array filled like the reference's."""

    _fn = None       # name of the mrs_bev_*_host entry point
    _planes = 3

    def __init__(self, point, size, max_length, max_height, n0, n1, num_height, last):
        point = np.asarray(point)
        if point.dtype != np.float32 or point.ndim != 1 or not point.flags["C_CONTIGUOUS"]:
            raise ValueError("Buffer dtype mismatch / ndim: expected C-contiguous 1-D float32")
        self._point = point            # the reference keeps the caller's pointer alive too
        self._size = int(size)
        self._cfg = _lib.BevCfg(int(max_length), int(max_height), int(n0), int(n1),
                                int(num_height), int(last))
        if point.size < self._size * self._planes_for(last):
            raise ValueError("point array shorter than size * planes")

    def _planes_for(self, last):
        return self._planes

    def _out_size(self):
        raise NotImplementedError

    def transform(self):
        _lib.load()

    def retreive(self):
        out = np.zeros(self._out_size(), dtype=np.float32)
        fn = getattr(_lib.load(), self._fn)
        _lib.check(fn(_lib.ctx(0), _lib.ptr(self._point), self._size, C.byref(self._cfg), _lib.ptr(out)))
        return out
