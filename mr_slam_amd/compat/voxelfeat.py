"""Drop-in for the reference module `voxelfeat` (feature BEV; feature extractor = row N1).

Mirrors generate_bev_pointfeat_cython/wrapper.pyx:17-59."""
from ._bevshim import HostTransformer


class GPUTransformer(HostTransformer):
    """GPUTransformer(point[F*n] channel-major, size, max_length, max_height, num_x, num_y,
    num_height, featsize); retreive() -> float32[num_x*num_y*num_height*featsize]."""
    _fn = "mrs_bev_feat_host"

    def _planes_for(self, last):
        return int(last)

    def _out_size(self):
        c = self._cfg
        return c.n0 * c.n1 * c.num_height * c.enough_large


class GPUFeatureExtractor:
    """GPUFeatureExtractor(point[3*size] row-major, size, featsize=13, k, neighbors_indices int32[size*k],
    eigens float32[size*5]); get_features() -> float32[size*13]
    (generate_bev_pointfeat_cython/wrapper.pyx:43-59)."""

    def __init__(self, point, size, featsize, k, neighbors_indices, eigens):
        import numpy as np
        if int(featsize) != 13:
            raise ValueError("featsize must be 13")
        self._p = np.ascontiguousarray(point, dtype=np.float32)
        self._k = np.ascontiguousarray(neighbors_indices, dtype=np.int32)
        self._e = np.ascontiguousarray(eigens, dtype=np.float32)
        self._n, self._kk = int(size), int(k)

    def get_features(self):
        import numpy as np
        from .. import _lib
        out = np.zeros(self._n * 13, dtype=np.float32)
        _lib.check(_lib.load().mrs_pointfeat_from_neighbors_host(_lib.ctx(0), _lib.ptr(self._p), self._n, self._kk,
                                                                 _lib.ptr(self._k), _lib.ptr(self._e), _lib.ptr(out)))
        return out
