"""Drop-in for the reference module `gputransform` (polar multi-layer BEV).

Mirrors disco_ros/tools/multi-layer-polar-{cpu,gpu}/cython/gputransform.pyx:14-39:
GPUTransformer(point, size, max_length, max_height, num_ring, num_sector, num_height,
enough_large) with .transform() and .retreive() (sic)."""
from ._bevshim import HostTransformer


class GPUTransformer(HostTransformer):
    _fn = "mrs_bev_polar_host"

    def _out_size(self):
        c = self._cfg
        return 3 * c.n0 * c.n1 * c.num_height * c.enough_large
