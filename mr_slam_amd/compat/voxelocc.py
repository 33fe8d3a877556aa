"""Drop-in for the reference module `voxelocc` (Cartesian max-z BEV).

Mirrors generate_bev_cython_binary/wrapper.pyx:13-39: GPUTransformer(point, size,
max_length, max_height, num_x, num_y, num_height, enough_large)."""
from ._bevshim import HostTransformer


class GPUTransformer(HostTransformer):
    _fn = "mrs_bev_cart_host"

    def _out_size(self):
        c = self._cfg  # the reference allocates the enough_large factor but fills slab 0 only
        return 3 * c.n0 * c.n1 * c.num_height * c.enough_large
