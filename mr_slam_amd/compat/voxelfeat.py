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
