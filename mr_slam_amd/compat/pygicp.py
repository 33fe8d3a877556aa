"""Drop-in for the part of `pygicp` (fast_gicp's Python binding) MR_SLAM uses
(LoopDetection/src/RING_ros/main_RING.py:81-104, disco_ros/main.py:174-197, main_SC.py:108-131):
downsample(points, resolution) and FastGICP with set_input_target / set_input_source /
set_num_threads / set_max_correspondence_distance / align / get_fitness_score /
get_final_transformation.  Points are float64 [N,3] numpy arrays like upstream src/python/main.cpp."""
import numpy as np

from .. import gicp as _gicp


def downsample(points, resolution, approximate=True, device="cuda:0"):
    """pygicp.downsample(points, resolution) -> float64 [m,3].

    approximate=True (default) is upstream's filter: pcl::ApproximateVoxelGrid (512-entry hash history streamed in input
    order, float centroids, flush order) on the GPU (mrs_voxel_downsample_approx), bit-identical to the sequential filter
    as restated in oracle/voxel_oracle.c.  approximate=False is the exact voxel-grid centroid filter
    (mrs_voxel_downsample, one centroid per occupied voxel)."""
    import torch
    from .. import preprocess
    p = torch.from_numpy(np.ascontiguousarray(np.asarray(points, dtype=np.float64)[:, :3])).to(device)
    if p.shape[0] == 0:
        return np.zeros((0, 3), np.float64)
    out = preprocess.approx_voxel_grid(p, float(resolution)) if approximate else preprocess.voxel_down_sample(p, float(resolution))
    return out.cpu().numpy()


class FastGICP:
    def __init__(self):
        self._b = _gicp.GicpBatch(1)
        self._src = self._tgt = None
        self._final = np.eye(4)

    def set_input_target(self, pts):
        self._tgt = np.asarray(pts, dtype=np.float64)[:, :3]
        self._b.set_targets([self._tgt])

    def set_input_source(self, pts):
        self._src = np.asarray(pts, dtype=np.float64)[:, :3]
        self._b.set_sources([self._src])

    def set_num_threads(self, n):       # OpenMP width of the CPU reference: meaningless here
        pass

    def set_max_correspondence_distance(self, d):
        self._b.set_params(max_correspondence_distance=float(d))

    def set_correspondence_randomness(self, k):
        self._b.set_params(k_correspondences=int(k))

    def set_max_iterations(self, n):
        self._b.set_params(max_iterations=int(n))

    def set_transformation_epsilon(self, e):
        self._b.set_params(transformation_epsilon=float(e))

    def set_rotation_epsilon(self, e):
        self._b.set_params(rotation_epsilon=float(e))

    def align(self, initial_guess=np.eye(4)):
        T, conv, its = self._b.align(np.asarray(initial_guess, dtype=np.float64)[None])
        self._final, self._conv, self._its = T[0], bool(conv[0]), int(its[0])
        return self._final

    def has_converged(self):
        return self._conv

    def get_final_transformation(self):
        return self._final

    def get_fitness_score(self, max_range=np.finfo(np.float64).max):
        return float(self._b.fitness(self._final[None], float(max_range))[0])

    def get_final_hessian(self):
        return self._b.hessian[0].reshape(6, 6)
