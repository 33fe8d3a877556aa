# cython: language_level=3
# gputransform -- compiled extension module with the public surface of the reference's
# LoopDetection/src/disco_ros/tools/multi-layer-polar-cpu/cython/gputransform.pyx:14-39 (class GPUTransformer, transform(), retreive()),
# bound to libmrslam_hip.so through the C ABI (mrs_bev_polar_host) instead of the CUDA manager class.
import numpy as np
cimport numpy as np
from _mrslam cimport *

assert sizeof(int) == sizeof(np.int32_t)

cdef mrs_ctx* _ctx = NULL

cdef mrs_ctx* _context() except NULL:
    global _ctx
    cdef int st
    if _ctx == NULL:
        st = mrs_ctx_create(0, &_ctx)
        if st != 0:
            _ctx = NULL
            raise RuntimeError("%s: %s" % (mrs_status_str(st).decode(), mrs_last_error().decode()))
    return _ctx


cdef class GPUTransformer:
    cdef np.ndarray point           # the reference keeps the caller's buffer alive through its raw pointer too
    cdef int size
    cdef mrs_bev_cfg cfg
    cdef int grid_size

    def __cinit__(self, np.ndarray[float, ndim=1, mode = "c"] point not None,
                    int size, int max_length, int max_height, int num_ring, int num_sector, int num_height, int enough_large):
        if point.shape[0] < 3 * size:
            raise ValueError("point array shorter than 3 * size")
        self.point = point
        self.size = size
        self.grid_size = num_ring * num_sector * num_height
        self.cfg.max_length = max_length
        self.cfg.max_height = max_height
        self.cfg.n0 = num_ring
        self.cfg.n1 = num_sector
        self.cfg.num_height = num_height
        self.cfg.enough_large = enough_large

    def transform(self):
        _context()                  # the rasteriser is one fused pass inside retreive(); fail here if no GPU

    def retreive(self):
        cdef np.ndarray[float, ndim=1, mode = "c"] point_transformed = np.zeros(self.grid_size * self.cfg.enough_large * 3, dtype=np.float32)
        cdef np.ndarray[float, ndim=1, mode = "c"] p = self.point
        cdef mrs_ctx* c = _context()
        cdef int st
        with nogil:                 # rospy callbacks enter concurrently (SURVEY.md 8(b)): do not hold the GIL on the GPU call
            st = mrs_bev_polar_host(c, &p[0], self.size, &self.cfg, &point_transformed[0])
        if st != 0:
            raise RuntimeError("%s: %s" % (mrs_status_str(st).decode(), mrs_last_error().decode()))
        return point_transformed
