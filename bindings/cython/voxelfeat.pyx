# cython: language_level=3
# voxelfeat -- compiled extension module with the public surface of the reference's
# LoopDetection/generate_bev_pointfeat_cython/wrapper.pyx:17-59 (GPUTransformer + GPUFeatureExtractor), bound to
# libmrslam_hip.so through the C ABI (mrs_bev_feat_host, mrs_pointfeat_from_neighbors_host).
import numpy as np
cimport numpy as np
from libc.stdint cimport int32_t
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
    cdef np.ndarray point
    cdef int size
    cdef int grid_size
    cdef mrs_bev_cfg cfg

    def __cinit__(self, np.ndarray[float, ndim=1, mode = "c"] point not None,
                    int size, int max_length, int max_height, int num_x, int num_y, int num_height, int featsize):
        if point.shape[0] < featsize * size:
            raise ValueError("point array shorter than featsize * size")
        self.point = point
        self.size = size
        self.grid_size = num_x * num_y * num_height * featsize
        self.cfg.max_length = max_length
        self.cfg.max_height = max_height
        self.cfg.n0 = num_x
        self.cfg.n1 = num_y
        self.cfg.num_height = num_height
        self.cfg.enough_large = featsize

    def transform(self):
        _context()

    def retreive(self):
        cdef np.ndarray[float, ndim=1, mode = "c"] point_out = np.zeros(self.grid_size, dtype=np.float32)
        cdef np.ndarray[float, ndim=1, mode = "c"] p = self.point
        cdef mrs_ctx* c = _context()
        cdef int st
        with nogil:
            st = mrs_bev_feat_host(c, &p[0], self.size, &self.cfg, &point_out[0])
        if st != 0:
            raise RuntimeError("%s: %s" % (mrs_status_str(st).decode(), mrs_last_error().decode()))
        return point_out


cdef class GPUFeatureExtractor:
    cdef np.ndarray point, neighbors, eigens
    cdef int size, k
    cdef int featmapsize

    def __cinit__(self, np.ndarray[float, ndim=1, mode = "c"] point not None,
                    int size, int featsize, int k,
                    np.ndarray[int, ndim=1, mode = "c"] neighbors_indices not None,
                    np.ndarray[float, ndim=1, mode = "c"] eigens not None,):
        if featsize != 13:
            raise ValueError("featsize must be 13")
        if point.shape[0] < 3 * size or neighbors_indices.shape[0] < size * k or eigens.shape[0] < 5 * size:
            raise ValueError("input arrays shorter than size requires")
        self.point, self.neighbors, self.eigens = point, neighbors_indices, eigens
        self.size, self.k = size, k
        self.featmapsize = featsize * size

    def get_features(self):
        cdef np.ndarray[float, ndim=1, mode = "c"] feature = np.zeros(self.featmapsize, dtype=np.float32)
        cdef np.ndarray[float, ndim=1, mode = "c"] p = self.point
        cdef np.ndarray[int, ndim=1, mode = "c"] nb = self.neighbors
        cdef np.ndarray[float, ndim=1, mode = "c"] eg = self.eigens
        cdef mrs_ctx* c = _context()
        cdef int st
        with nogil:
            st = mrs_pointfeat_from_neighbors_host(c, &p[0], self.size, self.k, <const int32_t*>&nb[0], &eg[0], &feature[0])
        if st != 0:
            raise RuntimeError("%s: %s" % (mrs_status_str(st).decode(), mrs_last_error().decode()))
        return feature
