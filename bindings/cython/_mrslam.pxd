# Declarations of the C ABI (include/mrslam_hip.h) used by the three BEV extension modules.
from libc.stdint cimport int32_t

cdef extern from "mrslam_hip.h":
    ctypedef struct mrs_ctx:
        pass
    ctypedef struct mrs_bev_cfg:
        int32_t max_length
        int32_t max_height
        int32_t n0
        int32_t n1
        int32_t num_height
        int32_t enough_large
    int mrs_ctx_create(int device, mrs_ctx** out_ctx) nogil
    int mrs_bev_polar_host(mrs_ctx* ctx, const float* h_xyz_soa, int32_t n, const mrs_bev_cfg* cfg, float* h_out) nogil
    int mrs_bev_cart_host(mrs_ctx* ctx, const float* h_xyz_soa, int32_t n, const mrs_bev_cfg* cfg, float* h_out) nogil
    int mrs_bev_feat_host(mrs_ctx* ctx, const float* h_pts_cm, int32_t n, const mrs_bev_cfg* cfg, float* h_out) nogil
    int mrs_pointfeat_from_neighbors_host(mrs_ctx* ctx, const float* h_points, int32_t n, int32_t k, const int32_t* h_knn,
                                          const float* h_eigens, float* h_features) nogil
    const char* mrs_last_error() nogil
    const char* mrs_status_str(int status) nogil
