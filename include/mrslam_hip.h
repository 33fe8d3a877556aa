/*
 * mrslam_hip.h -- C ABI of libmrslam_hip.so: the MI355X (gfx950) implementation of
 * MR_SLAM's loop-closure hot path (BEV rasterisers, Radon sinogram, FFT correlation,
 * GICP refinement).  Plain pointers and sizes only; no torch / Eigen / PCL types.
 *
 * Conventions
 *   - every entry point returns an int status (MRS_OK == 0); nothing here ever calls
 *     exit() (the reference's torch-radon does: include/utils.h:38-48) or throws;
 *   - `d_` arguments are DEVICE pointers, `h_` arguments are HOST pointers;
 *   - `stream` is a hipStream_t passed as void* (NULL = the legacy default stream);
 *     device-pointer entry points only ENQUEUE work on it, they do not synchronise;
 *   - a context (mrs_ctx) is bound to one device; entry points are re-entrant and may
 *     be called concurrently from several host threads (the reference modules are
 *     entered concurrently by rospy callback threads: main_RING.py:241-390);
 *   - batches are "ragged": scan b owns points [offsets[b], offsets[b+1]) of the packed
 *     cloud; each scan keeps the reference's own SoA layout [x0..xn-1,y0..,z0..]
 *     starting at float 3*offsets[b] (util.py:177 `pc.transpose().flatten()`).
 *
 * Each function names the reference interface it replaces (file:line under
 * /root/reference).  INTEGRATION.md shows the reference-side bindings.
 */
#ifndef MRSLAM_HIP_H
#define MRSLAM_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MRS_ABI_VERSION 1

enum mrs_status {
    MRS_OK = 0,
    MRS_ERR_ARG = 1,         /* bad argument (null pointer, non-positive size ...)      */
    MRS_ERR_HIP = 2,         /* a HIP runtime call failed; see mrs_last_error()          */
    MRS_ERR_UNSUPPORTED = 3, /* configuration outside what this build implements         */
    MRS_ERR_NO_DEVICE = 4,   /* no gfx950-class device visible                           */
    MRS_ERR_NOT_CONVERGED = 5
};

typedef struct mrs_ctx mrs_ctx;
typedef void* mrs_stream;

int mrs_abi_version(void);
const char* mrs_status_str(int status);
/* thread-local text of the last failure in the calling thread ("" if none) */
const char* mrs_last_error(void);

/* Bind a context to HIP device `device`.  Fails with MRS_ERR_NO_DEVICE when no GPU is
 * visible: there is NO CPU fallback anywhere behind this ABI. */
int mrs_ctx_create(int device, mrs_ctx** out_ctx);
int mrs_ctx_destroy(mrs_ctx* ctx);
int mrs_ctx_device(const mrs_ctx* ctx);

/* ------------------------------------------------------------------------------------
 * BEV rasterisers
 * ---------------------------------------------------------------------------------- */

/* Grid description shared by the three rasterisers.  Field meaning follows the reference
 * constructors: GPUTransformer(point, size, max_length, max_height, num_ring|num_x,
 * num_sector|num_y, num_height, enough_large|featsize)
 *   polar : disco_ros/tools/multi-layer-polar-cpu/cython/gputransform.pyx:19-30
 *   cart  : generate_bev_cython_binary/wrapper.pyx:18-29
 *   feat  : generate_bev_pointfeat_cython/wrapper.pyx:22-33 */
typedef struct mrs_bev_cfg {
    int32_t max_length;   /* half extent in x/y (cart) or max range (polar); reference int  */
    int32_t max_height;   /* half extent in z                                               */
    int32_t n0;           /* polar: num_ring    cart/feat: num_x                            */
    int32_t n1;           /* polar: num_sector  cart/feat: num_y                            */
    int32_t num_height;   /* height layers                                                  */
    int32_t enough_large; /* polar/cart: points kept per cell (cart ignores it, like the
                             reference); feat: featsize F                                   */
} mrs_bev_cfg;

/* output layouts */
enum mrs_bev_out {
    /* the reference `retreive()` array, bit for bit:
     *   polar/cart: float[3 * n0*n1*num_height * enough_large]  (x, y, value) triplets
     *   feat      : float[n0*n1*num_height * F]                 F interleaved channels  */
    MRS_BEV_OUT_REFERENCE = 0,
    /* only what the callers consume (util.py:186-187, disco_ros/main.py:121-123):
     *   polar/cart: float[n0*n1*num_height]   channel 2 (occupancy / max z), slab 0
     *   feat      : float[(F-3) * n0*n1*num_height] planar, channels 3..F-1 (util.py:231-240) */
    MRS_BEV_OUT_COMPACT = 1
};

/* Value written for a point the reference would mishandle (NaN, int overflow). */
#define MRS_BEV_DROPPED INT32_MIN

/* A1 -- per-point ring / sector / height of ONE scan.
 * Replaces GPUTransformer::transform() -> point2gridmap,
 * multi-layer-polar-cpu/cython/src/kernel.cpp:40-77 (GPU twin ...-gpu/.../kernel.cu:41-80).
 * Bit-exact with the CPU reference on every point it handles without UB. */
int mrs_bev_polar_indices(mrs_ctx* ctx, const float* d_xyz_soa, int32_t n, const mrs_bev_cfg* cfg,
                          int32_t* d_ring, int32_t* d_sector, int32_t* d_height, mrs_stream stream);

/* A3 -- per-point x / y / height cell of ONE scan.
 * Replaces point2gridmap, generate_bev_cython_binary/src/kernel.cu:14-61. */
int mrs_bev_cart_indices(mrs_ctx* ctx, const float* d_xyz_soa, int32_t n, const mrs_bev_cfg* cfg,
                         int32_t* d_ix, int32_t* d_iy, int32_t* d_ih, mrs_stream stream);

/* A1+A2 fused, batched: polar multi-layer occupancy BEV of `batch` scans.
 * Replaces transform()+retreive(), multi-layer-polar-cpu/cython/src/manager.cpp:36-58.
 * d_offsets: int64[batch+1] (device).  d_out: batch * (layout size) floats; the kernel
 * writes every element (no pre-zeroing needed). */
int mrs_bev_polar_batch(mrs_ctx* ctx, const float* d_xyz, const int64_t* d_offsets, int32_t batch,
                        const mrs_bev_cfg* cfg, int32_t out_layout, float* d_out, mrs_stream stream);

/* A3+A4 fused, batched: Cartesian max-z BEV.
 * Replaces transform()+retreive(), generate_bev_cython_binary/src/manager.cu:45-91. */
int mrs_bev_cart_batch(mrs_ctx* ctx, const float* d_xyz, const int64_t* d_offsets, int32_t batch,
                       const mrs_bev_cfg* cfg, int32_t out_layout, float* d_out, mrs_stream stream);

/* A5, batched: per-cell per-channel maximum of F channel-major planes [F*n] per scan
 * (planes 0..2 are x,y,z); scan b starts at float F*offsets[b].
 * Replaces point2gridmap + retreive, generate_bev_pointfeat_cython/src/kernel.cu:106-164,
 * src/manager.cu:54-63.  Deterministic true maximum (the reference races). */
int mrs_bev_feat_batch(mrs_ctx* ctx, const float* d_pts, const int64_t* d_offsets, int32_t batch,
                       const mrs_bev_cfg* cfg, int32_t out_layout, float* d_out, mrs_stream stream);

/* Host-buffer forms with the reference's calling convention (caller-owned numpy buffers in,
 * zero-initialised float array out; gputransform.pyx:32-39, wrapper.pyx:31-39).  They copy
 * H2D, run the kernels above, copy D2H and synchronise.  Layout MRS_BEV_OUT_REFERENCE. */
int mrs_bev_polar_host(mrs_ctx* ctx, const float* h_xyz_soa, int32_t n, const mrs_bev_cfg* cfg, float* h_out);
int mrs_bev_cart_host(mrs_ctx* ctx, const float* h_xyz_soa, int32_t n, const mrs_bev_cfg* cfg, float* h_out);
int mrs_bev_feat_host(mrs_ctx* ctx, const float* h_pts_cm, int32_t n, const mrs_bev_cfg* cfg, float* h_out);

#ifdef __cplusplus
}
#endif
#endif /* MRSLAM_HIP_H */
