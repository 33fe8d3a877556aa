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

/* ------------------------------------------------------------------------------------
 * Radon sinogram (rows R1, R2)
 * ---------------------------------------------------------------------------------- */
typedef struct mrs_radon_plan mrs_radon_plan;

/* Replaces the constructor torch_radon.ParallelBeam(det_count, angles, det_spacing, volume=None)
 * (torch_radon/radon.py:139-167) for a fixed image size; volume centre 0, voxel size 1
 * (torch_radon/volumes.py:13-21), which is how every MR_SLAM call site uses it
 * (RING_ros/util.py:192-195,241-245).  h_angles: n_angles floats on the HOST (radians);
 * their cos/sin are evaluated once, in double, when the plan is built, together with the per-ray
 * geometry table.  Images whose zero-bordered copy fits the LDS (up to ~190 x 190) take the LDS-resident
 * kernel; larger ones run the same sample loop against a padded copy in global memory. */
int mrs_radon_plan_create(mrs_ctx* ctx, const float* h_angles, int32_t n_angles, int32_t det_count,
                          float det_spacing, int32_t height, int32_t width, mrs_radon_plan** out_plan);
int mrs_radon_plan_destroy(mrs_radon_plan* plan);

/* R1 (+R2a): d_img float[batch][H][W] -> d_sino float[batch][n_angles][det_count].
 * Replaces torch_radon_cuda.forward / radon_forward_cuda<float>, parallel beam
 * (torch-radon/src/pytorch.cpp:42-81, src/forward.cu:12-178).
 * d_sino_norm (optional, may be NULL; d_sino may be NULL when it is given) receives
 * (S - mean(S)) / std(S) per image with the unbiased std: the fn.normalize(...) step of
 * generate_RING (RING_ros/util.py:197), fused so the sinogram never makes a second trip. */
int mrs_radon_forward(mrs_radon_plan* plan, const float* d_img, int32_t batch, float* d_sino,
                      float* d_sino_norm, mrs_stream stream);

/* Sinograms whose fused normalisation met a zero or non-finite standard deviation since the plan was created (or since
 * the last call with reset != 0): a blank / constant image.  The reference raises there (torchvision fn.normalize:
 * ValueError, RING_ros/util.py:197); the kernels write an all-zero normalised sinogram (finite, correlates with nothing:
 * dist = 1) and count the event so that the host mirror can raise the same error.  Synchronises the device. */
int mrs_radon_plan_degenerate_count(mrs_radon_plan* plan, int32_t reset, int32_t* out_count);

/* A3/A4 + R1 + R2a in one launch: the front half of generate_RING (RING_ros/util.py:174-197: voxelocc.GPUTransformer
 * transform() + retreive(), channel 2 -> ParallelBeam.forward -> fn.normalize) for a batch of scans.  One persistent workgroup
 * per compute unit rasterises two scans straight into the Radon kernel's LDS tile and marches the rays; the BEV image goes to
 * HBM only when d_bev is given.  Results are bit-identical to mrs_bev_cart_batch(MRS_BEV_OUT_COMPACT) followed by
 * mrs_radon_forward.  d_xyz / d_offsets as for mrs_bev_cart_batch; cfg: num_height == 1 and n0 x n1 == the plan's image
 * (H x W).  Each of d_bev float[batch][n0][n1], d_sino, d_sino_norm float[batch][n_angles][det_count] may be NULL (not
 * all).  MRS_ERR_UNSUPPORTED when the configuration needs the two-call path (num_height > 1, a tile that does not fit the
 * LDS twice, more than 16 384 rays). */
int mrs_ring_descriptors_batch(mrs_radon_plan* plan, const float* d_xyz, const int64_t* d_offsets, int32_t batch,
                               const mrs_bev_cfg* cfg, float* d_bev, float* d_sino, float* d_sino_norm, mrs_stream stream);

/* Tuning knobs of mrs_ring_descriptors_batch (results do not depend on them). */
enum mrs_radon_option {
    MRS_RADON_OPT_FUSED_STAGGER_US = 1, /* odd workgroups start this many microseconds late (default 70, 0 = off) */
    MRS_RADON_OPT_FUSED_PREFETCH = 2,   /* 16-byte load triplets in flight per lane while rasterising: 2, 4 or 6 */
    MRS_RADON_OPT_FUSED_GRID = 3,       /* persistent workgroups (0 = one per compute unit)                      */
    MRS_RADON_OPT_FUSED_SKIP = 5,       /* measurement aid, outputs meaningless: 1 = leave out the rasteriser, 2 = the ray march (phase floors) */
    MRS_RADON_OPT_FUSED_VARIANT = 4     /* 2 (default): rays dealt to lanes by length (slot tables), raw sums in registers, the sinogram written
                                           once; 1: the same with the raw sums parked in the output buffer; 0: (angle, detector) order */
};
int mrs_radon_plan_set_option(mrs_radon_plan* plan, int32_t option, int32_t value);

/* (x - mean) / std over n_groups consecutive groups of group_len floats (unbiased std):
 * torchvision fn.normalize(t, mean=t.mean(), std=t.std()) as RING_ros/util.py:197,339-340,429-430
 * use it (RING++ normalises a whole [C,H,W] descriptor with ONE mean/std -> group_len=C*H*W).
 * In-place allowed (d_out == d_in).  A constant group (std == 0; torchvision raises) is written as zeros. */
int mrs_normalize_groups(mrs_ctx* ctx, const float* d_in, float* d_out, int32_t n_groups,
                         int32_t group_len, mrs_stream stream);

/* ------------------------------------------------------------------------------------
 * RING / RING++ descriptors and rotation correlation (rows R2, C1, C2, C3)
 * ---------------------------------------------------------------------------------- */

/* R2: TIRING = ortho FFT over the ANGLE axis of (already normalised) sinograms.
 * Replaces torch.fft.fft2(x, dim=-2, norm="ortho") at RING_ros/util.py:198.
 * d_x float[n_img][n_angles][det] -> d_out interleaved complex64 [n_img][n_angles][det]. */
int mrs_fft_angle_r2c(mrs_ctx* ctx, const float* d_x, int32_t n_img, int32_t n_angles, int32_t det,
                      float* d_out, mrs_stream stream);

/* R2 (RING++): |ortho FFT over the DETECTOR axis|.  Replaces forward_row_fft,
 * RING_ros/util.py:295-300 (first return value).  float[n_img][n_angles][det] both ways. */
int mrs_fft_row_magnitude(mrs_ctx* ctx, const float* d_x, int32_t n_img, int32_t n_angles, int32_t det,
                          float* d_out, mrs_stream stream);

/* C1/C2 database sweep: every query against every database entry.
 * Replaces the Python loop `for idx in range(len(pc_candidates)): fast_corr(...)`
 * (RING_ros/main_RING.py:133-140 -> util.py:362-374; RING++: util.py:337-358 after its two
 * fn.normalize calls, i.e. feed mrs_normalize_groups output).
 * d_query float[n_query][channels][n_angles][det], d_db float[n_db][...]: NORMALISED REAL
 * sinograms (RING) / normalised row-FFT magnitudes (RING++), the inverse transform of the
 * spectra the reference stores.  d_dist float[n_query][n_db] = 1 - max / (0.15*C*A*D),
 * d_angle int32[n_query][n_db] = A/2 - argmax(fftshift(corr)); d_corr (optional, may be NULL)
 * float[n_query][n_db][n_angles] = the shifted correlation vector. */
int mrs_ring_corr_sweep(mrs_ctx* ctx, const float* d_query, int32_t n_query, const float* d_db, int32_t n_db,
                        int32_t channels, int32_t n_angles, int32_t det, float* d_dist, int32_t* d_angle,
                        float* d_corr, mrs_stream stream);

/* Pairwise form of the sweep: pair i = (d_a[i], d_b[i]); outputs have n_pairs entries
 * (d_corr: float[n_pairs][n_angles]). */
int mrs_ring_corr_pairs(mrs_ctx* ctx, const float* d_a, const float* d_b, int32_t n_pairs, int32_t channels,
                        int32_t n_angles, int32_t det, float* d_dist, int32_t* d_angle, float* d_corr,
                        mrs_stream stream);

/* C1 literal: fast_corr(a, b) on complex spectra, pair by pair (RING_ros/util.py:362-374).
 * d_a, d_b interleaved complex64 [n_pairs][channels][n_angles][det]; outputs per pair.
 * dist = 1 - max / (0.15 * n_angles * det): like the reference, NO channel factor (util.py:369). */
int mrs_ring_corr_spectra(mrs_ctx* ctx, const float* d_a, const float* d_b, int32_t n_pairs, int32_t channels,
                          int32_t n_angles, int32_t det, float* d_dist, int32_t* d_angle, float* d_corr,
                          mrs_stream stream);

/* C3: solve_translation(query, positive, rot_angle) (RING_ros/util.py:388-423,488-506).
 * d_query/d_positive float[n_pairs][channels][height][width] (RING sinograms, `positive`
 * already row-rolled by the caller as main_RING.py:172-178 does); d_angles float[height]
 * (linspace(0, 2*pi, height)); d_rot float[n_pairs].  d_xy_err float[n_pairs][3] = x, y,
 * residual norm; d_shifts (optional) float[n_pairs][height] = the per-row integer shifts. */
int mrs_ring_solve_translation(mrs_ctx* ctx, const float* d_query, const float* d_positive, int32_t n_pairs,
                               int32_t channels, int32_t height, int32_t width, const float* d_angles,
                               const float* d_rot, float* d_xy_err, float* d_shifts, mrs_stream stream);

/* ------------------------------------------------------------------------------------
 * GICP refinement (rows G2-G6), batched over independent submap pairs
 * ---------------------------------------------------------------------------------- */
typedef struct mrs_gicp_batch mrs_gicp_batch;

/* Parameters with fast_gicp's names and defaults (upstream include/fast_gicp/gicp/
 * {fast_gicp.hpp,lsq_registration.hpp}).  Setters they replace:
 * setCorrespondenceRandomness, setMaxCorrespondenceDistance, setMaximumIterations,
 * setRotationEpsilon, setTransformationEpsilon (global_manager.cpp:2437-2442,
 * main_RING.py:93-94).  setNumThreads has no meaning on the GPU and is accepted and ignored
 * by the host-side mirrors. */
typedef struct mrs_gicp_params {
    int32_t k_correspondences;          /* neighbours for the covariances (20; Mapping: 15)   */
    int32_t max_iterations;             /* outer LM iterations (64; Mapping: icp_iters = 50)  */
    int32_t lm_max_iterations;          /* inner LM trials per iteration (10)                  */
    int32_t force_iterations;           /* > 0: run exactly this many outer iterations with the
                                           convergence test disabled (benchmark timing only)   */
    double max_correspondence_distance; /* DBL_MAX = unbounded (RING: 5.0, Mapping: 100)       */
    double rotation_epsilon;            /* 2e-3                                                */
    double transformation_epsilon;      /* 5e-4 (Mapping: 1e-3)                                */
    double lm_init_lambda_factor;       /* 1e-9                                                */
    double voxel_resolution;            /* 0 = GICP (FastGICP); > 0 = voxelised GICP, row G7
                                           (FastVGICP / FastVGICPCuda::setResolution; Mapping: 0.5,
                                           global_manager.cpp:2450)                            */
    int32_t voxel_neighbors;            /* 1 / 7 / 27 = DIRECT1 / DIRECT7 / DIRECT27
                                           (setNeighborSearchMethod, global_manager.cpp:2452)  */
    int32_t reserved;
    double convergence_factor;          /* LsqRegistration::is_converged: max(f |R - I| / rotation_epsilon,
                                           f |t| / transformation_epsilon) < 1 with upstream's f = 10
                                           (0 selects 10)                                      */
} mrs_gicp_params;

void mrs_gicp_default_params(mrs_gicp_params* p);

/* One handle aligns n_pairs independent (source, target) pairs together.
 * Replaces n_pairs instances of fast_gicp::FastGICP<PointXYZI,PointXYZI> / pygicp.FastGICP
 * (factory at global_manager.cpp:2416-2461; pygicp use at main_RING.py:81-104). */
int mrs_gicp_batch_create(mrs_ctx* ctx, int32_t n_pairs, mrs_gicp_batch** out);
int mrs_gicp_batch_destroy(mrs_gicp_batch* h);
int mrs_gicp_batch_set_params(mrs_gicp_batch* h, const mrs_gicp_params* p);

/* setInputSource (which = 0) / setInputTarget (which = 1) for every pair at once.
 * d_points: packed DEVICE array of points, x y z in the first three of every `stride_floats`
 * floats (3 = pygicp's Nx3, 4 = pcl::PointXYZ, 8 = pcl::PointXYZI); h_offsets: HOST
 * int64[n_pairs+1] point offsets of the pairs.  Copies into the library's float4 layout. */
int mrs_gicp_batch_set_clouds(mrs_gicp_batch* h, int32_t which, const float* d_points, int32_t stride_floats,
                              const int64_t* h_offsets, mrs_stream stream);

/* The same with the points in HOST memory (the form a pcl::PointCloud / numpy caller has: setInputSource / setInputTarget of
 * the C++ adapter, pygicp.FastGICP.set_input_*): staged through the library's scratch cache, synchronous. */
int mrs_gicp_batch_set_clouds_host(mrs_gicp_batch* h, int32_t which, const float* h_points, int32_t stride_floats,
                                   const int64_t* h_offsets);
/* Submap store: a second batch used as a container of UNIQUE clouds (mrs_gicp_batch_create(ctx, n_clouds), set_clouds(which = 1, ...),
 * compute_covariances(1)).  Pair i's cloud of side `which` := cloud h_ids[i] of side `store_which` of `store`: the sorted points, the
 * covariances, the boxes and the octree-cell hierarchy are copied on the device, nothing is rebuilt -- a submap that takes part in several
 * pairs (a new scan checked against several stored candidates: main_RING.py:81-104, global_manager.cpp:2016-2021, where fast_gicp itself
 * recomputes both clouds' covariances for every pair) is sorted and gets its covariances once.  Same results as set_clouds with the same
 * points (tests/test_gicp_gpu.py).  Both batches must use the same k_correspondences. */
int mrs_gicp_batch_set_clouds_from(mrs_gicp_batch* h, int32_t which, mrs_gicp_batch* store, int32_t store_which, const int32_t* h_ids,
                                   mrs_stream stream);

/* G2: FastGICP::calculate_covariances (brute-force kNN, PLANE regularisation).  Called lazily
 * by align; exposed so that it can be timed / cached per submap.  d_knn_out (optional, may be
 * NULL): int32[total_points][k] neighbour indices (cloud-local, ascending distance). */
int mrs_gicp_batch_compute_covariances(mrs_gicp_batch* h, int32_t which, int32_t* d_knn_out, mrs_stream stream);
/* h_cov6: double[total_points][6] = xx xy xz yy yz zz of the regularised 3x3 block */
int mrs_gicp_batch_get_covariances(mrs_gicp_batch* h, int32_t which, double* h_cov6);

/* align(output, guess): h_guess double[n_pairs][16] row-major 4x4 (NULL = identity; narrowed to
 * float like the reference's Eigen::Matrix4f guess), h_final double[n_pairs][16] =
 * getFinalTransformation() (float precision), h_converged = hasConverged(), h_iterations =
 * outer iterations used, h_hessian double[n_pairs][36] (each optional).  Synchronises `stream`. */
int mrs_gicp_batch_align(mrs_gicp_batch* h, const double* h_guess, double* h_final, int32_t* h_converged,
                         int32_t* h_iterations, double* h_hessian, mrs_stream stream);

/* G3+G4 once at given poses (kernel-level parity hook): H double[n_pairs][36], b [n_pairs][6],
 * err [n_pairs]; d_corr (optional) int32[total source points] correspondences or -1. */
int mrs_gicp_batch_linearize(mrs_gicp_batch* h, const double* h_poses, double* h_H, double* h_b, double* h_err,
                             int32_t* d_corr, mrs_stream stream);

/* G6: pcl::Registration::getFitnessScore(max_range) at the given poses
 * (global_manager.cpp:2058-2071, main_RING.py:100). */
int mrs_gicp_batch_fitness(mrs_gicp_batch* h, const double* h_poses, double max_range, double* h_scores,
                           mrs_stream stream);

/* number of brute-force NN passes the last align() issued (for iterations/s accounting) */
double mrs_gicp_batch_last_nn_passes(const mrs_gicp_batch* h);

/* Measurement hook of bench.py (no reference counterpart): every kernel of one outer iteration launched alone between HIP events on
 * `stream`, `reps` times, at the given poses (double[n_pairs][16]).  out_ms float[8]: 0 k_linearize (28 sums), 1 k_linearize (error only:
 * an LM trial), 2 round-3 search of every point (warm), 3 k_nn_certify at an unchanged pose, 4 round-4 search of every point (warm),
 * 5 k-NN selection of the sources, 6 k_cov_from_knn of the sources, 7 certify + work-list search after a 1 mm step.  out_counts int64[3]:
 * source points, correspondences at the poses, queries on the work lists of [7].  The batch's search settings are restored on every way out
 * (errors included); its correspondences, warm-start seeds and certificates are OVERWRITTEN (they are those of the profiled poses afterwards). */
int mrs_gicp_batch_profile(mrs_gicp_batch* h, const double* h_poses, int32_t reps, float* out_ms, int64_t* out_counts, mrs_stream stream);

/* Which exact nearest-neighbour searches the batch uses for G2 / G3 (no reference counterpart: upstream fast_gicp searches a
 * kd-tree; every setting returns the exact neighbours, ties aside -- identical transforms, tests/test_gicp_gpu.py):
 *   1 (default) align(): the first pass and every pair whose last step moved it by more than 2 cm are searched by the round-3 kernel
 *               (1024-point tiles / 16-point minis, candidates shared by a wave: best for broad searches); the other pairs first
 *               CERTIFY the neighbours of the previous pass (triangle inequality: a query that moved by delta keeps its neighbour
 *               if that neighbour's new distance is below [the old lower bound of every other point's distance] - delta) and search
 *               only what could not be certified, on octree-cell leaves with per-query culling (csrc/nn_core.hpp).  k-NN for the
 *               covariances: round-3 kernel;
 *   2           like 1 without certificates (every pass searches every point);
 *   3           round-4 kernels everywhere (first pass and k-NN too; slower on lidar scans, kept for comparison);
 *   0           the round-3 kernels everywhere, no certificates (A/B measurements, cross-check in the tests).
 * Invalidates cached covariances when the k-NN kernel changes. */
int mrs_gicp_batch_set_search(mrs_gicp_batch* h, int32_t core);
/* share of (source point, nearest-neighbour pass) of the last align() that needed a search (1.0 without certificates) */
double mrs_gicp_batch_last_searched_fraction(const mrs_gicp_batch* h);

/* ------------------------------------------------------------------------------------
 * rocFFT-backed 2-D correlations: DiSCO (rows D1, D2) and RING++ BEV translation (row C4)
 * ---------------------------------------------------------------------------------- */

/* D1: DiSCO.forward with the UNet bypassed, as the ROS node runs it
 * (disco_ros/models/DiSCO.py:315-334, fftshift2d :280-294).
 * d_bev float[batch][num_height][num_ring][num_sector] (polar occupancy, mrs_bev_polar_batch COMPACT)
 * -> d_signature float[batch][4*col*col] (shifted magnitude, centre crop; col = 16 -> 1024-d) and
 *    d_spectrum interleaved complex64 [batch][num_ring][num_sector] (ortho fft2 of the height sum). */
int mrs_disco_descriptor(mrs_ctx* ctx, const float* d_bev, int32_t batch, int32_t num_height, int32_t num_ring,
                         int32_t num_sector, int32_t col, float* d_signature, float* d_spectrum, mrs_stream stream);

/* D2: phase_corr(a, b) (disco_ros/main.py:260-272) for n_pairs spectra pairs.
 * d_flat_argmax int32[n_pairs] = flat argmax of the shifted magnitude map (the reference then takes
 * `% num_sector`; the host mirror does that); d_corr (optional) float[n_pairs][num_ring][num_sector]. */
int mrs_disco_phase_corr(mrs_ctx* ctx, const float* d_a, const float* d_b, int32_t n_pairs, int32_t num_ring,
                         int32_t num_sector, int32_t* d_flat_argmax, float* d_corr, mrs_stream stream);

/* C4: solve_translation_bev(a, b) (RING_ros/util.py:427-450): d_a, d_b float[n_pairs][channels][H][W];
 * d_arg int32[n_pairs] = flat (row-major) index of the first maximum of the shifted correlation map,
 * d_max (optional) float[n_pairs] its value, d_corr (optional) float[n_pairs][H][W] the map. */
int mrs_bev_translation(mrs_ctx* ctx, const float* d_a, const float* d_b, int32_t n_pairs, int32_t channels,
                        int32_t height, int32_t width, int32_t* d_arg, float* d_max, float* d_corr, mrs_stream stream);

/* rotate_bev (RING_ros/util.py:67-70 -> torchvision rotate: nearest, about the centre, zero fill).
 * d_img float[n_images][H][W]; image i uses d_angle_deg[i / images_per_angle] (degrees, counter-clockwise). */
int mrs_rotate_nearest(mrs_ctx* ctx, const float* d_img, int32_t n_images, int32_t images_per_angle, int32_t height,
                       int32_t width, const float* d_angle_deg, float* d_out, mrs_stream stream);

/* ------------------------------------------------------------------------------------
 * RING++ point-feature front-end
 * ---------------------------------------------------------------------------------- */

/* Replaces voxelfeat.GPUFeatureExtractor(point, size, 13, k, neighbors_indices, eigens).get_features()
 * (generate_bev_pointfeat_cython/wrapper.pyx:43-59, src/kernel.cu:16-104): d_points float[n][3]
 * row-major, d_knn int32[n][k], d_eigens float[n][5] (3-D then 2-D eigenvalues, descending),
 * d_features float[n][13] = C,O,L,E,P,S,A,X,D,S2,L2,dZ,vZ.  k <= 32. */
int mrs_pointfeat_from_neighbors(mrs_ctx* ctx, const float* d_points, int32_t n, int32_t k, const int32_t* d_knn,
                                 const float* d_eigens, float* d_features, mrs_stream stream);
int mrs_pointfeat_from_neighbors_host(mrs_ctx* ctx, const float* h_points, int32_t n, int32_t k, const int32_t* h_knn,
                                      const float* h_eigens, float* h_features);

/* The whole front-end of generate_RINGplusplus on the GPU (RING_ros/util.py:163-170 build_neighbors_NN
 * [sklearn kd-tree kNN, k = 30], :123-160 covariation_eigenvalue [CPU eigvalsh], :218-228 features):
 * exact kNN (self included), covariance / (k-1), eigenvalues, 13 features, for `batch` clouds.
 * d_points: packed device points, xyz in the first 3 of every stride_floats floats; h_offsets HOST
 * int64[batch+1].  Outputs (each optional) in the caller's point order: d_knn int32[N][k],
 * d_eigens float[N][5], d_features float[N][13], d_feat_planes float[9*N]: per scan the channel-major
 * planes x,y,z,C,O,E,L2,dZ,vZ, i.e. exactly the input of mrs_bev_feat_batch with featsize 9.
 * Synchronises `stream`.  The Morton-ordered working copy of the clouds (~60 B per point) stays in the context between calls, per value of
 * `batch` (at most four) and grow-only, and is released by mrs_ctx_destroy; calls from several threads are safe (one of them uses the kept
 * buffers, the others temporary ones). */
int mrs_pointfeat_batch(mrs_ctx* ctx, const float* d_points, int32_t stride_floats, const int64_t* h_offsets,
                        int32_t batch, int32_t k, int32_t* d_knn, float* d_eigens, float* d_features,
                        float* d_feat_planes, mrs_stream stream);

/* ------------------------------------------------------------------------------------
 * FFT-domain RING correlation (rows R2, C1), specialised for num_ring = num_sector = 120
 * ---------------------------------------------------------------------------------- */

/* Half TIRING: the first n_angles/2+1 = 61 angle-frequency rows of torch.fft.fft2(x, dim=-2, norm="ortho")
 * (RING_ros/util.py:198) for real input; the remaining rows are their conjugates.
 * d_norm_sino float[n_img][120][120] -> d_half_spec interleaved complex64 [n_img][61][120].
 * This is the database format of the FFT-domain sweep (58 560 B per entry). */
int mrs_ring_half_spectrum(mrs_ctx* ctx, const float* d_norm_sino, int32_t n_img, int32_t n_angles, int32_t det,
                           float* d_half_spec, mrs_stream stream);

/* C1 sweep / pairs on half spectra: same outputs as mrs_ring_corr_sweep / mrs_ring_corr_pairs
 * (fast_corr, RING_ros/util.py:362-374; single channel). */
int mrs_ring_corr_fft_sweep(mrs_ctx* ctx, const float* d_query_spec, int32_t n_query, const float* d_db_spec,
                            int32_t n_db, float* d_dist, int32_t* d_angle, float* d_corr, mrs_stream stream);
int mrs_ring_corr_fft_pairs(mrs_ctx* ctx, const float* d_a_spec, const float* d_b_spec, int32_t n_pairs,
                            float* d_dist, int32_t* d_angle, float* d_corr, mrs_stream stream);
/* Several sweeps in one launch (the loop of main_RING.py:133-140 for a batch of new scans, each against ITS OWN slice of one descriptor
 * pool): query q is entry d_query_row[q] of d_spec ([entries][61][120] complex64) and sweeps the n_db entries that start at entry
 * d_db_first[q] (device int64 arrays, n_query <= 65535); d_dist / d_angle [n_query][n_db] as in mrs_ring_corr_fft_sweep.
 * PRECONDITION (not checked: the indices live on the device): every d_query_row[q] and every d_db_first[q] + n_db - 1 is an entry of d_spec. */
int mrs_ring_corr_fft_sweep_blocks(mrs_ctx* ctx, const float* d_spec, const int64_t* d_query_row, int32_t n_query, const int64_t* d_db_first,
                                   int32_t n_db, float* d_dist, int32_t* d_angle, mrs_stream stream);

/* DMA-tiled database entries: the 7 320 complex values of a half spectrum permuted so that every 1-KiB LDS-DMA wave-instruction of the
 * one-query sweep (gfx950 `global_load_lds_dwordx4`) reads ONE contiguous, 128-byte aligned block (csrc/ringfft.hip: kTiledEntryBytes):
 * 58 624 bytes per entry (58 560 + 64 of padding).  The array must be followed by at least 1 KiB of readable memory (row 30 is a half block).
 * mrs_ring_spec_to_tiled: d_half_spec [n][61][120] complex64 -> d_tiled [n][58 624 B].
 * mrs_ring_corr_fft_sweep_tiled: ONE query (row layout, [channels][61][120] complex64) against n_db entries of `channels` tiled planes each
 * (channels = 1: RING; 6: RING++); outputs as mrs_ring_corr_fft_sweep / _mc, bit-identical to them.  This is the sweep behind mrs_loopdb_query (the node's loop, main_RING.py:133-140). */
#define MRS_RING_TILED_ENTRY_BYTES 58624
int mrs_ring_spec_to_tiled(mrs_ctx* ctx, const float* d_half_spec, int32_t n, float* d_tiled, mrs_stream stream);
int mrs_ring_corr_fft_sweep_tiled(mrs_ctx* ctx, const float* d_query_spec, const float* d_db_tiled, int32_t n_db, int32_t channels, float* d_dist,
                                  int32_t* d_angle, mrs_stream stream);
/* The same for n_query queries at once ([n_query][channels][61][120] complex64, row layout; d_dist / d_angle [n_query][n_db]): one robot's scan
 * against the other robots' lists, or the three callbacks' scans against one list (main_RING.py:241-390; BASELINE configs[3]).  Every
 * (query, entry) goes through the one-query pipeline unchanged (bit-identical results); the workgroups that sweep the same entries for
 * different queries share an XCD, so an entry is fetched from HBM once per launch and served to the others by that XCD's L2.  Launches of at
 * most 32 (channels = 1) / 8 (channels > 1) queries; more are split. */
int mrs_ring_corr_fft_sweep_tiled_q(mrs_ctx* ctx, const float* d_query_spec, int32_t n_query, const float* d_db_tiled, int32_t n_db, int32_t channels,
                                    float* d_dist, int32_t* d_angle, mrs_stream stream);

/* One launch for the new-descriptor side of a batch of loop checks: half spectra of n_pairs freshly normalised
 * sinograms (d_half_spec and/or its fp16 replica, either may be null) and their correlation with one candidate
 * spectrum each (d_cand_spec [n_pairs][61][120] complex64).  Results are bitwise those of mrs_ring_half_spectrum
 * followed by mrs_ring_corr_fft_pairs. */
int mrs_ring_spectrum_corr_pairs(mrs_ctx* ctx, const float* d_norm_sino, const float* d_cand_spec, int32_t n_pairs,
                                 int32_t n_angles, int32_t det, float* d_half_spec, void* d_half_spec_f16, float* d_dist,
                                 int32_t* d_angle, mrs_stream stream);
/* Same launch with the candidates picked out of a database: candidate of pair i = row d_cand_index[i] of d_db_spec,
 * a [n_db][61][120] array of complex64 half spectra (db_is_f16 = 0) or of their fp16 replicas (db_is_f16 = 1:
 * IEEE binary16 pairs, the multi-GPU exchange format; arithmetic stays fp32).  This is the per-query step after a
 * candidate pre-selection over the replicated database (RING_ros/main_RING.py:133-140 scores every candidate).
 * An index outside [0, n_db) means "no candidate": dist = +inf, angle = 0 (the new spectrum is still written). */
int mrs_ring_spectrum_corr_pairs_db(mrs_ctx* ctx, const float* d_norm_sino, const void* d_db_spec, int32_t db_is_f16, int32_t n_db,
                                    const int32_t* d_cand_index, int32_t n_pairs, int32_t n_angles, int32_t det, float* d_half_spec,
                                    void* d_half_spec_f16, float* d_dist, int32_t* d_angle, mrs_stream stream);

/* Multi-channel forms (RING++, fast_corr_RINGplusplus, RING_ros/util.py:337-358): descriptors are
 * [channels][61][120] complex64 half spectra of the jointly normalised channels (mrs_normalize_groups over
 * channels*120*120, then mrs_ring_half_spectrum with n_img = n * channels); |corr| is summed over channels and
 * detectors, dist = 1 - max / (0.15 * channels * 120 * 120). */
int mrs_ring_corr_fft_sweep_mc(mrs_ctx* ctx, const float* d_query_spec, int32_t n_query, const float* d_db_spec, int32_t n_db,
                               int32_t channels, float* d_dist, int32_t* d_angle, float* d_corr, mrs_stream stream);
int mrs_ring_corr_fft_pairs_mc(mrs_ctx* ctx, const float* d_a_spec, const float* d_b_spec, int32_t n_pairs, int32_t channels,
                               float* d_dist, int32_t* d_angle, float* d_corr, mrs_stream stream);

/* fp16 replica format (multi-GPU exchange): every descriptor has to reach every GPU, and at
 * >1 M descriptors/s/GPU the fp32 half spectrum (58 560 B) exceeds what the xGMI links carry.  A rank keeps its
 * own descriptors in fp32 (exact scoring) and ships round-to-nearest fp16 copies (29 280 B) to the others, which
 * sweep them with the same kernel (fp32 arithmetic; |dist - fp32 dist| < 2e-3).  The reference has no multi-GPU
 * path; this is the RCCL all-gather format, not a change of the single-GPU results.
 * mrs_ring_half_spectrum_f16: like mrs_ring_half_spectrum, with either output optional (not both null);
 * d_half_spec_f16 = IEEE binary16 pairs [n_img][61][120][2]. */
int mrs_ring_half_spectrum_f16(mrs_ctx* ctx, const float* d_norm_sino, int32_t n_img, int32_t n_angles, int32_t det,
                               float* d_half_spec, void* d_half_spec_f16, mrs_stream stream);
int mrs_ring_corr_fft_sweep_f16(mrs_ctx* ctx, const float* d_query_spec, int32_t n_query, const void* d_db_spec_f16,
                                int32_t n_db, float* d_dist, int32_t* d_angle, float* d_corr, mrs_stream stream);

/* ------------------------------------------------------------------------------------
 * Scan pre-processing
 * ---------------------------------------------------------------------------------- */

/* open3d PointCloud.voxel_down_sample(voxel_size) as the RING / RING++ / SC nodes call it
 * (RING_ros/main_RING.py:257-259): d_points [n][stride] float (is_double = 0) or double (1), xyz first;
 * d_out double[<= n][3] = voxel centroids sorted by voxel index; *h_count = number of voxels.
 * Synchronises `stream`. */
int mrs_voxel_downsample(mrs_ctx* ctx, const void* d_points, int32_t is_double, int32_t stride, int32_t n,
                         double voxel_size, double* d_out, int32_t* h_count, mrs_stream stream);

/* The same filter for a batch of scans in one set of launches (hash grid instead of a sort, one host synchronisation per call instead of per
 * scan): d_points [total][stride], d_raw_offsets / h_raw_offsets int64[batch + 1] (device and host copies of the same offsets, starting at 0).
 * d_out double[<= total][3] receives the centroids scan after scan, each scan's voxels in order of FIRST OCCURRENCE (open3d's own order is
 * unspecified), d_out_offsets int64[batch + 1] (device) their extents.  Centroids are within 1e-13 m of mrs_voxel_downsample's. */
int mrs_voxel_downsample_batch(mrs_ctx* ctx, const void* d_points, int32_t is_double, int32_t stride, const int64_t* d_raw_offsets,
                               const int64_t* h_raw_offsets, int32_t batch, double voxel_size, double* d_out, int64_t* d_out_offsets,
                               mrs_stream stream);

/* G1: pygicp.downsample(points, resolution) (RING_ros/main_RING.py:84-85, disco_ros/main.py:177-178, main_SC.py:111-112)
 * = pcl::ApproximateVoxelGrid<pcl::PointXYZ> with leaf (r, r, r): points narrowed to float, a 512-entry direct-mapped
 * history streamed in input order (a voxel evicted by a colliding one and met again yields another output point),
 * centroids in float, output in flush order.  Same arguments as mrs_voxel_downsample; the result equals the sequential
 * filter bit for bit, order included.  Synchronises `stream`. */
int mrs_voxel_downsample_approx(mrs_ctx* ctx, const void* d_points, int32_t is_double, int32_t stride, int32_t n,
                                double leaf_size, double* d_out, int32_t* h_count, mrs_stream stream);
/* host arrays in and out (pygicp.downsample's calling convention); h_out: room for n x 3 doubles */
int mrs_voxel_downsample_approx_host(mrs_ctx* ctx, const void* h_points, int32_t is_double, int32_t stride, int32_t n,
                                     double leaf_size, double* h_out, int32_t* h_count);

/* load_pc_infer (RING_ros/util.py:91-112, disco_ros/main.py:94-113) for a batch of raw clouds: float32
 * cast, keep |x|,|y| < 70 and 0 < z < 30, divide by 70/70/30.  Raw cloud b = points
 * [raw_offsets[b], raw_offsets[b+1]) (the offsets are needed on both sides: d_ device, h_ host).
 * d_xyz_soa: float[3 * total_raw] (upper bound); d_out_offsets: int64[batch+1], written on the device.
 * (d_xyz_soa, d_out_offsets) is exactly the input of mrs_bev_*_batch: no host round trip. */
int mrs_crop_scale_batch(mrs_ctx* ctx, const void* d_points, int32_t is_double, int32_t stride,
                         const int64_t* d_raw_offsets, const int64_t* h_raw_offsets, int32_t batch,
                         float* d_xyz_soa, int64_t* d_out_offsets, mrs_stream stream);

/* ------------------------------------------------------------------------------------
 * Multi-GPU exchange of the descriptor database over RCCL / xGMI
 * ---------------------------------------------------------------------------------- */

/* The reference has no collective at all (its three robots share one process and one GPU).  One process per GPU here; every rank builds the
 * descriptors of its share of the scans, then either every rank receives everybody's (mrs_exchange_allgather: ONE ncclAllGather per launch;
 * the per-robot lists of main_RING.py:284-288, replicated) or only the candidate rows a rank asks for travel (mrs_exchange_fetch_rows).
 * A C++ host (the Mapping node) calls these directly; mr_slam_amd/shard.py calls the same entry points.  RCCL is looked up at run time
 * (symbols already in the process first, then librccl.so.1); mrs_exchange_available() says whether it was found. */
typedef struct mrs_exchange mrs_exchange;
int mrs_exchange_available(void);
/* ncclGetUniqueId: 128 bytes made by ONE rank and handed to the others by the host (file, socket, torch.distributed broadcast ...) */
int mrs_exchange_unique_id(uint8_t* out128);
/* ncclCommInitRank on the context's device; collective over all n_ranks */
int mrs_exchange_create(mrs_ctx* ctx, int32_t n_ranks, int32_t rank, const uint8_t* id128, mrs_exchange** out);
/* borrow a communicator the host already has (ncclComm_t as void*); it is not destroyed with the handle */
int mrs_exchange_create_from_comm(mrs_ctx* ctx, void* nccl_comm, int32_t n_ranks, int32_t rank, mrs_exchange** out);
int mrs_exchange_destroy(mrs_exchange* x);
int mrs_exchange_world(const mrs_exchange* x, int32_t* n_ranks, int32_t* rank);
/* d_all [n_ranks][n_local] entries of entry_bytes <- every rank's d_local [n_local], in rank order.  Enqueued on `stream`. */
int mrs_exchange_allgather(mrs_exchange* x, const void* d_local, int64_t n_local, int64_t entry_bytes, void* d_all, mrs_stream stream);
/* Rank r owns global rows [r * rows_per_rank, (r + 1) * rows_per_rank) (d_local_db).  d_out [n_rows] <- the entries of d_global_rows (device
 * int64 [n_rows], any owner, repeats allowed; the same n_rows on every rank), in request order: the requests are all-gathered, every owner
 * packs what it was asked for, one grouped send / receive per peer pair moves exactly those rows.  entry_bytes % 16 == 0.  Collective; blocking. */
int mrs_exchange_fetch_rows(mrs_exchange* x, const void* d_local_db, int64_t rows_per_rank, int64_t entry_bytes, const int64_t* d_global_rows,
                            int32_t n_rows, void* d_out, mrs_stream stream);
/* The same with the request phase done ahead of time (candidate rows are known long before the entries are needed: they come out of a coarse
 * search).  mrs_exchange_fetch_plan_create is a COLLECTIVE (every rank calls it, with its own d_global_rows [n_rows]; one all-gather of the
 * requests + one host synchronisation); mrs_exchange_fetch_planned then only enqueues work on `stream` (gather -> one grouped send / receive
 * per remote peer -> scatter to request order into d_out [n_rows][entry_bytes]): no host synchronisation, so it can be issued launches ahead
 * on a communication stream.  Fetches of ONE plan share the plan's staging buffers and must follow each other in stream order. */
typedef struct mrs_fetch_plan mrs_fetch_plan;
int mrs_exchange_fetch_plan_create(mrs_exchange* x, int64_t rows_per_rank, const int64_t* d_global_rows, int32_t n_rows, mrs_stream stream,
                                   mrs_fetch_plan** out);
int mrs_exchange_fetch_planned(mrs_fetch_plan* plan, const void* d_local_db, int64_t entry_bytes, void* d_out, mrs_stream stream);
int mrs_exchange_fetch_plan_counts(const mrs_fetch_plan* plan, int64_t* rows_sent_to_peers, int64_t* rows_received_from_peers);
int mrs_exchange_fetch_plan_destroy(mrs_fetch_plan* plan);

/* ------------------------------------------------------------------------------------
 * Loop database: the per-robot descriptor lists of the LoopDetection nodes, resident on the device
 * ---------------------------------------------------------------------------------- */

/* The reference keeps one Python list of descriptors per robot, appends one entry per callback and scores every new scan
 * against every entry of the other robots' lists in a Python loop:
 *   RING   : TIRING<k>.append(pc_TIRING) / `for idx in range(len(pc_candidates)): fast_corr(TIRING_current, TIRING_candidates[idx])`
 *            (RING_ros/main_RING.py:284-288, 126-140)
 *   RING++ : the same with fast_corr_RINGplusplus (RING_ros/main_RINGplusplus.py:126-134)
 *   DiSCO  : DiSCO<k>.append / FFT<k>.append, `KDTree(np.array(DiSCO_candidates)).query(DiSCO_current, k=1)` + phase_corr of the winner
 *            (disco_ros/main.py:276-291, 356-360)
 * A mrs_loopdb is that list on the device: entries are stored in the format the sweep kernels stream (RING: DMA-tiled half spectra,
 * RING++: [C][61][120] half spectra of the jointly normalised channels, DiSCO: 1024-d signatures + 40 x 120 spectra), an append writes
 * ONE slot (amortised doubling, nothing is re-uploaded), a query is ONE sweep over all entries.  Thread-safe (one lock per handle; the
 * reference's callbacks run on concurrent rospy threads); all device work of a handle runs on the handle's own stream, `stream` is the
 * stream that produced a DEVICE argument (the handle waits for it; for appends it is made to wait until the argument has been consumed). */
typedef struct mrs_loopdb mrs_loopdb;
enum mrs_loopdb_kind { MRS_LOOPDB_RING = 0, MRS_LOOPDB_RINGPP = 1, MRS_LOOPDB_DISCO = 2 };
/* where / what a RING or RING++ descriptor argument is:
 *   HOST / DEVICE : the reference's own object -- RING: pc_TIRING, complex64 [1][120][120] (util.py:198; its first 61 rows are what is
 *                   kept); RING++: pc_TIRING, float32 [C][120][120] magnitudes (util.py:247-250; normalised jointly + transformed along the
 *                   angle axis here, once, instead of at every comparison as util.py:339-343 does)
 *   DEVICE_SPEC   : half spectra in the product's row layout, complex64 [C][61][120] (mrs_ring_half_spectrum / mrs_ring_spectrum_corr_pairs) */
enum mrs_loopdb_form { MRS_LOOPDB_FORM_HOST = 0, MRS_LOOPDB_FORM_DEVICE = 1, MRS_LOOPDB_FORM_DEVICE_SPEC = 2 };

/* channels: 1 for RING, C (6 in the reference) for RING++, ignored for DiSCO; capacity_hint: entries to allocate up front (>= 1) */
int mrs_loopdb_create(mrs_ctx* ctx, int32_t kind, int32_t channels, int32_t capacity_hint, mrs_loopdb** out);
int mrs_loopdb_destroy(mrs_loopdb* db);
int mrs_loopdb_size(mrs_loopdb* db, int32_t* out_n);
int mrs_loopdb_reserve(mrs_loopdb* db, int32_t capacity);
int mrs_loopdb_clear(mrs_loopdb* db);
/* `TIRING<k>.append(descriptor)`.  count > 1 only with DEVICE_SPEC (a batch producer appending `count` consecutive half spectra). */
int mrs_loopdb_append(mrs_loopdb* db, const void* descriptor, int32_t form, int32_t count, mrs_stream stream);
/* The candidate loop of detect_loop_icp (main_RING.py:133-140): scores `descriptor` against every entry with one sweep and returns, in
 * index order, the entries with dist < dist_threshold (float32 comparison): h_index / h_dist / h_angle [max_out] and *h_count (the number
 * that qualified; when it exceeds max_out only the first max_out were written).  h_all_dist / h_all_angle (optional, [all_capacity]) receive
 * the scores of the first min(n, all_capacity) entries, *h_n (optional) the number of entries n this call scored -- another thread may append
 * between the caller's mrs_loopdb_size and this call, so arrays are never written past all_capacity.  Blocking; an empty database returns
 * *h_count = 0 without device work. */
int mrs_loopdb_query(mrs_loopdb* db, const void* descriptor, int32_t form, float dist_threshold, int32_t max_out, int32_t* h_index,
                     float* h_dist, int32_t* h_angle, int32_t* h_count, int32_t all_capacity, float* h_all_dist, int32_t* h_all_angle, int32_t* h_n,
                     mrs_stream stream);
/* `count` descriptors (consecutive in memory, all in the same form; 1 .. 1024) against every entry in ONE sweep
 * (mrs_ring_corr_fft_sweep_tiled_q): the shape of one robot's scan against several lists, of the three callbacks' scans against one list
 * (main_RING.py:241-390), and of BASELINE configs[3].  h_all_dist / h_all_angle [count][all_capacity]: row q = the scores of query q over
 * the first min(n, all_capacity) entries; *h_n = n.  Same bits as `count` calls of mrs_loopdb_query.  Blocking. */
int mrs_loopdb_query_multi(mrs_loopdb* db, const void* descriptors, int32_t form, int32_t count, int32_t all_capacity, float* h_all_dist,
                           int32_t* h_all_angle, int32_t* h_n, mrs_stream stream);
/* DiSCO: signature float32 [1024], spectrum complex64 [40][120] (DiSCO.forward's two outputs, disco_ros/models/DiSCO.py:315-334). */
int mrs_loopdb_append_disco(mrs_loopdb* db, const float* signature, const float* spectrum, int32_t on_device, mrs_stream stream);
/* disco_ros/main.py:284-291 in two launches: nearest signature (squared L2, ties to the lower index) and phase_corr(FFT_candidates[idx],
 * fft_current): *h_index (-1 for an empty database), *h_dist2, *h_flat_argmax = flat index of the first maximum of the shifted magnitude
 * map (the reference takes it `% num_sector`).  Blocking. */
int mrs_loopdb_query_disco(mrs_loopdb* db, const float* signature, const float* spectrum, int32_t on_device, int32_t* h_index, float* h_dist2,
                           int32_t* h_flat_argmax, mrs_stream stream);
/* the entries as they lie on the device (tests, exchange): valid until the next append that grows the capacity */
int mrs_loopdb_device_entries(mrs_loopdb* db, const float** d_entries, const float** d_signatures, int32_t* out_n, int64_t* entry_floats);

/* ------------------------------------------------------------------------------------
 * Mapping-side DiSCO matcher
 * ---------------------------------------------------------------------------------- */

/* GlobalManager::calcRelOri(newDiSCO, oldDiSCO) (Mapping/src/global_manager/src/global_manager.cpp:2719-2762),
 * literal including its quirks (non-conjugate cross term, no normalisation, unshifted argmax):
 * d_a, d_b interleaved complex64 [n_pairs][height][width] -> d_rel_angle_deg float[n_pairs]
 * = (argmax(real(IFFT2_double(cross))) % width) * 3.0. */
int mrs_disco_rel_ori_literal(mrs_ctx* ctx, const float* d_a, const float* d_b, int32_t n_pairs, int32_t height,
                              int32_t width, float* d_rel_angle_deg, mrs_stream stream);

/* Nearest DiSCO signature by squared L2 distance: replaces the kd-tree query over the 1024-d signatures
 * (global_manager.cpp:993-1188, src/kdtree.cpp; LoopDetection twin: sklearn KDTree k=1 at
 * disco_ros/main.py:284-285).  d_query float[n_query][dim], d_db float[n_db][dim]. */
int mrs_signature_search(mrs_ctx* ctx, const float* d_query, int32_t n_query, const float* d_db, int32_t n_db, int32_t dim,
                         int32_t* d_index, float* d_dist2, mrs_stream stream);

/* The k nearest signatures per query, ascending by squared distance (ties: lower index first): what
 * kdtree_knn_search(kdtree, query, NUM_CANDIDATES_FROM_TREE) + kdtree_knn_result return (global_manager.cpp:1002-1007,
 * src/kdtree.cpp:535-586; the reference reports sqrt of these distances and walks the list from entry 1, entry 0 being
 * the query's own descriptor, global_manager.cpp:1136-1139).  1 <= k <= 32; d_index / d_dist2 are [n_query][k], rows past
 * the database size hold -1 / +inf. */
int mrs_signature_knn(mrs_ctx* ctx, const float* d_query, int32_t n_query, const float* d_db, int32_t n_db, int32_t dim, int32_t k,
                      int32_t* d_index, float* d_dist2, mrs_stream stream);

/* ------------------------------------------------------------------------------------
 * Elevation mapping: the nine functions of the reference's libgpu.so
 * (Mapping/src/elevation_mapping_periodical/elevation_mapping/cuda/gpu_process.cu:938-1312; declared by hand
 * in elevation_mapping/src/ElevationMapping.cpp:44-50 and src/sensor_processors/SensorProcessorBase.cpp:34).
 * Same argument meaning, host arrays in/out like the reference; Eigen arguments become plain floats
 * (row-major 4x4 / 3x3, 3-vectors); the map state lives in a handle instead of __device__ globals.
 * ---------------------------------------------------------------------------------- */
typedef struct mrs_elev_map mrs_elev_map;

/* Init_GPU_elevationmap(length, resolution, mahalanobisDistanceThreshold, obstacle_threshold) */
int mrs_elev_create(mrs_ctx* ctx, int32_t length, float resolution, float mahalanobis_threshold, float obstacle_threshold,
                    mrs_elev_map** out);
int mrs_elev_destroy(mrs_elev_map* m);
/* Move(current_Position[3], resolution, length, Central_coordinate[2], Start_indice[2], alignedPositionShift[2]) */
int mrs_elev_move(mrs_elev_map* m, const float* h_position3, float* h_central2, int32_t* h_start2, float* h_aligned_shift2);
/* Process_points(map_index, point_x/y/z (read only: the reference never copies its device copy back), point_var, point_x/y/z_ts, transform, point_num, thresholds,
 * sensor model, sensorJacobian, rotationVariance, C_SB_transpose, P_mul_C_BM_transpose, B_r_BS_skew) */
int mrs_elev_process_points(mrs_elev_map* m, int32_t n, const float* h_x, const float* h_y, const float* h_z, const float* h_transform16,
                            double relative_lower_threshold, double relative_upper_threshold, float min_r, float beam_a,
                            float beam_c, const float* h_sensorJacobian3, const float* h_rotationVariance9,
                            const float* h_C_SB_transpose9, const float* h_P_mul_C_BM_transpose3, const float* h_B_r_BS_skew9,
                            int32_t* h_map_index, float* h_var, float* h_x_ts, float* h_y_ts, float* h_z_ts);
/* Fuse(length, point_num, point_index, colorR/G/B, intensity, height, var) */
int mrs_elev_fuse(mrs_elev_map* m, int32_t n, const int32_t* h_index, const int32_t* h_colorR, const int32_t* h_colorG,
                  const int32_t* h_colorB, const float* h_intensity, const float* h_height, const float* h_var);
/* Mapvar_update(length, var_update) */
int mrs_elev_mapvar_update(mrs_elev_map* m, float var_update);
/* Map_feature(length, elevation, var, colorR/G/B, rough, slope, traver, intensity): float/int32 [length*length] each */
int mrs_elev_map_feature(mrs_elev_map* m, float* h_elevation, float* h_var, int32_t* h_colorR, int32_t* h_colorG,
                         int32_t* h_colorB, float* h_rough, float* h_slope, float* h_traver, float* h_intensity);
/* Raytracing(length) */
int mrs_elev_raytracing(mrs_elev_map* m);
/* Map_optmove(opt_p[2], height_update, resolution, length, opt_alignedPosition[2]) */
int mrs_elev_map_optmove(mrs_elev_map* m, const float* h_opt_p2, float height_update, float* h_aligned2);
/* Map_closeloop(update_position[2], height_update, length, resolution) */
int mrs_elev_map_closeloop(mrs_elev_map* m, const float* h_update_position2, float height_update);
/* state readback for tests / debugging: which = 0 lowest, 1 elevation, 2 variance, 3 intensity, 4 traversability */
int mrs_elev_get_layer(mrs_elev_map* m, int32_t which, float* h_out);
int mrs_elev_get_frame(mrs_elev_map* m, float* h_central2, int32_t* h_start2);

#ifdef __cplusplus
}
#endif
#endif /* MRSLAM_HIP_H */
