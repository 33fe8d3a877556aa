// fft2d.hip -- rocFFT-backed 2-D correlations for gfx950.
//
// Reference behaviour reproduced (never copied):
//   D1  disco_ros/models/DiSCO.py:315-334 (+ fftshift2d :280-294): sum over height, fft2 (ortho),
//       magnitude (+1e-15), shift, centre crop -> signature; the complex spectrum is kept
//   D2  disco_ros/main.py:260-272 (phase_corr): ifft2(a * conj(b), ortho), magnitude (+1e-15),
//       fftshift2d, flat argmax % num_sector
//   C4  RING_ros/util.py:427-450 (solve_translation_bev): normalise, fft2 both, ifft2 of the
//       conjugate product, magnitude, sum over channels, fftshift, first global argmax;
//       util.py:67-70 (rotate_bev -> torchvision rotate, nearest, about the centre)
// The transforms themselves are batched rocFFT plans (cached per context); everything around
// them (height sum, conjugate product, ortho scaling, magnitude, shift, crop, argmax) is fused
// into the hand-written kernels below so each spectrum makes one trip through HBM per stage.
#include <rocfft/rocfft.h>

#include <cmath>

#include "common.hpp"

namespace {

struct PlanKey {
    int n0, n1, batch, inverse;  // inverse: 0 fwd f32, 1 inv f32, 3 inv f64
    bool operator<(const PlanKey& o) const
    {
        if (n0 != o.n0) return n0 < o.n0;
        if (n1 != o.n1) return n1 < o.n1;
        if (batch != o.batch) return batch < o.batch;
        return inverse < o.inverse;
    }
};

struct PlanEntry {
    rocfft_plan plan = nullptr;
    size_t work = 0;
};

std::mutex g_mu;
std::map<std::pair<const mrs_ctx*, PlanKey>, PlanEntry> g_plans;
bool g_setup = false;

#define MRS_FFT_TRY(expr)                                                         \
    do {                                                                          \
        rocfft_status s__ = (expr);                                               \
        if (s__ != rocfft_status_success) {                                       \
            ::mrs::set_error("%s failed with rocfft status %d", #expr, (int)s__); \
            return MRS_ERR_HIP;                                                   \
        }                                                                         \
    } while (0)

// in-place complex 2-D transform of `batch` contiguous [n0][n1] interleaved-complex images
int fft2_inplace(mrs_ctx* ctx, void* d, int n0, int n1, int batch, bool inverse, hipStream_t s, bool dbl = false)
{
    PlanEntry pe;
    {
        std::lock_guard<std::mutex> g(g_mu);
        if (!g_setup) { MRS_FFT_TRY(rocfft_setup()); g_setup = true; }
        const auto key = std::make_pair((const mrs_ctx*)ctx, PlanKey{n0, n1, batch, (inverse ? 1 : 0) | (dbl ? 2 : 0)});
        auto it = g_plans.find(key);
        if (it == g_plans.end()) {
            const size_t lengths[2] = {(size_t)n1, (size_t)n0};  // fastest dimension first
            MRS_FFT_TRY(rocfft_plan_create(&pe.plan, rocfft_placement_inplace,
                                           inverse ? rocfft_transform_type_complex_inverse : rocfft_transform_type_complex_forward,
                                           dbl ? rocfft_precision_double : rocfft_precision_single, 2, lengths, (size_t)batch, nullptr));
            MRS_FFT_TRY(rocfft_plan_get_work_buffer_size(pe.plan, &pe.work));
            it = g_plans.emplace(key, pe).first;
        }
        pe = it->second;
    }
    rocfft_execution_info info = nullptr;
    MRS_FFT_TRY(rocfft_execution_info_create(&info));
    mrs::Scratch work;
    int st = MRS_OK;
    do {
        if (rocfft_execution_info_set_stream(info, s) != rocfft_status_success) { mrs::set_error("rocfft set_stream failed"); st = MRS_ERR_HIP; break; }
        if (pe.work) {
            st = work.alloc(pe.work, s);
            if (st != MRS_OK) break;
            if (rocfft_execution_info_set_work_buffer(info, work.p, pe.work) != rocfft_status_success) { mrs::set_error("rocfft set_work_buffer failed"); st = MRS_ERR_HIP; break; }
        }
        void* in[1] = {d};
        if (rocfft_execute(pe.plan, in, nullptr, info) != rocfft_status_success) { mrs::set_error("rocfft_execute failed"); st = MRS_ERR_HIP; }
    } while (0);
    rocfft_execution_info_destroy(info);
    return st;
}

__device__ __forceinline__ float wave_max_arg(float v, int& idx)
{
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(v, o, 64);
        const int oi = __shfl_xor(idx, o, 64);
        if (ov > v || (ov == v && oi < idx)) { v = ov; idx = oi; }
    }
    return v;
}

// D1a: sum over the height layers -> complex image (imaginary part 0)
__global__ void k_height_sum(const float* __restrict__ bev, int H, int cells, float2* __restrict__ out)
{
    const int b = blockIdx.y;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < cells; i += gridDim.x * blockDim.x) {
        float s = 0.0f;
        for (int h = 0; h < H; ++h) s += bev[((size_t)b * H + h) * cells + i];
        out[(size_t)b * cells + i] = make_float2(s, 0.0f);
    }
}

// D1b: ortho scaling of the spectrum (kept for phase_corr) + shifted, cropped magnitude signature
__global__ void k_disco_finish(float2* __restrict__ spec, int R, int S, int col, float scale, float* __restrict__ sig)
{
    const int b = blockIdx.y;
    float2* sp = spec + (size_t)b * R * S;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < R * S; i += gridDim.x * blockDim.x) {
        float2 v = sp[i];
        v.x *= scale; v.y *= scale;
        sp[i] = v;
        // fftshift2d: shifted[r][s] = x[(r + ceil(R/2)) % R][(s + ceil(S/2)) % S]  (DiSCO.py:280-294)
        const int r = i / S, s = i - r * S;
        const int rs = (r + R - (R + 1) / 2) % R, ss = (s + S - (S + 1) / 2) % S;  // where (r,s) lands
        const int cr = rs - (R / 2 - col), cs = ss - (S / 2 - col);
        if (cr >= 0 && cr < 2 * col && cs >= 0 && cs < 2 * col)
            sig[(size_t)b * 4 * col * col + cr * 2 * col + cs] = sqrtf(v.x * v.x + v.y * v.y + 1e-15f);
    }
}

__global__ void k_conj_product(const float2* __restrict__ a, const float2* __restrict__ b, size_t n, float2* __restrict__ out)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float2 u = a[i], v = b[i];
        out[i] = make_float2(u.x * v.x + u.y * v.y, u.y * v.x - u.x * v.y);
    }
}

__global__ void k_real_to_complex(const float* __restrict__ x, size_t n, float2* __restrict__ out)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        out[i] = make_float2(x[i], 0.0f);
}

// magnitude of `C` complex images per pair (scaled), summed over C, shifted, first argmax.
// One workgroup per pair.  eps is added under the square root (D2) or not (C4).
// out_arg[pair] = flat index (row-major) of the first maximum of the SHIFTED map.
__global__ __launch_bounds__(1024) void k_mag_shift_argmax(const float2* __restrict__ corr, int C, int R, int S,
                                                           float scale, float eps, int* __restrict__ out_arg,
                                                           float* __restrict__ out_max, float* __restrict__ out_map)
{
    __shared__ float bv[16];
    __shared__ int bi[16];
    const int pair = blockIdx.x;
    const float2* base = corr + (size_t)pair * C * R * S;
    float best = -1.0f;
    int bidx = 0x7fffffff;
    for (int m = threadIdx.x; m < R * S; m += 1024) {  // m: index in the shifted map
        const int r = m / S, s = m - r * S;
        const int sr = (r + (R + 1) / 2) % R, ss = (s + (S + 1) / 2) % S;  // source of shifted (r,s)
        float tot = 0.0f;
        for (int c = 0; c < C; ++c) {
            const float2 v = base[((size_t)c * R + sr) * S + ss];
            const float re = v.x * scale, im = v.y * scale;
            tot += sqrtf(re * re + im * im + eps);
        }
        if (out_map) out_map[(size_t)pair * R * S + m] = tot;
        if (tot > best) { best = tot; bidx = m; }
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    best = wave_max_arg(best, bidx);
    if (lane == 0) { bv[wave] = best; bi[wave] = bidx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 16; ++w)
            if (bv[w] > best || (bv[w] == best && bi[w] < bidx)) { best = bv[w]; bidx = bi[w]; }
        out_arg[pair] = bidx;
        if (out_max) out_max[pair] = best;
    }
}

// rotate_bev: torchvision rotate(img, angle) defaults = nearest, about the centre, zero fill.
// theta maps output pixel centres to input coordinates (inverse rotation); see the restatement
// in oracle/corr_oracle.py::rotate_nearest for the derivation.
__global__ void k_rotate_nearest(const float* __restrict__ in, int H, int W, const float* __restrict__ angle_deg,
                                 int imgs_per_angle, float* __restrict__ out)
{
    const int img = blockIdx.y;
    const float rot = -angle_deg[img / imgs_per_angle] * 0.017453292519943295f;
    const float a = cosf(rot), b = sinf(rot);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < H * W; i += gridDim.x * blockDim.x) {
        const int y = i / W, x = i - y * W;
        const float bx = (float)x + 0.5f - 0.5f * W, by = (float)y + 0.5f - 0.5f * H;  // pixel centre rel. to centre
        // normalised grid coordinate, then grid_sample(align_corners=False) un-normalisation
        const float gx = (a * bx + b * by) / (0.5f * W), gy = (-b * bx + a * by) / (0.5f * H);
        const float ix = ((gx + 1.0f) * W - 1.0f) * 0.5f, iy = ((gy + 1.0f) * H - 1.0f) * 0.5f;
        const int sx = (int)nearbyintf(ix), sy = (int)nearbyintf(iy);
        float v = 0.0f;
        if (sx >= 0 && sx < W && sy >= 0 && sy < H) v = in[(size_t)img * H * W + sy * W + sx];
        out[(size_t)img * H * W + i] = v;
    }
}


// N4: GlobalManager::calcRelOri (Mapping/src/global_manager/src/global_manager.cpp:2719-2762), LITERAL:
// cross = (ra*rb + ia*ib) + i (ra*ib + rb*ia)   [float products; note the + in the imaginary part, which is
// not a conjugate product -- reproduced on purpose], unnormalised backward 2-D FFT in double, real part
// narrowed to float, first argmax, relAngle = (argmax % width) * 3.0.
__global__ void k_relori_cross(const float2* __restrict__ a, const float2* __restrict__ b, size_t n, double2* __restrict__ out)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float2 u = a[i], v = b[i];
        out[i] = make_double2((double)(u.x * v.x + u.y * v.y), (double)(u.x * v.y + v.x * u.y));
    }
}

__global__ __launch_bounds__(1024) void k_relori_argmax(const double2* __restrict__ corr, int R, int S, float* __restrict__ rel_angle)
{
    __shared__ float bv[16];
    __shared__ int bi[16];
    const double2* base = corr + (size_t)blockIdx.x * R * S;
    float best = -INFINITY;
    int bidx = 0x7fffffff;
    for (int m = threadIdx.x; m < R * S; m += 1024) {
        const float v = (float)base[m].x;
        if (v > best) { best = v; bidx = m; }
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    best = wave_max_arg(best, bidx);
    if (lane == 0) { bv[wave] = best; bi[wave] = bidx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 16; ++w)
            if (bv[w] > best || (bv[w] == best && bi[w] < bidx)) { best = bv[w]; bidx = bi[w]; }
        rel_angle[blockIdx.x] = (float)(bidx % S) * 3.0f;
    }
}

// N4: nearest DiSCO signature (the role of the kd-tree of global_manager.cpp:993-1188 / src/kdtree.cpp):
// exact brute-force squared-L2 1-NN in the difference form sum (q - r)^2 (what an L2 kd-tree metric evaluates;
// no |q|^2 + |r|^2 - 2 q.r cancellation).  Register-tiled like an SGEMM: a workgroup owns 64 queries x 64
// database rows, every lane a 4 x 4 block of distances, the operands staged through LDS in slices of 16
// dimensions (transposed, so that the inner loop reads two float4 per dimension for 16 sub+fma pairs); the
// database is read once per 64 queries.  Winners are merged with a 64-bit atomicMin on (distance bits, index):
// distances are >= 0, so their bit patterns order like the values and ties go to the smaller index.
constexpr int kSigTile = 64, kSigSlice = 16;

// ALL: every distance goes to dmat[nq][n] (the k-nearest selection reads it) instead of the per-query minimum.
template <bool ALL>
__global__ __launch_bounds__(256) void k_signature_tile(const float* __restrict__ q, int nq, const float* __restrict__ db, int n,
                                                        int dim, unsigned long long* __restrict__ best, float* __restrict__ dmat)
{
    __shared__ __attribute__((aligned(16))) float qs[kSigSlice][kSigTile];
    __shared__ __attribute__((aligned(16))) float rs[kSigSlice][kSigTile];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int r0 = blockIdx.x * kSigTile, q0 = blockIdx.y * kSigTile;
    const int lrow = threadIdx.x >> 2, lc = (threadIdx.x & 3) * 4;   // staging: row lrow of the tile, 4 consecutive dimensions
    float acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = 0.0f;
    for (int c0 = 0; c0 < dim; c0 += kSigSlice) {
        float qv[4], rv[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int c = c0 + lc + e;
            qv[e] = (q0 + lrow < nq && c < dim) ? q[(size_t)(q0 + lrow) * dim + c] : 0.0f;
            rv[e] = (r0 + lrow < n && c < dim) ? db[(size_t)(r0 + lrow) * dim + c] : 0.0f;
        }
        __syncthreads();   // previous slice consumed
#pragma unroll
        for (int e = 0; e < 4; ++e) { qs[lc + e][lrow] = qv[e]; rs[lc + e][lrow] = rv[e]; }
        __syncthreads();
#pragma unroll
        for (int c = 0; c < kSigSlice; ++c) {
            const float4 qa = *reinterpret_cast<const float4*>(&qs[c][4 * ty]);
            const float4 rb = *reinterpret_cast<const float4*>(&rs[c][4 * tx]);
            const float qe[4] = {qa.x, qa.y, qa.z, qa.w}, re[4] = {rb.x, rb.y, rb.z, rb.w};
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const float d = qe[a] - re[b];
                    acc[a][b] = __builtin_fmaf(d, d, acc[a][b]);
                }
        }
    }
    if (ALL) {
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const int qi = q0 + 4 * ty + a;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int row = r0 + 4 * tx + b;
                if (qi < nq && row < n) dmat[(size_t)qi * n + row] = acc[a][b];
            }
        }
        return;
    }
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        unsigned long long key = ~0ull;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int row = r0 + 4 * tx + b;
            if (row < n) {
                const unsigned long long k = ((unsigned long long)__float_as_uint(acc[a][b]) << 32) | (unsigned)row;
                key = k < key ? k : key;
            }
        }
        for (int o = 1; o < 16; o <<= 1) {   // the 16 lanes that share this query are consecutive
            const unsigned long long other = __shfl_xor(key, o, 64);
            key = other < key ? other : key;
        }
        const int qi = q0 + 4 * ty + a;
        if (tx == 0 && qi < nq && key != ~0ull) atomicMin(&best[qi], key);
    }
}

__global__ void k_signature_unpack(const unsigned long long* __restrict__ best, int nq, int* __restrict__ out_idx,
                                   float* __restrict__ out_d2)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nq) return;
    out_idx[i] = (int)(unsigned)(best[i] & 0xffffffffull);
    out_d2[i] = __uint_as_float((unsigned)(best[i] >> 32));
}

// k smallest (distance, index) pairs of one row of the distance matrix, ascending; ties by the lower index.  One workgroup per
// query: every thread keeps the KMAX best of its strided share as packed keys (distance bits << 32 | index; distances are
// >= 0, so the unsigned order is the float order), then the 256 x KMAX keys are sorted in LDS (bitonic) and the first k leave.
template <int KMAX>
__global__ __launch_bounds__(256) void k_signature_select(const float* __restrict__ dmat, int n, int k, int* __restrict__ out_idx,
                                                          float* __restrict__ out_d2)
{
    extern __shared__ unsigned long long keys[];   // 256 * KMAX
    const int qi = blockIdx.x;
    const float* row = dmat + (size_t)qi * n;
    unsigned long long mine[KMAX];
#pragma unroll
    for (int s = 0; s < KMAX; ++s) mine[s] = ~0ull;
    for (int j = threadIdx.x; j < n; j += 256) {
        unsigned long long key = ((unsigned long long)__float_as_uint(row[j]) << 32) | (unsigned)j;
        if (key < mine[KMAX - 1]) {
#pragma unroll
            for (int s = 0; s < KMAX; ++s) {   // sorted insertion, static register indices
                const unsigned long long cur = mine[s];
                const bool take = key < cur;
                mine[s] = take ? key : cur;
                key = take ? cur : key;
            }
        }
    }
    constexpr int N = 256 * KMAX;
#pragma unroll
    for (int s = 0; s < KMAX; ++s) keys[s * 256 + threadIdx.x] = mine[s];
    __syncthreads();
    for (int size = 2; size <= N; size <<= 1)
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int i = threadIdx.x; i < N / 2; i += 256) {
                const int lo = (i / stride) * 2 * stride + (i % stride), hi = lo + stride;
                const bool up = ((lo / size) & 1) == 0;
                const unsigned long long a = keys[lo], b = keys[hi];
                if ((a > b) == up) { keys[lo] = b; keys[hi] = a; }
            }
            __syncthreads();
        }
    if (threadIdx.x < k) {
        const unsigned long long key = keys[threadIdx.x];
        const bool none = key == ~0ull;                      // fewer than k database rows
        out_idx[(size_t)qi * k + threadIdx.x] = none ? -1 : (int)(unsigned)(key & 0xffffffffull);
        out_d2[(size_t)qi * k + threadIdx.x] = none ? INFINITY : __uint_as_float((unsigned)(key >> 32));
    }
}

inline int grid_for(size_t n) { size_t b = (n + 255) / 256; return (int)(b > 4096 ? 4096 : (b ? b : 1)); }

}  // namespace

extern "C" {

int mrs_disco_descriptor(mrs_ctx* ctx, const float* d_bev, int32_t batch, int32_t num_height, int32_t num_ring,
                         int32_t num_sector, int32_t col, float* d_signature, float* d_spectrum, mrs_stream stream)
{
    MRS_REQUIRE(ctx && d_bev && d_signature && d_spectrum, "null pointer");
    MRS_REQUIRE(batch > 0 && num_height > 0 && num_ring > 0 && num_sector > 0, "sizes must be positive");
    MRS_REQUIRE(col > 0 && 2 * col <= num_ring && 2 * col <= num_sector, "crop does not fit the spectrum");
    MRS_HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t s = (hipStream_t)stream;
    const int cells = num_ring * num_sector;
    float2* spec = reinterpret_cast<float2*>(d_spectrum);
    MRS_REQUIRE(batch <= mrs::kMaxGridY, "at most 65535 scans per call (split the batch)");
    hipLaunchKernelGGL(k_height_sum, dim3(grid_for(cells), batch), dim3(256), 0, s, d_bev, num_height, cells, spec);
    int st = fft2_inplace(ctx, spec, num_ring, num_sector, batch, false, s);
    if (st != MRS_OK) return st;
    hipLaunchKernelGGL(k_disco_finish, dim3(grid_for(cells), batch), dim3(256), 0, s, spec, num_ring, num_sector, col,
                       1.0f / sqrtf((float)cells), d_signature);
    MRS_HIP_TRY(hipGetLastError());
    return MRS_OK;
}

int mrs_disco_phase_corr(mrs_ctx* ctx, const float* d_a, const float* d_b, int32_t n_pairs, int32_t num_ring,
                         int32_t num_sector, int32_t* d_yaw, float* d_corr, mrs_stream stream)
{
    MRS_REQUIRE(ctx && d_a && d_b && d_yaw, "null pointer");
    MRS_REQUIRE(n_pairs > 0 && num_ring > 0 && num_sector > 0, "sizes must be positive");
    MRS_HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t s = (hipStream_t)stream;
    const size_t n = (size_t)n_pairs * num_ring * num_sector;
    mrs::Scratch prod, arg;
    int st = prod.alloc(n * sizeof(float2), s);
    if (st != MRS_OK) return st;
    st = arg.alloc((size_t)n_pairs * sizeof(int), s);
    if (st != MRS_OK) return st;
    hipLaunchKernelGGL(k_conj_product, dim3(grid_for(n)), dim3(256), 0, s, reinterpret_cast<const float2*>(d_a),
                       reinterpret_cast<const float2*>(d_b), n, prod.as<float2>());
    st = fft2_inplace(ctx, prod.as<float2>(), num_ring, num_sector, n_pairs, true, s);
    if (st != MRS_OK) return st;
    hipLaunchKernelGGL(k_mag_shift_argmax, dim3(n_pairs), dim3(1024), 0, s, prod.as<float2>(), 1, num_ring, num_sector,
                       1.0f / sqrtf((float)(num_ring * num_sector)), 1e-15f, d_yaw, (float*)nullptr, d_corr);
    MRS_HIP_TRY(hipGetLastError());
    // flat argmax -> % num_sector (disco_ros/main.py:269-270): done by the caller-side mirror to
    // keep the raw index available; here we return the raw flat index.
    return MRS_OK;
}

int mrs_bev_translation(mrs_ctx* ctx, const float* d_a, const float* d_b, int32_t n_pairs, int32_t channels,
                        int32_t height, int32_t width, int32_t* d_arg, float* d_max, float* d_corr, mrs_stream stream)
{
    MRS_REQUIRE(ctx && d_a && d_b && d_arg, "null pointer");
    MRS_REQUIRE(n_pairs > 0 && channels > 0 && height > 0 && width > 0, "sizes must be positive");
    MRS_HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t s = (hipStream_t)stream;
    const int imgs = n_pairs * channels;
    const size_t n = (size_t)imgs * height * width;
    mrs::Scratch na, fa, fb;
    int st = na.alloc(n * sizeof(float), s);
    if (st != MRS_OK) return st;
    st = fa.alloc(n * sizeof(float2), s);
    if (st != MRS_OK) return st;
    st = fb.alloc(n * sizeof(float2), s);
    if (st != MRS_OK) return st;
    const int group = channels * height * width;  // one mean/std per [C,H,W] descriptor (util.py:429-430)
    for (int which = 0; which < 2; ++which) {
        st = mrs_normalize_groups(ctx, which ? d_b : d_a, na.as<float>(), n_pairs, group, stream);
        if (st != MRS_OK) return st;
        float2* f = which ? fb.as<float2>() : fa.as<float2>();
        hipLaunchKernelGGL(k_real_to_complex, dim3(grid_for(n)), dim3(256), 0, s, na.as<float>(), n, f);
        st = fft2_inplace(ctx, f, height, width, imgs, false, s);
        if (st != MRS_OK) return st;
    }
    hipLaunchKernelGGL(k_conj_product, dim3(grid_for(n)), dim3(256), 0, s, fa.as<float2>(), fb.as<float2>(), n, fa.as<float2>());
    st = fft2_inplace(ctx, fa.as<float2>(), height, width, imgs, true, s);
    if (st != MRS_OK) return st;
    // two ortho forwards (1/sqrt(HW) each) and one ortho inverse (1/sqrt(HW)) on unnormalised transforms
    const float hw = (float)(height * width);
    hipLaunchKernelGGL(k_mag_shift_argmax, dim3(n_pairs), dim3(1024), 0, s, fa.as<float2>(), channels, height, width,
                       1.0f / (hw * sqrtf(hw)), 0.0f, d_arg, d_max, d_corr);
    MRS_HIP_TRY(hipGetLastError());
    return MRS_OK;
}


int mrs_disco_rel_ori_literal(mrs_ctx* ctx, const float* d_a, const float* d_b, int32_t n_pairs, int32_t height,
                              int32_t width, float* d_rel_angle_deg, mrs_stream stream)
{
    MRS_REQUIRE(ctx && d_a && d_b && d_rel_angle_deg, "null pointer");
    MRS_REQUIRE(n_pairs > 0 && height > 0 && width > 0, "sizes must be positive");
    MRS_HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t s = (hipStream_t)stream;
    const size_t n = (size_t)n_pairs * height * width;
    mrs::Scratch cross;
    int st = cross.alloc(n * sizeof(double2), s);
    if (st != MRS_OK) return st;
    hipLaunchKernelGGL(k_relori_cross, dim3(grid_for(n)), dim3(256), 0, s, reinterpret_cast<const float2*>(d_a),
                       reinterpret_cast<const float2*>(d_b), n, cross.as<double2>());
    st = fft2_inplace(ctx, cross.p, height, width, n_pairs, true, s, true);
    if (st != MRS_OK) return st;
    hipLaunchKernelGGL(k_relori_argmax, dim3(n_pairs), dim3(1024), 0, s, cross.as<double2>(), height, width, d_rel_angle_deg);
    MRS_HIP_TRY(hipGetLastError());
    return MRS_OK;
}

int mrs_signature_search(mrs_ctx* ctx, const float* d_query, int32_t n_query, const float* d_db, int32_t n_db, int32_t dim,
                         int32_t* d_index, float* d_dist2, mrs_stream stream)
{
    MRS_REQUIRE(ctx && d_query && d_db && d_index && d_dist2, "null pointer");
    MRS_REQUIRE(n_query > 0 && n_db > 0 && dim > 0, "sizes must be positive");
    MRS_HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t s = (hipStream_t)stream;
    mrs::Scratch best;
    int st = best.alloc((size_t)n_query * sizeof(unsigned long long), s);
    if (st != MRS_OK) return st;
    MRS_HIP_TRY(hipMemsetAsync(best.p, 0xff, (size_t)n_query * sizeof(unsigned long long), s));
    MRS_REQUIRE((n_query + kSigTile - 1) / kSigTile <= mrs::kMaxGridY, "too many queries per call (split the batch)");
    hipLaunchKernelGGL(k_signature_tile<false>, dim3((n_db + kSigTile - 1) / kSigTile, (n_query + kSigTile - 1) / kSigTile), dim3(256), 0, s,
                       d_query, n_query, d_db, n_db, dim, best.as<unsigned long long>(), (float*)nullptr);
    hipLaunchKernelGGL(k_signature_unpack, dim3((n_query + 255) / 256), dim3(256), 0, s, best.as<unsigned long long>(), n_query,
                       d_index, d_dist2);
    MRS_HIP_TRY(hipGetLastError());
    return MRS_OK;
}

int mrs_signature_knn(mrs_ctx* ctx, const float* d_query, int32_t n_query, const float* d_db, int32_t n_db, int32_t dim, int32_t k,
                      int32_t* d_index, float* d_dist2, mrs_stream stream)
{
    MRS_REQUIRE(ctx && d_query && d_db && d_index && d_dist2, "null pointer");
    MRS_REQUIRE(n_query > 0 && n_db > 0 && dim > 0, "sizes must be positive");
    MRS_REQUIRE(k >= 1 && k <= 32, "k must be in [1, 32]");
    MRS_HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t s = (hipStream_t)stream;
    // the distance matrix of a chunk of queries stays below 1 GiB (256 M floats): 2 560 queries at a time against 100 k rows
    const int chunk = (int)std::max<size_t>(1, std::min<size_t>((size_t)n_query, ((size_t)1 << 28) / (size_t)n_db));
    MRS_REQUIRE((chunk + kSigTile - 1) / kSigTile <= mrs::kMaxGridY, "too many queries per call (split the batch)");
    mrs::Scratch dmat;
    int st = dmat.alloc((size_t)chunk * n_db * sizeof(float), s);
    if (st != MRS_OK) return st;
    for (int q0 = 0; q0 < n_query; q0 += chunk) {
        const int nq = std::min(chunk, n_query - q0);
        const float* q = d_query + (size_t)q0 * dim;
        int32_t* oi = d_index + (size_t)q0 * k;
        float* od = d_dist2 + (size_t)q0 * k;
        hipLaunchKernelGGL(k_signature_tile<true>, dim3((n_db + kSigTile - 1) / kSigTile, (nq + kSigTile - 1) / kSigTile), dim3(256), 0, s, q, nq,
                           d_db, n_db, dim, (unsigned long long*)nullptr, dmat.as<float>());
        if (k <= 8) {
            hipLaunchKernelGGL(k_signature_select<8>, dim3(nq), dim3(256), 256 * 8 * sizeof(unsigned long long), s, dmat.as<float>(), n_db, k, oi, od);
        } else if (k <= 16) {
            hipLaunchKernelGGL(k_signature_select<16>, dim3(nq), dim3(256), 256 * 16 * sizeof(unsigned long long), s, dmat.as<float>(), n_db, k, oi, od);
        } else {
            auto kern = k_signature_select<32>;
            const size_t lds = 256 * 32 * sizeof(unsigned long long);
            MRS_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL(kern, dim3(nq), dim3(256), lds, s, dmat.as<float>(), n_db, k, oi, od);
        }
    }
    MRS_HIP_TRY(hipGetLastError());
    return MRS_OK;
}

int mrs_rotate_nearest(mrs_ctx* ctx, const float* d_img, int32_t n_images, int32_t images_per_angle, int32_t height,
                       int32_t width, const float* d_angle_deg, float* d_out, mrs_stream stream)
{
    MRS_REQUIRE(ctx && d_img && d_angle_deg && d_out, "null pointer");
    MRS_REQUIRE(n_images > 0 && images_per_angle > 0 && height > 0 && width > 0, "sizes must be positive");
    MRS_REQUIRE(d_img != d_out, "in-place rotation is not supported");
    MRS_HIP_TRY(hipSetDevice(ctx->device));
    MRS_REQUIRE(n_images <= mrs::kMaxGridY, "at most 65535 images per call (split the batch)");
    hipLaunchKernelGGL(k_rotate_nearest, dim3(grid_for((size_t)height * width), n_images), dim3(256), 0,
                       (hipStream_t)stream, d_img, height, width, d_angle_deg, images_per_angle, d_out);
    MRS_HIP_TRY(hipGetLastError());
    return MRS_OK;
}

}  // extern "C"
