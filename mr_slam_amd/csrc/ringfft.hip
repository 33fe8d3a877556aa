// ringfft.hip -- FFT-domain RING rotation correlation for gfx950,
// specialised for the reference configuration num_ring = num_sector = 120 (RING_ros/config.py:7-8).
//
// What the reference does per candidate (RING_ros/util.py:362-374): corr = ifft_angle(a * conj(b)),
// |corr|, sum over detectors, fftshift, max/argmax, with a, b the TIRING spectra (util.py:198).
// Here the database holds the Hermitian HALF spectrum of every normalised sinogram
// ([61][120] complex64, ortho scaled = the first 61 angle-frequency rows of the reference's TIRING,
// 58 560 B per entry instead of 115 200 B), the query's half spectrum sits in LDS, and every lane
// owns one detector column: it streams its 61 complex values of the candidate from HBM (coalesced
// across lanes), forms the conjugate product, and runs a 120-point real inverse transform entirely in
// registers (half-length trick + generated straight-line complex FFT-60, csrc/fft_codelets.hpp).
// The 120 |corr| values per lane are summed across the 120 lanes with a wave reduce-scatter.
// ~0.3 MFLOP and 58.56 KB per pair: with one query the sweep is near both the HBM and the VALU limit
// (64 M pairs/s = 3.8 TB/s); with several queries per launch the database is fetched once (L2 shares it
// between the query rows) and the in-register FFT (VALU) binds at ~90-100 M pairs/s.  RING++ descriptors
// ([C][61][120]) run the same code in a channel loop; fp16 replicas of the database are accepted as well.
#include <algorithm>
#include <cmath>

#include <hip/hip_fp16.h>

#include "common.hpp"
#include "fft_codelets.hpp"

namespace {

constexpr int kA = 120;        // angles
constexpr int kD = 120;        // detectors
constexpr int kHalf = 61;      // kA / 2 + 1
constexpr int kSlotThreads = 128;

// Z[k] = E + i O for one k, from P[k] = (ar, ai) and P[60-k] = (br, bi):
// E = P[k] + conj(P[60-k]),  O = (P[k] - conj(P[60-k])) * exp(+2 pi i k / 120)
template <int K>
__device__ __forceinline__ void irfft_pre(float ar, float ai, float br, float bi, float& zr, float& zi)
{
    const float er = ar + br, ei = ai - bi;
    const float dr = ar - br, di = ai + bi;
    const float orr = dr * kCos120[K] - di * kSin120[K];
    const float oi = dr * kSin120[K] + di * kCos120[K];
    zr = er - oi;
    zi = ei + orr;
}

// Conjugate product of two half spectra streamed straight into the pre-processed FFT input, then the
// 60-point inverse codelet: on return x[2m] = re[m], x[2m+1] = im[m] with
// x[n] = sum_{k=0}^{119} P_full[k] exp(+2 pi i k n / 120), P = a * conj(b).
// load(k, ar, ai, br, bi) fetches a[k] and b[k] of this lane's column.
template <class Load>
__device__ __forceinline__ void corr_irfft120(Load load, float (&re)[60], float (&im)[60])
{
    auto prod = [&](int k, float& pr, float& pi) {
        float ar, ai, br, bi;
        load(k, ar, ai, br, bi);
        pr = ar * br + ai * bi;   // a * conj(b)
        pi = ai * br - ar * bi;
    };
    {
        float p0r, p0i, p60r, p60i;
        prod(0, p0r, p0i);
        prod(60, p60r, p60i);
        irfft_pre<0>(p0r, p0i, p60r, p60i, re[0], im[0]);
    }
#define MRS_IRFFT_PAIR(J)                                                   \
    {                                                                       \
        float ur, ui, wr, wi;                                               \
        prod(J, ur, ui);                                                    \
        prod(60 - J, wr, wi);                                               \
        irfft_pre<J>(ur, ui, wr, wi, re[J], im[J]);                         \
        irfft_pre<60 - J>(wr, wi, ur, ui, re[60 - J], im[60 - J]);          \
    }
    MRS_IRFFT_PAIR(1) MRS_IRFFT_PAIR(2) MRS_IRFFT_PAIR(3) MRS_IRFFT_PAIR(4) MRS_IRFFT_PAIR(5) MRS_IRFFT_PAIR(6)
    MRS_IRFFT_PAIR(7) MRS_IRFFT_PAIR(8) MRS_IRFFT_PAIR(9) MRS_IRFFT_PAIR(10) MRS_IRFFT_PAIR(11) MRS_IRFFT_PAIR(12)
    MRS_IRFFT_PAIR(13) MRS_IRFFT_PAIR(14) MRS_IRFFT_PAIR(15) MRS_IRFFT_PAIR(16) MRS_IRFFT_PAIR(17) MRS_IRFFT_PAIR(18)
    MRS_IRFFT_PAIR(19) MRS_IRFFT_PAIR(20) MRS_IRFFT_PAIR(21) MRS_IRFFT_PAIR(22) MRS_IRFFT_PAIR(23) MRS_IRFFT_PAIR(24)
    MRS_IRFFT_PAIR(25) MRS_IRFFT_PAIR(26) MRS_IRFFT_PAIR(27) MRS_IRFFT_PAIR(28) MRS_IRFFT_PAIR(29)
#undef MRS_IRFFT_PAIR
    {
        float pr, pi;
        prod(30, pr, pi);
        irfft_pre<30>(pr, pi, pr, pi, re[30], im[30]);
    }
    cfft60_inv(re, im);
}

// real x (x[2m] = re[m], x[2m+1] = im[m]) -> X[0..60] = sum_n x[n] exp(-2 pi i k n / 120), handed to
// store(k, xr, xi) one frequency at a time (keeps the register footprint at the codelet's)
template <class Store>
__device__ __forceinline__ void rfft120(float (&re)[60], float (&im)[60], Store store)
{
    cfft60_fwd(re, im);
#pragma unroll
    for (int k = 0; k <= 60; ++k) {
        const int k0 = k % 60, k1 = (60 - k) % 60;
        const float ar = re[k0], ai = im[k0];
        const float br = re[k1], bi = -im[k1];                     // conj(Z[60-k])
        const float sr = ar + br, si = ai + bi;
        const float dr = ar - br, di = ai - bi;
        // X = 0.5 * S - 0.5 i * exp(-i theta_k) * D,   exp(-i theta) = cos - i sin
        const float tr = dr * kCos120[k] + di * kSin120[k];
        const float ti = di * kCos120[k] - dr * kSin120[k];
        store(k, 0.5f * (sr + ti), 0.5f * (si - tr));
    }
}

// R2: half spectrum (ortho) of normalised sinograms.  grid = images, 128 lanes (120 columns).
// out16 (optional): the same values rounded to nearest-even fp16 = the exchange / replica format (29 280 B)
__global__ __launch_bounds__(kSlotThreads) void k_ring_half_spectrum(const float* __restrict__ x, float2* __restrict__ out,
                                                                     __half2* __restrict__ out16)
{
    const int d = min((int)threadIdx.x, kD - 1);
    const float* src = x + (size_t)blockIdx.x * kA * kD + d;
    float re[60], im[60];
#pragma unroll
    for (int m = 0; m < 60; ++m) {
        re[m] = src[(2 * m) * kD];
        im[m] = src[(2 * m + 1) * kD];
    }
    const bool live = threadIdx.x < kD;
    const float sc = 0.09128709291752769f;  // 1/sqrt(120)
    const size_t o = (size_t)blockIdx.x * kHalf * kD + d;
    rfft120(re, im, [&](int k, float xr, float xi) {
        if (!live) return;
        const float2 v = make_float2(xr * sc, xi * sc);
        if (out) out[o + k * kD] = v;
        if (out16) out16[o + k * kD] = __floats2half2_rn(v.x, v.y);
    });
}

// Sum over the 64 lanes of a wave of |x[n]|, n = 0..119, where x[2m] = re[m], x[2m+1] = im[m]
// (reduce-scatter: 6 halving steps).  On return lane L holds the totals of n = 2L (w0) and 2L+1 (w1).
__device__ __forceinline__ void wave_abs_reduce_scatter(const float (&re)[60], const float (&im)[60], bool live,
                                                        float& w0, float& w1)
{
    const int lane = threadIdx.x & 63;
    float w[64];
    {
        const bool hi = (lane & 32) != 0;   // step 0: indices [0,64) stay in the low half-wave, [64,128) in the high
#pragma unroll
        for (int i = 0; i < 64; ++i) {
            const float lo_v = live ? fabsf((i & 1) ? im[i >> 1] : re[i >> 1]) : 0.0f;
            const int j = 64 + i;
            const float hi_v = (j < 120 && live) ? fabsf((j & 1) ? im[(j >> 1) % 60] : re[(j >> 1) % 60]) : 0.0f;
            const float send = hi ? lo_v : hi_v;
            const float mine = hi ? hi_v : lo_v;
            w[i] = mine + __shfl_xor(send, 32, 64);
        }
    }
#pragma unroll
    for (int step = 1; step < 6; ++step) {
        const int keep = 64 >> step;
        const int mask = 32 >> step;
        const bool hi = (lane & mask) != 0;
#pragma unroll
        for (int i = 0; i < keep; ++i) {
            const float send = hi ? w[i] : w[keep + i];
            const float mine = hi ? w[keep + i] : w[i];
            w[i] = mine + __shfl_xor(send, mask, 64);
        }
    }
    w0 = w[0];
    w1 = w[1];
}

struct FftCorrP {
    int nq, ndb, pairwise, channels;
    float denom;  // 0.15 * C * A * D
};

__device__ __forceinline__ float2 load_spec(const float2* p) { return *p; }
__device__ __forceinline__ float2 load_spec(const __half2* p) { return __half22float2(*p); }

// grid = (blocks, nq); NSLOT pair streams of 128 lanes.  QLDS: the (single-channel) query spectrum is staged in
// LDS and shared by the slots; otherwise (pairwise mode, C > 1) it is read through L2 like the candidate.
// DBT = float2 (the database format) or __half2 (fp16 replicas received from other ranks).
// Descriptors with C channels are [C][61][120]: |corr| is summed over channels and detectors
// (fast_corr_RINGplusplus, RING_ros/util.py:337-358; C = 1: fast_corr, util.py:362-374).
// PAIRWISE: candidate = query index (one per block row); MULTI: runtime channel count (else C = 1, no loop).
template <int NSLOT, bool QLDS, bool PAIRWISE, bool MULTI, typename DBT>
__global__ __launch_bounds__(NSLOT* kSlotThreads) void k_ring_corr_fft(const float2* __restrict__ Q, const DBT* __restrict__ DB,
                                                                       FftCorrP p, float* __restrict__ dist,
                                                                       int* __restrict__ angle, float* __restrict__ corr_out)
{
    extern __shared__ __attribute__((aligned(16))) float2 qs[];  // [61][120] query half spectrum (QLDS)
    __shared__ float xbuf[NSLOT][128];
    const int q = blockIdx.y;
    const int slot = threadIdx.x / kSlotThreads;
    const int t = threadIdx.x % kSlotThreads;
    const int wave_in_slot = t >> 6, lane = t & 63;
    const int d = min(t, kD - 1);
    const bool live_col = t < kD;
    const int C = MULTI ? p.channels : 1;
    const size_t entry = (size_t)C * kHalf * kD;
    const float2* qsrc = Q + (size_t)q * entry;
    if (QLDS) {
        for (int i = threadIdx.x; i < kHalf * kD; i += NSLOT * kSlotThreads) qs[i] = qsrc[i];
        __syncthreads();
    }
    const int ncand = PAIRWISE ? 1 : p.ndb;
    const int stride = gridDim.x * NSLOT;
    const int rounds = (ncand + stride - 1) / stride;
    for (int r = 0; r < rounds; ++r) {
        const int cand = (r * gridDim.x + blockIdx.x) * NSLOT + slot;
        const bool live = cand < ncand;
        float v[2] = {0.0f, 0.0f};
        for (int c = 0; c < C; ++c) {
            float re[60], im[60];
            if (live) {
                const DBT* b = DB + (size_t)(PAIRWISE ? q : cand) * entry + (size_t)c * kHalf * kD + d;
                const float2* a = qsrc + (size_t)c * kHalf * kD + d;
                corr_irfft120([&](int k, float& ar, float& ai, float& br, float& bi) {
                    const float2 bv = load_spec(b + k * kD);
                    const float2 av = QLDS ? qs[k * kD + d] : a[k * kD];
                    ar = av.x; ai = av.y; br = bv.x; bi = bv.y;
                }, re, im);
            } else {
#pragma unroll
                for (int m = 0; m < 60; ++m) { re[m] = 0.0f; im[m] = 0.0f; }
            }
            float w0, w1;
            wave_abs_reduce_scatter(re, im, live && live_col, w0, w1);
            v[0] += w0;
            v[1] += w1;
        }
        if (wave_in_slot == 1) { xbuf[slot][2 * lane] = v[0]; xbuf[slot][2 * lane + 1] = v[1]; }
        __syncthreads();
        if (wave_in_slot == 0 && live) {
            const float sc = 0.09128709291752769f;  // ifft ortho factor 1/sqrt(120)
            const float s0 = (v[0] + xbuf[slot][2 * lane]) * sc, s1 = (v[1] + xbuf[slot][2 * lane + 1]) * sc;
            // fftshift: shifted index m = (n + 60) % 120; first maximum in m order (util.py:367-371)
            const int n0 = 2 * lane, n1 = 2 * lane + 1;
            const int m0 = n0 < 60 ? n0 + 60 : n0 - 60, m1 = n1 < 60 ? n1 + 60 : n1 - 60;
            const size_t o = (size_t)q * (PAIRWISE ? 1 : p.ndb) + (PAIRWISE ? 0 : cand);
            float best = -1.0f;
            int bm = 1 << 30;
            if (lane < 60) {
                if (corr_out) { corr_out[o * kA + m0] = s0; corr_out[o * kA + m1] = s1; }
                best = s0; bm = m0;
                if (s1 > best || (s1 == best && m1 < bm)) { best = s1; bm = m1; }
            }
            for (int off = 32; off > 0; off >>= 1) {
                const float ob = __shfl_xor(best, off, 64);
                const int om = __shfl_xor(bm, off, 64);
                if (ob > best || (ob == best && om < bm)) { best = ob; bm = om; }
            }
            if (lane == 0) {
                dist[o] = 1.0f - best / p.denom;
                angle[o] = kA / 2 - bm;
            }
        }
        __syncthreads();  // xbuf reuse
    }
}

// New-descriptor path of a loop check in ONE launch (row R2 + C1 for pairs): half spectrum of the freshly
// normalised sinogram (written out: it becomes a database entry / travels to the other ranks) and, straight from
// the LDS copy of that spectrum, its correlation with the candidate's.  Same arithmetic, op for op, as
// k_ring_half_spectrum followed by the pairwise k_ring_corr_fft (bitwise identical results), one launch and one
// round trip of the spectrum through HBM less.  grid = pairs, 128 lanes.
// CT = float2 (exact database entries) or __half2 (fp16 replicas received from other ranks); cand_idx (optional):
// the candidate of pair i is row cand_idx[i] of `cand` (a pre-selected row of the replicated database) instead of row i.
template <typename CT>
__global__ __launch_bounds__(kSlotThreads) void k_ring_spec_corr_pairs(const float* __restrict__ x, const CT* __restrict__ cand,
                                                                       const int* __restrict__ cand_idx, int n_db,
                                                                       float2* __restrict__ out, __half2* __restrict__ out16,
                                                                       float denom, float* __restrict__ dist,
                                                                       int* __restrict__ angle)
{
    extern __shared__ __attribute__((aligned(16))) float2 qs[];  // [61][120]
    __shared__ float xbuf[128];
    const int pair = blockIdx.x;
    const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
    const int d = min(t, kD - 1);
    const bool live_col = t < kD;
    {
        const float* src = x + (size_t)pair * kA * kD + d;
        float re[60], im[60];
#pragma unroll
        for (int m = 0; m < 60; ++m) {
            re[m] = src[(2 * m) * kD];
            im[m] = src[(2 * m + 1) * kD];
        }
        const float sc = 0.09128709291752769f;  // 1/sqrt(120)
        const size_t o = (size_t)pair * kHalf * kD + d;
        rfft120(re, im, [&](int k, float xr, float xi) {
            if (!live_col) return;
            const float2 v = make_float2(xr * sc, xi * sc);
            qs[k * kD + d] = v;
            if (out) out[o + k * kD] = v;
            if (out16) out16[o + k * kD] = __floats2half2_rn(v.x, v.y);
        });
    }
    __syncthreads();
    float re[60], im[60];
    const int row = cand_idx ? cand_idx[pair] : pair;
    if (cand_idx && (row < 0 || row >= n_db)) {   // no such database row: the new spectrum is still written, the score says "no match"
        if (t == 0) { dist[pair] = INFINITY; angle[pair] = 0; }
        return;
    }
    const CT* b = cand + (size_t)row * kHalf * kD + d;
    corr_irfft120([&](int k, float& ar, float& ai, float& br, float& bi) {
        const float2 bv = load_spec(b + k * kD);
        const float2 av = qs[k * kD + d];
        ar = av.x; ai = av.y; br = bv.x; bi = bv.y;
    }, re, im);
    float v0, v1;
    wave_abs_reduce_scatter(re, im, live_col, v0, v1);
    if (wave == 1) { xbuf[2 * lane] = v0; xbuf[2 * lane + 1] = v1; }
    __syncthreads();
    if (wave == 0) {
        const float sc = 0.09128709291752769f;
        const float s0 = (v0 + xbuf[2 * lane]) * sc, s1 = (v1 + xbuf[2 * lane + 1]) * sc;
        const int n0 = 2 * lane, n1 = 2 * lane + 1;
        const int m0 = n0 < 60 ? n0 + 60 : n0 - 60, m1 = n1 < 60 ? n1 + 60 : n1 - 60;
        float best = -1.0f;
        int bm = 1 << 30;
        if (lane < 60) {
            best = s0; bm = m0;
            if (s1 > best || (s1 == best && m1 < bm)) { best = s1; bm = m1; }
        }
        for (int off = 32; off > 0; off >>= 1) {
            const float ob = __shfl_xor(best, off, 64);
            const int om = __shfl_xor(bm, off, 64);
            if (ob > best || (ob == best && om < bm)) { best = ob; bm = om; }
        }
        if (lane == 0) {
            dist[pair] = 1.0f - best / denom;
            angle[pair] = kA / 2 - bm;
        }
    }
}

}  // namespace

template <typename CT>
static int spectrum_corr_pairs_launch(mrs_ctx* ctx, const float* d_norm_sino, const CT* cand, const int32_t* d_cand_index,
                                      int32_t n_db, int32_t n_pairs, int32_t n_angles, int32_t det, float* d_half_spec, void* d_half_spec_f16,
                                      float* d_dist, int32_t* d_angle, mrs_stream stream)
{
    MRS_REQUIRE(ctx && d_norm_sino && cand && d_dist && d_angle, "null pointer");
    MRS_REQUIRE(n_pairs > 0, "n_pairs must be positive");
    if (n_angles != kA || det != kD) {
        mrs::set_error("ring_spectrum_corr_pairs is specialised for 120 x 120 (got %d x %d)", n_angles, det);
        return MRS_ERR_UNSUPPORTED;
    }
    MRS_HIP_TRY(hipSetDevice(ctx->device));
    const size_t lds = (size_t)kHalf * kD * sizeof(float2);
    MRS_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_ring_spec_corr_pairs<CT>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)lds));
    hipLaunchKernelGGL(k_ring_spec_corr_pairs<CT>, dim3(n_pairs), dim3(kSlotThreads), lds, (hipStream_t)stream, d_norm_sino, cand,
                       d_cand_index, n_db, reinterpret_cast<float2*>(d_half_spec), reinterpret_cast<__half2*>(d_half_spec_f16),
                       (float)(0.15 * kA * kD), d_dist, d_angle);
    MRS_HIP_TRY(hipGetLastError());
    return MRS_OK;
}

extern "C" {

static int half_spectrum_launch(mrs_ctx* ctx, const float* d_norm_sino, int32_t n_img, int32_t n_angles, int32_t det,
                                float* d_half_spec, void* d_half_spec_f16, mrs_stream stream)
{
    MRS_REQUIRE(ctx && d_norm_sino && (d_half_spec || d_half_spec_f16), "null pointer");
    MRS_REQUIRE(n_img > 0, "n_img must be positive");
    if (n_angles != kA || det != kD) {
        mrs::set_error("ring_half_spectrum is specialised for 120 x 120 (got %d x %d)", n_angles, det);
        return MRS_ERR_UNSUPPORTED;
    }
    MRS_HIP_TRY(hipSetDevice(ctx->device));
    hipLaunchKernelGGL(k_ring_half_spectrum, dim3(n_img), dim3(kSlotThreads), 0, (hipStream_t)stream, d_norm_sino,
                       reinterpret_cast<float2*>(d_half_spec), reinterpret_cast<__half2*>(d_half_spec_f16));
    MRS_HIP_TRY(hipGetLastError());
    return MRS_OK;
}

int mrs_ring_half_spectrum(mrs_ctx* ctx, const float* d_norm_sino, int32_t n_img, int32_t n_angles, int32_t det,
                           float* d_half_spec, mrs_stream stream)
{
    MRS_REQUIRE(d_half_spec, "null pointer");
    return half_spectrum_launch(ctx, d_norm_sino, n_img, n_angles, det, d_half_spec, nullptr, stream);
}

int mrs_ring_half_spectrum_f16(mrs_ctx* ctx, const float* d_norm_sino, int32_t n_img, int32_t n_angles, int32_t det,
                               float* d_half_spec, void* d_half_spec_f16, mrs_stream stream)
{
    return half_spectrum_launch(ctx, d_norm_sino, n_img, n_angles, det, d_half_spec, d_half_spec_f16, stream);
}

extern "C++" {
template <typename DBT>
static int corr_fft_launch(mrs_ctx* ctx, const float* d_q, int32_t n_q, const void* d_db, int32_t n_db, int32_t channels,
                           float* d_dist, int32_t* d_angle, float* d_corr, mrs_stream stream, bool pairwise)
{
    MRS_REQUIRE(ctx && d_q && d_db && d_dist && d_angle, "null pointer");
    MRS_REQUIRE(n_q > 0 && n_db > 0 && channels > 0, "counts must be positive");
    MRS_HIP_TRY(hipSetDevice(ctx->device));
    FftCorrP p;
    p.nq = n_q; p.ndb = n_db; p.pairwise = pairwise ? 1 : 0; p.channels = channels;
    p.denom = (float)(0.15 * channels * kA * kD);
    constexpr int NSLOT = 2;
    hipStream_t s = (hipStream_t)stream;
    const float2* q2 = reinterpret_cast<const float2*>(d_q);
    const DBT* db2 = reinterpret_cast<const DBT*>(d_db);
    const size_t entry = (size_t)channels * kHalf * kD;
    for (int q0 = 0; q0 < n_q; q0 += mrs::kMaxGridY) {   // grid.y is limited to 65535 rows
        const int nq = std::min(n_q - q0, mrs::kMaxGridY);
        p.nq = nq;
        const size_t ro = (size_t)q0 * (pairwise ? 1 : n_db);
        const float2* qq = q2 + q0 * entry;
        const DBT* dd = pairwise ? db2 + q0 * entry : db2;
        float* dist_c = d_dist + ro;
        int32_t* angle_c = d_angle + ro;
        float* corr_c = d_corr ? d_corr + ro * kA : nullptr;
        if (pairwise) {
            if (channels == 1)
                hipLaunchKernelGGL((k_ring_corr_fft<1, false, true, false, DBT>), dim3(1, nq), dim3(kSlotThreads), 0, s, qq, dd, p, dist_c,
                                   angle_c, corr_c);
            else
                hipLaunchKernelGGL((k_ring_corr_fft<1, false, true, true, DBT>), dim3(1, nq), dim3(kSlotThreads), 0, s, qq, dd, p, dist_c,
                                   angle_c, corr_c);
        } else {
            int blocks = 2 * (ctx->num_cu > 0 ? ctx->num_cu : 256);
            if (n_q > 1) blocks = (blocks + n_q - 1) / n_q;
            const int need = (n_db + NSLOT - 1) / NSLOT;
            if (blocks > need) blocks = need;
            if (blocks < 1) blocks = 1;
            if (channels == 1) {
                const size_t lds = (size_t)kHalf * kD * sizeof(float2);
                auto kern = k_ring_corr_fft<NSLOT, true, false, false, DBT>;
                MRS_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                hipLaunchKernelGGL(kern, dim3(blocks, nq), dim3(NSLOT * kSlotThreads), lds, s, qq, dd, p, dist_c, angle_c, corr_c);
            } else {
                hipLaunchKernelGGL((k_ring_corr_fft<NSLOT, false, false, true, DBT>), dim3(blocks, nq), dim3(NSLOT * kSlotThreads), 0, s, qq, dd, p,
                                   dist_c, angle_c, corr_c);
            }
        }
    }
    MRS_HIP_TRY(hipGetLastError());
    return MRS_OK;
}
}  // extern "C++"

int mrs_ring_corr_fft_sweep(mrs_ctx* ctx, const float* d_query_spec, int32_t n_query, const float* d_db_spec,
                            int32_t n_db, float* d_dist, int32_t* d_angle, float* d_corr, mrs_stream stream)
{
    return corr_fft_launch<float2>(ctx, d_query_spec, n_query, d_db_spec, n_db, 1, d_dist, d_angle, d_corr, stream, false);
}

int mrs_ring_corr_fft_sweep_f16(mrs_ctx* ctx, const float* d_query_spec, int32_t n_query, const void* d_db_spec_f16,
                                int32_t n_db, float* d_dist, int32_t* d_angle, float* d_corr, mrs_stream stream)
{
    return corr_fft_launch<__half2>(ctx, d_query_spec, n_query, d_db_spec_f16, n_db, 1, d_dist, d_angle, d_corr, stream, false);
}

int mrs_ring_corr_fft_pairs(mrs_ctx* ctx, const float* d_a_spec, const float* d_b_spec, int32_t n_pairs, float* d_dist,
                            int32_t* d_angle, float* d_corr, mrs_stream stream)
{
    return corr_fft_launch<float2>(ctx, d_a_spec, n_pairs, d_b_spec, n_pairs, 1, d_dist, d_angle, d_corr, stream, true);
}

/* multi-channel (RING++) forms: descriptors are [channels][61][120] complex64 */
int mrs_ring_corr_fft_sweep_mc(mrs_ctx* ctx, const float* d_query_spec, int32_t n_query, const float* d_db_spec, int32_t n_db,
                               int32_t channels, float* d_dist, int32_t* d_angle, float* d_corr, mrs_stream stream)
{
    return corr_fft_launch<float2>(ctx, d_query_spec, n_query, d_db_spec, n_db, channels, d_dist, d_angle, d_corr, stream, false);
}

int mrs_ring_corr_fft_pairs_mc(mrs_ctx* ctx, const float* d_a_spec, const float* d_b_spec, int32_t n_pairs, int32_t channels,
                               float* d_dist, int32_t* d_angle, float* d_corr, mrs_stream stream)
{
    return corr_fft_launch<float2>(ctx, d_a_spec, n_pairs, d_b_spec, n_pairs, channels, d_dist, d_angle, d_corr, stream, true);
}

int mrs_ring_spectrum_corr_pairs(mrs_ctx* ctx, const float* d_norm_sino, const float* d_cand_spec, int32_t n_pairs,
                                 int32_t n_angles, int32_t det, float* d_half_spec, void* d_half_spec_f16, float* d_dist,
                                 int32_t* d_angle, mrs_stream stream)
{
    return spectrum_corr_pairs_launch<float2>(ctx, d_norm_sino, reinterpret_cast<const float2*>(d_cand_spec), nullptr, n_pairs, n_pairs, n_angles,
                                              det, d_half_spec, d_half_spec_f16, d_dist, d_angle, stream);
}

int mrs_ring_spectrum_corr_pairs_db(mrs_ctx* ctx, const float* d_norm_sino, const void* d_db_spec, int32_t db_is_f16, int32_t n_db,
                                    const int32_t* d_cand_index, int32_t n_pairs, int32_t n_angles, int32_t det, float* d_half_spec,
                                    void* d_half_spec_f16, float* d_dist, int32_t* d_angle, mrs_stream stream)
{
    MRS_REQUIRE(d_cand_index, "null candidate index");
    MRS_REQUIRE(n_db > 0, "n_db must be positive");
    return db_is_f16 ? spectrum_corr_pairs_launch<__half2>(ctx, d_norm_sino, reinterpret_cast<const __half2*>(d_db_spec), d_cand_index,
                                                          n_db, n_pairs, n_angles, det, d_half_spec, d_half_spec_f16, d_dist, d_angle, stream)
                     : spectrum_corr_pairs_launch<float2>(ctx, d_norm_sino, reinterpret_cast<const float2*>(d_db_spec), d_cand_index,
                                                          n_db, n_pairs, n_angles, det, d_half_spec, d_half_spec_f16, d_dist, d_angle, stream);
}

}  // extern "C"
