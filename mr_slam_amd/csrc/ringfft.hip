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
// 58.56 KB and ~1 400 packed / scalar VALU instructions per 120-lane column pass.  With several queries per launch the database is fetched
// once (L2 shares it between the query rows) and the in-register FFT (VALU) binds at 115-135 M pairs/s (k_ring_corr_fft: the candidate's column
// in registers).  With ONE query -- the node's loop, main_RING.py:133 -- the sweep is a stream of the database and runs as an LDS-DMA pipeline per
// wave (k_ring_sweep_dma, round 5: `global_load_lds_dwordx4` into a private ring of LDS slots, DMA-tiled entries): 98-105 M pairs/s = 0.72-0.77
// of the HBM peak (0.60 with the column in registers); RING++ descriptors ([C][61][120]) one channel per workgroup (16-17 M pairs/s = 0.72-0.74;
// k_ring_sweep_mc, channel-outer, for several queries).  fp16 replicas of the database are accepted as well.
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdlib>
#include <type_traits>

#include <hip/hip_fp16.h>

#include "common.hpp"
#include "fft_codelets.hpp"

namespace {

constexpr int kA = 120;        // angles
constexpr int kD = 120;        // detectors
constexpr int kHalf = 61;      // kA / 2 + 1
constexpr int kSlotThreads = 128;

constexpr int kLdsStride = 128;  // LDS row stride of a staged half spectrum: columns 120..127 are zero, so the 8 lanes past the
                                 // last detector of a 128-lane slot multiply by zero instead of being masked value by value

// A complex value is one v2f (re, im) in an aligned VGPR pair (csrc/fft_codelets.hpp): every line below is one packed
// fp32 instruction (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32, swizzles and signs in the operand modifiers).
__device__ __forceinline__ v2f cmul_conj(v2f a, v2f b)   // a * conj(b) = (ar br + ai bi, ai br - ar bi)
{
    const v2f t = a.yx * b.yy * (v2f){1.0f, -1.0f};
    return __builtin_elementwise_fma(a, b.xx, t);
}

__device__ __forceinline__ v2f twiddle120(int K, v2f d)   // d * exp(+2 pi i K / 120); K is a constant after unrolling
{
    const v2f t = d.yx * (v2f){-kSin120[K], kSin120[K]};
    return __builtin_elementwise_fma(d, (v2f){kCos120[K], kCos120[K]}, t);
}

// Pre-processing of the half-length inverse real transform for the pair (J, 60 - J) from P[J] = u and P[60-J] = w:
// Z[k] = E + i O with E = P[k] + conj(P[60-k]), O = (P[k] - conj(P[60-k])) exp(+2 pi i k / 120); the partner needs no
// arithmetic of its own: E' = conj(E), O' = conj(O)  =>  Z[60-J] = (Er + Oi, Or - Ei).
__device__ __forceinline__ void irfft_pre_pair(int J, v2f u, v2f w, v2f& zj, v2f& zp)
{
    const v2f e = __builtin_elementwise_fma(w, (v2f){1.0f, -1.0f}, u);
    const v2f d = __builtin_elementwise_fma(w, (v2f){-1.0f, 1.0f}, u);
    const v2f o = twiddle120(J, d);
    zj = __builtin_elementwise_fma(o.yx, (v2f){-1.0f, 1.0f}, e);
    zp = __builtin_elementwise_fma(e, (v2f){1.0f, -1.0f}, o.yx);
}

// Conjugate product of two half spectra streamed straight into the pre-processed FFT input, then the
// 60-point inverse codelet: on return the 120 real samples are x[m] = (sample 2m, sample 2m + 1) of
// s[n] = sum_{k=0}^{119} P_full[k] exp(+2 pi i k n / 120), P = a * conj(b).
// la(k) / lb(k) fetch a[k] and b[k] (k = 0..60) of this lane's column.  The a values (LDS or L2) of the NEXT group of
// frequency pairs are requested before the arithmetic of the current group: left to itself the scheduler places every read
// right in front of its use and the wave sits out the full latency 30 times per column.
constexpr int kPairsPerGroup = 5;   // 6 groups cover the pairs (J, 60 - J), J = 0..29; J = 30 rides with the last group
template <class LoadA, class LoadB>
__device__ __forceinline__ void corr_irfft120(LoadA la, LoadB lb, v2f (&x)[60])
{
    constexpr int G = kPairsPerGroup, NG = 30 / kPairsPerGroup;
    v2f abuf[2][2 * G + 1];
    auto request = [&](int g, v2f (&dst)[2 * G + 1]) {
#pragma unroll
        for (int j = 0; j < G; ++j) {
            dst[2 * j] = la(g * G + j);
            dst[2 * j + 1] = la(60 - (g * G + j));
        }
        if (g == NG - 1) dst[2 * G] = la(30);
    };
    request(0, abuf[0]);
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        if (g + 1 < NG) request(g + 1, abuf[(g + 1) & 1]);
        __builtin_amdgcn_sched_barrier(0);
        const v2f(&a)[2 * G + 1] = abuf[g & 1];
#pragma unroll
        for (int j = 0; j < G; ++j) {
            const int J = g * G + j;
            v2f zj, zp;
            irfft_pre_pair(J, cmul_conj(a[2 * j], lb(J)), cmul_conj(a[2 * j + 1], lb(60 - J)), zj, zp);
            x[J] = zj;
            if (J != 0) x[60 - J] = zp;
        }
        if (g == NG - 1) {
            const v2f pm = cmul_conj(a[2 * G], lb(30));
            v2f unused;
            irfft_pre_pair(30, pm, pm, x[30], unused);
        }
    }
    cfft60_inv(x);
}

// real samples (x[m] = (sample 2m, sample 2m + 1)) -> TWICE the spectrum, 2 X[k] = 2 sum_n s[n] exp(-2 pi i k n / 120),
// k = 0..60, handed to store(k, value) (the caller folds the exact factor 0.5 into its own scale).
// Z = FFT60(x); S = Z[k] + conj(Z[60-k]), T = (Z[k] - conj(Z[60-k])) exp(-2 pi i k / 120), 2 X[k] = S - i T; the partner
// frequency shares all of it: S' = conj(S), T' = conj(T)  =>  2 X[60-k] = (Sr - Ti, -Si - Tr).
__device__ __forceinline__ void rfft_post_pair(int K, v2f zk, v2f zp, v2f& xk, v2f& xp)
{
    const v2f sum = __builtin_elementwise_fma(zp, (v2f){1.0f, -1.0f}, zk);
    const v2f d = __builtin_elementwise_fma(zp, (v2f){-1.0f, 1.0f}, zk);
    const v2f tt = d.yx * (v2f){kSin120[K], -kSin120[K]};
    const v2f t = __builtin_elementwise_fma(d, (v2f){kCos120[K], kCos120[K]}, tt);      // d * (cos - i sin)
    xk = __builtin_elementwise_fma(t.yx, (v2f){1.0f, -1.0f}, sum);
    xp = __builtin_elementwise_fma(sum, (v2f){1.0f, -1.0f}, -t.yx);
}

template <class Store>
__device__ __forceinline__ void rfft120(v2f (&x)[60], Store store)
{
    cfft60_fwd(x);
#pragma unroll
    for (int K = 0; K <= 30; ++K) {
        v2f xk, xp;
        rfft_post_pair(K, x[K], x[(60 - K) % 60], xk, xp);
        store(K, xk);
        if (K != 30) store(60 - K, xp);
    }
}

constexpr float kOrtho120 = 0.09128709291752769f;   // 1 / sqrt(120)

// R2: half spectrum (ortho) of normalised sinograms.  grid = images, 128 lanes (120 columns).
// out16 (optional): the same values rounded to nearest-even fp16 = the exchange / replica format (29 280 B)
__global__ __launch_bounds__(kSlotThreads) void k_ring_half_spectrum(const float* __restrict__ x, float2* __restrict__ out,
                                                                     __half2* __restrict__ out16)
{
    const int d = min((int)threadIdx.x, kD - 1);
    const float* src = x + (size_t)blockIdx.x * kA * kD + d;
    v2f s[60];
#pragma unroll
    for (int m = 0; m < 60; ++m) s[m] = (v2f){src[(2 * m) * kD], src[(2 * m + 1) * kD]};
    const bool live = threadIdx.x < kD;
    const size_t o = (size_t)blockIdx.x * kHalf * kD + d;
    rfft120(s, [&](int k, v2f twice) {
        if (!live) return;
        const v2f v = twice * (v2f){0.5f * kOrtho120, 0.5f * kOrtho120};
        if (out) out[o + k * kD] = make_float2(v.x, v.y);
        if (out16) out16[o + k * kD] = __floats2half2_rn(v.x, v.y);
    });
}

// Sum over the 64 lanes of a wave of |s[n]|, n = 0..119, s[2m] = x[m].x, s[2m+1] = x[m].y (reduce-scatter: 6 halving
// steps).  On return lane L holds the totals of n = 2L (w0) and 2L+1 (w1).  Lanes that must not contribute hold zeros.
// Steps 0 and 1 (96 of the 126 exchanges) are v_permlane32_swap / v_permlane16_swap: the two values a lane would select
// between are swapped across the wave halves / row pairs in place, so a halving step is one swap + one add.
__device__ __forceinline__ float swap32_abs_add(float a, float b)
{
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    return fabsf(__uint_as_float(r[0])) + fabsf(__uint_as_float(r[1]));   // low half: |a[L]| + |a[L+32]|, high half: |b[L-32]| + |b[L]|
}
__device__ __forceinline__ float swap16_add(float a, float b)
{
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);                  // even rows: a + a[L^16], odd rows: b[L^16] + b
}

__device__ __forceinline__ void wave_abs_reduce_scatter(const v2f (&x)[60], float& w0, float& w1)
{
    const int lane = threadIdx.x & 63;
    float w[64];
    // step 0: indices [0,64) stay in the low half-wave, [64,128) in the high one (120..127 do not exist: zero)
#pragma unroll
    for (int i = 0; i < 64; ++i) {
        const int j = 64 + i;
        const float lo_v = (i & 1) ? x[i >> 1].y : x[i >> 1].x;
        const float hi_v = j < 120 ? ((j & 1) ? x[j >> 1].y : x[j >> 1].x) : 0.0f;
        w[i] = swap32_abs_add(lo_v, hi_v);
    }
#pragma unroll
    for (int i = 0; i < 32; ++i) w[i] = swap16_add(w[i], w[32 + i]);
#pragma unroll
    for (int step = 2; step < 6; ++step) {
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
    const long long* db_first;   // optional (sweeps): query q sweeps the ndb entries that start at entry db_first[q] of DB (NULL: entry 0)
    const long long* q_row;      // optional (sweeps): query q is entry q_row[q] of Q (NULL: entry q)
    float2* mc_partial;          // k_ring_sweep_dma<MC>: per-lane |corr| sums of every (query, channel, candidate, half): [nq][C][ndb][128]
    int units;                   // k_ring_sweep_dma with nq > 1: (channel, slice) units of the launch (see there); 0 otherwise
};

__device__ __forceinline__ float2 load_spec(const float2* p) { return *p; }
__device__ __forceinline__ float2 load_spec(const __half2* p) { return __half22float2(*p); }

// grid = (blocks, nq); NSLOT pair streams of 128 lanes.  QLDS: the (single-channel) query spectrum is staged in
// LDS and shared by the slots; otherwise (pairwise mode, C > 1) it is read through L2 like the candidate.
// DBT = float2 (the database format) or __half2 (fp16 replicas received from other ranks).
// Descriptors with C channels are [C][61][120]: |corr| is summed over channels and detectors
// (fast_corr_RINGplusplus, RING_ros/util.py:337-358; C = 1: fast_corr, util.py:362-374).
// PAIRWISE: candidate = query index (one per block row); MULTI: runtime channel count (else C = 1, no loop).
template <int NSLOT, bool QLDS, bool PAIRWISE, bool MULTI, typename DBT, bool PRELOAD = false>
__global__ __launch_bounds__(NSLOT* kSlotThreads) void k_ring_corr_fft(const float2* __restrict__ Q, const DBT* __restrict__ DB,
                                                                       FftCorrP p, float* __restrict__ dist,
                                                                       int* __restrict__ angle, float* __restrict__ corr_out)
{
    extern __shared__ __attribute__((aligned(16))) v2f qs[];  // [61][128] query half spectrum, columns 120..127 zero (QLDS)
    __shared__ float xbuf[NSLOT][128];
    const int q = blockIdx.y;
    const int slot = threadIdx.x / kSlotThreads;
    const int t = threadIdx.x % kSlotThreads;
    const int wave_in_slot = t >> 6, lane = t & 63;
    const int d = min(t, kD - 1);
    const bool live_col = t < kD;
    const int C = MULTI ? p.channels : 1;
    const size_t entry = (size_t)C * kHalf * kD;
    const float2* qsrc = Q + (size_t)((!PAIRWISE && p.q_row) ? p.q_row[q] : q) * entry;
    if (!PAIRWISE && p.db_first) DB += (size_t)p.db_first[q] * entry;
    if (QLDS) {
        for (int i = threadIdx.x; i < kHalf * kLdsStride; i += NSLOT * kSlotThreads) {
            const int k = i / kLdsStride, col = i % kLdsStride;
            v2f v = {0.0f, 0.0f};
            if (col < kD) { const float2 g = qsrc[k * kD + col]; v = (v2f){g.x, g.y}; }
            qs[i] = v;
        }
        __syncthreads();
    }
    const int ncand = PAIRWISE ? 1 : p.ndb;
    const int stride = gridDim.x * NSLOT;
    const int rounds = (ncand + stride - 1) / stride;
    for (int r = 0; r < rounds; ++r) {
        const int cand = (r * gridDim.x + blockIdx.x) * NSLOT + slot;
        const bool live = cand < ncand;      // uniform over the slot's two waves
        float v[2] = {0.0f, 0.0f};
        if (live) {
            for (int c = 0; c < C; ++c) {
                v2f x[60];
                const DBT* b = DB + (size_t)(PAIRWISE ? q : cand) * entry + (size_t)c * kHalf * kD + d;
                const float2* a = qsrc + (size_t)c * kHalf * kD + d;
                auto la = [&](int k) {
                    if (QLDS) return qs[k * kLdsStride + t];
                    const float2 g = a[k * kD];       // lanes past the last detector: zero query -> zero correlation
                    return live_col ? (v2f){g.x, g.y} : (v2f){0.0f, 0.0f};
                };
                if (PRELOAD) {
                    DBT raw[kHalf];                   // the whole column of the candidate in flight before the first use
#pragma unroll
                    for (int k = 0; k < kHalf; ++k) raw[k] = b[k * kD];
                    __builtin_amdgcn_sched_barrier(0);
                    corr_irfft120(la, [&](int k) { const float2 g = load_spec(&raw[k]); return (v2f){g.x, g.y}; }, x);
                } else {
                    corr_irfft120(la, [&](int k) { const float2 g = load_spec(b + k * kD); return (v2f){g.x, g.y}; }, x);
                }
                float w0, w1;
                wave_abs_reduce_scatter(x, w0, w1);
                v[0] += w0;
                v[1] += w1;
            }
        }
        if (wave_in_slot == 1) { xbuf[slot][2 * lane] = v[0]; xbuf[slot][2 * lane + 1] = v[1]; }
        __syncthreads();
        if (wave_in_slot == 0 && live) {
            const float sc = kOrtho120;  // ifft ortho factor
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

// The single-channel database sweep with the candidate of the NEXT round already in flight while this round's
// correlation runs (software pipeline in registers: 61 x 8 B per lane, or 61 x 4 B for fp16 replicas).  Same arithmetic and
// epilogue as k_ring_corr_fft<NSLOT, true, false, false>; one query per blockIdx.y.
template <int NSLOT, typename DBT, int WAVES>
__global__ __launch_bounds__(NSLOT* kSlotThreads) __attribute__((amdgpu_waves_per_eu(WAVES, WAVES))) void k_ring_sweep_pipe(
    const float2* __restrict__ Q, const DBT* __restrict__ DB, FftCorrP p, float* __restrict__ dist, int* __restrict__ angle,
    float* __restrict__ corr_out)
{
    extern __shared__ __attribute__((aligned(16))) v2f qs[];  // [61][128] query half spectrum, columns 120..127 zero
    __shared__ float xbuf[NSLOT][128];
    const int q = blockIdx.y;
    const int slot = threadIdx.x / kSlotThreads;
    const int t = threadIdx.x % kSlotThreads;
    const int wave_in_slot = t >> 6, lane = t & 63;
    const int d = min(t, kD - 1);
    const size_t entry = (size_t)kHalf * kD;
    const float2* qsrc = Q + (size_t)q * entry;
    const int ncand = p.ndb;
    const int stride = gridDim.x * NSLOT;
    const int rounds = (ncand + stride - 1) / stride;
    int cand = blockIdx.x * NSLOT + slot;
    DBT nxt[kHalf];
    auto fetch = [&](int c) {
        const DBT* b = DB + (size_t)c * entry + d;
#pragma unroll
        for (int k = 0; k < kHalf; ++k) nxt[k] = b[k * kD];
    };
    if (cand < ncand) fetch(cand);           // in flight while the query is staged
    for (int i = threadIdx.x; i < kHalf * kLdsStride; i += NSLOT * kSlotThreads) {
        const int k = i / kLdsStride, col = i % kLdsStride;
        v2f v = {0.0f, 0.0f};
        if (col < kD) { const float2 g = qsrc[k * kD + col]; v = (v2f){g.x, g.y}; }
        qs[i] = v;
    }
    __syncthreads();
    for (int r = 0; r < rounds; ++r) {
        const bool live = cand < ncand;      // uniform over the slot's two waves
        DBT cur[kHalf];
#pragma unroll
        for (int k = 0; k < kHalf; ++k) cur[k] = nxt[k];
        const int next = cand + stride;
        if (next < ncand) fetch(next);
        __builtin_amdgcn_sched_barrier(0);
        float v0 = 0.0f, v1 = 0.0f;
        if (live) {
            v2f x[60];
            corr_irfft120([&](int k) { return qs[k * kLdsStride + t]; },
                          [&](int k) { const float2 g = load_spec(&cur[k]); return (v2f){g.x, g.y}; }, x);
            wave_abs_reduce_scatter(x, v0, v1);
        }
        if (wave_in_slot == 1) { xbuf[slot][2 * lane] = v0; xbuf[slot][2 * lane + 1] = v1; }
        __syncthreads();
        if (wave_in_slot == 0 && live) {
            const float sc = kOrtho120;
            const float s0 = (v0 + xbuf[slot][2 * lane]) * sc, s1 = (v1 + xbuf[slot][2 * lane + 1]) * sc;
            const int n0 = 2 * lane, n1 = 2 * lane + 1;
            const int m0 = n0 < 60 ? n0 + 60 : n0 - 60, m1 = n1 < 60 ? n1 + 60 : n1 - 60;
            const size_t o = (size_t)q * p.ndb + cand;
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
        cand = next;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// One-query database sweep with the candidate staged by LDS-DMA (gfx950 `global_load_lds_dwordx4`) instead of in registers.
//
// ONE workgroup per compute unit; the query's half spectrum lies in LDS once ([61][120] complex64, compact rows + a
// 64-byte zero tail) and every WAVE is its own pipeline: it owns a ring of kRing 1-KiB LDS slots that the DMA engine fills
// with (row J, row 60 - J) x 64 columns of its candidate -- 16 B per lane, lanes 0-31 row J, lanes 32-63 row 60 - J -- while
// the wave transforms what has landed.  A wave-instruction moves 1 KiB without touching a VGPR, stays in flight across
// everything the wave does (only the issuing wave's vmcnt orders it), and the 122 registers per lane that held the
// pre-requested column in k_ring_corr_fft are gone: the kernel needs the 120 registers of the in-register transform plus a few
// frequency pairs of LDS look-ahead, so three waves per SIMD fit without spills and there is no workgroup barrier after the query
// has been staged.
//
// Work unit = (candidate, column half h): columns h * 64 .. h * 64 + 63 (h = 1: 56 live columns; the dead lanes fetch a
// duplicate of columns 118-119 and are switched off in the first step of the sum over detectors by a per-lane 0 / 1 factor, one
// FMA in place of the ADD).  A unit is a stream of kElems = 32 elements: the 30 frequency pairs, row 30, and one filler, so that
// element e always lives in ring slot e % kRing and every LDS offset is an immediate.  The element kRing ahead is requested as
// soon as an element has been consumed -- across units, so the ring is full while the wave runs the 60-point transform and
// the |.|-sum; exactly one DMA per element (fillers fetch one 16-byte piece) keeps the `s_waitcnt vmcnt(N)` counts static.
// SPLIT = false: a wave runs both halves of its candidate in turn and adds them in registers.  SPLIT = true: waves 2 i and
// 2 i + 1 take the two halves of the same candidate (half the quantisation loss when a wave has only a few candidates); the
// second to finish reads the partner's per-lane sums from LDS and writes the result -- no barrier, two LDS counters per pair.
// Arithmetic per column, the transform, the sum over lanes and the order (half 0 + half 1) are those of k_ring_corr_fft:
// the outputs are bit-identical.
constexpr int kRing = 8;
constexpr int kElems = 32;
constexpr int kSlotBytes = 1024;
// DMA-tiled database entry (mrs_ring_spec_to_tiled): the same 7 320 complex values as the [61][120] half spectrum, permuted so that the 2 x 64 (h = 0)
// or 2 x 56 (h = 1) values of stream element e of half h are ONE contiguous, 128-byte aligned block:
//   h = 0: bytes [1024 e, 1024 e + 512) = row e, columns 0-63;   [1024 e + 512, 1024 (e + 1)) = row 60 - e, columns 0-63   (e < 30; e = 30: row 30 only)
//   h = 1: bytes 31 232 + [896 e, 896 e + 448) = row e, columns 64-119;  the next 448 = row 60 - e, columns 64-119        (e < 30; e = 30: row 30 only)
// 58 560 bytes + 64 of padding: entries are 58 624 bytes apart, every DMA reads whole 128-byte lines that nobody else needs.
constexpr int kTiledHalf1 = 30 * 1024 + 512;            // byte offset of the h = 1 blocks
constexpr int kTiledEntryBytes = 58624;
constexpr int kTiledEntry = kTiledEntryBytes / 8;       // in complex values


// POL: cache policy of the DMA load -- 0 default, 1 nt (streaming: the entry is read once), 2 sc1, 3 sc0 sc1, 4 sc1 nt, 5 sc0 sc1 nt
#define MRS_GLDS(MOD) asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" MOD "\n\ts_mov_b32 m0, %0" \
                                   : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory")
template <int POL>
__device__ __forceinline__ void glds16(const void* sbase, unsigned voff, unsigned lds_dst)
{
    unsigned keep;   // M0 = LDS destination of the wave-instruction (lane l lands at M0 + 16 l); written in the statement that reads it
    if (POL == 1) MRS_GLDS(" nt");
    else if (POL == 2) MRS_GLDS(" sc1");
    else if (POL == 3) MRS_GLDS(" sc0 sc1");
    else if (POL == 4) MRS_GLDS(" sc1 nt");
    else if (POL == 5) MRS_GLDS(" sc0 sc1 nt");
    else MRS_GLDS("");
}
#undef MRS_GLDS

template <int N>
__device__ __forceinline__ void wait_vmcnt()
{
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// wave_abs_reduce_scatter with the second operand of step 0 scaled by `m` (1: live, 0: this lane's partner / this lane is
// past the last detector): fma(|b|, 1, |a|) rounds exactly like |a| + |b|
__device__ __forceinline__ void wave_abs_reduce_scatter_masked(const v2f (&x)[60], float m, float& w0, float& w1)
{
    const int lane = threadIdx.x & 63;
    float w[64];
#pragma unroll
    for (int i = 0; i < 64; ++i) {
        const int j = 64 + i;
        const float lo_v = (i & 1) ? x[i >> 1].y : x[i >> 1].x;
        const float hi_v = j < 120 ? ((j & 1) ? x[j >> 1].y : x[j >> 1].x) : 0.0f;
        const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(lo_v), __float_as_uint(hi_v), false, false);
        w[i] = __builtin_fmaf(fabsf(__uint_as_float(r[1])), m, fabsf(__uint_as_float(r[0])));
    }
#pragma unroll
    for (int i = 0; i < 32; ++i) w[i] = swap16_add(w[i], w[32 + i]);
#pragma unroll
    for (int step = 2; step < 6; ++step) {
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

// fftshift + first maximum in shifted order + dist / angle (util.py:367-374); s0, s1 = this lane's samples n = 2 lane, 2 lane + 1
__device__ __forceinline__ void sweep_epilogue(float s0, float s1, int lane, float denom, float* dist_o, int* angle_o)
{
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
        *dist_o = 1.0f - best / denom;
        *angle_o = kA / 2 - bm;
    }
}

struct DmaUnit {            // one (candidate, half) as the DMA sees it; wave-uniform
    const float2* base;     // first value of the candidate (SGPR pair)
    int half;               // 0 / 1
    bool live;              // false: past the end of this wave's work (fillers only)
};

template <int WAVES, bool SPLIT, int NT, int RING = kRing, bool QDMA = true, bool TILED = false, bool PRIO = false, bool MC = false, int PF = 2>   // (!SPLIT: candidates are handed out by a per-workgroup LDS ticket)
__global__ __launch_bounds__(WAVES * 64) void k_ring_sweep_dma(const float2* __restrict__ Q, const float2* __restrict__ DB, FftCorrP p,
                                                               float* __restrict__ dist, int* __restrict__ angle)
{
    // LDS: ring [WAVES][RING][1 KiB] | query [61 * 120 + 8] v2f | SPLIT: partial [WAVES / 2][2 halves][64] v2f, counters
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    v2f* const qs = reinterpret_cast<v2f*>(smem + WAVES * RING * kSlotBytes);
    constexpr int kQueryVals = kHalf * kD + 8;
    v2f* const partial = qs + kQueryVals;                                        // SPLIT only
    unsigned* const counters = reinterpret_cast<unsigned*>(partial + (WAVES / 2) * 2 * 64);   // [WAVES / 2][arrived, done]
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
    const unsigned ring_lds = lds0 + wave * (RING * kSlotBytes);               // wave-uniform LDS byte address of this wave's ring
    const v2f* const ring = reinterpret_cast<const v2f*>(smem + wave * (RING * kSlotBytes)) + lane;

    const size_t plane = TILED ? (size_t)kTiledEntry : (size_t)kHalf * kD;      // float2 values of one channel of an entry
    const size_t entry = (MC ? (size_t)p.channels : (size_t)1) * plane;          // ... from one candidate to the next
    const size_t qentry = (size_t)kHalf * kD;                                   // the query is always in row layout
    const int ncand = p.ndb;
    // this wave's candidates: c0, c0 + cstride, ...
    // MC (RING++, C channels): a workgroup keeps ONE channel for its lifetime (its query plane is staged once, no barrier ever after) and
    // shares the candidates with the other workgroups of that channel; the per-lane sums go to p.mc_partial and k_ring_mc_finish adds
    // the channels in order, so the bits are those of the channel-outer kernel (k_ring_sweep_mc).
    // Several queries per launch (round 6, p.units > 0): workgroup i -> XCD x = i % 8 (where the hardware puts it: MI355X_MICROARCH.md, "block b runs
    // on XCD b % 8" -- a speed assumption only), k = i / 8; query k % nq, unit (k / nq) * 8 + x.  The nq workgroups that sweep the SAME unit's
    // candidates for different queries therefore sit on ONE XCD and, doing the same work per candidate, move through the database in step: an
    // entry is fetched from HBM once and served to the other nq - 1 workgroups by that XCD's L2 (the DMA loads of this form use the default
    // cache policy, not `nt`).  Every (query, candidate) is scored by the one-query pipeline unchanged: same bits.
    const bool multi = p.units > 0;
    const int bx = multi ? (int)((blockIdx.x >> 3) / p.nq) * 8 + (int)(blockIdx.x & 7) : (int)blockIdx.x;     // (channel, slice) unit
    const int nbx = multi ? p.units : (int)gridDim.x;
    const int qy = multi ? (int)((blockIdx.x >> 3) % p.nq) : (int)blockIdx.y;                                    // query
    if (multi && bx >= nbx) return;                      // grid padding (whole workgroup, before any barrier)
    const int C = MC ? p.channels : 1;
    const int ch = MC ? bx % C : 0;
    const int slice = MC ? bx / C : bx;
    const int nslices = MC ? (nbx - ch + C - 1) / C : nbx;
    const int c0 = SPLIT ? slice * (WAVES / 2) + (wave >> 1) : slice * WAVES + wave;
    const int cstride = nslices * (SPLIT ? WAVES / 2 : WAVES);
    const int my_half = SPLIT ? (wave & 1) : 0;
    // !SPLIT: the workgroup's candidates are the sequence t = 0, 1, 2 ... -> slice * WAVES + t % WAVES + (t / WAVES) * cstride; wave w starts with
    // t = w and draws every further one from a ticket in LDS (a wave that finishes early takes what is left: no wave idles through a last round
    // that only some of them have a candidate for).  Which wave scores a candidate does not touch its result.
    unsigned* const ticket = reinterpret_cast<unsigned*>(qs + kQueryVals);       // first word behind the query (the SPLIT form's partial sums live there)
    auto cand_of = [&](unsigned t) { return slice * WAVES + (int)(t % WAVES) + (int)(t / WAVES) * cstride; };

    // per-lane pieces of a DMA: lanes 0-31 row J, lanes 32-63 row 60 - J; 16 B = 2 columns per lane.
    // Row layout ([61][120]): the two rows are 960 (60 - 2 J) bytes apart and a half-row starts on a 64-byte boundary.
    // Tiled layout (mrs_ring_spec_to_tiled): the 1024 (h = 0) / 896 (h = 1) bytes of an element are contiguous and 128-byte aligned.
    const int piece = lane & 31;
    const bool upper = lane >= 32;
    const unsigned voff_h0 = TILED ? (unsigned)lane * 16u : (upper ? 60u * 960u : 0u) + (unsigned)piece * 16u;
    const unsigned voff_h1 = TILED ? (unsigned)kTiledHalf1 + (upper ? 448u : 0u) + (unsigned)min(piece, 27) * 16u
                                   : (upper ? 60u * 960u : 0u) + 512u + (unsigned)min(piece, 27) * 16u;   // pieces 28-31 would leave the row
    const int vdelta = upper ? -960 : 960;
    // The byte offset of the next DMA is a RUNNING register (+- 960 per element; row 30 needs no case of its own: both half-waves
    // arrive at it), re-read through an empty asm at every unit: written as base + e * delta it is loop-invariant where a wave keeps
    // its half, and the 31 offsets of a unit become 31 registers spilled to scratch -- whose reloads share vmcnt with the DMAs.
    unsigned run = 0;
    auto start_unit = [&](const DmaUnit& u) {
        run = u.half ? voff_h1 : voff_h0;
        asm volatile("" : "+v"(run));
    };
    auto issue = [&](const DmaUnit& u, int e) {          // e: element index 0..31 (compile-time after unrolling), strictly in stream order
        if (e == 0) start_unit(u);
        const unsigned voff = (!u.live || e == kElems - 1) ? 0u : run;             // filler: one 16-byte piece for all lanes
        glds16<NT>(u.base, voff, ring_lds + (e % RING) * kSlotBytes);
        if (TILED) run += u.half ? 896u : 1024u;
        else run += (unsigned)vdelta;
    };
    auto unit_of = [&](int c, int h) {
        DmaUnit u;
        u.live = c < ncand;
        u.base = DB + (size_t)(u.live ? c : 0) * entry + (size_t)ch * plane;
        u.half = h;
        return u;
    };

    const float2* const qsrc = Q + ((size_t)qy * C + (size_t)ch) * qentry;           // queries are [nq][C][61][120], row layout
    if (QDMA) {
        // the query rides the same engine: 58 pieces of 1 KiB dealt to the waves (the last one 192 B: lanes 0-11), then the unit's
        // first RING elements behind them -- the barrier below needs only the former (vmcnt(RING)), the latter stay in flight across it
        constexpr int kPieces = (kHalf * kD * 8 + kSlotBytes - 1) / kSlotBytes;    // 58
        const unsigned q_lds = lds0 + WAVES * (RING * kSlotBytes);
        if (threadIdx.x < 8) qs[kHalf * kD + threadIdx.x] = (v2f){0.0f, 0.0f};      // zero tail (read by the dead lanes of row 60)
        for (int pc = wave; pc < kPieces; pc += WAVES)
            if (pc * kSlotBytes + lane * 16 < kHalf * kD * 8) glds16<0>(qsrc, (unsigned)(pc * kSlotBytes + lane * 16), q_lds + pc * kSlotBytes);
    }
    DmaUnit cur = unit_of(c0, my_half);
#pragma unroll
    for (int e = 0; e < RING; ++e) issue(cur, e);        // in flight while the query is staged
    if (QDMA) {
        wait_vmcnt<RING>();
        if (SPLIT && threadIdx.x < (WAVES / 2) * 2) counters[threadIdx.x] = 0;
        if (!SPLIT && threadIdx.x == 0) *ticket = WAVES;
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    } else {   // query -> LDS through registers (compact rows), zero tail; counters
        for (int i = threadIdx.x; i < kQueryVals; i += WAVES * 64) {
            v2f v = {0.0f, 0.0f};
            if (i < kHalf * kD) { const float2 g = qsrc[i]; v = (v2f){g.x, g.y}; }
            qs[i] = v;
        }
        if (SPLIT && threadIdx.x < (WAVES / 2) * 2) counters[threadIdx.x] = 0;
        if (!SPLIT && threadIdx.x == 0) *ticket = WAVES;
        __syncthreads();
    }

    // PF = LDS look-ahead in elements
    int c = c0;
    int round = 0;
    float accA0 = 0.0f, accA1 = 0.0f;                    // !SPLIT: sums of half 0 while half 1 runs
    while (cur.live) {
        DmaUnit nxt;
        int c_next = c + cstride;
        if (SPLIT) nxt = unit_of(c_next, my_half);
        else if (cur.half == 0) nxt = unit_of(c, 1);
        else {                                           // the candidate after this one: drawn now, its first elements are requested at the end of this unit
            unsigned t = 0;
            if (lane == 0) t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            c_next = cand_of(__builtin_amdgcn_readfirstlane(t));
            nxt = unit_of(c_next, 0);
        }
        const int h = cur.half;
        const v2f* const qcol = qs + h * 64 + lane;
        const float m = (h == 1 && piece >= 24) ? 0.0f : 1.0f;

        v2f x[60];
        v2f qa[PF + 1][2], cb[PF + 1][2];                // element e lives in [e % (PF + 1)]
        auto read_elem = [&](int e) {
            const int J = e;
            qa[e % (PF + 1)][0] = qcol[J * kD];
            cb[e % (PF + 1)][0] = ring[(e % RING) * 128];
            if (e < 30) {
                qa[e % (PF + 1)][1] = qcol[(60 - J) * kD];
                cb[e % (PF + 1)][1] = ring[(e % RING) * 128 + 64];
            }
        };
        if (PRIO) __builtin_amdgcn_s_setprio(1);      // the phase that hands ring slots back to the DMA engine goes first
        // elements 0 .. PF - 1 of this unit: their DMAs are followed by RING - 1 - e younger ones
        wait_vmcnt<RING - 1>(); read_elem(0);
        if (PF > 1) { wait_vmcnt<RING - 2>(); read_elem(1); }
#pragma unroll
        for (int e = 0; e < kElems; ++e) {
            if (e + PF <= 30) {
                wait_vmcnt<RING - 1 - PF>();            // element e + PF has landed: RING - 1 - PF younger DMAs may be in flight
                read_elem(e + PF);
            }
            if (e < 30) {
                v2f zj, zp;
                irfft_pre_pair(e, cmul_conj(qa[e % (PF + 1)][0], cb[e % (PF + 1)][0]), cmul_conj(qa[e % (PF + 1)][1], cb[e % (PF + 1)][1]), zj, zp);
                x[e] = zj;
                if (e != 0) x[60 - e] = zp;
            } else if (e == 30) {
                const v2f pm = cmul_conj(qa[e % (PF + 1)][0], cb[e % (PF + 1)][0]);
                v2f unused;
                irfft_pre_pair(30, pm, pm, x[30], unused);
            }
            __builtin_amdgcn_sched_barrier(0);           // the slot's values are in registers before the slot is handed back
            if (e + RING < kElems) issue(cur, e + RING);
            else issue(nxt, e + RING - kElems);
        }
        if (PRIO) __builtin_amdgcn_s_setprio(0);
        cfft60_inv(x);
        float w0, w1;
        wave_abs_reduce_scatter_masked(x, m, w0, w1);

        if (MC) {
            static_assert(!MC || !SPLIT, "the multi-channel form keeps both halves of a candidate on one wave");
            p.mc_partial[(((size_t)qy * C + ch) * p.ndb + c) * 128 + h * 64 + lane] = make_float2(w0, w1);
            if (h == 1) c = c_next;
        } else if (!SPLIT) {
            if (h == 0) { accA0 = w0; accA1 = w1; }
            else {
                const float s0 = (accA0 + w0) * kOrtho120, s1 = (accA1 + w1) * kOrtho120;
                const size_t o = (size_t)qy * p.ndb + c;
                sweep_epilogue(s0, s1, lane, p.denom, dist + o, angle + o);
                c = c_next;
            }
        } else {
            const int pi = wave >> 1;
            v2f* const mine = partial + (pi * 2 + h) * 64 + lane;
            const v2f* const theirs = partial + (pi * 2 + (1 - h)) * 64 + lane;
            unsigned* const arrived = counters + pi * 2;
            unsigned* const done = arrived + 1;
            // one slot per pair: the sums of the previous round have been read by whoever finished it second
            while (__hip_atomic_load(done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < (unsigned)round) __builtin_amdgcn_s_sleep(2);
            *mine = (v2f){w0, w1};
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            unsigned old = 0;
            if (lane == 0) old = __hip_atomic_fetch_add(arrived, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP);
            old = __builtin_amdgcn_readfirstlane(old);
            if (old & 1u) {                               // the partner's sums are there: half 0 + half 1, as the two-wave slot adds them
                const v2f t = *theirs;
                const float a0 = h == 0 ? w0 : t.x, a1 = h == 0 ? w1 : t.y;
                const float b0 = h == 0 ? t.x : w0, b1 = h == 0 ? t.y : w1;
                const float s0 = (a0 + b0) * kOrtho120, s1 = (a1 + b1) * kOrtho120;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                if (lane == 0) __hip_atomic_fetch_add(done, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                const size_t o = (size_t)qy * p.ndb + c;
                sweep_epilogue(s0, s1, lane, p.denom, dist + o, angle + o);
            }
            c += cstride;
            ++round;
        }
        cur = nxt;
    }
    wait_vmcnt<0>();                                     // no DMA may land in LDS that the next workgroup owns
}

// RING++: channel sums of k_ring_sweep_dma<MC> in channel order (((0 + w_0) + w_1) + ...), the two column halves, fftshift, maximum.
// One wave per candidate.
__global__ __launch_bounds__(256) void k_ring_mc_finish(const float2* __restrict__ partial, int ndb, int C, float denom, float* __restrict__ dist,
                                                        int* __restrict__ angle)
{
    const int cand = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (cand >= ndb) return;
    partial += (size_t)blockIdx.y * C * ndb * 128;       // query blockIdx.y
    float2 lo = make_float2(0.0f, 0.0f), hi = make_float2(0.0f, 0.0f);
    for (int c = 0; c < C; ++c) {
        const float2 a = partial[((size_t)c * ndb + cand) * 128 + lane], b = partial[((size_t)c * ndb + cand) * 128 + 64 + lane];
        lo.x += a.x; lo.y += a.y; hi.x += b.x; hi.y += b.y;
    }
    const size_t o = (size_t)blockIdx.y * ndb + cand;
    sweep_epilogue((lo.x + hi.x) * kOrtho120, (lo.y + hi.y) * kOrtho120, lane, denom, dist + o, angle + o);
}

// [n][61][120] half spectra (row layout) -> the DMA-tiled entries above; grid = entries, 256 lanes
__global__ __launch_bounds__(256) void k_ring_spec_to_tiled(const float2* __restrict__ src, float2* __restrict__ dst)
{
    const float2* s = src + (size_t)blockIdx.x * kHalf * kD;
    float2* d = dst + (size_t)blockIdx.x * kTiledEntry;
    for (int i = threadIdx.x; i < kTiledEntry; i += 256) {
        float2 v = make_float2(0.0f, 0.0f);
        if (i < kHalf * kD) {
            int row, col;
            if (i < kTiledHalf1 / 8) {                   // h = 0: 128 values per element (64 of row e, 64 of row 60 - e)
                const int e = i / 128, r = i % 128;
                row = r < 64 ? e : 60 - e;
                col = r % 64;
            } else {                                     // h = 1: 112 values per element
                const int j = i - kTiledHalf1 / 8;
                const int e = j / 112, r = j % 112;
                row = r < 56 ? e : 60 - e;
                col = 64 + r % 56;
            }
            v = s[row * kD + col];
        }
        d[i] = v;
    }
}

// RING++ database sweep, channel-outer: one channel of the query is staged in LDS at a time (the six channels together,
// 375 KB, do not fit) and swept over all of the workgroup's candidates before the next channel replaces it, so that the inner
// loop is the single-channel kernel's (query from LDS, candidate column requested up front, two waves per SIMD).  The
// per-candidate |corr| sums of every lane wait in LDS between channels: MAXR rounds x NSLOT x 128 lanes x 2 floats.
// grid = (queries, chunks): workgroups that follow each other sweep the SAME candidates for different queries and share them
// through L2.  Channel sums are added in channel order per lane, like the channel-inner loop of k_ring_corr_fft (same bits).
template <int NSLOT, int MAXR, typename DBT>
__global__ __launch_bounds__(NSLOT* kSlotThreads) void k_ring_sweep_mc(const float2* __restrict__ Q, const DBT* __restrict__ DB, FftCorrP p,
                                                                       float* __restrict__ dist, int* __restrict__ angle,
                                                                       float* __restrict__ corr_out)
{
    extern __shared__ __attribute__((aligned(16))) v2f qs[];  // [61][128] one query channel, then acc[MAXR][NSLOT][128]
    v2f* acc = qs + kHalf * kLdsStride;
    const int q = blockIdx.x, chunk = blockIdx.y, nchunks = gridDim.y;
    const int slot = threadIdx.x / kSlotThreads;
    const int t = threadIdx.x % kSlotThreads;
    const int lane = t & 63;
    const int d = min(t, kD - 1);
    const int C = p.channels;
    const size_t plane = (size_t)kHalf * kD, entry = (size_t)C * plane;
    const float2* qsrc = Q + (size_t)q * entry;
    const int ncand = p.ndb;
    const int rounds = (ncand + nchunks * NSLOT - 1) / (nchunks * NSLOT);   // <= MAXR (launcher)
    for (int r = 0; r < rounds; ++r) acc[(r * NSLOT + slot) * kSlotThreads + t] = (v2f){0.0f, 0.0f};
    for (int c = 0; c < C; ++c) {
        __syncthreads();   // the previous channel has been read by every wave
        for (int i = threadIdx.x; i < kHalf * kLdsStride; i += NSLOT * kSlotThreads) {
            const int k = i / kLdsStride, col = i % kLdsStride;
            v2f v = {0.0f, 0.0f};
            if (col < kD) { const float2 g = qsrc[c * plane + k * kD + col]; v = (v2f){g.x, g.y}; }
            qs[i] = v;
        }
        __syncthreads();
        for (int r = 0; r < rounds; ++r) {
            const int cand = (r * nchunks + chunk) * NSLOT + slot;
            if (cand >= ncand) continue;   // uniform over the slot's two waves; no barrier inside this loop
            const DBT* b = DB + (size_t)cand * entry + c * plane + d;
            DBT raw[kHalf];
#pragma unroll
            for (int k = 0; k < kHalf; ++k) raw[k] = b[k * kD];
            __builtin_amdgcn_sched_barrier(0);
            v2f x[60];
            corr_irfft120([&](int k) { return qs[k * kLdsStride + t]; },
                          [&](int k) { const float2 g = load_spec(&raw[k]); return (v2f){g.x, g.y}; }, x);
            float w0, w1;
            wave_abs_reduce_scatter(x, w0, w1);
            v2f* a = &acc[(r * NSLOT + slot) * kSlotThreads + t];
            const v2f old = *a;
            *a = (v2f){old.x + w0, old.y + w1};
        }
    }
    __syncthreads();
    if (t >= 64) return;   // the first wave of each slot adds the second wave's sums and picks the maximum
    for (int r = 0; r < rounds; ++r) {
        const int cand = (r * nchunks + chunk) * NSLOT + slot;
        if (cand >= ncand) continue;
        const v2f lo = acc[(r * NSLOT + slot) * kSlotThreads + lane], hi = acc[(r * NSLOT + slot) * kSlotThreads + 64 + lane];
        const float sc = kOrtho120;
        const float s0 = (lo.x + hi.x) * sc, s1 = (lo.y + hi.y) * sc;
        const int n0 = 2 * lane, n1 = 2 * lane + 1;
        const int m0 = n0 < 60 ? n0 + 60 : n0 - 60, m1 = n1 < 60 ? n1 + 60 : n1 - 60;
        const size_t o = (size_t)q * p.ndb + cand;
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
}

// New-descriptor path of a loop check in ONE launch (row R2 + C1 for pairs): half spectrum of the freshly
// normalised sinogram (written out: it becomes a database entry / travels to the other ranks) and, straight from
// the LDS copy of that spectrum, its correlation with the candidate's.  Same arithmetic, op for op, as
// k_ring_half_spectrum followed by the pairwise k_ring_corr_fft (bitwise identical results), one launch and one
// round trip of the spectrum through HBM less.  grid = pairs, 128 lanes.
// CT = float2 (exact database entries) or __half2 (fp16 replicas received from other ranks); cand_idx (optional):
// the candidate of pair i is row cand_idx[i] of `cand` (a pre-selected row of the replicated database) instead of row i.
template <typename CT>
__global__ __launch_bounds__(kSlotThreads) __attribute__((amdgpu_waves_per_eu(1, 1))) void k_ring_spec_corr_pairs(const float* __restrict__ x, const CT* __restrict__ cand,
                                                                       const int* __restrict__ cand_idx, int n_db,
                                                                       float2* __restrict__ out, __half2* __restrict__ out16,
                                                                       float denom, float* __restrict__ dist,
                                                                       int* __restrict__ angle)
{
    extern __shared__ __attribute__((aligned(16))) v2f qs[];  // [61][128], columns 120..127 zero
    __shared__ float xbuf[128];
    const int pair = blockIdx.x;
    const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
    const int d = min(t, kD - 1);
    const bool live_col = t < kD;
    // the candidate's column is requested first: it arrives while the forward transform of the new sinogram runs (the
    // workgroup's LDS footprint allows one wave per SIMD anyway, so the 61 - 122 extra registers cost no occupancy)
    const int row = cand_idx ? cand_idx[pair] : pair;
    const bool no_row = cand_idx && (row < 0 || row >= n_db);
    CT raw[kHalf];
    {
        const CT* b = cand + (size_t)(no_row ? 0 : row) * kHalf * kD + d;
#pragma unroll
        for (int k = 0; k < kHalf; ++k) raw[k] = b[k * kD];
    }
    {
        const float* src = x + (size_t)pair * kA * kD + d;
        v2f s[60];
#pragma unroll
        for (int m = 0; m < 60; ++m) s[m] = (v2f){src[(2 * m) * kD], src[(2 * m + 1) * kD]};
        const size_t o = (size_t)pair * kHalf * kD + d;
        rfft120(s, [&](int k, v2f twice) {
            const v2f v = twice * (v2f){0.5f * kOrtho120, 0.5f * kOrtho120};
            qs[k * kLdsStride + t] = live_col ? v : (v2f){0.0f, 0.0f};
            if (!live_col) return;
            if (out) out[o + k * kD] = make_float2(v.x, v.y);
            if (out16) out16[o + k * kD] = __floats2half2_rn(v.x, v.y);
        });
    }
    __syncthreads();
    if (no_row) {   // no such database row: the new spectrum is still written, the score says "no match"
        if (t == 0) { dist[pair] = INFINITY; angle[pair] = 0; }
        return;
    }
    v2f c[60];
    corr_irfft120([&](int k) { return qs[k * kLdsStride + t]; },
                  [&](int k) { const float2 g = load_spec(&raw[k]); return (v2f){g.x, g.y}; }, c);
    float v0, v1;
    wave_abs_reduce_scatter(c, v0, v1);
    if (wave == 1) { xbuf[2 * lane] = v0; xbuf[2 * lane + 1] = v1; }
    __syncthreads();
    if (wave == 0) {
        const float sc = kOrtho120;
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
    const size_t lds = (size_t)kHalf * kLdsStride * sizeof(v2f);
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
constexpr int kSweepDmaDefault = 11008;
constexpr int kSweepDmaTiledDefault = 11008;
template <int WAVES, bool SPLIT, int NT, bool TILED, bool PRIO, bool MC = false>
static hipError_t sweep_dma_launch_t(int num_cu, hipStream_t s, const float2* q, const float2* db, const FftCorrP& p, float* dist, int* angle)
{
    const size_t lds = (size_t)WAVES * kRing * kSlotBytes + (size_t)(kHalf * kD + 8) * sizeof(v2f) +
                       (SPLIT ? (size_t)(WAVES / 2) * 2 * 64 * sizeof(v2f) + (WAVES / 2) * 2 * sizeof(unsigned) : 16 /* the ticket */);
    auto kern = k_ring_sweep_dma<WAVES, SPLIT, NT, kRing, true, TILED, PRIO, MC>;
    static std::atomic<unsigned> attr_set{0};          // per instantiation, one bit per device: the attribute call costs microseconds of a 100-us query
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (!(attr_set.load(std::memory_order_relaxed) & (1u << (dev & 31)))) {
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr_set.fetch_or(1u << (dev & 31), std::memory_order_relaxed);
    }
    const int per_wg = SPLIT ? WAVES / 2 : WAVES;
    const int C = MC ? p.channels : 1;
    const int need = C * ((p.ndb + per_wg - 1) / per_wg);             // units that have work
    if (p.nq > 1) {
        // nq workgroups per unit, all of them on the unit's XCD (see the kernel): groups of 8 units x nq queries
        FftCorrP pm = p;
        pm.units = std::max(C, std::min(std::max(8, num_cu / (8 * p.nq) * 8), need));
        const int grid = (pm.units + 7) / 8 * 8 * p.nq;
        hipLaunchKernelGGL(kern, dim3(grid, 1), dim3(WAVES * 64), lds, s, q, db, pm, dist, angle);
        return hipGetLastError();
    }
    int blocks = std::max(1, std::min(num_cu, need));
    if (MC) blocks = std::max(p.channels, blocks);                    // every channel needs a workgroup
    hipLaunchKernelGGL(kern, dim3(blocks, 1), dim3(WAVES * 64), lds, s, q, db, p, dist, angle);
    return hipGetLastError();
}
// variant = waves per workgroup (8 / 12) + 100 (the halves of a candidate on two waves) + 1000 (non-temporal DMA loads: the entry is read
// once) + 10000 (s_setprio 1 while a wave consumes / re-requests ring slots)
static hipError_t sweep_dma_launch(int variant, int num_cu, hipStream_t s, const float2* q, const float2* db, const FftCorrP& p, float* dist, int* angle, bool tiled = false)
{
    const bool prio = (variant / 10000) % 10 != 0, split = (variant / 100) % 10 != 0, nt = (variant / 1000) % 10 != 0;
    const int waves = variant % 100;
#define MRS_DMA_1(W, S, N, T, P) if (tiled == T && waves == W && split == S && nt == (N != 0) && prio == P) return sweep_dma_launch_t<W, S, N, T, P>(num_cu, s, q, db, p, dist, angle);
#define MRS_DMA_4(N, T, P) MRS_DMA_1(8, false, N, T, P) MRS_DMA_1(8, true, N, T, P) MRS_DMA_1(12, false, N, T, P) MRS_DMA_1(12, true, N, T, P)
    MRS_DMA_4(0, false, false) MRS_DMA_4(1, false, false) MRS_DMA_4(0, false, true) MRS_DMA_4(1, false, true)
    MRS_DMA_4(0, true, false) MRS_DMA_4(1, true, false) MRS_DMA_4(0, true, true) MRS_DMA_4(1, true, true)
#undef MRS_DMA_4
#undef MRS_DMA_1
    return hipErrorInvalidValue;
}

// Several queries per sweep on the LDS-DMA pipeline: at most this many per launch (the queries of a launch share the compute units: 256 / (8 nq)
// groups of 8 units each)
constexpr int kSweepDmaMaxQ = 32;
constexpr int kSweepDmaMaxQMc = 8;       // RING++: the partial sums of a launch are nq x C x ndb x 1 KiB of scratch
constexpr int kSweepDmaMultiDefault = 10012;   // default cache policy (the other queries' workgroups find the entry in L2), priority, 12 waves

// RING++ (C channels), p.nq queries (<= kSweepDmaMaxQMc): channel-per-workgroup DMA sweep into per-lane partial sums + the finishing kernel
static int sweep_dma_mc(mrs_ctx* ctx, hipStream_t s, const float2* q, const float2* db, FftCorrP p, float* dist, int* angle, bool tiled)
{
    mrs::Scratch part;
    int st = part.alloc((size_t)p.nq * p.channels * p.ndb * 128 * sizeof(float2), s);
    if (st != MRS_OK) return st;
    p.mc_partial = part.as<float2>();
    const int num_cu = ctx->num_cu > 0 ? ctx->num_cu : 256;
    if (p.nq > 1) {
        const char* v = mrs::dev_env("MRS_SWEEP_MQ_MC_WAVES");       // 12 waves (three per SIMD): 23.3-23.5 M pairs/s at 4 queries against 22.7 with 8
        if (!v || atoi(v) == 12) {
            if (tiled) MRS_HIP_TRY((sweep_dma_launch_t<12, false, 0, true, true, true>(num_cu, s, q, db, p, dist, angle)));
            else MRS_HIP_TRY((sweep_dma_launch_t<12, false, 0, false, true, true>(num_cu, s, q, db, p, dist, angle)));
        } else {
            if (tiled) MRS_HIP_TRY((sweep_dma_launch_t<8, false, 0, true, true, true>(num_cu, s, q, db, p, dist, angle)));
            else MRS_HIP_TRY((sweep_dma_launch_t<8, false, 0, false, true, true>(num_cu, s, q, db, p, dist, angle)));
        }
    } else {
        if (tiled) MRS_HIP_TRY((sweep_dma_launch_t<8, false, 1, true, true, true>(num_cu, s, q, db, p, dist, angle)));
        else MRS_HIP_TRY((sweep_dma_launch_t<8, false, 1, false, true, true>(num_cu, s, q, db, p, dist, angle)));
    }
    hipLaunchKernelGGL(k_ring_mc_finish, dim3((p.ndb + 3) / 4, p.nq), dim3(256), 0, s, p.mc_partial, p.ndb, p.channels, p.denom, dist, angle);
    MRS_HIP_TRY(hipGetLastError());
    return MRS_OK;
}

// 1 .. any number of queries over exact fp32 entries (row layout or DMA-tiled) through the LDS-DMA pipeline, in launches of at most
// kSweepDmaMaxQ (single channel) / kSweepDmaMaxQMc (RING++) queries; dist / angle are [n_q][ndb]
static int sweep_dma_queries(mrs_ctx* ctx, hipStream_t s, const float2* q, int n_q, const float2* db, FftCorrP p, float* dist, int* angle, bool tiled)
{
    const int num_cu = ctx->num_cu > 0 ? ctx->num_cu : 256;
    const int C = p.channels;
    const int maxq = C > 1 ? kSweepDmaMaxQMc : kSweepDmaMaxQ;
    const size_t qentry = (size_t)C * kHalf * kD;
    for (int q0 = 0; q0 < n_q; q0 += maxq) {
        p.nq = std::min(maxq, n_q - q0);
        const float2* qq = q + (size_t)q0 * qentry;
        float* dd = dist + (size_t)q0 * p.ndb;
        int* aa = angle + (size_t)q0 * p.ndb;
        if (C > 1) {
            const int st = sweep_dma_mc(ctx, s, qq, db, p, dd, aa, tiled);
            if (st != MRS_OK) return st;
            continue;
        }
        int variant = p.nq > 1 ? kSweepDmaMultiDefault : (tiled ? kSweepDmaTiledDefault : kSweepDmaDefault);
        if (const char* v = mrs::dev_env(p.nq > 1 ? "MRS_SWEEP_MQ_VARIANT" : "MRS_SWEEP_VARIANT")) variant = atoi(v) > 0 ? atoi(v) : variant;
        MRS_HIP_TRY(sweep_dma_launch(variant, num_cu, s, qq, db, p, dd, aa, tiled));
    }
    return MRS_OK;
}

template <typename DBT>
static int corr_fft_launch(mrs_ctx* ctx, const float* d_q, int32_t n_q, const void* d_db, int32_t n_db, int32_t channels,
                           float* d_dist, int32_t* d_angle, float* d_corr, mrs_stream stream, bool pairwise,
                           const long long* d_db_first = nullptr, const long long* d_q_row = nullptr)
{
    MRS_REQUIRE(ctx && d_q && d_db && d_dist && d_angle, "null pointer");
    MRS_REQUIRE(n_q > 0 && n_db > 0 && channels > 0, "counts must be positive");
    MRS_HIP_TRY(hipSetDevice(ctx->device));
    FftCorrP p;
    p.nq = n_q; p.ndb = n_db; p.pairwise = pairwise ? 1 : 0; p.channels = channels;
    p.denom = (float)(0.15 * channels * kA * kD);
    p.db_first = d_db_first; p.q_row = d_q_row; p.mc_partial = nullptr; p.units = 0;
    MRS_REQUIRE(!(d_db_first || d_q_row) || (!pairwise && channels == 1 && std::is_same<DBT, float2>::value && n_q <= mrs::kMaxGridY),
                "per-query database blocks: single-channel fp32 sweeps of at most 65535 queries");
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
                const size_t lds = (size_t)kHalf * kLdsStride * sizeof(v2f);
                // the candidate's whole column is requested before the arithmetic starts (PRELOAD); fp16 replicas (half the
                // registers per value) swept by one or two queries (the HBM-side regime): the NEXT round's candidate is already
                // in flight during this round's arithmetic
                bool launched = false;
                if constexpr (std::is_same<DBT, float2>::value) {
                    // one query, exact entries (the node's loop: main_RING.py:133): the LDS-DMA pipeline, one workgroup per compute unit
                    // ... and, since round 6, a few queries at once (one robot's scan against the other robots' lists: BASELINE configs[3]):
                    // the same pipeline, the queries' workgroups grouped per XCD so that the entry is fetched from HBM once
                    const char* v1 = mrs::dev_env("MRS_SWEEP_VARIANT");
                    const char* vq = mrs::dev_env("MRS_SWEEP_MQ_VARIANT");
                    const bool off = nq == 1 ? (v1 && atoi(v1) == 0) : (vq && atoi(vq) == 0);
                    if (nq <= kSweepDmaMaxQ && !off && !d_corr && !d_db_first && !d_q_row) {
                        const int st = sweep_dma_queries(ctx, s, qq, nq, dd, p, dist_c, angle_c, false);
                        if (st != MRS_OK) return st;
                        launched = true;
                    }
                }
                if constexpr (std::is_same<DBT, __half2>::value) {
                    if (n_q <= 2) {
                        auto kern = k_ring_sweep_pipe<NSLOT, DBT, 2>;
                        MRS_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                        hipLaunchKernelGGL(kern, dim3(blocks, nq), dim3(NSLOT * kSlotThreads), lds, s, qq, dd, p, dist_c, angle_c, corr_c);
                        launched = true;
                    }
                }
                if (!launched) {
                    auto kern = k_ring_corr_fft<NSLOT, true, false, false, DBT, true>;
                    MRS_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                    hipLaunchKernelGGL(kern, dim3(blocks, nq), dim3(NSLOT * kSlotThreads), lds, s, qq, dd, p, dist_c, angle_c, corr_c);
                }
            } else if (std::is_same<DBT, float2>::value && nq <= kSweepDmaMaxQMc && !d_corr &&
                       !(mrs::dev_env(nq == 1 ? "MRS_SWEEP_VARIANT" : "MRS_SWEEP_MQ_VARIANT") && atoi(mrs::dev_env(nq == 1 ? "MRS_SWEEP_VARIANT" : "MRS_SWEEP_MQ_VARIANT")) == 0)) {
                // one query (the node's loop, main_RINGplusplus.py:131-134) or a few: LDS-DMA pipeline, one channel per workgroup
                const int st = sweep_dma_queries(ctx, s, qq, nq, reinterpret_cast<const float2*>(dd), p, dist_c, angle_c, false);
                if (st != MRS_OK) return st;
            } else {
                constexpr int MAXR = 8;
                // chunks = a whole number of "waves" of resident workgroups (two per CU, shared by the queries of the launch), the
                // smallest that keeps the rounds per workgroup within MAXR: 10 000 entries, one query -> 1024 chunks x 5 rounds
                // (625 chunks x 8 rounds would run 512 + 113 workgroups: the second wave almost empty)
                const int resident = std::max(1, 2 * (ctx->num_cu > 0 ? ctx->num_cu : 256) / nq);
                const int waves = (n_db + NSLOT * MAXR * resident - 1) / (NSLOT * MAXR * resident);
                const int chunks = std::max(1, std::min(waves * resident, (n_db + NSLOT - 1) / NSLOT));
                const size_t lds = (size_t)(kHalf * kLdsStride + MAXR * NSLOT * kSlotThreads) * sizeof(v2f);
                if (chunks <= mrs::kMaxGridY) {
                    auto kern = k_ring_sweep_mc<NSLOT, MAXR, DBT>;
                    MRS_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                    hipLaunchKernelGGL(kern, dim3(nq, chunks), dim3(NSLOT * kSlotThreads), lds, s, qq, dd, p, dist_c, angle_c, corr_c);
                } else {   // more than 65535 x 16 candidates in one call: the channel-inner kernel has no such limit
                    hipLaunchKernelGGL((k_ring_corr_fft<NSLOT, false, false, true, DBT>), dim3(blocks, nq), dim3(NSLOT * kSlotThreads), 0, s, qq, dd,
                                       p, dist_c, angle_c, corr_c);
                }
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

int mrs_ring_spec_to_tiled(mrs_ctx* ctx, const float* d_half_spec, int32_t n, float* d_tiled, mrs_stream stream)
{
    MRS_REQUIRE(ctx && d_half_spec && d_tiled, "null pointer");
    MRS_REQUIRE(n > 0, "n must be positive");
    MRS_HIP_TRY(hipSetDevice(ctx->device));
    hipLaunchKernelGGL(k_ring_spec_to_tiled, dim3(n), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const float2*>(d_half_spec),
                       reinterpret_cast<float2*>(d_tiled));
    MRS_HIP_TRY(hipGetLastError());
    return MRS_OK;
}

int mrs_ring_corr_fft_sweep_tiled_q(mrs_ctx* ctx, const float* d_query_spec, int32_t n_query, const float* d_db_tiled, int32_t n_db, int32_t channels,
                                    float* d_dist, int32_t* d_angle, mrs_stream stream)
{
    MRS_REQUIRE(ctx && d_query_spec && d_db_tiled && d_dist && d_angle, "null pointer");
    MRS_REQUIRE(n_query > 0 && n_db > 0 && channels > 0, "counts must be positive");
    MRS_HIP_TRY(hipSetDevice(ctx->device));
    FftCorrP p;
    p.nq = 1; p.ndb = n_db; p.pairwise = 0; p.channels = channels;
    p.denom = (float)(0.15 * channels * kA * kD);
    p.db_first = nullptr; p.q_row = nullptr; p.mc_partial = nullptr; p.units = 0;
    return sweep_dma_queries(ctx, (hipStream_t)stream, reinterpret_cast<const float2*>(d_query_spec), n_query, reinterpret_cast<const float2*>(d_db_tiled), p,
                             d_dist, d_angle, true);
}

int mrs_ring_corr_fft_sweep_tiled(mrs_ctx* ctx, const float* d_query_spec, const float* d_db_tiled, int32_t n_db, int32_t channels, float* d_dist,
                                  int32_t* d_angle, mrs_stream stream)
{
    return mrs_ring_corr_fft_sweep_tiled_q(ctx, d_query_spec, 1, d_db_tiled, n_db, channels, d_dist, d_angle, stream);
}

int mrs_ring_corr_fft_sweep_blocks(mrs_ctx* ctx, const float* d_spec, const int64_t* d_query_row, int32_t n_query, const int64_t* d_db_first,
                                   int32_t n_db, float* d_dist, int32_t* d_angle, mrs_stream stream)
{
    MRS_REQUIRE(d_query_row && d_db_first, "null pointer");
    static_assert(sizeof(long long) == sizeof(int64_t), "int64_t is long long here");
    return corr_fft_launch<float2>(ctx, d_spec, n_query, d_spec, n_db, 1, d_dist, d_angle, nullptr, stream, false,
                                   reinterpret_cast<const long long*>(d_db_first), reinterpret_cast<const long long*>(d_query_row));
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
