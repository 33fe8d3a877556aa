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
// 58.56 KB and ~1 400 packed / scalar VALU instructions per 120-lane column pass: with one query the sweep runs at the
// device's copy rate (81 M pairs/s = 4.8 TB/s, profiles/r02_*); with several queries per launch the database is fetched once
// (L2 shares it between the query rows) and the in-register FFT (VALU) binds at 115-135 M pairs/s.  RING++ descriptors
// ([C][61][120]) are swept channel-outer (k_ring_sweep_mc); fp16 replicas of the database are accepted as well.
#include <algorithm>
#include <cmath>
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
    p.db_first = d_db_first; p.q_row = d_q_row;
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
