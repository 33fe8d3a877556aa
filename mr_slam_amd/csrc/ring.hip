// ring.hip -- RING / RING++ descriptor finishing and rotation correlation for gfx950
// (SURVEY.md 8(a) rows R2, C1, C2, C3).
//
// Reference behaviour reproduced (never copied), all in LoopDetection/src/RING_ros/util.py:
//   generate_RING:198 (FFT over the angle axis), forward_row_fft:295-300,
//   fast_corr:362-374, fast_corr_RINGplusplus:337-358, solve_translation:388-423.
//
// fast_corr computes, for TIRING spectra a = F(x), b = F(y) (ortho FFT over the angle axis of
// the normalised sinograms x, y):  corr[n,d] = ifft(a * conj(b))[n,d] = A^-1/2 * sum_j
// x[(j+n) mod A, d] * y[j, d]  -- a circular cross-correlation per detector column --
// followed by  s[n] = sum_c sum_d |corr[c,n,d]|,  fftshift, max / argmax.
// The database sweep (k_ring_corr) evaluates exactly that sum in the sinogram domain: the
// database stores the NORMALISED REAL sinograms (A*D*4 bytes per channel, half of the complex
// spectrum the reference stores), each candidate is streamed from HBM once into LDS and
// correlated against an LDS-resident query with register-tiled sliding windows (2 LDS reads
// per 15 FMA).  HBM bytes per pair = C*A*D*4.
// The literal spectral form (any complex input, k_corr_spectra) backs the drop-in
// fast_corr(a, b) signature.
#include <algorithm>
#include <cmath>

#include "common.hpp"

namespace {

constexpr int kWG = 1024;
constexpr int kNB = 15;  // shifts per lane (register tile)

__device__ __forceinline__ float wave_sum_f(float v)
{
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

struct CorrP {
    int C, A, D;
    int nq, ndb;
    int pairwise;      // 1: query q is matched against database entry q only
    float inv_sqrt_a;  // 1/sqrt(A)
    float denom;       // 0.15 * C * A * D
};

// grid = (blocks, nq).  LDS: q tile [A][D], y tile [A][D], part [A][4], s [A].
__global__ __launch_bounds__(kWG) void k_ring_corr(const float* __restrict__ Q, const float* __restrict__ Y,
                                                   CorrP p, float* __restrict__ dist, int* __restrict__ angle,
                                                   float* __restrict__ corr_out)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int AD = p.A * p.D;
    float* qt = smem;
    float* yt = qt + AD;
    float* part = yt + AD;      // [A][4]
    float* s = part + 4 * p.A;  // [A]
    const int q = blockIdx.y;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int nsb = p.A / kNB;
    const float* qsrc = Q + (size_t)q * p.C * AD;
    if (p.pairwise) Y += (size_t)q * p.C * AD;

    for (int cand = blockIdx.x; cand < p.ndb; cand += gridDim.x) {
        __syncthreads();  // wave 0 has finished reading s[] of the previous candidate
        const float* ysrc = Y + (size_t)cand * p.C * AD;
        for (int i = threadIdx.x; i < p.A; i += kWG) s[i] = 0.0f;
        for (int c = 0; c < p.C; ++c) {
            __syncthreads();  // previous channel / candidate done with the tiles
            if (c > 0 || cand == (int)blockIdx.x || p.C > 1) {
                const float4* s4 = reinterpret_cast<const float4*>(qsrc + (size_t)c * AD);
                float4* d4 = reinterpret_cast<float4*>(qt);
                for (int i = threadIdx.x; i < AD / 4; i += kWG) d4[i] = s4[i];
            }
            {
                const float4* s4 = reinterpret_cast<const float4*>(ysrc + (size_t)c * AD);
                float4* d4 = reinterpret_cast<float4*>(yt);
                for (int i = threadIdx.x; i < AD / 4; i += kWG) d4[i] = s4[i];
            }
            for (int i = threadIdx.x; i < 4 * p.A; i += kWG) part[i] = 0.0f;
            __syncthreads();
            // tasks: (shift block sb, 64-column chunk dc); wave w starts at (w>>1, w&1)
            for (int sb = wave >> 1; sb < nsb; sb += 8) {
                const int n0 = sb * kNB;
                for (int dc = wave & 1; dc * 64 < p.D; dc += 2) {
                    const int d = dc * 64 + lane;
                    const bool live = d < p.D;
                    const int dd = live ? d : 0;
                    float acc[kNB], xw[kNB];
#pragma unroll
                    for (int t = 0; t < kNB; ++t) acc[t] = 0.0f;
                    int row = n0;  // row of the window's first element, (j + n0) mod A
#pragma unroll
                    for (int t = 0; t < kNB - 1; ++t) {
                        int r = row + t; if (r >= p.A) r -= p.A;
                        xw[t] = qt[r * p.D + dd];
                    }
                    int rnew = n0 + kNB - 1; if (rnew >= p.A) rnew -= p.A;
                    for (int j0 = 0; j0 < p.A; j0 += kNB) {
#pragma unroll
                        for (int u = 0; u < kNB; ++u) {
                            xw[(u + kNB - 1) % kNB] = qt[rnew * p.D + dd];
                            ++rnew; if (rnew >= p.A) rnew -= p.A;
                            const float y = yt[(j0 + u) * p.D + dd];
#pragma unroll
                            for (int t = 0; t < kNB; ++t)
                                acc[t] = __builtin_fmaf(xw[(u + t) % kNB], y, acc[t]);
                        }
                    }
#pragma unroll
                    for (int t = 0; t < kNB; ++t) {
                        const float v = wave_sum_f(live ? fabsf(acc[t]) : 0.0f);
                        if (lane == 0) part[(n0 + t) * 4 + (dc & 3)] += v;  // one writer per (n, dc&3, pass)
                    }
                }
            }
            __syncthreads();
            for (int i = threadIdx.x; i < p.A; i += kWG)
                s[i] += (part[4 * i] + part[4 * i + 1]) + (part[4 * i + 2] + part[4 * i + 3]);
        }
        __syncthreads();
        // fftshift + first-max argmax by wave 0 (util.py:367-371)
        if (wave == 0) {
            float best = -1.0f;
            int bm = 0;
            const int half = p.A / 2;  // fftshift of an even-length vector: shifted[m] = s[(m + A/2) % A]
            for (int m = lane; m < p.A; m += 64) {
                int n = m + (p.A - half); if (n >= p.A) n -= p.A;
                const float v = s[n] * p.inv_sqrt_a;
                if (corr_out) corr_out[((size_t)q * p.ndb + cand) * p.A + m] = v;
                if (v > best) { best = v; bm = m; }
            }
            for (int o = 32; o > 0; o >>= 1) {
                const float ob = __shfl_xor(best, o, 64);
                const int om = __shfl_xor(bm, o, 64);
                if (ob > best || (ob == best && om < bm)) { best = ob; bm = om; }
            }
            if (lane == 0) {
                dist[(size_t)q * p.ndb + cand] = 1.0f - best / p.denom;
                angle[(size_t)q * p.ndb + cand] = half - bm;
            }
        }
    }
}

// ---- small DFTs along one axis of [n_img][A][D] tiles held in LDS (N <= 256) ---------------
// twiddle table: tw[k] = (cos(2 pi k / N), sin(2 pi k / N)), computed on the host in double.

// R2 (util.py:198): X[k][d] = N^-1/2 * sum_j x[j][d] * exp(-2 pi i j k / N), along the angle axis.
// out is interleaved complex64 [img][A][D].  One workgroup per image.
__global__ __launch_bounds__(kWG) void k_dft_angle_r2c(const float* __restrict__ x, int A, int D,
                                                       const float2* __restrict__ tw_g, float2* __restrict__ out)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* tile = smem;                                   // [A][D]
    float2* tw = reinterpret_cast<float2*>(tile + A * D);  // [A]
    const float* src = x + (size_t)blockIdx.x * A * D;
    for (int i = threadIdx.x; i < A * D; i += kWG) tile[i] = src[i];
    for (int i = threadIdx.x; i < A; i += kWG) tw[i] = tw_g[i];
    __syncthreads();
    const float sc = 1.0f / sqrtf((float)A);
    float2* dst = out + (size_t)blockIdx.x * A * D;
    const int half = A / 2;
    for (int t = threadIdx.x; t < (half + 1) * D; t += kWG) {
        const int k = t / D, d = t - k * D;
        float re = 0.0f, im = 0.0f;
        int idx = 0;
        for (int j = 0; j < A; ++j) {
            const float v = tile[j * D + d];
            const float2 w = tw[idx];
            re = __builtin_fmaf(v, w.x, re);
            im = __builtin_fmaf(-v, w.y, im);
            idx += k; if (idx >= A) idx -= A;
        }
        re *= sc; im *= sc;
        dst[k * D + d] = make_float2(re, im);
        if (k > 0 && k < A - k) dst[(A - k) * D + d] = make_float2(re, -im);  // Hermitian mirror
    }
}

// forward_row_fft (util.py:295-300): |ortho FFT along the detector axis|.  One WG per image.
__global__ __launch_bounds__(kWG) void k_dft_row_mag(const float* __restrict__ x, int A, int D,
                                                     const float2* __restrict__ tw_g, float* __restrict__ out)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* tile = smem;                                       // [A][D+1] padded
    float2* tw = reinterpret_cast<float2*>(tile + A * (D + 1));
    const float* src = x + (size_t)blockIdx.x * A * D;
    for (int i = threadIdx.x; i < A * D; i += kWG) {
        const int r = i / D, c = i - r * D;
        tile[r * (D + 1) + c] = src[i];
    }
    for (int i = threadIdx.x; i < D; i += kWG) tw[i] = tw_g[i];
    __syncthreads();
    const float sc = 1.0f / sqrtf((float)D);
    float* dst = out + (size_t)blockIdx.x * A * D;
    for (int t = threadIdx.x; t < A * D; t += kWG) {
        const int r = t / D, k = t - r * D;
        float re = 0.0f, im = 0.0f;
        int idx = 0;
        for (int j = 0; j < D; ++j) {
            const float v = tile[r * (D + 1) + j];
            const float2 w = tw[idx];
            re = __builtin_fmaf(v, w.x, re);
            im = __builtin_fmaf(-v, w.y, im);
            idx += k; if (idx >= D) idx -= D;
        }
        re *= sc; im *= sc;
        dst[t] = sqrtf(re * re + im * im);
    }
}

// Literal fast_corr (util.py:362-374) on arbitrary complex spectra a, b [C][A][D]:
// corr = ifft_angle(a * conj(b), ortho); |corr|; sum over C and D; fftshift; max/argmax.
// One workgroup per pair; the product is staged through LDS one channel at a time.
__global__ __launch_bounds__(kWG) void k_corr_spectra(const float2* __restrict__ a, const float2* __restrict__ b,
                                                      CorrP p, const float2* __restrict__ tw_g,
                                                      float* __restrict__ dist, int* __restrict__ angle,
                                                      float* __restrict__ corr_out)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int AD = p.A * p.D;
    float2* prod = reinterpret_cast<float2*>(smem);  // [A][D]
    float2* tw = prod + AD;                          // [A]
    float* s = reinterpret_cast<float*>(tw + p.A);   // [A]
    const int pair = blockIdx.x;
    const float2* pa = a + (size_t)pair * p.C * AD;
    const float2* pb = b + (size_t)pair * p.C * AD;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < p.A; i += kWG) { tw[i] = tw_g[i]; s[i] = 0.0f; }
    for (int c = 0; c < p.C; ++c) {
        __syncthreads();
        for (int i = threadIdx.x; i < AD; i += kWG) {
            const float2 u = pa[(size_t)c * AD + i], v = pb[(size_t)c * AD + i];
            prod[i] = make_float2(u.x * v.x + u.y * v.y, u.y * v.x - u.x * v.y);  // u * conj(v)
        }
        __syncthreads();
        for (int n = wave; n < p.A; n += kWG / 64) {  // s[n] is owned by one wave: no sync needed
            float mag = 0.0f;
            for (int d = lane; d < p.D; d += 64) {
                float re = 0.0f, im = 0.0f;
                int idx = 0;
                for (int k = 0; k < p.A; ++k) {
                    const float2 z = prod[k * p.D + d];
                    const float2 w = tw[idx];  // exp(+2 pi i k n / A)
                    re += z.x * w.x - z.y * w.y;
                    im += z.x * w.y + z.y * w.x;
                    idx += n; if (idx >= p.A) idx -= p.A;
                }
                re *= p.inv_sqrt_a; im *= p.inv_sqrt_a;
                mag += sqrtf(re * re + im * im);
            }
            mag = wave_sum_f(mag);
            if (lane == 0) s[n] += mag;
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const int half = p.A / 2;
        float best = -1.0f;
        int bm = 0;
        for (int m = 0; m < p.A; ++m) {
            int n = m + (p.A - half); if (n >= p.A) n -= p.A;
            const float v = s[n];
            if (corr_out) corr_out[(size_t)pair * p.A + m] = v;
            if (v > best) { best = v; bm = m; }
        }
        dist[pair] = 1.0f - best / p.denom;
        angle[pair] = half - bm;
    }
}

// C3 (util.py:388-423): per sinogram row i, circular cross-correlation along the detector axis
// between query[:, i, :] and positive[:, i, :], |.|, fftshift, sum over channels, first argmax
// -> b[i] = W/2 - argmax; then the 2-unknown least squares  [cos(th_i + psi) sin(th_i + psi)] *
// [x y]^T = b_i  (the reference's SVD pseudo-inverse; solved here through the 2x2 normal
// equations in double) and the residual norm.  One workgroup per pair.
// NOTE the reference sums |corr| over channels AFTER the shift; same thing.
__global__ __launch_bounds__(kWG) void k_solve_translation(const float* __restrict__ qs, const float* __restrict__ ps,
                                                           int C, int H, int W, const float* __restrict__ angles_g,
                                                           const float* __restrict__ rot, float* __restrict__ xy_err,
                                                           float* __restrict__ shifts)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int pair = blockIdx.x;
    float* bsh = smem;  // [H]
    const float* q = qs + (size_t)pair * C * H * W;
    const float* p = ps + (size_t)pair * C * H * W;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const float sc = 1.0f / sqrtf((float)W);
    const int half = W / 2;
    // one wave per row; lane handles shifts m = lane, lane+64 (shifted index), W <= 128
    for (int i = wave; i < H; i += kWG / 64) {
        float best = -1.0f;
        int bm = 0;
        for (int m = lane; m < W; m += 64) {
            int n = m + (W - half); if (n >= W) n -= W;  // unshifted lag
            float tot = 0.0f;
            for (int c = 0; c < C; ++c) {
                const float* qr = q + ((size_t)c * H + i) * W;
                const float* pr = p + ((size_t)c * H + i) * W;
                float acc = 0.0f;
                int jj = n;
                for (int j = 0; j < W; ++j) {
                    acc = __builtin_fmaf(qr[jj], pr[j], acc);
                    ++jj; if (jj >= W) jj -= W;
                }
                tot += fabsf(acc * sc);
            }
            if (tot > best) { best = tot; bm = m; }
        }
        for (int o = 32; o > 0; o >>= 1) {
            const float ob = __shfl_xor(best, o, 64);
            const int om = __shfl_xor(bm, o, 64);
            if (ob > best || (ob == best && om < bm)) { best = ob; bm = om; }
        }
        if (lane == 0) bsh[i] = (float)(half - bm);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double saa = 0, sab = 0, sbb = 0, ra = 0, rb = 0;
        const float psi = rot[pair];
        for (int i = 0; i < H; ++i) {
            const float th = angles_g[i] + psi;
            const double ca = (double)cosf(th), sa = (double)sinf(th);
            saa += ca * ca; sab += ca * sa; sbb += sa * sa;
            ra += ca * bsh[i]; rb += sa * bsh[i];
        }
        const double det = saa * sbb - sab * sab;
        const double x = (sbb * ra - sab * rb) / det;
        const double y = (saa * rb - sab * ra) / det;
        double e2 = 0;
        for (int i = 0; i < H; ++i) {
            const float th = angles_g[i] + psi;
            const double r = (double)cosf(th) * x + (double)sinf(th) * y - bsh[i];
            e2 += r * r;
        }
        xy_err[3 * pair + 0] = (float)x;
        xy_err[3 * pair + 1] = (float)y;
        xy_err[3 * pair + 2] = (float)sqrt(e2);
    }
    if (shifts)
        for (int i = threadIdx.x; i < H; i += kWG) shifts[(size_t)pair * H + i] = bsh[i];
}

int get_twiddles(mrs_ctx* ctx, int N, const float2** out)
{
    std::lock_guard<std::mutex> g(ctx->mu);
    auto it = ctx->twiddles.find(N);
    if (it == ctx->twiddles.end()) {
        std::vector<float> h(2 * (size_t)N);
        for (int k = 0; k < N; ++k) {
            const double a = 2.0 * M_PI * (double)k / (double)N;
            h[2 * k] = (float)cos(a);
            h[2 * k + 1] = (float)sin(a);
        }
        float* d = nullptr;
        MRS_HIP_TRY(hipMalloc(&d, h.size() * sizeof(float)));
        MRS_HIP_TRY(hipMemcpy(d, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice));
        it = ctx->twiddles.emplace(N, d).first;
    }
    *out = reinterpret_cast<const float2*>(it->second);
    return MRS_OK;
}

template <class K>
int allow_lds(K kernel, size_t bytes)
{
    if (bytes > 48 * 1024)
        MRS_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    return MRS_OK;
}

}  // namespace

extern "C" {

static int corr_launch(mrs_ctx* ctx, const float* d_query, int32_t n_query, const float* d_db, int32_t n_db,
                       int32_t channels, int32_t n_angles, int32_t det, float* d_dist, int32_t* d_angle,
                       float* d_corr, mrs_stream stream, int pairwise)
{
    MRS_REQUIRE(ctx && d_query && d_db && d_dist && d_angle, "null pointer");
    MRS_REQUIRE(n_query > 0 && n_db > 0 && channels > 0, "counts must be positive");
    MRS_REQUIRE(n_angles > 0 && det > 0, "sizes must be positive");
    if (n_angles % kNB != 0 || n_angles % 2 != 0 || (n_angles * det) % 4 != 0 || det > 256) {
        mrs::set_error("ring_corr_sweep: needs n_angles %% 30 == 0, n_angles*det %% 4 == 0, det <= 256 (got %d x %d)", n_angles, det);
        return MRS_ERR_UNSUPPORTED;
    }
    const size_t lds = ((size_t)2 * n_angles * det + 5 * (size_t)n_angles) * sizeof(float);
    if (lds > ctx->lds_bytes) {
        mrs::set_error("ring_corr_sweep: %d x %d tiles do not fit LDS", n_angles, det);
        return MRS_ERR_UNSUPPORTED;
    }
    MRS_HIP_TRY(hipSetDevice(ctx->device));
    int st = allow_lds(k_ring_corr, lds);
    if (st != MRS_OK) return st;
    CorrP p;
    p.C = channels; p.A = n_angles; p.D = det; p.nq = n_query; p.ndb = pairwise ? 1 : n_db; p.pairwise = pairwise;
    p.inv_sqrt_a = 1.0f / sqrtf((float)n_angles);
    p.denom = (float)(0.15 * channels * n_angles * det);
    if (pairwise) n_db = 1;
    int blocks = ctx->num_cu > 0 ? ctx->num_cu : 256;
    if (n_query > 1) blocks = (blocks + n_query - 1) / n_query;
    if (blocks > n_db) blocks = n_db;
    if (blocks < 1) blocks = 1;
    const size_t entry = (size_t)channels * n_angles * det;
    for (int q0 = 0; q0 < n_query; q0 += mrs::kMaxGridY) {   // grid.y is limited to 65535 rows
        const int nq = std::min(n_query - q0, mrs::kMaxGridY);
        p.nq = nq;
        const size_t ro = (size_t)q0 * (pairwise ? 1 : n_db);
        hipLaunchKernelGGL(k_ring_corr, dim3(blocks, nq), dim3(kWG), lds, (hipStream_t)stream, d_query + q0 * entry,
                           pairwise ? d_db + q0 * entry : d_db, p, d_dist + ro, d_angle + ro, d_corr ? d_corr + ro * n_angles : nullptr);
    }
    MRS_HIP_TRY(hipGetLastError());
    return MRS_OK;
}

int mrs_ring_corr_sweep(mrs_ctx* ctx, const float* d_query, int32_t n_query, const float* d_db, int32_t n_db,
                        int32_t channels, int32_t n_angles, int32_t det, float* d_dist, int32_t* d_angle,
                        float* d_corr, mrs_stream stream)
{
    return corr_launch(ctx, d_query, n_query, d_db, n_db, channels, n_angles, det, d_dist, d_angle, d_corr, stream, 0);
}

int mrs_ring_corr_pairs(mrs_ctx* ctx, const float* d_a, const float* d_b, int32_t n_pairs, int32_t channels,
                        int32_t n_angles, int32_t det, float* d_dist, int32_t* d_angle, float* d_corr,
                        mrs_stream stream)
{
    return corr_launch(ctx, d_a, n_pairs, d_b, n_pairs, channels, n_angles, det, d_dist, d_angle, d_corr, stream, 1);
}

int mrs_ring_corr_spectra(mrs_ctx* ctx, const float* d_a, const float* d_b, int32_t n_pairs, int32_t channels,
                          int32_t n_angles, int32_t det, float* d_dist, int32_t* d_angle, float* d_corr,
                          mrs_stream stream)
{
    MRS_REQUIRE(ctx && d_a && d_b && d_dist && d_angle, "null pointer");
    MRS_REQUIRE(n_pairs > 0 && channels > 0 && n_angles > 0 && det > 0, "sizes must be positive");
    MRS_REQUIRE(n_angles % 2 == 0, "n_angles must be even");
    const size_t lds = ((size_t)2 * n_angles * det + 3 * (size_t)n_angles + 16) * sizeof(float);
    if (lds > ctx->lds_bytes) {
        mrs::set_error("ring_corr_spectra: %d x %d spectrum does not fit LDS", n_angles, det);
        return MRS_ERR_UNSUPPORTED;
    }
    MRS_HIP_TRY(hipSetDevice(ctx->device));
    const float2* tw;
    int st = get_twiddles(ctx, n_angles, &tw);
    if (st != MRS_OK) return st;
    st = allow_lds(k_corr_spectra, lds);
    if (st != MRS_OK) return st;
    CorrP p;
    p.C = channels; p.A = n_angles; p.D = det; p.nq = 1; p.ndb = n_pairs; p.pairwise = 0;
    p.inv_sqrt_a = 1.0f / sqrtf((float)n_angles);
    p.denom = (float)(0.15 * n_angles * det);   // util.py:369: no channel factor in fast_corr (only fast_corr_RINGplusplus has one)
    hipLaunchKernelGGL(k_corr_spectra, dim3(n_pairs), dim3(kWG), lds, (hipStream_t)stream,
                       reinterpret_cast<const float2*>(d_a), reinterpret_cast<const float2*>(d_b), p, tw, d_dist,
                       d_angle, d_corr);
    MRS_HIP_TRY(hipGetLastError());
    return MRS_OK;
}

int mrs_fft_angle_r2c(mrs_ctx* ctx, const float* d_x, int32_t n_img, int32_t n_angles, int32_t det, float* d_out,
                      mrs_stream stream)
{
    MRS_REQUIRE(ctx && d_x && d_out, "null pointer");
    MRS_REQUIRE(n_img > 0 && n_angles > 0 && det > 0, "sizes must be positive");
    const size_t lds = ((size_t)n_angles * det + 2 * (size_t)n_angles) * sizeof(float);
    if (lds > ctx->lds_bytes) { mrs::set_error("fft_angle_r2c: tile does not fit LDS"); return MRS_ERR_UNSUPPORTED; }
    MRS_HIP_TRY(hipSetDevice(ctx->device));
    const float2* tw;
    int st = get_twiddles(ctx, n_angles, &tw);
    if (st != MRS_OK) return st;
    st = allow_lds(k_dft_angle_r2c, lds);
    if (st != MRS_OK) return st;
    hipLaunchKernelGGL(k_dft_angle_r2c, dim3(n_img), dim3(kWG), lds, (hipStream_t)stream, d_x, n_angles, det, tw,
                       reinterpret_cast<float2*>(d_out));
    MRS_HIP_TRY(hipGetLastError());
    return MRS_OK;
}

int mrs_fft_row_magnitude(mrs_ctx* ctx, const float* d_x, int32_t n_img, int32_t n_angles, int32_t det, float* d_out,
                          mrs_stream stream)
{
    MRS_REQUIRE(ctx && d_x && d_out, "null pointer");
    MRS_REQUIRE(n_img > 0 && n_angles > 0 && det > 0, "sizes must be positive");
    const size_t lds = ((size_t)n_angles * (det + 1) + 2 * (size_t)det) * sizeof(float);
    if (lds > ctx->lds_bytes) { mrs::set_error("fft_row_magnitude: tile does not fit LDS"); return MRS_ERR_UNSUPPORTED; }
    MRS_HIP_TRY(hipSetDevice(ctx->device));
    const float2* tw;
    int st = get_twiddles(ctx, det, &tw);
    if (st != MRS_OK) return st;
    st = allow_lds(k_dft_row_mag, lds);
    if (st != MRS_OK) return st;
    hipLaunchKernelGGL(k_dft_row_mag, dim3(n_img), dim3(kWG), lds, (hipStream_t)stream, d_x, n_angles, det, tw, d_out);
    MRS_HIP_TRY(hipGetLastError());
    return MRS_OK;
}

int mrs_ring_solve_translation(mrs_ctx* ctx, const float* d_query, const float* d_positive, int32_t n_pairs,
                               int32_t channels, int32_t height, int32_t width, const float* d_angles,
                               const float* d_rot, float* d_xy_err, float* d_shifts, mrs_stream stream)
{
    MRS_REQUIRE(ctx && d_query && d_positive && d_angles && d_rot && d_xy_err, "null pointer");
    MRS_REQUIRE(n_pairs > 0 && channels > 0 && height > 1 && width > 0, "sizes must be positive");
    MRS_REQUIRE(width % 2 == 0, "width must be even");
    MRS_HIP_TRY(hipSetDevice(ctx->device));
    hipLaunchKernelGGL(k_solve_translation, dim3(n_pairs), dim3(kWG), (size_t)height * sizeof(float),
                       (hipStream_t)stream, d_query, d_positive, channels, height, width, d_angles, d_rot, d_xy_err,
                       d_shifts);
    MRS_HIP_TRY(hipGetLastError());
    return MRS_OK;
}

}  // extern "C"
