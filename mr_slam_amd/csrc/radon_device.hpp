// radon_device.hpp -- the parts of the Radon kernels shared by radon.hip and fused.hip: plan object, ray table view,
// the two-image sample loop (march2 / trace_ray2) and the fused normalisation.  See radon.hip for the design notes.
#pragma once
#include "common.hpp"

struct mrs_radon_plan {
    mrs_ctx* ctx = nullptr;
    int n_angles = 0, det = 0, H = 0, W = 0;
    float spacing = 1.0f;
    bool in_lds = true;      // the zero-bordered image fits the LDS (else: global-memory path)
    int* d_meta = nullptr;   // ray table, one allocation: meta | base | q | vm | n, each [n_angles*det]
    int* d_degenerate = nullptr;  // sinograms with zero / non-finite std seen by the fused normalisation
    bool two_in_lds = false; // two interleaved images fit the LDS (k_radon2)
    // tuning of the fused rasterise + Radon kernel (fused.hip; mrs_radon_plan_set_option)
    int fused_stagger_us = 70;  // odd workgroups start this many microseconds late (phase-shifts the HBM-bound and the VALU-bound halves)
    int fused_prefetch = 2;     // 16-byte load triplets in flight per lane while rasterising (2 or 4)
    int fused_grid = 0;         // persistent workgroups (0 = one per compute unit)
    int fused_skip = 0;         // measurement aid (MRS_RADON_OPT_FUSED_SKIP): 1 = the slot-table kernel without its rasteriser, 2 = without its ray march
    int fused_variant = 2;      // 2 (default): slot tables (rays sorted by orientation and length) + rolled ray loop, raw sums in registers, sinogram written once;
                                // 1: the same with the raw sums parked in the output buffer (re-read, rewritten); 0: the table in ray order, 15 rays unrolled
    // slot tables of the two-image kernels: lane slot s = k * 1024 + lane carries ray slot_ray[s] (-1: idle).  Rays are sorted by
    // (orientation, step count), so that the 64 lanes of a wave march rays of (almost) the same length: a wave executes the longest
    // of its rays, and in (angle, detector) order 16 % of the lane-steps were idle
    int4* d_slot = nullptr;     // {n_steps | ydom << 16, LDS byte offset of the first texel line (single-image units), q bits, vm bits}
    float* d_slot_nrm = nullptr;
    int* d_slot_ray = nullptr;
    int slot_per_lane = 0;
};

namespace {

constexpr int kRadonWG = 1024; // 16 waves = 4 per SIMD; 120x120 rays = 14 full rounds + one of 64 rays
constexpr int kPad = 2;

// The geometry of a ray does not depend on the image: it is evaluated once, on the host, when the
// plan is built (fp32 op for op like forward.cu:32-112, cos/sin in double), and every workgroup
// streams the table from L2 instead of redoing ~10 IEEE divisions per ray per image.
struct RadonP {
    int A, D, H, W, stride;
    const int* meta;    // n_steps | ydom << 16  (n_steps == 0: the ray misses the image)
    const int* base;    // LDS byte offset of the first sample's dominant-axis texel line
    const float* q;     // minor-axis coordinate of the first sample, shifted by +1.5 (border + centre)
    const float* vm;    // its increment per sample
    const float* nrm;   // length of one step
};

// slot tables of the two-image kernels (mrs_radon_plan::d_slot ...): lane slot s = k * 1024 + lane marches ray ray[s] (-1: idle)
struct SlotP {
    const int4* slot;
    const float* nrm;
    const int* ray;
};

typedef float v2f __attribute__((ext_vector_type(2)));
typedef const __attribute__((address_space(3))) char* lds_cptr;

// ---- two images per workgroup -------------------------------------------------------------------------------
// The geometry of a ray is the same for every image, so a lane that marches ray r through TWO images shares the whole
// index chain (q += vm, fract, cvt, address, 1 - fr) between them: per sample and image 4.5 VALU instead of 7.  The two
// images are interleaved texel by texel in the LDS ((A,B) cells of 8 bytes): one ds_read_b64 fetches a tap of both, and
// the packed FMA (A_t, B_t) * (w, w) advances both images' running sums.  Arithmetic per image is exactly the
// single-image loop's (same operations, same order): results are bit-identical.
// volatile: keeps every tap a ds_read_b64 (2 LDS cycles per wave, 64 banks); the load/store optimiser would otherwise
// fuse pairs into ds_read2_b64, which moves the same bytes at half the rate (MI355X_MICROARCH.md, LDS table)
typedef const volatile __attribute__((address_space(3))) v2f* lds_v2ptr;
__device__ __forceinline__ v2f lds_cell(unsigned addr) { return *(lds_v2ptr)(uintptr_t)addr; }

template <bool YDOM, int STRIDE>
__device__ __forceinline__ void march2(unsigned off, float q, float vm, int n_steps, int rstride, float& outA, float& outB)
{
    const int stride = STRIDE > 0 ? STRIDE : rstride;
    const int unit = (YDOM ? 1 : stride) * 8;   // bytes between the two taps == bytes per minor index
    const int lstep = (YDOM ? stride : 1) * 8;  // bytes per sample along the dominant axis
    constexpr int U = 6;
    v2f acc0 = {0.0f, 0.0f}, acc1 = {0.0f, 0.0f};   // per tap: (image A, image B)
    int j = 0;
#pragma nounroll
    for (; j + U <= n_steps; j += U) {
        v2f t0[U], t1[U];
        float w0[U], w1[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const float fr = __builtin_amdgcn_fractf(q);
            const unsigned a = off + (unsigned)(YDOM ? (int)q * 8 : __mul24((int)q, unit));
            t0[u] = lds_cell(a + u * lstep);
            t1[u] = lds_cell(a + u * lstep + unit);
            w0[u] = 1.0f - fr;
            w1[u] = fr;
            q += vm;
        }
        off += U * lstep;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const v2f W0 = {w0[u], w0[u]}, W1 = {w1[u], w1[u]};
            acc0 = __builtin_elementwise_fma(t0[u], W0, acc0);
            acc1 = __builtin_elementwise_fma(t1[u], W1, acc1);
        }
    }
    for (; j < n_steps; ++j) {
        const float fr = __builtin_amdgcn_fractf(q);
        const unsigned a = off + (unsigned)(YDOM ? (int)q * 8 : __mul24((int)q, unit));
        const v2f t0 = lds_cell(a), t1 = lds_cell(a + unit);
        const float w0 = 1.0f - fr;
        const v2f W0 = {w0, w0}, W1 = {fr, fr};
        acc0 = __builtin_elementwise_fma(t0, W0, acc0);
        acc1 = __builtin_elementwise_fma(t1, W1, acc1);
        q += vm;
        off += lstep;
    }
    outA = acc0.x + acc1.x;
    outB = acc0.y + acc1.y;
}

template <int STRIDE>
__device__ __forceinline__ void trace_ray2(const v2f* cells, const RadonP& p, int ray, float& outA, float& outB)
{
    const int meta = p.meta[ray];
    const int n_steps = meta & 0xffff;
    if (n_steps == 0) { outA = 0.0f; outB = 0.0f; return; }
    const float q = p.q[ray], vm = p.vm[ray], n = p.nrm[ray];
    const unsigned tile = (unsigned)(uintptr_t)(lds_cptr)reinterpret_cast<const char*>(cells) + 2u * (unsigned)p.base[ray];
    float a, b;
    if (meta >> 16) march2<true, STRIDE>(tile, q, vm, n_steps, p.stride, a, b);
    else march2<false, STRIDE>(tile, q, vm, n_steps, p.stride, a, b);
    outA = a * n;
    outB = b * n;
}

__device__ __forceinline__ double wave_sum(double v)
{
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Mean / unbiased std of `rays` values held as val[k] by the workgroup's lanes; returns (mean, sd).  A constant
// sinogram (blank image) has sd == 0: torchvision's fn.normalize raises there (util.py:197); here the normalised
// output is written as zeros and *degenerate is counted up so that the host mirror can raise the same error.
template <int N>
__device__ __forceinline__ void normalize_store(const float (&val)[N], int rays, double (&red)[2][16], float* __restrict__ dst,
                                                int* __restrict__ degenerate, int tid = -1)
{
    // tid: the caller's (possibly opaque) copy of threadIdx.x, so that a persistent kernel's per-round addresses are not hoisted out of its loop
    if (tid < 0) tid = (int)threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    double s1 = 0.0;
#pragma unroll
    for (int k = 0; k < N; ++k)
        if (tid + k * kRadonWG < rays) s1 += (double)val[k];
    s1 = wave_sum(s1);
    __syncthreads();
    if (lane == 0) red[0][wave] = s1;
    __syncthreads();
    double tot = 0.0;
    for (int w = 0; w < kRadonWG / 64; ++w) tot += red[0][w];
    const double mean_d = tot / (double)rays;
    const float mean = (float)mean_d;
    double s2 = 0.0;
#pragma unroll
    for (int k = 0; k < N; ++k)
        if (tid + k * kRadonWG < rays) {
            const double dlt = (double)val[k] - mean_d;
            s2 += dlt * dlt;
        }
    s2 = wave_sum(s2);
    if (lane == 0) red[1][wave] = s2;
    __syncthreads();
    double tot2 = 0.0;
    for (int w = 0; w < kRadonWG / 64; ++w) tot2 += red[1][w];
    const float sd = (float)sqrt(tot2 / (double)(rays - 1));
    const bool ok = sd > 0.0f && sd < INFINITY;
    if (!ok && tid == 0 && degenerate) atomicAdd(degenerate, 1);
#pragma unroll
    for (int k = 0; k < N; ++k) {
        const int ray = tid + k * kRadonWG;
        if (ray < rays) dst[ray] = ok ? (val[k] - mean) / sd : 0.0f;
    }
}

}  // namespace
