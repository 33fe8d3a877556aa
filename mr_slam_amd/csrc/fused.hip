// fused.hip -- the front half of generate_RING (RING_ros/util.py:174-197) in ONE persistent kernel for gfx950:
// Cartesian max-z BEV (rows A3/A4, generate_bev_cython_binary/src/kernel.cu:14-61 + manager.cu:53-91) rasterised straight
// into the Radon kernel's LDS tile, then the sinogram (row R1, torch-radon/src/forward.cu:12-124) and its normalisation
// (row R2a, util.py:197).  Nothing is copied from the reference; the arithmetic is the one of k_cart_lds (bev.hip) and
// k_radon2 (radon.hip), shared through bev_cart.hpp / radon_device.hpp, so the results are bit-identical to running the
// two kernels back to back.
//
// Why fuse: the rasteriser is HBM-bound (12 B per point, 0.8 of the HBM peak, VALU 43 % busy) and the Radon march is
// VALU-bound (115 KB of HBM traffic per image).  Run back to back each leaves the other resource idle.  Here one workgroup
// per compute unit (the two interleaved images take 124 KB of the 160 KB LDS) loops over pairs of scans: rasterise both
// scans into the (A, B) texel cells, march the rays, take the next pair from a global counter.  Workgroups drift out of phase
// (a workgroup that streams points while its neighbours march rays has the memory system to itself and gets ahead: the
// drift feeds itself; `fused_stagger_us` seeds it by starting every other workgroup half a period late), so that at any time
// some compute units pull points from HBM while the others are busy in the VALU.  The BEV image never goes to HBM unless asked for.
// Measured: 0.39-0.41 ms per 1024 scans from 8192 scans per launch against 0.50-0.54 for the two kernels; a round issues
// about 0.8 of the VALU slots it has, i.e. the fused kernel sits near the VALU roof with 3.8 TB/s of point traffic underneath.
#include "common.hpp"
#include "bev_cart.hpp"
#include "radon_device.hpp"

#include <algorithm>
#include <cstdlib>

namespace {

// Rasterise one scan into image `which` (0 = A, 1 = B) of the interleaved tile: the per-point logic of k_cart_lds (bev.hip),
// with cell (ix, iy) at int index 2 * ((ix + kPad) * stride + iy + kPad) + which.  One workgroup per compute unit means 4 waves
// per SIMD instead of the stand-alone rasteriser's 8, and the fused kernel as a whole is VALU-bound (rasterising + ray march
// issue ~80 % of the VALU slots of a round), so this loop counts instructions:
//   * full stages of PF point quads per lane run without bounds tests, the next stage's loads in flight while the current one is
//     rasterised (two register stages); the ragged end of the scan takes the simple per-point form;
//   * the 4 * PF cells of a stage are read from the LDS together, then compared, then (rarely) raised by ds_max: one LDS round
//     trip per stage instead of one per point.  A read that is stale by the time of its compare only costs a redundant ds_max;
//   * the cell offset is formed in fp32 (fx * stride + fy is an exact small integer: one fma + one conversion instead of two
//     conversions and a quarter-rate integer multiply);
//   * eps_fast (bev_cart.hpp): the distance from a bin edge below which a point leaves the fp32 path is 3.4x the worst-case error
//     of the fma quotient instead of the stand-alone kernel's 17x, so that 1.3 % instead of 6 % of the wave-points drag their
//     wave through the exact (fp64 division) path.  Same bits: tools/ab_fused.py compares billions of points per run.
template <int PF>
__device__ __forceinline__ void rasterise_scan(int* icells, int which, const float* __restrict__ px, const float* __restrict__ py,
                                               const float* __restrict__ pz, int n, const CartP& p, int stride)
{
    // Only z > 0 can change the map (max_h starts at 0: manager.cu:57,69).  Common case in one test: 0 < z < 1 and
    // 0 < |x|,|y| <= 1 and both quotients at least eps away from a bin edge -> the fp32 quotient's floor IS the reference's
    // double floor.  Everything else (rare) takes the exact per-axis path of cart_lin().
    const float eps = p.eps_fast;
    const float inv_x = p.inv_x, inv_y = p.inv_y;
    const float fstride = (float)stride;
    const int NY = p.NY;
    int* const origin = icells + 2 * (kPad * stride + kPad) + which;
    // offset of the point's cell from `origin` and the bits of its z; (0, 0) for a point that cannot raise anything through the
    // fast path (0 never exceeds a cell), after the exact path has dealt with it
    auto prep = [&](float x, float y, float z, int& off, int& zi) {
        const float gx = __builtin_fmaf(x, inv_x, inv_x), gy = __builtin_fmaf(y, inv_y, inv_y);
        const float fx = floorf(gx), fy = floorf(gy);
        const float ex = 0.5f - fabsf((gx - fx) - 0.5f), ey = 0.5f - fabsf((gy - fy) - 0.5f);  // distance to a bin edge
        // every comparison is false for a NaN operand: NaN x or y leave the fast path
        const bool fast = (bool)((int)(z > 0.0f) & (int)(z < 1.0f) & (int)(fabsf(x) <= 1.0f) & (int)(fabsf(y) <= 1.0f) & (int)(x * y != 0.0f) &
                                 (int)(ex >= eps) & (int)(ey >= eps));
        const int cell = (int)__builtin_fmaf(fx, fstride, fy);    // exact for the fast path's 0 <= fx, fy < 2^11; discarded otherwise
        off = fast ? 2 * cell : 0;
        zi = fast ? __float_as_int(z) : 0;
        if (!fast && z > 0.0f) {
            int col;
            const int lin = cart_lin(p, x, y, z, col);
            if (lin >= 0) {
                const int ix = lin / NY, iy = lin - ix * NY;
                atomicMax(origin + 2 * (ix * stride + iy), __float_as_int(z));
            }
        }
    };
    auto put = [&](float x, float y, float z) {
        int off, zi;
        prep(x, y, z, off, zi);
        if (origin[off] < zi) atomicMax(origin + off, zi);
    };
    int done = 0;
    if (aligned16(px) && aligned16(py) && aligned16(pz)) {
        const int n4 = n >> 2;
        const float4* x4 = reinterpret_cast<const float4*>(px);
        const float4* y4 = reinterpret_cast<const float4*>(py);
        const float4* z4 = reinterpret_cast<const float4*>(pz);
        constexpr int kStage = PF * kRadonWG;       // quads per stage of the whole workgroup
        const int stages = n4 / kStage;             // full stages: every lane owns PF valid quads, no bounds tests
        auto fetch = [&](int s, float4 (&A)[PF], float4 (&Bv)[PF], float4 (&Cv)[PF]) {
#pragma unroll
            for (int u = 0; u < PF; ++u) {
                const int k = s * kStage + u * kRadonWG + (int)threadIdx.x;
                A[u] = stream_load4(x4 + k); Bv[u] = stream_load4(y4 + k); Cv[u] = stream_load4(z4 + k);
            }
        };
        auto rasterise = [&](const float4 (&A)[PF], const float4 (&Bv)[PF], const float4 (&Cv)[PF]) {
            int off[4 * PF], zi[4 * PF], cur[4 * PF];
#pragma unroll
            for (int u = 0; u < PF; ++u) {
                prep(A[u].x, Bv[u].x, Cv[u].x, off[4 * u + 0], zi[4 * u + 0]);
                prep(A[u].y, Bv[u].y, Cv[u].y, off[4 * u + 1], zi[4 * u + 1]);
                prep(A[u].z, Bv[u].z, Cv[u].z, off[4 * u + 2], zi[4 * u + 2]);
                prep(A[u].w, Bv[u].w, Cv[u].w, off[4 * u + 3], zi[4 * u + 3]);
            }
#pragma unroll
            for (int k = 0; k < 4 * PF; ++k) cur[k] = origin[off[k]];
#pragma unroll
            for (int k = 0; k < 4 * PF; ++k)
                if (cur[k] < zi[k]) atomicMax(origin + off[k], zi[k]);
        };
        if (stages > 0) {
            float4 X[PF], Y[PF], Z[PF];
            fetch(0, X, Y, Z);
#pragma nounroll
            for (int s = 1; s < stages; ++s) {
                float4 Xn[PF], Yn[PF], Zn[PF];
                fetch(s, Xn, Yn, Zn);                                 // in flight while stage s - 1 is rasterised
                rasterise(X, Y, Z);
#pragma unroll
                for (int u = 0; u < PF; ++u) { X[u] = Xn[u]; Y[u] = Yn[u]; Z[u] = Zn[u]; }
            }
            rasterise(X, Y, Z);
        }
        for (int k = stages * kStage + (int)threadIdx.x; k < n4; k += kRadonWG) {      // the ragged end: fewer quads than lanes x PF
            const float4 A = stream_load4(x4 + k), Bv = stream_load4(y4 + k), Cv = stream_load4(z4 + k);
            put(A.x, Bv.x, Cv.x);
            put(A.y, Bv.y, Cv.y);
            put(A.z, Bv.z, Cv.z);
            put(A.w, Bv.w, Cv.w);
        }
        done = n4 << 2;
    }
    for (int i = done + threadIdx.x; i < n; i += kRadonWG) put(px[i], py[i], pz[i]);
}

// grid = persistent workgroups (one per compute unit); pair 2k, 2k+1 of the batch per round, rounds handed out by *next_pair
// (zeroed by the host before the launch).  bev_out / sino_raw / sino_norm may each be null.
template <int MAX_RAYS_PER_LANE, int STRIDE, int PF>
__global__ __launch_bounds__(kRadonWG) void k_bev_radon2(const float* __restrict__ xyz, const int64_t* __restrict__ offs, CartP cp, RadonP p,
                                                         int batch, float* __restrict__ bev_out, float* __restrict__ sino_raw,
                                                         float* __restrict__ sino_norm, int* __restrict__ degenerate,
                                                         unsigned* __restrict__ next_pair, unsigned stagger_ticks)
{
    extern __shared__ __attribute__((aligned(16))) int lds_i[];   // [rows][stride] cells of (A, B) texels, as ints while rasterising
    __shared__ double red[2][16];
    __shared__ unsigned s_next;
    const v2f* cells = reinterpret_cast<const v2f*>(lds_i);
    const int pairs = (batch + 1) >> 1;
    const int rows = p.H + 2 * kPad;
    const int rays = p.A * p.D;
    const int hw = p.H * p.W;
    if (stagger_ticks != 0u && (blockIdx.x & 1u)) {
        const unsigned long long t0 = wall_clock64();   // constant-rate counter (100 MHz)
        while (wall_clock64() - t0 < (unsigned long long)stagger_ticks) __builtin_amdgcn_s_sleep(64);
    }
    unsigned pair = blockIdx.x;
    while (pair < (unsigned)pairs) {
        // opaque copy of the lane id: the ray-table addresses of 15 rays x 5 arrays per lane are loop-invariant and would otherwise be
        // hoisted out of the persistent loop (150 VGPRs -> scratch spills)
        int tid = (int)threadIdx.x;
        asm volatile("" : "+v"(tid));
        const int b0 = 2 * (int)pair, b1 = b0 + 1;
        const bool two = b1 < batch;
        int2* z2 = reinterpret_cast<int2*>(lds_i);
        for (int i = threadIdx.x; i < rows * p.stride; i += kRadonWG) z2[i] = make_int2(0, 0);
        __syncthreads();
        {
            const int64_t o = offs[b0];
            const int n = (int)(offs[b0 + 1] - o);
            const float* px = xyz + 3 * o;
            rasterise_scan<PF>(lds_i, 0, px, px + n, px + 2 * (size_t)n, n, cp, p.stride);
        }
        if (two) {
            const int64_t o = offs[b1];
            const int n = (int)(offs[b1 + 1] - o);
            const float* px = xyz + 3 * o;
            rasterise_scan<PF>(lds_i, 1, px, px + n, px + 2 * (size_t)n, n, cp, p.stride);
        }
        __syncthreads();
        if (bev_out) {   // the COMPACT layout of mrs_bev_cart_batch: [b][ix][iy]
            for (int i = threadIdx.x; i < hw; i += kRadonWG) {
                const int y = i / p.W, x = i - y * p.W;
                const int2 c = z2[(y + kPad) * p.stride + x + kPad];
                bev_out[(size_t)b0 * hw + i] = __int_as_float(c.x);
                if (two) bev_out[(size_t)b1 * hw + i] = __int_as_float(c.y);
            }
        }
        float va[MAX_RAYS_PER_LANE], vb[MAX_RAYS_PER_LANE];
#pragma unroll
        for (int k = 0; k < MAX_RAYS_PER_LANE; ++k) {
            const int ray = tid + k * kRadonWG;
            float a = 0.0f, b = 0.0f;
            if (ray < rays) {
                trace_ray2<STRIDE>(cells, p, ray, a, b);
                if (sino_raw) {
                    sino_raw[(size_t)b0 * rays + ray] = a;
                    if (two) sino_raw[(size_t)b1 * rays + ray] = b;
                }
            }
            va[k] = a; vb[k] = b;
        }
        if (sino_norm) {
            normalize_store<MAX_RAYS_PER_LANE>(va, rays, red, sino_norm + (size_t)b0 * rays, degenerate);
            if (two) normalize_store<MAX_RAYS_PER_LANE>(vb, rays, red, sino_norm + (size_t)b1 * rays, degenerate);
        }
        __syncthreads();   // every wave is done with the tile (and with red) before the next round clears it
        if (threadIdx.x == 0) s_next = gridDim.x + atomicAdd(next_pair, 1u);
        __syncthreads();
        pair = s_next;
    }
}

// Variant with slot tables (radon_device.hpp: mrs_radon_plan::d_slot): lane slot s = k * 1024 + lane marches ray slot_ray[s].  The rays of a
// wave have (almost) the same length, the ray loop is a real loop (one 16-byte table entry per ray, the next one requested while the current
// ray is marched) and the raw sums go to the output buffer instead of 30 registers: the normalisation re-reads them after a barrier in the
// library-wide lane <-> ray order (threadIdx.x + k * 1024), so every reduction runs in the order of normalize_store / k_normalize and the
// bits do not change.  The registers this frees pay for deeper point prefetch in the rasteriser stage (PF quads x 2 stages per lane).

// ONCE: the raw sums stay in the lane's registers through the march (a 16-float vector indexed by the wave-uniform ray counter), go to the
// then-free tile after the march's barrier and are read back in the library-wide lane <-> ray order for the normalisation: the sinogram is
// written to HBM once instead of parked raw, re-read and rewritten (PMC traffic 1.12 x -> ~1.0 x the algorithmic bytes); same reduction
// order, same bits.
typedef float v16f __attribute__((ext_vector_type(16)));

template <int MAX_RAYS_PER_LANE, int STRIDE, int PF, bool ONCE = false>
__global__ __launch_bounds__(kRadonWG) void k_bev_radon3(const float* __restrict__ xyz, const int64_t* __restrict__ offs, CartP cp, RadonP p, SlotP sp,
                                                         int batch, float* __restrict__ bev_out, float* __restrict__ sino_raw,
                                                         float* __restrict__ sino_norm, float* __restrict__ park, int* __restrict__ degenerate,
                                                         unsigned* __restrict__ next_pair, unsigned stagger_ticks,
                                                         unsigned long long* __restrict__ prof, int dev_skip)
{
    // dev_skip (development aid, MRS_FUSED_SKIP): 1 = no rasterising, 2 = no ray march (results are then meaningless)
    // prof (development aid, MRS_FUSED_PROF=1): 100 MHz ticks per phase summed over workgroups and rounds: clear, rasterise, march, normalise
    unsigned long long tp = 0;
    auto stamp = [&](int phase) {
        if (prof && threadIdx.x == 0) {
            const unsigned long long now = wall_clock64();
            if (phase >= 0) atomicAdd(prof + phase, now - tp);
            tp = now;
        }
    };
    extern __shared__ __attribute__((aligned(16))) int lds_i[];   // [rows][stride] cells of (A, B) texels, as ints while rasterising
    __shared__ double red[2][16];
    __shared__ unsigned s_next;
    const v2f* cells = reinterpret_cast<const v2f*>(lds_i);
    const int pairs = (batch + 1) >> 1;
    const int rows = p.H + 2 * kPad;
    const int rays = p.A * p.D;
    const int hw = p.H * p.W;
    const unsigned tile0 = (unsigned)(uintptr_t)(lds_cptr)reinterpret_cast<const char*>(cells);
    if (stagger_ticks != 0u && (blockIdx.x & 1u)) {
        const unsigned long long t0 = wall_clock64();   // constant-rate counter (100 MHz)
        while (wall_clock64() - t0 < (unsigned long long)stagger_ticks) __builtin_amdgcn_s_sleep(64);
    }
    unsigned pair = blockIdx.x;
    while (pair < (unsigned)pairs) {
        // opaque copy of the lane id: per-round addresses derived from it stay inside the persistent loop instead of being hoisted into
        // (and spilled from) dozens of registers
        int tid = (int)threadIdx.x;
        asm volatile("" : "+v"(tid));
        const int b0 = 2 * (int)pair, b1 = b0 + 1;
        const bool two = b1 < batch;
        int2* z2 = reinterpret_cast<int2*>(lds_i);
        stamp(-1);
        for (int i = threadIdx.x; i < rows * p.stride; i += kRadonWG) z2[i] = make_int2(0, 0);
        __syncthreads();
        stamp(0);
        if (!(dev_skip & 1)) {
            const int64_t o = offs[b0];
            const int n = (int)(offs[b0 + 1] - o);
            const float* px = xyz + 3 * o;
            rasterise_scan<PF>(lds_i, 0, px, px + n, px + 2 * (size_t)n, n, cp, p.stride);
        }
        if (two && !(dev_skip & 1)) {
            const int64_t o = offs[b1];
            const int n = (int)(offs[b1 + 1] - o);
            const float* px = xyz + 3 * o;
            rasterise_scan<PF>(lds_i, 1, px, px + n, px + 2 * (size_t)n, n, cp, p.stride);
        }
        __syncthreads();
        stamp(1);
        if (bev_out) {   // the COMPACT layout of mrs_bev_cart_batch: [b][ix][iy]
            for (int i = threadIdx.x; i < hw; i += kRadonWG) {
                const int y = i / p.W, x = i - y * p.W;
                const int2 c = z2[(y + kPad) * p.stride + x + kPad];
                bev_out[(size_t)b0 * hw + i] = __int_as_float(c.x);
                if (two) bev_out[(size_t)b1 * hw + i] = __int_as_float(c.y);
            }
        }
        // raw sums are parked where the normalisation will read them: the normalised output itself (overwritten in place), else the
        // raw output, else this workgroup's scratch rows
        float* const dA = sino_norm ? sino_norm + (size_t)b0 * rays : sino_raw ? sino_raw + (size_t)b0 * rays : park + (size_t)blockIdx.x * 2 * rays;
        float* const dB = sino_norm ? sino_norm + (size_t)b1 * rays : sino_raw ? sino_raw + (size_t)b1 * rays : dA + rays;
        float* const rA = sino_norm && sino_raw ? sino_raw + (size_t)b0 * rays : nullptr;      // both outputs wanted: raw goes there as well
        int4 e = sp.slot[tid];
        float nrm = sp.nrm[tid];
        int ray = sp.ray[tid];
        const bool keep = ONCE && sino_norm != nullptr;     // raw sums wait in registers, not in the output buffer
        v16f ka = 0.0f, kb = 0.0f;
#pragma nounroll
        for (int k = 0; k < MAX_RAYS_PER_LANE; ++k) {
            const int4 ce = e;
            const float cn = nrm;
            const int cr = ray;
            if (k + 1 < MAX_RAYS_PER_LANE) {        // the next ray's table entry travels while this one is marched
                const int s = tid + (k + 1) * kRadonWG;
                e = sp.slot[s]; nrm = sp.nrm[s]; ray = sp.ray[s];
            }
            const int n_steps = (dev_skip & 2) ? 0 : (ce.x & 0xffff);
            if (cr >= 0) {
                float a = 0.0f, b = 0.0f;
                if (n_steps > 0) {
                    const unsigned tile = tile0 + 2u * (unsigned)ce.y;
                    const float q = __int_as_float(ce.z), vm = __int_as_float(ce.w);
                    if (ce.x >> 16) march2<true, STRIDE>(tile, q, vm, n_steps, p.stride, a, b);
                    else march2<false, STRIDE>(tile, q, vm, n_steps, p.stride, a, b);
                    a *= cn; b *= cn;
                }
                if (keep) { ka[k] = a; kb[k] = b; }
                else {
                    dA[cr] = a;
                    if (two) dB[cr] = b;
                }
                if (rA) {
                    rA[cr] = a;
                    if (two) rA[rays + cr] = b;
                }
            }
        }
        __syncthreads();   // every ray's raw sum is in place (stores of this workgroup are visible to it after the barrier); the tile is free
        float* const tA = reinterpret_cast<float*>(lds_i);      // ONCE: the tile as two rows of `rays` raw sums
        float* const tB = tA + rays;
        if (keep) {
#pragma unroll
            for (int k = 0; k < MAX_RAYS_PER_LANE; ++k) {
                const int cr = sp.ray[tid + k * kRadonWG];
                if (cr >= 0) { tA[cr] = ka[k]; tB[cr] = kb[k]; }
            }
            __syncthreads();
        }
        stamp(2);
        if (sino_norm) {
            float va[MAX_RAYS_PER_LANE], vb[MAX_RAYS_PER_LANE];
#pragma unroll
            for (int k = 0; k < MAX_RAYS_PER_LANE; ++k) {
                const int r = tid + k * kRadonWG;
                va[k] = r < rays ? (keep ? tA[r] : dA[r]) : 0.0f;
                vb[k] = (two && r < rays) ? (keep ? tB[r] : dB[r]) : 0.0f;
            }
            normalize_store<MAX_RAYS_PER_LANE>(va, rays, red, dA, degenerate, tid);
            if (two) normalize_store<MAX_RAYS_PER_LANE>(vb, rays, red, dB, degenerate, tid);
            __syncthreads();   // red is reused by the next round
        }
        if (threadIdx.x == 0) s_next = gridDim.x + atomicAdd(next_pair, 1u);
        __syncthreads();
        stamp(3);
        pair = s_next;
    }
}

template <int M, int S>
void* fused_kernel(int pf)
{
    // `pf` load triplets in flight per lane = two register stages of pf / 2 quads each
    return pf >= 4 ? reinterpret_cast<void*>(k_bev_radon2<M, S, 2>) : reinterpret_cast<void*>(k_bev_radon2<M, S, 1>);
}

template <int M, int S, bool ONCE = false>
void* fused_kernel_slots(int pf)
{
    return pf >= 6   ? reinterpret_cast<void*>(k_bev_radon3<M, S, 3, ONCE>)
           : pf >= 4 ? reinterpret_cast<void*>(k_bev_radon3<M, S, 2, ONCE>)
                     : reinterpret_cast<void*>(k_bev_radon3<M, S, 1, ONCE>);
}

}  // namespace

extern "C" {

int mrs_radon_plan_set_option(mrs_radon_plan* plan, int32_t option, int32_t value)
{
    MRS_REQUIRE(plan, "null plan");
    switch (option) {
        case MRS_RADON_OPT_FUSED_STAGGER_US:
            MRS_REQUIRE(value >= 0 && value <= 100000, "stagger must be within [0, 100000] microseconds");
            plan->fused_stagger_us = value;
            return MRS_OK;
        case MRS_RADON_OPT_FUSED_PREFETCH:
            MRS_REQUIRE(value == 2 || value == 4 || value == 6, "prefetch depth must be 2, 4 or 6");
            plan->fused_prefetch = value;
            return MRS_OK;
        case MRS_RADON_OPT_FUSED_GRID:
            MRS_REQUIRE(value >= 0 && value <= 65535, "workgroup count must be within [0, 65535]");
            plan->fused_grid = value;
            return MRS_OK;
        case MRS_RADON_OPT_FUSED_VARIANT:
            MRS_REQUIRE(value >= 0 && value <= 2, "variant must be 0 (ray-order table, unrolled), 1 (slot tables) or 2 (slot tables, sinogram written once)");
            MRS_REQUIRE(value == 0 || plan->d_slot, "this plan has no slot tables");
            plan->fused_variant = value;
            return MRS_OK;
        case MRS_RADON_OPT_FUSED_SKIP:
            MRS_REQUIRE(value >= 0 && value <= 2, "skip must be 0 (run everything), 1 (no rasterising) or 2 (no ray march)");
            plan->fused_skip = value;
            return MRS_OK;
        default:
            mrs::set_error("unknown plan option %d", (int)option);
            return MRS_ERR_ARG;
    }
}

int mrs_ring_descriptors_batch(mrs_radon_plan* plan, const float* d_xyz, const int64_t* d_offsets, int32_t batch,
                               const mrs_bev_cfg* cfg, float* d_bev, float* d_sino, float* d_sino_norm, mrs_stream stream)
{
    MRS_REQUIRE(plan && d_xyz && d_offsets && cfg, "null pointer");
    MRS_REQUIRE(d_bev || d_sino || d_sino_norm, "at least one output required");
    MRS_REQUIRE(batch > 0, "batch must be positive");
    MRS_HIP_TRY(hipSetDevice(plan->ctx->device));
    CartP cp;
    int st = make_cart(cfg, false, cp);
    if (st != MRS_OK) return st;
    const int rays = plan->n_angles * plan->det;
    const int per_lane = (rays + kRadonWG - 1) / kRadonWG;
    if (cfg->num_height != 1 || cfg->n0 != plan->H || cfg->n1 != plan->W || !plan->two_in_lds || per_lane > 16) {
        mrs::set_error("fused descriptor kernel needs num_height == 1, a %d x %d grid (the plan's image), two images in the LDS and at most "
                       "16384 rays: run mrs_bev_cart_batch + mrs_radon_forward instead", plan->H, plan->W);
        return MRS_ERR_UNSUPPORTED;
    }
    RadonP p;
    p.A = plan->n_angles; p.D = plan->det; p.H = plan->H; p.W = plan->W;
    p.stride = (plan->W + 2 * kPad) | 1;
    const size_t nr = (size_t)p.A * p.D;
    p.meta = plan->d_meta;
    p.base = plan->d_meta + nr;
    p.q = reinterpret_cast<const float*>(plan->d_meta + 2 * nr);
    p.vm = p.q + nr;
    p.nrm = p.vm + nr;
    const size_t lds = 2 * (size_t)(p.H + 2 * kPad) * p.stride * sizeof(float);
    hipStream_t s = (hipStream_t)stream;
    mrs::Scratch ctr;
    if ((st = ctr.alloc(256, s)) != MRS_OK) return st;
    MRS_HIP_TRY(hipMemsetAsync(ctr.p, 0, sizeof(unsigned), s));
    const int pf = plan->fused_prefetch;
    const int pairs = (batch + 1) / 2;
    const int grid = std::min(pairs, plan->fused_grid > 0 ? plan->fused_grid : std::max(plan->ctx->num_cu, 1));
    unsigned stagger_ticks = (unsigned)plan->fused_stagger_us * 100u;   // wall_clock64 ticks at 100 MHz
    if (pairs <= grid) stagger_ticks = 0;   // a single round: nothing to phase-shift, the delay would only add latency
    unsigned* d_ctr = ctr.as<unsigned>();
    int* d_deg = plan->d_degenerate;
    if (plan->fused_variant >= 1 && plan->d_slot && plan->slot_per_lane == per_lane) {
        // variant 2 parks 2 x rays raw sums in the tile after the march: they must fit it
        const bool once = plan->fused_variant == 2 && 2 * (size_t)rays * sizeof(float) <= lds;
        void* kern = once ? (per_lane <= 15 ? (p.stride == 125 ? fused_kernel_slots<15, 125, true>(pf) : fused_kernel_slots<15, 0, true>(pf))
                                            : fused_kernel_slots<16, 0, true>(pf))
                          : (per_lane <= 15 ? (p.stride == 125 ? fused_kernel_slots<15, 125>(pf) : fused_kernel_slots<15, 0>(pf))
                                            : fused_kernel_slots<16, 0>(pf));
        if (lds > 48 * 1024) MRS_HIP_TRY(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        SlotP sp;
        sp.slot = plan->d_slot; sp.nrm = plan->d_slot_nrm; sp.ray = plan->d_slot_ray;
        mrs::Scratch parkbuf;          // only a BEV-only call has no output to park the raw sums in (they are then never read)
        float* d_park = nullptr;
        if (!d_sino && !d_sino_norm) {
            if ((st = parkbuf.alloc((size_t)grid * 2 * rays * sizeof(float), s)) != MRS_OK) return st;
            d_park = parkbuf.as<float>();
        }
        static const bool want_prof = mrs::dev_env("MRS_FUSED_PROF") != nullptr;     // development aid: phase times on stderr (synchronises)
        static const char* const dev_skip_s = mrs::dev_env("MRS_FUSED_SKIP");
        static const int dev_skip_env = dev_skip_s ? atoi(dev_skip_s) : 0;
        int dev_skip = dev_skip_env | plan->fused_skip;
        unsigned long long* d_prof = nullptr;
        if (want_prof) {
            MRS_HIP_TRY(hipMalloc(&d_prof, 4 * sizeof(unsigned long long)));
            MRS_HIP_TRY(hipMemsetAsync(d_prof, 0, 4 * sizeof(unsigned long long), s));
        }
        void* args[] = {(void*)&d_xyz, (void*)&d_offsets, (void*)&cp, (void*)&p, (void*)&sp, (void*)&batch, (void*)&d_bev, (void*)&d_sino,
                        (void*)&d_sino_norm, (void*)&d_park, (void*)&d_deg, (void*)&d_ctr, (void*)&stagger_ticks, (void*)&d_prof, (void*)&dev_skip};
        MRS_HIP_TRY(hipLaunchKernel(kern, dim3(grid), dim3(kRadonWG), args, lds, s));
        if (want_prof) {
            unsigned long long h[4];
            MRS_HIP_TRY(hipStreamSynchronize(s));
            MRS_HIP_TRY(hipMemcpy(h, d_prof, sizeof(h), hipMemcpyDeviceToHost));
            (void)hipFree(d_prof);
            const double per = 0.01 / (double)pairs;     // 100 MHz ticks -> microseconds per pair of scans
            fprintf(stderr, "[fused prof] %d scans, grid %d, pf %d: clear %.1f us, rasterise %.1f us, march %.1f us, normalise + next %.1f us per pair\n", batch,
                    grid, pf, h[0] * per, h[1] * per, h[2] * per, h[3] * per);
        }
        return MRS_OK;
    }
    void* kern = per_lane <= 15 ? (p.stride == 125 ? fused_kernel<15, 125>(pf) : fused_kernel<15, 0>(pf)) : fused_kernel<16, 0>(pf);
    if (lds > 48 * 1024) MRS_HIP_TRY(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    void* args[] = {(void*)&d_xyz, (void*)&d_offsets, (void*)&cp, (void*)&p, (void*)&batch, (void*)&d_bev, (void*)&d_sino,
                    (void*)&d_sino_norm, (void*)&d_deg, (void*)&d_ctr, (void*)&stagger_ticks};
    MRS_HIP_TRY(hipLaunchKernel(kern, dim3(grid), dim3(kRadonWG), args, lds, s));
    return MRS_OK;
}

}  // extern "C"
