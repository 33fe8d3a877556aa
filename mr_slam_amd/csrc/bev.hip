// bev.hip -- BEV rasterisers for gfx950.
//
// Reference behaviour reproduced (never copied):
//   polar : disco_ros/tools/multi-layer-polar-cpu/cython/src/kernel.cpp:23-77, manager.cpp:41-58
//   cart  : generate_bev_cython_binary/src/kernel.cu:14-61, manager.cu:53-91
//   feat  : generate_bev_pointfeat_cython/src/kernel.cu:106-164
//
// Design (HBM-bound byte/integer work, no MFMA):
//   * one workgroup (16 waves) per scan streams the three SoA planes with non-temporal 16-byte loads
//     and rasterises into an LDS-private grid (ds_max on the Cartesian max-z grid; an interleaved
//     power-of-two occupancy bitset with read-before-ds_or for the polar one), then writes the grid
//     once, coalesced: HBM traffic = 12 B/point + 4 B/cell, no global atomics on the hot path;
//     the common case is ONE combined test per point (fast path), everything else re-evaluates exactly;
//   * cell indices are bit-exact with the CPU reference, which evaluates atan/sqrt/div in
//     double: each axis takes an fp32 fast path whose error is bounded far below `eps`
//     bins, and any lane whose quotient lands within eps of a bin edge re-evaluates the
//     reference formula exactly (IEEE double add/div/sqrt; the sector, whose atan comes
//     from the host libm, through a host-built table of change points);
//   * the bit-for-bit `retreive()` layouts (order-dependent x/y payloads, enough_large > 1,
//     num_height > 1) go through deterministic multi-pass kernels on global scratch.
#include <hipcub/hipcub.hpp>

#include <climits>
#include <cmath>

#include "common.hpp"
#include "bev_cart.hpp"

namespace {

constexpr int kDrop = INT32_MIN;
constexpr int kWG = 1024;

// ------------------------------------------------------------------------------------------
// parameters
// ------------------------------------------------------------------------------------------
struct PolarP {
    float gap_ring, gap_sector, gap_height;  // exactly the floats the reference computes
    float inv_ring, inv_sector, inv_height;
    float mh;                                // (float)max_height
    float eps_ring, eps_sector, eps_height;
    int R, S, H, K;                          // K = enough_large
    const float* thr;                        // sector change points, [4][stride]
    const int* val;
    int stride;
    int cnt[4];
};

// ------------------------------------------------------------------------------------------
// per-point cell computation
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ bool floor_to_int(double q, int& k)
{
    double f = floor(q);
    if (!(f > -1073741824.0 && f < 1073741824.0)) return false;
    k = (int)f;
    return true;
}

// atan(t), t in [0,1], |err| < 1.5e-7 rad in fp32 (odd minimax polynomial, degree 15)
__device__ __forceinline__ float atan01(float t)
{
    const float u = t * t;
    float p = -0.004054562654346228f;
    p = __builtin_fmaf(p, u, 0.02186294086277485f);
    p = __builtin_fmaf(p, u, -0.05591230094432831f);
    p = __builtin_fmaf(p, u, 0.09642195701599121f);
    p = __builtin_fmaf(p, u, -0.1390862911939621f);
    p = __builtin_fmaf(p, u, 0.19946566224098206f);
    p = __builtin_fmaf(p, u, -0.33329859375953674f);
    p = __builtin_fmaf(p, u, 0.9999993443489075f);
    return p * t;
}

// kernel.cpp:40-77 for one point.  Returns false for points the reference mishandles.
__device__ __forceinline__ bool polar_cell(const PolarP& p, float x, float y, float z, int& kr,
                                           int& ks, int& kh)
{
    if (x == 0.0f) x = 0.0001f;
    if (y == 0.0f) y = 0.0001f;
    if (z == 0.0f) z = 0.0001f;
    if (!(x == x) || !(y == y) || !(z == z)) return false;

    // ---- ring: floor( float(sqrt(double(x^2+y^2))) / gap_ring ), clamped to R-1
    {
        const float s = __builtin_fmaf(x, x, y * y);
        const float g = __builtin_amdgcn_sqrtf(s) * p.inv_ring;
        const float f = floorf(g);
        const float fr = g - f;
        bool exact_needed;
        if (g >= (float)p.R + 1.0f)
            exact_needed = !(g < 5.0e8f);  // far outside: clamps, unless the int cast overflows
        else
            exact_needed = !(fr >= p.eps_ring && fr <= 1.0f - p.eps_ring);
        if (exact_needed) {
            const double sd = fma((double)x, (double)x, (double)y * (double)y);
            const float far = (float)sqrt(sd);
            if (!floor_to_int((double)(far / p.gap_ring), kr)) return false;
        } else {
            kr = g >= (float)p.R ? p.R : (int)f;
        }
        if (kr >= p.R) kr = p.R - 1;
    }
    // ---- sector: floor( theta / gap_sector ), theta from xy2theta (kernel.cpp:23-36)
    {
        const float ax = fabsf(x), ay = fabsf(y);
        const bool steep = ay > ax;
        const float t = (steep ? ax : ay) * __builtin_amdgcn_rcpf(steep ? ay : ax);
        float a = atan01(t) * 57.295779513f;
        if (steep) a = 90.0f - a;
        const int quad = (x < 0.0f ? 1 : 0) ^ (y < 0.0f ? 3 : 0);  // 0:I 1:II 2:III 3:IV
        float theta = a;
        if (quad == 1) theta = 180.0f - a;
        if (quad == 2) theta = 180.0f + a;
        if (quad == 3) theta = 360.0f - a;
        const float g = theta * p.inv_sector;
        const float f = floorf(g);
        const float fr = g - f;
        if (fr >= p.eps_sector && fr <= 1.0f - p.eps_sector) {
            ks = (int)f;
        } else {
            const float q = ay / ax;  // IEEE division == the reference's _y/_x in every quadrant
            if (!(q == q)) return false;
            const float* thr = p.thr + quad * p.stride;
            int lo = 0, hi = p.cnt[quad];  // last j with thr[j] <= q  (thr[0] == 0)
            while (hi - lo > 1) {
                const int mid = (lo + hi) >> 1;
                if (thr[mid] <= q) lo = mid; else hi = mid;
            }
            ks = p.val[quad * p.stride + lo];
            if (ks == kDrop) return false;
        }
    }
    // ---- height: floor( (z + max_height) / gap_height ) in float
    {
        const float g = (z + p.mh) * p.inv_height;
        const float f = floorf(g);
        const float fr = g - f;
        const bool far_out = !(fabsf(g) < (float)(2 * p.H + 16));
        if (!far_out && fr >= p.eps_height && fr <= 1.0f - p.eps_height) {
            kh = (int)f;
        } else {
            if (!floor_to_int((double)((z + p.mh) / p.gap_height), kh)) return false;
        }
    }
    return true;
}

__device__ __forceinline__ int polar_lin(const PolarP& p, float x, float y, float z)
{
    int kr, ks, kh;
    if (!polar_cell(p, x, y, z, kr, ks, kh)) return -1;
    const long long lin = (long long)ks + (long long)kr * p.S + (long long)kh * p.S * p.R;
    const long long cells = (long long)p.R * p.S * p.H;
    return (lin < 0 || lin >= cells) ? -1 : (int)lin;
}

// ------------------------------------------------------------------------------------------
// A1 / A3: index kernels (one scan)
// ------------------------------------------------------------------------------------------
__global__ void k_polar_indices(const float* __restrict__ xyz, int n, PolarP p, int* __restrict__ ring,
                                int* __restrict__ sector, int* __restrict__ height)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        int kr, ks, kh;
        const bool ok = polar_cell(p, xyz[i], xyz[i + n], xyz[i + 2 * (size_t)n], kr, ks, kh);
        ring[i] = ok ? kr : kDrop;
        sector[i] = ok ? ks : kDrop;
        height[i] = ok ? kh : kDrop;
    }
}

__global__ void k_cart_indices(const float* __restrict__ xyz, int n, CartP p, int* __restrict__ ox,
                               int* __restrict__ oy, int* __restrict__ oh)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        int ix, iy, ih;
        const bool ok = cart_cell(p, xyz[i], xyz[i + n], xyz[i + 2 * (size_t)n], ix, iy, ih);
        ox[i] = ok ? ix : kDrop;
        oy[i] = ok ? iy : kDrop;
        oh[i] = ok ? ih : kDrop;
    }
}

// ------------------------------------------------------------------------------------------
// hot path: LDS-private rasterisers, one workgroup per scan, COMPACT output
// ------------------------------------------------------------------------------------------
// Cartesian max-z (num_height == 1): ch2 = max positive z per cell (manager.cu:57,69-72 with
// max_h initialised to 0).  Positive floats order like their bit patterns -> integer ds_max.
__global__ __launch_bounds__(kWG) void k_cart_lds(const float* __restrict__ xyz,
                                                  const int64_t* __restrict__ offs, CartP p,
                                                  float* __restrict__ out)
{
    extern __shared__ __attribute__((aligned(16))) int grid[];
    const int b = blockIdx.x;
    const int64_t o = offs[b];
    const int n = (int)(offs[b + 1] - o);
    const float* px = xyz + 3 * o;
    const float* py = px + n;
    const float* pz = py + n;
    const int cells = p.NX * p.NY;
    for (int i = threadIdx.x; i < cells; i += kWG) grid[i] = 0;
    __syncthreads();

    // Only z > 0 can change this map (max_h starts at 0: manager.cu:57,69).  Common case in one test:
    // 0 < z < 1 and 0 < |x|,|y| <= 1 (no zero substitution, no clamp, height layer 0) and both quotients
    // at least eps away from a bin edge -> the fp32 quotient's floor IS the reference's double floor.
    // Everything else (rare) takes the exact per-axis path of cart_lin().
    const float eps = fmaxf(p.eps_x, p.eps_y);
    const float inv_x = p.inv_x, inv_y = p.inv_y;
    const int NY = p.NY;
    auto put = [&](float x, float y, float z) {
        const float gx = __builtin_fmaf(x, inv_x, inv_x), gy = __builtin_fmaf(y, inv_y, inv_y);
        const float fx = floorf(gx), fy = floorf(gy);
        const float ex = 0.5f - fabsf((gx - fx) - 0.5f), ey = 0.5f - fabsf((gy - fy) - 0.5f);  // distance to a bin edge
        // every comparison is false for a NaN operand (fmaxf / fminf would drop the NaN instead): NaN x or y leave the fast path
        const bool fast = (bool)((int)(z > 0.0f) & (int)(z < 1.0f) & (int)(fabsf(x) <= 1.0f) & (int)(fabsf(y) <= 1.0f) & (int)(x * y != 0.0f) &
                                 (int)(ex >= eps) & (int)(ey >= eps));
        if (fast) {
            // consecutive lidar returns share cells: a plain read broadcasts where same-address atomics serialise,
            // and most points do not raise the maximum
            int* cell = &grid[(int)fy + (int)fx * NY];
            const int zi = __float_as_int(z);
            if (*cell < zi) atomicMax(cell, zi);
        } else if (z > 0.0f) {
            int col;
            const int lin = cart_lin(p, x, y, z, col);
            if (lin >= 0) atomicMax(&grid[lin], __float_as_int(z));
        }
    };
    int done = 0;
    if (aligned16(px) && aligned16(py) && aligned16(pz)) {
        const int n4 = n >> 2;
        const float4* x4 = reinterpret_cast<const float4*>(px);
        const float4* y4 = reinterpret_cast<const float4*>(py);
        const float4* z4 = reinterpret_cast<const float4*>(pz);
#pragma unroll 2
        for (int i = threadIdx.x; i < n4; i += kWG) {
            const float4 X = stream_load4(x4 + i), Y = stream_load4(y4 + i), Z = stream_load4(z4 + i);
            put(X.x, Y.x, Z.x);
            put(X.y, Y.y, Z.y);
            put(X.z, Y.z, Z.z);
            put(X.w, Y.w, Z.w);
        }
        done = n4 << 2;
    }
    for (int i = done + threadIdx.x; i < n; i += kWG) put(px[i], py[i], pz[i]);
    __syncthreads();
    float* dst = out + (size_t)b * cells;
    if ((cells & 3) == 0) {  // 16-byte LDS reads and stores (dst is 16-byte aligned when cells % 4 == 0 and out is)
        const int4* g4 = reinterpret_cast<const int4*>(grid);
        float4* d4 = reinterpret_cast<float4*>(dst);
        if (aligned16(dst)) {
            for (int i = threadIdx.x; i < (cells >> 2); i += kWG) {
                const int4 v = g4[i];
                stream_store4(d4 + i, make_float4(__int_as_float(v.x), __int_as_float(v.y), __int_as_float(v.z), __int_as_float(v.w)));
            }
            return;
        }
    }
    for (int i = threadIdx.x; i < cells; i += kWG) dst[i] = __int_as_float(grid[i]);
}

// Polar occupancy (enough_large == 1): ch2 = 1 for any cell that received a point.
__global__ __launch_bounds__(kWG) void k_polar_lds(const float* __restrict__ xyz,
                                                   const int64_t* __restrict__ offs, PolarP p,
                                                   float* __restrict__ out)
{
    extern __shared__ __attribute__((aligned(16))) unsigned bits[];
    const int b = blockIdx.x;
    const int64_t o = offs[b];
    const int n = (int)(offs[b + 1] - o);
    const float* px = xyz + 3 * o;
    const float* py = px + n;
    const float* pz = py + n;
    const int cells = p.R * p.S * p.H;
    int wshift = 0;  // words = smallest power of two >= cells / 32: cell c -> word c & (words-1), bit c >> wshift
    while ((32 << wshift) < cells) ++wshift;
    const int words = 1 << wshift, wmask = words - 1;
    for (int i = threadIdx.x; i < words; i += kWG) bits[i] = 0u;
    __syncthreads();

    // Common case in ONE test: no zero coordinate, and the ring / sector / height quotients all at least
    // eps away from a bin edge (fp32 fast path == the reference's double path there); everything else
    // (rare) re-evaluates exactly through polar_lin().
    const float fR = (float)p.R, hmax = (float)(2 * p.H + 16);
    const float hi_ring = 1.0f - p.eps_ring, hi_sector = 1.0f - p.eps_sector, hi_height = 1.0f - p.eps_height;
    auto put = [&](float x, float y, float z) {
        // quotients g >= 0 on the fast path: bin = (int)g (truncation == floor), distance to a bin edge from v_fract
        const float gr = __builtin_amdgcn_sqrtf(__builtin_fmaf(x, x, y * y)) * p.inv_ring;
        const float ax = fabsf(x), ay = fabsf(y);
        const bool steep = ay > ax;
        const float t = (steep ? ax : ay) * __builtin_amdgcn_rcpf(steep ? ay : ax);
        float a = atan01(t) * 57.295779513f;
        a = steep ? 90.0f - a : a;
        const bool xn = x < 0.0f, yn = y < 0.0f;
        const float base = xn ? 180.0f : (yn ? 360.0f : 0.0f);
        const float theta = (xn != yn) ? base - a : base + a;
        const float gs = theta * p.inv_sector;
        const float sh = z + p.mh;
        float gh = sh * p.inv_height;
        // height edge (ground returns sit right on the z = 0 bin edge after the reference's z > 0 crop): the
        // reference quotient is a plain fp32 division -> redo just that, IEEE-rounded, instead of leaving the fast path
        {
            const float fh0 = __builtin_amdgcn_fractf(gh);
            if (!(fh0 >= p.eps_height) | !(fh0 <= hi_height)) gh = sh / p.gap_height;
        }
        const float frr = __builtin_amdgcn_fractf(gr), frs = __builtin_amdgcn_fractf(gs);
        // every comparison is false for NaN; gr >= R + 1 clamps to the last ring whatever its fraction
        const bool ring_ok = (gr >= fR + 1.0f) | ((frr >= p.eps_ring) & (frr <= hi_ring));
        const bool sect_ok = (frs >= p.eps_sector) & (frs <= hi_sector);
        const bool fast = (bool)((int)(x * y * z != 0.0f) & (int)ring_ok & (int)sect_ok & (int)(gr < 5.0e8f) & (int)(gh >= 0.0f) &
                                 (int)(gh < hmax));
        const float fr = gr, fs = gs, fh = gh;   // (int) of these below
        int lin;
        if (fast) {
            const int kr = gr >= fR ? p.R - 1 : (int)fr;
            lin = (int)fs + kr * p.S + (int)fh * (p.S * p.R);
            if ((unsigned)lin >= (unsigned)cells) lin = -1;
        } else {
            lin = polar_lin(p, x, y, z);
        }
        if (lin >= 0) {
            // bit of cell c lives in word c % words, position c / words: neighbouring cells (consecutive lidar
            // returns) land in different words and banks instead of fighting over one dword
            const int w = lin & wmask;
            const unsigned m = 1u << (lin >> wshift);
            if (!(bits[w] & m)) atomicOr(&bits[w], m);
        }
    };
    int done = 0;
    if (aligned16(px) && aligned16(py) && aligned16(pz)) {
        const int n4 = n >> 2;
        const float4* x4 = reinterpret_cast<const float4*>(px);
        const float4* y4 = reinterpret_cast<const float4*>(py);
        const float4* z4 = reinterpret_cast<const float4*>(pz);
#pragma unroll 2
        for (int i = threadIdx.x; i < n4; i += kWG) {
            const float4 X = stream_load4(x4 + i), Y = stream_load4(y4 + i), Z = stream_load4(z4 + i);
            put(X.x, Y.x, Z.x);
            put(X.y, Y.y, Z.y);
            put(X.z, Y.z, Z.z);
            put(X.w, Y.w, Z.w);
        }
        done = n4 << 2;
    }
    for (int i = done + threadIdx.x; i < n; i += kWG) put(px[i], py[i], pz[i]);
    __syncthreads();
    float* dst = out + (size_t)b * cells;
    if ((cells & 3) == 0 && wshift >= 2 && aligned16(dst)) {
        // cells 4j..4j+3 sit in four consecutive words at the same bit position: one 16-byte LDS read, one 16-byte store
        const uint4* b4 = reinterpret_cast<const uint4*>(bits);
        float4* d4 = reinterpret_cast<float4*>(dst);
        const int w4mask = wmask >> 2;
        for (int j = threadIdx.x; j < (cells >> 2); j += kWG) {
            const uint4 v = b4[j & w4mask];
            const int sh = (4 * j) >> wshift;
            stream_store4(d4 + j, make_float4((float)((v.x >> sh) & 1u), (float)((v.y >> sh) & 1u), (float)((v.z >> sh) & 1u),
                                              (float)((v.w >> sh) & 1u)));
        }
        return;
    }
    for (int i = threadIdx.x; i < cells; i += kWG)
        dst[i] = (bits[i & wmask] >> (i >> wshift)) & 1u ? 1.0f : 0.0f;
}

// ------------------------------------------------------------------------------------------
// bit-for-bit `retreive()` layouts: deterministic multi-pass kernels on global scratch.
// grid = (blocks, batch); every kernel is a grid-stride loop over the scan's points / cells.
// ------------------------------------------------------------------------------------------
struct ScanView {
    const float* px;
    const float* py;
    const float* pz;
    int n;
};
__device__ __forceinline__ ScanView scan_view(const float* xyz, const int64_t* offs, int b, int planes)
{
    const int64_t o = offs[b];
    ScanView v;
    v.n = (int)(offs[b + 1] - o);
    v.px = xyz + (int64_t)planes * o;
    v.py = v.px + v.n;
    v.pz = v.py + v.n;
    return v;
}

__global__ void k_fill_i32(int* p, size_t n, int v)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        p[i] = v;
}

// polar pass k: smallest point index per cell that is larger than pass k-1's (manager.cpp:46-56:
// the first `enough_large` points of a cell in input order)
__global__ void k_polar_kth(const float* __restrict__ xyz, const int64_t* __restrict__ offs, PolarP p,
                            int k, int* __restrict__ kidx /* [batch][K][cells] */)
{
    const int b = blockIdx.y;
    const ScanView v = scan_view(xyz, offs, b, 3);
    const size_t cells = (size_t)p.R * p.S * p.H;
    int* cur = kidx + ((size_t)b * p.K + k) * cells;
    const int* prev = k ? cur - cells : nullptr;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < v.n; i += gridDim.x * blockDim.x) {
        const int lin = polar_lin(p, v.px[i], v.py[i], v.pz[i]);
        if (lin < 0) continue;
        if (prev && i <= prev[lin]) continue;
        // a plain (possibly stale, i.e. too large) read first: points arrive roughly in index order, so after the first few
        // a hot cell no longer sees same-address atomics queueing up in L2
        if (__atomic_load_n(&cur[lin], __ATOMIC_RELAXED) <= i) continue;
        atomicMin(&cur[lin], i);
    }
}

__global__ void k_polar_emit(const float* __restrict__ xyz, const int64_t* __restrict__ offs, PolarP p,
                             const int* __restrict__ kidx, float* __restrict__ out)
{
    const int b = blockIdx.y;
    const ScanView v = scan_view(xyz, offs, b, 3);
    const size_t slots = (size_t)p.R * p.S * p.H * p.K;  // [k][cell] == reference slab order
    const int* src = kidx + (size_t)b * slots;
    float* dst = out + (size_t)b * slots * 3;
    for (size_t c = blockIdx.x * (size_t)blockDim.x + threadIdx.x; c < slots; c += (size_t)gridDim.x * blockDim.x) {
        const int i = src[c];
        const bool hit = i != INT_MAX;
        dst[3 * c + 0] = hit ? v.px[i] : 0.0f;
        dst[3 * c + 1] = hit ? v.py[i] : 0.0f;
        dst[3 * c + 2] = hit ? 1.0f : 0.0f;
    }
}

// Cartesian pass 1: last point per cell (x/y payload), and either the max positive z
// (H == 1) or the first positive-z point per cell (H > 1, feeds pass 2)
__global__ void k_cart_pass1(const float* __restrict__ xyz, const int64_t* __restrict__ offs, CartP p,
                             int* __restrict__ last_idx, int* __restrict__ zbits, int* __restrict__ first_pos)
{
    const int b = blockIdx.y;
    const ScanView v = scan_view(xyz, offs, b, 3);
    const size_t cells = (size_t)p.NX * p.NY * p.H;
    last_idx += b * cells;
    zbits += b * cells;
    if (first_pos) first_pos += b * cells;
    // LAST point per cell: walk the scan from its end, so that the first arrival at a hot cell already holds (nearly) the final
    // answer and the plain (possibly stale, i.e. too small) read lets the rest skip their same-address atomics
    for (int g = blockIdx.x * blockDim.x + threadIdx.x; g < v.n; g += gridDim.x * blockDim.x) {
        const int i = v.n - 1 - g;
        int col;
        const float z = v.pz[i];
        const int lin = cart_lin(p, v.px[i], v.py[i], z, col);
        if (lin < 0) continue;
        if (__atomic_load_n(&last_idx[lin], __ATOMIC_RELAXED) < i) atomicMax(&last_idx[lin], i);
        if (z > 0.0f) {
            if (first_pos) atomicMin(&first_pos[lin], i);
            else if (__atomic_load_n(&zbits[lin], __ATOMIC_RELAXED) < __float_as_int(z)) atomicMax(&zbits[lin], __float_as_int(z));
        }
    }
}

// Cartesian pass 2 (H > 1).  The reference keeps ONE running max per column while writing
// the value into the point's own layer, so a point i in layer h sets the cell iff no earlier
// point of the column is at least as high.  Layers are monotone in z, hence: i must precede
// the first positive point of every higher layer, and among those the cell keeps the max.
__global__ void k_cart_pass2(const float* __restrict__ xyz, const int64_t* __restrict__ offs, CartP p,
                             const int* __restrict__ first_pos, int* __restrict__ zbits)
{
    const int b = blockIdx.y;
    const ScanView v = scan_view(xyz, offs, b, 3);
    const size_t cols = (size_t)p.NX * p.NY;
    const size_t cells = cols * p.H;
    first_pos += b * cells;
    zbits += b * cells;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < v.n; i += gridDim.x * blockDim.x) {
        int col;
        const float z = v.pz[i];
        if (!(z > 0.0f)) continue;
        const int lin = cart_lin(p, v.px[i], v.py[i], z, col);
        if (lin < 0) continue;
        const int h = lin / (int)cols;
        int t = INT_MAX;
        for (int hh = h + 1; hh < p.H; ++hh) t = min(t, first_pos[col + (size_t)hh * cols]);
        if (i < t) atomicMax(&zbits[lin], __float_as_int(z));
    }
}

__global__ void k_cart_emit(const float* __restrict__ xyz, const int64_t* __restrict__ offs, CartP p,
                            const int* __restrict__ last_idx, const int* __restrict__ zbits,
                            float* __restrict__ out)
{
    const int b = blockIdx.y;
    const ScanView v = scan_view(xyz, offs, b, 3);
    const size_t cells = (size_t)p.NX * p.NY * p.H;
    last_idx += b * cells;
    zbits += b * cells;
    float* dst = out + (size_t)b * cells * 3;
    for (size_t c = blockIdx.x * (size_t)blockDim.x + threadIdx.x; c < cells; c += (size_t)gridDim.x * blockDim.x) {
        const int i = last_idx[c];
        dst[3 * c + 0] = i >= 0 ? v.px[i] : 0.0f;
        dst[3 * c + 1] = i >= 0 ? v.py[i] : 0.0f;
        dst[3 * c + 2] = __int_as_float(zbits[c]);
    }
}

__global__ void k_copy_i32_as_f32(const int* __restrict__ src, float* __restrict__ dst, size_t n)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        dst[i] = __int_as_float(src[i]);
}

// A5: per-cell, per-channel max of positive values.  acc is [batch][cells][F] int bits
// (REFERENCE layout) or [batch][F-3][cells] (COMPACT, channels 3..F-1), pre-zeroed.
template <bool COMPACT>
__global__ void k_feat_max(const float* __restrict__ pts, const int64_t* __restrict__ offs, CartP p,
                           int* __restrict__ acc)
{
    const int b = blockIdx.y;
    const ScanView v = scan_view(pts, offs, b, p.F);
    const size_t cols = (size_t)p.NX * p.NY;
    const size_t cells = cols * p.H;
    int* dst = acc + (size_t)b * cells * (COMPACT ? p.F - 3 : p.F);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < v.n; i += gridDim.x * blockDim.x) {
        int col;
        const int lin = cart_lin(p, v.px[i], v.py[i], v.pz[i], col);
        if (lin < 0) continue;
        for (int j = COMPACT ? 3 : 0; j < p.F; ++j) {
            const float f = v.px[i + (size_t)j * v.n];
            if (f > 0.0f) {   // most points do not raise a cell's maximum: plain (possibly stale = too small) read before the atomic
                int* cell = COMPACT ? &dst[(size_t)(j - 3) * cells + lin] : &dst[(size_t)lin * p.F + j];
                if (__atomic_load_n(cell, __ATOMIC_RELAXED) < __float_as_int(f)) atomicMax(cell, __float_as_int(f));
            }
        }
    }
}

// A5, compact layout, one height layer (what generate_RINGplusplus consumes): one workgroup per (channel, scan) with the channel's grid in
// the LDS -- no global atomics, one coalesced write-out, no memset.  The x, y, z planes are read once per channel (16 B per point and
// channel instead of the 36 B a single pass over 9 planes would need): 64 scans x 6 channels in 0.15 ms against 2.35 ms for k_feat_max.
__global__ __launch_bounds__(1024) void k_feat_lds(const float* __restrict__ pts, const int64_t* __restrict__ offs, CartP p, float* __restrict__ out)
{
    extern __shared__ int grid_i[];
    const int b = blockIdx.y, j = 3 + (int)blockIdx.x;
    const ScanView v = scan_view(pts, offs, b, p.F);
    const int cells = p.NX * p.NY;
    for (int c = threadIdx.x; c < cells; c += blockDim.x) grid_i[c] = 0;
    __syncthreads();
    const float* plane = v.px + (size_t)j * v.n;
    for (int i = threadIdx.x; i < v.n; i += blockDim.x) {
        const float f = __builtin_nontemporal_load(plane + i);
        if (!(f > 0.0f)) continue;
        int col;
        const int lin = cart_lin(p, v.px[i], v.py[i], v.pz[i], col);
        if (lin >= 0 && grid_i[lin] < __float_as_int(f)) atomicMax(&grid_i[lin], __float_as_int(f));
    }
    __syncthreads();
    float* dst = out + ((size_t)b * (p.F - 3) + (j - 3)) * cells;
    for (int c = threadIdx.x; c < cells; c += blockDim.x) dst[c] = __int_as_float(grid_i[c]);
}

// A5 with num_height > 1.  The reference keeps ONE running maximum per (column, channel) while it writes the value into
// the point's own height layer (kernel.cu:151-158: max_h is indexed without the layer), so what a layer's cell holds
// depends on the ORDER of the column's points: it is the last point of that layer that raised the column's running
// maximum.  Sequential reading (threads in gid order = what the host build of the reference executes): stable sort of
// the points by (scan, column), then one thread walks each column's points in input order with the running maxima in
// registers.  Values <= 0 never pass `max_h < value` (max_h starts at 0).
constexpr int kFeatMax = 16;

__global__ void k_feat_keys(const float* __restrict__ pts, const int64_t* __restrict__ offs, CartP p,
                            unsigned long long* __restrict__ keys, int* __restrict__ vals)
{
    const int b = blockIdx.y;
    const ScanView v = scan_view(pts, offs, b, p.F);
    const int64_t o = offs[b];
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < v.n; i += gridDim.x * blockDim.x) {
        int col = 0;
        const int lin = cart_lin(p, v.px[i], v.py[i], v.pz[i], col);
        keys[o + i] = lin < 0 ? ~0ull : (((unsigned long long)b << 32) | (unsigned)col);
        vals[o + i] = i;
    }
}

template <bool COMPACT>
__global__ void k_feat_columns(const float* __restrict__ pts, const int64_t* __restrict__ offs, CartP p,
                               const unsigned long long* __restrict__ keys, const int* __restrict__ perm, size_t total,
                               float* __restrict__ out)
{
    const size_t cols = (size_t)p.NX * p.NY;
    const size_t cells = cols * p.H;
    for (size_t j = blockIdx.x * (size_t)blockDim.x + threadIdx.x; j < total; j += (size_t)gridDim.x * blockDim.x) {
        const unsigned long long k = keys[j];
        if (k == ~0ull || (j > 0 && keys[j - 1] == k)) continue;      // not the head of a column's run
        const int b = (int)(k >> 32);
        const ScanView v = scan_view(pts, offs, b, p.F);
        float* dst = out + (size_t)b * cells * (COMPACT ? p.F - 3 : p.F);
        float m[kFeatMax];
#pragma unroll
        for (int c = 0; c < kFeatMax; ++c) m[c] = 0.0f;
        for (size_t e = j; e < total && keys[e] == k; ++e) {
            const int i = perm[e];
            int col;
            const int lin = cart_lin(p, v.px[i], v.py[i], v.pz[i], col);
#pragma unroll
            for (int c = 0; c < kFeatMax; ++c) {
                if (c >= p.F) break;
                const float f = v.px[i + (size_t)c * v.n];
                if (m[c] < f) {
                    m[c] = f;
                    if (COMPACT) { if (c >= 3) dst[(size_t)(c - 3) * cells + lin] = f; }
                    else dst[(size_t)lin * p.F + c] = f;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
// The reference's sector as a function of quadrant and q = |y|/|x| (kernel.cpp:23-36,64),
// evaluated with the HOST libm exactly as the reference CPU path evaluates it.
int host_sector(int quad, float q, float gap_sector)
{
    const double k = 180 / M_PI;
    const double a = k * atan((double)q);
    float theta;
    switch (quad) {
        case 0: theta = (float)a; break;
        case 1: theta = (float)(180 - a); break;
        case 2: theta = (float)(180 + a); break;
        default: theta = (float)(360 - a); break;
    }
    const double f = floor((double)(theta / gap_sector));
    if (!(f > -1073741824.0 && f < 1073741824.0)) return kDrop;
    return (int)f;
}

inline float bits_to_float(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

int build_sector_lut(mrs_ctx* ctx, int S, mrs::SectorLut& lut)
{
    const float gap = (float)(360.0 / (float)S);
    std::vector<float> thr[4];
    std::vector<int> val[4];
    const uint32_t kInf = 0x7f800000u;
    size_t longest = 0;
    for (int quad = 0; quad < 4; ++quad) {
        uint32_t cur = 0;  // q = +0
        int v = host_sector(quad, 0.0f, gap);
        thr[quad].push_back(0.0f);
        val[quad].push_back(v);
        while (cur < kInf) {
            if (host_sector(quad, bits_to_float(kInf), gap) == v) break;  // constant to +inf
            uint32_t lo = cur, hi = kInf;  // sector(lo) == v, sector(hi) != v (monotone in q)
            while (hi - lo > 1) {
                const uint32_t mid = lo + ((hi - lo) >> 1);
                if (host_sector(quad, bits_to_float(mid), gap) == v) lo = mid; else hi = mid;
            }
            cur = hi;
            v = host_sector(quad, bits_to_float(cur), gap);
            thr[quad].push_back(bits_to_float(cur));
            val[quad].push_back(v);
            if (thr[quad].size() > (size_t)S + 8) {
                mrs::set_error("sector table for num_sector=%d does not converge", S);
                return MRS_ERR_UNSUPPORTED;
            }
        }
        longest = std::max(longest, thr[quad].size());
    }
    lut.stride = (int)longest;
    std::vector<float> ht(4 * longest, INFINITY);
    std::vector<int> hv(4 * longest, kDrop);
    for (int quad = 0; quad < 4; ++quad) {
        lut.cnt[quad] = (int)thr[quad].size();
        for (size_t j = 0; j < thr[quad].size(); ++j) {
            ht[quad * longest + j] = thr[quad][j];
            hv[quad * longest + j] = val[quad][j];
        }
    }
    MRS_HIP_TRY(hipMalloc(&lut.d_thr, ht.size() * sizeof(float)));
    MRS_HIP_TRY(hipMalloc(&lut.d_val, hv.size() * sizeof(int)));
    MRS_HIP_TRY(hipMemcpy(lut.d_thr, ht.data(), ht.size() * sizeof(float), hipMemcpyHostToDevice));
    MRS_HIP_TRY(hipMemcpy(lut.d_val, hv.data(), hv.size() * sizeof(int), hipMemcpyHostToDevice));
    (void)ctx;
    return MRS_OK;
}

int make_polar(mrs_ctx* ctx, const mrs_bev_cfg* c, PolarP& p)
{
    MRS_REQUIRE(c->n0 > 0 && c->n1 > 0 && c->num_height > 0 && c->enough_large > 0, "grid sizes must be positive");
    MRS_REQUIRE(c->max_length > 0 && c->max_height > 0, "max_length / max_height must be positive");
    MRS_REQUIRE((int64_t)c->n0 * c->n1 * c->num_height * c->enough_large < (1ll << 28), "grid too large");
    p.R = c->n0; p.S = c->n1; p.H = c->num_height; p.K = c->enough_large;
    p.gap_ring = (float)c->max_length / (float)c->n0;                       // kernel.cpp:46
    p.gap_sector = (float)(360.0 / (float)c->n1);                           // kernel.cpp:47
    p.gap_height = (float)(2.0 * (float)c->max_height / (float)c->num_height);  // kernel.cpp:48
    p.inv_ring = 1.0f / p.gap_ring;
    p.inv_sector = 1.0f / p.gap_sector;
    p.inv_height = 1.0f / p.gap_height;
    p.mh = (float)c->max_height;
    p.eps_ring = eps_for(p.R);
    p.eps_sector = eps_for(p.S);
    // |(z+mh)*inv - reference quotient| <= ~2.5e-7 * |quotient| (three fp32 roundings each side): 8x margin
    p.eps_height = fmaxf(2e-5f, (float)(2 * p.H + 16) * 2e-6f);
    {
        std::lock_guard<std::mutex> g(ctx->mu);
        auto it = ctx->sector_luts.find(p.S);
        if (it == ctx->sector_luts.end()) {
            mrs::SectorLut lut;
            const int st = build_sector_lut(ctx, p.S, lut);
            if (st != MRS_OK) return st;
            it = ctx->sector_luts.emplace(p.S, lut).first;
        }
        p.thr = it->second.d_thr;
        p.val = it->second.d_val;
        p.stride = it->second.stride;
        for (int q = 0; q < 4; ++q) p.cnt[q] = it->second.cnt[q];
    }
    return MRS_OK;
}

inline int blocks_for(size_t work, int threads, int cap) {
    size_t b = (work + threads - 1) / threads;
    if (b < 1) b = 1;
    if (b > (size_t)cap) b = cap;
    return (int)b;
}

template <class K>
int allow_lds(K kernel, size_t bytes)
{
    if (bytes > 48 * 1024)
        MRS_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    return MRS_OK;
}

int check_batch(mrs_ctx* ctx, const void* a, const void* b, int32_t batch, const mrs_bev_cfg* cfg,
                int32_t layout, const void* out)
{
    MRS_REQUIRE(ctx != nullptr, "ctx");
    MRS_REQUIRE(cfg != nullptr, "cfg");
    MRS_REQUIRE(a != nullptr && b != nullptr && out != nullptr, "null device pointer");
    MRS_REQUIRE(batch > 0, "batch must be positive");
    MRS_REQUIRE(layout == MRS_BEV_OUT_REFERENCE || layout == MRS_BEV_OUT_COMPACT, "unknown out_layout");
    MRS_HIP_TRY(hipSetDevice(ctx->device));
    return MRS_OK;
}

}  // namespace

// ------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------
extern "C" {

int mrs_bev_polar_indices(mrs_ctx* ctx, const float* d_xyz, int32_t n, const mrs_bev_cfg* cfg,
                          int32_t* d_ring, int32_t* d_sector, int32_t* d_height, mrs_stream stream)
{
    MRS_REQUIRE(ctx && cfg && d_xyz && d_ring && d_sector && d_height, "null pointer");
    MRS_REQUIRE(n >= 0, "n must be >= 0");
    MRS_HIP_TRY(hipSetDevice(ctx->device));
    PolarP p;
    int st = make_polar(ctx, cfg, p);
    if (st != MRS_OK) return st;
    if (n == 0) return MRS_OK;
    hipLaunchKernelGGL(k_polar_indices, dim3(blocks_for(n, 256, 4096)), dim3(256), 0, (hipStream_t)stream,
                       d_xyz, n, p, d_ring, d_sector, d_height);
    MRS_HIP_TRY(hipGetLastError());
    return MRS_OK;
}

int mrs_bev_cart_indices(mrs_ctx* ctx, const float* d_xyz, int32_t n, const mrs_bev_cfg* cfg,
                         int32_t* d_ix, int32_t* d_iy, int32_t* d_ih, mrs_stream stream)
{
    MRS_REQUIRE(ctx && cfg && d_xyz && d_ix && d_iy && d_ih, "null pointer");
    MRS_REQUIRE(n >= 0, "n must be >= 0");
    MRS_HIP_TRY(hipSetDevice(ctx->device));
    CartP p;
    int st = make_cart(cfg, false, p);
    if (st != MRS_OK) return st;
    if (n == 0) return MRS_OK;
    hipLaunchKernelGGL(k_cart_indices, dim3(blocks_for(n, 256, 4096)), dim3(256), 0, (hipStream_t)stream,
                       d_xyz, n, p, d_ix, d_iy, d_ih);
    MRS_HIP_TRY(hipGetLastError());
    return MRS_OK;
}

int mrs_bev_polar_batch(mrs_ctx* ctx, const float* d_xyz, const int64_t* d_offsets, int32_t batch,
                        const mrs_bev_cfg* cfg, int32_t layout, float* d_out, mrs_stream stream)
{
    int st = check_batch(ctx, d_xyz, d_offsets, batch, cfg, layout, d_out);
    if (st != MRS_OK) return st;
    PolarP p;
    st = make_polar(ctx, cfg, p);
    if (st != MRS_OK) return st;
    hipStream_t s = (hipStream_t)stream;
    const size_t cells = (size_t)p.R * p.S * p.H;
    size_t lds = 4;
    while (lds * 8 < cells) lds <<= 1;  // power-of-two bitset, see k_polar_lds
    if (layout == MRS_BEV_OUT_COMPACT && p.K == 1 && lds <= ctx->lds_bytes) {
        st = allow_lds(k_polar_lds, lds);
        if (st != MRS_OK) return st;
        hipLaunchKernelGGL(k_polar_lds, dim3(batch), dim3(kWG), lds, s, d_xyz, d_offsets, p, d_out);
        MRS_HIP_TRY(hipGetLastError());
        return MRS_OK;
    }
    // deterministic multi-pass path
    mrs::Scratch kidx;
    const size_t slots = (size_t)batch * p.K * cells;
    st = kidx.alloc(slots * sizeof(int), s);
    if (st != MRS_OK) return st;
    hipLaunchKernelGGL(k_fill_i32, dim3(blocks_for(slots, 256, 4096)), dim3(256), 0, s, kidx.as<int>(), slots, INT_MAX);
    MRS_REQUIRE(batch <= mrs::kMaxGridY, "at most 65535 scans per call in this layout (split the batch)");
    const dim3 pg(512, batch);
    for (int k = 0; k < p.K; ++k)
        hipLaunchKernelGGL(k_polar_kth, pg, dim3(256), 0, s, d_xyz, d_offsets, p, k, kidx.as<int>());
    if (layout == MRS_BEV_OUT_REFERENCE) {
        hipLaunchKernelGGL(k_polar_emit, dim3(blocks_for(cells * p.K, 256, 1024), batch), dim3(256), 0, s,
                           d_xyz, d_offsets, p, kidx.as<int>(), d_out);
    } else {
        mrs::set_error("polar COMPACT layout needs enough_large == 1");
        return MRS_ERR_UNSUPPORTED;
    }
    MRS_HIP_TRY(hipGetLastError());
    return MRS_OK;
}

int mrs_bev_cart_batch(mrs_ctx* ctx, const float* d_xyz, const int64_t* d_offsets, int32_t batch,
                       const mrs_bev_cfg* cfg, int32_t layout, float* d_out, mrs_stream stream)
{
    int st = check_batch(ctx, d_xyz, d_offsets, batch, cfg, layout, d_out);
    if (st != MRS_OK) return st;
    CartP p;
    st = make_cart(cfg, false, p);
    if (st != MRS_OK) return st;
    hipStream_t s = (hipStream_t)stream;
    const size_t cells = (size_t)p.NX * p.NY * p.H;
    if (layout == MRS_BEV_OUT_COMPACT && p.H == 1 && cells * 4 <= ctx->lds_bytes) {
        st = allow_lds(k_cart_lds, cells * 4);
        if (st != MRS_OK) return st;
        hipLaunchKernelGGL(k_cart_lds, dim3(batch), dim3(kWG), cells * 4, s, d_xyz, d_offsets, p, d_out);
        MRS_HIP_TRY(hipGetLastError());
        return MRS_OK;
    }
    const size_t tot = (size_t)batch * cells;
    mrs::Scratch buf;
    const int planes = p.H > 1 ? 3 : 2;
    st = buf.alloc(tot * planes * sizeof(int), s);
    if (st != MRS_OK) return st;
    int* last_idx = buf.as<int>();
    int* zbits = last_idx + tot;
    int* first_pos = p.H > 1 ? zbits + tot : nullptr;
    const int fb = blocks_for(tot, 256, 4096);
    hipLaunchKernelGGL(k_fill_i32, dim3(fb), dim3(256), 0, s, last_idx, tot, -1);
    hipLaunchKernelGGL(k_fill_i32, dim3(fb), dim3(256), 0, s, zbits, tot, 0);
    if (first_pos) hipLaunchKernelGGL(k_fill_i32, dim3(fb), dim3(256), 0, s, first_pos, tot, INT_MAX);
    MRS_REQUIRE(batch <= mrs::kMaxGridY, "at most 65535 scans per call in this layout (split the batch)");
    const dim3 pg(512, batch);
    hipLaunchKernelGGL(k_cart_pass1, pg, dim3(256), 0, s, d_xyz, d_offsets, p, last_idx, zbits, first_pos);
    if (first_pos) hipLaunchKernelGGL(k_cart_pass2, pg, dim3(256), 0, s, d_xyz, d_offsets, p, first_pos, zbits);
    if (layout == MRS_BEV_OUT_REFERENCE)
        hipLaunchKernelGGL(k_cart_emit, dim3(blocks_for(cells, 256, 1024), batch), dim3(256), 0, s, d_xyz,
                           d_offsets, p, last_idx, zbits, d_out);
    else
        hipLaunchKernelGGL(k_copy_i32_as_f32, dim3(fb), dim3(256), 0, s, zbits, d_out, tot);
    MRS_HIP_TRY(hipGetLastError());
    return MRS_OK;
}

int mrs_bev_feat_batch(mrs_ctx* ctx, const float* d_pts, const int64_t* d_offsets, int32_t batch,
                       const mrs_bev_cfg* cfg, int32_t layout, float* d_out, mrs_stream stream)
{
    int st = check_batch(ctx, d_pts, d_offsets, batch, cfg, layout, d_out);
    if (st != MRS_OK) return st;
    CartP p;
    st = make_cart(cfg, true, p);
    if (st != MRS_OK) return st;
    if (layout == MRS_BEV_OUT_COMPACT && p.F <= 3) {
        mrs::set_error("feature BEV COMPACT layout needs featsize > 3");
        return MRS_ERR_ARG;
    }
    hipStream_t s = (hipStream_t)stream;
    if (p.H != 1) {   // order-dependent semantics: sorted columns, one thread per column (see k_feat_columns)
        MRS_REQUIRE(p.F <= kFeatMax, "feature BEV with num_height > 1 supports at most 16 channels");
        MRS_REQUIRE(batch <= mrs::kMaxGridY, "at most 65535 scans per call (split the batch)");
        const size_t cellsH = (size_t)p.NX * p.NY * p.H;
        const size_t totH = (size_t)batch * cellsH * (layout == MRS_BEV_OUT_COMPACT ? p.F - 3 : p.F);
        MRS_HIP_TRY(hipMemsetAsync(d_out, 0, totH * sizeof(float), s));
        int64_t total = 0;
        MRS_HIP_TRY(hipMemcpyAsync(&total, d_offsets + batch, sizeof(int64_t), hipMemcpyDeviceToHost, s));
        MRS_HIP_TRY(hipStreamSynchronize(s));
        if (total == 0) return MRS_OK;
        MRS_REQUIRE(total < (1ll << 31), "more than 2^31 points in one call");
        mrs::Scratch k_in, k_out, v_in, v_out, tmp;
        if ((st = k_in.alloc((size_t)total * 8, s)) != MRS_OK) return st;
        if ((st = k_out.alloc((size_t)total * 8, s)) != MRS_OK) return st;
        if ((st = v_in.alloc((size_t)total * 4, s)) != MRS_OK) return st;
        if ((st = v_out.alloc((size_t)total * 4, s)) != MRS_OK) return st;
        hipLaunchKernelGGL(k_feat_keys, dim3(512, batch), dim3(256), 0, s, d_pts, d_offsets, p, k_in.as<unsigned long long>(), v_in.as<int>());
        size_t bytes = 0;
        MRS_HIP_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, k_in.as<unsigned long long>(), k_out.as<unsigned long long>(),
                                                       v_in.as<int>(), v_out.as<int>(), (int)total, 0, 64, s));
        if ((st = tmp.alloc(bytes, s)) != MRS_OK) return st;
        MRS_HIP_TRY(hipcub::DeviceRadixSort::SortPairs(tmp.p, bytes, k_in.as<unsigned long long>(), k_out.as<unsigned long long>(),
                                                       v_in.as<int>(), v_out.as<int>(), (int)total, 0, 64, s));
        const int fb = blocks_for((size_t)total, 256, 4096);
        if (layout == MRS_BEV_OUT_COMPACT)
            hipLaunchKernelGGL(k_feat_columns<true>, dim3(fb), dim3(256), 0, s, d_pts, d_offsets, p, k_out.as<unsigned long long>(),
                               v_out.as<int>(), (size_t)total, d_out);
        else
            hipLaunchKernelGGL(k_feat_columns<false>, dim3(fb), dim3(256), 0, s, d_pts, d_offsets, p, k_out.as<unsigned long long>(),
                               v_out.as<int>(), (size_t)total, d_out);
        MRS_HIP_TRY(hipGetLastError());
        return MRS_OK;
    }
    const size_t cells = (size_t)p.NX * p.NY;
    const size_t tot = (size_t)batch * cells * (layout == MRS_BEV_OUT_COMPACT ? p.F - 3 : p.F);
    if (layout == MRS_BEV_OUT_COMPACT && p.F > 3 && cells * sizeof(int) <= std::min<size_t>(ctx->lds_bytes, 96 * 1024) && batch <= mrs::kMaxGridY) {
        const size_t lds = cells * sizeof(int);
        if (lds > 48 * 1024)
            MRS_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_feat_lds), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(k_feat_lds, dim3(p.F - 3, batch), dim3(1024), lds, s, d_pts, d_offsets, p, d_out);
        MRS_HIP_TRY(hipGetLastError());
        return MRS_OK;
    }
    // positive floats order like ints: accumulate straight into the output buffer
    MRS_HIP_TRY(hipMemsetAsync(d_out, 0, tot * sizeof(float), s));
    MRS_REQUIRE(batch <= mrs::kMaxGridY, "at most 65535 scans per call in this layout (split the batch)");
    const dim3 pg(512, batch);
    if (layout == MRS_BEV_OUT_COMPACT)
        hipLaunchKernelGGL(k_feat_max<true>, pg, dim3(256), 0, s, d_pts, d_offsets, p, reinterpret_cast<int*>(d_out));
    else
        hipLaunchKernelGGL(k_feat_max<false>, pg, dim3(256), 0, s, d_pts, d_offsets, p, reinterpret_cast<int*>(d_out));
    MRS_HIP_TRY(hipGetLastError());
    return MRS_OK;
}

// ---- host-buffer forms (reference calling convention) -------------------------------------
// One scan per call, host arrays in and out: what a rospy callback does.  Latency matters here, so the device buffers come
// from the library's scratch cache and every caller thread keeps one non-blocking stream for the life of the thread
// (callbacks of different subscriptions run on different threads and therefore overlap on the GPU).
static hipStream_t thread_stream(int device)   // the calling thread's stream on `device` (current device already set)
{
    static thread_local std::map<int, hipStream_t> streams;
    auto it = streams.find(device);
    if (it != streams.end()) return it->second;
    hipStream_t s = nullptr;
    if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) return nullptr;
    streams[device] = s;
    return s;
}

static int bev_host(mrs_ctx* ctx, const float* h_in, int32_t n, int planes, const mrs_bev_cfg* cfg,
                    float* h_out, size_t out_floats, int which)
{
    MRS_REQUIRE(ctx && cfg && h_in && h_out, "null pointer");
    MRS_REQUIRE(n >= 0, "n must be >= 0");
    MRS_HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t s = thread_stream(ctx->device);
    if (!s) { mrs::set_error("could not create the caller thread's stream"); return MRS_ERR_HIP; }
    mrs::Scratch in, out, off;
    int st;
    if ((st = in.alloc((size_t)(n ? n : 1) * planes * sizeof(float), s)) != MRS_OK) return st;
    if ((st = out.alloc(out_floats * sizeof(float), s)) != MRS_OK) return st;
    if ((st = off.alloc(2 * sizeof(int64_t), s)) != MRS_OK) return st;
    const int64_t offs[2] = {0, n};
    MRS_HIP_TRY(hipMemcpyAsync(in.p, h_in, (size_t)n * planes * sizeof(float), hipMemcpyHostToDevice, s));
    MRS_HIP_TRY(hipMemcpyAsync(off.p, offs, sizeof(offs), hipMemcpyHostToDevice, s));
    if (which == 0) st = mrs_bev_polar_batch(ctx, in.as<float>(), off.as<int64_t>(), 1, cfg, MRS_BEV_OUT_REFERENCE, out.as<float>(), s);
    else if (which == 1) st = mrs_bev_cart_batch(ctx, in.as<float>(), off.as<int64_t>(), 1, cfg, MRS_BEV_OUT_REFERENCE, out.as<float>(), s);
    else st = mrs_bev_feat_batch(ctx, in.as<float>(), off.as<int64_t>(), 1, cfg, MRS_BEV_OUT_REFERENCE, out.as<float>(), s);
    if (st != MRS_OK) { (void)hipStreamSynchronize(s); return st; }
    if (hipMemcpyAsync(h_out, out.p, out_floats * sizeof(float), hipMemcpyDeviceToHost, s) != hipSuccess ||
        hipStreamSynchronize(s) != hipSuccess) {
        mrs::set_error("D2H copy / synchronise failed: %s", hipGetErrorString(hipGetLastError()));
        return MRS_ERR_HIP;
    }
    return MRS_OK;
}

int mrs_bev_polar_host(mrs_ctx* ctx, const float* h_xyz, int32_t n, const mrs_bev_cfg* cfg, float* h_out)
{
    MRS_REQUIRE(cfg, "cfg");
    return bev_host(ctx, h_xyz, n, 3, cfg, h_out,
                    (size_t)3 * cfg->n0 * cfg->n1 * cfg->num_height * cfg->enough_large, 0);
}

int mrs_bev_cart_host(mrs_ctx* ctx, const float* h_xyz, int32_t n, const mrs_bev_cfg* cfg, float* h_out)
{
    MRS_REQUIRE(cfg, "cfg");
    return bev_host(ctx, h_xyz, n, 3, cfg, h_out, (size_t)3 * cfg->n0 * cfg->n1 * cfg->num_height, 1);
}

int mrs_bev_feat_host(mrs_ctx* ctx, const float* h_pts, int32_t n, const mrs_bev_cfg* cfg, float* h_out)
{
    MRS_REQUIRE(cfg, "cfg");
    return bev_host(ctx, h_pts, n, cfg->enough_large, cfg, h_out,
                    (size_t)cfg->n0 * cfg->n1 * cfg->num_height * cfg->enough_large, 2);
}

}  // extern "C"
