// radon.hip -- parallel-beam Radon forward projection for gfx950.
//
// Reference behaviour reproduced (never copied):
//   torch-radon/src/forward.cu:12-124 (radon_forward_kernel<true,1,float>), texture semantics of
//   src/texture.cu:133-143, API torch_radon/radon.py:62-87,139-167; the global mean / unbiased
//   std normalisation of RING_ros/util.py:197 is fused in as an optional second output.
//
// Design: CDNA has no texture sampler, so the image lives in LDS with a 2-texel zero border
// (border address mode) and an odd row stride; one workgroup per image, one ray per lane with
// lanes = consecutive detectors.  The geometry of a ray does not depend on the image: it is evaluated
// once per plan on the host into a table (start line, minor coordinate, slope, step count, norm) that
// every workgroup streams from L2.  Samples are aligned exactly to texel centres along the ray's
// dominant axis (integer stepping, always towards increasing index), 2-tap fp32 interpolation along
// the minor axis: 7 VALU + one ds_read2_b32 per sample, the two taps accumulated by one packed FMA.
// HBM traffic is 4*H*W in + 4*A*D out per image; the kernels are bound by VALU issue and the LDS array, not by HBM.
// Batches run two images per workgroup (k_radon2 / march2 below: the two images share the per-sample index chain;
// VALU issue 76 %, LDS array 65 % busy of which 41 % bank conflicts: adjacent rays are 1-1.41 texels apart).
// Numerics are shared with oracle/radon_oracle.c op for op (cos/sin evaluated on the host in
// double when the plan is built), so HIP == oracle bit for bit.
#include <algorithm>
#include <cmath>
#include <vector>

#include "common.hpp"
#include "radon_device.hpp"

namespace {


struct HostRay {
    int n_steps, ydom, major;
    float q, vm, n;
};

// forward.cu:32-112 for one ray (a, r); samples are taken along INCREASING dominant-axis index
// (a ray running the other way is entered at its last sample).
HostRay ray_setup(int H, int W, float cs, float sn, int r, int det, float spacing, float L)
{
    HostRay o = {0, 0, 0, 0.0f, 0.0f, 0.0f};
    const float sx = ((float)r - (float)det * 0.5f + 0.5f) * spacing;
    const float sy = L, ex = sx, ey = -L;
    float rsx = sx * cs + sy * sn;
    float rsy = -sx * sn + sy * cs;
    float rdx = ex * cs + ey * sn - rsx;
    float rdy = -ex * sn + ey * cs - rsy;
    rsx = rsx - (-0.5f * (float)W);
    rsy = rsy - (-0.5f * (float)H);
    const float dx = rdx >= 0 ? fmaxf(rdx, 1e-6f) : fminf(rdx, -1e-6f);
    const float dy = rdy >= 0 ? fmaxf(rdy, 1e-6f) : fminf(rdy, -1e-6f);
    const float axm = (-rsx) / dx, axp = ((float)W - rsx) / dx;
    const float aym = (-rsy) / dy, ayp = ((float)H - rsy) / dy;
    const float as = fmaxf(fminf(axp, axm), fminf(ayp, aym));
    const float ae = fminf(fmaxf(axp, axm), fmaxf(ayp, aym));
    if ((double)as > (double)ae - 1e-6) return o;
    rsx += rdx * as;
    rsy += rdy * as;
    rdx *= (ae - as);
    rdy *= (ae - as);
    const float m = fmaxf(fabsf(rdx), fabsf(rdy));
    const int n_steps = (int)rintf(m);
    const float vx = rdx / m, vy = rdy / m;
    o.n = sqrtf(vx * vx + vy * vy);
    const bool ydom = fabsf(rdy) >= fabsf(rdx);
    float step;
    if (ydom) {
        const float inc = 0.5f - rsy + rintf(rsy);
        step = inc / vy;
        step += (vy < 0) ? 1.0f : 0.0f;
    } else {
        const float inc = 0.5f - rsx + rintf(rsx);
        step = inc / vx;
        step += (vx < 0) ? 1.0f : 0.0f;
    }
    rsx += step * vx;
    rsy += step * vy;
    o.ydom = ydom ? 1 : 0;
    o.major = (int)floorf(ydom ? rsy : rsx);
    o.q = (ydom ? rsx : rsy) + 1.5f;
    o.vm = ydom ? vx : vy;
    o.n_steps = n_steps;
    if (n_steps > 0 && (ydom ? vy : vx) < 0) {
        o.major -= n_steps - 1;
        o.q = fmaf((float)(n_steps - 1), o.vm, o.q);
        o.vm = -o.vm;
    }
    return o;
}

// byte offset of minor index i: a shift when the minor axis is x, a full-rate 24-bit multiply (not the
// quarter-rate v_mul_lo_u32) when it is y
template <bool YDOM>
__device__ __forceinline__ int minor_bytes(int i, int unit)
{
    return YDOM ? i * 4 : __mul24(i, unit);
}

typedef const __attribute__((address_space(3))) float* lds_fptr;

__device__ __forceinline__ float lds_at(unsigned addr) { return *(lds_fptr)(uintptr_t)addr; }

// Sample loop of one ray.  off: absolute LDS byte address of the first sample's texel line; the line
// advances by one row (YDOM) / one column (!YDOM) per sample, the two taps sit one column / one row apart.
// Per sample: q += vm, fract, cvt, address shift-add (or 24-bit mad), 1 - fr, one packed FMA
// (t0,t1)*(1-fr,fr) -> two running sums; both taps in one ds_read2_b32 whose immediate offsets absorb
// the line advance of the unrolled samples.  The chain of each running sum is strictly sequential
// (matches the oracle bit for bit).
// GLOBAL: the padded image sits in global memory (images too large for the LDS); off is then a byte offset from gbase.
template <bool YDOM, int STRIDE, bool GLOBAL = false>
__device__ __forceinline__ float march(unsigned off, float q, float vm, int n_steps, int rstride, const char* gbase = nullptr)
{
    auto tap = [&](unsigned a) { return GLOBAL ? *reinterpret_cast<const float*>(gbase + a) : lds_at(a); };
    const int stride = STRIDE > 0 ? STRIDE : rstride;
    const int unit = (YDOM ? 1 : stride) * 4;   // bytes between the two taps == bytes per minor index
    const int lstep = (YDOM ? stride : 1) * 4;  // bytes per sample along the dominant axis
    constexpr int U = 6;  // samples in flight; ds_read2_b32 offsets are 8-bit dword counts (2*125+1 fits), so
                          // the unrolled samples hang off two line addresses, three immediates each
    v2f acc = {0.0f, 0.0f};
    int j = 0;
    unsigned off2 = off + 3 * lstep;
#pragma nounroll
    for (; j + U <= n_steps; j += U) {
        v2f w[U], t[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const float fr = __builtin_amdgcn_fractf(q);
            const unsigned a = (u < 3 ? off : off2) + (unsigned)minor_bytes<YDOM>((int)q, unit);
            t[u].x = tap(a + (u % 3) * lstep);
            t[u].y = tap(a + (u % 3) * lstep + unit);
            w[u].x = 1.0f - fr;
            w[u].y = fr;
            q += vm;
        }
        off += U * lstep;
        off2 += U * lstep;
#pragma unroll
        for (int u = 0; u < U; ++u) acc = __builtin_elementwise_fma(t[u], w[u], acc);
    }
    for (; j < n_steps; ++j) {
        const float fr = __builtin_amdgcn_fractf(q);
        const unsigned a = off + (unsigned)minor_bytes<YDOM>((int)q, unit);
        v2f t, w;
        t.x = tap(a);
        t.y = tap(a + unit);
        w.x = 1.0f - fr;
        w.y = fr;
        acc = __builtin_elementwise_fma(t, w, acc);
        q += vm;
        off += lstep;
    }
    return acc.x + acc.y;
}

template <int STRIDE>
__device__ __forceinline__ float trace_ray(const float* img, const RadonP& p, int ray)
{
    const int meta = p.meta[ray];
    const int n_steps = meta & 0xffff;
    if (n_steps == 0) return 0.0f;
    const int off = p.base[ray];
    const float q = p.q[ray], vm = p.vm[ray], n = p.nrm[ray];
    const unsigned tile = (unsigned)(uintptr_t)(lds_cptr)reinterpret_cast<const char*>(img) + (unsigned)off;
    // the two orientations get their own loop (wave-uniform branch except for the one wave in 15 that
    // straddles the 45-degree switch)
    const float acc = (meta >> 16) ? march<true, STRIDE>(tile, q, vm, n_steps, p.stride)
                                   : march<false, STRIDE>(tile, q, vm, n_steps, p.stride);
    return acc * n;
}

// the same ray against a padded image in global memory (generic row stride)
__device__ __forceinline__ float trace_ray_global(const float* padded, const RadonP& p, int ray)
{
    const int meta = p.meta[ray];
    const int n_steps = meta & 0xffff;
    if (n_steps == 0) return 0.0f;
    const char* g = reinterpret_cast<const char*>(padded);
    const float acc = (meta >> 16) ? march<true, 0, true>((unsigned)p.base[ray], p.q[ray], p.vm[ray], n_steps, p.stride, g)
                                   : march<false, 0, true>((unsigned)p.base[ray], p.q[ray], p.vm[ray], n_steps, p.stride, g);
    return acc * p.nrm[ray];
}

// One workgroup per image.  sino_raw / sino_norm may each be null.
// sino_norm = (S - mean(S)) / std(S) with the unbiased std over the whole sinogram
// (util.py:197: fn.normalize(pc_RING, mean=pc_RING.mean(), std=pc_RING.std())).
template <int MAX_RAYS_PER_LANE, int STRIDE>
__global__ __launch_bounds__(kRadonWG) void k_radon(const float* __restrict__ img, RadonP p,
                                                    float* __restrict__ sino_raw,
                                                    float* __restrict__ sino_norm, int* __restrict__ degenerate)
{
    extern __shared__ __attribute__((aligned(16))) float tile[];
    __shared__ double red[2][16];
    const int b = blockIdx.x;
    const float* src = img + (size_t)b * p.H * p.W;
    const int rows = p.H + 2 * kPad;
    for (int i = threadIdx.x; i < rows * p.stride; i += kRadonWG) tile[i] = 0.0f;
    __syncthreads();
    for (int i = threadIdx.x; i < p.H * p.W; i += kRadonWG) {
        const int y = i / p.W, x = i - y * p.W;
        tile[(y + kPad) * p.stride + x + kPad] = src[i];
    }
    __syncthreads();

    const int rays = p.A * p.D;
    float val[MAX_RAYS_PER_LANE];
#pragma unroll
    for (int k = 0; k < MAX_RAYS_PER_LANE; ++k) {
        const int ray = threadIdx.x + k * kRadonWG;
        float v = 0.0f;
        if (ray < rays) {
            v = trace_ray<STRIDE>(tile, p, ray);
            if (sino_raw) sino_raw[(size_t)b * rays + ray] = v;
        }
        val[k] = v;
    }
    if (!sino_norm) return;
    normalize_store<MAX_RAYS_PER_LANE>(val, rays, red, sino_norm + (size_t)b * rays, degenerate);
}

// Few images (a rospy callback hands over ONE): grid = (slices of 1024 rays, images), every workgroup stages the image and each
// lane traces a single ray, so that one image occupies 15 workgroups instead of one; k_normalize (same lane <-> ray mapping and
// reduction order as normalize_store: identical bits) follows.
template <int STRIDE>
__global__ __launch_bounds__(kRadonWG) void k_radon_split(const float* __restrict__ img, RadonP p, float* __restrict__ sino_raw)
{
    extern __shared__ __attribute__((aligned(16))) float tile[];
    const int b = blockIdx.y;
    const float* src = img + (size_t)b * p.H * p.W;
    const int rows = p.H + 2 * kPad;
    for (int i = threadIdx.x; i < rows * p.stride; i += kRadonWG) tile[i] = 0.0f;
    __syncthreads();
    for (int i = threadIdx.x; i < p.H * p.W; i += kRadonWG) {
        const int y = i / p.W, x = i - y * p.W;
        tile[(y + kPad) * p.stride + x + kPad] = src[i];
    }
    __syncthreads();
    const int rays = p.A * p.D;
    const int ray = blockIdx.x * kRadonWG + threadIdx.x;
    if (ray < rays) sino_raw[(size_t)b * rays + ray] = trace_ray<STRIDE>(tile, p, ray);
}

// Two images per workgroup (see march2).  grid = ceil(batch / 2); an odd batch leaves the last workgroup's second slot
// empty (zero image, nothing stored).
template <int MAX_RAYS_PER_LANE, int STRIDE>
__global__ __launch_bounds__(kRadonWG) void k_radon2(const float* __restrict__ img, RadonP p, int batch,
                                                     float* __restrict__ sino_raw, float* __restrict__ sino_norm,
                                                     int* __restrict__ degenerate)
{
    extern __shared__ __attribute__((aligned(16))) v2f cells[];
    __shared__ double red[2][16];
    const int b0 = 2 * blockIdx.x, b1 = b0 + 1;
    const bool two = b1 < batch;
    const float* srcA = img + (size_t)b0 * p.H * p.W;
    const float* srcB = img + (size_t)(two ? b1 : b0) * p.H * p.W;
    const int rows = p.H + 2 * kPad;
    const v2f zero = {0.0f, 0.0f};
    for (int i = threadIdx.x; i < rows * p.stride; i += kRadonWG) cells[i] = zero;
    __syncthreads();
    for (int i = threadIdx.x; i < p.H * p.W; i += kRadonWG) {
        const int y = i / p.W, x = i - y * p.W;
        const v2f c = {srcA[i], two ? srcB[i] : 0.0f};
        cells[(y + kPad) * p.stride + x + kPad] = c;
    }
    __syncthreads();

    const int rays = p.A * p.D;
    float va[MAX_RAYS_PER_LANE], vb[MAX_RAYS_PER_LANE];
#pragma unroll
    for (int k = 0; k < MAX_RAYS_PER_LANE; ++k) {
        const int ray = threadIdx.x + k * kRadonWG;
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
    if (!sino_norm) return;
    normalize_store<MAX_RAYS_PER_LANE>(va, rays, red, sino_norm + (size_t)b0 * rays, degenerate);
    if (two) normalize_store<MAX_RAYS_PER_LANE>(vb, rays, red, sino_norm + (size_t)b1 * rays, degenerate);
}

// generic fallback for sinograms with more than 16 rays per lane: no register residency,
// raw output only (normalisation then runs as its own kernel)
__global__ __launch_bounds__(kRadonWG) void k_radon_big(const float* __restrict__ img, RadonP p,
                                                        float* __restrict__ sino_raw)
{
    extern __shared__ __attribute__((aligned(16))) float tile[];
    const int b = blockIdx.x;
    const float* src = img + (size_t)b * p.H * p.W;
    const int rows = p.H + 2 * kPad;
    for (int i = threadIdx.x; i < rows * p.stride; i += kRadonWG) tile[i] = 0.0f;
    __syncthreads();
    for (int i = threadIdx.x; i < p.H * p.W; i += kRadonWG) {
        const int y = i / p.W, x = i - y * p.W;
        tile[(y + kPad) * p.stride + x + kPad] = src[i];
    }
    __syncthreads();
    const int rays = p.A * p.D;
    for (int ray = threadIdx.x; ray < rays; ray += kRadonWG)
        sino_raw[(size_t)b * rays + ray] = trace_ray<0>(tile, p, ray);
}

// Images that do not fit the LDS (torch_radon.ParallelBeam takes any size; MR_SLAM itself only uses 120 x 120):
// k_radon_pad builds the zero-bordered copy in global memory, k_radon_global traces the rays against it through L2.
__global__ void k_radon_pad(const float* __restrict__ img, RadonP p, float* __restrict__ padded)
{
    const int b = blockIdx.y;
    const size_t plane = (size_t)(p.H + 2 * kPad) * p.stride;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < p.H * p.W; i += gridDim.x * blockDim.x) {
        const int y = i / p.W, x = i - y * p.W;
        padded[b * plane + (size_t)(y + kPad) * p.stride + x + kPad] = img[(size_t)b * p.H * p.W + i];
    }
}

__global__ void k_radon_global(const float* __restrict__ padded, RadonP p, float* __restrict__ sino_raw)
{
    const int b = blockIdx.y;
    const size_t plane = (size_t)(p.H + 2 * kPad) * p.stride;
    const int rays = p.A * p.D;
    for (int ray = blockIdx.x * blockDim.x + threadIdx.x; ray < rays; ray += gridDim.x * blockDim.x)
        sino_raw[(size_t)b * rays + ray] = trace_ray_global(padded + b * plane, p, ray);
}

// (x - mean) / std over `group` consecutive floats per block (unbiased std), in place or not.
// util.py:339-340 (RING++ normalises a whole [C,H,W] descriptor with one mean/std).
__global__ __launch_bounds__(1024) void k_normalize(const float* __restrict__ in, float* __restrict__ out, int group,
                                                    int* __restrict__ degenerate)
{
    __shared__ double red[2][16];
    const float* src = in + (size_t)blockIdx.x * group;
    float* dst = out + (size_t)blockIdx.x * group;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    double s1 = 0.0;
    for (int i = threadIdx.x; i < group; i += 1024) s1 += (double)src[i];
    s1 = wave_sum(s1);
    if (lane == 0) red[0][wave] = s1;
    __syncthreads();
    double tot = 0.0;
    for (int w = 0; w < 16; ++w) tot += red[0][w];
    const double mean_d = tot / (double)group;
    double s2 = 0.0;
    for (int i = threadIdx.x; i < group; i += 1024) {
        const double d = (double)src[i] - mean_d;
        s2 += d * d;
    }
    s2 = wave_sum(s2);
    if (lane == 0) red[1][wave] = s2;
    __syncthreads();
    double tot2 = 0.0;
    for (int w = 0; w < 16; ++w) tot2 += red[1][w];
    const float mean = (float)mean_d;
    const float sd = (float)sqrt(tot2 / (double)(group - 1));
    const bool ok = sd > 0.0f && sd < INFINITY;   // constant group: zeros instead of NaN (torchvision raises there)
    if (!ok && threadIdx.x == 0 && degenerate) atomicAdd(degenerate, 1);
    for (int i = threadIdx.x; i < group; i += 1024) dst[i] = ok ? (src[i] - mean) / sd : 0.0f;
}

}  // namespace

extern "C" {

int mrs_radon_plan_create(mrs_ctx* ctx, const float* h_angles, int32_t n_angles, int32_t det_count,
                          float det_spacing, int32_t height, int32_t width, mrs_radon_plan** out_plan)
{
    MRS_REQUIRE(ctx && h_angles && out_plan, "null pointer");
    MRS_REQUIRE(n_angles > 0 && det_count > 0 && height > 0 && width > 0, "sizes must be positive");
    MRS_REQUIRE(det_spacing > 0.0f, "det_spacing must be positive");
    *out_plan = nullptr;
    const int stride = (width + 2 * kPad) | 1;
    const size_t lds = (size_t)(height + 2 * kPad) * stride * sizeof(float);
    MRS_REQUIRE((size_t)(height + 2 * kPad) * stride * sizeof(float) < (1ull << 31), "image too large");
    MRS_HIP_TRY(hipSetDevice(ctx->device));
    MRS_REQUIRE(height + width < 60000, "image too large for the 16-bit step count");
    const size_t rays = (size_t)n_angles * det_count;
    const float L = sqrtf((width * 0.5f) * (width * 0.5f) + (height * 0.5f) * (height * 0.5f));  // forward.cu:33
    std::vector<int> tab(5 * rays);
    int* meta = tab.data();
    int* base = meta + rays;
    float* q = reinterpret_cast<float*>(base + rays);
    float* vm = q + rays;
    float* nrm = vm + rays;
    for (int a = 0; a < n_angles; ++a) {
        const float cs = (float)cos((double)h_angles[a]);
        const float sn = (float)sin((double)h_angles[a]);
        for (int r = 0; r < det_count; ++r) {
            const HostRay g = ray_setup(height, width, cs, sn, r, det_count, det_spacing, L);
            const size_t k = (size_t)a * det_count + r;
            meta[k] = g.n_steps | (g.ydom << 16);
            base[k] = 4 * (g.ydom ? (g.major + kPad) * stride : g.major + kPad);
            q[k] = g.q; vm[k] = g.vm; nrm[k] = g.n;
            if (g.n_steps > 0) {  // every tap must stay inside the zero border
                const int last = g.major + g.n_steps - 1, lim = g.ydom ? height : width, mlim = g.ydom ? width : height;
                const float q_end = g.q + (float)(g.n_steps - 1) * g.vm;
                if (g.major < -kPad || last >= lim + kPad || fminf(g.q, q_end) < 0.0f ||
                    fmaxf(g.q, q_end) >= (float)(mlim + 2 * kPad - 1)) {
                    mrs::set_error("Radon ray (%d,%d) leaves the %d-texel border", a, r, kPad);
                    return MRS_ERR_UNSUPPORTED;
                }
            }
        }
    }
    mrs_radon_plan* pl = new mrs_radon_plan();
    pl->ctx = ctx;
    pl->n_angles = n_angles; pl->det = det_count; pl->H = height; pl->W = width;
    pl->spacing = det_spacing;
    pl->in_lds = lds <= ctx->lds_bytes;
    pl->two_in_lds = 2 * lds + 1024 <= ctx->lds_bytes;
    if (hipMalloc(&pl->d_degenerate, sizeof(int)) != hipSuccess || hipMemset(pl->d_degenerate, 0, sizeof(int)) != hipSuccess ||
        hipMalloc(&pl->d_meta, tab.size() * sizeof(int)) != hipSuccess ||
        hipMemcpy(pl->d_meta, tab.data(), tab.size() * sizeof(int), hipMemcpyHostToDevice) != hipSuccess) {
        mrs::set_error("could not upload the ray table");
        if (pl->d_meta) (void)hipFree(pl->d_meta);
        if (pl->d_degenerate) (void)hipFree(pl->d_degenerate);
        delete pl;
        return MRS_ERR_HIP;
    }
    // slot tables (two-image kernels): rays sorted by (orientation, step count, ray id), 64 consecutive slots per wave and round
    const int per_lane = (int)((rays + kRadonWG - 1) / kRadonWG);
    if (pl->two_in_lds && per_lane <= 16) {
        std::vector<int> order(rays);
        for (size_t i = 0; i < rays; ++i) order[i] = (int)i;
        std::stable_sort(order.begin(), order.end(), [&](int a, int b) {
            const int ya = meta[a] >> 16, yb = meta[b] >> 16, na = meta[a] & 0xffff, nb = meta[b] & 0xffff;
            return ya != yb ? ya > yb : na > nb;
        });
        const size_t slots = (size_t)per_lane * kRadonWG;
        std::vector<int4> h_slot(slots, make_int4(0, 0, 0, 0));
        std::vector<float> h_nrm(slots, 0.0f);
        std::vector<int> h_ray(slots, -1);
        for (size_t i = 0; i < rays; ++i) {
            const int r = order[i];
            int qb, vb;
            memcpy(&qb, &q[r], 4); memcpy(&vb, &vm[r], 4);
            h_slot[i] = make_int4(meta[r], base[r], qb, vb);
            h_nrm[i] = nrm[r];
            h_ray[i] = r;
        }
        if (hipMalloc(&pl->d_slot, slots * sizeof(int4)) != hipSuccess || hipMalloc(&pl->d_slot_nrm, slots * sizeof(float)) != hipSuccess ||
            hipMalloc(&pl->d_slot_ray, slots * sizeof(int)) != hipSuccess ||
            hipMemcpy(pl->d_slot, h_slot.data(), slots * sizeof(int4), hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(pl->d_slot_nrm, h_nrm.data(), slots * sizeof(float), hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(pl->d_slot_ray, h_ray.data(), slots * sizeof(int), hipMemcpyHostToDevice) != hipSuccess) {
            mrs::set_error("could not upload the slot tables");
            mrs_radon_plan_destroy(pl);
            return MRS_ERR_HIP;
        }
        pl->slot_per_lane = per_lane;
    }
    *out_plan = pl;
    return MRS_OK;
}

int mrs_radon_plan_destroy(mrs_radon_plan* plan)
{
    if (!plan) return MRS_OK;
    (void)hipSetDevice(plan->ctx->device);
    if (plan->d_slot) (void)hipFree(plan->d_slot);
    if (plan->d_slot_nrm) (void)hipFree(plan->d_slot_nrm);
    if (plan->d_slot_ray) (void)hipFree(plan->d_slot_ray);
    if (plan->d_meta) (void)hipFree(plan->d_meta);
    if (plan->d_degenerate) (void)hipFree(plan->d_degenerate);
    delete plan;
    return MRS_OK;
}

int mrs_radon_plan_degenerate_count(mrs_radon_plan* plan, int32_t reset, int32_t* out_count)
{
    MRS_REQUIRE(plan && out_count, "null pointer");
    MRS_HIP_TRY(hipSetDevice(plan->ctx->device));
    MRS_HIP_TRY(hipDeviceSynchronize());
    int v = 0;
    MRS_HIP_TRY(hipMemcpy(&v, plan->d_degenerate, sizeof(int), hipMemcpyDeviceToHost));
    if (reset) MRS_HIP_TRY(hipMemset(plan->d_degenerate, 0, sizeof(int)));
    *out_count = v;
    return MRS_OK;
}

int mrs_normalize_groups(mrs_ctx* ctx, const float* d_in, float* d_out, int32_t n_groups, int32_t group_len,
                         mrs_stream stream)
{
    MRS_REQUIRE(ctx && d_in && d_out, "null pointer");
    MRS_REQUIRE(n_groups > 0 && group_len > 1, "need n_groups > 0 and group_len > 1");
    MRS_HIP_TRY(hipSetDevice(ctx->device));
    hipLaunchKernelGGL(k_normalize, dim3(n_groups), dim3(1024), 0, (hipStream_t)stream, d_in, d_out, group_len, (int*)nullptr);
    MRS_HIP_TRY(hipGetLastError());
    return MRS_OK;
}

int mrs_radon_forward(mrs_radon_plan* plan, const float* d_img, int32_t batch, float* d_sino,
                      float* d_sino_norm, mrs_stream stream)
{
    MRS_REQUIRE(plan && d_img, "null pointer");
    MRS_REQUIRE(d_sino || d_sino_norm, "at least one output required");
    MRS_REQUIRE(batch > 0, "batch must be positive");
    MRS_HIP_TRY(hipSetDevice(plan->ctx->device));
    RadonP p;
    p.A = plan->n_angles; p.D = plan->det; p.H = plan->H; p.W = plan->W;
    p.stride = (plan->W + 2 * kPad) | 1;
    const size_t nr = (size_t)p.A * p.D;
    p.meta = plan->d_meta;
    p.base = plan->d_meta + nr;
    p.q = reinterpret_cast<const float*>(plan->d_meta + 2 * nr);
    p.vm = p.q + nr;
    p.nrm = p.vm + nr;
    const size_t lds = (size_t)(p.H + 2 * kPad) * p.stride * sizeof(float);
    hipStream_t s = (hipStream_t)stream;
    const int rays = p.A * p.D;
    const int per_lane = (rays + kRadonWG - 1) / kRadonWG;
    if (!plan->in_lds) {
        mrs::Scratch padded, tmp;
        const size_t plane = (size_t)(p.H + 2 * kPad) * p.stride;
        int st = padded.alloc((size_t)batch * plane * sizeof(float), s);
        if (st != MRS_OK) return st;
        MRS_HIP_TRY(hipMemsetAsync(padded.p, 0, (size_t)batch * plane * sizeof(float), s));
        float* raw = d_sino;
        if (!raw) {
            if ((st = tmp.alloc((size_t)batch * rays * sizeof(float), s)) != MRS_OK) return st;
            raw = tmp.as<float>();
        }
        hipLaunchKernelGGL(k_radon_pad, dim3(std::min((p.H * p.W + 255) / 256, 1024), batch), dim3(256), 0, s, d_img, p, padded.as<float>());
        hipLaunchKernelGGL(k_radon_global, dim3(std::min((rays + 255) / 256, 4096), batch), dim3(256), 0, s, padded.as<float>(), p, raw);
        if (d_sino_norm) hipLaunchKernelGGL(k_normalize, dim3(batch), dim3(1024), 0, s, raw, d_sino_norm, rays, plan->d_degenerate);
        MRS_HIP_TRY(hipGetLastError());
        return MRS_OK;
    }
    if (per_lane <= 16 && batch <= 128) {
        // latency path: spread every image over ceil(rays / 1024) workgroups
        mrs::Scratch tmp;
        float* raw = d_sino;
        if (!raw) {
            int st = tmp.alloc((size_t)batch * rays * sizeof(float), s);
            if (st != MRS_OK) return st;
            raw = tmp.as<float>();
        }
        auto kern = p.stride == 125 ? k_radon_split<125> : k_radon_split<0>;
        if (lds > 48 * 1024)
            MRS_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(kern, dim3(per_lane, batch), dim3(kRadonWG), lds, s, d_img, p, raw);
        if (d_sino_norm) hipLaunchKernelGGL(k_normalize, dim3(batch), dim3(1024), 0, s, raw, d_sino_norm, rays, plan->d_degenerate);
    } else if (per_lane <= 16 && plan->two_in_lds && batch > 1) {
        // two images per workgroup share the per-sample index arithmetic (march2).  The slot tables of the fused descriptor kernel (rays dealt
        // to lanes by length, rolled ray loop, sums parked in the output) were tried here too: same bits, +13 % time (0.298 vs 0.262 ms per 1024
        // images) -- stand-alone, without the rasteriser's register pressure, the unrolled loop with its hoisted table loads wins
        auto kern = per_lane <= 15 ? (p.stride == 125 ? k_radon2<15, 125> : k_radon2<15, 0>) : k_radon2<16, 0>;
        if (2 * lds > 48 * 1024)
            MRS_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)(2 * lds)));
        hipLaunchKernelGGL(kern, dim3((batch + 1) / 2), dim3(kRadonWG), 2 * lds, s, d_img, p, batch, d_sino, d_sino_norm,
                           plan->d_degenerate);
    } else if (per_lane <= 16) {
        auto kern = per_lane <= 15 ? (p.stride == 125 ? k_radon<15, 125> : k_radon<15, 0>) : k_radon<16, 0>;
        if (lds > 48 * 1024)
            MRS_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(kern, dim3(batch), dim3(kRadonWG), lds, s, d_img, p, d_sino, d_sino_norm, plan->d_degenerate);
    } else {
        mrs::Scratch tmp;
        float* raw = d_sino;
        if (!raw) {
            int st = tmp.alloc((size_t)batch * rays * sizeof(float), s);
            if (st != MRS_OK) return st;
            raw = tmp.as<float>();
        }
        if (lds > 48 * 1024)
            MRS_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_radon_big),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(k_radon_big, dim3(batch), dim3(kRadonWG), lds, s, d_img, p, raw);
        if (d_sino_norm)
            hipLaunchKernelGGL(k_normalize, dim3(batch), dim3(1024), 0, s, raw, d_sino_norm, rays, plan->d_degenerate);
    }
    MRS_HIP_TRY(hipGetLastError());
    return MRS_OK;
}

}  // extern "C"
