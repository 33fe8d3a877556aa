// radon.hip -- parallel-beam Radon forward projection for gfx950.
//
// Reference behaviour reproduced (never copied):
//   torch-radon/src/forward.cu:12-124 (radon_forward_kernel<true,1,float>), texture semantics of
//   src/texture.cu:133-143, API torch_radon/radon.py:62-87,139-167; the global mean / unbiased
//   std normalisation of RING_ros/util.py:197 is fused in as an optional second output.
//
// Design: CDNA has no texture sampler, so the image lives in LDS with a 2-texel zero border
// (border address mode) and an odd row stride (bank-conflict-free for rays marching along
// either axis); one workgroup per image, one ray per lane with lanes = consecutive detectors,
// samples aligned exactly to texel centres along the ray's dominant axis (integer stepping), 2-tap
// fp32 interpolation along the minor axis (~14 VALU + 2 ds_read per sample).  HBM traffic is
// 4*H*W in + 4*A*D out per image; the kernel is VALU/LDS bound, not HBM bound.
// Numerics are shared with oracle/radon_oracle.c op for op (cos/sin evaluated on the host in
// double when the plan is built), so HIP == oracle bit for bit.
#include <cmath>

#include "common.hpp"

struct mrs_radon_plan {
    mrs_ctx* ctx = nullptr;
    int n_angles = 0, det = 0, H = 0, W = 0;
    float spacing = 1.0f;
    float L = 0.0f;
    float* d_cs = nullptr;  // [n_angles][2] = (cos, sin)
};

namespace {

constexpr int kRadonWG = 960;  // 15 waves; 120x120 rays = 15 rays per lane exactly
constexpr int kPad = 2;

struct RadonP {
    int A, D, H, W, stride;
    float spacing, L;
    const float* cs;
};

// Sample loop of one ray.  line: LDS address of (dominant texel line, minor index 0); lstep: its
// increment per sample; mc/vm: minor-axis coordinate and increment.  STRIDE > 0: compile-time row
// stride.  Accumulation order is strictly sequential (matches the oracle bit for bit).
template <bool YDOM, int STRIDE>
__device__ __forceinline__ float march(const float* line, int lstep, float mc, float vm, int n_steps, int rstride)
{
    const int unit = YDOM ? 1 : (STRIDE > 0 ? STRIDE : rstride);  // distance between the two taps
    float acc = 0.0f;
    int j = 0;
    for (; j + 4 <= n_steps; j += 4) {   // 4 samples in flight: addresses first, then the taps
        float fr[4], t0[4], t1[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float mb = mc - 0.5f;
            const float fl = floorf(mb);
            fr[u] = mb - fl;
            const float* q = line + (int)fl * unit;
            t0[u] = q[0];
            t1[u] = q[unit];
            mc += vm;
            line += lstep;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) acc += __builtin_fmaf(fr[u], t1[u] - t0[u], t0[u]);
    }
    for (; j < n_steps; ++j) {
        const float mb = mc - 0.5f;
        const float fl = floorf(mb);
        const float* q = line + (int)fl * unit;
        const float t0 = q[0], t1 = q[unit];
        acc += __builtin_fmaf(mb - fl, t1 - t0, t0);
        mc += vm;
        line += lstep;
    }
    return acc;
}

// forward.cu:18-123 for one ray (a, r)
template <int STRIDE>
__device__ __forceinline__ float trace_ray(const float* img, const RadonP& p, int a, int r)
{
    const int stride = STRIDE > 0 ? STRIDE : p.stride;
    const float cs = p.cs[2 * a], sn = p.cs[2 * a + 1];
    const float sx = ((float)r - (float)p.D * 0.5f + 0.5f) * p.spacing;
    const float sy = p.L, ex = sx, ey = -p.L;
    float rsx = sx * cs + sy * sn;
    float rsy = -sx * sn + sy * cs;
    float rdx = ex * cs + ey * sn - rsx;
    float rdy = -ex * sn + ey * cs - rsy;
    rsx = rsx - (-0.5f * (float)p.W);
    rsy = rsy - (-0.5f * (float)p.H);
    const float dx = rdx >= 0 ? fmaxf(rdx, 1e-6f) : fminf(rdx, -1e-6f);
    const float dy = rdy >= 0 ? fmaxf(rdy, 1e-6f) : fminf(rdy, -1e-6f);
    const float axm = (-rsx) / dx, axp = ((float)p.W - rsx) / dx;
    const float aym = (-rsy) / dy, ayp = ((float)p.H - rsy) / dy;
    const float as = fmaxf(fminf(axp, axm), fminf(ayp, aym));
    const float ae = fminf(fmaxf(axp, axm), fmaxf(ayp, aym));
    if ((double)as > (double)ae - 1e-6) return 0.0f;
    rsx += rdx * as;
    rsy += rdy * as;
    rdx *= (ae - as);
    rdy *= (ae - as);
    const float m = fmaxf(fabsf(rdx), fabsf(rdy));
    const int n_steps = (int)rintf(m);
    const float vx = rdx / m, vy = rdy / m;
    const float n = sqrtf(vx * vx + vy * vy);
    float step;
    if (fabsf(rdy) >= fabsf(rdx)) {
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
    // dominant axis: integer texel line stepping by +-1; minor axis: cumulative float coordinate,
    // 2-tap interpolation.  The two orientations get their own loop (wave-uniform branch except
    // for the one wave in 15 that straddles the 45-degree switch) so that both taps come from one
    // ds_read2_b32 and no per-sample integer multiply is needed.  Indices stay inside the 2-texel
    // zero border: the alignment step is in [0,1] and n_steps = rint(length), so a ray overshoots
    // its exit point by at most half a texel.
    const bool ydom = fabsf(rdy) >= fabsf(rdx);
    const int major = (int)floorf(ydom ? rsy : rsx);
    const bool neg = (ydom ? vy : vx) < 0;
    float acc = 0.0f;
    if (ydom)
        acc = march<true, STRIDE>(img + (major + kPad) * stride + kPad, neg ? -stride : stride, rsx, vx, n_steps, stride);
    else
        acc = march<false, STRIDE>(img + kPad * stride + (major + kPad), neg ? -1 : 1, rsy, vy, n_steps, stride);
    return acc * n;
}

__device__ __forceinline__ double wave_sum(double v)
{
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// One workgroup per image.  sino_raw / sino_norm may each be null.
// sino_norm = (S - mean(S)) / std(S) with the unbiased std over the whole sinogram
// (util.py:197: fn.normalize(pc_RING, mean=pc_RING.mean(), std=pc_RING.std())).
template <int MAX_RAYS_PER_LANE, int STRIDE>
__global__ __launch_bounds__(kRadonWG) void k_radon(const float* __restrict__ img, RadonP p,
                                                    float* __restrict__ sino_raw,
                                                    float* __restrict__ sino_norm)
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
    double s1 = 0.0;
#pragma unroll
    for (int k = 0; k < MAX_RAYS_PER_LANE; ++k) {
        const int ray = threadIdx.x + k * kRadonWG;
        float v = 0.0f;
        if (ray < rays) {
            const int a = ray / p.D, r = ray - a * p.D;
            v = trace_ray<STRIDE>(tile, p, a, r);
            if (sino_raw) sino_raw[(size_t)b * rays + ray] = v;
            s1 += (double)v;
        }
        val[k] = v;
    }
    if (!sino_norm) return;
    // mean, then centred sum of squares (two-pass, double accumulation)
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    s1 = wave_sum(s1);
    if (lane == 0) red[0][wave] = s1;
    __syncthreads();
    double tot = 0.0;
    for (int w = 0; w < kRadonWG / 64; ++w) tot += red[0][w];
    const double mean_d = tot / (double)rays;
    const float mean = (float)mean_d;
    double s2 = 0.0;
#pragma unroll
    for (int k = 0; k < MAX_RAYS_PER_LANE; ++k) {
        const int ray = threadIdx.x + k * kRadonWG;
        if (ray < rays) {
            const double dlt = (double)val[k] - mean_d;
            s2 += dlt * dlt;
        }
    }
    s2 = wave_sum(s2);
    if (lane == 0) red[1][wave] = s2;
    __syncthreads();
    double tot2 = 0.0;
    for (int w = 0; w < kRadonWG / 64; ++w) tot2 += red[1][w];
    const float sd = (float)sqrt(tot2 / (double)(rays - 1));
#pragma unroll
    for (int k = 0; k < MAX_RAYS_PER_LANE; ++k) {
        const int ray = threadIdx.x + k * kRadonWG;
        if (ray < rays) sino_norm[(size_t)b * rays + ray] = (val[k] - mean) / sd;
    }
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
    for (int ray = threadIdx.x; ray < rays; ray += kRadonWG) {
        const int a = ray / p.D, r = ray - a * p.D;
        sino_raw[(size_t)b * rays + ray] = trace_ray<0>(tile, p, a, r);
    }
}

// (x - mean) / std over `group` consecutive floats per block (unbiased std), in place or not.
// util.py:339-340 (RING++ normalises a whole [C,H,W] descriptor with one mean/std).
__global__ __launch_bounds__(1024) void k_normalize(const float* __restrict__ in, float* __restrict__ out, int group)
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
    for (int i = threadIdx.x; i < group; i += 1024) dst[i] = (src[i] - mean) / sd;
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
    if (lds > ctx->lds_bytes) {
        mrs::set_error("Radon image %dx%d does not fit the %zu-byte LDS tile", height, width, ctx->lds_bytes);
        return MRS_ERR_UNSUPPORTED;
    }
    MRS_HIP_TRY(hipSetDevice(ctx->device));
    std::vector<float> cs(2 * (size_t)n_angles);
    for (int a = 0; a < n_angles; ++a) {
        cs[2 * a] = (float)cos((double)h_angles[a]);
        cs[2 * a + 1] = (float)sin((double)h_angles[a]);
    }
    mrs_radon_plan* pl = new mrs_radon_plan();
    pl->ctx = ctx;
    pl->n_angles = n_angles; pl->det = det_count; pl->H = height; pl->W = width;
    pl->spacing = det_spacing;
    pl->L = sqrtf((width * 0.5f) * (width * 0.5f) + (height * 0.5f) * (height * 0.5f));
    if (hipMalloc(&pl->d_cs, cs.size() * sizeof(float)) != hipSuccess ||
        hipMemcpy(pl->d_cs, cs.data(), cs.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) {
        mrs::set_error("could not upload the angle table");
        if (pl->d_cs) (void)hipFree(pl->d_cs);
        delete pl;
        return MRS_ERR_HIP;
    }
    *out_plan = pl;
    return MRS_OK;
}

int mrs_radon_plan_destroy(mrs_radon_plan* plan)
{
    if (!plan) return MRS_OK;
    (void)hipSetDevice(plan->ctx->device);
    if (plan->d_cs) (void)hipFree(plan->d_cs);
    delete plan;
    return MRS_OK;
}

int mrs_normalize_groups(mrs_ctx* ctx, const float* d_in, float* d_out, int32_t n_groups, int32_t group_len,
                         mrs_stream stream)
{
    MRS_REQUIRE(ctx && d_in && d_out, "null pointer");
    MRS_REQUIRE(n_groups > 0 && group_len > 1, "need n_groups > 0 and group_len > 1");
    MRS_HIP_TRY(hipSetDevice(ctx->device));
    hipLaunchKernelGGL(k_normalize, dim3(n_groups), dim3(1024), 0, (hipStream_t)stream, d_in, d_out, group_len);
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
    p.spacing = plan->spacing; p.L = plan->L; p.cs = plan->d_cs;
    const size_t lds = (size_t)(p.H + 2 * kPad) * p.stride * sizeof(float);
    hipStream_t s = (hipStream_t)stream;
    const int rays = p.A * p.D;
    const int per_lane = (rays + kRadonWG - 1) / kRadonWG;
    if (per_lane <= 16) {
        auto kern = per_lane <= 15 ? (p.stride == 125 ? k_radon<15, 125> : k_radon<15, 0>) : k_radon<16, 0>;
        if (lds > 48 * 1024)
            MRS_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(kern, dim3(batch), dim3(kRadonWG), lds, s, d_img, p, d_sino, d_sino_norm);
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
            hipLaunchKernelGGL(k_normalize, dim3(batch), dim3(1024), 0, s, raw, d_sino_norm, rays);
    }
    MRS_HIP_TRY(hipGetLastError());
    return MRS_OK;
}

}  // extern "C"
