// tools/ubench/radon_march.hip -- development harness: variants of the two-image Radon march (radon_device.hpp march2, the VALU/LDS half
// of the fused descriptor kernel) timed against each other on one box, each checked bit for bit against the library's k_radon2.
// Links against libmrslam_hip.so for the plan (host-built ray table) and the reference output.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -I../../mr_slam_amd/csrc radon_march.hip \
//         -L../../mr_slam_amd -lmrslam_hip -Wl,-rpath,'$ORIGIN/../../mr_slam_amd' -o radon_march
// Variants (all: one workgroup of 1024 lanes per image pair, tile staged from global memory like k_radon2):
//   base      : the library's structure (15 rays per lane unrolled, values kept in registers for the normalisation)
//   rolled    : the ray loop not unrolled; raw sums parked in the output buffer and normalised in place
//   rolled + U: chunk length U of the sample loop (cells of a chunk requested together)
//   pipe      : chunk k + 1's cells requested before chunk k's FMAs (two register stages)
#include "common.hpp"
#include "radon_device.hpp"

#include <cstdio>
#include <algorithm>
#include <cstdlib>
#include <vector>

#define CHECK(e)                                                                              \
    do {                                                                                      \
        hipError_t s__ = (e);                                                                 \
        if (s__ != hipSuccess) {                                                              \
            fprintf(stderr, "%s: %s (%s:%d)\n", #e, hipGetErrorString(s__), __FILE__, __LINE__); \
            exit(1);                                                                          \
        }                                                                                     \
    } while (0)

namespace {

// ---- sample-loop variants ---------------------------------------------------------------------------------------------------
template <bool YDOM, int STRIDE, int U, bool PIPE>
__device__ __forceinline__ void march2v(unsigned off, float q, float vm, int n_steps, float& outA, float& outB)
{
    constexpr int unit = (YDOM ? 1 : STRIDE) * 8;
    constexpr int lstep = (YDOM ? STRIDE : 1) * 8;
    v2f acc0 = {0.0f, 0.0f}, acc1 = {0.0f, 0.0f};
    auto issue = [&](v2f (&t0)[U], v2f (&t1)[U], float (&fr)[U]) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            fr[u] = __builtin_amdgcn_fractf(q);
            const unsigned a = off + (unsigned)(YDOM ? (int)q * 8 : __mul24((int)q, unit));
            t0[u] = lds_cell(a + u * lstep);
            t1[u] = lds_cell(a + u * lstep + unit);
            q += vm;
        }
        off += U * lstep;
    };
    auto consume = [&](const v2f (&t0)[U], const v2f (&t1)[U], const float (&fr)[U]) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const float w0 = 1.0f - fr[u];
            const v2f W0 = {w0, w0}, W1 = {fr[u], fr[u]};
            acc0 = __builtin_elementwise_fma(t0[u], W0, acc0);
            acc1 = __builtin_elementwise_fma(t1[u], W1, acc1);
        }
    };
    const int chunks = n_steps / U;
    int j = chunks * U;
    if (PIPE) {
        if (chunks > 0) {
            v2f a0[U], a1[U], b0[U], b1[U];
            float af[U], bf[U];
            int c = 0;
            issue(a0, a1, af);
#pragma nounroll
            while (true) {
                if (c + 1 < chunks) issue(b0, b1, bf);
                consume(a0, a1, af);
                if (++c >= chunks) break;
                if (c + 1 < chunks) issue(a0, a1, af);
                consume(b0, b1, bf);
                if (++c >= chunks) break;
            }
        }
    } else {
#pragma nounroll
        for (int c = 0; c < chunks; ++c) {
            v2f t0[U], t1[U];
            float fr[U];
            issue(t0, t1, fr);
            consume(t0, t1, fr);
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

template <int STRIDE, int U, bool PIPE>
__device__ __forceinline__ void trace_ray2v(const v2f* cells, const RadonP& p, int ray, float& outA, float& outB)
{
    const int meta = p.meta[ray];
    const int n_steps = meta & 0xffff;
    if (n_steps == 0) { outA = 0.0f; outB = 0.0f; return; }
    const float q = p.q[ray], vm = p.vm[ray], n = p.nrm[ray];
    const unsigned tile = (unsigned)(uintptr_t)(lds_cptr)reinterpret_cast<const char*>(cells) + 2u * (unsigned)p.base[ray];
    float a, b;
    if (meta >> 16) march2v<true, STRIDE, U, PIPE>(tile, q, vm, n_steps, a, b);
    else march2v<false, STRIDE, U, PIPE>(tile, q, vm, n_steps, a, b);
    outA = a * n;
    outB = b * n;
}

// normalize_store with the raw values parked in dst (same lane <-> ray mapping, same reduction order: same bits)
__device__ __forceinline__ void normalize_inplace(float* __restrict__ dst, int rays, int per_lane, double (&red)[2][16], int* __restrict__ degenerate)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    double s1 = 0.0;
    for (int k = 0; k < per_lane; ++k) {
        const int ray = threadIdx.x + k * kRadonWG;
        if (ray < rays) s1 += (double)dst[ray];
    }
    s1 = wave_sum(s1);
    __syncthreads();
    if (lane == 0) red[0][wave] = s1;
    __syncthreads();
    double tot = 0.0;
    for (int w = 0; w < kRadonWG / 64; ++w) tot += red[0][w];
    const double mean_d = tot / (double)rays;
    const float mean = (float)mean_d;
    double s2 = 0.0;
    for (int k = 0; k < per_lane; ++k) {
        const int ray = threadIdx.x + k * kRadonWG;
        if (ray < rays) {
            const double dlt = (double)dst[ray] - mean_d;
            s2 += dlt * dlt;
        }
    }
    s2 = wave_sum(s2);
    if (lane == 0) red[1][wave] = s2;
    __syncthreads();
    double tot2 = 0.0;
    for (int w = 0; w < kRadonWG / 64; ++w) tot2 += red[1][w];
    const float sd = (float)sqrt(tot2 / (double)(rays - 1));
    const bool ok = sd > 0.0f && sd < INFINITY;
    if (!ok && threadIdx.x == 0 && degenerate) atomicAdd(degenerate, 1);
    for (int k = 0; k < per_lane; ++k) {
        const int ray = threadIdx.x + k * kRadonWG;
        if (ray < rays) dst[ray] = ok ? (dst[ray] - mean) / sd : 0.0f;
    }
}

__device__ __forceinline__ void stage_pair(v2f* cells, const float* __restrict__ img, const RadonP& p, int b0, bool two)
{
    const float* srcA = img + (size_t)b0 * p.H * p.W;
    const float* srcB = img + (size_t)(two ? b0 + 1 : b0) * p.H * p.W;
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
}

// the library's structure
template <int STRIDE>
__global__ __launch_bounds__(kRadonWG) void k_base(const float* __restrict__ img, RadonP p, int batch, float* __restrict__ sino_norm,
                                                   int* __restrict__ degenerate)
{
    extern __shared__ __attribute__((aligned(16))) v2f cells[];
    __shared__ double red[2][16];
    const int b0 = 2 * blockIdx.x, b1 = b0 + 1;
    const bool two = b1 < batch;
    stage_pair(cells, img, p, b0, two);
    const int rays = p.A * p.D;
    float va[15], vb[15];
#pragma unroll
    for (int k = 0; k < 15; ++k) {
        const int ray = threadIdx.x + k * kRadonWG;
        float a = 0.0f, b = 0.0f;
        if (ray < rays) trace_ray2<STRIDE>(cells, p, ray, a, b);
        va[k] = a; vb[k] = b;
    }
    normalize_store<15>(va, rays, red, sino_norm + (size_t)b0 * rays, degenerate);
    if (two) normalize_store<15>(vb, rays, red, sino_norm + (size_t)b1 * rays, degenerate);
}

template <int STRIDE, int U, bool PIPE>
__global__ __launch_bounds__(kRadonWG) void k_rolled(const float* __restrict__ img, RadonP p, int batch, float* __restrict__ sino_norm,
                                                     int* __restrict__ degenerate)
{
    extern __shared__ __attribute__((aligned(16))) v2f cells[];
    __shared__ double red[2][16];
    const int b0 = 2 * blockIdx.x, b1 = b0 + 1;
    const bool two = b1 < batch;
    stage_pair(cells, img, p, b0, two);
    const int rays = p.A * p.D;
    const int per_lane = (rays + kRadonWG - 1) / kRadonWG;
    float* dA = sino_norm + (size_t)b0 * rays;
    float* dB = sino_norm + (size_t)(two ? b1 : b0) * rays;
#pragma nounroll
    for (int k = 0; k < per_lane; ++k) {
        const int ray = threadIdx.x + k * kRadonWG;
        if (ray < rays) {
            float a, b;
            trace_ray2v<STRIDE, U, PIPE>(cells, p, ray, a, b);
            dA[ray] = a;
            if (two) dB[ray] = b;
        }
    }
    normalize_inplace(dA, rays, per_lane, red, degenerate);
    if (two) normalize_inplace(dB, rays, per_lane, red, degenerate);
}

// ---- ablations of the base structure (timing only: results differ on purpose) ------------------------------------------------
//   ABL 1: both taps read ONE fixed cell (same address in every lane: broadcast, no bank conflicts)   -> what the conflicts cost
//   ABL 2: no tail loop (n_steps rounded down to whole chunks)                                         -> what the per-step tails cost
//   ABL 3: no FMAs (loaded cells xor-ed into a dummy)                                                   -> what the packed FMAs cost
//   ABL 4: no LDS reads at all (cells = the weights)                                                    -> VALU-only time
//   ABL 5: every lane's ray forced to 120 steps of its own geometry is not possible; instead all lanes march ray 60 of their angle
template <bool YDOM, int STRIDE, int ABL>
__device__ __forceinline__ void march2a(unsigned off, float q, float vm, int n_steps, float& outA, float& outB)
{
    constexpr int unit = (YDOM ? 1 : STRIDE) * 8;
    constexpr int lstep = (YDOM ? STRIDE : 1) * 8;
    constexpr int U = 6;
    v2f acc0 = {0.0f, 0.0f}, acc1 = {0.0f, 0.0f};
    const unsigned fixed = off;
    int j = 0;
#pragma nounroll
    for (; j + U <= n_steps; j += U) {
        v2f t0[U], t1[U];
        float w0[U], w1[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const float fr = __builtin_amdgcn_fractf(q);
            unsigned a = off + (unsigned)(YDOM ? (int)q * 8 : __mul24((int)q, unit));
            if (ABL == 1) { asm volatile("" : "+v"(a)); a = fixed; }
            if (ABL == 4) {
                const v2f c = {fr, __int_as_float(a)};
                t0[u] = c; t1[u] = c;
            } else {
                t0[u] = lds_cell(a + u * lstep);
                t1[u] = lds_cell(a + u * lstep + unit);
            }
            w0[u] = 1.0f - fr;
            w1[u] = fr;
            q += vm;
        }
        off += U * lstep;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (ABL == 3) {
                acc0.x = __int_as_float(__float_as_int(acc0.x) ^ __float_as_int(t0[u].x) ^ __float_as_int(t1[u].y) ^ __float_as_int(w0[u]));
                acc1.y = __int_as_float(__float_as_int(acc1.y) ^ __float_as_int(t0[u].y) ^ __float_as_int(t1[u].x) ^ __float_as_int(w1[u]));
            } else {
                const v2f W0 = {w0[u], w0[u]}, W1 = {w1[u], w1[u]};
                acc0 = __builtin_elementwise_fma(t0[u], W0, acc0);
                acc1 = __builtin_elementwise_fma(t1[u], W1, acc1);
            }
        }
    }
    if (ABL != 2)
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

template <int STRIDE, int ABL>
__global__ __launch_bounds__(kRadonWG) void k_abl(const float* __restrict__ img, RadonP p, int batch, float* __restrict__ sino_norm,
                                                  int* __restrict__ degenerate)
{
    extern __shared__ __attribute__((aligned(16))) v2f cells[];
    __shared__ double red[2][16];
    const int b0 = 2 * blockIdx.x, b1 = b0 + 1;
    const bool two = b1 < batch;
    stage_pair(cells, img, p, b0, two);
    const int rays = p.A * p.D;
    float va[15], vb[15];
#pragma unroll
    for (int k = 0; k < 15; ++k) {
        int ray = threadIdx.x + k * kRadonWG;
        float a = 0.0f, b = 0.0f;
        if (ray < rays) {
            if (ABL == 5) ray = (ray / p.D) * p.D + p.D / 2;      // every lane the central ray of its angle: no divergence in n_steps
            const int meta = p.meta[ray];
            const int n_steps = meta & 0xffff;
            if (n_steps > 0) {
                const float q = p.q[ray], vm = p.vm[ray], n = p.nrm[ray];
                const unsigned tile = (unsigned)(uintptr_t)(lds_cptr)reinterpret_cast<const char*>(cells) + 2u * (unsigned)p.base[ray];
                if (meta >> 16) march2a<true, STRIDE, ABL>(tile, q, vm, n_steps, a, b);
                else march2a<false, STRIDE, ABL>(tile, q, vm, n_steps, a, b);
                a *= n; b *= n;
            }
        }
        va[k] = a; vb[k] = b;
    }
    normalize_store<15>(va, rays, red, sino_norm + (size_t)b0 * rays, degenerate);
    if (two) normalize_store<15>(vb, rays, red, sino_norm + (size_t)b1 * rays, degenerate);
}

// ---- rays re-assigned to lanes: slot s = k * 1024 + wave * 64 + lane carries ray rayid[s] (-1: idle), tables in slot order ----
template <int STRIDE, int PER_LANE>
__global__ __launch_bounds__(kRadonWG) void k_slots(const float* __restrict__ img, RadonP p, const int* __restrict__ rayid, int batch,
                                                    float* __restrict__ sino_norm, int* __restrict__ degenerate)
{
    extern __shared__ __attribute__((aligned(16))) v2f cells[];
    __shared__ double red[2][16];
    const int b0 = 2 * blockIdx.x, b1 = b0 + 1;
    const bool two = b1 < batch;
    stage_pair(cells, img, p, b0, two);
    const int rays = p.A * p.D;
    float* dA = sino_norm + (size_t)b0 * rays;
    float* dB = sino_norm + (size_t)(two ? b1 : b0) * rays;
#pragma unroll
    for (int k = 0; k < PER_LANE; ++k) {
        const int slot = threadIdx.x + k * kRadonWG;
        const int ray = rayid[slot];
        if (ray >= 0) {
            float a, b;
            trace_ray2<STRIDE>(cells, p, slot, a, b);      // the tables are in slot order
            dA[ray] = a;
            if (two) dB[ray] = b;
        }
    }
    __syncthreads();     // raw sums of every ray are in place (same workgroup: visible after the barrier)
    const int per_lane = (rays + kRadonWG - 1) / kRadonWG;
    normalize_inplace(dA, rays, per_lane, red, degenerate);
    if (two) normalize_inplace(dB, rays, per_lane, red, degenerate);
}

struct Variant {
    const char* name;
    void* fn;
};

}  // namespace

int main(int argc, char** argv)
{
    const int B = argc > 1 ? atoi(argv[1]) : 2048;
    const int H = 120, W = 120, A = 120, D = 120;
    mrs_ctx* ctx = nullptr;
    if (mrs_ctx_create(0, &ctx) != MRS_OK) { fprintf(stderr, "ctx: %s\n", mrs_last_error()); return 1; }
    std::vector<float> ang(A);
    for (int i = 0; i < A; ++i) ang[i] = (float)(2.0 * M_PI * i / (A - 1));   // linspace(0, 2 pi, 120)
    mrs_radon_plan* plan = nullptr;
    if (mrs_radon_plan_create(ctx, ang.data(), A, D, 1.0f, H, W, &plan) != MRS_OK) { fprintf(stderr, "plan: %s\n", mrs_last_error()); return 1; }
    // sparse images like max-z BEVs: ~12 % of the texels non-zero
    std::vector<float> h((size_t)B * H * W);
    unsigned s = 12345u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (float)(s >> 8) * (1.0f / 16777216.0f); };
    for (auto& v : h) { const float r = rnd(); v = r < 0.12f ? rnd() : 0.0f; }
    float *d_img, *d_ref, *d_out;
    const size_t ibytes = h.size() * sizeof(float), obytes = (size_t)B * A * D * sizeof(float);
    CHECK(hipMalloc(&d_img, ibytes)); CHECK(hipMalloc(&d_ref, obytes)); CHECK(hipMalloc(&d_out, obytes));
    CHECK(hipMemcpy(d_img, h.data(), ibytes, hipMemcpyHostToDevice));
    if (mrs_radon_forward(plan, d_img, B, nullptr, d_ref, nullptr) != MRS_OK) { fprintf(stderr, "forward: %s\n", mrs_last_error()); return 1; }
    CHECK(hipDeviceSynchronize());
    std::vector<float> ref((size_t)B * A * D), out((size_t)B * A * D);
    CHECK(hipMemcpy(ref.data(), d_ref, obytes, hipMemcpyDeviceToHost));

    RadonP p;
    p.A = A; p.D = D; p.H = H; p.W = W; p.stride = (W + 2 * kPad) | 1;
    const size_t nr = (size_t)A * D;
    p.meta = plan->d_meta; p.base = plan->d_meta + nr;
    p.q = reinterpret_cast<const float*>(plan->d_meta + 2 * nr); p.vm = p.q + nr; p.nrm = p.vm + nr;
    const size_t lds = 2 * (size_t)(H + 2 * kPad) * p.stride * sizeof(float);
    int* d_deg = plan->d_degenerate;
    int batch = B;
    const Variant vars[] = {
        {"base (15 rays unrolled, U=6)", (void*)k_base<125>},
        {"rolled U=6", (void*)k_rolled<125, 6, false>},
        {"rolled U=8", (void*)k_rolled<125, 8, false>},
        {"rolled U=10", (void*)k_rolled<125, 10, false>},
        {"rolled U=12", (void*)k_rolled<125, 12, false>},
        {"rolled pipe U=4", (void*)k_rolled<125, 4, true>},
        {"rolled pipe U=6", (void*)k_rolled<125, 6, true>},
        {"rolled pipe U=8", (void*)k_rolled<125, 8, true>},
        {"ablation 0 (= base, own copy)", (void*)k_abl<125, 0>},
        {"ablation 1: all taps one fixed cell (no bank conflicts)", (void*)k_abl<125, 1>},
        {"ablation 2: no tail loop", (void*)k_abl<125, 2>},
        {"ablation 3: no FMAs", (void*)k_abl<125, 3>},
        {"ablation 4: no LDS reads", (void*)k_abl<125, 4>},
        {"ablation 5: every lane the central ray of its angle (120 steps, no divergence; 1.18x the samples)", (void*)k_abl<125, 5>},
    };
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    printf("{\"images\": %d, \"library_k_radon2_ms_per_1024\": ", B);
    {
        for (int i = 0; i < 2; ++i) mrs_radon_forward(plan, d_img, B, nullptr, d_ref, nullptr);
        CHECK(hipEventRecord(e0));
        for (int i = 0; i < 5; ++i) mrs_radon_forward(plan, d_img, B, nullptr, d_ref, nullptr);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        printf("%.4f, \"variants\": [\n", ms / 5 * 1024 / B);
    }
    bool first = true;
    for (const Variant& v : vars) {
        CHECK(hipFuncSetAttribute(v.fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        void* args[] = {(void*)&d_img, (void*)&p, (void*)&batch, (void*)&d_out, (void*)&d_deg};
        CHECK(hipMemset(d_out, 0xff, obytes));
        for (int i = 0; i < 2; ++i) CHECK(hipLaunchKernel(v.fn, dim3((B + 1) / 2), dim3(kRadonWG), args, lds, 0));
        CHECK(hipEventRecord(e0));
        for (int i = 0; i < 5; ++i) CHECK(hipLaunchKernel(v.fn, dim3((B + 1) / 2), dim3(kRadonWG), args, lds, 0));
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        CHECK(hipMemcpy(out.data(), d_out, obytes, hipMemcpyDeviceToHost));
        size_t bad = 0;
        for (size_t i = 0; i < out.size(); ++i) bad += memcmp(&out[i], &ref[i], 4) != 0;
        hipFuncAttributes fa;
        CHECK(hipFuncGetAttributes(&fa, v.fn));
        printf("%s  {\"name\": \"%s\", \"ms_per_1024\": %.4f, \"mismatching_values\": %zu, \"vgprs\": %d, \"scratch_bytes\": %zu}", first ? "" : ",\n", v.name,
               ms / 5 * 1024 / B, bad, fa.numRegs, (size_t)fa.localSizeBytes);
        first = false;
    }
    // ---- slot tables: rays sorted by (orientation, n_steps) in chunks of 64, chunks dealt to the 16 waves longest first ----------
    {
        std::vector<int> meta(nr), base(nr);
        std::vector<float> q(nr), vm(nr), nrm(nr);
        CHECK(hipMemcpy(meta.data(), p.meta, nr * 4, hipMemcpyDeviceToHost));
        CHECK(hipMemcpy(base.data(), p.base, nr * 4, hipMemcpyDeviceToHost));
        CHECK(hipMemcpy(q.data(), p.q, nr * 4, hipMemcpyDeviceToHost));
        CHECK(hipMemcpy(vm.data(), p.vm, nr * 4, hipMemcpyDeviceToHost));
        CHECK(hipMemcpy(nrm.data(), p.nrm, nr * 4, hipMemcpyDeviceToHost));
        for (int mode = 0; mode < 3; ++mode) {
            constexpr int PL = 15;
            const int slots = PL * kRadonWG;
            std::vector<int> order(nr);
            for (size_t i = 0; i < nr; ++i) order[i] = (int)i;
            if (mode >= 1)
                std::stable_sort(order.begin(), order.end(), [&](int a, int b) {
                    const int ya = meta[a] >> 16, yb = meta[b] >> 16, na = meta[a] & 0xffff, nb = meta[b] & 0xffff;
                    if (ya != yb) return ya > yb;
                    return na > nb;
                });
            const int chunks = (int)((nr + 63) / 64);
            std::vector<int> chunk_slot(chunks);           // chunk -> (k, wave) position
            if (mode == 2) {                               // longest chunk first onto the least loaded wave (at most PL chunks per wave)
                std::vector<long> load(16, 0);
                std::vector<int> cnt(16, 0);
                std::vector<int> cost(chunks), idx(chunks);
                for (int c = 0; c < chunks; ++c) {
                    int m = 0;
                    for (int l = 0; l < 64 && (size_t)(c * 64 + l) < nr; ++l) m = std::max(m, meta[order[c * 64 + l]] & 0xffff);
                    cost[c] = m; idx[c] = c;
                }
                std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) { return cost[a] > cost[b]; });
                for (int c : idx) {
                    int w = -1;
                    for (int i = 0; i < 16; ++i) if (cnt[i] < PL && (w < 0 || load[i] < load[w])) w = i;
                    chunk_slot[c] = cnt[w] * 16 + w;
                    cnt[w]++; load[w] += cost[c];
                }
            } else {
                for (int c = 0; c < chunks; ++c) chunk_slot[c] = c;      // k = c / 16, wave = c % 16: the library's dealing
            }
            std::vector<int> s_meta(slots, 0), s_base(slots, 0), s_ray(slots, -1);
            std::vector<float> s_q(slots, 0.0f), s_vm(slots, 0.0f), s_nrm(slots, 0.0f);
            for (int c = 0; c < chunks; ++c)
                for (int l = 0; l < 64 && (size_t)(c * 64 + l) < nr; ++l) {
                    const int r = order[c * 64 + l];
                    const int k = chunk_slot[c] / 16, w = chunk_slot[c] % 16;
                    const int sl = k * kRadonWG + w * 64 + l;
                    s_meta[sl] = meta[r]; s_base[sl] = base[r]; s_q[sl] = q[r]; s_vm[sl] = vm[r]; s_nrm[sl] = nrm[r]; s_ray[sl] = r;
                }
            int* d_tab; int* d_ray;
            CHECK(hipMalloc(&d_tab, (size_t)slots * 5 * 4)); CHECK(hipMalloc(&d_ray, (size_t)slots * 4));
            CHECK(hipMemcpy(d_tab, s_meta.data(), slots * 4, hipMemcpyHostToDevice));
            CHECK(hipMemcpy(d_tab + slots, s_base.data(), slots * 4, hipMemcpyHostToDevice));
            CHECK(hipMemcpy(d_tab + 2 * slots, s_q.data(), slots * 4, hipMemcpyHostToDevice));
            CHECK(hipMemcpy(d_tab + 3 * slots, s_vm.data(), slots * 4, hipMemcpyHostToDevice));
            CHECK(hipMemcpy(d_tab + 4 * slots, s_nrm.data(), slots * 4, hipMemcpyHostToDevice));
            CHECK(hipMemcpy(d_ray, s_ray.data(), slots * 4, hipMemcpyHostToDevice));
            RadonP ps = p;
            ps.meta = d_tab; ps.base = d_tab + slots;
            ps.q = reinterpret_cast<const float*>(d_tab + 2 * slots); ps.vm = ps.q + slots; ps.nrm = ps.vm + slots;
            void* fn = (void*)k_slots<125, PL>;
            CHECK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            void* args[] = {(void*)&d_img, (void*)&ps, (void*)&d_ray, (void*)&batch, (void*)&d_out, (void*)&d_deg};
            CHECK(hipMemset(d_out, 0xff, obytes));
            for (int i = 0; i < 2; ++i) CHECK(hipLaunchKernel(fn, dim3((B + 1) / 2), dim3(kRadonWG), args, lds, 0));
            CHECK(hipEventRecord(e0));
            for (int i = 0; i < 5; ++i) CHECK(hipLaunchKernel(fn, dim3((B + 1) / 2), dim3(kRadonWG), args, lds, 0));
            CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            CHECK(hipMemcpy(out.data(), d_out, obytes, hipMemcpyDeviceToHost));
            size_t bad = 0;
            for (size_t i = 0; i < out.size(); ++i) bad += memcmp(&out[i], &ref[i], 4) != 0;
            const char* names[] = {"slot tables, natural order (in-place normalisation)", "slot tables, rays sorted by (orientation, n_steps)",
                                   "slot tables, sorted + chunks dealt longest-first to the 16 waves"};
            printf(",\n  {\"name\": \"%s\", \"ms_per_1024\": %.4f, \"mismatching_values\": %zu}", names[mode], ms / 5 * 1024 / B, bad);
            CHECK(hipFree(d_tab)); CHECK(hipFree(d_ray));
        }
    }
    printf("\n]}\n");
    return 0;
}
