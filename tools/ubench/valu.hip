// tools/ubench/valu.hip -- issue-rate microbenchmark for gfx950: how many shader cycles does one wave64 instruction of a
// given kind occupy its SIMD for?  (bench.py priced SQ_INSTS_VALU at 4 cycles, the micro-architecture
// guide says 2.)  Every kernel runs 8 independent dependency chains of ONE instruction kind, 8 x 32 instructions per asm block,
// 64 blocks per wave; the grid puts k = 1, 2, 4, 8 waves on every SIMD of every compute unit (256-thread workgroups: one wave
// per SIMD each).  Time per wave from s_memtime (shader clock) and, for the whole launch, from HIP events.
//   cycles per wave-instruction per SIMD = mean wave time / (k x instructions per wave)
// Output: one JSON object on stdout.  Build: hipcc --offload-arch=gfx950 -O2 valu.hip -o valu
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#define CHECK(e)                                                                              \
    do {                                                                                      \
        hipError_t s__ = (e);                                                                 \
        if (s__ != hipSuccess) {                                                              \
            fprintf(stderr, "%s: %s (%s:%d)\n", #e, hipGetErrorString(s__), __FILE__, __LINE__); \
            exit(1);                                                                          \
        }                                                                                     \
    } while (0)

typedef float v2f __attribute__((ext_vector_type(2)));
constexpr int kBlocks = 64;     // asm blocks per wave
constexpr int kPerBlock = 256;  // instructions per block (8 chains x 32 repeats)

// 32-bit destination / sources
#define KERNEL32(NAME, TEXT)                                                                                          \
    __global__ __launch_bounds__(256) void NAME(unsigned long long* __restrict__ t, float* __restrict__ sink, float s) \
    {                                                                                                                 \
        float r0 = s + threadIdx.x, r1 = r0 + 1, r2 = r0 + 2, r3 = r0 + 3, r4 = r0 + 4, r5 = r0 + 5, r6 = r0 + 6,    \
              r7 = r0 + 7;                                                                                            \
        float a = s * 0.999f, b = s * 1e-3f;                                                                          \
        asm volatile("s_mov_b32 s20, 0x55555555\n s_mov_b32 s21, 0x55555555" ::: "s20", "s21");                                     \
        const unsigned long long t0 = __builtin_readcyclecounter();                                                   \
        for (int i = 0; i < kBlocks; ++i)                                                                             \
            asm volatile(".rept 32\n" TEXT(0) TEXT(1) TEXT(2) TEXT(3) TEXT(4) TEXT(5) TEXT(6) TEXT(7) ".endr" \
                         : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7)             \
                         : "v"(a), "v"(b)                                                                             \
                         : "vcc", "s20", "s21");                                                                      \
        const unsigned long long t1 = __builtin_readcyclecounter();                                                   \
        if ((threadIdx.x & 63) == 0) t[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;                                \
        if (r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7 == 12345.678f) sink[0] = r0;                                        \
    }

// 64-bit destination (packed fp32 / fp64)
#define KERNEL64(NAME, TEXT)                                                                                          \
    __global__ __launch_bounds__(256) void NAME(unsigned long long* __restrict__ t, float* __restrict__ sink, float s) \
    {                                                                                                                 \
        v2f r0 = {s + threadIdx.x, s}, r1 = r0 + 1, r2 = r0 + 2, r3 = r0 + 3, r4 = r0 + 4, r5 = r0 + 5, r6 = r0 + 6,  \
            r7 = r0 + 7;                                                                                              \
        v2f a = {s * 0.999f, s * 0.998f}, b = {s * 1e-3f, s * 2e-3f};                                                 \
        const unsigned long long t0 = __builtin_readcyclecounter();                                                   \
        for (int i = 0; i < kBlocks; ++i)                                                                             \
            asm volatile(".rept 32\n" TEXT(0) TEXT(1) TEXT(2) TEXT(3) TEXT(4) TEXT(5) TEXT(6) TEXT(7) ".endr" \
                         : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7)             \
                         : "v"(a), "v"(b));                                                                           \
        const unsigned long long t1 = __builtin_readcyclecounter();                                                   \
        if ((threadIdx.x & 63) == 0) t[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;                                \
        const v2f q = r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7;                                                          \
        if (q.x + q.y == 12345.678f) sink[0] = q.x;                                                                   \
    }

#define T_FMA(R) "v_fma_f32 %" #R ", %8, %" #R ", %9\n"
#define T_ADD(R) "v_add_f32 %" #R ", %8, %" #R "\n"
#define T_MUL(R) "v_mul_f32 %" #R ", %8, %" #R "\n"
#define T_MAX(R) "v_max_f32 %" #R ", %8, %" #R "\n"
#define T_FRACT(R) "v_fract_f32 %" #R ", %" #R "\n"
#define T_FLOOR(R) "v_floor_f32 %" #R ", %" #R "\n"
#define T_CVTI(R) "v_cvt_i32_f32 %" #R ", %" #R "\n"
#define T_CVTF(R) "v_cvt_f32_i32 %" #R ", %" #R "\n"
#define T_MUL24(R) "v_mul_u32_u24 %" #R ", %8, %" #R "\n"
#define T_MAD24(R) "v_mad_u32_u24 %" #R ", %8, %" #R ", %9\n"
#define T_MULLO(R) "v_mul_lo_u32 %" #R ", %8, %" #R "\n"
#define T_LSHLADD(R) "v_lshl_add_u32 %" #R ", %" #R ", 3, %8\n"
#define T_ADDU(R) "v_add_u32 %" #R ", %8, %" #R "\n"
#define T_AND(R) "v_and_b32 %" #R ", %8, %" #R "\n"
#define T_CNDMASK(R) "v_cndmask_b32 %" #R ", %8, %" #R ", vcc\n"
#define T_RCP(R) "v_rcp_f32 %" #R ", %" #R "\n"
#define T_SQRT(R) "v_sqrt_f32 %" #R ", %" #R "\n"
#define T_MOV(R) "v_mov_b32 %" #R ", %8\n"
#define T_CMP(R) "v_cmp_lt_f32 vcc, %8, %" #R "\n"
#define T_CMP64(R) "v_cmp_lt_f32 s[20:21], %8, %" #R "\n"
#define T_CMPCND(R) "v_cmp_lt_f32 vcc, %8, %" #R "\n s_nop 1\n v_cndmask_b32 %" #R ", %9, %" #R ", vcc\n"
#define T_CMPCND64(R) "v_cmp_lt_f32 s[20:21], %8, %" #R "\n s_nop 1\n v_cndmask_b32 %" #R ", %9, %" #R ", s[20:21]\n"
#define T_CNDS(R) "v_cndmask_b32 %" #R ", %8, %" #R ", s[20:21]\n"
#define T_MED3(R) "v_med3_f32 %" #R ", %8, %" #R ", %9\n"
#define T_MIN(R) "v_min_f32 %" #R ", %8, %" #R "\n"
#define T_MAX3(R) "v_max3_f32 %" #R ", %8, %" #R ", %9\n"
#define T_SUB(R) "v_sub_f32 %" #R ", %8, %" #R "\n"
#define T_XOR(R) "v_xor_b32 %" #R ", %8, %" #R "\n"
#define T_LSHL(R) "v_lshlrev_b32 %" #R ", 3, %" #R "\n"
#define T_BFE(R) "v_bfe_u32 %" #R ", %" #R ", 3, 9\n"
#define T_ADD3(R) "v_add3_u32 %" #R ", %8, %" #R ", %9\n"
#define T_DPP(R) "v_mov_b32_dpp %" #R ", %" #R " row_shr:1 row_mask:0xf bank_mask:0xf\n"
#define T_BPERM(R) "ds_bpermute_b32 %" #R ", %8, %" #R "\n s_waitcnt lgkmcnt(0)\n"
#define T_PKFMA(R) "v_pk_fma_f32 %" #R ", %8, %" #R ", %9\n"
#define T_PKADD(R) "v_pk_add_f32 %" #R ", %8, %" #R "\n"
#define T_PKMUL(R) "v_pk_mul_f32 %" #R ", %8, %" #R "\n"
#define T_FMA64(R) "v_fma_f64 %" #R ", %8, %" #R ", %9\n"
#define T_ADD64(R) "v_add_f64 %" #R ", %8, %" #R "\n"
#define T_MUL64(R) "v_mul_f64 %" #R ", %8, %" #R "\n"

KERNEL32(k_fma, T_FMA)
KERNEL32(k_add, T_ADD)
KERNEL32(k_mul, T_MUL)
KERNEL32(k_max, T_MAX)
KERNEL32(k_fract, T_FRACT)
KERNEL32(k_floor, T_FLOOR)
KERNEL32(k_cvt_i32_f32, T_CVTI)
KERNEL32(k_cvt_f32_i32, T_CVTF)
KERNEL32(k_mul_u32_u24, T_MUL24)
KERNEL32(k_mad_u32_u24, T_MAD24)
KERNEL32(k_mul_lo_u32, T_MULLO)
KERNEL32(k_lshl_add_u32, T_LSHLADD)
KERNEL32(k_add_u32, T_ADDU)
KERNEL32(k_and_b32, T_AND)
KERNEL32(k_cndmask, T_CNDMASK)
KERNEL32(k_rcp, T_RCP)
KERNEL32(k_sqrt, T_SQRT)
KERNEL32(k_mov, T_MOV)
KERNEL32(k_cmp_vcc, T_CMP)
KERNEL32(k_cmp_sgpr, T_CMP64)
KERNEL32(k_cmp_cndmask_vcc, T_CMPCND)
KERNEL32(k_cmp_cndmask_sgpr, T_CMPCND64)
KERNEL32(k_cndmask_sgpr, T_CNDS)
KERNEL32(k_med3, T_MED3)
KERNEL32(k_min, T_MIN)
KERNEL32(k_max3, T_MAX3)
KERNEL32(k_sub, T_SUB)
KERNEL32(k_xor, T_XOR)
KERNEL32(k_lshl, T_LSHL)
KERNEL32(k_bfe, T_BFE)
KERNEL32(k_add3, T_ADD3)
KERNEL32(k_dpp, T_DPP)
KERNEL32(k_bperm, T_BPERM)
KERNEL64(k_pk_fma, T_PKFMA)
KERNEL64(k_pk_add, T_PKADD)
KERNEL64(k_pk_mul, T_PKMUL)
KERNEL64(k_fma_f64, T_FMA64)
KERNEL64(k_add_f64, T_ADD64)
KERNEL64(k_mul_f64, T_MUL64)

// LDS read rates at the Radon kernel's access shapes: 8 independent ds_read_b64 per group, addresses per lane
//   mode 0: lane * 8 (conflict-free, consecutive cells)            mode 1: lane * stride * 8 with stride = 125 cells (x-dominant rays of
//   adjacent detectors: 125 * 2 dwords = 250 -> banks 0, 58, 52, ... : 2-way conflicts inside 32 lanes)
//   mode 2: lane * 8 * 2 (every other cell)                          mode 3: random cells (hash of the lane)
__global__ __launch_bounds__(1024) void k_lds_b64(unsigned long long* __restrict__ t, float* __restrict__ sink, int mode)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    for (int i = threadIdx.x; i < 124 * 125 * 2; i += blockDim.x) lds[i] = (float)i;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    unsigned cell;
    if (mode == 0) cell = lane;
    else if (mode == 1) cell = lane * 125;
    else if (mode == 2) cell = lane * 2;
    else cell = (lane * 2654435761u >> 7) % (124u * 125u - 64u);
    unsigned addr = cell * 8 + (threadIdx.x >> 6) * 8 * 125 * 2;
    v2f acc = {0.0f, 0.0f};
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < kBlocks; ++i) {
        v2f v0, v1, v2, v3, v4, v5, v6, v7;
        asm volatile(
            ".rept 4\n"
            "ds_read_b64 %0, %8\n ds_read_b64 %1, %8 offset:8\n ds_read_b64 %2, %8 offset:16\n ds_read_b64 %3, %8 offset:24\n"
            "ds_read_b64 %4, %8 offset:32\n ds_read_b64 %5, %8 offset:40\n ds_read_b64 %6, %8 offset:48\n ds_read_b64 %7, %8 offset:56\n"
            ".endr\n s_waitcnt lgkmcnt(0)"
            : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3), "=&v"(v4), "=&v"(v5), "=&v"(v6), "=&v"(v7)
            : "v"(addr)
            : "memory");
        acc += v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7;
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if ((threadIdx.x & 63) == 0) t[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
    if (acc.x + acc.y == 12345.678f) sink[0] = acc.x;
}

struct Entry {
    const char* name;
    void (*fn)(unsigned long long*, float*, float);
};

int main()
{
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    unsigned long long* d_t;
    float* d_sink;
    CHECK(hipMalloc(&d_t, sizeof(unsigned long long) * cus * 8 * 16));
    CHECK(hipMalloc(&d_sink, 64));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const Entry entries[] = {
        {"v_fma_f32", k_fma}, {"v_add_f32", k_add}, {"v_mul_f32", k_mul}, {"v_max_f32", k_max}, {"v_fract_f32", k_fract},
        {"v_floor_f32", k_floor}, {"v_cvt_i32_f32", k_cvt_i32_f32}, {"v_cvt_f32_i32", k_cvt_f32_i32}, {"v_mul_u32_u24", k_mul_u32_u24},
        {"v_mad_u32_u24", k_mad_u32_u24}, {"v_mul_lo_u32", k_mul_lo_u32}, {"v_lshl_add_u32", k_lshl_add_u32}, {"v_add_u32", k_add_u32},
        {"v_and_b32", k_and_b32}, {"v_cndmask_b32", k_cndmask}, {"v_rcp_f32", k_rcp}, {"v_sqrt_f32", k_sqrt}, {"v_mov_b32", k_mov},
        {"v_cmp_lt_f32 -> vcc", k_cmp_vcc}, {"v_cmp_lt_f32 -> sgpr pair", k_cmp_sgpr},
        {"v_cmp(vcc) + s_nop 1 + v_cndmask(vcc) [2 VALU]", k_cmp_cndmask_vcc},
        {"v_cmp(sgpr) + s_nop 1 + v_cndmask(sgpr) [2 VALU]", k_cmp_cndmask_sgpr}, {"v_cndmask_b32 (sgpr pair mask)", k_cndmask_sgpr},
        {"v_med3_f32", k_med3}, {"v_min_f32", k_min}, {"v_max3_f32", k_max3}, {"v_sub_f32", k_sub}, {"v_xor_b32", k_xor},
        {"v_lshlrev_b32", k_lshl}, {"v_bfe_u32", k_bfe}, {"v_add3_u32", k_add3}, {"v_mov_b32 dpp row_shr:1", k_dpp},
        {"ds_bpermute_b32 + wait", k_bperm},
        {"v_pk_fma_f32", k_pk_fma}, {"v_pk_add_f32", k_pk_add}, {"v_pk_mul_f32", k_pk_mul}, {"v_fma_f64", k_fma_f64},
        {"v_add_f64", k_add_f64}, {"v_mul_f64", k_mul_f64}};
    const long insts = (long)kBlocks * kPerBlock;
    printf("{\"device\": \"%s\", \"cus\": %d, \"clock_mhz_reported\": %d, \"instructions_per_wave\": %ld,\n \"valu\": {\n", prop.gcnArchName, cus,
           prop.clockRate / 1000, insts);
    std::vector<unsigned long long> h(cus * 8 * 16);
    bool first = true;
    for (const Entry& e : entries) {
        printf("%s  \"%s\": {", first ? "" : ",\n", e.name);
        first = false;
        for (int k = 1; k <= 8; k *= 2) {
            const int grid = cus * k;
            e.fn<<<grid, 256>>>(d_t, d_sink, 1.0f);   // warm-up (clock ramp, code fetch)
            CHECK(hipEventRecord(e0));
            e.fn<<<grid, 256>>>(d_t, d_sink, 1.0f);
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float ms;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            CHECK(hipMemcpy(h.data(), d_t, sizeof(unsigned long long) * grid * 4, hipMemcpyDeviceToHost));
            double sum = 0;
            for (int i = 0; i < grid * 4; ++i) sum += (double)h[i];
            const double mean = sum / (grid * 4);
            const double cyc = mean / ((double)k * insts);
            const double wall_cyc_24 = ms * 1e-3 * 2.4e9 / ((double)k * insts);   // wall time priced at 2.4 GHz (includes launch overhead)
            printf("%s\"w%d\": {\"cyc_per_inst_memtime\": %.3f, \"wall_us\": %.1f, \"cyc_at_2.4GHz_wall\": %.3f}", k == 1 ? "" : ", ", k, cyc,
                   ms * 1e3, wall_cyc_24);
        }
        printf("}");
    }
    printf("\n },\n \"lds_ds_read_b64\": {\n");
    const char* modes[] = {"consecutive_cells", "stride125_cells", "stride2_cells", "random_cells"};
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_lds_b64), hipFuncAttributeMaxDynamicSharedMemorySize, 124 * 125 * 8));
    for (int m = 0; m < 4; ++m) {
        printf("%s  \"%s\": {", m ? ",\n" : "", modes[m]);
        for (int wg = 256; wg <= 1024; wg *= 2) {   // one workgroup per CU (the tile takes 124 KB): 1, 2, 4 waves per SIMD
            const int grid = cus, waves = wg / 64;
            hipLaunchKernelGGL(k_lds_b64, dim3(grid), dim3(wg), 124 * 125 * 8, 0, d_t, d_sink, m);
            CHECK(hipEventRecord(e0));
            hipLaunchKernelGGL(k_lds_b64, dim3(grid), dim3(wg), 124 * 125 * 8, 0, d_t, d_sink, m);
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float ms;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            CHECK(hipMemcpy(h.data(), d_t, sizeof(unsigned long long) * grid * waves, hipMemcpyDeviceToHost));
            double sum = 0;
            for (int i = 0; i < grid * waves; ++i) sum += (double)h[i];
            const double mean = sum / (grid * waves);
            const double reads = (double)kBlocks * 32;   // ds_read_b64 wave-instructions per wave
            printf("%s\"waves_per_cu_%d\": {\"cyc_per_wave_read_per_cu\": %.3f, \"bytes_per_clk_per_cu\": %.1f}", wg == 256 ? "" : ", ", waves,
                   mean / (waves * reads), 512.0 * waves * reads / mean);
        }
        printf("}");
    }
    printf("\n }\n}\n");
    return 0;
}
