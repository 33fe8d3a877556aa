// loopdb.hip -- device-resident, append-as-you-go descriptor database of one robot (C ABI mrs_loopdb_*).
//
// What it replaces: the Python lists the LoopDetection nodes keep per robot and walk for every new scan --
// TIRING1/2/3 + `for idx in range(len(pc_candidates)): fast_corr(TIRING_current, TIRING_candidates[idx])`
// (RING_ros/main_RING.py:126-140, 284-288), the RING++ twin (main_RINGplusplus.py:126-134) and the DiSCO node's
// `KDTree(np.array(DiSCO_candidates)).query(...)` + `phase_corr(FFT_candidates[idx], fft_current)` (disco_ros/main.py:276-291).
// There one descriptor is appended per callback (host tensors) and a query costs one torch FFT correlation per stored
// entry; here the entries live in HBM in the format the sweep kernels stream (RING: DMA-tiled half spectra, ringfft.hip),
// an append is one small copy / transform into the next slot (capacity doubles when it runs out: no re-upload), and a
// query is ONE sweep launch + one copy of (dist, angle) per entry + the `dist < threshold` filter in index order.
// All work of a handle runs on the handle's own stream; calls are serialised by a mutex (the reference's callbacks run on
// concurrent rospy threads, each appending to its own list and reading the other robots').
#include "common.hpp"
#include "fft_codelets.hpp"

#include <algorithm>
#include <cmath>
#include <cstdlib>

namespace {

constexpr int kA = 120, kD = 120, kHalf = 61;
constexpr size_t kSpecFloats = (size_t)kHalf * kD * 2;       // one [61][120] complex64 plane
constexpr size_t kTiledFloats = MRS_RING_TILED_ENTRY_BYTES / 4;
constexpr int kStageSlots = 4;

struct Pinned {
    void* p = nullptr;
    hipEvent_t ev = nullptr;
    bool used = false;
};

// ---- DiSCO query kernels ---------------------------------------------------------------------------------------------
// nearest signature by squared L2 distance in the difference form (what a kd-tree's L2 metric evaluates); one wave per signature
// at a time, 4 KiB per signature streamed as 4 x float4 per lane; winners merged with a 64-bit atomicMin on (distance bits, index).
__global__ __launch_bounds__(256) void k_sig_nearest(const float* __restrict__ q, const float* __restrict__ db, int n, int dim,
                                                     unsigned long long* __restrict__ best)
{
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = gridDim.x * 4;
    unsigned long long mine = ~0ull;
    for (int i = wave; i < n; i += nwaves) {
        const float* r = db + (size_t)i * dim;
        float acc = 0.0f;
        for (int c = lane * 4; c < dim; c += 256) {
            const float4 a = *reinterpret_cast<const float4*>(q + c);
            const float4 b = *reinterpret_cast<const float4*>(r + c);
            float d = a.x - b.x; acc = __builtin_fmaf(d, d, acc);
            d = a.y - b.y; acc = __builtin_fmaf(d, d, acc);
            d = a.z - b.z; acc = __builtin_fmaf(d, d, acc);
            d = a.w - b.w; acc = __builtin_fmaf(d, d, acc);
        }
        for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
        const unsigned long long key = ((unsigned long long)__float_as_uint(acc) << 32) | (unsigned)i;
        mine = key < mine ? key : mine;
    }
    if (lane == 0 && mine != ~0ull) atomicMin(best, mine);
}

// phase_corr(a = candidate, b = current) for the ONE candidate the search picked (disco_ros/main.py:260-272, 288-291):
// corr = ifft2(a conj(b), ortho), |corr| (+ 1e-15 under the root), fftshift2d, first maximum.  One workgroup, everything in LDS, 40 x 120:
// rows: 120-point inverse transforms as two in-register 60-point codelets (even / odd samples; 80 lanes) + one radix-2 step;
// columns: 40-point direct transforms (192 k complex multiply-adds spread over 960 work items, the lane's 40 inputs in registers).
constexpr int kPR = 40, kPS = 120, kPhaseThreads = 512;
__global__ __launch_bounds__(kPhaseThreads) void k_disco_phase_one(const float2* __restrict__ spectra, unsigned long long* __restrict__ best,
                                                                   const float2* __restrict__ cur, const float2* __restrict__ tw,
                                                                   int* __restrict__ out_index, float* __restrict__ out_d2, int* __restrict__ out_arg)
{
    extern __shared__ float2 sm[];        // A [R][S] | B [R][S] | twiddles exp(+2 pi i k / S), k < S | exp(+2 pi i k / R), k < R
    float2* A = sm;
    float2* B = sm + kPR * kPS;
    float2* twS = B + kPR * kPS;
    float2* twR = twS + kPS;
    __shared__ float bv[kPhaseThreads / 64];
    __shared__ int bi[kPhaseThreads / 64];
    const int t = threadIdx.x;
    const unsigned long long won = *best;                      // (distance bits, index) of the nearest signature
    const int idx = (int)(unsigned)(won & 0xffffffffull);
    const float2* a = spectra + (size_t)idx * kPR * kPS;
    for (int i = t; i < kPR * kPS; i += kPhaseThreads) {
        const float2 x = a[i], y = cur[i];                     // x conj(y)
        A[i] = make_float2(x.x * y.x + x.y * y.y, x.y * y.x - x.x * y.y);
    }
    for (int i = t; i < kPS; i += kPhaseThreads) twS[i] = tw[i];
    for (int i = t; i < kPR; i += kPhaseThreads) twR[i] = tw[kPS + i];
    __syncthreads();
    if (t < 2 * kPR) {                                         // E / O: 60-point inverse transforms of a row's even / odd samples
        const int r = t >> 1, par = t & 1;
        v2f x[60];
#pragma unroll
        for (int m = 0; m < 60; ++m) { const float2 v = A[r * kPS + 2 * m + par]; x[m] = (v2f){v.x, v.y}; }
        cfft60_inv(x);
#pragma unroll
        for (int k = 0; k < 60; ++k) B[r * kPS + par * 60 + k] = make_float2(x[k].x, x[k].y);
    }
    __syncthreads();
    for (int o = t; o < kPR * kPS; o += kPhaseThreads) {      // X[n] = E[n mod 60] + w^n O[n mod 60]
        const int r = o / kPS, n = o - r * kPS, k = n >= 60 ? n - 60 : n;
        const float2 e = B[r * kPS + k], od = B[r * kPS + 60 + k], w = twS[n];
        A[o] = make_float2(e.x + (w.x * od.x - w.y * od.y), e.y + (w.x * od.y + w.y * od.x));
    }
    __syncthreads();
    const float scale = 1.0f / sqrtf((float)(kPR * kPS));
    float bestv = -1.0f;
    int bidx = 0x7fffffff;
    for (int wi = t; wi < 8 * kPS; wi += kPhaseThreads) {     // work item = (column n, 5 consecutive output rows)
        const int g = wi / kPS, n = wi - g * kPS;
        float2 in[kPR];
#pragma unroll
        for (int j = 0; j < kPR; ++j) in[j] = A[j * kPS + n];
        for (int mm = 0; mm < 5; ++mm) {
            const int m = g * 5 + mm;
            float re = 0.0f, im = 0.0f;
            int k = 0;
#pragma unroll
            for (int j = 0; j < kPR; ++j) {
                const float2 w = twR[k];
                re = __builtin_fmaf(in[j].x, w.x, __builtin_fmaf(-in[j].y, w.y, re));
                im = __builtin_fmaf(in[j].x, w.y, __builtin_fmaf(in[j].y, w.x, im));
                k += m; if (k >= kPR) k -= kPR;
            }
            re *= scale; im *= scale;
            const float tot = sqrtf(re * re + im * im + 1e-15f);
            // fftshift2d (even sizes): source (m, n) appears at ((m + R/2) % R, (n + S/2) % S) of the shifted map
            const int flat = ((m + kPR / 2) % kPR) * kPS + (n + kPS / 2) % kPS;
            if (tot > bestv || (tot == bestv && flat < bidx)) { bestv = tot; bidx = flat; }
        }
    }
    const int wave = t >> 6, lane = t & 63;
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(bestv, o, 64);
        const int oi = __shfl_xor(bidx, o, 64);
        if (ov > bestv || (ov == bestv && oi < bidx)) { bestv = ov; bidx = oi; }
    }
    if (lane == 0) { bv[wave] = bestv; bi[wave] = bidx; }
    __syncthreads();
    if (t == 0) {
        for (int w = 1; w < kPhaseThreads / 64; ++w)
            if (bv[w] > bestv || (bv[w] == bestv && bi[w] < bidx)) { bestv = bv[w]; bidx = bi[w]; }
        // the three results go straight to pinned host memory (no copy after the kernel: the stream synchronisation that follows makes them
        // visible), and `best` is armed for the next query (no memset before it)
        *out_arg = bidx;
        *out_index = idx;
        *out_d2 = __uint_as_float((unsigned)(won >> 32));
        __threadfence_system();
        *best = ~0ull;
    }
}

}  // namespace

struct mrs_loopdb {
    mrs_ctx* ctx = nullptr;
    int kind = 0, channels = 1;
    int sig_dim = 0, R = 0, S = 0;            // DiSCO
    int n = 0, cap = 0;
    size_t entry_floats = 0;                  // floats from one entry of d_entries to the next
    size_t in_floats = 0;                     // floats of a descriptor in the reference's form (what append / query take)
    float* d_entries = nullptr;               // RING / RING++: [cap + 1][C] DMA-tiled planes (one entry of slack); DiSCO: spectra [cap][R][S] complex64
    float* d_sigs = nullptr;                  // DiSCO: [cap][dim]
    float* d_dist = nullptr;                  // [2 cap]: [n] distances, [n] angles
    int32_t* d_angle = nullptr;               // = d_dist + n of the query at hand (one allocation, one copy)
    float* h_dist = nullptr;                  // pinned [2 cap]
    int32_t* h_angle = nullptr;
    float* d_in = nullptr;                    // one descriptor in the reference's form (device copy of a host argument)
    float* d_tmp = nullptr;                   // RING++: the normalised channels
    float* d_query = nullptr;                 // the query in database form (row layout: [C][61][120] complex64)
    float* d_tw = nullptr;                    // DiSCO: exp(+2 pi i k / S), k < S, then exp(+2 pi i k / R), k < R
    unsigned long long* d_best = nullptr;     // DiSCO: packed (distance bits, index)
    int32_t* d_small = nullptr;               // DiSCO: index, distance bits, argmax
    int32_t* h_small = nullptr;               // pinned
    bool phase_attr_set = false;              // DiSCO: the phase kernel's LDS attribute has been set on this handle's device
    Pinned stage[kStageSlots];
    int stage_next = 0;
    size_t stage_bytes = 0;
    hipStream_t s = nullptr;
    hipEvent_t ev_in = nullptr, ev_out = nullptr;
    std::mutex mu;
};

namespace {

void free_arrays(mrs_loopdb* db)
{
    if (db->d_entries) (void)hipFree(db->d_entries);
    if (db->d_sigs) (void)hipFree(db->d_sigs);
    if (db->d_dist) (void)hipFree(db->d_dist);
    if (db->h_dist) (void)hipHostFree(db->h_dist);
    db->d_entries = db->d_sigs = db->d_dist = nullptr;
    db->d_angle = nullptr; db->h_dist = nullptr; db->h_angle = nullptr;
}

// capacity >= want (doubling): new arrays, device-to-device copy of what is there, old ones freed after the copy; lock held
int reserve_locked(mrs_loopdb* db, int want)
{
    if (want <= db->cap) return MRS_OK;
    int cap = std::max(db->cap, 256);
    while (cap < want) cap *= 2;
    float *ne = nullptr, *ns = nullptr, *nd = nullptr, *hd = nullptr;
    int32_t *na = nullptr, *ha = nullptr;
    const size_t slack = db->kind == MRS_LOOPDB_DISCO ? 0 : 1;   // the tiled sweep reads up to 1 KiB past the last entry
    // every failure path releases what this call has allocated so far (the out-of-memory case must not get worse on retry)
    auto grow = [&]() -> hipError_t {
        hipError_t e = hipMalloc(&ne, ((size_t)cap + slack) * db->entry_floats * sizeof(float));
        if (e != hipSuccess) return e;
        if (slack && (e = hipMemsetAsync(ne + (size_t)cap * db->entry_floats, 0, db->entry_floats * sizeof(float), db->s)) != hipSuccess) return e;
        if (db->kind == MRS_LOOPDB_DISCO && (e = hipMalloc(&ns, (size_t)cap * db->sig_dim * sizeof(float))) != hipSuccess) return e;
        // distances and angles of a query lie back to back ([n] floats, [n] ints: the angles start at element n, wherever n stands) so that ONE
        // copy brings both to the host; d_dist / h_dist own the 2 x cap elements, d_angle / h_angle are not separate allocations
        if ((e = hipMalloc(&nd, (size_t)2 * cap * sizeof(float))) != hipSuccess) return e;
        if ((e = hipHostMalloc(&hd, (size_t)2 * cap * sizeof(float), hipHostMallocDefault)) != hipSuccess) return e;
        if (db->n > 0) {
            if ((e = hipMemcpyAsync(ne, db->d_entries, (size_t)db->n * db->entry_floats * sizeof(float), hipMemcpyDeviceToDevice, db->s)) != hipSuccess) return e;
            if (ns && (e = hipMemcpyAsync(ns, db->d_sigs, (size_t)db->n * db->sig_dim * sizeof(float), hipMemcpyDeviceToDevice, db->s)) != hipSuccess) return e;
        }
        return hipStreamSynchronize(db->s);
    };
    const hipError_t e = grow();
    if (e != hipSuccess) {
        (void)hipStreamSynchronize(db->s);
        if (ne) (void)hipFree(ne);
        if (ns) (void)hipFree(ns);
        if (nd) (void)hipFree(nd);
        if (hd) (void)hipHostFree(hd);
        mrs::set_error("growing a loop database to %d entries failed: %s", cap, hipGetErrorString(e));
        return MRS_ERR_HIP;
    }
    free_arrays(db);
    db->d_entries = ne; db->d_sigs = ns; db->d_dist = nd; db->d_angle = na; db->h_dist = hd; db->h_angle = ha;
    db->cap = cap;
    return MRS_OK;
}

// the caller's stream has produced a device argument: the handle's stream waits for it
int join_in(mrs_loopdb* db, hipStream_t user)
{
    MRS_HIP_TRY(hipEventRecord(db->ev_in, user));
    MRS_HIP_TRY(hipStreamWaitEvent(db->s, db->ev_in, 0));
    return MRS_OK;
}

// host argument -> device buffer `dst` through a pinned slot (the caller's memory is free again when this returns)
int upload(mrs_loopdb* db, const void* h_src, size_t bytes, void* dst)
{
    Pinned& p = db->stage[db->stage_next];
    db->stage_next = (db->stage_next + 1) % kStageSlots;
    if (p.used) MRS_HIP_TRY(hipEventSynchronize(p.ev));
    memcpy(p.p, h_src, bytes);
    MRS_HIP_TRY(hipMemcpyAsync(dst, p.p, bytes, hipMemcpyHostToDevice, db->s));
    MRS_HIP_TRY(hipEventRecord(p.ev, db->s));
    p.used = true;
    return MRS_OK;
}

// a RING / RING++ descriptor in the given form -> row-layout half spectra [C][61][120] at `d_spec` (device), on the handle's stream
int to_half_spectrum(mrs_loopdb* db, const void* desc, int form, float* d_spec, hipStream_t user)
{
    const int C = db->channels;
    if (form == MRS_LOOPDB_FORM_DEVICE_SPEC) {
        int st = join_in(db, user);
        if (st != MRS_OK) return st;
        MRS_HIP_TRY(hipMemcpyAsync(d_spec, desc, (size_t)C * kSpecFloats * sizeof(float), hipMemcpyDeviceToDevice, db->s));
        return MRS_OK;
    }
    if (db->kind == MRS_LOOPDB_RING) {
        // the reference's TIRING [1][120][120] complex64 (util.py:198): its first 61 angle-frequency rows ARE the half spectrum
        if (form == MRS_LOOPDB_FORM_HOST) return upload(db, desc, kSpecFloats * sizeof(float), d_spec);
        int st = join_in(db, user);
        if (st != MRS_OK) return st;
        MRS_HIP_TRY(hipMemcpyAsync(d_spec, desc, kSpecFloats * sizeof(float), hipMemcpyDeviceToDevice, db->s));
        return MRS_OK;
    }
    // RING++: the reference's TIRING [C][120][120] float32 magnitudes; fast_corr_RINGplusplus normalises jointly and transforms along the
    // angle axis at every comparison (util.py:339-343) -- done once here
    const float* src = static_cast<const float*>(desc);
    if (form == MRS_LOOPDB_FORM_HOST) {
        int st = upload(db, desc, db->in_floats * sizeof(float), db->d_in);
        if (st != MRS_OK) return st;
        src = db->d_in;
    } else {
        int st = join_in(db, user);
        if (st != MRS_OK) return st;
    }
    int st = mrs_normalize_groups(db->ctx, src, db->d_tmp, 1, C * kA * kD, db->s);
    if (st != MRS_OK) return st;
    return mrs_ring_half_spectrum(db->ctx, db->d_tmp, C, kA, kD, d_spec, db->s);
}

}  // namespace

extern "C" {

int mrs_loopdb_create(mrs_ctx* ctx, int32_t kind, int32_t channels, int32_t capacity_hint, mrs_loopdb** out)
{
    MRS_REQUIRE(ctx && out, "null pointer");
    *out = nullptr;
    MRS_REQUIRE(kind == MRS_LOOPDB_RING || kind == MRS_LOOPDB_RINGPP || kind == MRS_LOOPDB_DISCO, "unknown kind");
    MRS_REQUIRE(kind != MRS_LOOPDB_RING || channels == 1, "RING descriptors have one channel");
    MRS_REQUIRE(channels >= 1 && channels <= 16, "channels out of range");
    MRS_HIP_TRY(hipSetDevice(ctx->device));
    mrs_loopdb* db = new mrs_loopdb();
    db->ctx = ctx; db->kind = kind; db->channels = channels;
    if (kind == MRS_LOOPDB_RING) { db->entry_floats = kTiledFloats; db->in_floats = (size_t)kA * kD * 2; }
    else if (kind == MRS_LOOPDB_RINGPP) { db->entry_floats = (size_t)channels * kTiledFloats; db->in_floats = (size_t)channels * kA * kD; }
    else { db->sig_dim = 1024; db->R = kPR; db->S = kPS; db->entry_floats = (size_t)db->R * db->S * 2; db->in_floats = db->entry_floats; }
    auto fail = [&](int st) { mrs_loopdb_destroy(db); return st; };
#define LDB_TRY(expr) do { if ((expr) != hipSuccess) { mrs::set_error("%s failed (%s:%d)", #expr, __FILE__, __LINE__); return fail(MRS_ERR_HIP); } } while (0)
    LDB_TRY(hipStreamCreateWithFlags(&db->s, hipStreamNonBlocking));
    LDB_TRY(hipEventCreateWithFlags(&db->ev_in, hipEventDisableTiming));
    LDB_TRY(hipEventCreateWithFlags(&db->ev_out, hipEventDisableTiming));
    db->stage_bytes = (kind == MRS_LOOPDB_DISCO ? db->in_floats + (size_t)db->sig_dim : db->in_floats) * sizeof(float);      // DiSCO: signature | spectrum in one slot
    for (Pinned& p : db->stage) {
        LDB_TRY(hipHostMalloc(&p.p, db->stage_bytes, hipHostMallocDefault));
        LDB_TRY(hipEventCreateWithFlags(&p.ev, hipEventDisableTiming));
    }
    LDB_TRY(hipMalloc(&db->d_in, (db->in_floats + (size_t)db->sig_dim) * sizeof(float)));
    LDB_TRY(hipMalloc(&db->d_tmp, db->in_floats * sizeof(float)));
    LDB_TRY(hipMalloc(&db->d_query, std::max((size_t)channels * kSpecFloats, db->entry_floats) * sizeof(float)));
    if (kind == MRS_LOOPDB_DISCO) {
        std::vector<float> tw(2 * (size_t)(db->S + db->R));
        for (int k = 0; k < db->S; ++k) { tw[2 * k] = (float)cos(2.0 * M_PI * k / db->S); tw[2 * k + 1] = (float)sin(2.0 * M_PI * k / db->S); }
        for (int k = 0; k < db->R; ++k) { tw[2 * (db->S + k)] = (float)cos(2.0 * M_PI * k / db->R); tw[2 * (db->S + k) + 1] = (float)sin(2.0 * M_PI * k / db->R); }
        LDB_TRY(hipMalloc(&db->d_tw, tw.size() * sizeof(float)));
        LDB_TRY(hipMemcpy(db->d_tw, tw.data(), tw.size() * sizeof(float), hipMemcpyHostToDevice));
        LDB_TRY(hipMalloc(&db->d_best, sizeof(unsigned long long)));
        LDB_TRY(hipMemset(db->d_best, 0xff, sizeof(unsigned long long)));       // armed; k_disco_phase_one re-arms it after every query
        LDB_TRY(hipMalloc(&db->d_small, 4 * sizeof(int32_t)));
        LDB_TRY(hipHostMalloc(&db->h_small, 4 * sizeof(int32_t), hipHostMallocDefault));
    }
#undef LDB_TRY
    {
        std::lock_guard<std::mutex> lk(db->mu);
        const int st = reserve_locked(db, std::max(capacity_hint, 1));
        if (st != MRS_OK) return fail(st);
    }
    *out = db;
    return MRS_OK;
}

int mrs_loopdb_destroy(mrs_loopdb* db)
{
    if (!db) return MRS_OK;
    (void)hipSetDevice(db->ctx->device);
    if (db->s) (void)hipStreamSynchronize(db->s);
    free_arrays(db);
    for (Pinned& p : db->stage) {
        if (p.p) (void)hipHostFree(p.p);
        if (p.ev) (void)hipEventDestroy(p.ev);
    }
    if (db->d_in) (void)hipFree(db->d_in);
    if (db->d_tmp) (void)hipFree(db->d_tmp);
    if (db->d_query) (void)hipFree(db->d_query);
    if (db->d_tw) (void)hipFree(db->d_tw);
    if (db->d_best) (void)hipFree(db->d_best);
    if (db->d_small) (void)hipFree(db->d_small);
    if (db->h_small) (void)hipHostFree(db->h_small);
    if (db->ev_in) (void)hipEventDestroy(db->ev_in);
    if (db->ev_out) (void)hipEventDestroy(db->ev_out);
    if (db->s) (void)hipStreamDestroy(db->s);
    delete db;
    return MRS_OK;
}

int mrs_loopdb_size(mrs_loopdb* db, int32_t* out_n)
{
    MRS_REQUIRE(db && out_n, "null pointer");
    std::lock_guard<std::mutex> lk(db->mu);
    *out_n = db->n;
    return MRS_OK;
}

int mrs_loopdb_reserve(mrs_loopdb* db, int32_t capacity)
{
    MRS_REQUIRE(db, "null pointer");
    MRS_HIP_TRY(hipSetDevice(db->ctx->device));
    std::lock_guard<std::mutex> lk(db->mu);
    return reserve_locked(db, capacity);
}

int mrs_loopdb_clear(mrs_loopdb* db)
{
    MRS_REQUIRE(db, "null pointer");
    std::lock_guard<std::mutex> lk(db->mu);
    db->n = 0;
    return MRS_OK;
}

int mrs_loopdb_append(mrs_loopdb* db, const void* descriptor, int32_t form, int32_t count, mrs_stream stream)
{
    MRS_REQUIRE(db && descriptor, "null pointer");
    MRS_REQUIRE(db->kind != MRS_LOOPDB_DISCO, "DiSCO entries are appended with mrs_loopdb_append_disco");
    MRS_REQUIRE(form == MRS_LOOPDB_FORM_HOST || form == MRS_LOOPDB_FORM_DEVICE || form == MRS_LOOPDB_FORM_DEVICE_SPEC, "unknown form");
    MRS_REQUIRE(count >= 1 && (count == 1 || form == MRS_LOOPDB_FORM_DEVICE_SPEC), "several entries per call only as device half spectra");
    MRS_HIP_TRY(hipSetDevice(db->ctx->device));
    std::lock_guard<std::mutex> lk(db->mu);
    int st = reserve_locked(db, db->n + count);
    if (st != MRS_OK) return st;
    const int C = db->channels;
    const float* spec = db->d_query;
    if (form == MRS_LOOPDB_FORM_DEVICE_SPEC) {             // already row-layout half spectra on the device: permuted straight into the slots
        st = join_in(db, (hipStream_t)stream);
        if (st != MRS_OK) return st;
        spec = static_cast<const float*>(descriptor);
    } else {
        st = to_half_spectrum(db, descriptor, form, db->d_query, (hipStream_t)stream);
        if (st != MRS_OK) return st;
    }
    st = mrs_ring_spec_to_tiled(db->ctx, spec, count * C, db->d_entries + (size_t)db->n * db->entry_floats, db->s);
    if (st != MRS_OK) return st;
    if (form != MRS_LOOPDB_FORM_HOST) {   // the caller may overwrite its device buffer once ITS stream has passed this point
        MRS_HIP_TRY(hipEventRecord(db->ev_out, db->s));
        MRS_HIP_TRY(hipStreamWaitEvent((hipStream_t)stream, db->ev_out, 0));
    }
    db->n += count;
    return MRS_OK;
}

int mrs_loopdb_query(mrs_loopdb* db, const void* descriptor, int32_t form, float dist_threshold, int32_t max_out, int32_t* h_index,
                     float* h_dist, int32_t* h_angle, int32_t* h_count, int32_t all_capacity, float* h_all_dist, int32_t* h_all_angle, int32_t* h_n,
                     mrs_stream stream)
{
    MRS_REQUIRE(db && descriptor && h_count, "null pointer");
    MRS_REQUIRE(db->kind != MRS_LOOPDB_DISCO, "DiSCO databases are queried with mrs_loopdb_query_disco");
    MRS_REQUIRE(form == MRS_LOOPDB_FORM_HOST || form == MRS_LOOPDB_FORM_DEVICE || form == MRS_LOOPDB_FORM_DEVICE_SPEC, "unknown form");
    MRS_REQUIRE(max_out >= 0 && (max_out == 0 || (h_index && h_dist && h_angle)), "output arrays");
    MRS_REQUIRE(all_capacity >= 0 && (all_capacity == 0 || h_all_dist || h_all_angle), "all_capacity without an array");
    MRS_HIP_TRY(hipSetDevice(db->ctx->device));
    std::lock_guard<std::mutex> lk(db->mu);
    *h_count = 0;
    const int n = db->n;
    if (h_n) *h_n = n;                                     // the entries this call scored (another thread may append right after)
    if (n == 0) return MRS_OK;                             // `for idx in range(0)`: no candidates, no work
    int st = to_half_spectrum(db, descriptor, form, db->d_query, (hipStream_t)stream);
    if (st != MRS_OK) return st;
    db->d_angle = reinterpret_cast<int32_t*>(db->d_dist + n);
    db->h_angle = reinterpret_cast<int32_t*>(db->h_dist + n);
    st = mrs_ring_corr_fft_sweep_tiled(db->ctx, db->d_query, db->d_entries, n, db->channels, db->d_dist, db->d_angle, db->s);
    if (st != MRS_OK) return st;
    MRS_HIP_TRY(hipMemcpyAsync(db->h_dist, db->d_dist, (size_t)2 * n * sizeof(float), hipMemcpyDeviceToHost, db->s));
    MRS_HIP_TRY(hipStreamSynchronize(db->s));
    // `if dist < cfg.dist_threshold: idxs.append(idx) ...` in index order (main_RING.py:133-140)
    int cnt = 0;
    for (int i = 0; i < n; ++i)
        if (db->h_dist[i] < dist_threshold) {
            if (cnt < max_out) { h_index[cnt] = i; h_dist[cnt] = db->h_dist[i]; h_angle[cnt] = db->h_angle[i]; }
            ++cnt;
        }
    *h_count = cnt;                                         // may exceed max_out: the caller then asks again with larger arrays
    const size_t m = (size_t)std::min(n, all_capacity);     // never past the caller's arrays, whatever was appended since it sized them
    if (h_all_dist) memcpy(h_all_dist, db->h_dist, m * sizeof(float));
    if (h_all_angle) memcpy(h_all_angle, db->h_angle, m * sizeof(int32_t));
    return MRS_OK;
}

int mrs_loopdb_query_multi(mrs_loopdb* db, const void* descriptors, int32_t form, int32_t count, int32_t all_capacity, float* h_all_dist,
                           int32_t* h_all_angle, int32_t* h_n, mrs_stream stream)
{
    MRS_REQUIRE(db && descriptors && h_all_dist && h_all_angle && h_n, "null pointer");
    MRS_REQUIRE(db->kind != MRS_LOOPDB_DISCO, "DiSCO databases are queried with mrs_loopdb_query_disco");
    MRS_REQUIRE(form == MRS_LOOPDB_FORM_HOST || form == MRS_LOOPDB_FORM_DEVICE || form == MRS_LOOPDB_FORM_DEVICE_SPEC, "unknown form");
    MRS_REQUIRE(count >= 1 && count <= 1024 && all_capacity >= 0, "count in 1..1024");
    MRS_HIP_TRY(hipSetDevice(db->ctx->device));
    std::lock_guard<std::mutex> lk(db->mu);
    const int n = db->n;
    *h_n = n;
    if (n == 0) return MRS_OK;
    const int C = db->channels;
    const size_t spec_floats = (size_t)C * kSpecFloats;
    mrs::Scratch qbuf, out;
    int st = qbuf.alloc((size_t)count * spec_floats * sizeof(float), db->s);
    if (st != MRS_OK) return st;
    st = out.alloc((size_t)count * 2 * n * sizeof(float), db->s);
    if (st != MRS_OK) return st;
    float* d_q = qbuf.as<float>();
    if (form == MRS_LOOPDB_FORM_DEVICE_SPEC) {
        st = join_in(db, (hipStream_t)stream);
        if (st != MRS_OK) return st;
        MRS_HIP_TRY(hipMemcpyAsync(d_q, descriptors, (size_t)count * spec_floats * sizeof(float), hipMemcpyDeviceToDevice, db->s));
    } else {
        const size_t step = db->kind == MRS_LOOPDB_RING ? (size_t)kA * kD * 2 : db->in_floats;     // floats from one descriptor to the next in the caller's array
        for (int i = 0; i < count; ++i) {
            st = to_half_spectrum(db, static_cast<const float*>(descriptors) + (size_t)i * step, form, d_q + (size_t)i * spec_floats, (hipStream_t)stream);
            if (st != MRS_OK) return st;
        }
    }
    float* d_dist = out.as<float>();                                     // [count][n] distances, then [count][n] angles
    int32_t* d_angle = reinterpret_cast<int32_t*>(d_dist + (size_t)count * n);
    st = mrs_ring_corr_fft_sweep_tiled_q(db->ctx, d_q, count, db->d_entries, n, C, d_dist, d_angle, db->s);
    if (st != MRS_OK) return st;
    const size_t m = (size_t)std::min(n, all_capacity);
    if (m > 0) {
        MRS_HIP_TRY(hipMemcpy2DAsync(h_all_dist, (size_t)all_capacity * sizeof(float), d_dist, (size_t)n * sizeof(float), m * sizeof(float), count,
                                     hipMemcpyDeviceToHost, db->s));
        MRS_HIP_TRY(hipMemcpy2DAsync(h_all_angle, (size_t)all_capacity * sizeof(int32_t), d_angle, (size_t)n * sizeof(int32_t), m * sizeof(int32_t), count,
                                     hipMemcpyDeviceToHost, db->s));
    }
    if (form != MRS_LOOPDB_FORM_HOST) {
        MRS_HIP_TRY(hipEventRecord(db->ev_out, db->s));
        MRS_HIP_TRY(hipStreamWaitEvent((hipStream_t)stream, db->ev_out, 0));
    }
    MRS_HIP_TRY(hipStreamSynchronize(db->s));
    return MRS_OK;
}

int mrs_loopdb_append_disco(mrs_loopdb* db, const float* signature, const float* spectrum, int32_t on_device, mrs_stream stream)
{
    MRS_REQUIRE(db && signature && spectrum, "null pointer");
    MRS_REQUIRE(db->kind == MRS_LOOPDB_DISCO, "not a DiSCO database");
    MRS_HIP_TRY(hipSetDevice(db->ctx->device));
    std::lock_guard<std::mutex> lk(db->mu);
    int st = reserve_locked(db, db->n + 1);
    if (st != MRS_OK) return st;
    float* sig = db->d_sigs + (size_t)db->n * db->sig_dim;
    float* spec = db->d_entries + (size_t)db->n * db->entry_floats;
    if (on_device) {
        st = join_in(db, (hipStream_t)stream);
        if (st != MRS_OK) return st;
        MRS_HIP_TRY(hipMemcpyAsync(sig, signature, (size_t)db->sig_dim * sizeof(float), hipMemcpyDeviceToDevice, db->s));
        MRS_HIP_TRY(hipMemcpyAsync(spec, spectrum, db->entry_floats * sizeof(float), hipMemcpyDeviceToDevice, db->s));
        MRS_HIP_TRY(hipEventRecord(db->ev_out, db->s));
        MRS_HIP_TRY(hipStreamWaitEvent((hipStream_t)stream, db->ev_out, 0));
    } else {
        st = upload(db, signature, (size_t)db->sig_dim * sizeof(float), sig);
        if (st != MRS_OK) return st;
        st = upload(db, spectrum, db->entry_floats * sizeof(float), spec);
        if (st != MRS_OK) return st;
    }
    db->n += 1;
    return MRS_OK;
}

int mrs_loopdb_query_disco(mrs_loopdb* db, const float* signature, const float* spectrum, int32_t on_device, int32_t* h_index, float* h_dist2,
                           int32_t* h_flat_argmax, mrs_stream stream)
{
    MRS_REQUIRE(db && signature && spectrum && h_index && h_dist2 && h_flat_argmax, "null pointer");
    MRS_REQUIRE(db->kind == MRS_LOOPDB_DISCO, "not a DiSCO database");
    MRS_HIP_TRY(hipSetDevice(db->ctx->device));
    std::lock_guard<std::mutex> lk(db->mu);
    *h_index = -1; *h_dist2 = INFINITY; *h_flat_argmax = 0;
    const int n = db->n;
    if (n == 0) return MRS_OK;
    const float *sig = signature, *spec = spectrum;
    int st;
    if (on_device) {
        st = join_in(db, (hipStream_t)stream);
        if (st != MRS_OK) return st;
    } else {
        // signature | spectrum through ONE pinned slot and ONE copy
        Pinned& p = db->stage[db->stage_next];
        db->stage_next = (db->stage_next + 1) % kStageSlots;
        if (p.used) MRS_HIP_TRY(hipEventSynchronize(p.ev));
        const size_t sb = (size_t)db->sig_dim * sizeof(float), eb = db->entry_floats * sizeof(float);
        memcpy(p.p, signature, sb);
        memcpy(static_cast<char*>(p.p) + sb, spectrum, eb);
        MRS_HIP_TRY(hipMemcpyAsync(db->d_in, p.p, sb + eb, hipMemcpyHostToDevice, db->s));
        MRS_HIP_TRY(hipEventRecord(p.ev, db->s));
        p.used = true;
        sig = db->d_in; spec = db->d_in + db->sig_dim;
    }
    const int blocks = std::max(1, std::min((n + 3) / 4, 4 * (db->ctx->num_cu > 0 ? db->ctx->num_cu : 256)));
    hipLaunchKernelGGL(k_sig_nearest, dim3(blocks), dim3(256), 0, db->s, sig, db->d_sigs, n, db->sig_dim, db->d_best);
    const size_t lds = (size_t)(2 * db->R * db->S + db->S + db->R) * sizeof(float2);
    if (!db->phase_attr_set) {
        MRS_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_disco_phase_one), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        db->phase_attr_set = true;
    }
    hipLaunchKernelGGL(k_disco_phase_one, dim3(1), dim3(kPhaseThreads), lds, db->s, reinterpret_cast<const float2*>(db->d_entries), db->d_best,
                       reinterpret_cast<const float2*>(spec), reinterpret_cast<const float2*>(db->d_tw), db->h_small,
                       reinterpret_cast<float*>(db->h_small + 1), db->h_small + 2);
    MRS_HIP_TRY(hipGetLastError());
    MRS_HIP_TRY(hipStreamSynchronize(db->s));
    *h_index = db->h_small[0];
    memcpy(h_dist2, &db->h_small[1], sizeof(float));
    *h_flat_argmax = db->h_small[2];
    return MRS_OK;
}

int mrs_loopdb_device_entries(mrs_loopdb* db, const float** d_entries, const float** d_signatures, int32_t* out_n, int64_t* entry_floats)
{
    MRS_REQUIRE(db && d_entries && out_n, "null pointer");
    std::lock_guard<std::mutex> lk(db->mu);
    MRS_HIP_TRY(hipSetDevice(db->ctx->device));
    MRS_HIP_TRY(hipStreamSynchronize(db->s));
    *d_entries = db->d_entries;
    if (d_signatures) *d_signatures = db->d_sigs;
    *out_n = db->n;
    if (entry_floats) *entry_floats = (int64_t)db->entry_floats;
    return MRS_OK;
}

}  // extern "C"
