// capi.hip -- context management and status reporting of the C ABI (include/mrslam_hip.h).
#include "common.hpp"

#include <cstdlib>
#include <string>

namespace mrs {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

const char* dev_env(const char* name)
{
    const char* v = getenv(name);
    if (!v) return nullptr;
    const char* d = getenv("MRS_DEV");
    static std::mutex mu;
    static std::map<std::string, bool> told;
    std::lock_guard<std::mutex> lk(mu);
    const bool on = d && atoi(d) == 1;
    if (!told[name]) {
        told[name] = true;
        if (on) fprintf(stderr, "[mrslam] development switch %s=%s is ACTIVE (MRS_DEV=1): outputs and timings of this process are not production results\n", name, v);
        else fprintf(stderr, "[mrslam] %s is set but ignored: development switches need MRS_DEV=1\n", name);
    }
    return on ? v : nullptr;
}

int side_acquire(mrs_ctx* ctx, SideSlot* out)
{
    {
        std::lock_guard<std::mutex> lk(ctx->side_mu);
        if (!ctx->side_free.empty()) {
            *out = ctx->side_free.back();
            ctx->side_free.pop_back();
            return MRS_OK;
        }
    }
    SideSlot sl;
    hipError_t e = hipStreamCreateWithFlags(&sl.stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&sl.fork, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&sl.join, hipEventDisableTiming);
    if (e == hipSuccess) e = hipHostMalloc(&sl.pinned, kSidePinnedInts * sizeof(int), hipHostMallocDefault);
    if (e != hipSuccess) {
        if (sl.pinned) (void)hipHostFree(sl.pinned);
        if (sl.join) (void)hipEventDestroy(sl.join);
        if (sl.fork) (void)hipEventDestroy(sl.fork);
        if (sl.stream) (void)hipStreamDestroy(sl.stream);
        set_error("side stream slot: %s", hipGetErrorString(e));
        return MRS_ERR_HIP;
    }
    *out = sl;
    return MRS_OK;
}

void side_release(mrs_ctx* ctx, const SideSlot& slot)
{
    if (!slot.stream) return;
    std::lock_guard<std::mutex> lk(ctx->side_mu);
    ctx->side_free.push_back(slot);
}

// ---- scratch allocator (see common.hpp) -------------------------------------------------------------------------
namespace {
struct ScratchBlock {
    void* p;
    size_t cap;
    int device;
    hipStream_t last_stream;
    hipEvent_t ev;       // recorded on last_stream at release
    bool busy;
};
std::mutex g_scratch_mu;
std::vector<ScratchBlock> g_scratch;   // a few dozen blocks at most: linear search

// idle blocks kept for reuse: MRS_SCRATCH_IDLE_CAP_MB (default 8192).  A process that shares the GPU with another allocator
// (torch's) can set it low; 0 returns every block as soon as its work is done
size_t idle_cap()
{
    static const size_t cap = [] {
        const char* e = getenv("MRS_SCRATCH_IDLE_CAP_MB");
        const long long mb = e ? atoll(e) : 8192;
        return (size_t)(mb < 0 ? 0 : mb) << 20;
    }();
    return cap;
}

// give idle blocks whose last use has completed back to the driver (all of them, or until `keep` bytes remain idle); lock held
void trim_idle_locked(size_t keep, const void* spare)
{
    size_t idle = 0;
    for (const ScratchBlock& b : g_scratch) idle += b.busy ? 0 : b.cap;
    for (size_t i = 0; i < g_scratch.size() && idle > keep;) {
        ScratchBlock& b = g_scratch[i];
        if (!b.busy && b.p != spare && hipEventQuery(b.ev) == hipSuccess) {
            idle -= b.cap;
            (void)hipFree(b.p);
            (void)hipEventDestroy(b.ev);
            g_scratch.erase(g_scratch.begin() + i);
        } else {
            ++i;
        }
    }
}
}  // namespace

void* scratch_acquire(size_t bytes, hipStream_t stream)
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    const size_t want = (bytes + 255) & ~(size_t)255;
    {
        std::lock_guard<std::mutex> lock(g_scratch_mu);
        int best = -1;
        for (int i = 0; i < (int)g_scratch.size(); ++i) {
            ScratchBlock& b = g_scratch[i];
            if (b.busy || b.device != dev || b.cap < want || b.cap > 4 * want + (1 << 20)) continue;
            // same stream: stream order protects the previous user's work; other stream: only once that work is done
            if (b.last_stream != stream && hipEventQuery(b.ev) != hipSuccess) continue;
            if (best < 0 || b.cap < g_scratch[best].cap) best = i;
        }
        if (best >= 0) {
            g_scratch[best].busy = true;
            return g_scratch[best].p;
        }
    }
    ScratchBlock nb{nullptr, want, dev, stream, nullptr, true};
    if (hipMalloc(&nb.p, want) != hipSuccess) {
        // out of memory with idle blocks in the cache: return every finished one to the driver and try once more
        (void)hipGetLastError();
        {
            std::lock_guard<std::mutex> lock(g_scratch_mu);
            trim_idle_locked(0, nullptr);
        }
        if (hipMalloc(&nb.p, want) != hipSuccess) return nullptr;
    }
    if (hipEventCreateWithFlags(&nb.ev, hipEventDisableTiming) != hipSuccess) { (void)hipFree(nb.p); return nullptr; }
    std::lock_guard<std::mutex> lock(g_scratch_mu);
    g_scratch.push_back(nb);
    return nb.p;
}

void scratch_release(void* p, hipStream_t stream)
{
    std::lock_guard<std::mutex> lock(g_scratch_mu);
    for (ScratchBlock& b : g_scratch)
        if (b.p == p) {
            b.last_stream = stream;
            (void)hipEventRecord(b.ev, stream);
            b.busy = false;
            break;
        }
    // keep the cache bounded: beyond the idle cap, give finished blocks back (not the one just released: its work is in flight)
    trim_idle_locked(idle_cap(), p);
}

}  // namespace mrs

extern "C" {

int mrs_abi_version(void) { return MRS_ABI_VERSION; }

const char* mrs_status_str(int status)
{
    switch (status) {
        case MRS_OK: return "ok";
        case MRS_ERR_ARG: return "bad argument";
        case MRS_ERR_HIP: return "HIP runtime error";
        case MRS_ERR_UNSUPPORTED: return "unsupported configuration";
        case MRS_ERR_NO_DEVICE: return "no HIP device";
        case MRS_ERR_NOT_CONVERGED: return "not converged";
        default: return "unknown status";
    }
}

const char* mrs_last_error(void) { return mrs::g_err; }

int mrs_ctx_create(int device, mrs_ctx** out_ctx)
{
    MRS_REQUIRE(out_ctx != nullptr, "out_ctx");
    *out_ctx = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) {
        mrs::set_error("no HIP device visible (this library has no CPU fallback)");
        return MRS_ERR_NO_DEVICE;
    }
    MRS_REQUIRE(device >= 0 && device < count, "device index out of range");
    MRS_HIP_TRY(hipSetDevice(device));
    hipDeviceProp_t prop;
    MRS_HIP_TRY(hipGetDeviceProperties(&prop, device));
    // the code objects in this library are gfx950 only (no other target is built): refuse every other device here instead of failing at the
    // first launch.  MRS_ALLOW_ANY_ARCH=1 skips the test (a gfx950 part that reports a name this check does not know).
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0 && !(getenv("MRS_ALLOW_ANY_ARCH") && atoi(getenv("MRS_ALLOW_ANY_ARCH")) == 1)) {
        mrs::set_error("device %d is %s: this library is built for gfx950 (MI355X) only", device, prop.gcnArchName);
        return MRS_ERR_NO_DEVICE;
    }
    mrs_ctx* c = new mrs_ctx();
    c->device = device;
    c->num_cu = prop.multiProcessorCount;
    c->lds_bytes = prop.maxSharedMemoryPerMultiProcessor ? (size_t)prop.maxSharedMemoryPerMultiProcessor
                                                         : (size_t)prop.sharedMemPerBlock;
    if (c->lds_bytes > 160 * 1024) c->lds_bytes = 160 * 1024;
    *out_ctx = c;
    return MRS_OK;
}

int mrs_ctx_destroy(mrs_ctx* ctx)
{
    if (!ctx) return MRS_OK;
    (void)hipSetDevice(ctx->device);
    for (auto& kv : ctx->sector_luts) {
        if (kv.second.d_thr) (void)hipFree(kv.second.d_thr);
        if (kv.second.d_val) (void)hipFree(kv.second.d_val);
    }
    for (auto& kv : ctx->twiddles)
        if (kv.second) (void)hipFree(kv.second);
    if (ctx->pointfeat_free)
        for (auto& kv : ctx->pointfeat_cache) ctx->pointfeat_free(kv.second);
    for (auto& sl : ctx->side_free) {
        (void)hipStreamSynchronize(sl.stream);
        (void)hipHostFree(sl.pinned);
        (void)hipEventDestroy(sl.join);
        (void)hipEventDestroy(sl.fork);
        (void)hipStreamDestroy(sl.stream);
    }
    delete ctx;
    return MRS_OK;
}

int mrs_ctx_device(const mrs_ctx* ctx) { return ctx ? ctx->device : -1; }

}  // extern "C"
