// capi.hip -- context management and status reporting of the C ABI (include/mrslam_hip.h).
#include "common.hpp"

namespace mrs {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
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
    delete ctx;
    return MRS_OK;
}

int mrs_ctx_device(const mrs_ctx* ctx) { return ctx ? ctx->device : -1; }

}  // extern "C"
