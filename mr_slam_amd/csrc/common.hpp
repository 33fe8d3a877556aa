// common.hpp -- shared host-side plumbing of libmrslam_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <vector>

#include "../../include/mrslam_hip.h"

namespace mrs {
constexpr int kMaxGridY = 65535;  // HIP grid limit in y: entry points that put the batch there check or chunk


void set_error(const char* fmt, ...);

// Development switches (phase timers, ablations, debug counters) are read from the environment ONLY when MRS_DEV=1 is set as well, and
// each active one is announced on stderr once: a stray variable in a production environment can neither change results nor add
// synchronisation silently.  Returns the variable's value, or nullptr.
const char* dev_env(const char* name);

#define MRS_HIP_TRY(expr)                                                                   \
    do {                                                                                    \
        hipError_t e__ = (expr);                                                            \
        if (e__ != hipSuccess) {                                                            \
            ::mrs::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), __FILE__, \
                             __LINE__);                                                     \
            return MRS_ERR_HIP;                                                             \
        }                                                                                   \
    } while (0)

#define MRS_REQUIRE(cond, msg)                                     \
    do {                                                           \
        if (!(cond)) {                                             \
            ::mrs::set_error("bad argument: %s (%s)", msg, #cond); \
            return MRS_ERR_ARG;                                    \
        }                                                          \
    } while (0)

// Scratch buffer of one call, handed out by a small caching allocator of this library (capi.hip): blocks are
// hipMalloc'ed once and recycled.  A released block carries an event recorded on the stream that used it; it is handed
// out again to the SAME stream at once (stream order makes that safe) and to another stream only after the event has
// completed.  Not hipMallocAsync: with a second process active on the same GPU, small stream-ordered-pool allocations
// were observed to come back overlapping (the memset of one scratch buffer zeroed its neighbour: 6-7 of 30 runs of the
// adapter test under a GPU-holding parent process; 0 of 40 with plain allocations), silently corrupting results.
void* scratch_acquire(size_t bytes, hipStream_t stream);
void scratch_release(void* p, hipStream_t stream);

struct Scratch {
    void* p = nullptr;
    hipStream_t s = nullptr;
    ~Scratch() {
        if (p) scratch_release(p, s);
    }
    int alloc(size_t bytes, hipStream_t stream) {
        s = stream;
        p = scratch_acquire(bytes ? bytes : 16, stream);
        if (!p) {
            set_error("scratch allocation of %zu bytes failed: %s", bytes, hipGetErrorString(hipGetLastError()));
            return MRS_ERR_HIP;
        }
        return MRS_OK;
    }
    template <class T>
    T* as() const { return static_cast<T*>(p); }
};

// Device lookup table that reproduces the reference's sector() step function exactly
// (built on the host with the host libm, see bev.hip).
struct SectorLut {
    float* d_thr = nullptr;  // [4][stride] ascending change points of q=|y|/|x| per quadrant
    int* d_val = nullptr;    // [4][stride] sector value from thr[j] on
    int stride = 0;
    int cnt[4] = {0, 0, 0, 0};
};

}  // namespace mrs

namespace mrs {
// A second stream + fork / join events + a small pinned buffer, kept in a per-context pool: a registration of one pair (the nodes' shape)
// borrows one for the duration of mrs_gicp_batch_align.  Creating them per handle cost 0.4 ms per FastGICP object (the nodes construct one
// per registration, main_RING.py:84); concurrent callers (three rospy callback threads) each get their own slot.
struct SideSlot {
    hipStream_t stream = nullptr;
    hipEvent_t fork = nullptr, join = nullptr;
    int* pinned = nullptr;      // kSidePinnedInts ints
};
constexpr int kSidePinnedInts = 64;
int side_acquire(mrs_ctx* ctx, SideSlot* out);   // MRS_OK / MRS_ERR_HIP
void side_release(mrs_ctx* ctx, const SideSlot& slot);
}  // namespace mrs

struct mrs_ctx {
    std::mutex side_mu;
    std::vector<mrs::SideSlot> side_free;
    int device = 0;
    int num_cu = 0;
    size_t lds_bytes = 0;
    std::mutex mu;
    std::map<int, mrs::SectorLut> sector_luts;  // keyed by num_sector
    // device-resident constant tables for the correlation / Radon kernels (lazily built)
    std::map<int, float*> twiddles;             // keyed by FFT length
    // mrs_pointfeat_batch keeps its Morton-ordered cloud container between calls (keyed by the number of scans per call, at most 4 sizes):
    // creating and freeing ~1.5 GB of device buffers per call cost as much as the kernels on some boxes.  One caller at a time uses a cached
    // container (try_lock); a concurrent caller works on a temporary one.
    std::mutex pointfeat_mu;
    std::map<int, void*> pointfeat_cache;
    void (*pointfeat_free)(void*) = nullptr;    // set by gicp.hip when it caches a container
};
