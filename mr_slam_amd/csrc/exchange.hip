// exchange.hip -- the multi-GPU exchange step of the descriptor database behind the C ABI (mrs_exchange_*): RCCL all-gather of the
// descriptors every rank just built, and the request-based alternative (only the candidate rows asked for travel).
//
// The reference has no collective; this is the "RCCL all-gather of the descriptor database over xGMI ... behind a
// thin C-ABI" of BASELINE.json's north_star, callable from a C++ host (the Mapping node, global_manager.cpp:2016-2021, would shard its
// candidate pairs with it) as well as from mr_slam_amd/shard.py.  One process per GPU; the communicator is either created here from a
// unique id (ncclCommInitRank) or borrowed from the host.  RCCL is NOT a link-time dependency of libmrslam_hip.so: its entry points are
// looked up in the process first (a host that already carries RCCL -- torch, a ROS node linked against it -- keeps its copy) and
// `librccl.so.1` is opened only when they are not there.
#include "common.hpp"

#include <dlfcn.h>

#include <algorithm>
#include <rccl/rccl.h>

namespace {

struct Rccl {
    bool ok = false;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

const Rccl& rccl()
{
    static const Rccl r = [] {
        Rccl x;
        void* h = nullptr;
        auto sym = [&](const char* name) -> void* {
            void* p = dlsym(RTLD_DEFAULT, name);
            if (!p) {
                if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
                if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
                if (h) p = dlsym(h, name);
            }
            return p;
        };
        x.GetUniqueId = reinterpret_cast<decltype(x.GetUniqueId)>(sym("ncclGetUniqueId"));
        x.CommInitRank = reinterpret_cast<decltype(x.CommInitRank)>(sym("ncclCommInitRank"));
        x.CommDestroy = reinterpret_cast<decltype(x.CommDestroy)>(sym("ncclCommDestroy"));
        x.AllGather = reinterpret_cast<decltype(x.AllGather)>(sym("ncclAllGather"));
        x.Send = reinterpret_cast<decltype(x.Send)>(sym("ncclSend"));
        x.Recv = reinterpret_cast<decltype(x.Recv)>(sym("ncclRecv"));
        x.GroupStart = reinterpret_cast<decltype(x.GroupStart)>(sym("ncclGroupStart"));
        x.GroupEnd = reinterpret_cast<decltype(x.GroupEnd)>(sym("ncclGroupEnd"));
        x.GetErrorString = reinterpret_cast<decltype(x.GetErrorString)>(sym("ncclGetErrorString"));
        x.ok = x.GetUniqueId && x.CommInitRank && x.CommDestroy && x.AllGather && x.Send && x.Recv && x.GroupStart && x.GroupEnd;
        return x;
    }();
    return r;
}

#define MRS_NCCL_TRY(expr)                                                                                        \
    do {                                                                                                          \
        ncclResult_t r__ = (expr);                                                                                \
        if (r__ != ncclSuccess) {                                                                                 \
            ::mrs::set_error("%s failed: %s (%s:%d)", #expr, rccl().GetErrorString ? rccl().GetErrorString(r__) : "?", __FILE__, __LINE__); \
            return MRS_ERR_HIP;                                                                                   \
        }                                                                                                         \
    } while (0)

// dst[i] = src[idx[i]] for rows of `units` 16-byte pieces; grid = rows
__global__ void k_gather_rows16(const uint4* __restrict__ src, const int64_t* __restrict__ idx, int64_t units, uint4* __restrict__ dst)
{
    const uint4* s = src + idx[blockIdx.x] * units;
    uint4* d = dst + (int64_t)blockIdx.x * units;
    for (int64_t i = threadIdx.x; i < units; i += blockDim.x) d[i] = s[i];
}

// dst[pos[i]] = src[i]
__global__ void k_scatter_rows16(const uint4* __restrict__ src, const int64_t* __restrict__ pos, int64_t units, uint4* __restrict__ dst)
{
    const uint4* s = src + (int64_t)blockIdx.x * units;
    uint4* d = dst + pos[blockIdx.x] * units;
    for (int64_t i = threadIdx.x; i < units; i += blockDim.x) d[i] = s[i];
}

}  // namespace

struct mrs_exchange {
    mrs_ctx* ctx = nullptr;
    ncclComm_t comm = nullptr;
    bool own = false;
    int n_ranks = 1, rank = 0;
};

extern "C" {

int mrs_exchange_available(void) { return rccl().ok ? 1 : 0; }

int mrs_exchange_unique_id(uint8_t* out128)
{
    MRS_REQUIRE(out128, "null pointer");
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
    if (!rccl().ok) { mrs::set_error("RCCL entry points not found (librccl.so.1)"); return MRS_ERR_UNSUPPORTED; }
    ncclUniqueId id;
    MRS_NCCL_TRY(rccl().GetUniqueId(&id));
    memcpy(out128, &id, sizeof(id));
    return MRS_OK;
}

int mrs_exchange_create(mrs_ctx* ctx, int32_t n_ranks, int32_t rank, const uint8_t* id128, mrs_exchange** out)
{
    MRS_REQUIRE(ctx && id128 && out, "null pointer");
    *out = nullptr;
    MRS_REQUIRE(n_ranks >= 1 && rank >= 0 && rank < n_ranks, "rank outside the world");
    if (!rccl().ok) { mrs::set_error("RCCL entry points not found (librccl.so.1)"); return MRS_ERR_UNSUPPORTED; }
    MRS_HIP_TRY(hipSetDevice(ctx->device));
    ncclUniqueId id;
    memcpy(&id, id128, sizeof(id));
    ncclComm_t comm = nullptr;
    MRS_NCCL_TRY(rccl().CommInitRank(&comm, n_ranks, id, rank));
    mrs_exchange* x = new mrs_exchange();
    x->ctx = ctx; x->comm = comm; x->own = true; x->n_ranks = n_ranks; x->rank = rank;
    *out = x;
    return MRS_OK;
}

int mrs_exchange_create_from_comm(mrs_ctx* ctx, void* nccl_comm, int32_t n_ranks, int32_t rank, mrs_exchange** out)
{
    MRS_REQUIRE(ctx && nccl_comm && out, "null pointer");
    *out = nullptr;
    MRS_REQUIRE(n_ranks >= 1 && rank >= 0 && rank < n_ranks, "rank outside the world");
    if (!rccl().ok) { mrs::set_error("RCCL entry points not found (librccl.so.1)"); return MRS_ERR_UNSUPPORTED; }
    mrs_exchange* x = new mrs_exchange();
    x->ctx = ctx; x->comm = static_cast<ncclComm_t>(nccl_comm); x->own = false; x->n_ranks = n_ranks; x->rank = rank;
    *out = x;
    return MRS_OK;
}

int mrs_exchange_destroy(mrs_exchange* x)
{
    if (!x) return MRS_OK;
    if (x->own && x->comm && rccl().ok) (void)rccl().CommDestroy(x->comm);
    delete x;
    return MRS_OK;
}

int mrs_exchange_world(const mrs_exchange* x, int32_t* n_ranks, int32_t* rank)
{
    MRS_REQUIRE(x, "null handle");
    if (n_ranks) *n_ranks = x->n_ranks;
    if (rank) *rank = x->rank;
    return MRS_OK;
}

int mrs_exchange_allgather(mrs_exchange* x, const void* d_local, int64_t n_local, int64_t entry_bytes, void* d_all, mrs_stream stream)
{
    MRS_REQUIRE(x && d_local && d_all, "null pointer");
    MRS_REQUIRE(n_local > 0 && entry_bytes > 0, "sizes must be positive");
    MRS_HIP_TRY(hipSetDevice(x->ctx->device));
    MRS_NCCL_TRY(rccl().AllGather(d_local, d_all, (size_t)(n_local * entry_bytes), ncclChar, x->comm, (hipStream_t)stream));
    return MRS_OK;
}

// ---- row fetch with the request phase done ahead of time -----------------------------------------------------------------------------
// Candidate rows are usually known long before the entries are needed (they come out of a coarse search; bench.py knows them before a step
// starts), so the request exchange, its host synchronisation and the index tables are paid ONCE, in mrs_exchange_fetch_plan_create; a fetch
// is then gather -> one grouped send / receive per remote peer -> scatter, all stream-ordered, no host synchronisation: it can be issued
// launches ahead on a communication stream and waited for with an event.
struct mrs_fetch_plan {
    mrs_exchange* x = nullptr;
    int n_rows = 0;
    int64_t rows_per_rank = 0;
    std::vector<int64_t> send_off, recv_off;      // [W + 1] prefix counts per peer
    int64_t* d_sidx = nullptr;                     // local rows I send, grouped by peer, in the peer's request order
    int64_t* d_rpos = nullptr;                     // position in the caller's output of every row I receive, grouped by owner
    void* sendbuf = nullptr;
    void* recvbuf = nullptr;
    int64_t buf_entry_bytes = 0;                   // sendbuf / recvbuf are sized for entries of this many bytes
};

extern "C++" {
namespace {
void free_plan(mrs_fetch_plan* p)
{
    if (!p) return;
    if (p->d_sidx) (void)hipFree(p->d_sidx);
    if (p->d_rpos) (void)hipFree(p->d_rpos);
    if (p->sendbuf) (void)hipFree(p->sendbuf);
    if (p->recvbuf) (void)hipFree(p->recvbuf);
    delete p;
}
}  // namespace
}

int mrs_exchange_fetch_plan_create(mrs_exchange* x, int64_t rows_per_rank, const int64_t* d_global_rows, int32_t n_rows, mrs_stream stream,
                                   mrs_fetch_plan** out)
{
    MRS_REQUIRE(x && d_global_rows && out, "null pointer");
    *out = nullptr;
    MRS_REQUIRE(rows_per_rank > 0 && n_rows > 0, "sizes must be positive");
    MRS_HIP_TRY(hipSetDevice(x->ctx->device));
    hipStream_t s = (hipStream_t)stream;
    const int W = x->n_ranks, me = x->rank;
    // everybody's requests to everybody (n_rows int64 per rank: a few KB)
    std::vector<int64_t> req((size_t)W * n_rows);
    {
        mrs::Scratch req_all;
        int st = req_all.alloc((size_t)W * n_rows * sizeof(int64_t), s);
        if (st != MRS_OK) return st;
        MRS_NCCL_TRY(rccl().AllGather(d_global_rows, req_all.p, (size_t)n_rows * sizeof(int64_t), ncclChar, x->comm, s));
        MRS_HIP_TRY(hipMemcpyAsync(req.data(), req_all.p, req.size() * sizeof(int64_t), hipMemcpyDeviceToHost, s));
        MRS_HIP_TRY(hipStreamSynchronize(s));
    }
    for (int64_t r : req) MRS_REQUIRE(r >= 0 && r < rows_per_rank * W, "requested row outside the database");
    mrs_fetch_plan* p = new mrs_fetch_plan();
    p->x = x; p->n_rows = n_rows; p->rows_per_rank = rows_per_rank;
    // what I send to every peer (local row indices, in the peer's request order) and where what I receive goes (positions in the output)
    std::vector<int64_t> send_idx, recv_pos;
    p->send_off.assign(W + 1, 0); p->recv_off.assign(W + 1, 0);
    for (int q = 0; q < W; ++q) {
        for (int k = 0; k < n_rows; ++k)
            if (req[(size_t)q * n_rows + k] / rows_per_rank == me) send_idx.push_back(req[(size_t)q * n_rows + k] - (int64_t)me * rows_per_rank);
        p->send_off[q + 1] = (int64_t)send_idx.size();
    }
    for (int o = 0; o < W; ++o) {
        for (int k = 0; k < n_rows; ++k)
            if (req[(size_t)me * n_rows + k] / rows_per_rank == o) recv_pos.push_back(k);
        p->recv_off[o + 1] = (int64_t)recv_pos.size();
    }
    hipError_t e = hipMalloc(&p->d_sidx, std::max<size_t>(send_idx.size(), 1) * sizeof(int64_t));
    if (e == hipSuccess) e = hipMalloc(&p->d_rpos, (size_t)n_rows * sizeof(int64_t));
    if (e == hipSuccess && !send_idx.empty()) e = hipMemcpy(p->d_sidx, send_idx.data(), send_idx.size() * sizeof(int64_t), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(p->d_rpos, recv_pos.data(), recv_pos.size() * sizeof(int64_t), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        free_plan(p);
        mrs::set_error("fetch plan tables: %s", hipGetErrorString(e));
        return MRS_ERR_HIP;
    }
    *out = p;
    return MRS_OK;
}

int mrs_exchange_fetch_plan_destroy(mrs_fetch_plan* p)
{
    if (p) {
        (void)hipSetDevice(p->x->ctx->device);
        (void)hipDeviceSynchronize();              // a fetch of this plan may still be in flight on some stream
    }
    free_plan(p);
    return MRS_OK;
}

int mrs_exchange_fetch_plan_counts(const mrs_fetch_plan* p, int64_t* rows_sent_to_peers, int64_t* rows_received_from_peers)
{
    MRS_REQUIRE(p, "null plan");
    const int W = p->x->n_ranks, me = p->x->rank;
    if (rows_sent_to_peers) *rows_sent_to_peers = p->send_off[W] - (p->send_off[me + 1] - p->send_off[me]);
    if (rows_received_from_peers) *rows_received_from_peers = p->recv_off[W] - (p->recv_off[me + 1] - p->recv_off[me]);
    return MRS_OK;
}

int mrs_exchange_fetch_planned(mrs_fetch_plan* p, const void* d_local_db, int64_t entry_bytes, void* d_out, mrs_stream stream)
{
    MRS_REQUIRE(p && d_local_db && d_out, "null pointer");
    MRS_REQUIRE(entry_bytes > 0 && entry_bytes % 16 == 0, "entry_bytes must be a positive multiple of 16");
    mrs_exchange* x = p->x;
    MRS_HIP_TRY(hipSetDevice(x->ctx->device));
    hipStream_t s = (hipStream_t)stream;
    const int W = x->n_ranks, me = x->rank;
    const int64_t units = entry_bytes / 16;
    const int64_t n_send = p->send_off[W];
    if (p->buf_entry_bytes < entry_bytes) {        // first use (or larger entries): the plan owns its staging buffers; fetches of ONE plan must
        MRS_HIP_TRY(hipStreamSynchronize(s));      // follow each other in stream order (they share them)
        if (p->sendbuf) (void)hipFree(p->sendbuf);
        if (p->recvbuf) (void)hipFree(p->recvbuf);
        p->sendbuf = p->recvbuf = nullptr;
        p->buf_entry_bytes = 0;
        MRS_HIP_TRY(hipMalloc(&p->sendbuf, (size_t)std::max<int64_t>(n_send, 1) * entry_bytes));
        MRS_HIP_TRY(hipMalloc(&p->recvbuf, (size_t)p->n_rows * entry_bytes));
        p->buf_entry_bytes = entry_bytes;
    }
    if (n_send > 0)
        hipLaunchKernelGGL(k_gather_rows16, dim3((unsigned)n_send), dim3(256), 0, s, static_cast<const uint4*>(d_local_db), p->d_sidx, units,
                           static_cast<uint4*>(p->sendbuf));
    char* sb = static_cast<char*>(p->sendbuf);
    char* rb = static_cast<char*>(p->recvbuf);
    if (W > 1) {
        MRS_NCCL_TRY(rccl().GroupStart());
        ncclResult_t bad = ncclSuccess;
        for (int q = 0; q < W && bad == ncclSuccess; ++q) {
            if (q == me) continue;
            const int64_t ns = p->send_off[q + 1] - p->send_off[q], nr = p->recv_off[q + 1] - p->recv_off[q];
            if (ns > 0) bad = rccl().Send(sb + p->send_off[q] * entry_bytes, (size_t)(ns * entry_bytes), ncclChar, q, x->comm, s);
            if (nr > 0 && bad == ncclSuccess) bad = rccl().Recv(rb + p->recv_off[q] * entry_bytes, (size_t)(nr * entry_bytes), ncclChar, q, x->comm, s);
        }
        const ncclResult_t end = rccl().GroupEnd();    // ALWAYS closed: a group left open would wedge every later collective of the communicator
        if (bad != ncclSuccess || end != ncclSuccess) {
            mrs::set_error("row fetch send / receive failed: %s", rccl().GetErrorString ? rccl().GetErrorString(bad != ncclSuccess ? bad : end) : "?");
            return MRS_ERR_HIP;
        }
    }
    const int64_t mine = p->send_off[me + 1] - p->send_off[me];
    if (mine > 0)
        MRS_HIP_TRY(hipMemcpyAsync(rb + p->recv_off[me] * entry_bytes, sb + p->send_off[me] * entry_bytes, (size_t)(mine * entry_bytes), hipMemcpyDeviceToDevice, s));
    hipLaunchKernelGGL(k_scatter_rows16, dim3((unsigned)p->n_rows), dim3(256), 0, s, static_cast<const uint4*>(p->recvbuf), p->d_rpos, units, static_cast<uint4*>(d_out));
    MRS_HIP_TRY(hipGetLastError());
    return MRS_OK;
}

int mrs_exchange_fetch_rows(mrs_exchange* x, const void* d_local_db, int64_t rows_per_rank, int64_t entry_bytes, const int64_t* d_global_rows,
                            int32_t n_rows, void* d_out, mrs_stream stream)
{
    MRS_REQUIRE(x && d_local_db && d_global_rows && d_out, "null pointer");
    MRS_REQUIRE(entry_bytes > 0 && entry_bytes % 16 == 0, "entry_bytes must be a positive multiple of 16");
    mrs_fetch_plan* p = nullptr;
    int st = mrs_exchange_fetch_plan_create(x, rows_per_rank, d_global_rows, n_rows, stream, &p);
    if (st != MRS_OK) return st;
    st = mrs_exchange_fetch_planned(p, d_local_db, entry_bytes, d_out, stream);
    if (st == MRS_OK && hipStreamSynchronize((hipStream_t)stream) != hipSuccess) { mrs::set_error("row fetch: stream synchronisation failed"); st = MRS_ERR_HIP; }
    free_plan(p);
    return st;
}

}  // extern "C"
