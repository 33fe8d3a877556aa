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

int mrs_exchange_fetch_rows(mrs_exchange* x, const void* d_local_db, int64_t rows_per_rank, int64_t entry_bytes, const int64_t* d_global_rows,
                            int32_t n_rows, void* d_out, mrs_stream stream)
{
    MRS_REQUIRE(x && d_local_db && d_global_rows && d_out, "null pointer");
    MRS_REQUIRE(rows_per_rank > 0 && n_rows > 0, "sizes must be positive");
    MRS_REQUIRE(entry_bytes > 0 && entry_bytes % 16 == 0, "entry_bytes must be a positive multiple of 16");
    MRS_HIP_TRY(hipSetDevice(x->ctx->device));
    hipStream_t s = (hipStream_t)stream;
    const int W = x->n_ranks, me = x->rank;
    const int64_t units = entry_bytes / 16;
    // 1. everybody's requests to everybody (n_rows int64 per rank: a few KB)
    mrs::Scratch req_all;
    int st = req_all.alloc((size_t)W * n_rows * sizeof(int64_t), s);
    if (st != MRS_OK) return st;
    MRS_NCCL_TRY(rccl().AllGather(d_global_rows, req_all.p, (size_t)n_rows * sizeof(int64_t), ncclChar, x->comm, s));
    std::vector<int64_t> req((size_t)W * n_rows);
    MRS_HIP_TRY(hipMemcpyAsync(req.data(), req_all.p, req.size() * sizeof(int64_t), hipMemcpyDeviceToHost, s));
    MRS_HIP_TRY(hipStreamSynchronize(s));
    for (int64_t r : req) MRS_REQUIRE(r >= 0 && r < rows_per_rank * W, "requested row outside the database");
    // 2. what I send to every peer (local row indices, in the peer's request order) and where what I receive goes (positions in d_out)
    std::vector<int64_t> send_idx, recv_pos;
    std::vector<int64_t> send_off(W + 1, 0), recv_off(W + 1, 0);
    for (int p = 0; p < W; ++p) {
        for (int k = 0; k < n_rows; ++k)
            if (req[(size_t)p * n_rows + k] / rows_per_rank == me) send_idx.push_back(req[(size_t)p * n_rows + k] - (int64_t)me * rows_per_rank);
        send_off[p + 1] = (int64_t)send_idx.size();
    }
    for (int o = 0; o < W; ++o) {
        for (int k = 0; k < n_rows; ++k)
            if (req[(size_t)me * n_rows + k] / rows_per_rank == o) recv_pos.push_back(k);
        recv_off[o + 1] = (int64_t)recv_pos.size();
    }
    mrs::Scratch d_sidx, d_rpos, sendbuf, recvbuf;
    if ((st = d_sidx.alloc(std::max<size_t>(send_idx.size(), 1) * sizeof(int64_t), s)) != MRS_OK) return st;
    if ((st = d_rpos.alloc(std::max<size_t>(recv_pos.size(), 1) * sizeof(int64_t), s)) != MRS_OK) return st;
    if ((st = sendbuf.alloc(std::max<size_t>(send_idx.size(), 1) * entry_bytes, s)) != MRS_OK) return st;
    if ((st = recvbuf.alloc((size_t)n_rows * entry_bytes, s)) != MRS_OK) return st;
    if (!send_idx.empty()) MRS_HIP_TRY(hipMemcpyAsync(d_sidx.p, send_idx.data(), send_idx.size() * sizeof(int64_t), hipMemcpyHostToDevice, s));
    MRS_HIP_TRY(hipMemcpyAsync(d_rpos.p, recv_pos.data(), recv_pos.size() * sizeof(int64_t), hipMemcpyHostToDevice, s));
    if (!send_idx.empty())
        hipLaunchKernelGGL(k_gather_rows16, dim3((unsigned)send_idx.size()), dim3(256), 0, s, static_cast<const uint4*>(d_local_db), d_sidx.as<int64_t>(), units,
                           sendbuf.as<uint4>());
    // 3. the rows travel: one send and one receive per remote peer, grouped; my own rows are a device copy
    char* sb = sendbuf.as<char>();
    char* rb = recvbuf.as<char>();
    if (W > 1) {
        MRS_NCCL_TRY(rccl().GroupStart());
        for (int p = 0; p < W; ++p) {
            if (p == me) continue;
            const int64_t ns = send_off[p + 1] - send_off[p], nr = recv_off[p + 1] - recv_off[p];
            if (ns > 0) MRS_NCCL_TRY(rccl().Send(sb + send_off[p] * entry_bytes, (size_t)(ns * entry_bytes), ncclChar, p, x->comm, s));
            if (nr > 0) MRS_NCCL_TRY(rccl().Recv(rb + recv_off[p] * entry_bytes, (size_t)(nr * entry_bytes), ncclChar, p, x->comm, s));
        }
        MRS_NCCL_TRY(rccl().GroupEnd());
    }
    const int64_t mine = send_off[me + 1] - send_off[me];
    if (mine > 0)
        MRS_HIP_TRY(hipMemcpyAsync(rb + recv_off[me] * entry_bytes, sb + send_off[me] * entry_bytes, (size_t)(mine * entry_bytes), hipMemcpyDeviceToDevice, s));
    // 4. back to request order
    hipLaunchKernelGGL(k_scatter_rows16, dim3((unsigned)n_rows), dim3(256), 0, s, recvbuf.as<uint4>(), d_rpos.as<int64_t>(), units, static_cast<uint4*>(d_out));
    MRS_HIP_TRY(hipGetLastError());
    MRS_HIP_TRY(hipStreamSynchronize(s));      // the index tables are host temporaries; the scratch buffers return to the cache
    return MRS_OK;
}

}  // extern "C"
