// nn_core.hpp -- exact nearest-neighbour search core of the GICP / RING++ front ends (gfx950, wave64).
//
// Round-3 design: one traversal of a (1024-point tile, 16-point mini) hierarchy per WAVE; every lane evaluated every candidate any lane
// of the wave needed: ~1000 distance evaluations per query for a 1-NN whose ball holds a handful of points.  Two
// things were wrong with it: the leaves and the sharing.
//   * Leaves.  16 CONSECUTIVE points of a Morton-ordered cloud are not compact: the run crosses octree cell boundaries (and a spinning
//     lidar's rings interleave along the curve), so their boxes are fat and overlap; a 3 cm ball touches 5.4 such boxes (86 points) on the
//     bench's scans.  Here a leaf is an OCTREE CELL: the coarsest cell of the cloud's Morton grid that holds <= 16 points (variable
//     size, disjoint cells, boxes = tight boxes of the points inside): the same ball touches 1.5 leaves (13 points).  Leaves are runs of
//     the Morton-ordered point array, found from the sorted codes alone (k_leaf_level: the level of point j is 1 + the longest common
//     octal prefix of codes i and i + 16 over the windows that contain j), so nothing moves.
//   * Sharing.  A wave still walks the hierarchy once for its 64 queries (supers of 64 tiles of 64 leaves, lane-parallel box tests, one
//     ballot per 64 boxes), but the leaves it finds go to an LDS list that every GROUP of GS = 8 consecutive queries filters against its
//     own box; a group's surviving leaves are expanded into a per-group queue of candidate indices, and a lane only evaluates the
//     candidates of its own group (group-uniform addresses: 8 distinct 16-byte loads per wave instruction): ~60 evaluations per query
//     instead of ~1000.  Everything is conservative (0.9999 slack on box distances), candidates reach a lane in ascending index order,
//     so results are the exact neighbours with ties resolved to the smaller index -- a pure function of the data, whatever the grouping.
#pragma once
#include <hip/hip_runtime.h>

#include <cfloat>
#include <cmath>
#include <cstdint>

namespace nnc {

constexpr int kLeafMax = 16;   // points per leaf (octree cell) at most
constexpr int kWL = 128;       // wave list: leaves found by the shared traversal, waiting for the per-group filter

struct LeafHier {              // one cloud's hierarchy (cloud-local arrays)
    const float4* llo;         // [nleaf] leaf box min, .w = first point (int bits, cloud-local index)
    const float4* lhi;         // [nleaf] leaf box max, .w = number of points (int bits, 1..16)
    const float4* tlo;         // [ntile] boxes of 64 consecutive leaves
    const float4* thi;
    const float4* slo;         // [nsuper] boxes of 64 consecutive tiles
    const float4* shi;
    int nleaf, ntile, nsuper;
};

template <int GS>
struct __align__(16) GrpLds {
    static constexpr int NG = 64 / GS;
    static constexpr int QCAP = 16 * GS + 36;      // a chunk of GS leaves adds <= 16 GS candidates; 164 for GS = 8: rows start 36 banks apart (no conflicts between the groups' 16-byte reads)
    int gq[NG][QCAP];          // per-group queue of candidate point indices
    int wl[kWL];
};

__device__ __forceinline__ void wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ float d2_point(float qx, float qy, float qz, const float4& b)
{
    const float dx = qx - b.x, dy = qy - b.y, dz = qz - b.z;
    return __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
}

__device__ __forceinline__ float d2_boxes(const float4& lo, const float4& hi, const float (&qlo)[3], const float (&qhi)[3])
{
    const float dx = fmaxf(fmaxf(lo.x - qhi[0], qlo[0] - hi.x), 0.0f);
    const float dy = fmaxf(fmaxf(lo.y - qhi[1], qlo[1] - hi.y), 0.0f);
    const float dz = fmaxf(fmaxf(lo.z - qhi[2], qlo[2] - hi.z), 0.0f);
    return dx * dx + dy * dy + dz * dz;
}

// Exact neighbour search of 64 queries (one per lane) over one Morton-ordered cloud.
//   lim()            : the lane's current squared search radius (a candidate farther than that cannot matter to it); may shrink while the
//                      search runs, is re-read after every batch of candidates.  Lanes without a query pass live = false.
//   visit(j, d2, ok) : called for every candidate j (cloud-local index) of the lane's GROUP, in ascending j, each at most once with ok = true;
//                      ok = false marks the (at most 3) padding repeats at the end of a batch.  Every point within lim() of the query is
//                      visited; most visited points are not within lim().
// The whole wave must call it together (ballots, shuffles, wave-private LDS).
template <int GS, class Lim, class Visit>
__device__ __forceinline__ void grp_search(const float4* __restrict__ pts, const LeafHier& H, GrpLds<GS>& L, float qx, float qy, float qz,
                                           bool live, Lim lim, Visit visit)
{
    constexpr int QCAP = GrpLds<GS>::QCAP;
    const int lane = threadIdx.x & 63, g = lane / GS, c = lane % GS;
    float glo[3] = {live ? qx : INFINITY, live ? qy : INFINITY, live ? qz : INFINITY};
    float ghi[3] = {live ? qx : -INFINITY, live ? qy : -INFINITY, live ? qz : -INFINITY};
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int o = 1; o < GS; o <<= 1) {
            glo[a] = fminf(glo[a], __shfl_xor(glo[a], o, 64));
            ghi[a] = fmaxf(ghi[a], __shfl_xor(ghi[a], o, 64));
        }
    float wlo[3], whi[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        wlo[a] = glo[a]; whi[a] = ghi[a];
#pragma unroll
        for (int o = GS; o < 64; o <<= 1) {
            wlo[a] = fminf(wlo[a], __shfl_xor(wlo[a], o, 64));
            whi[a] = fmaxf(whi[a], __shfl_xor(whi[a], o, 64));
        }
    }
    float greach, wreach;
    auto refresh = [&]() {
        float r = live ? lim() : -1.0f;
#pragma unroll
        for (int o = 1; o < GS; o <<= 1) r = fmaxf(r, __shfl_xor(r, o, 64));
        greach = r;
#pragma unroll
        for (int o = GS; o < 64; o <<= 1) r = fmaxf(r, __shfl_xor(r, o, 64));
        wreach = r;
    };
    refresh();
    int wcnt = 0;   // wave-uniform: leaves in L.wl
    int qcnt = 0;   // group-uniform: candidates in L.gq[g]

    auto flush_queue = [&]() {
        int maxq = qcnt;
#pragma unroll
        for (int o = GS; o < 64; o <<= 1) maxq = max(maxq, __shfl_xor(maxq, o, 64));
        if (maxq == 0) return;
        if (qcnt > 0 && c < 3) L.gq[g][qcnt + c] = L.gq[g][qcnt - 1];   // pad to a multiple of 4 with repeats (flagged ok = false)
        wave_lds_sync();
        for (int t = 0; t < maxq; t += 4) {
            if (t < qcnt) {
                const int4 id = *reinterpret_cast<const int4*>(&L.gq[g][t]);
                const float4 p0 = pts[id.x], p1 = pts[id.y], p2 = pts[id.z], p3 = pts[id.w];
                visit(id.x, d2_point(qx, qy, qz, p0), true);
                visit(id.y, d2_point(qx, qy, qz, p1), t + 1 < qcnt);
                visit(id.z, d2_point(qx, qy, qz, p2), t + 2 < qcnt);
                visit(id.w, d2_point(qx, qy, qz, p3), t + 3 < qcnt);
            }
        }
        wave_lds_sync();
        qcnt = 0;
        refresh();
    };

    auto flush_list = [&]() {
        wave_lds_sync();
        for (int j0 = 0; j0 < wcnt; j0 += GS) {
            const int e = j0 + c;
            bool hit = false;
            int start = 0, cnt = 0;
            if (e < wcnt) {
                const int lf = L.wl[e];
                const float4 lo = H.llo[lf], hi = H.lhi[lf];
                hit = d2_boxes(lo, hi, glo, ghi) * 0.9999f <= greach;
                start = __float_as_int(lo.w);
                cnt = __float_as_int(hi.w);
            }
            const int add = hit ? cnt : 0;
            int inc = add;   // inclusive prefix sum over the group's lanes
#pragma unroll
            for (int o = 1; o < GS; o <<= 1) {
                const int v = __shfl_up(inc, o, GS);
                if (c >= o) inc += v;
            }
            const int total = __shfl(inc, GS - 1, GS);
            if (__any(qcnt + total > QCAP - 3)) flush_queue();   // (the hit decisions taken with the older, larger reach stay valid: conservative)
            if (hit) {
                const int off = qcnt + inc - add;
                for (int u = 0; u < cnt; ++u) L.gq[g][off + u] = start + u;
            }
            qcnt += total;
        }
        wcnt = 0;
    };

    for (int sb = 0; sb < H.nsuper; sb += 64) {
        const int s = sb + lane;
        const bool sh = s < H.nsuper && d2_boxes(H.slo[s], H.shi[s], wlo, whi) * 0.9999f <= wreach;
        unsigned long long sm = __ballot(sh);
        while (sm) {
            const int S = sb + (int)__builtin_ctzll(sm);
            sm &= sm - 1;
            const int t = S * 64 + lane;
            const bool th = t < H.ntile && d2_boxes(H.tlo[t], H.thi[t], wlo, whi) * 0.9999f <= wreach;
            unsigned long long tm = __ballot(th);
            while (tm) {
                const int T = S * 64 + (int)__builtin_ctzll(tm);
                tm &= tm - 1;
                const int lf = T * 64 + lane;
                const bool lh = lf < H.nleaf && d2_boxes(H.llo[lf], H.lhi[lf], wlo, whi) * 0.9999f <= wreach;
                const unsigned long long lm = __ballot(lh);
                const int nh = __popcll(lm);
                if (nh == 0) continue;
                if (wcnt + nh > kWL) flush_list();
                if (lh) L.wl[wcnt + (int)__popcll(lm & ((1ull << lane) - 1ull))] = lf;
                wcnt += nh;
            }
        }
    }
    flush_list();
    flush_queue();
}

// The same evaluation over a plain index range [first, first + count) shared by the lane's group (first, count group-uniform; count may
// differ between groups): the seed phase of the k-NN search.
template <int GS, class Visit>
__device__ __forceinline__ void grp_eval_range(const float4* __restrict__ pts, float qx, float qy, float qz, int first, int count, Visit visit)
{
    int most = count;
#pragma unroll
    for (int o = GS; o < 64; o <<= 1) most = max(most, __shfl_xor(most, o, 64));
    for (int t = 0; t < most; t += 4) {
        if (t < count) {
            const int j0 = first + t, j1 = first + min(t + 1, count - 1), j2 = first + min(t + 2, count - 1), j3 = first + min(t + 3, count - 1);
            const float4 p0 = pts[j0], p1 = pts[j1], p2 = pts[j2], p3 = pts[j3];
            visit(j0, d2_point(qx, qy, qz, p0), true);
            visit(j1, d2_point(qx, qy, qz, p1), t + 1 < count);
            visit(j2, d2_point(qx, qy, qz, p2), t + 2 < count);
            visit(j3, d2_point(qx, qy, qz, p3), t + 3 < count);
        }
    }
}

// ---- building the leaves from the sorted keys (cloud id << 42 | 42-bit Morton code) ------------------------------------------------
// common leading octal digits (0..14) of the two codes; -1 when the keys belong to different clouds
__device__ __forceinline__ int common_octal_prefix(unsigned long long a, unsigned long long b)
{
    const unsigned long long x = a ^ b;
    if (x >> 42) return -1;
    if (x == 0) return 14;
    const int hb = 63 - __builtin_clzll(x);   // highest differing bit, 0..41
    return (41 - hb) / 3;
}

// cellhead[j] = j if point j is the first of its octree leaf cell, else 0 (max-scanned into "first point of my cell" by the caller).
// Level of point j = the smallest l such that the level-l cell that holds j has <= kLeafMax points = 1 + max over i in [j - 16, j] of the
// common octal prefix of keys i and i + 16 (sorted keys: i <= j <= i + 16 share a level-l cell iff both ends do), clamped to the 14 levels
// of the code: more than 16 points with one and the same code stay in one cell and are cut into runs of 16 afterwards (k_leaf_heads).
__global__ void k_leaf_level(const unsigned long long* __restrict__ keys, size_t n, int* __restrict__ cellhead)
{
    for (size_t j = blockIdx.x * (size_t)blockDim.x + threadIdx.x; j < n; j += (size_t)gridDim.x * blockDim.x) {
        const unsigned long long kj = keys[j];
        int lvl = 0;
        const size_t i0 = j >= (size_t)kLeafMax ? j - kLeafMax : 0;
        for (size_t i = i0; i <= j; ++i) {
            if (i + kLeafMax >= n) break;
            lvl = max(lvl, 1 + common_octal_prefix(keys[i], keys[i + kLeafMax]));
        }
        lvl = min(lvl, 14);
        const int sh = 42 - 3 * lvl;
        const bool head = j == 0 || (kj >> sh) != (keys[j - 1] >> sh);
        cellhead[j] = head ? (int)j : 0;
    }
}

// head[j] = 1 if point j starts a leaf: first of its cell, or a multiple of 16 points into it
__global__ void k_leaf_heads(const int* __restrict__ cellstart, size_t n, int* __restrict__ head)
{
    for (size_t j = blockIdx.x * (size_t)blockDim.x + threadIdx.x; j < n; j += (size_t)gridDim.x * blockDim.x)
        head[j] = (((int)j - cellstart[j]) % kLeafMax == 0) ? 1 : 0;
}

// leaf_first[c] = index (over all clouds) of cloud c's first leaf; leaf_first[n_clouds] = number of leaves
__global__ void k_leaf_first(const int* __restrict__ head, const int* __restrict__ leafid, const int64_t* __restrict__ offs, int n_clouds,
                             int* __restrict__ leaf_first)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c > n_clouds) return;
    if (c == n_clouds) {
        const int64_t last = offs[n_clouds] - 1;
        leaf_first[c] = leafid[last] + head[last];
    } else {
        leaf_first[c] = leafid[offs[c]];
    }
}

// one lane per leaf head: tight box of the leaf's points, first point (cloud-local) and count in the .w components
__global__ void k_leaf_boxes(const float4* __restrict__ pts, const unsigned long long* __restrict__ keys, const int* __restrict__ head,
                             const int* __restrict__ leafid, const int64_t* __restrict__ offs, size_t n, float4* __restrict__ llo,
                             float4* __restrict__ lhi)
{
    for (size_t j = blockIdx.x * (size_t)blockDim.x + threadIdx.x; j < n; j += (size_t)gridDim.x * blockDim.x) {
        if (!head[j]) continue;
        const int cloud = (int)(keys[j] >> 42);
        const size_t end = (size_t)offs[cloud + 1];
        float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
        int cnt = 0;
        for (size_t u = j; u < end && cnt < kLeafMax && (u == j || !head[u]); ++u) {
            const float4 p = pts[u];
            lo[0] = fminf(lo[0], p.x); lo[1] = fminf(lo[1], p.y); lo[2] = fminf(lo[2], p.z);
            hi[0] = fmaxf(hi[0], p.x); hi[1] = fmaxf(hi[1], p.y); hi[2] = fmaxf(hi[2], p.z);
            ++cnt;
        }
        const int lid = leafid[j];
        llo[lid] = make_float4(lo[0], lo[1], lo[2], __int_as_float((int)(j - (size_t)offs[cloud])));
        lhi[lid] = make_float4(hi[0], hi[1], hi[2], __int_as_float(cnt));
    }
}

// boxes of 64 consecutive children: out[out_first[c] + t] = box of children [64 t, 64 t + 64) of cloud c (children of cloud c:
// in[in_first[c] .. in_first[c + 1]) ).  One wave per output box; grid = (most outputs of any cloud, clouds).
__global__ __launch_bounds__(64) void k_group_boxes(const float4* __restrict__ ilo, const float4* __restrict__ ihi, const int* __restrict__ in_first,
                                                     const int* __restrict__ out_first, float4* __restrict__ olo, float4* __restrict__ ohi)
{
    const int c = blockIdx.y, t = blockIdx.x;
    const int n_in = in_first[c + 1] - in_first[c];
    if (t * 64 >= n_in) return;
    const int i = t * 64 + (int)threadIdx.x;
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    if (i < n_in) {
        const float4 a = ilo[in_first[c] + i], b = ihi[in_first[c] + i];
        lo[0] = a.x; lo[1] = a.y; lo[2] = a.z; hi[0] = b.x; hi[1] = b.y; hi[2] = b.z;
    }
#pragma unroll
    for (int a = 0; a < 3; ++a)
        for (int o = 32; o > 0; o >>= 1) {
            lo[a] = fminf(lo[a], __shfl_xor(lo[a], o, 64));
            hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], o, 64));
        }
    if (threadIdx.x == 0) {
        olo[out_first[c] + t] = make_float4(lo[0], lo[1], lo[2], 0.f);
        ohi[out_first[c] + t] = make_float4(hi[0], hi[1], hi[2], 0.f);
    }
}

}  // namespace nnc
