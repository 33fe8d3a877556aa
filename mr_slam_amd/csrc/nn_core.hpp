// nn_core.hpp -- exact nearest-neighbour search core of the GICP / RING++ front ends (gfx950, wave64).
//
// Round-3 design: one traversal of a (1024-point tile, 16-point mini) hierarchy per WAVE; every lane evaluated every candidate any lane
// of the wave needed: ~1000 distance evaluations per query for a 1-NN whose ball holds a handful of points.  Two
// things were wrong with it: the leaves and the sharing.
//   * Leaves.  16 CONSECUTIVE points of a Morton-ordered cloud are not compact: the run crosses octree cell boundaries (and a spinning
//     lidar's rings interleave along the curve), so their boxes are fat and overlap; a 3 cm ball touches 5.4 such boxes (86 points) on the
//     bench's scans.  Here a leaf is an OCTREE CELL: the coarsest cell of the cloud's Morton grid that holds <= 8 points (variable size,
//     disjoint cells, boxes = tight boxes of the points inside): the same ball touches 1.4 leaves (6 points).  Leaves are runs of the
//     Morton-ordered point array, found from the sorted codes alone (k_leaf_level: the level of point j is 1 + the longest common octal
//     prefix of codes i and i + 8 over the windows that contain j), so nothing moves.
//   * Sharing.  A wave still walks the hierarchy once for its 64 queries (supers of 64 tiles of 64 leaves, lane-parallel box tests, one
//     ballot per 64 boxes), but tiles and leaves are tested against the boxes of the 8 GROUPS of 8 consecutive queries (kept in scalar
//     registers), never against the wave's own box alone: 64 consecutive points of a Morton-ordered cloud regularly straddle a jump of
//     the curve, and a wave box that spans half the scene would drag thousands of leaves in (the first version of this file did that: its
//     slowest waves ran for milliseconds).  A leaf that touches a group goes to that group's queue; queues are drained 8 leaves at a
//     time: each lane of the group fetches one leaf's points (8 independent 16-byte loads in flight per lane), they are staged in LDS
//     and every lane evaluates the staged points of ITS group only: ~60 evaluations per query instead of ~1000, one memory round
//     trip per 64 candidates.
// Everything is conservative (0.9999 slack on box distances), candidates reach a lane in ascending index order, so results are the
// exact neighbours with ties resolved to the smaller index -- a pure function of the data, whatever the grouping.
#pragma once
#include <hip/hip_runtime.h>

#include <cfloat>
#include <cmath>
#include <cstdint>

namespace nnc {

constexpr int kLeafMax = 8;    // points per leaf (octree cell) at most
constexpr int kTilePts = 256;  // points per tile (octree cell one level of the hierarchy up) at most
constexpr int kLQ = 64;        // leaf-queue entries per group (a round of 64 leaves always fits an empty queue): 10 KB of LDS per wave
constexpr int kStage = 64;     // staged points per group: 8 leaves x 8 points; point t of group g sits in column (t + g) & 63 (rows 4 banks apart)

struct LeafHier {              // one cloud's hierarchy (cloud-local arrays)
    const float4* llo;         // [nleaf] leaf box min, .w = first point (int bits, cloud-local index)
    const float4* lhi;         // [nleaf] leaf box max, .w = number of points (int bits, 1..8)
    const float4* tlo;         // [ntile] tile = octree cell of <= 256 points: box min, .w = first leaf (int bits, cloud-local)
    const float4* thi;         //         box max, .w = number of leaves
    const float4* slo;         // [nsuper] boxes of 64 consecutive tiles
    const float4* shi;
    int nleaf, ntile, nsuper;
};

struct __align__(16) GrpLds {  // wave-private
    float4 stage[8][kStage];   // .w = the point's cloud-local index (int bits)
    int lq[8][kLQ];            // first point | (count - 1) << 28
};

__device__ __forceinline__ void wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// cross-lane moves on the DPP path (no LDS crossbar): quad swaps, mirror within 8 / 16 lanes, shifts within a row of 16
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}
template <int CTRL>
__device__ __forceinline__ int dpp_i(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, false); }

__device__ __forceinline__ float grp_max(float v)   // max over the 8 lanes of the group, in every lane
{
    v = fmaxf(v, dpp_f<0xB1>(v));    // quad_perm [1,0,3,2]
    v = fmaxf(v, dpp_f<0x4E>(v));    // quad_perm [2,3,0,1]
    return fmaxf(v, dpp_f<0x141>(v));  // row_half_mirror
}
__device__ __forceinline__ float grp_min(float v)
{
    v = fminf(v, dpp_f<0xB1>(v));
    v = fminf(v, dpp_f<0x4E>(v));
    return fminf(v, dpp_f<0x141>(v));
}
__device__ __forceinline__ int grp_max_i(int v)
{
    v = max(v, dpp_i<0xB1>(v));
    v = max(v, dpp_i<0x4E>(v));
    return max(v, dpp_i<0x141>(v));
}
__device__ __forceinline__ float readlane_f(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }

__device__ __forceinline__ float d2_point(float qx, float qy, float qz, const float4& b)
{
    const float dx = qx - b.x, dy = qy - b.y, dz = qz - b.z;
    return __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
}

// squared distance between box (lo, hi) and the box q = {min x, y, z, max x, y, z}
__device__ __forceinline__ float d2_boxes(const float4& lo, const float4& hi, const float (&q)[6])
{
    const float dx = fmaxf(fmaxf(lo.x - q[3], q[0] - hi.x), 0.0f);
    const float dy = fmaxf(fmaxf(lo.y - q[4], q[1] - hi.y), 0.0f);
    const float dz = fmaxf(fmaxf(lo.z - q[5], q[2] - hi.z), 0.0f);
    return dx * dx + dy * dy + dz * dz;
}

// Evaluate the staged points of the lane's group: points 0 .. total of row g (rotated by g columns), 4 per trip.
template <class Visit>
__device__ __forceinline__ void eval_stage(const GrpLds& L, int g, int total, float qx, float qy, float qz, Visit& visit)
{
    for (int t = 0; __any(t < total); t += 4) {
        if (t < total) {
            const float4 s0 = L.stage[g][(t + g) & 63], s1 = L.stage[g][(t + 1 + g) & 63], s2 = L.stage[g][(t + 2 + g) & 63],
                         s3 = L.stage[g][(t + 3 + g) & 63];
            visit(__float_as_int(s0.w), d2_point(qx, qy, qz, s0), true);
            visit(__float_as_int(s1.w), d2_point(qx, qy, qz, s1), t + 1 < total);
            visit(__float_as_int(s2.w), d2_point(qx, qy, qz, s2), t + 2 < total);
            visit(__float_as_int(s3.w), d2_point(qx, qy, qz, s3), t + 3 < total);
        }
    }
}

__device__ __forceinline__ float d2_point_box(const float4& lo, const float4& hi, float x, float y, float z)
{
    const float dx = fmaxf(fmaxf(lo.x - x, x - hi.x), 0.0f);
    const float dy = fmaxf(fmaxf(lo.y - y, y - hi.y), 0.0f);
    const float dz = fmaxf(fmaxf(lo.z - z, z - hi.z), 0.0f);
    return dx * dx + dy * dy + dz * dz;
}

// Exact neighbour search of 64 queries (one per lane) over one Morton-ordered cloud.
//   lim()            : the lane's current squared search radius (a candidate farther than that cannot matter to it); may shrink while the
//                      search runs, is re-read after every drained batch.  Lanes without a query pass live = false.
//   visit(j, d2, ok) : called for the candidates j (cloud-local index) of the lane's GROUP in ascending j, each at most once with ok = true;
//                      calls with ok = false carry padding and must be ignored.  Every point within lim() of the query is visited; most
//                      visited points are not within lim().  Must not use wave-wide operations (it runs under divergent control flow).
// Supers and tiles are found with the wave's box (lanes = nodes, one ballot per 64 boxes), then every decision is PER QUERY:
//   tile  : the tile's box goes to scalar registers, every lane tests its own ball against it -> the set of queries that need the tile;
//   leaf  : lanes = the tile's 64 leaves; each query that needs the tile goes to scalar registers in turn and is tested against all 64
//           leaf boxes at once; the leaves any query of a group needs are queued for that group.
// Per-query tests are what bounds the work: 8 consecutive points of a Morton-ordered cloud regularly straddle a jump of the curve, the
// box of such a group spans metres (up to the whole scene) while its queries' balls are centimetres wide -- queued by group box, 1 % of the
// groups carried 27 % of all queued leaves, a typical tile was tested for 5-6 groups of which one needed it, and the slowest wave of a
// launch ran for milliseconds.
// The whole wave must call it together (ballots, cross-lane moves, wave-private LDS).
// development counters (MRS_DEV=1 MRS_NN_PROF=1): shader-clock cycles per phase and event counts, summed over the waves of a launch
struct Prof {
    unsigned long long cyc_top = 0, cyc_leaf = 0, cyc_drain = 0, tiles_near = 0, tiles_needed = 0, grp_tiles = 0, query_tests = 0, queued = 0,
                       batches = 0, staged = 0, drains = 0;
};
__device__ unsigned long long g_prof[16];

template <bool PROF = false, class Lim, class Visit>
__device__ __forceinline__ void grp_search(const float4* __restrict__ pts, const LeafHier& H, GrpLds& L, float qx, float qy, float qz,
                                           bool live, Lim lim, Visit visit, Prof* prof = nullptr)
{
    const int lane = threadIdx.x & 63, g = lane >> 3, c = lane & 7;
    unsigned long long tk = 0;
    auto tick = [&](unsigned long long& acc) __attribute__((always_inline)) {
        if (PROF) { const unsigned long long now = __builtin_readcyclecounter(); acc += now - tk; tk = now; }
    };
    if (PROF) tk = __builtin_readcyclecounter();
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    // the first 64 supers are requested before anything else (they do not depend on the radii, whose first evaluation may wait for a load)
    const int s_first = min(lane, H.nsuper - 1);
    const float4 s0lo = H.slo[s_first], s0hi = H.shi[s_first];
    float wb[6] = {live ? qx : INFINITY, live ? qy : INFINITY, live ? qz : INFINITY, live ? qx : -INFINITY, live ? qy : -INFINITY, live ? qz : -INFINITY};
#pragma unroll
    for (int a = 0; a < 6; ++a) {      // the wave's box, wave-uniform
        float v = a < 3 ? grp_min(wb[a]) : grp_max(wb[a]);
        float r = readlane_f(v, 0);
#pragma unroll
        for (int gg = 1; gg < 8; ++gg) r = a < 3 ? fminf(r, readlane_f(v, 8 * gg)) : fmaxf(r, readlane_f(v, 8 * gg));
        wb[a] = r;
    }
    float mylim, wreach;   // the lane's radius; the largest of the wave
    auto refresh = [&]() __attribute__((always_inline)) {
        mylim = live ? lim() : -1.0f;
        const float r = grp_max(mylim);
        wreach = readlane_f(r, 0);
#pragma unroll
        for (int gg = 1; gg < 8; ++gg) wreach = fmaxf(wreach, readlane_f(r, 8 * gg));
    };
    refresh();
    int lqmax = 0;                           // the longest group queue (wave-uniform)
    int lqv = 0;                             // lane gg < 8 holds the queue length of group gg (dynamic indexing without unrolling the group loop)
    int mylqn = 0;                           // the lane's group's queue length

    auto drain = [&]() __attribute__((always_inline)) {
        if (lqmax == 0) return;
        if (PROF) { tick(prof->cyc_leaf); ++prof->drains; }
        wave_lds_sync();
        for (int r0 = 0; r0 < lqmax; r0 += 8) {
            if (PROF) ++prof->batches;
            const int e = r0 + c;
            int start = 0, cnt = 0;
            if (e < mylqn) {
                const int ent = L.lq[g][e];
                start = ent & 0x0fffffff;
                cnt = (int)((unsigned)ent >> 28) + 1;
            }
            float4 p[kLeafMax];
#pragma unroll
            for (int u = 0; u < kLeafMax; ++u)
                if (u < cnt) p[u] = pts[start + u];
            int inc = cnt;   // inclusive prefix sum over the group's lanes (row_shr within the row of 16: guarded, so nothing crosses a group)
            { const int v = dpp_i<0x111>(inc); if (c >= 1) inc += v; }
            { const int v = dpp_i<0x112>(inc); if (c >= 2) inc += v; }
            { const int v = dpp_i<0x114>(inc); if (c >= 4) inc += v; }
            const int total = grp_max_i(inc);
            if (PROF && lane == 63) prof->staged += (unsigned long long)total;
            const int off = inc - cnt;
#pragma unroll
            for (int u = 0; u < kLeafMax; ++u)
                if (u < cnt) L.stage[g][(off + u + g) & 63] = make_float4(p[u].x, p[u].y, p[u].z, __int_as_float(start + u));
            wave_lds_sync();
            eval_stage(L, g, total, qx, qy, qz, visit);
            wave_lds_sync();
        }
        lqv = 0;
        lqmax = 0; mylqn = 0;
        refresh();
        if (PROF) tick(prof->cyc_drain);
    };

    // one round = up to 64 consecutive leaves (lanes = leaves) against the queries `qm` says need their tile.  `done`: the groups already
    // served (bit per group).  Returns true when a group's queue cannot take what the round adds: the caller drains and calls again.
    auto leaf_round = [&](const float4& lo, const float4& hi, int nleaf, unsigned long long qm, unsigned& done) __attribute__((always_inline)) {
        const bool lin = lane < nleaf;
        const int entry = __float_as_int(lo.w) | ((__float_as_int(hi.w) - 1) << 28);
        for (int gg = 0; gg < 8; ++gg) {
            unsigned byte = (unsigned)(qm >> (8 * gg)) & 0xffu;     // the group's queries that need this tile (wave-uniform)
            if (!byte || ((done >> gg) & 1u)) continue;
            unsigned long long lm = 0;                               // leaves of the round any of them needs
            if (PROF) { ++prof->grp_tiles; prof->query_tests += __popc(byte); }
            while (byte) {
                const int ql = 8 * gg + (int)__builtin_ctz(byte);
                byte &= byte - 1;
                const float sx = readlane_f(qx, ql), sy = readlane_f(qy, ql), sz = readlane_f(qz, ql), sl = readlane_f(mylim, ql);
                lm |= __ballot(lin && d2_point_box(lo, hi, sx, sy, sz) * 0.9999f <= sl);
            }
            if (lm) {
                const int base = __builtin_amdgcn_readlane(lqv, gg);
                const int n = (int)__popcll(lm);
                if (base + n > kLQ) return true;
                if ((lm >> lane) & 1ull) L.lq[gg][base + (int)__popcll(lm & lt_mask)] = entry;
                if (PROF) prof->queued += n;
                if (lane == gg) lqv += n;
                lqmax = max(lqmax, base + n);
                if (g == gg) mylqn += n;
            }
            done |= 1u << gg;
        }
        return false;
    };
    // Up to three rounds wait with their query masks (lane i < 3 holds round i) while their leaf boxes are fetched together: one memory
    // round trip for the batch.  A queue that fills up (64 needed leaves for one group: large radii only) is drained and the batch
    // re-run for the groups not served yet.
    int pend = 0;                              // rounds waiting (wave-uniform)
    int vb = 0, vn = 1;
    unsigned vql = 0, vqh = 0;
    auto run_pending = [&]() __attribute__((always_inline)) {
        if (pend == 0) return;
        unsigned done0 = 0, done1 = 0, done2 = 0;
        for (;;) {
            int pb[3], pn[3];
            float4 a[3], b[3];
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                pb[i] = __builtin_amdgcn_readlane(vb, i); pn[i] = __builtin_amdgcn_readlane(vn, i);
                if (i < pend) { a[i] = H.llo[pb[i] + min(lane, pn[i] - 1)]; b[i] = H.lhi[pb[i] + min(lane, pn[i] - 1)]; }
            }
            auto qmask = [&](int i) __attribute__((always_inline)) {
                return ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)vqh, i) << 32) | (unsigned)__builtin_amdgcn_readlane((int)vql, i);
            };
            bool full = leaf_round(a[0], b[0], pn[0], qmask(0), done0);
            if (!full && pend > 1) full = leaf_round(a[1], b[1], pn[1], qmask(1), done1);
            if (!full && pend > 2) full = leaf_round(a[2], b[2], pn[2], qmask(2), done2);
            if (!full) break;
            if (PROF) tick(prof->cyc_leaf);
            drain();
        }
        pend = 0;
        if (PROF) tick(prof->cyc_leaf);
    };
    auto push_round = [&](int base, int n, unsigned long long qm) __attribute__((always_inline)) {
        if (pend == 3) { if (PROF) tick(prof->cyc_top); run_pending(); }
        if (lane == pend) { vb = base; vn = n; vql = (unsigned)qm; vqh = (unsigned)(qm >> 32); }
        ++pend;
    };
    // per-query tile test of the tiles of one super (lanes = tiles, boxes already in registers): the tile's box goes to scalar registers,
    // every lane tests its own ball; the rounds of leaves of the tiles some query needs are pushed
    auto tile_round = [&](const float4& tlo, const float4& thi, bool tin) __attribute__((always_inline)) {
        const bool th = tin && d2_boxes(tlo, thi, wb) * 0.9999f <= wreach;
        for (unsigned long long tm0 = __ballot(th); tm0; tm0 &= tm0 - 1) {
            const int tl = (int)__builtin_ctzll(tm0);
            float4 blo, bhi;
            blo.x = readlane_f(tlo.x, tl); blo.y = readlane_f(tlo.y, tl); blo.z = readlane_f(tlo.z, tl);
            bhi.x = readlane_f(thi.x, tl); bhi.y = readlane_f(thi.y, tl); bhi.z = readlane_f(thi.z, tl);
            const unsigned long long qm = __ballot(d2_point_box(blo, bhi, qx, qy, qz) * 0.9999f <= mylim);
            if (PROF) ++prof->tiles_near;
            if (!qm) continue;
            if (PROF) ++prof->tiles_needed;
            const int first = __builtin_amdgcn_readlane(__float_as_int(tlo.w), tl);
            const int nl = __builtin_amdgcn_readlane(__float_as_int(thi.w), tl);
            for (int r0 = 0; r0 < nl; r0 += 64) push_round(first + r0, min(64, nl - r0), qm);
        }
    };

    for (int sb = 0; sb < H.nsuper; sb += 64) {
        const int s = sb + lane;
        float4 slo = s0lo, shi = s0hi;
        if (sb > 0) { slo = H.slo[min(s, H.nsuper - 1)]; shi = H.shi[min(s, H.nsuper - 1)]; }
        const bool sh = s < H.nsuper && d2_boxes(slo, shi, wb) * 0.9999f <= wreach;
        unsigned long long sm = __ballot(sh);
        // the tiles of the next hit super are requested before those of the current one are used
        int S = sm ? sb + (int)__builtin_ctzll(sm) : -1;
        sm &= sm - 1;
        float4 nlo = H.tlo[min(max(S, 0) * 64 + lane, H.ntile - 1)], nhi = H.thi[min(max(S, 0) * 64 + lane, H.ntile - 1)];
        while (S >= 0) {
            const float4 tlo = nlo, thi = nhi;
            const bool tin = S * 64 + lane < H.ntile;
            S = sm ? sb + (int)__builtin_ctzll(sm) : -1;
            sm &= sm - 1;
            if (S >= 0) { nlo = H.tlo[min(S * 64 + lane, H.ntile - 1)]; nhi = H.thi[min(S * 64 + lane, H.ntile - 1)]; }
            tile_round(tlo, thi, tin);
            if (PROF) tick(prof->cyc_top);
        }
    }
    run_pending();
    drain();
}

// The same evaluation over a plain index range [first, first + count), count <= 64, shared by the lane's group (first, count group-
// uniform; they may differ between groups): the seed phase of the k-NN search.  One staging round trip.
template <class Visit>
__device__ __forceinline__ void grp_eval_range(const float4* __restrict__ pts, GrpLds& L, float qx, float qy, float qz, int first, int count,
                                               Visit visit)
{
    const int lane = threadIdx.x & 63, g = lane >> 3, c = lane & 7;
    float4 p[8];
#pragma unroll
    for (int r = 0; r < 8; ++r)
        if (c + 8 * r < count) p[r] = pts[first + c + 8 * r];
#pragma unroll
    for (int r = 0; r < 8; ++r)
        if (c + 8 * r < count) L.stage[g][(c + 8 * r + g) & 63] = make_float4(p[r].x, p[r].y, p[r].z, __int_as_float(first + c + 8 * r));
    wave_lds_sync();
    eval_stage(L, g, count, qx, qy, qz, visit);
    wave_lds_sync();
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
// Level of point j = the smallest l such that the level-l cell that holds j has <= kLeafMax points = 1 + max over i in [j - 8, j] of the
// common octal prefix of keys i and i + 8 (sorted keys: i <= j <= i + 8 share a level-l cell iff both ends do), clamped to the 14 levels
// of the code: more than 8 points with one and the same code stay in one cell and are cut into runs of 8 afterwards (k_leaf_heads).
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

// Tiles: the same construction one level up (cells of <= kTilePts points).  d[i] = common octal prefix of keys i and i + kTilePts ...
__global__ void k_tile_prefix(const unsigned long long* __restrict__ keys, size_t n, signed char* __restrict__ d)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        d[i] = (signed char)(i + kTilePts < n ? common_octal_prefix(keys[i], keys[i + kTilePts]) : -1);
}

// ... and tilehead[j] = 1 if point j is the first of its tile cell: level = 1 + max of d over [j - kTilePts, j] (a window staged in the LDS),
// clamped to 14; a cell that still holds more than kTilePts points (one code repeated) stays one tile with many leaves.
__global__ __launch_bounds__(1024) void k_tile_heads(const unsigned long long* __restrict__ keys, const signed char* __restrict__ d, size_t n,
                                                      int* __restrict__ tilehead)
{
    __shared__ signed char win[1024 + kTilePts];
    for (size_t b0 = (size_t)blockIdx.x * 1024; b0 < n; b0 += (size_t)gridDim.x * 1024) {
        __syncthreads();
        for (int u = threadIdx.x; u < 1024 + kTilePts; u += 1024) {
            const long long i = (long long)b0 - kTilePts + u;
            win[u] = (i >= 0 && (size_t)i < n) ? d[i] : (signed char)-1;
        }
        __syncthreads();
        const size_t j = b0 + threadIdx.x;
        if (j >= n) continue;
        int m = -1;
        for (int u = 0; u <= kTilePts; ++u) m = max(m, (int)win[threadIdx.x + u]);
        const int lvl = min(m + 1, 14);
        const int sh = 42 - 3 * lvl;
        tilehead[j] = (j == 0 || (keys[j] >> sh) != (keys[j - 1] >> sh)) ? 1 : 0;
    }
}

// one lane per tile head: box of the tile's leaves, first leaf (cloud-local) and leaf count in the .w components.
// leafid: exclusive scan of the leaf heads (leaf of point j), tileid: exclusive scan of the tile heads.
__global__ void k_tile_boxes(const unsigned long long* __restrict__ keys, const int* __restrict__ tilehead, const int* __restrict__ tileid,
                             const int* __restrict__ leafhead, const int* __restrict__ leafid, const int* __restrict__ leaf_first,
                             const int64_t* __restrict__ offs, size_t n, const float4* __restrict__ llo, const float4* __restrict__ lhi,
                             float4* __restrict__ tlo, float4* __restrict__ thi)
{
    for (size_t j = blockIdx.x * (size_t)blockDim.x + threadIdx.x; j < n; j += (size_t)gridDim.x * blockDim.x) {
        if (!tilehead[j]) continue;
        const int cloud = (int)(keys[j] >> 42);
        const size_t end = (size_t)offs[cloud + 1];
        const int l0 = leafid[j];
        float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
        int nl = 0;
        for (size_t u = j; u < end && (u == j || !tilehead[u]); ++u) {
            if (!leafhead[u]) continue;
            const float4 a = llo[l0 + nl], b = lhi[l0 + nl];
            lo[0] = fminf(lo[0], a.x); lo[1] = fminf(lo[1], a.y); lo[2] = fminf(lo[2], a.z);
            hi[0] = fmaxf(hi[0], b.x); hi[1] = fmaxf(hi[1], b.y); hi[2] = fmaxf(hi[2], b.z);
            ++nl;
        }
        const int tid = tileid[j];
        tlo[tid] = make_float4(lo[0], lo[1], lo[2], __int_as_float(l0 - leaf_first[cloud]));
        thi[tid] = make_float4(hi[0], hi[1], hi[2], __int_as_float(nl));
    }
}

// head[j] = 1 if point j starts a leaf: first of its cell, or a multiple of 8 points into it
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
