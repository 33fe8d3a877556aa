// gicp.hip -- batched GICP refinement for gfx950.
//
// Behaviour reproduced: fast_gicp's FastGICP (un-vendored submodule of the reference; algorithm
// per SURVEY.md App. A.2) as configured at Mapping/src/global_manager/src/global_manager.cpp:
// 2435-2443 and LoopDetection/src/RING_ros/main_RING.py:81-104:
//   G2 calculate_covariances : kNN (k) in the own cloud, covariance of the neighbours (double),
//                              PLANE regularisation U diag(1,1,1e-3) V^T == I - 0.999 n n^T
//   G3 update_correspondences: 1-NN of the float-transformed source point, reject d^2 >= max^2,
//                              M_i = (C_B + R C_A R^T)^-1
//   G4 linearize             : e = b - T a, J = [skew(T a) | -I], H += J^T M J, b += J^T M e
//   G5 LM optimiser          : LsqRegistration::step_lm / is_converged / se3_exp
//   G6 getFitnessScore       : mean squared NN distance with d^2 <= max_range
//
// Design: kd-tree-free.  Every nearest-neighbour query is an exact brute-force scan: target
// points are staged through LDS in tiles of 1024 float4 and read back as wave-wide broadcasts
// (one ds_read_b128 per candidate per wave), four source points per lane, ~7 VALU ops per
// (source, target) pair; at 120k x 120k the un-culled scan runs at ~95 % of the VALU issue peak.
// Both clouds are stored in Morton order (rocPRIM radix sort at set_clouds time), so a tile of 1024
// consecutive points is spatially compact and carries an axis-aligned bounding box: a workgroup
// (1024 consecutive, i.e. equally compact, queries; a wave owns 4 x 64 consecutive ones) visits the tiles
// in order of increasing box-to-box distance, stops at the first tile beyond its largest search radius,
// and inside a staged tile every wave skips the 128- and 16-candidate sub-tiles (boxes built while
// staging) that none of its lanes can use.  Every query starts from the neighbour it had in the previous
// pass (any target point is an upper bound).  The culling is conservative, so the neighbours are still
// the exact ones.  The scan is VALU-bound (7 lane-ops per surviving (source, target) pair);
// the per-point 3x3 algebra and the 28-term fp64 reductions (wave __shfl butterflies -> one
// partial per workgroup -> fixed-order final sum) are noise next to it.  All pairs of a batch
// advance together; the Levenberg-Marquardt bookkeeping runs on the device (one lane per
// pair), so the host only polls two counters (pairs to linearise, pairs in an LM trial).
// LM trial semantics are upstream's: linearize(x0) is the only step that searches (one NN pass per
// outer iteration); every trial pose delta * x0 is scored by compute_error, i.e. on the cached
// correspondences with the Mahalanobis matrices of the linearisation pose (recomputed on the fly
// from x0 -- the same values, cheaper than storing 48 B per point).
#include <hipcub/hipcub.hpp>

#include <cfloat>
#include <cstdlib>
#include <cmath>

#include "common.hpp"
#include "nn_core.hpp"
#include "eig3.hpp"

namespace {

constexpr int kTile = 1024;      // target points per LDS tile
constexpr int kNNThreads = 256;  // lanes per workgroup
constexpr int kPts = 4;          // source points per lane
constexpr int kTerms = 28;       // 21 (H upper) + 6 (b) + 1 (error)

struct LmState {
    double x[16];      // accepted pose (row-major 4x4) = linearisation pose of the current outer iteration
    double xi[16];     // pose being evaluated: == x while linearising (phase 0), the LM candidate in phase 1
    double delta[16];  // last increment
    double H[36];      // linearisation at x
    double b[6];
    double d[6];       // last LM step
    double y0;
    double lambda;
    double nu;
    double final_H[36];
    int phase;         // 0: linearize at x (NN search + H, b, y0), 1: LM trial (compute_error at xi), 2: done
    int inner;
    int outer;
    int trials;
    int converged;
    int failed;
    int active;
    int pad;
};

struct GicpParams {
    double max_corr2;     // squared correspondence distance threshold (inf if unbounded)
    double rot_eps, trans_eps;
    double conv_factor;   // upstream is_converged: factor 10 on both scaled deltas
    double lm_init_factor;
    int max_iter;
    int lm_max_iter;
    int force_iters;      // >0: run exactly this many outer iterations, no convergence test
    int k;
    double voxel_res;     // > 0: VGICP (voxelised target, G7); 0: GICP
    int voxel_neighbors;  // 1, 7 or 27 (DIRECT1 / DIRECT7 / DIRECT27)
    float cert_margin;    // metres the round-4 search looks beyond the neighbour it found (what later passes certify against)
    float motion_switch;  // round-4 schedule: a pair whose last step moved it farther than this (metres) is searched by the round-3 kernel
    int pad2;
};

// How far the last accepted LM increment moved the source cloud: |translation| + rotation angle x 60 m (metres, an upper estimate for
// points within 60 m of the origin).  Decides which search a pair gets in the round-4 schedule (nn_pass).
__device__ __forceinline__ float pair_motion(const LmState& S)
{
    const double tx = S.delta[3], ty = S.delta[7], tz = S.delta[11];
    const double c = fmin(fmax(0.5 * (S.delta[0] + S.delta[5] + S.delta[10] - 1.0), -1.0), 1.0);
    return (float)(sqrt(tx * tx + ty * ty + tz * tz) + 60.0 * sqrt(fmax(2.0 - 2.0 * c, 0.0)));
}

__device__ __forceinline__ double wave_sum_d(double v)
{
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__device__ __forceinline__ float dist2(float qx, float qy, float qz, const float4& b)
{
    const float dx = qx - b.x, dy = qy - b.y, dz = qz - b.z;
    return __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
}

// PLANE regularisation (the only one implemented, fast_gicp's default): C = U diag(1, 1, 1e-3) V^T = I - 0.999 n n^T with n the unit normal.
// The library keeps n (3 doubles, 24 B per point) instead of the 6 doubles of C: every reader rebuilds C with THESE expressions (fp64,
// no contraction: -ffp-contract=off), i.e. the very doubles the covariance kernels used to store -- half the bytes k_linearize streams and gathers.
constexpr int kCovDoubles = 3;
__host__ __device__ __forceinline__ void cov6_from_normal(const double* __restrict__ n, double (&c)[6])
{
    const double n0 = n[0], n1 = n[1], n2 = n[2];
    c[0] = 1.0 - 0.999 * n0 * n0;
    c[1] = -0.999 * n0 * n1;
    c[2] = -0.999 * n0 * n2;
    c[3] = 1.0 - 0.999 * n1 * n1;
    c[4] = -0.999 * n1 * n2;
    c[5] = 1.0 - 0.999 * n2 * n2;
}

// Bounding boxes of the Morton-ordered cloud at two granularities, in global memory (built once per set_clouds by k_boxes):
//   tile t  = points [1024 t, 1024 t + 1024),  mini 64 t + m = points [1024 t + 16 m, + 16)   (slots past the cloud: empty boxes, lo = +inf, hi = -inf)
struct Hier {
    const float4* tlo;  // [ntiles]
    const float4* thi;
    const float4* mlo;  // [64 * ntiles]
    const float4* mhi;
    int ntiles;
};

__device__ __forceinline__ float box_point_d2(const float4& lo, const float4& hi, float x, float y, float z)
{
    const float dx = fmaxf(fmaxf(lo.x - x, x - hi.x), 0.0f);
    const float dy = fmaxf(fmaxf(lo.y - y, y - hi.y), 0.0f);
    const float dz = fmaxf(fmaxf(lo.z - z, z - hi.z), 0.0f);
    return dx * dx + dy * dy + dz * dz;
}

// lower bound of the squared distance between any point of box (lo, hi) and any point of box (qlo, qhi)
__device__ __forceinline__ float box_box_d2(const float4& lo, const float4& hi, const float (&qlo)[3], const float (&qhi)[3])
{
    const float dx = fmaxf(fmaxf(lo.x - qhi[0], qlo[0] - hi.x), 0.0f);
    const float dy = fmaxf(fmaxf(lo.y - qhi[1], qlo[1] - hi.y), 0.0f);
    const float dz = fmaxf(fmaxf(lo.z - qhi[2], qlo[2] - hi.z), 0.0f);
    return dx * dx + dy * dy + dz * dz;
}

// bounding box of a wave's live queries (every lane returns the same values; +inf / -inf without live queries)
__device__ __forceinline__ void wave_bbox(float (&lo)[3], float (&hi)[3])
{
#pragma unroll
    for (int a = 0; a < 3; ++a)
        for (int o = 32; o > 0; o >>= 1) {
            lo[a] = fminf(lo[a], __shfl_xor(lo[a], o, 64));
            hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], o, 64));
        }
}

__device__ __forceinline__ float wave_max(float v)
{
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// ---- the traversal of the k-NN selection ------------------------------------------------------------------------------------------
// No LDS tile, no workgroup barrier: a WAVE (its queries are 64 consecutive Morton-ordered points, i.e. spatially compact) walks the
// two-level box hierarchy on its own.  What bounded the round-3 / round-4 form of this walk (one bounding box per wave, a per-mini test, then
// the mini's candidates) was not arithmetic but the LENGTH OF ITS DEPENDENCY CHAINS: a workgroup of k_knn_cov<30> lived 0.9 ms (0.55 ms now)
// because every mini cost two dependent round trips to the L2 (its box -> test -> its 16 candidates -> distances), taken one after the other.
// And a wave whose 64 queries straddle a jump of the Morton curve has a bounding box the size of the scene: tested against THAT box, every mini
// of the cloud passed the coarse test and was then rejected one round trip at a time -- 3 ms for one wave, the tail of the whole launch.
//   * coarse tests against QUAD boxes: the queries of 4 consecutive lanes share a box and a bound (16 per wave; a jump of the curve
//     spoils one of them, not the wave).  Lane l tests tile / mini l against the 16 quads (v_readlane broadcasts, ~20 VALU instructions
//     per quad): one ballot per 64 boxes and NO per-mini test afterwards -- whatever passes is evaluated;
//   * tiles nearest first (smallest box distance to any quad) inside a chunk of 64 tiles, the chunk of the wave's own tile first; the
//     bounds are asked again before every tile (`bound()`), so a lane whose seed was poor holds the walk only until its neighbours' tile
//     has been seen;
//   * the 16 candidates of a mini arrive by scalar loads in two halves, the next half REQUESTED BEFORE the current one is evaluated
//     (scalar loads return out of order, so only lgkmcnt(0) exists: an empty asm that reads one register of the current half makes the
//     compiler wait for it before the next requests are issued -- everything outstanding during the arithmetic belongs to the next half).
// Conservative at every level (0.9999 slack on the box distances), so the neighbours found are the exact ones.
#ifndef MRS_KNN_QUAD
#define MRS_KNN_QUAD 4
#endif
constexpr int kQL = MRS_KNN_QUAD;        // lanes per group of the coarse tests (4: "quads"; 8 was measured: see DESIGN.md 4)
constexpr int kQG = 64 / kQL;
__device__ __forceinline__ float lane_f(float v, int l) { return __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(v), l)); }
__device__ __forceinline__ float first_f(float v) { return __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(v))); }

// box of the live queries of this lane's group of 4 consecutive lanes (empty: +inf / -inf)
__device__ __forceinline__ void quad_box(bool live, const float4& q, float (&lo)[3], float (&hi)[3])
{
    lo[0] = live ? q.x : INFINITY; lo[1] = live ? q.y : INFINITY; lo[2] = live ? q.z : INFINITY;
    hi[0] = live ? q.x : -INFINITY; hi[1] = live ? q.y : -INFINITY; hi[2] = live ? q.z : -INFINITY;
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int o = 1; o < kQL; o <<= 1) {
            lo[a] = fminf(lo[a], __shfl_xor(lo[a], o, 64));
            hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], o, 64));
        }
}
__device__ __forceinline__ float quad_max(float v)
{
#pragma unroll
    for (int o = 1; o < kQL; o <<= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// does THIS lane's box (blo, bhi) come within the bound of any query of the wave?  16 quad tests (box against the quad's box and largest
// bound); a quad in `wide` (its box is larger than its bound: the 4 points straddle a jump of the Morton curve, and everything between the
// two ends of the jump would "touch" the box) is tested query by query instead.  dmin: the smallest distance seen.
__device__ __forceinline__ bool quads_hit(const float4& blo, const float4& bhi, const float (&qlo)[3], const float (&qhi)[3], float qT,
                                          unsigned wide, const float4& q, float T, float& dmin)
{
    bool hit = false;
    dmin = INFINITY;
#pragma unroll
    for (int g = 0; g < kQG; ++g) {
        if (wide >> g & 1) {        // wave-uniform
#pragma unroll
            for (int l = kQL * g; l < kQL * g + kQL; ++l) {
                const float d = box_point_d2(blo, bhi, lane_f(q.x, l), lane_f(q.y, l), lane_f(q.z, l));
                hit |= d * 0.9999f <= lane_f(T, l);         // a dead lane's T is -1
                dmin = fminf(dmin, lane_f(T, l) >= 0.0f ? d : INFINITY);
            }
        } else {
            const float l[3] = {lane_f(qlo[0], kQL * g), lane_f(qlo[1], kQL * g), lane_f(qlo[2], kQL * g)};
            const float h[3] = {lane_f(qhi[0], kQL * g), lane_f(qhi[1], kQL * g), lane_f(qhi[2], kQL * g)};
            const float d = box_box_d2(blo, bhi, l, h);
            hit |= d * 0.9999f <= lane_f(qT, kQL * g);
            dmin = fminf(dmin, d);
        }
    }
    return hit;
}

// quads whose box is larger than their bound (bit g: lanes 4g .. 4g + 3)
__device__ __forceinline__ unsigned wide_quads(const float (&qlo)[3], const float (&qhi)[3], float qT)
{
    const float ex = qhi[0] - qlo[0], ey = qhi[1] - qlo[1], ez = qhi[2] - qlo[2];
    const bool w = ex * ex + ey * ey + ez * ez > qT;           // (an empty quad: -inf extents, inf > -1: tested lane by lane, every lane dead)
    const unsigned long long m = __ballot(w);
    unsigned out = 0;
#pragma unroll
    for (int g = 0; g < kQG; ++g) out |= (unsigned)(m >> (kQL * g) & 1ull) << g;
    return out;
}

struct Cand8 { float4 c[8]; };
__device__ __forceinline__ void cand_request(Cand8& o, const float4* __restrict__ pts, int j0)       // wave-uniform j0: scalar loads
{
#pragma unroll
    for (int u = 0; u < 8; ++u) o.c[u] = pts[j0 + u];       // 128 contiguous bytes; past the cloud's end: the next cloud's points or the 16 points of
                                                            // slack behind the last one (prepare_side), masked in cand_dist
    asm volatile("" ::: "memory");        // the requests stay where they are written
}
__device__ __forceinline__ void cand_arrived(const Cand8& a)
{
    // a use of the half: the compiler's s_waitcnt lgkmcnt(0) lands HERE, before the next requests.  The .w lanes (never read by the arithmetic)
    // are named too: left dead, the register allocator hands them out as scratch while the loads are in flight, and every such write
    // costs a wait for everything outstanding
    asm volatile("" ::"s"(a.c[0].x), "s"(a.c[0].w), "s"(a.c[1].w), "s"(a.c[2].w), "s"(a.c[3].w), "s"(a.c[4].w), "s"(a.c[5].w), "s"(a.c[6].w), "s"(a.c[7].w) : "memory");
}
__device__ __forceinline__ void cand_dist(const Cand8& a, int j0, int n, float qx, float qy, float qz, float (&dd)[8])
{
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const float dx = qx - a.c[u].x, dy = qy - a.c[u].y, dz = qz - a.c[u].z;
        dd[u] = __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
    }
    if (j0 + 8 > n) {           // the cloud's last points (wave-uniform): what was read beyond them does not exist
#pragma unroll
        for (int u = 0; u < 8; ++u) dd[u] = j0 + u < n ? dd[u] : INFINITY;
    }
}

// the candidates of the minis `ids(r)`, r = 0 .. count - 1 (wave-uniform), 8 at a time: visit(first index, the 8 candidates).  The visitor
// starts with cand_pin() on a coordinate of its query: its arithmetic then stays behind the requests for the next 8.
__device__ __forceinline__ void cand_pin(float& x) { asm volatile("" : "+v"(x)); }      // (volatile asm statements keep their order)
template <class Ids, class Visit>
__device__ __forceinline__ void stream_minis(const float4* __restrict__ pts, int count, Ids ids, Visit visit)
{
    if (count <= 0) return;
    Cand8 A, B;
    int j0 = ids(0) * 16;
    cand_request(A, pts, j0);
    for (int r = 0; r < count; ++r) {
        cand_arrived(A);
        cand_request(B, pts, j0 + 8);
        visit(j0, A);
        cand_arrived(B);
        const int jn = r + 1 < count ? ids(r + 1) * 16 : j0;
        if (r + 1 < count) cand_request(A, pts, jn);
        visit(j0 + 8, B);
        j0 = jn;
    }
}

// Walk of the hierarchy for the 64 queries of a wave.  bound(): the lane's current bound (called by the whole wave before every tile; it may
// do wave-wide bookkeeping first); a dead lane's bound is ignored.  visit(first index, 8 candidates): see stream_minis.  NEAREST: tiles nearest
// first (pass 1: bounds shrink); otherwise tiles and minis in index order (pass 2 without the list of pass 1: candidates must arrive in
// ascending index order).  rec(id): every mini visited.  Returns the number of tiles visited (a development counter).
template <bool NEAREST, class Bound, class Visit, class Rec>
__device__ __forceinline__ int knn_walk(const float4* __restrict__ pts, int n, const Hier& H, bool live, const float4& q, int home_tile,
                                         Bound bound, Visit visit, Rec rec)
{
    const int lane = threadIdx.x & 63;
    float qlo[3], qhi[3];
    quad_box(live, q, qlo, qhi);
    const int nchunks = (H.ntiles + 63) >> 6;
    const int hc = min(home_tile, H.ntiles - 1) >> 6;
    int visited = 0;        // tiles (returned: a development counter)
    for (int ci = 0; ci < nchunks; ++ci) {
        const int ch = !NEAREST ? ci : (ci == 0 ? hc : (ci <= hc ? ci - 1 : ci));       // NEAREST: the chunk of the wave's own tile first
        const int t = ch * 64 + lane;
        float T = bound();
        T = live ? T : -1.0f;
        float qT = quad_max(T);
        unsigned wide = wide_quads(qlo, qhi, qT);
        float key = INFINITY;           // box distance of a tile still to be visited; +inf: not (or no longer) a candidate
        {
            const int tc = min(t, H.ntiles - 1);
            float dmin;
            const bool hit = quads_hit(H.tlo[tc], H.thi[tc], qlo, qhi, qT, wide, q, T, dmin);
            if (hit && t < H.ntiles) key = dmin;
        }
        unsigned long long tmask = __ballot(key < INFINITY);
        while (tmask) {
            int tl;
            if (NEAREST) {
                float best = key;
                for (int o = 32; o > 0; o >>= 1) best = fminf(best, __shfl_xor(best, o, 64));
                best = first_f(best);
                tl = (int)__builtin_ctzll(__ballot(key == best));
                T = bound();
                T = live ? T : -1.0f;
                qT = quad_max(T);
                wide = wide_quads(qlo, qhi, qT);
                if (!(best * 0.9999f <= first_f(wave_max(T)))) break;        // every tile left is at least as far from every query
            } else {
                tl = (int)__builtin_ctzll(tmask);
            }
            tmask &= ~(1ull << tl);
            ++visited;
            if (lane == tl) key = INFINITY;
            const int tt = ch * 64 + tl;
            float dmin;
            const bool mh = quads_hit(H.mlo[tt * 64 + lane], H.mhi[tt * 64 + lane], qlo, qhi, qT, wide, q, T, dmin);
            // (minis past the cloud's end have empty boxes, at distance +inf -- which an infinite bound, a cloud smaller than k, would accept)
            const unsigned long long mmask = __ballot(mh && (tt * 64 + lane) * 16 < n);
            const int cnt = __builtin_popcountll(mmask);
            unsigned long long left = mmask;         // ids(r) is asked for r = 0, 1, 2, ... in turn
            stream_minis(pts, cnt, [&](int) { const int m = (int)__builtin_ctzll(left); left &= left - 1; rec(tt * 64 + m); return tt * 64 + m; }, visit);
        }
    }
    return visited;
}

// Exact 1-NN of P query points per lane over the Morton-ordered cloud tgt[0..m): squared distance and (sorted-space) index; `maxc2` is
// the rejection radius (inf = none).  seed: any valid target index per query (last pass's neighbour, or the Morton seed of a cold
// start): its distance is the initial bound.  Waves are independent (no barrier inside).
constexpr int kNNRejMax = 256;     // minis a wave of nn_scan may reject before it changes to the quad-box walk
template <int P>
__device__ __forceinline__ void nn_scan(const float4* __restrict__ tgt, int m, const Hier& H, float maxc2,
                                        const float (&qx)[P], const float (&qy)[P], const float (&qz)[P],
                                        const bool (&live)[P], float (&best)[P], int (&bidx)[P], const int (&seed)[P])
{
    int grp[P];
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    float r = 0.0f;
#pragma unroll
    for (int p = 0; p < P; ++p) {
        best[p] = INFINITY; grp[p] = -1;
        if (live[p]) {
            lo[0] = fminf(lo[0], qx[p]); hi[0] = fmaxf(hi[0], qx[p]);
            lo[1] = fminf(lo[1], qy[p]); hi[1] = fmaxf(hi[1], qy[p]);
            lo[2] = fminf(lo[2], qz[p]); hi[2] = fmaxf(hi[2], qz[p]);
            // warm start: any target point is an upper bound; last pass's neighbour is nearly always the winner
            if (seed[p] >= 0 && seed[p] < m) { best[p] = dist2(qx[p], qy[p], qz[p], tgt[seed[p]]); grp[p] = seed[p] & ~7; }
            r = fmaxf(r, fminf(best[p], maxc2));
        }
    }
    wave_bbox(lo, hi);
    const float reach = wave_max(r);
    // the P queries of a lane share one traversal: `need` / `visit` loop over them
    auto need = [&](const float4& blo, const float4& bhi) {
        bool w = false;
#pragma unroll
        for (int p = 0; p < P; ++p) w |= live[p] && box_point_d2(blo, bhi, qx[p], qy[p], qz[p]) * 0.9999f <= fminf(best[p], maxc2);
        return w;
    };
    const int lane = threadIdx.x & 63;
    // A wave whose queries straddle a jump of the Morton curve has a box the size of the scene: every mini passes the two coarse tests and is
    // then rejected by `need`, one dependent round trip each (such a workgroup lived 1.4 ms, the median one 0.08 ms: the tail of every
    // launch, and most of a small one).  The walk counts its rejections; past kNNRejMax it is abandoned for the k-NN selection's walk
    // (quad boxes, per-query tests at the jump, nearest tile first), which starts over with the bounds found so far.
    int rejected = 0;
    for (int tb = 0; tb < H.ntiles && rejected <= kNNRejMax; tb += 64) {
        const int t = tb + lane;
        bool hit = false;
        if (t < H.ntiles) hit = box_box_d2(H.tlo[t], H.thi[t], lo, hi) * 0.9999f <= reach;
        unsigned long long tmask = __ballot(hit);
        while (tmask && rejected <= kNNRejMax) {
            const int tt = tb + (int)__builtin_ctzll(tmask);
            tmask &= tmask - 1;
            const bool mhit = box_box_d2(H.mlo[tt * 64 + lane], H.mhi[tt * 64 + lane], lo, hi) * 0.9999f <= reach;
            unsigned long long mmask = __ballot(mhit);
            while (mmask) {
                const int mm = (int)__builtin_ctzll(mmask);
                mmask &= mmask - 1;
                if (!__any(need(H.mlo[tt * 64 + mm], H.mhi[tt * 64 + mm]))) { ++rejected; continue; }
                const int j0 = tt * kTile + mm * 16;
                float4 c16[16];        // all 16 candidates requested before the first use (wave-uniform addresses: scalar loads)
#pragma unroll
                for (int u = 0; u < 16; ++u) c16[u] = j0 + u < m ? tgt[j0 + u] : make_float4(INFINITY, INFINITY, INFINITY, 0.f);
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const float4* c = c16 + 8 * h;
#pragma unroll
                    for (int p = 0; p < P; ++p) {
                        float d[8];
#pragma unroll
                        for (int u = 0; u < 8; ++u) d[u] = dist2(qx[p], qy[p], qz[p], c[u]);
                        const float mn = fminf(fminf(fminf(d[0], d[1]), fminf(d[2], d[3])), fminf(fminf(d[4], d[5]), fminf(d[6], d[7])));
                        if (mn < best[p]) { best[p] = mn; grp[p] = j0 + 8 * h; }
                    }
                }
            }
        }
    }
    if (rejected > kNNRejMax) {
#pragma unroll
        for (int p = 0; p < P; ++p) {
            const float4 q = make_float4(qx[p], qy[p], qz[p], 0.f);
            const int home_tile = __builtin_amdgcn_readfirstlane(max(seed[p], 0)) >> 10;
            knn_walk<true>(tgt, m, H, live[p], q, home_tile, [&]() { return fminf(best[p], maxc2); },
                           [&](int j0, const Cand8& cand) {
                               float x = q.x, d[8];
                               cand_pin(x);
                               cand_dist(cand, j0, m, x, q.y, q.z, d);
                               const float mn = fminf(fminf(fminf(d[0], d[1]), fminf(d[2], d[3])), fminf(fminf(d[4], d[5]), fminf(d[6], d[7])));
                               if (live[p] && mn < best[p]) { best[p] = mn; grp[p] = j0; }
                           },
                           [](int) {});
        }
    }
    // resolve the index inside the winning group of 8
#pragma unroll
    for (int p = 0; p < P; ++p) {
        bidx[p] = -1;
        if (grp[p] >= 0) {
            for (int u = 7; u >= 0; --u) {
                const int j = grp[p] + u;
                if (j < m && dist2(qx[p], qy[p], qz[p], tgt[j]) == best[p]) bidx[p] = j;
            }
        }
    }
}

// ---- Morton ordering of the clouds (set_clouds) ------------------------------------------------
__device__ __forceinline__ int float_to_ordered(float f)
{
    const int i = __float_as_int(f);
    return i >= 0 ? i : i ^ 0x7fffffff;
}
__device__ __forceinline__ float ordered_to_float(int i) { return __int_as_float(i >= 0 ? i : i ^ 0x7fffffff); }

// per-cloud bounding box: bbox[c] = {min x,y,z, max x,y,z} as ordered ints (pre-set to +-max)
__global__ void k_cloud_bbox(const float* __restrict__ src, int stride, const int64_t* __restrict__ offs, int* __restrict__ bbox)
{
    const int c = blockIdx.y;
    const int64_t o = offs[c];
    const int n = (int)(offs[c + 1] - o);
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const float* p = src + (size_t)(o + i) * stride;
#pragma unroll
        for (int a = 0; a < 3; ++a) { lo[a] = fminf(lo[a], p[a]); hi[a] = fmaxf(hi[a], p[a]); }
    }
    // wave butterflies -> one partial per wave in LDS -> ONE set of atomics per workgroup (six hot addresses per cloud)
    __shared__ float red[4][6];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        for (int s = 32; s > 0; s >>= 1) {
            lo[a] = fminf(lo[a], __shfl_xor(lo[a], s, 64));
            hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], s, 64));
        }
        if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6][a] = lo[a]; red[threadIdx.x >> 6][3 + a] = hi[a]; }
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        const int a = threadIdx.x;
        if (a < 3)
            atomicMin(&bbox[6 * c + a], float_to_ordered(fminf(fminf(red[0][a], red[1][a]), fminf(red[2][a], red[3][a]))));
        else
            atomicMax(&bbox[6 * c + a], float_to_ordered(fmaxf(fmaxf(red[0][a], red[1][a]), fmaxf(red[2][a], red[3][a]))));
    }
}

__device__ __forceinline__ unsigned long long spread3(unsigned v)  // 14 bits -> every third bit
{
    unsigned long long x = v & 0x3fffu;
    x = (x | (x << 32)) & 0x1f00000000ffffull;
    x = (x | (x << 16)) & 0x1f0000ff0000ffull;
    x = (x | (x << 8)) & 0x100f00f00f00f00full;
    x = (x | (x << 4)) & 0x10c30c30c30c30c3ull;
    x = (x | (x << 2)) & 0x1249249249249249ull;
    return x;
}

// 42-bit Morton code of (x, y, z) on the cubic grid (origin lo, scale sc) of a cloud
__device__ __forceinline__ unsigned long long morton42(float x, float y, float z, const float (&lo)[3], float sc)
{
    const float p[3] = {x, y, z};
    unsigned q[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float v = (p[a] - lo[a]) * sc;
        q[a] = v >= 0.0f ? (unsigned)fminf(v, 16383.0f) : 0u;  // NaN -> 0
    }
    return spread3(q[0]) | (spread3(q[1]) << 1) | (spread3(q[2]) << 2);
}

__device__ __forceinline__ float morton_grid(const int* __restrict__ bbox, int c, float (&lo)[3])
{
    float ext = 0.0f;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        lo[a] = ordered_to_float(bbox[6 * c + a]);
        ext = fmaxf(ext, ordered_to_float(bbox[6 * c + 3 + a]) - lo[a]);
    }
    return ext > 0.0f ? 16383.0f / ext : 0.0f;
}

// Cold start of the NN scan: the target point whose Morton code is closest to the query's (binary search over the
// Morton-ordered cloud, codes recomputed from the points) or its predecessor, whichever is nearer.  Any index is
// a valid upper bound; this one is usually within a cell or two of the true neighbour.
__device__ __forceinline__ int morton_seed(const float4* __restrict__ tgt, int m, float qx, float qy, float qz,
                                           const float (&lo)[3], float sc)
{
    const unsigned long long key = morton42(qx, qy, qz, lo, sc);
    int a = 0, b = m;  // first index with code >= key
    while (a < b) {
        const int mid = (a + b) >> 1;
        const float4 t = tgt[mid];
        if (morton42(t.x, t.y, t.z, lo, sc) < key) a = mid + 1; else b = mid;
    }
    const int j1 = min(a, m - 1), j0 = max(j1 - 1, 0);
    return dist2(qx, qy, qz, tgt[j0]) < dist2(qx, qy, qz, tgt[j1]) ? j0 : j1;
}

// key = cloud id (high bits) | 42-bit Morton code on a cubic grid spanning the cloud's bounding box
__global__ void k_morton_keys(const float* __restrict__ src, int stride, const int64_t* __restrict__ offs,
                              const int* __restrict__ bbox, unsigned long long* __restrict__ keys, int* __restrict__ vals)
{
    const int c = blockIdx.y;
    const int64_t o = offs[c];
    const int n = (int)(offs[c + 1] - o);
    float lo[3];
    const float sc = morton_grid(bbox, c, lo);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const float* p = src + (size_t)(o + i) * stride;
        keys[o + i] = ((unsigned long long)c << 42) | morton42(p[0], p[1], p[2], lo, sc);
        vals[o + i] = (int)(o + i);
    }
}

// dst[k] = point perm[k]; .w carries its ORIGINAL cloud-local index
__global__ void k_gather_sorted(const float* __restrict__ src, int stride, const int64_t* __restrict__ offs,
                                const int* __restrict__ perm, float4* __restrict__ dst)
{
    const int c = blockIdx.y;
    const int64_t o = offs[c];
    const int n = (int)(offs[c + 1] - o);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int g = perm[o + i];
        const float* p = src + (size_t)g * stride;
        dst[o + i] = make_float4(p[0], p[1], p[2], __int_as_float(g - (int)o));
    }
}

// bounding boxes of every 1024-point tile and of its 64 minis of 16 points; tile_base[c] = first tile of cloud c (minis: 64 x that)
__global__ __launch_bounds__(256) void k_boxes(const float4* __restrict__ pts, const int64_t* __restrict__ offs, const int* __restrict__ tile_base,
                                               float4* __restrict__ tlo, float4* __restrict__ thi, float4* __restrict__ mlo,
                                               float4* __restrict__ mhi)
{
    __shared__ float red[4][6];
    const int c = blockIdx.y;
    const int64_t o = offs[c];
    const int n = (int)(offs[c + 1] - o);
    const int t = blockIdx.x;
    if (t * kTile >= n) return;
    const size_t tile = (size_t)tile_base[c] + t;
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
#pragma unroll
    for (int j = 0; j < kTile / 256; ++j) {
        const int i = t * kTile + j * 256 + (int)threadIdx.x;      // 16 consecutive lanes = one mini
        const bool in = i < n;
        const float4 p = pts[o + (in ? i : 0)];
        float b[6] = {in ? p.x : INFINITY, in ? p.y : INFINITY, in ? p.z : INFINITY, in ? p.x : -INFINITY, in ? p.y : -INFINITY, in ? p.z : -INFINITY};
#pragma unroll
        for (int a = 0; a < 3; ++a)
            for (int s = 1; s < 16; s <<= 1) {
                b[a] = fminf(b[a], __shfl_xor(b[a], s, 64));
                b[3 + a] = fmaxf(b[3 + a], __shfl_xor(b[3 + a], s, 64));
            }
        if ((threadIdx.x & 15) == 0) {
            const size_t mi = tile * 64 + (size_t)((j * 256 + (int)threadIdx.x) >> 4);
            mlo[mi] = make_float4(b[0], b[1], b[2], 0.f);
            mhi[mi] = make_float4(b[3], b[4], b[5], 0.f);
        }
#pragma unroll
        for (int a = 0; a < 3; ++a) { lo[a] = fminf(lo[a], b[a]); hi[a] = fmaxf(hi[a], b[3 + a]); }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        for (int s = 32; s > 0; s >>= 1) {
            lo[a] = fminf(lo[a], __shfl_xor(lo[a], s, 64));
            hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], s, 64));
        }
        if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6][a] = lo[a]; red[threadIdx.x >> 6][3 + a] = hi[a]; }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float l[3], h[3];
        for (int a = 0; a < 3; ++a) {
            l[a] = fminf(fminf(red[0][a], red[1][a]), fminf(red[2][a], red[3][a]));
            h[a] = fmaxf(fmaxf(red[0][3 + a], red[1][3 + a]), fmaxf(red[2][3 + a], red[3][3 + a]));
        }
        tlo[tile] = make_float4(l[0], l[1], l[2], 0.f);
        thi[tile] = make_float4(h[0], h[1], h[2], 0.f);
    }
}

// smallest_eigvec / sym3_eigvals: eig3.hpp (closed form + Rayleigh-quotient steps; the cyclic Jacobi of rounds 1-5 is gone)

// ---- exact k nearest neighbours in two passes ----------------------------------------------------------------------------------
// A sorted insertion that carries the index costs ~8 VALU instructions per slot and, in SIMT, every lane of a wave pays for every
// lane's insertions.  Here the scan keeps only the KMAX smallest DISTANCES, sorted, by a chain of v_med3_f32
// (dk[s] = med3(dk[s-1], d, dk[s]): one instruction per slot, no predicate: a distance beyond the list leaves it unchanged).
//   pass 1 (the k-th distance tau): seed = the 64 neighbours along the Morton curve; then the hierarchy walk.  A candidate closer than
//     the lane's bound T = dk[KMAX - 1] is only NOTED in a lane-private LDS buffer (one predicated ds_write); the chain runs when some
//     lane's buffer is more than half full: once per candidate a LANE accepted, not once per candidate ANY lane of the wave accepted
//     (rounds 3-4: ~480 chain passes per wave, 15 k of its 44 k VALU instructions; now ~50).  T is refreshed at every flush; a stale T
//     is a valid bound (bounds only shrink), it merely lets a few more candidates through;
//   pass 2 (the indices): tau is exact, and so is the number of candidates strictly inside it (nless = #{dk[s] < tau}); the walk over
//     the minis pass 1 noted appends every candidate with d < tau and the first kk - nless exact ties with tau (they arrive in
//     ascending index order): never more than kk entries, whatever the number of duplicates (no overflow path);
//   order: the rank of a collected candidate is the number of dk[s] below its distance (+ the equal ones already placed: a bit mask),
//     2 VALU instructions per slot in place of the 8 of a (distance, index) insertion, and no index registers.
// Exactly equal distances resolve to the smaller (Morton-space) index.  The whole workgroup must call it together.

// development counters (MRS_KNN_DBG=1 prints them): per wave -- candidate groups of 8 visited in pass 1, groups in which some lane noted
// a candidate, chain passes (seed excluded), pass 2 groups, entries ranked, query waves
__device__ unsigned long long g_knn_dbg[8];
__device__ unsigned long long g_knn_clk[4];       // wave clocks spent in the seed / the pass-1 walk / pass 2; tiles visited in pass 1
__device__ int g_knn_dbg_on;       // set by the host when MRS_KNN_DBG is in the environment
__device__ unsigned long long g_nn_trace[2 * 65536];        // the same for k_nn_scan (MRS_NN_TRACE_FILE, written by mrs_gicp_batch_profile)
__device__ unsigned long long g_knn_trace[2 * 65536];       // MRS_KNN_DBG=1: (start, end) of every workgroup of the last launch on the 100 MHz wall clock
__device__ int g_knn_norec;        // development aid (MRS_KNN_REC=0): pass 2 walks the hierarchy again instead of revisiting pass 1's minis

constexpr int kHome = 64;         // Morton-curve neighbours that seed the bound (the walk skips exactly that index range)
constexpr int kKnnBuf = 16;       // LDS slots per lane for noted candidates (in the first slots of the index list: pass 1 is over before pass 2 writes)
constexpr int kKnnRec = 64;       // minis a wave can note in pass 1 for pass 2 (more: pass 2 walks the hierarchy again)
constexpr int kKnnBlk = 64;       // neighbour lists leave the selection in blocks of 64 points, slot-major (knn_at)

template <int KMAX>
__device__ __forceinline__ void dist_insert(float (&dk)[KMAX], float d)
{
#pragma unroll
    for (int s = KMAX - 1; s > 0; --s) dk[s] = __builtin_amdgcn_fmed3f(dk[s - 1], d, dk[s]);
    dk[0] = fminf(dk[0], d);
}

// (distance, index) ordered insertion (k_knn_select, the round-4 search core)
template <int KMAX>
__device__ __forceinline__ void knn_insert_tie(float (&dk)[KMAX], int (&ik)[KMAX], float d, int j)
{
#pragma unroll
    for (int s = KMAX - 1; s > 0; --s) {
        const bool up = dk[s - 1] > d || (dk[s - 1] == d && ik[s - 1] > j);
        const bool here = !up && (dk[s] > d || (dk[s] == d && ik[s] > j));
        dk[s] = up ? dk[s - 1] : (here ? d : dk[s]);
        ik[s] = up ? ik[s - 1] : (here ? j : ik[s]);
    }
    if (dk[0] > d || (dk[0] == d && ik[0] > j)) { dk[0] = d; ik[0] = j; }
}

// Neighbour lists between the selection and its consumers (k_cov_from_knn, k_feat_from_knn): cloud-local sorted-space indices, per cloud
// in blocks of 64 points, slot-major inside a block -- entry (point i, slot s) of cloud c (first point o) lies at
//   knn[(o + 64 c) k + (i / 64) 64 k + 64 s + i % 64]
// so that a wave's loads of one slot are ONE 256-byte row (as [point][k] every lane walked its own 4 k bytes: 64 lines per load
// instruction, re-fetched from HBM whenever the L1 / L2 lost them: k_feat_from_knn read 1.1 KB per point).  Room: knn_ints().
__host__ __device__ inline size_t knn_ints(int64_t points, int64_t clouds, int k) { return (size_t)(points + kKnnBlk * clouds) * (size_t)k; }
__device__ __forceinline__ size_t knn_at(int64_t o, int c, int i, int k, int s)
{
    return (size_t)(o + (int64_t)kKnnBlk * c) * k + (size_t)(i >> 6) * (kKnnBlk * k) + (size_t)(s << 6) + (size_t)(i & 63);
}

// The k nearest of point i (itself included) in (distance, index) order: emit(rank, index) once per neighbour, ranks 0 .. found - 1;
// returns the number found (< k only in a cloud with fewer than k points).  list: KMAX x kNNThreads ints of LDS, slot-major.
template <int KMAX, class Emit>
__device__ __forceinline__ int knn_two_pass(int* __restrict__ list, const float4* __restrict__ pts, int n, const Hier& H,
                                            int i, bool live, const float4& q, int k, int* __restrict__ rec_ids /* wave-private, kKnnRec ints of LDS, or null */,
                                            Emit emit)
{
    static_assert(KMAX <= 32 && KMAX >= kKnnBuf, "rank mask is 32 bits; the note buffer lives in the list");
    const int tid = (int)threadIdx.x;
    const long long t_begin = g_knn_dbg_on ? clock64() : 0;
    float dk[KMAX];
#pragma unroll
    for (int s = 0; s < KMAX; ++s) dk[s] = INFINITY;
    // pass 1: the KMAX smallest distances.  Seed: the 64 neighbours along the Morton curve, 8 loads in flight at a time
    const int home = max(0, min(i - kHome / 2, n - kHome));
    for (int u0 = 0; u0 < kHome; u0 += 8) {
        if (home + u0 >= n) break;             // n < kHome: wave-uniform
        float4 hp[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) hp[u] = pts[(live && home + u0 + u < n) ? home + u0 + u : 0];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const float d = dist2(q.x, q.y, q.z, hp[u]);
            dist_insert<KMAX>(dk, (live && home + u0 + u < n && d == d) ? d : INFINITY);
        }
    }
    const int home_tile = __builtin_amdgcn_readfirstlane(i) >> 10;      // kTile = 1024
    float* const buf = reinterpret_cast<float*>(list);
    int nb = 0;                     // candidates noted since the last flush
    int c_g1 = 0, c_ins = 0, c_chain = 0, c_g2 = 0;
    const long long t_seed = g_knn_dbg_on ? clock64() : 0;
    auto flush = [&]() {
#pragma unroll 1
        for (int s = 0; s < kKnnBuf; ++s) {
            if (!__any(s < nb)) break;
            const float v = s < nb ? buf[s * kNNThreads + tid] : INFINITY;
            dist_insert<KMAX>(dk, v);
            ++c_chain;
        }
        nb = 0;
    };
    int nrec = 0;       // wave-uniform
    const int c_tiles = knn_walk<true>(pts, n, H, live, q, home_tile,
               [&]() { if (__any(nb > 0)) flush(); return dk[KMAX - 1]; },       // before every tile: bounds up to date
               [&](int j0, const Cand8& cand) {
                   ++c_g1;
                   float qx = q.x, dd[8];
                   cand_pin(qx);
                   cand_dist(cand, j0, n, qx, q.y, q.z, dd);
                   const float T = dk[KMAX - 1];
                   const float mn = fminf(fminf(fminf(dd[0], dd[1]), fminf(dd[2], dd[3])), fminf(fminf(dd[4], dd[5]), fminf(dd[6], dd[7])));
                   if (!__any(live && mn < T)) return;
                   ++c_ins;
#pragma unroll
                   for (int u = 0; u < 8; ++u) {
                       const bool use = live && (unsigned)(j0 + u - home) >= (unsigned)kHome && dd[u] < T;       // NaN: false
                       if (use) { buf[nb * kNNThreads + tid] = dd[u]; ++nb; }
                   }
                   if (__any(nb > kKnnBuf - 8)) flush();
               },
               [&](int id) {       // every mini within some quad's bound of the moment (a superset of the minis within the final bounds)
                   if (rec_ids && nrec < kKnnRec && (threadIdx.x & 63) == 0) rec_ids[nrec] = id;
                   ++nrec;
               });
    if (__any(nb > 0)) flush();
    const long long t_walk = g_knn_dbg_on ? clock64() : 0;
    // pass 2: the candidates within the k-th distance, home range included, in ascending index order
    const int kk = k < KMAX ? k : KMAX;
    float tau = dk[KMAX - 1];
    int nless = 0;
#pragma unroll
    for (int s = 0; s < KMAX - 1; ++s) tau = (s == kk - 1) ? dk[s] : tau;
#pragma unroll
    for (int s = 0; s < KMAX; ++s) {
        dk[s] = s < kk ? dk[s] : INFINITY;        // the ranks below count dk[s] < d over every slot
        nless += dk[s] < tau ? 1 : 0;
    }
    if (!live) tau = -1.0f;
    const int room = kk - nless;                  // exact ties with tau that belong to the k nearest
    int cnt = 0, nt = 0;
    auto visit2 = [&](int j0, const Cand8& cand) {
        ++c_g2;
        float qx = q.x, dd[8];
        cand_pin(qx);
        cand_dist(cand, j0, n, qx, q.y, q.z, dd);
        const float mn = fminf(fminf(fminf(dd[0], dd[1]), fminf(dd[2], dd[3])), fminf(fminf(dd[4], dd[5]), fminf(dd[6], dd[7])));
        if (!__any(mn <= tau)) return;
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (dd[u] <= tau && dd[u] < INFINITY) {     // tau is +inf for a cloud with fewer than k points: padding stays out
                const bool tie = dd[u] == tau;
                if (!tie || nt < room) {
                    list[cnt * kNNThreads + tid] = j0 + u;
                    ++cnt;
                    nt += tie ? 1 : 0;
                }
            }
    };
    if (rec_ids && nrec <= kKnnRec) {       // the minis pass 1 visited, without walking the hierarchy again -- in INDEX order (ties resolve by arrival)
        nnc::wave_lds_sync();
        const int lane = threadIdx.x & 63;
        const int id = lane < nrec ? rec_ids[lane] : 0x7fffffff;
        int rank = 0;
        for (int m = 0; m < nrec; ++m) rank += __builtin_amdgcn_readlane(id, m) < id ? 1 : 0;
        nnc::wave_lds_sync();
        if (lane < nrec) rec_ids[rank] = id;
        nnc::wave_lds_sync();
        // (no box test: nearly every one of them holds a candidate of some lane, and the test would be a round trip per mini)
        stream_minis(pts, nrec, [&](int r) { return __builtin_amdgcn_readfirstlane(rec_ids[r]); }, visit2);
    } else {
        knn_walk<false>(pts, n, H, live, q, 0, [&]() { return tau; }, visit2, [](int) {});
    }
    const long long t_pass2 = g_knn_dbg_on ? clock64() : 0;
    cnt = min(cnt, kk);       // (cannot exceed it: nless candidates are closer than tau, at most room ties were taken)
    // order: rank = #{dk < d} + the equal ones placed before (entries arrive in ascending index order)
    int most = cnt;
    for (int o = 32; o > 0; o >>= 1) most = max(most, __shfl_xor(most, o, 64));
    if (g_knn_dbg_on && (threadIdx.x & 63) == 0) {
        atomicAdd(&g_knn_dbg[1], (unsigned long long)c_g1); atomicAdd(&g_knn_dbg[2], (unsigned long long)c_ins);
        atomicAdd(&g_knn_dbg[3], (unsigned long long)c_chain);
        atomicAdd(&g_knn_dbg[4], (unsigned long long)c_g2); atomicAdd(&g_knn_dbg[5], (unsigned long long)most);
        atomicAdd(&g_knn_dbg[6], 1ull);
        atomicMax(&g_knn_dbg[7], (unsigned long long)c_g1); atomicMax(&g_knn_dbg[0], (unsigned long long)c_chain);
        atomicAdd(&g_knn_clk[0], (unsigned long long)(t_seed - t_begin)); atomicAdd(&g_knn_clk[1], (unsigned long long)(t_walk - t_seed));
        atomicAdd(&g_knn_clk[2], (unsigned long long)(t_pass2 - t_walk)); atomicAdd(&g_knn_clk[3], (unsigned long long)c_tiles);
    }
    unsigned used = 0;
    int j = cnt > 0 ? list[tid] : 0;
    float4 pj = pts[j];
#pragma unroll 1
    for (int c = 0; c < most; ++c) {
        const bool h = c < cnt;
        const int jn = c + 1 < cnt ? list[(c + 1) * kNNThreads + tid] : 0;
        const float4 pn = pts[jn];          // the next entry's point is on its way while this one is ranked
        const float d = dist2(q.x, q.y, q.z, pj);
        int r = 0;
#pragma unroll
        for (int s = 0; s < KMAX; ++s) r += dk[s] < d ? 1 : 0;
        r = min(r, 31);
        r += __builtin_ctz(~(used >> r));
        if (h && r < kk) {
            used |= 1u << r;
            emit(r, j);
        }
        j = jn; pj = pn;
    }
    return cnt;
}

// G2 / N1 selection: exact kNN (KMAX slots, the first k are used) on the Morton-ordered cloud with tile / mini culling (bound = the lane's
// current KMAX-th distance).  grid = (blocks, clouds); cloud c spans pts[offs[c] .. offs[c+1]).  The neighbours go to knn (layout: knn_at)
// as cloud-local SORTED-space indices, -1 in the slots a cloud with fewer than k points cannot fill; k_cov_from_knn / k_feat_from_knn do
// the fp64 tails (without their state the selection keeps fewer registers alive: more waves per SIMD).
template <int KMAX, int WAVES = (KMAX <= 20 ? 6 : (KMAX <= 30 ? 5 : 4))>
__global__ __launch_bounds__(kNNThreads) __attribute__((amdgpu_waves_per_eu(WAVES, WAVES))) void k_knn_cov(const float4* __restrict__ pts_all,
                                                        const int64_t* __restrict__ offs, const int* __restrict__ tile_base,
                                                        const float4* __restrict__ tlo, const float4* __restrict__ thi,
        const float4* __restrict__ mlo, const float4* __restrict__ mhi, int k, int* __restrict__ knn)
{
    __shared__ int knn_list[KMAX * kNNThreads];
    __shared__ int knn_rec[kNNThreads / 64][kKnnRec];
    const int c = blockIdx.y;
    const int64_t o = offs[c];
    const int n = (int)(offs[c + 1] - o);
    const float4* pts = pts_all + o;
    Hier H;
    H.tlo = tlo + tile_base[c]; H.thi = thi + tile_base[c];
    H.mlo = mlo + (size_t)64 * tile_base[c]; H.mhi = mhi + (size_t)64 * tile_base[c];
    H.ntiles = (n + kTile - 1) / kTile;
    const unsigned wg = blockIdx.y * gridDim.x + blockIdx.x;
    if (g_knn_dbg_on && threadIdx.x == 0 && wg < 65536) g_knn_trace[2 * wg] = wall_clock64();
    for (int base = blockIdx.x * kNNThreads; base < n; base += gridDim.x * kNNThreads) {
        const int i = base + threadIdx.x;
        const bool live = i < n;
        const float4 q = pts[live ? i : 0];
        int* const out = knn + knn_at(o, c, live ? i : 0, k, 0);
        const int found = knn_two_pass<KMAX>(knn_list, pts, n, H, i, live, q, k, g_knn_norec ? nullptr : knn_rec[threadIdx.x >> 6],
                                             [&](int r, int j) { out[r << 6] = j; });
        if (live && found < k)
            for (int s = found; s < k; ++s) out[s << 6] = -1;
    }
    if (g_knn_dbg_on && wg < 65536) {
        __syncthreads();
        if (threadIdx.x == 0) g_knn_trace[2 * wg + 1] = wall_clock64();
    }
}


// ---- RING++ point-feature front-end -----------------------------------
// calculate_features (generate_bev_pointfeat_cython/src/kernel.cu:16-104) for one point, given its
// 5 eigenvalues (3-D descending, 2-D descending) and the z of its k neighbours.
__device__ __forceinline__ void point_features(const float* e, const float* nz, int k, float* f)
{
    const float e0 = e[0], e1 = e[1], e2 = e[2];
    const float sum = e0 + e1 + e2, prod = e0 * e1 * e2, sum2 = e[3] + e[4];
    f[0] = e2 / sum;                                                  // C_
    f[1] = (float)pow((double)(prod / (sum * sum * sum)), 1.0 / 3.0);  // O_
    f[2] = (e0 - e1) / e0;                                            // L_
    float ent = 0.0f;
    ent += (e0 / sum) * logf(e0 / sum);
    ent += (e1 / sum) * logf(e1 / sum);
    ent += (e2 / sum) * logf(e2 / sum);
    f[3] = -ent;                                                      // E_
    f[4] = (e1 - e2) / e0;                                            // P_
    f[5] = e2 / e0;                                                   // S_
    f[6] = (e0 - e2) / e0;                                            // A_
    f[7] = sum;                                                       // X_
    f[8] = (float)((double)(3 * k) / (4.0 * M_PI * (double)prod));    // D_
    f[9] = sum2;                                                      // S_2
    f[10] = e[4] / e[3];                                              // L_2
    float mean = 0.0f, mn = 10000.0f;
    for (int i = 0; i < k; ++i) { mean += nz[i]; mn = fminf(mn, nz[i]); }
    mean /= (float)k;
    float dz = -100000.0f, vz = 0.0f;
    for (int i = 0; i < k; ++i) {
        dz = fmaxf(dz, nz[i] - mn);
        const float d = fabsf(nz[i] - mean);
        vz += d * d;
    }
    f[11] = dz;                                                       // dZ_
    f[12] = vz / (float)k;                                            // vZ_
}

// drop-in kernel of voxelfeat.GPUFeatureExtractor: neighbours and eigenvalues supplied by the caller
__global__ void k_features_from_neighbors(const float* __restrict__ pts /* [n][3] */, int n, int k,
                                          const int* __restrict__ knn, const float* __restrict__ eig,
                                          float* __restrict__ feat /* [n][13] */)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        float nz[32];
        for (int j = 0; j < k; ++j) nz[j] = pts[(size_t)knn[(size_t)i * k + j] * 3 + 2];
        float f[13];
        point_features(eig + (size_t)i * 5, nz, k, f);
        for (int j = 0; j < 13; ++j) feat[(size_t)i * 13 + j] = f[j];
    }
}

__device__ __forceinline__ bool inv3_sym(const double* a, double* r)
{
    // a: full 3x3 symmetric, r: full 3x3
    const double c0 = a[4] * a[8] - a[5] * a[7], c1 = a[5] * a[6] - a[3] * a[8], c2 = a[3] * a[7] - a[4] * a[6];
    const double det = a[0] * c0 + a[1] * c1 + a[2] * c2;
    if (det == 0.0) return false;
    const double id = 1.0 / det;
    r[0] = c0 * id; r[1] = (a[2] * a[7] - a[1] * a[8]) * id; r[2] = (a[1] * a[5] - a[2] * a[4]) * id;
    r[3] = c1 * id; r[4] = (a[0] * a[8] - a[2] * a[6]) * id; r[5] = (a[2] * a[3] - a[0] * a[5]) * id;
    r[6] = c2 * id; r[7] = (a[1] * a[6] - a[0] * a[7]) * id; r[8] = (a[0] * a[4] - a[1] * a[3]) * id;
    return true;
}

// G3a: exact 1-NN of every (float-)transformed source point (tile-culled brute force).
// grid = (blocks, pairs).  Kept free of the fp64 algebra so that it runs at full occupancy.  P source points per
// lane: 2 in a batch, 1 for a single pair so that its ~235 workgroups become ~470 (see launch_nn_scan).
// corr[so + i] = target index (sorted space), or -1 when d^2 >= max_corr^2.
template <int P>
__global__ __launch_bounds__(kNNThreads) void k_nn_scan(
    const float4* __restrict__ src_all, const int64_t* __restrict__ src_offs,
    const float4* __restrict__ tgt_all, const int64_t* __restrict__ tgt_offs,
    const int* __restrict__ tgt_tile_base, const float4* __restrict__ tlo, const float4* __restrict__ thi,
        const float4* __restrict__ mlo, const float4* __restrict__ mhi,
    const LmState* __restrict__ st, GicpParams prm, int* __restrict__ corr, int* __restrict__ nn_seed,
    const int* __restrict__ tgt_bbox, float* __restrict__ lb_out, int gate)
{
    const int pair = blockIdx.y;
    const LmState& S = st[pair];
    if (!S.active || S.phase != 0) return;   // LM trials reuse the cached correspondences (upstream compute_error)
    if (gate && !(pair_motion(S) > prm.motion_switch)) return;   // round-4 schedule: this pair is certified / searched by k_nn_scan_g
    const int64_t so = src_offs[pair], to = tgt_offs[pair];
    const int n = (int)(src_offs[pair + 1] - so), m = (int)(tgt_offs[pair + 1] - to);
    const float4* src = src_all + so;
    const float4* tgt = tgt_all + to;
    Hier H;
    H.tlo = tlo + tgt_tile_base[pair]; H.thi = thi + tgt_tile_base[pair];
    H.mlo = mlo + (size_t)64 * tgt_tile_base[pair]; H.mhi = mhi + (size_t)64 * tgt_tile_base[pair];
    H.ntiles = (m + kTile - 1) / kTile;
    const float maxc2 = prm.max_corr2 < 3.0e38 ? (float)prm.max_corr2 * 1.0001f : INFINITY;
    float Tf[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) Tf[i] = (float)S.x[i];
    float glo[3];
    const float gsc = morton_grid(tgt_bbox, pair, glo);
    const int per_block = kNNThreads * P;
    const unsigned wg = blockIdx.y * gridDim.x + blockIdx.x;
    if (g_knn_dbg_on && threadIdx.x == 0 && wg < 65536) g_nn_trace[2 * wg] = wall_clock64();
    for (int base = blockIdx.x * per_block; base < n; base += gridDim.x * per_block) {
        float qx[P], qy[P], qz[P];
        int si[P], seed[P];
        bool live[P];
#pragma unroll
        for (int p = 0; p < P; ++p) {
            // a wave owns 4 x 64 CONSECUTIVE (Morton-ordered, i.e. spatially compact) source points
            si[p] = base + (threadIdx.x >> 6) * (64 * P) + p * 64 + (threadIdx.x & 63);
            live[p] = si[p] < n;
            const float4 a = src[live[p] ? si[p] : 0];
            qx[p] = Tf[0] * a.x + Tf[1] * a.y + Tf[2] * a.z + Tf[3];
            qy[p] = Tf[4] * a.x + Tf[5] * a.y + Tf[6] * a.z + Tf[7];
            qz[p] = Tf[8] * a.x + Tf[9] * a.y + Tf[10] * a.z + Tf[11];
            seed[p] = live[p] ? nn_seed[so + si[p]] : -1;   // last pass's nearest neighbour (the rejected ones too)
            if (live[p] && seed[p] < 0 && m > 0) seed[p] = morton_seed(tgt, m, qx[p], qy[p], qz[p], glo, gsc);  // cold start
        }
        float best[P];
        int bidx[P];
        nn_scan<P>(tgt, m, H, maxc2, qx, qy, qz, live, best, bidx, seed);
#pragma unroll
        for (int p = 0; p < P; ++p)
            if (live[p]) {
                corr[so + si[p]] = (bidx[p] >= 0 && (double)best[p] < prm.max_corr2) ? bidx[p] : -1;
                nn_seed[so + si[p]] = bidx[p];
                if (lb_out) lb_out[so + si[p]] = 0.0f;     // this search leaves no certificate
            }
    }
    if (g_knn_dbg_on && wg < 65536) {
        __syncthreads();
        if (threadIdx.x == 0) g_nn_trace[2 * wg + 1] = wall_clock64();
    }
}

// G3b + G4: Mahalanobis matrices, residuals and the 28 fp64 sums for the correspondences found by
// k_nn_scan.  grid = (blocks, pairs), one source point per lane; partial[pair][block][28].
// Two poses: the Mahalanobis matrices belong to the linearisation pose S.x (upstream caches them in
// update_correspondences), the residuals to the evaluated pose S.xi.  Phase 0: xi == x, all 28 sums
// (FastGICP::linearize); phase 1: only the error sum (FastGICP::compute_error of an LM trial).
// WAVES: waves per SIMD the register budget is cut for; PREN: the normals of the lane's NEXT point (own 24 B streamed, neighbour's 24 B gathered)
// travel one point ahead like the points themselves, instead of being requested where the algebra needs them
template <int WAVES, bool PREN>
__global__ __launch_bounds__(kNNThreads) __attribute__((amdgpu_waves_per_eu(WAVES, WAVES))) void k_linearize(
    const float4* __restrict__ src_all, const int64_t* __restrict__ src_offs, const double* __restrict__ src_cov,
    const float4* __restrict__ tgt_all, const int64_t* __restrict__ tgt_offs, const double* __restrict__ tgt_cov,
    const LmState* __restrict__ st, const int* __restrict__ corr, double* __restrict__ partial, int max_blocks, int trial_only)
{
    __shared__ double red[kNNThreads / 64][kTerms];
    const int pair = blockIdx.y;
    const LmState& S = st[pair];
    if (!S.active) return;
    if (trial_only && S.phase == 0) return;     // a tick enqueued WITHOUT its search kernels (mrs_gicp_batch_align, one pair): a pair that needs a
                                                // linearisation sits this tick out (k_lm_update leaves its state alone) and takes the next, full one
    const int64_t so = src_offs[pair], to = tgt_offs[pair];
    const int n = (int)(src_offs[pair + 1] - so);
    const float4* src = src_all + so;
    const float4* tgt = tgt_all + to;
    double* pout = partial + ((size_t)pair * max_blocks + blockIdx.x) * kTerms;
    const bool error_only = S.phase == 1;
    double T[12], TL[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) { T[i] = S.xi[i]; TL[i] = S.x[i]; }
    double acc[kTerms];
#pragma unroll
    for (int i = 0; i < kTerms; ++i) acc[i] = 0.0;
    const int per_block = kNNThreads * kPts;  // same point -> block mapping as the scan (fixed summation order)
    for (int base = blockIdx.x * per_block; base < n; base += gridDim.x * per_block) {
        // software pipeline over the lane's kPts points (a rolled loop: one copy of the algebra): the correspondence index travels two
        // points ahead, the point's own data and its gathered neighbour one point ahead of the algebra (the gathers are what the kernel
        // waits for); the order of the sums does not change
        auto idx_of = [&](int p) { const int i = base + p * kNNThreads + (int)threadIdx.x; return (p < kPts && i < n) ? corr[so + i] : -1; };
        int j_cur = idx_of(0), j_nx = idx_of(1);
        float4 a_nx = make_float4(0.f, 0.f, 0.f, 0.f), b_nx = a_nx;
        double na_nx[3] = {0.0, 0.0, 0.0}, nb_nx[3] = {0.0, 0.0, 0.0};
        auto fetch_normals = [&](int i_next, int j_next) {
            const double* pa = src_cov + kCovDoubles * (size_t)(so + i_next);
            const double* pb = tgt_cov + kCovDoubles * (size_t)(to + j_next);
            na_nx[0] = pa[0]; na_nx[1] = pa[1]; na_nx[2] = pa[2];
            nb_nx[0] = pb[0]; nb_nx[1] = pb[1]; nb_nx[2] = pb[2];
        };
        if (j_cur >= 0) {
            a_nx = src[base + threadIdx.x]; b_nx = tgt[j_cur];
            if (PREN) fetch_normals(base + (int)threadIdx.x, j_cur);
        }
#pragma unroll 1
        for (int p = 0; p < kPts; ++p) {
            const int i = base + p * kNNThreads + threadIdx.x;
            const int j = j_cur;
            const float4 a = a_nx, bb = b_nx;
            const double na[3] = {na_nx[0], na_nx[1], na_nx[2]}, nbv[3] = {nb_nx[0], nb_nx[1], nb_nx[2]};
            j_cur = j_nx;
            j_nx = idx_of(p + 2);
            if (j_cur >= 0) {
                a_nx = src[i + kNNThreads]; b_nx = tgt[j_cur];
                if (PREN) fetch_normals(i + kNNThreads, j_cur);
            }
            if (j < 0) continue;
            double ca[6], cb[6];
            cov6_from_normal(PREN ? na : src_cov + kCovDoubles * (size_t)(so + i), ca);
            cov6_from_normal(PREN ? nbv : tgt_cov + kCovDoubles * (size_t)(to + j), cb);
            const double CA[9] = {ca[0], ca[1], ca[2], ca[1], ca[3], ca[4], ca[2], ca[4], ca[5]};
            double RC[9], RCR[9], M[9];
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int c = 0; c < 3; ++c)
                    RC[3 * r + c] = TL[4 * r] * CA[c] + TL[4 * r + 1] * CA[3 + c] + TL[4 * r + 2] * CA[6 + c];
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int c = 0; c < 3; ++c)
                    RCR[3 * r + c] = RC[3 * r] * TL[4 * c] + RC[3 * r + 1] * TL[4 * c + 1] + RC[3 * r + 2] * TL[4 * c + 2];
            RCR[0] += cb[0]; RCR[1] += cb[1]; RCR[2] += cb[2];
            RCR[3] += cb[1]; RCR[4] += cb[3]; RCR[5] += cb[4];
            RCR[6] += cb[2]; RCR[7] += cb[4]; RCR[8] += cb[5];
            if (!inv3_sym(RCR, M)) continue;
            double ta[3], e[3], Me[3];
#pragma unroll
            for (int r = 0; r < 3; ++r)
                ta[r] = T[4 * r] * (double)a.x + T[4 * r + 1] * (double)a.y + T[4 * r + 2] * (double)a.z + T[4 * r + 3];
            e[0] = (double)bb.x - ta[0]; e[1] = (double)bb.y - ta[1]; e[2] = (double)bb.z - ta[2];
#pragma unroll
            for (int r = 0; r < 3; ++r) Me[r] = M[3 * r] * e[0] + M[3 * r + 1] * e[1] + M[3 * r + 2] * e[2];
            acc[27] += e[0] * Me[0] + e[1] * Me[1] + e[2] * Me[2];
            if (error_only) continue;
            const double J[18] = {0, -ta[2], ta[1], -1, 0, 0,
                                  ta[2], 0, -ta[0], 0, -1, 0,
                                  -ta[1], ta[0], 0, 0, 0, -1};
            double MJ[18];
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int c = 0; c < 6; ++c)
                    MJ[6 * r + c] = M[3 * r] * J[c] + M[3 * r + 1] * J[6 + c] + M[3 * r + 2] * J[12 + c];
#pragma unroll
            for (int r = 0; r < 6; ++r) {
#pragma unroll
                for (int c = r; c < 6; ++c)
                    acc[r * 6 - (r * (r - 1)) / 2 + (c - r)] += J[r] * MJ[c] + J[6 + r] * MJ[6 + c] + J[12 + r] * MJ[12 + c];
            }
#pragma unroll
            for (int r = 0; r < 6; ++r) acc[21 + r] += J[r] * Me[0] + J[6 + r] * Me[1] + J[12 + r] * Me[2];
        }
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (int i = 0; i < kTerms; ++i) {
        const double v = wave_sum_d(acc[i]);
        if (lane == 0) red[wave][i] = v;
    }
    __syncthreads();
    if (threadIdx.x < kTerms) {
        double v = 0;
        for (int w = 0; w < kNNThreads / 64; ++w) v += red[w][threadIdx.x];
        pout[threadIdx.x] = v;
    }
}


// ================================================================================================================================
// Round 4 search core (nn_core.hpp): octree-cell leaves + query groups.  The kernels below replace k_nn_scan / k_knn_cov;
// the round-3 kernels stay selectable (mrs_gicp_batch_set_search(h, 0)) for A/B runs and as a cross-check in the tests.
struct HierArrays {
    const float4* llo; const float4* lhi; const float4* tlo; const float4* thi; const float4* slo; const float4* shi;
    const int* leaf_first; const int* tile_first; const int* super_first;   // [clouds + 1]
};

__device__ __forceinline__ nnc::LeafHier cloud_hier(const HierArrays& A, int c)
{
    nnc::LeafHier H;
    const int l0 = A.leaf_first[c], t0 = A.tile_first[c], s0 = A.super_first[c];
    H.llo = A.llo + l0; H.lhi = A.lhi + l0; H.nleaf = A.leaf_first[c + 1] - l0;
    H.tlo = A.tlo + t0; H.thi = A.thi + t0; H.ntile = A.tile_first[c + 1] - t0;
    H.slo = A.slo + s0; H.shi = A.shi + s0; H.nsuper = A.super_first[c + 1] - s0;
    return H;
}

// G3a, round 4: exact 1-NN of every (float-)transformed source point.  One query per lane; semantics of k_nn_scan (corr = target index in
// sorted space or -1 when d^2 >= max_corr^2; nn_seed = the neighbour found, warm start of the next pass), ties to the smaller index.
// Certificates (round 4).  After a search the lane knows lb = a lower bound of the distance from its query to every target point OTHER
// than the neighbour found: the second smallest distance it evaluated, or the radius it searched (neighbour distance + a margin), whichever
// is smaller.  When the pose changes, a query moves by delta = |T_new a - T_prev a|, so every other point is still at least lb - delta
// away; if the old neighbour's new distance is below that, it is still THE nearest neighbour -- exactly, by the triangle inequality --
// and no search is needed (k_nn_certify).  Queries that cannot be certified go to a per-pair work list and are searched as before.
// Late iterations of an alignment move the cloud by less than the gap between a point's nearest and second nearest neighbour, so most of
// their passes reduce to one streaming kernel.  Float evaluation error is covered by a relative 1e-5 + absolute (1e-6 m + 4 ulps of the
// largest coordinate) slack on both sides -- certificates are exact up to that evaluation error; an exact tie (two points at one distance) leaves no gap and is always searched, so ties still resolve to the smaller index.
struct CertArrays {
    float* lb;             // [source points] lower bound described above (0: none)
    float* t_prev;         // [pairs][12] pose of the pair's last nearest-neighbour pass (float, like the searches use it)
    int* work;             // [source points] per pair (at the pair's source offset): source indices that need a search
    int* bcount;           // [pairs][nb] entries in the work list of each block of 1024 consecutive source points (its list starts at the block)
    int nb;                // blocks of the longest source cloud
    unsigned long long* searched;   // [pairs][kStatStride] statistics of an align(), slots 0 / 1 of every pair: queries searched, queries due (points of the
                                    // pairs that searched, per pass).  One 128-byte line per pair: 30 000 workgroups adding to ONE word serialise in its L2
                                    // channel (~12 ns each: 0.7 ms of a 0.77 ms k_nn_certify); the host sums the pairs
};

__device__ __forceinline__ void pose_f(const LmState& S, float (&Tf)[12])
{
#pragma unroll
    for (int i = 0; i < 12; ++i) Tf[i] = (float)S.x[i];
}

constexpr int kStatStride = 16;    // unsigned long longs per pair in CertArrays::searched (128 bytes)
constexpr int kCertBlock = 1024;   // consecutive source points whose uncertified members form one work list (searched by one workgroup)

// One workgroup per 1024 consecutive source points of every pair that is about to search: certify the old neighbour or put the point on
// the block's work list, in index order (lists of consecutive points keep the search's waves spatially compact; appended with atomics in
// completion order, the waves of a sparse list spanned the whole cloud and tested ~1000 tiles each).
__global__ __launch_bounds__(256) void k_nn_certify(const float4* __restrict__ src_all, const int64_t* __restrict__ src_offs,
                                                   const float4* __restrict__ tgt_all, const int64_t* __restrict__ tgt_offs,
                                                   const LmState* __restrict__ st, GicpParams prm, int* __restrict__ corr,
                                                   const int* __restrict__ nn_seed, CertArrays C)
{
    __shared__ int wcnt[16];
    const int pair = blockIdx.y;
    const LmState& S = st[pair];
    if (!S.active || S.phase != 0) return;
    if (pair_motion(S) > prm.motion_switch) return;      // a pair that moved this far goes to the round-3 kernel (nn_pass)
    const int64_t so = src_offs[pair], to = tgt_offs[pair];
    const int n = (int)(src_offs[pair + 1] - so), m = (int)(tgt_offs[pair + 1] - to);
    const int b0 = (int)blockIdx.x * kCertBlock;
    if (b0 >= n) return;
    const float4* src = src_all + so;
    const float4* tgt = tgt_all + to;
    float Tf[12], Tp[12];
    pose_f(S, Tf);
#pragma unroll
    for (int i = 0; i < 12; ++i) Tp[i] = C.t_prev[12 * pair + i];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int rank[4];
    bool need[4];
    // the four points of a lane: every load of a stage is requested before the first one is used (the gathers are what the kernel waits for)
    float4 a[4], nb[4];
    int seed[4];
    float lb[4];
    bool live[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int i = b0 + j * 256 + (int)threadIdx.x;
        live[j] = i < n;
        a[j] = src[live[j] ? i : b0];
        seed[j] = live[j] ? nn_seed[so + i] : -1;
        lb[j] = live[j] ? C.lb[so + i] : 0.0f;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) nb[j] = tgt[(seed[j] >= 0 && seed[j] < m) ? seed[j] : 0];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int i = b0 + j * 256 + (int)threadIdx.x;
        bool certified = false;
        if (live[j] && seed[j] >= 0 && seed[j] < m && lb[j] > 0.0f) {
            const float qx = Tf[0] * a[j].x + Tf[1] * a[j].y + Tf[2] * a[j].z + Tf[3];
            const float qy = Tf[4] * a[j].x + Tf[5] * a[j].y + Tf[6] * a[j].z + Tf[7];
            const float qz = Tf[8] * a[j].x + Tf[9] * a[j].y + Tf[10] * a[j].z + Tf[11];
            const float dx = qx - (Tp[0] * a[j].x + Tp[1] * a[j].y + Tp[2] * a[j].z + Tp[3]);
            const float dy = qy - (Tp[4] * a[j].x + Tp[5] * a[j].y + Tp[6] * a[j].z + Tp[7]);
            const float dz = qz - (Tp[8] * a[j].x + Tp[9] * a[j].y + Tp[10] * a[j].z + Tp[11]);
            const float delta = sqrtf(dx * dx + dy * dy + dz * dz);
            // slack for the float evaluation of T a on both sides of the comparison: 1e-5 relative + 1e-6 m + 4 ulps of the largest
            // coordinate (an ulp at lidar range is 4e-6 m at 60 m: an absolute micrometre alone would rest on this kernel and the search
            // rounding T a identically).  A certificate is exact up to that evaluation error; what fails the test is searched.
            const float slack = 1e-6f + 4.0f * FLT_EPSILON * fmaxf(fmaxf(fabsf(qx), fabsf(qy)), fabsf(qz));
            const float lbn = lb[j] - delta * 1.00001f - slack;
            const float d1sq = dist2(qx, qy, qz, nb[j]);
            if (sqrtf(d1sq) * 1.00001f + slack < lbn) {       // (false for NaN)
                certified = true;
                corr[so + i] = (double)d1sq < prm.max_corr2 ? seed[j] : -1;
                C.lb[so + i] = lbn;
            }
        }
        need[j] = live[j] && !certified;
        const unsigned long long mk = __ballot(need[j]);
        rank[j] = (int)__popcll(mk & ((1ull << lane) - 1ull));
        if (lane == 0) wcnt[j * 4 + wave] = (int)__popcll(mk);
    }
    __syncthreads();
    int total = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        int off = 0;
        for (int u = 0; u < j * 4 + wave; ++u) off += wcnt[u];
        if (need[j]) C.work[so + b0 + off + rank[j]] = b0 + j * 256 + (int)threadIdx.x;
    }
    if (threadIdx.x == 0) {
        for (int u = 0; u < 16; ++u) total += wcnt[u];
        C.bcount[(size_t)pair * C.nb + blockIdx.x] = total;
        const int members = min(kCertBlock, n - b0);
        atomicAdd(&C.searched[(size_t)pair * kStatStride], (unsigned long long)(2 * total > members ? members : total));
        atomicAdd(&C.searched[(size_t)pair * kStatStride + 1], (unsigned long long)members);
    }
}

// the pose of this pass becomes t_prev of every pair that searched; the work-list sizes go to the statistics
__global__ void k_nn_store_pose(const LmState* __restrict__ st, int n_pairs, CertArrays C, int worklists, const int64_t* __restrict__ src_offs,
                                float motion_switch)
{
    const int pair = blockIdx.x * blockDim.x + threadIdx.x;
    if (pair >= n_pairs) return;
    const LmState& S = st[pair];
    if (!S.active || S.phase != 0) return;
    float Tf[12];
    pose_f(S, Tf);
#pragma unroll
    for (int i = 0; i < 12; ++i) C.t_prev[12 * pair + i] = Tf[i];
    const int n_src = (int)(src_offs[pair + 1] - src_offs[pair]);
    if (worklists && !(pair_motion(S) > motion_switch)) return;        // counted block by block in k_nn_certify
    C.searched[(size_t)pair * kStatStride] += (unsigned long long)n_src;      // this pair's own line, one thread per pair, stream-ordered
    C.searched[(size_t)pair * kStatStride + 1] += (unsigned long long)n_src;
}

// G3a, round 4: exact 1-NN of every (float-)transformed source point.  One query per lane; semantics of k_nn_scan (corr = target index in
// sorted space or -1 when d^2 >= max_corr^2; nn_seed = the neighbour found, warm start of the next pass), ties to the smaller index.
// WORK: the queries are the entries of the pair's work list (k_nn_certify) instead of all source points.  Always leaves the certificate
// bound of every query it searched in C.lb.
template <bool PROF, bool WORK>
__global__ __launch_bounds__(kNNThreads) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_nn_scan_g(
    const float4* __restrict__ src_all, const int64_t* __restrict__ src_offs,
    const float4* __restrict__ tgt_all, const int64_t* __restrict__ tgt_offs, HierArrays HA,
    const LmState* __restrict__ st, GicpParams prm, int* __restrict__ corr, int* __restrict__ nn_seed,
    const int* __restrict__ tgt_bbox, CertArrays C)
{
    __shared__ nnc::GrpLds lds[kNNThreads / 64];
    const int pair = blockIdx.y;
    const LmState& S = st[pair];
    if (!S.active || S.phase != 0) return;
    if (WORK && pair_motion(S) > prm.motion_switch) return;
    const int64_t so = src_offs[pair], to = tgt_offs[pair];
    const int n_src = (int)(src_offs[pair + 1] - so), m = (int)(tgt_offs[pair + 1] - to);
    // one workgroup per block of 1024 consecutive source points: its work list, or (no lists, or more than half of the block listed) all of it
    const int b0 = (int)blockIdx.x * kCertBlock;
    if (b0 >= n_src) return;
    const int members = min(kCertBlock, n_src - b0);
    const int listed_n = WORK ? C.bcount[(size_t)pair * C.nb + blockIdx.x] : members;
    const bool listed = WORK && 2 * listed_n <= members;
    const int n = listed ? listed_n : members;
    const float4* src = src_all + so;
    const float4* tgt = tgt_all + to;
    const nnc::LeafHier H = cloud_hier(HA, pair);
    const float maxc2 = prm.max_corr2 < 3.0e38 ? (float)prm.max_corr2 * 1.0001f : INFINITY;
    float Tf[12];
    pose_f(S, Tf);
    float glo[3];
    const float gsc = morton_grid(tgt_bbox, pair, glo);
    nnc::GrpLds& L = lds[threadIdx.x >> 6];
    const unsigned wg = blockIdx.y * gridDim.x + blockIdx.x;
    if (g_knn_dbg_on && threadIdx.x == 0 && wg < 65536) g_nn_trace[2 * wg] = wall_clock64();
    for (int base = 0; base < n; base += kNNThreads) {
        const unsigned long long t_wave = PROF ? __builtin_readcyclecounter() : 0ull;
        const int w = base + (int)threadIdx.x;
        const bool live = w < n;
        const int i = listed ? C.work[so + b0 + (live ? w : 0)] : b0 + (live ? w : 0);
        const float4 a = src[i];
        const float qx = Tf[0] * a.x + Tf[1] * a.y + Tf[2] * a.z + Tf[3];
        const float qy = Tf[4] * a.x + Tf[5] * a.y + Tf[6] * a.z + Tf[7];
        const float qz = Tf[8] * a.x + Tf[9] * a.y + Tf[10] * a.z + Tf[11];
        // margin of the search radius beyond the neighbour's distance: what a later pass may certify against.  The volume searched grows
        // with the cube of the radius, so it stays small: it pays in the late passes of an alignment, where a point moves by well under a
        // millimetre per pass (a 6 cm margin made the searches of the early passes 70 times as long and certified nothing there)
        const float margin = prm.cert_margin;
        int seed = live ? nn_seed[so + i] : -1;
        if (live && seed < 0 && m > 0) seed = morton_seed(tgt, m, qx, qy, qz, glo, gsc);   // cold start
        // any target point is an upper bound: last pass's neighbour is nearly always the winner again.  Its distance is formed inside the
        // search's first radius evaluation, after the search has requested the top of the hierarchy: the two loads overlap
        const bool has_seed = live && seed >= 0 && seed < m;
        const float4 sp = tgt[has_seed ? seed : 0];
        float best = INFINITY, second = INFINITY;
        int bidx = -1;
        nnc::Prof prof;
        const unsigned long long t_in = PROF ? __builtin_readcyclecounter() : 0ull;
        auto radius2 = [&]() {       // squared search radius: (distance of the best candidate so far + margin)^2, capped by the threshold
            const float r = sqrtf(fminf(has_seed ? dist2(qx, qy, qz, sp) : INFINITY, best)) + margin;
            return fminf(maxc2, r * r);
        };
        nnc::grp_search<PROF>(tgt, H, L, qx, qy, qz, live, radius2,
                              [&](int j, float d, bool ok) {
                                  if (!ok) return;
                                  second = __builtin_amdgcn_fmed3f(best, d, second);      // second smallest of everything evaluated
                                  if (d < best) { best = d; bidx = j; }
                              }, &prof);
        if (PROF) {
            const unsigned long long t_out = __builtin_readcyclecounter();
            if ((threadIdx.x & 63) == 0) {
                atomicAdd(&nnc::g_prof[0], 1ull); atomicAdd(&nnc::g_prof[1], t_in - t_wave); atomicAdd(&nnc::g_prof[2], prof.cyc_top);
                atomicAdd(&nnc::g_prof[3], prof.cyc_leaf); atomicAdd(&nnc::g_prof[4], prof.cyc_drain); atomicAdd(&nnc::g_prof[5], prof.tiles_near);
                atomicAdd(&nnc::g_prof[6], prof.tiles_needed); atomicAdd(&nnc::g_prof[7], prof.grp_tiles); atomicAdd(&nnc::g_prof[8], prof.query_tests);
                atomicAdd(&nnc::g_prof[9], prof.queued); atomicAdd(&nnc::g_prof[10], prof.batches); atomicAdd(&nnc::g_prof[12], prof.drains);
                atomicAdd(&nnc::g_prof[13], t_out - t_wave); atomicMax(&nnc::g_prof[14], t_out - t_wave);
            }
            if ((threadIdx.x & 63) == 63) atomicAdd(&nnc::g_prof[11], prof.staged);
        }
        if (live) {
            corr[so + i] = (bidx >= 0 && (double)best < prm.max_corr2) ? bidx : -1;
            nn_seed[so + i] = bidx;
            // every point that was not evaluated lies beyond the final radius (radii only shrink while the search runs)
            C.lb[so + i] = bidx >= 0 ? fminf(sqrtf(second), sqrtf(radius2()) * 0.9999f) : 0.0f;
        }
    }
    if (g_knn_dbg_on && wg < 65536) {
        __syncthreads();
        if (threadIdx.x == 0) g_nn_trace[2 * wg + 1] = wall_clock64();
    }
}

template <int KMAX>
__device__ __forceinline__ float kth_of(const float (&dk)[KMAX], int k)
{
    float v = dk[KMAX - 1];
#pragma unroll
    for (int s = 0; s < KMAX - 1; ++s) v = (s == k - 1) ? dk[s] : v;
    return v;
}

// G2 / N1, round 4: exact k nearest neighbours (the point itself included) of every point of every cloud, as cloud-local sorted-space
// indices in (distance, index) order, layout knn_at(); -1 in the slots a cloud with fewer than k points cannot fill.
// Seed: the group's 32 (64 for k > 16) neighbours along the Morton curve give a bound close to the final one; pass 1: the KMAX smallest
// DISTANCES (v_med3 chain, no indices) over the leaves within the shrinking bound -> tau = the exact k-th distance; pass 2: every candidate
// within tau, strictly closer ones from the bottom of a k-slot LDS list, exact ties from its top (they arrive in ascending index order, and a
// tie is only kept while the list still has room for it: at most k - #closer can be needed), so a cluster of duplicates can neither
// overflow the list nor push a closer point out; selection: (distance, index) insertion of the <= k collected.
template <int KMAX>
__global__ __launch_bounds__(kNNThreads) void k_knn_select(const float4* __restrict__ pts_all, const int64_t* __restrict__ offs, HierArrays HA,
                                                           int k, int* __restrict__ knn)
{
    __shared__ nnc::GrpLds lds[kNNThreads / 64];
    __shared__ int lst[KMAX * kNNThreads];          // slot-major: slot s of lane t at lst[s * kNNThreads + t]
    const int c = blockIdx.y;
    const int64_t o = offs[c];
    const int n = (int)(offs[c + 1] - o);
    const float4* pts = pts_all + o;
    const nnc::LeafHier H = cloud_hier(HA, c);
    nnc::GrpLds& L = lds[threadIdx.x >> 6];
    constexpr int HS = KMAX <= 16 ? 32 : 64;
    constexpr int GS = 8;
    const int tid = (int)threadIdx.x;
    for (int base = blockIdx.x * kNNThreads; base < n; base += gridDim.x * kNNThreads) {
        const int i = base + tid;
        const bool live = i < n;
        const float4 q = pts[live ? i : 0];
        float dk[KMAX];
#pragma unroll
        for (int s = 0; s < KMAX; ++s) dk[s] = INFINITY;
        // seed phase: HS consecutive points around the group's queries
        const int hc = min(HS, n);
        const int h0 = max(0, min(base + (tid & ~(GS - 1)) + GS / 2 - HS / 2, n - hc));
        nnc::grp_eval_range(pts, L, q.x, q.y, q.z, h0, hc, [&](int, float d, bool ok) { dist_insert<KMAX>(dk, (ok && live && d == d) ? d : INFINITY); });
        // pass 1: the k-th distance
        nnc::grp_search(pts, H, L, q.x, q.y, q.z, live, [&]() { return kth_of<KMAX>(dk, k); },
                            [&](int j, float d, bool ok) {
                                const bool use = ok && live && (unsigned)(j - h0) >= (unsigned)hc && d == d;
                                dist_insert<KMAX>(dk, use ? d : INFINITY);
                            });
        const float tau = live ? kth_of<KMAX>(dk, k) : -1.0f;
        // pass 2: indices within tau
        int nlt = 0, ntie = 0;
        nnc::grp_search(pts, H, L, q.x, q.y, q.z, live, [&]() { return tau; },
                            [&](int j, float d, bool ok) {
                                if (!(ok && live)) return;
                                if (d < tau) {
                                    lst[min(nlt, KMAX - 1) * kNNThreads + tid] = j;     // at most k - 1 of these; a tie in the way was not needed
                                    ++nlt;
                                    ntie = min(ntie, k - nlt);
                                } else if (d == tau && d < INFINITY && nlt + ntie < k) {
                                    lst[(k - 1 - ntie) * kNNThreads + tid] = j;
                                    ++ntie;
                                }
                            });
        // selection in (distance, index) order
        int ik[KMAX];
#pragma unroll
        for (int s = 0; s < KMAX; ++s) { dk[s] = INFINITY; ik[s] = -1; }
        int most = max(nlt, ntie);
        for (int w = 32; w > 0; w >>= 1) most = max(most, __shfl_xor(most, w, 64));
        for (int e = 0; e < most; ++e) {
            const bool ha = e < nlt, hb = e < ntie;
            const int ja = ha ? lst[e * kNNThreads + tid] : 0;
            const int jb = hb ? lst[(k - 1 - e) * kNNThreads + tid] : 0;
            const float da = dist2(q.x, q.y, q.z, pts[ja]), db = dist2(q.x, q.y, q.z, pts[jb]);
            if (ha) knn_insert_tie<KMAX>(dk, ik, da, ja);
            if (hb) knn_insert_tie<KMAX>(dk, ik, db, jb);
        }
        if (live) {
            int* out = knn + knn_at(o, c, i, k, 0);
#pragma unroll
            for (int s = 0; s < KMAX; ++s)
                if (s < k) out[s << 6] = ik[s];
        }
    }
}

// Second moments of a point's k neighbours in ONE pass over them: sums of d and d d^T with d = p - q taken about the query point q
// (fp64; |d| is a neighbourhood radius, so the subtraction  sum d d^T - n m m^T  cancels a few digits of 16 at most), instead of a pass
// for the mean and a second one about it: half the gathers (30 random 16-byte reads per point instead of 60).
// Memory schedule (round 6): rounds 1-5 walked the slots one by one behind two data-dependent branches each, so that a wave sat out an index
// load AND a dependent gather per neighbour, ~30 round trips in a row (10 k cycles per wave for ~4 k cycles of arithmetic).  Now the slots
// are taken in chunks of kMomChunk: the indices of a chunk (one coalesced 256-byte row per slot and wave; slots past k re-read slot k - 1) are
// requested two chunks ahead of the sums and its points one chunk ahead, so that loads of 2 x kMomChunk neighbours are in flight while a chunk is summed.  An empty slot (-1: a cloud with fewer than k points)
// gathers the query itself and contributes exact zeros, so the sums are those of the slot-by-slot loop bit for bit.
// cv: full symmetric 3x3 of  sum (p - mean)(p - mean)^T  (not yet divided); returns the number of neighbours.
constexpr int kMomChunk = 5;      // 15 / 20 / 30 neighbours (GICP default, the oracle's k, RING++) are whole chunks
template <int KMAX, bool WANT_Z>
__device__ __forceinline__ int neighbour_moments(const float4* __restrict__ pts, const int* __restrict__ nb /* slot s at nb[64 s] */, int k, int self,
                                                 const float4& q, double (&cv)[9], float (&nz)[KMAX])
{
#pragma clang fp contract(fast)
    constexpr int NCH = (KMAX + kMomChunk - 1) / kMomChunk;
    int idx[3][kMomChunk];                                 // chunk c lives in [c % 3]: indices run two chunks ahead of the sums, points one
    float px[2][kMomChunk], py[2][kMomChunk], pz[2][kMomChunk];
    auto indices = [&](int c) {
#pragma unroll
        for (int j = 0; j < kMomChunk; ++j) idx[c % 3][j] = nb[min(c * kMomChunk + j, k - 1) << 6];
    };
    auto gather = [&](int c) {
#pragma unroll
        for (int j = 0; j < kMomChunk; ++j) {
            const int id = idx[c % 3][j];
            const float4 p = pts[id >= 0 ? id : self];
            px[c & 1][j] = p.x; py[c & 1][j] = p.y; pz[c & 1][j] = p.z;
        }
    };
    double sd[3] = {0, 0, 0}, sc[6] = {0, 0, 0, 0, 0, 0};
    int cnt = 0;
    indices(0);
    if (NCH > 1) indices(1);
    gather(0);
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        if (c * kMomChunk < k) {                                       // uniform: k is a kernel argument
            if (c + 2 < NCH && (c + 2) * kMomChunk < k) indices(c + 2);
            if (c + 1 < NCH && (c + 1) * kMomChunk < k) gather(c + 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < kMomChunk; ++j) {
                const int s = c * kMomChunk + j;
                const bool ok = s < k && idx[c % 3][j] >= 0;
                const float fx = px[c & 1][j], fy = py[c & 1][j], fz = pz[c & 1][j];
                const double dx = ok ? (double)fx - (double)q.x : 0.0, dy = ok ? (double)fy - (double)q.y : 0.0, dz = ok ? (double)fz - (double)q.z : 0.0;
                sd[0] += dx; sd[1] += dy; sd[2] += dz;
                sc[0] += dx * dx; sc[1] += dx * dy; sc[2] += dx * dz;
                sc[3] += dy * dy; sc[4] += dy * dz; sc[5] += dz * dz;
                if (WANT_Z && s < KMAX) nz[s] = ok ? fz : 0.0f;
                cnt += ok ? 1 : 0;
            }
        } else if (WANT_Z) {
#pragma unroll
            for (int j = 0; j < kMomChunk; ++j)
                if (c * kMomChunk + j < KMAX) nz[c * kMomChunk + j] = 0.0f;
        }
    }
    const double inv = 1.0 / (double)cnt;
    cv[0] = sc[0] - sd[0] * sd[0] * inv; cv[1] = sc[1] - sd[0] * sd[1] * inv; cv[2] = sc[2] - sd[0] * sd[2] * inv;
    cv[4] = sc[3] - sd[1] * sd[1] * inv; cv[5] = sc[4] - sd[1] * sd[2] * inv; cv[8] = sc[5] - sd[2] * sd[2] * inv;
    cv[3] = cv[1]; cv[6] = cv[2]; cv[7] = cv[5];
    return cnt;
}

// G2 tail: covariance of the k neighbours (fp64) + PLANE regularisation.  One point per lane; knn = the selection's output (knn_at).
// knn_out optional, ORIGINAL indexing.
__global__ __launch_bounds__(256) void k_cov_from_knn(const float4* __restrict__ pts_all, const int64_t* __restrict__ offs, int k,
                                                     const int* __restrict__ knn, double* __restrict__ cov_all, int* __restrict__ knn_out)
{
    const int c = blockIdx.y;
    const int64_t o = offs[c];
    const int n = (int)(offs[c + 1] - o);
    const float4* pts = pts_all + o;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int* nb = knn + knn_at(o, c, i, k, 0);       // slot s at nb[64 s]: one row per wave and slot
        const float4 q = pts[i];
        double cv[9];
        float unused[32];
        const int cnt = neighbour_moments<32, false>(pts, nb, k, i, q, cv, unused);
        for (int a = 0; a < 9; ++a) cv[a] /= cnt;
        double nrm[3];
        mrs::smallest_eigvec(cv, nrm);
        double* out = cov_all + kCovDoubles * (size_t)(o + i);      // the unit normal: C = I - 0.999 n n^T is rebuilt by the readers (cov6_from_normal)
        out[0] = nrm[0]; out[1] = nrm[1]; out[2] = nrm[2];
        if (knn_out) {
            const int oi = __float_as_int(q.w);
            for (int s = 0; s < k; ++s) knn_out[(size_t)(o + oi) * k + s] = nb[s << 6] >= 0 ? __float_as_int(pts[nb[s << 6]].w) : -1;
        }
    }
}

// N1 tail: covariance P^T P / (k - 1) (util.py:123-131) -> eigenvalues of the 3x3 and of its xy 2x2 block, both descending (util.py:134-158) ->
// the 13 hand-crafted features of every point from its k neighbours.  Outputs in the caller's ORIGINAL point order; feat_planes (optional)
// receives the channel-major [9][n] planes x,y,z,C,O,E,L2,dZ,vZ that generate_RINGplusplus feeds to the feature BEV (util.py:220-228).
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4))) void k_feat_from_knn(const float4* __restrict__ pts_all, const int64_t* __restrict__ offs, int k,
                                                      const int* __restrict__ knn, int* __restrict__ knn_out, float* __restrict__ eig_out,
                                                      float* __restrict__ feat_out, float* __restrict__ feat_planes)
{
    const int c = blockIdx.y;
    const int64_t o = offs[c];
    const int n = (int)(offs[c + 1] - o);
    const float4* pts = pts_all + o;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int* nb = knn + knn_at(o, c, i, k, 0);       // slot s at nb[64 s]
        const float4 q = pts[i];
        const int oi = __float_as_int(q.w);
        double cv[9];
        float nz[32];
        const int cnt = neighbour_moments<32, true>(pts, nb, k, i, q, cv, nz);
        for (int a = 0; a < 9; ++a) cv[a] /= (double)(cnt - 1);
        double w[3];
        mrs::sym3_eigvals(cv, w);
        const double hm = 0.5 * (cv[0] + cv[4]), hd = 0.5 * (cv[0] - cv[4]);
        const double rad = sqrt(hd * hd + cv[1] * cv[1]);
        float e[5] = {(float)w[0], (float)w[1], (float)w[2], (float)(hm + rad), (float)(hm - rad)};
        float f[13];
        point_features(e, nz, k, f);
        const size_t gi = (size_t)(o + oi);
        if (knn_out)
            for (int s = 0; s < k; ++s) knn_out[gi * k + s] = nb[s << 6] >= 0 ? __float_as_int(pts[nb[s << 6]].w) : -1;
        if (eig_out) for (int j = 0; j < 5; ++j) eig_out[gi * 5 + j] = e[j];
        if (feat_out) for (int j = 0; j < 13; ++j) feat_out[gi * 13 + j] = f[j];
        if (feat_planes) {
            float* pl = feat_planes + (size_t)9 * o;  // scan-local channel-major planes
            pl[0 * (size_t)n + oi] = q.x; pl[1 * (size_t)n + oi] = q.y; pl[2 * (size_t)n + oi] = q.z;
            pl[3 * (size_t)n + oi] = f[0]; pl[4 * (size_t)n + oi] = f[1]; pl[5 * (size_t)n + oi] = f[3];
            pl[6 * (size_t)n + oi] = f[10]; pl[7 * (size_t)n + oi] = f[11]; pl[8 * (size_t)n + oi] = f[12];
        }
    }
}

// ---- G7: voxelised GICP (fast_gicp FastVGICP / FastVGICPCuda; Koide et al., ICRA 2021) ---------------------
// The target is summarised per voxel of edge `res`: mean of its points and mean of their (regularised)
// covariances (ADDITIVE accumulation).  A transformed source point corresponds to the voxel that contains it
// (DIRECT1) and optionally its 6 / 26 neighbours; each correspondence is a distribution-to-distribution term
// weighted by sqrt(points in the voxel).  max_correspondence_distance is not used.  Parity unpinned: restated
// from the publication and SURVEY.md row G7 (the submodule is absent).  Conventions of upstream's CUDA voxel map:
// voxel coordinate = floor(x / resolution - 0.5) in float arithmetic on the float-transformed point
// (calc_voxel_coord); correspondences and (C_voxel + R C_A R^T)^-1 belong to the linearisation pose, LM trials
// (compute_error) only re-evaluate the residuals (x_linearized / x_eval in upstream's kernels).
__device__ __forceinline__ int voxel_coord_f(float v, float res) { return (int)floorf(v / res - 0.5f); }

__device__ __forceinline__ unsigned long long voxel_key(int cloud, int ix, int iy, int iz)
{
    return ((unsigned long long)cloud << 48) | ((unsigned long long)(ix & 0xffff) << 32) |
           ((unsigned long long)(iy & 0xffff) << 16) | (unsigned long long)(iz & 0xffff);
}

__global__ void k_vox_keys(const float4* __restrict__ pts, const int64_t* __restrict__ offs, double res,
                           unsigned long long* __restrict__ keys, int* __restrict__ vals)
{
    const int c = blockIdx.y;
    const int64_t o = offs[c];
    const int n = (int)(offs[c + 1] - o);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const float4 p = pts[o + i];
        const float resf = (float)res;
        keys[o + i] = voxel_key(c, voxel_coord_f(p.x, resf) + 32768, voxel_coord_f(p.y, resf) + 32768,
                                voxel_coord_f(p.z, resf) + 32768);
        vals[o + i] = (int)(o + i);
    }
}

__global__ void k_vox_heads(const unsigned long long* __restrict__ keys, size_t n, int* __restrict__ head)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        head[i] = (i == 0 || keys[i] != keys[i - 1]) ? 1 : 0;
}

// one lane per voxel head: key, mean (w = count) and mean covariance of the voxel
__global__ void k_vox_build(const float4* __restrict__ pts, const double* __restrict__ cov, const unsigned long long* __restrict__ keys,
                            const int* __restrict__ perm, const int* __restrict__ head, const int* __restrict__ slot, size_t n,
                            unsigned long long* __restrict__ vkeys, float4* __restrict__ vmean, double* __restrict__ vcov)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        if (!head[i]) continue;
        const unsigned long long k = keys[i];
        double m[3] = {0, 0, 0}, c[6] = {0, 0, 0, 0, 0, 0};
        int cnt = 0;
        for (size_t j = i; j < n && keys[j] == k; ++j) {
            const float4 p = pts[perm[j]];
            m[0] += (double)p.x; m[1] += (double)p.y; m[2] += (double)p.z;
            double cp[6];
            cov6_from_normal(cov + kCovDoubles * (size_t)perm[j], cp);
            for (int a = 0; a < 6; ++a) c[a] += cp[a];
            ++cnt;
        }
        const int v = slot[i];
        vkeys[v] = k;
        vmean[v] = make_float4((float)(m[0] / cnt), (float)(m[1] / cnt), (float)(m[2] / cnt), (float)cnt);
        for (int a = 0; a < 6; ++a) vcov[6 * (size_t)v + a] = c[a] / cnt;
    }
}

// G7 linearisation: voxel lookup (binary search in the sorted voxel keys) fused with the 28 fp64 sums.
__global__ __launch_bounds__(kNNThreads) void k_linearize_voxel(
    const float4* __restrict__ src_all, const int64_t* __restrict__ src_offs, const double* __restrict__ src_cov,
    const unsigned long long* __restrict__ vkeys, const float4* __restrict__ vmean, const double* __restrict__ vcov,
    int n_voxels, const LmState* __restrict__ st, GicpParams prm, double* __restrict__ partial, int max_blocks)
{
    __shared__ double red[kNNThreads / 64][kTerms];
    const int pair = blockIdx.y;
    const LmState& S = st[pair];
    if (!S.active) return;
    const int64_t so = src_offs[pair];
    const int n = (int)(src_offs[pair + 1] - so);
    const float4* src = src_all + so;
    double* pout = partial + ((size_t)pair * max_blocks + blockIdx.x) * kTerms;
    const bool error_only = S.phase == 1;
    double T[12], TL[12];
    float Tf[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) { T[i] = S.xi[i]; TL[i] = S.x[i]; Tf[i] = (float)S.x[i]; }
    double acc[kTerms];
#pragma unroll
    for (int i = 0; i < kTerms; ++i) acc[i] = 0.0;
    const int per_block = kNNThreads * kPts;
    const int nn = prm.voxel_neighbors;
    const float resf = (float)prm.voxel_res;
    for (int base = blockIdx.x * per_block; base < n; base += gridDim.x * per_block) {
#pragma unroll 1
        for (int p = 0; p < kPts; ++p) {
            const int i = base + p * kNNThreads + threadIdx.x;
            if (i >= n) continue;
            const float4 a = src[i];
            double ta[3];
#pragma unroll
            for (int r = 0; r < 3; ++r)
                ta[r] = T[4 * r] * (double)a.x + T[4 * r + 1] * (double)a.y + T[4 * r + 2] * (double)a.z + T[4 * r + 3];
            // voxel of the float-transformed point at the LINEARISATION pose
            const int cx = voxel_coord_f(Tf[0] * a.x + Tf[1] * a.y + Tf[2] * a.z + Tf[3], resf) + 32768,
                      cy = voxel_coord_f(Tf[4] * a.x + Tf[5] * a.y + Tf[6] * a.z + Tf[7], resf) + 32768,
                      cz = voxel_coord_f(Tf[8] * a.x + Tf[9] * a.y + Tf[10] * a.z + Tf[11], resf) + 32768;
            double ca[6];
            cov6_from_normal(src_cov + kCovDoubles * (size_t)(so + i), ca);
            const double CA[9] = {ca[0], ca[1], ca[2], ca[1], ca[3], ca[4], ca[2], ca[4], ca[5]};
            double RC[9], RCRa[9];
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int c = 0; c < 3; ++c)
                    RC[3 * r + c] = TL[4 * r] * CA[c] + TL[4 * r + 1] * CA[3 + c] + TL[4 * r + 2] * CA[6 + c];
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int c = 0; c < 3; ++c)
                    RCRa[3 * r + c] = RC[3 * r] * TL[4 * c] + RC[3 * r + 1] * TL[4 * c + 1] + RC[3 * r + 2] * TL[4 * c + 2];
#pragma unroll 1
            for (int o = 0; o < 27; ++o) {
                const int dx = o % 3 - 1, dy = (o / 3) % 3 - 1, dz = o / 9 - 1;
                const int man = abs(dx) + abs(dy) + abs(dz);
                if ((nn == 1 && man != 0) || (nn == 7 && man > 1)) continue;
                const unsigned long long key = voxel_key(pair, cx + dx, cy + dy, cz + dz);
                int lo = 0, hi = n_voxels;  // first index with vkeys >= key
                while (lo < hi) {
                    const int mid = (lo + hi) >> 1;
                    if (vkeys[mid] < key) lo = mid + 1; else hi = mid;
                }
                if (lo >= n_voxels || vkeys[lo] != key) continue;
                const float4 vm = vmean[lo];
                const double* cb = vcov + 6 * (size_t)lo;
                double RCR[9], M[9];
                RCR[0] = RCRa[0] + cb[0]; RCR[1] = RCRa[1] + cb[1]; RCR[2] = RCRa[2] + cb[2];
                RCR[3] = RCRa[3] + cb[1]; RCR[4] = RCRa[4] + cb[3]; RCR[5] = RCRa[5] + cb[4];
                RCR[6] = RCRa[6] + cb[2]; RCR[7] = RCRa[7] + cb[4]; RCR[8] = RCRa[8] + cb[5];
                if (!inv3_sym(RCR, M)) continue;
                const double w = sqrt((double)vm.w);
                double e[3] = {(double)vm.x - ta[0], (double)vm.y - ta[1], (double)vm.z - ta[2]}, Me[3];
#pragma unroll
                for (int r = 0; r < 3; ++r) Me[r] = M[3 * r] * e[0] + M[3 * r + 1] * e[1] + M[3 * r + 2] * e[2];
                acc[27] += w * (e[0] * Me[0] + e[1] * Me[1] + e[2] * Me[2]);
                if (error_only) continue;
                const double J[18] = {0, -ta[2], ta[1], -1, 0, 0,
                                      ta[2], 0, -ta[0], 0, -1, 0,
                                      -ta[1], ta[0], 0, 0, 0, -1};
                double MJ[18];
#pragma unroll
                for (int r = 0; r < 3; ++r)
#pragma unroll
                    for (int c = 0; c < 6; ++c)
                        MJ[6 * r + c] = M[3 * r] * J[c] + M[3 * r + 1] * J[6 + c] + M[3 * r + 2] * J[12 + c];
#pragma unroll
                for (int r = 0; r < 6; ++r) {
#pragma unroll
                    for (int c = r; c < 6; ++c)
                        acc[r * 6 - (r * (r - 1)) / 2 + (c - r)] += w * (J[r] * MJ[c] + J[6 + r] * MJ[6 + c] + J[12 + r] * MJ[12 + c]);
                }
#pragma unroll
                for (int r = 0; r < 6; ++r) acc[21 + r] += w * (J[r] * Me[0] + J[6 + r] * Me[1] + J[12 + r] * Me[2]);
            }
        }
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (int i = 0; i < kTerms; ++i) {
        const double v = wave_sum_d(acc[i]);
        if (lane == 0) red[wave][i] = v;
    }
    __syncthreads();
    if (threadIdx.x < kTerms) {
        double v = 0;
        for (int w = 0; w < kNNThreads / 64; ++w) v += red[w][threadIdx.x];
        pout[threadIdx.x] = v;
    }
}

// ---- device-side LM bookkeeping (one lane per pair) ------------------------------------------
__device__ void mul4d(const double* a, const double* b, double* c)
{
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            double s = 0;
            for (int k = 0; k < 4; ++k) s += a[4 * i + k] * b[4 * k + j];
            c[4 * i + j] = s;
        }
}

__device__ void se3_exp_d(const double* a, double* T)
{
    const double wx = a[0], wy = a[1], wz = a[2];
    const double theta_sq = wx * wx + wy * wy + wz * wz;
    double imag, real, theta = 0;
    if (theta_sq < 1e-10) {
        const double t4 = theta_sq * theta_sq;
        imag = 0.5 - (1.0 / 48.0) * theta_sq + (1.0 / 3840.0) * t4;
        real = 1.0 - (1.0 / 8.0) * theta_sq + (1.0 / 384.0) * t4;
    } else {
        theta = sqrt(theta_sq);
        imag = sin(0.5 * theta) / theta;
        real = cos(0.5 * theta);
    }
    double qw = real, qx = imag * wx, qy = imag * wy, qz = imag * wz;
    const double nq = sqrt(qw * qw + qx * qx + qy * qy + qz * qz);
    qw /= nq; qx /= nq; qy /= nq; qz /= nq;
    const double R[9] = {1 - 2 * (qy * qy + qz * qz), 2 * (qx * qy - qz * qw), 2 * (qx * qz + qy * qw),
                         2 * (qx * qy + qz * qw), 1 - 2 * (qx * qx + qz * qz), 2 * (qy * qz - qx * qw),
                         2 * (qx * qz - qy * qw), 2 * (qy * qz + qx * qw), 1 - 2 * (qx * qx + qy * qy)};
    double V[9];
    if (theta < 1e-10) {
        for (int i = 0; i < 9; ++i) V[i] = R[i];
    } else {
        const double O[9] = {0, -wz, wy, wz, 0, -wx, -wy, wx, 0};
        double O2[9];
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) O2[3 * i + j] = O[3 * i] * O[j] + O[3 * i + 1] * O[3 + j] + O[3 * i + 2] * O[6 + j];
        const double c1 = (1.0 - cos(theta)) / theta_sq, c2 = (theta - sin(theta)) / (theta_sq * theta);
        for (int i = 0; i < 9; ++i) V[i] = (i % 4 == 0 ? 1.0 : 0.0) + c1 * O[i] + c2 * O2[i];
    }
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) T[4 * i + j] = R[3 * i + j];
        T[4 * i + 3] = V[3 * i] * a[3] + V[3 * i + 1] * a[4] + V[3 * i + 2] * a[5];
    }
    T[12] = T[13] = T[14] = 0; T[15] = 1;
}

__device__ bool solve6_d(const double* Hin, const double* rhs, double* x)
{
    double L[36], D[6];
    for (int i = 0; i < 36; ++i) L[i] = 0;
    for (int j = 0; j < 6; ++j) {
        double d = Hin[6 * j + j];
        for (int k = 0; k < j; ++k) d -= L[6 * j + k] * L[6 * j + k] * D[k];
        D[j] = d;
        if (d == 0.0 || !(d == d)) return false;
        for (int i = j + 1; i < 6; ++i) {
            double s = Hin[6 * i + j];
            for (int k = 0; k < j; ++k) s -= L[6 * i + k] * L[6 * j + k] * D[k];
            L[6 * i + j] = s / d;
        }
    }
    double y[6];
    for (int i = 0; i < 6; ++i) { double s = rhs[i]; for (int k = 0; k < i; ++k) s -= L[6 * i + k] * y[k]; y[i] = s; }
    for (int i = 0; i < 6; ++i) y[i] /= D[i];
    for (int i = 5; i >= 0; --i) { double s = y[i]; for (int k = i + 1; k < 6; ++k) s -= L[6 * k + i] * x[k]; x[i] = s; }
    return true;
}

__device__ bool is_converged_d(const GicpParams& p, const double* delta)
{
    double mr = 0, mt = 0;
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) mr = fmax(mr, p.conv_factor * fabs(delta[4 * i + j] - (i == j ? 1.0 : 0.0)) / p.rot_eps);
        mt = fmax(mt, p.conv_factor * fabs(delta[4 * i + 3]) / p.trans_eps);
    }
    return fmax(mr, mt) < 1.0;
}

// propose the next candidate from (H, b, lambda); marks the pair failed if the solve breaks down
__device__ void propose(LmState& S)
{
    double Hl[36], rhs[6];
    for (int i = 0; i < 36; ++i) Hl[i] = S.H[i];
    for (int i = 0; i < 6; ++i) { Hl[7 * i] += S.lambda; rhs[i] = -S.b[i]; }
    if (!solve6_d(Hl, rhs, S.d)) { S.failed = 1; S.active = 0; S.phase = 2; return; }
    se3_exp_d(S.d, S.delta);
    mul4d(S.delta, S.x, S.xi);
    ++S.trials;
}

// LsqRegistration::step_lm / computeTransformation bookkeeping; grid = pairs, 64 lanes each.
//   phase 0 result = linearize(x0): H, b, y0 -> first LM candidate, phase 1
//   phase 1 result = compute_error(delta * x0) on the cached correspondences -> rho -> accept (x0 <- xi, next
//   outer iteration linearises again: phase 0) or reject (lambda *= nu, next candidate, stay in phase 1)
// n_next[0] counts the pairs that need a linearisation next tick, n_next[1] the pairs in an LM trial, n_next[2] those of [0] that moved
// farther than prm.motion_switch in the step just accepted.
constexpr int kLmThreads = 256;     // four waves share the 28 terms of the final sum (rounds 1-5: one wave, 28 dependent reductions in a row)
__global__ __launch_bounds__(kLmThreads) void k_lm_update(LmState* __restrict__ st, const double* __restrict__ partial, const int* __restrict__ nblocks,
                            int max_blocks, GicpParams prm, int* __restrict__ n_next, int trial_only)
{
    const int pair = blockIdx.x;
    LmState& S = st[pair];
    if (!S.active) return;
    if (trial_only && S.phase == 0) {                   // see k_linearize: the pair waits for the next full tick; it still counts as "to linearise"
        if (threadIdx.x == 0) {
            atomicAdd(&n_next[0], 1);
            if (pair_motion(S) > prm.motion_switch) atomicAdd(&n_next[2], 1);
        }
        return;
    }
    __shared__ double sum[kTerms];
    {   // fixed-order (deterministic) final sum of the per-workgroup partials: lane l adds blocks l, l + 64, ...
        // in ascending order, then one wave butterfly per term; wave w takes the terms t0 + w, t0 + w + 4, ... (the order INSIDE a term is
        // what fixes its bits, and that is unchanged)
        const double* p = partial + (size_t)pair * max_blocks * kTerms;
        const int nb = nblocks[pair];
        const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
        for (int t = (S.phase == 1 ? kTerms - 1 : 0) + wave; t < kTerms; t += kLmThreads / 64) {
            double v = 0;
            for (int b = lane; b < nb; b += 64) v += p[(size_t)b * kTerms + t];
            v = wave_sum_d(v);
            if (lane == 0) sum[t] = v;
        }
    }
    __syncthreads();
    if (threadIdx.x != 0) return;
    const double y = sum[27];
    const int limit = prm.force_iters > 0 ? prm.force_iters : prm.max_iter;
    if (S.phase == 0) n_next[3] = 1;                    // this tick carried a linearisation (= a nearest-neighbour pass): counted by the host

    if (S.phase == 0) {
        int t = 0;
        for (int r = 0; r < 6; ++r)
            for (int c = r; c < 6; ++c) { S.H[6 * r + c] = sum[t]; S.H[6 * c + r] = sum[t]; ++t; }
        for (int r = 0; r < 6; ++r) S.b[r] = sum[21 + r];
        S.y0 = y;
        if (S.lambda < 0.0) {
            double mx = 0;
            for (int i = 0; i < 6; ++i) mx = fmax(mx, fabs(S.H[7 * i]));
            S.lambda = prm.lm_init_factor * mx;
        }
        S.nu = 2.0;
        S.inner = 0;
        S.phase = 1;
        propose(S);
    } else {
        double denom = 0;
        for (int i = 0; i < 6; ++i) denom += S.d[i] * (S.lambda * S.d[i] - S.b[i]);
        const double rho = (S.y0 - y) / denom;
        bool stepped = false;   // step_lm returned true: one outer iteration is complete
        if (!(rho == rho)) {
            S.failed = 1; S.active = 0; S.phase = 2;
        } else if (rho < 0) {
            if (is_converged_d(prm, S.delta)) {
                stepped = true;     // returns true without moving; the outer loop then sees a converged delta
            } else {
                S.lambda = S.nu * S.lambda;
                S.nu = 2 * S.nu;
                ++S.inner;
                if (S.inner >= prm.lm_max_iter) { S.failed = 1; S.active = 0; S.phase = 2; }  // "lm not converged"
                else propose(S);
            }
        } else {
            for (int i = 0; i < 16; ++i) S.x[i] = S.xi[i];
            const double w = 2 * rho - 1;
            S.lambda = S.lambda * fmax(1.0 / 3.0, 1.0 - w * w * w);
            for (int i = 0; i < 36; ++i) S.final_H[i] = S.H[i];
            stepped = true;
        }
        if (stepped) {
            ++S.outer;
            const bool conv = prm.force_iters > 0 ? false : is_converged_d(prm, S.delta);
            if (conv) S.converged = 1;
            if (conv || S.outer >= limit) { S.active = 0; S.phase = 2; }
            else {
                for (int i = 0; i < 16; ++i) S.xi[i] = S.x[i];
                S.phase = 0;
            }
        }
    }
    if (S.active) atomicAdd(&n_next[S.phase == 0 ? 0 : 1], 1);
    if (S.active && S.phase == 0 && pair_motion(S) > prm.motion_switch) atomicAdd(&n_next[2], 1);   // pairs whose next search is a broad one
}

// G6: fitness partials: [pair][block][2] = (sum of d^2 <= max_range, count)
__global__ __launch_bounds__(kNNThreads) void k_fitness(const float4* __restrict__ src_all,
                                                        const int64_t* __restrict__ src_offs,
                                                        const float4* __restrict__ tgt_all,
                                                        const int64_t* __restrict__ tgt_offs,
                                                        const int* __restrict__ tgt_tile_base,
                                                        const float4* __restrict__ tlo, const float4* __restrict__ thi,
        const float4* __restrict__ mlo, const float4* __restrict__ mhi,
                                                        const double* __restrict__ poses /* [pairs][16] */,
                                                        double max_range, double* __restrict__ partial, int max_blocks,
                                                        const int* __restrict__ nn_seed /* optional warm start */)
{
    __shared__ double red[kNNThreads / 64][2];
    const int pair = blockIdx.y;
    const int64_t so = src_offs[pair], to = tgt_offs[pair];
    const int n = (int)(src_offs[pair + 1] - so), m = (int)(tgt_offs[pair + 1] - to);
    const float4* src = src_all + so;
    const float4* tgt = tgt_all + to;
    Hier H;
    H.tlo = tlo + tgt_tile_base[pair]; H.thi = thi + tgt_tile_base[pair];
    H.mlo = mlo + (size_t)64 * tgt_tile_base[pair]; H.mhi = mhi + (size_t)64 * tgt_tile_base[pair];
    H.ntiles = (m + kTile - 1) / kTile;
    const float maxc2 = max_range < 3.0e38 ? (float)max_range * 1.0001f : INFINITY;
    float Tf[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) Tf[i] = (float)poses[(size_t)pair * 16 + i];
    constexpr int PF = 2;  // source points per lane (see launch_nn_scan); the grid-stride loop covers any grid
    const int per_block = kNNThreads * PF;
    double s = 0, c = 0;
    for (int base = blockIdx.x * per_block; base < n; base += gridDim.x * per_block) {
        float qx[PF], qy[PF], qz[PF];
        int si[PF];
        bool live[PF];
#pragma unroll
        for (int p = 0; p < PF; ++p) {
            si[p] = base + (threadIdx.x >> 6) * (64 * PF) + p * 64 + (threadIdx.x & 63);
            live[p] = si[p] < n;
            const float4 a = src[live[p] ? si[p] : 0];
            qx[p] = Tf[0] * a.x + Tf[1] * a.y + Tf[2] * a.z + Tf[3];
            qy[p] = Tf[4] * a.x + Tf[5] * a.y + Tf[6] * a.z + Tf[7];
            qz[p] = Tf[8] * a.x + Tf[9] * a.y + Tf[10] * a.z + Tf[11];
        }
        float best[PF];
        int bidx[PF];
        int seed[PF];   // the neighbours of the last alignment pass: valid upper bounds at any pose
#pragma unroll
        for (int p = 0; p < PF; ++p) seed[p] = (nn_seed && live[p]) ? nn_seed[so + si[p]] : -1;
        nn_scan<PF>(tgt, m, H, maxc2, qx, qy, qz, live, best, bidx, seed);
#pragma unroll
        for (int p = 0; p < PF; ++p)
            if (live[p] && bidx[p] >= 0 && (double)best[p] <= max_range) { s += (double)best[p]; c += 1.0; }
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    s = wave_sum_d(s); c = wave_sum_d(c);
    if (lane == 0) { red[wave][0] = s; red[wave][1] = c; }
    __syncthreads();
    if (threadIdx.x < 2) {
        double v = 0;
        for (int w = 0; w < kNNThreads / 64; ++w) v += red[w][threadIdx.x];
        partial[((size_t)pair * max_blocks + blockIdx.x) * 2 + threadIdx.x] = v;
    }
}

// corr (sorted space) -> caller's indexing: out[so + orig_src] = orig_tgt (or -1)
__global__ void k_corr_to_original(const float4* __restrict__ src_all, const int64_t* __restrict__ src_offs,
                                   const float4* __restrict__ tgt_all, const int64_t* __restrict__ tgt_offs,
                                   const int* __restrict__ corr, int* __restrict__ out)
{
    const int pair = blockIdx.y;
    const int64_t so = src_offs[pair], to = tgt_offs[pair];
    const int n = (int)(src_offs[pair + 1] - so);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int j = corr[so + i];
        out[so + __float_as_int(src_all[so + i].w)] = j >= 0 ? __float_as_int(tgt_all[to + j].w) : -1;
    }
}

}  // namespace

// Small batches (<= kLmWindowPairs pairs: ONE registration at a time is how the nodes call it, main_RING.py:81-104, global_manager.cpp:2016-2021)
// run the LM schedule in windows of kLmWindow ticks without a host round trip in between: every kernel of a tick gates itself on the pair's
// device-side state (active / phase / motion), so a tick launched for a pair that has converged, or the search kernels of a tick that is an LM
// trial, are empty launches.  The host reads the per-tick counters once per window instead of copying + synchronising after every tick
// (rounds 1-5: ~10 round trips of 40-60 us per registration of two 35 k-point clouds).  Same kernels on the same state: same bits.
constexpr int kLmWindowPairs = 8;
constexpr int kLmWindow = 4;
constexpr int kLmWindowMax = 16;
// ------------------------------------------------------------------------------------------------
struct mrs_gicp_batch {
    mrs_ctx* ctx = nullptr;
    int n_pairs = 0;
    GicpParams prm;
    std::vector<int64_t> offs[2];   // host copies: [0] source, [1] target
    int64_t* d_offs[2] = {nullptr, nullptr};
    float4* d_pts[2] = {nullptr, nullptr};
    double* d_cov[2] = {nullptr, nullptr};   // sorted space: the unit normal of every point (kCovDoubles = 3 doubles; cov6_from_normal)
    int* d_tile_base[2] = {nullptr, nullptr};  // [n_pairs] first tile of each cloud
    float4* d_tlo[2] = {nullptr, nullptr};     // tile bounding boxes
    float4* d_thi[2] = {nullptr, nullptr};
    float4* d_mlo[2] = {nullptr, nullptr};     // boxes of the 64 minis (16 points) of every tile
    float4* d_mhi[2] = {nullptr, nullptr};
    int max_tiles[2] = {0, 0};                 // tiles of the largest cloud
    int64_t cap_points[2] = {0, 0};            // capacity of d_pts / d_cov (points), d_tlo / d_thi (cap_tiles): buffers are kept
    int cap_tiles[2] = {0, 0};                 // across setInput* calls and only re-allocated when a cloud outgrows them
    bool cov_valid[2] = {false, false};
    LmState* d_state = nullptr;
    double* d_partial = nullptr;
    int* d_nblocks = nullptr;
    int* d_nactive = nullptr;
    int* d_bbox[2] = {nullptr, nullptr};  // [n_pairs][6] bounding box of each cloud (ordered ints), the Morton grid
    int* d_corr = nullptr;          // [total source points] correspondences of the current evaluation
    int* d_seed = nullptr;          // [total source points] last nearest neighbour (warm start of the next NN pass)
    size_t n_seed = 0;
    unsigned long long* d_vkeys = nullptr;  // G7 voxel map of the targets (sorted keys, all pairs)
    float4* d_vmean = nullptr;
    double* d_vcov = nullptr;
    int n_voxels = 0;
    double vox_res_built = 0.0;
    int max_blocks = 0;             // workgroups per pair of the reduction kernels for the CURRENT clouds (ensure_state)
    int cap_blocks = 0;             // ... d_partial was allocated for
    int longest_src = 0;            // points in the largest source cloud (grid of the NN scan)
    double last_nn_passes = 0;
    // round-4 search structure (nn_core.hpp): octree-cell leaves of <= 16 points, tiles of 64 leaves, supers of 64 tiles, per cloud
    float4* d_llo[2] = {nullptr, nullptr};
    float4* d_lhi[2] = {nullptr, nullptr};
    float4* d_t2lo[2] = {nullptr, nullptr};
    float4* d_t2hi[2] = {nullptr, nullptr};
    float4* d_slo[2] = {nullptr, nullptr};
    float4* d_shi[2] = {nullptr, nullptr};
    int* d_leaf_first[2] = {nullptr, nullptr};    // [n_pairs + 1] each
    int* d_tile_first[2] = {nullptr, nullptr};
    int* d_super_first[2] = {nullptr, nullptr};
    int cap_leaves[2] = {0, 0}, cap_tiles2[2] = {0, 0}, cap_supers[2] = {0, 0};
    std::vector<int> h_leaf_first[2], h_tile_first[2], h_super_first[2];   // host copies (set_clouds_from re-bases them)
    int n_leaves[2] = {0, 0};
    int search_core = 1;            // 1: octree leaves + query groups (round 4), 0: round-3 wave-shared traversal (A/B, cross-check)
    int cold_core = 0;              // search_core 1: kernel of the FIRST pass of an align() (0: round-3 kernel, 1: round-4 kernel)
    bool use_certificates = true;   // search_core 1: certify unchanged neighbours before searching (k_nn_certify)
    CertArrays cert = {nullptr, nullptr, nullptr, nullptr, 0, nullptr};
    double last_searched = 0;       // share of (source point, pass) that needed a search in the last align()
    bool want_leaf_hier = true;     // build the octree-cell hierarchy in set_clouds (false: RING++ front end)
    bool no_cov = false;            // RING++ front end: no covariance buffers
    bool hier_valid[2] = {false, false};
    int big_movers = 1;             // pairs whose last step exceeded motion_switch (counted by k_lm_update): do they need the round-3 kernel?
    HierArrays hier(int w) const
    {
        return HierArrays{d_llo[w], d_lhi[w], d_t2lo[w], d_t2hi[w], d_slo[w], d_shi[w], d_leaf_first[w], d_tile_first[w], d_super_first[w]};
    }
};

namespace {

// dst[seg.dst + i] = src[seg.src + i], i < seg.count, for every segment (one per pair; blockIdx.y): how set_clouds_from moves a stored cloud's
// arrays into a pair's slot.  seg = {src offset, dst offset, count} in units of T.
template <class T>
__global__ void k_copy_segments(const T* __restrict__ src, T* __restrict__ dst, const int64_t* __restrict__ seg, int64_t scale)
{
    const int64_t so = seg[3 * blockIdx.y] * scale, dofs = seg[3 * blockIdx.y + 1] * scale, n = seg[3 * blockIdx.y + 2] * scale;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) dst[dofs + i] = src[so + i];
}

void free_cloud(mrs_gicp_batch* h, int w)
{
    if (h->d_offs[w]) (void)hipFree(h->d_offs[w]);
    if (h->d_pts[w]) (void)hipFree(h->d_pts[w]);
    if (h->d_cov[w]) (void)hipFree(h->d_cov[w]);
    if (h->d_tile_base[w]) (void)hipFree(h->d_tile_base[w]);
    if (h->d_bbox[w]) (void)hipFree(h->d_bbox[w]);
    if (h->d_tlo[w]) (void)hipFree(h->d_tlo[w]);
    if (h->d_thi[w]) (void)hipFree(h->d_thi[w]);
    if (h->d_mlo[w]) (void)hipFree(h->d_mlo[w]);
    if (h->d_mhi[w]) (void)hipFree(h->d_mhi[w]);
    for (void* p : {(void*)h->d_llo[w], (void*)h->d_lhi[w], (void*)h->d_t2lo[w], (void*)h->d_t2hi[w], (void*)h->d_slo[w], (void*)h->d_shi[w],
                    (void*)h->d_leaf_first[w], (void*)h->d_tile_first[w], (void*)h->d_super_first[w]})
        if (p) (void)hipFree(p);
    h->d_llo[w] = h->d_lhi[w] = h->d_t2lo[w] = h->d_t2hi[w] = h->d_slo[w] = h->d_shi[w] = nullptr;
    h->d_leaf_first[w] = h->d_tile_first[w] = h->d_super_first[w] = nullptr;
    h->cap_leaves[w] = h->cap_tiles2[w] = h->cap_supers[w] = 0; h->n_leaves[w] = 0;
    h->d_mlo[w] = nullptr; h->d_mhi[w] = nullptr;
    h->d_offs[w] = nullptr; h->d_pts[w] = nullptr; h->d_cov[w] = nullptr;
    h->d_tile_base[w] = nullptr; h->d_tlo[w] = nullptr; h->d_thi[w] = nullptr; h->d_bbox[w] = nullptr;
    h->cov_valid[w] = false;
    h->cap_points[w] = 0; h->cap_tiles[w] = 0;
}

int blocks_for_points(int n) { return (n + kNNThreads * kPts - 1) / (kNNThreads * kPts); }
constexpr int kLinChunks = 4;
constexpr int kLinVariant = 3;    // launch_linearize     // blocks of 1024 points per workgroup of the reduction kernels (ensure_state)

// Source points per lane in the NN scan.  Fewer points per wave = a more compact query set = sharper sub-tile
// culling; more = every LDS candidate read serves more distance evaluations.  Measured (120k x 120k, MI355X):
// 2 beats 4 at every batch size (23.5k vs 21.4k it/s at 256 pairs, 14.3k vs 11.1k at 16) and 1 only wins when a
// single pair would otherwise leave most CUs idle.
template <class... Args>
void launch_nn_scan(int longest_src, int n_pairs, int num_cu, hipStream_t s, Args... args)
{
    const int cus = num_cu > 0 ? num_cu : 256;
    auto wgs = [&](int P) { return (long)n_pairs * ((longest_src + kNNThreads * P - 1) / (kNNThreads * P)); };
    static const char* const force_p_s = mrs::dev_env("MRS_NN_P");
    static const int force_p = force_p_s ? atoi(force_p_s) : 0;     // development aid: 1 or 2 source points per lane
    if (force_p == 2 || (force_p != 1 && wgs(2) >= 3L * cus))
        hipLaunchKernelGGL(k_nn_scan<2>, dim3((unsigned)(wgs(2) / n_pairs), n_pairs), dim3(kNNThreads), 0, s, args...);
    else
        hipLaunchKernelGGL(k_nn_scan<1>, dim3((unsigned)(wgs(1) / n_pairs), n_pairs), dim3(kNNThreads), 0, s, args...);
}

// k_linearize instantiation: 3 (default) = 3 waves per SIMD (142 registers, no spills), normals one point ahead: 0.55 ms per 256 pairs; 1 = 4 waves
// (128 registers, 3 spilled), normals requested at use: 0.69 ms; 2 = 4 waves + look-ahead (26 spills): 0.95 ms; 4 = 3 waves, no look-ahead: 0.58 ms; 5 = 2 waves.  MRS_DEV=1 MRS_LIN_VARIANT=<n> switches for A/B runs (same sums, same bits).
template <class... Args>
void launch_linearize(dim3 grid, hipStream_t s, Args... args)
{
    static const char* const v_s = mrs::dev_env("MRS_LIN_VARIANT");
    static const int v = v_s ? atoi(v_s) : kLinVariant;
    if (v == 2) hipLaunchKernelGGL((k_linearize<4, true>), grid, dim3(kNNThreads), 0, s, args...);
    else if (v == 3) hipLaunchKernelGGL((k_linearize<3, true>), grid, dim3(kNNThreads), 0, s, args...);
    else if (v == 4) hipLaunchKernelGGL((k_linearize<3, false>), grid, dim3(kNNThreads), 0, s, args...);
    else if (v == 5) hipLaunchKernelGGL((k_linearize<2, true>), grid, dim3(kNNThreads), 0, s, args...);
    else hipLaunchKernelGGL((k_linearize<4, false>), grid, dim3(kNNThreads), 0, s, args...);
}

// Octree-cell leaves + tiles + supers of every cloud of side `w` from the sorted keys (set_clouds).  Synchronises (the leaf counts size the arrays).
int build_leaf_hier(mrs_gicp_batch* h, int w, const unsigned long long* d_keys, int64_t total, hipStream_t s)
{
    const int P = h->n_pairs;
    mrs::Scratch cellhead, cellstart, head, leafid, tmp;
    int st;
    for (mrs::Scratch* b : {&cellhead, &cellstart, &head, &leafid})
        if ((st = b->alloc((size_t)total * sizeof(int), s)) != MRS_OK) return st;
    const int fb = (int)std::min<int64_t>((total + 255) / 256, 8192);
    hipLaunchKernelGGL(nnc::k_leaf_level, dim3(fb), dim3(256), 0, s, d_keys, (size_t)total, cellhead.as<int>());
    size_t b1 = 0, b2 = 0;
    MRS_HIP_TRY(hipcub::DeviceScan::InclusiveScan(nullptr, b1, cellhead.as<int>(), cellstart.as<int>(), hipcub::Max(), (int)total, s));
    MRS_HIP_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, b2, head.as<int>(), leafid.as<int>(), (int)total, s));
    if ((st = tmp.alloc(std::max(b1, b2), s)) != MRS_OK) return st;
    MRS_HIP_TRY(hipcub::DeviceScan::InclusiveScan(tmp.p, b1, cellhead.as<int>(), cellstart.as<int>(), hipcub::Max(), (int)total, s));
    hipLaunchKernelGGL(nnc::k_leaf_heads, dim3(fb), dim3(256), 0, s, cellstart.as<int>(), (size_t)total, head.as<int>());
    MRS_HIP_TRY(hipcub::DeviceScan::ExclusiveSum(tmp.p, b2, head.as<int>(), leafid.as<int>(), (int)total, s));
    if (!h->d_leaf_first[w]) {
        MRS_HIP_TRY(hipMalloc(&h->d_leaf_first[w], (size_t)(P + 1) * sizeof(int)));
        MRS_HIP_TRY(hipMalloc(&h->d_tile_first[w], (size_t)(P + 1) * sizeof(int)));
        MRS_HIP_TRY(hipMalloc(&h->d_super_first[w], (size_t)(P + 1) * sizeof(int)));
    }
    hipLaunchKernelGGL(nnc::k_leaf_first, dim3((P + 1 + 255) / 256), dim3(256), 0, s, head.as<int>(), leafid.as<int>(), h->d_offs[w], P,
                       h->d_leaf_first[w]);
    MRS_HIP_TRY(hipGetLastError());
    // tiles: octree cells of <= 256 points (every leaf cell lies inside one of them)
    mrs::Scratch tpre, thead, tid;
    if ((st = tpre.alloc((size_t)total, s)) != MRS_OK) return st;
    if ((st = thead.alloc((size_t)total * sizeof(int), s)) != MRS_OK) return st;
    if ((st = tid.alloc((size_t)total * sizeof(int), s)) != MRS_OK) return st;
    hipLaunchKernelGGL(nnc::k_tile_prefix, dim3(fb), dim3(256), 0, s, d_keys, (size_t)total, tpre.as<signed char>());
    hipLaunchKernelGGL(nnc::k_tile_heads, dim3((unsigned)std::min<int64_t>((total + 1023) / 1024, 16384)), dim3(1024), 0, s, d_keys,
                       (const signed char*)tpre.as<signed char>(), (size_t)total, thead.as<int>());
    MRS_HIP_TRY(hipcub::DeviceScan::ExclusiveSum(tmp.p, b2, thead.as<int>(), tid.as<int>(), (int)total, s));
    hipLaunchKernelGGL(nnc::k_leaf_first, dim3((P + 1 + 255) / 256), dim3(256), 0, s, thead.as<int>(), tid.as<int>(), h->d_offs[w], P,
                       h->d_tile_first[w]);
    MRS_HIP_TRY(hipGetLastError());
    std::vector<int> lf(P + 1), tf(P + 1), sf(P + 1);
    MRS_HIP_TRY(hipMemcpyAsync(lf.data(), h->d_leaf_first[w], lf.size() * sizeof(int), hipMemcpyDeviceToHost, s));
    MRS_HIP_TRY(hipMemcpyAsync(tf.data(), h->d_tile_first[w], tf.size() * sizeof(int), hipMemcpyDeviceToHost, s));
    MRS_HIP_TRY(hipStreamSynchronize(s));
    int most_supers = 0;
    sf[0] = 0;
    for (int c = 0; c < P; ++c) {
        const int nt = tf[c + 1] - tf[c], ns = (nt + 63) / 64;
        sf[c + 1] = sf[c] + ns;
        most_supers = std::max(most_supers, ns);
    }
    h->n_leaves[w] = lf[P];
    h->h_leaf_first[w] = lf; h->h_tile_first[w] = tf; h->h_super_first[w] = sf;
    auto grow = [](float4*& a, float4*& b, int& cap, int need) -> hipError_t {
        if (need <= cap && a) return hipSuccess;
        if (a) (void)hipFree(a);
        if (b) (void)hipFree(b);
        a = b = nullptr;
        cap = need + need / 8 + 1;
        hipError_t e = hipMalloc(&a, (size_t)cap * sizeof(float4));
        return e != hipSuccess ? e : hipMalloc(&b, (size_t)cap * sizeof(float4));
    };
    MRS_HIP_TRY(grow(h->d_llo[w], h->d_lhi[w], h->cap_leaves[w], lf[P]));
    MRS_HIP_TRY(grow(h->d_t2lo[w], h->d_t2hi[w], h->cap_tiles2[w], tf[P]));
    MRS_HIP_TRY(grow(h->d_slo[w], h->d_shi[w], h->cap_supers[w], sf[P]));
    MRS_HIP_TRY(hipMemcpyAsync(h->d_super_first[w], sf.data(), sf.size() * sizeof(int), hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(nnc::k_leaf_boxes, dim3(fb), dim3(256), 0, s, (const float4*)h->d_pts[w], d_keys, head.as<int>(), leafid.as<int>(),
                       h->d_offs[w], (size_t)total, h->d_llo[w], h->d_lhi[w]);
    hipLaunchKernelGGL(nnc::k_tile_boxes, dim3(fb), dim3(256), 0, s, d_keys, (const int*)thead.as<int>(), (const int*)tid.as<int>(),
                       (const int*)head.as<int>(), (const int*)leafid.as<int>(), (const int*)h->d_leaf_first[w], (const int64_t*)h->d_offs[w],
                       (size_t)total, (const float4*)h->d_llo[w], (const float4*)h->d_lhi[w], h->d_t2lo[w], h->d_t2hi[w]);
    hipLaunchKernelGGL(nnc::k_group_boxes, dim3(most_supers, P), dim3(64), 0, s, (const float4*)h->d_t2lo[w], (const float4*)h->d_t2hi[w],
                       (const int*)h->d_tile_first[w], (const int*)h->d_super_first[w], h->d_slo[w], h->d_shi[w]);
    MRS_HIP_TRY(hipGetLastError());
    MRS_HIP_TRY(hipStreamSynchronize(s));     // tf / sf are temporaries
    return MRS_OK;
}

// k_nn_scan_g, or its instrumented twin under MRS_DEV=1 MRS_NN_PROF=1 (per-launch phase cycles and event counts on stderr; synchronises)
template <bool WORK, class... Args>
void launch_nn_scan_g(dim3 grid, hipStream_t s, Args... args)
{
    static const bool prof = mrs::dev_env("MRS_NN_PROF") != nullptr;
    if (!prof) {
        hipLaunchKernelGGL((k_nn_scan_g<false, WORK>), grid, dim3(kNNThreads), 0, s, args...);
        return;
    }
    unsigned long long z[16] = {0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(nnc::g_prof), z, sizeof(z));
    hipLaunchKernelGGL((k_nn_scan_g<true, WORK>), grid, dim3(kNNThreads), 0, s, args...);
    (void)hipStreamSynchronize(s);
    (void)hipMemcpyFromSymbol(z, HIP_SYMBOL(nnc::g_prof), sizeof(z));
    const double w = (double)(z[0] ? z[0] : 1);
    fprintf(stderr, "[nn prof] %llu waves%s: cycles/wave total %.0f (max %llu) = setup %.0f + top %.0f + leaf %.0f + drain %.0f; per wave: tiles near %.1f needed %.1f, "
                    "(group,tile) %.1f, query-vs-leaves tests %.1f, leaves queued %.1f, drains %.2f, batches %.2f, staged points (max group) %.1f\n",
            z[0], WORK ? " (work list)" : "", z[13] / w, z[14], z[1] / w, z[2] / w, z[3] / w, z[4] / w, z[5] / w, z[6] / w, z[7] / w, z[8] / w, z[9] / w, z[12] / w, z[10] / w, z[11] / w);
}

// One nearest-neighbour pass for every pair in phase 0 (h->d_state): fills h->d_corr / h->d_seed.
// mode 0: first pass of an align(), 1: later pass, 2: one plain search with the selected core (linearize hook).
// search_core 0: the round-3 kernel, every point, every pass.  search_core 1 (round-4 schedule):
//   * first pass, and every pair whose last step moved it by more than prm.motion_switch: the round-3 kernel -- a search whose radius
//     is decimetres is a broad search, and brute force over fat minis is at its best there (measured: 17 against 26 ms for 5 cold
//     passes of 64 pairs); it leaves no certificates;
//   * the other pairs: certify the previous pass's neighbours, search what could not be certified (k_nn_scan_g leaves certificates).
int nn_pass(mrs_gicp_batch* h, int mode, hipStream_t s)
{
    const dim3 wg((unsigned)((h->longest_src + kCertBlock - 1) / kCertBlock), h->n_pairs);      // one workgroup per 1024 source points
    const dim3 pg((h->n_pairs + 255) / 256);
    MRS_REQUIRE(h->search_core == 0 || h->hier_valid[1], "target hierarchy missing: set the target clouds after choosing the search setting");
    auto round3 = [&](int gate) {
        launch_nn_scan(h->longest_src, h->n_pairs, h->ctx->num_cu, s, h->d_pts[0], h->d_offs[0], h->d_pts[1], h->d_offs[1],
                       h->d_tile_base[1], h->d_tlo[1], h->d_thi[1], h->d_mlo[1], h->d_mhi[1], h->d_state, h->prm, h->d_corr, h->d_seed,
                       (const int*)h->d_bbox[1], h->search_core == 1 ? h->cert.lb : (float*)nullptr, gate);
    };
    if (h->search_core == 0) {
        round3(0);
    } else if (mode == 0 && h->cold_core == 0) {
        round3(0);
        hipLaunchKernelGGL(k_nn_store_pose, pg, dim3(256), 0, s, (const LmState*)h->d_state, h->n_pairs, h->cert, 0, (const int64_t*)h->d_offs[0], h->prm.motion_switch);
    } else if (mode != 1 || !h->use_certificates) {
        launch_nn_scan_g<false>(wg, s, (const float4*)h->d_pts[0], (const int64_t*)h->d_offs[0], (const float4*)h->d_pts[1], (const int64_t*)h->d_offs[1],
                                h->hier(1), (const LmState*)h->d_state, h->prm, h->d_corr, h->d_seed, (const int*)h->d_bbox[1], h->cert);
        hipLaunchKernelGGL(k_nn_store_pose, pg, dim3(256), 0, s, (const LmState*)h->d_state, h->n_pairs, h->cert, 0, (const int64_t*)h->d_offs[0], h->prm.motion_switch);
    } else {
        if (h->big_movers > 0) round3(1);
        hipLaunchKernelGGL(k_nn_certify, wg, dim3(256), 0, s, (const float4*)h->d_pts[0],
                           (const int64_t*)h->d_offs[0], (const float4*)h->d_pts[1], (const int64_t*)h->d_offs[1], (const LmState*)h->d_state, h->prm,
                           h->d_corr, (const int*)h->d_seed, h->cert);
        launch_nn_scan_g<true>(wg, s, (const float4*)h->d_pts[0], (const int64_t*)h->d_offs[0], (const float4*)h->d_pts[1], (const int64_t*)h->d_offs[1],
                               h->hier(1), (const LmState*)h->d_state, h->prm, h->d_corr, h->d_seed, (const int*)h->d_bbox[1], h->cert);
        hipLaunchKernelGGL(k_nn_store_pose, pg, dim3(256), 0, s, (const LmState*)h->d_state, h->n_pairs, h->cert, 1, (const int64_t*)h->d_offs[0], h->prm.motion_switch);
    }
    MRS_HIP_TRY(hipGetLastError());
    return MRS_OK;
}

// k nearest neighbours of every point of side `w` (k_knn_select) into knn (layout: knn_at)
int launch_knn_select(mrs_gicp_batch* h, int w, int k, int* d_knn, hipStream_t s)
{
    int64_t longest = 0;
    for (int i = 0; i < h->n_pairs; ++i) longest = std::max(longest, h->offs[w][i + 1] - h->offs[w][i]);
    const dim3 grid((unsigned)((longest + kNNThreads - 1) / kNNThreads), h->n_pairs);
    const HierArrays HA = h->hier(w);
    if (k <= 16)
        hipLaunchKernelGGL((k_knn_select<16>), grid, dim3(kNNThreads), 0, s, (const float4*)h->d_pts[w], (const int64_t*)h->d_offs[w], HA, k, d_knn);
    else if (k <= 20)
        hipLaunchKernelGGL((k_knn_select<20>), grid, dim3(kNNThreads), 0, s, (const float4*)h->d_pts[w], (const int64_t*)h->d_offs[w], HA, k, d_knn);
    else
        hipLaunchKernelGGL((k_knn_select<32>), grid, dim3(kNNThreads), 0, s, (const float4*)h->d_pts[w], (const int64_t*)h->d_offs[w], HA, k, d_knn);
    MRS_HIP_TRY(hipGetLastError());
    return MRS_OK;
}

// the selection kernel of the default search setting over clouds [c0, c0 + nc) of side w; knn: see knn_at (cloud numbers count from c0)
void launch_knn_cov(mrs_gicp_batch* h, int w, int c0, int nc, int64_t longest, int k, int* d_knn, hipStream_t s)
{
    const dim3 g((unsigned)((longest + kNNThreads - 1) / kNNThreads), nc);
    const int64_t* const offs_c = h->d_offs[w] + c0;
    const int* const tb_c = h->d_tile_base[w] + c0;
    static const char* const wv_s = mrs::dev_env("MRS_KNN_WAVES");      // development aid: waves per SIMD of the selection kernel
    const int wv = wv_s ? atoi(wv_s) : 0;
    if (k <= 16 && wv == 5)
        hipLaunchKernelGGL((k_knn_cov<16, 5>), g, dim3(kNNThreads), 0, s, h->d_pts[w], offs_c, tb_c, h->d_tlo[w], h->d_thi[w], h->d_mlo[w], h->d_mhi[w], k, d_knn);
    else if (k <= 16)
        hipLaunchKernelGGL(k_knn_cov<16>, g, dim3(kNNThreads), 0, s, h->d_pts[w], offs_c, tb_c, h->d_tlo[w], h->d_thi[w], h->d_mlo[w], h->d_mhi[w], k, d_knn);
    else if (k <= 20)
        hipLaunchKernelGGL(k_knn_cov<20>, g, dim3(kNNThreads), 0, s, h->d_pts[w], offs_c, tb_c, h->d_tlo[w], h->d_thi[w], h->d_mlo[w], h->d_mhi[w], k, d_knn);
    else if (k <= 30 && wv == 4)
        hipLaunchKernelGGL((k_knn_cov<30, 4>), g, dim3(kNNThreads), 0, s, h->d_pts[w], offs_c, tb_c, h->d_tlo[w], h->d_thi[w], h->d_mlo[w], h->d_mhi[w], k, d_knn);
    else if (k <= 30)       // RING++'s k: 30 list slots = 30 KB of LDS = five workgroups per compute unit (32: four)
        hipLaunchKernelGGL(k_knn_cov<30>, g, dim3(kNNThreads), 0, s, h->d_pts[w], offs_c, tb_c, h->d_tlo[w], h->d_thi[w], h->d_mlo[w], h->d_mhi[w], k, d_knn);
    else
        hipLaunchKernelGGL(k_knn_cov<32>, g, dim3(kNNThreads), 0, s, h->d_pts[w], offs_c, tb_c, h->d_tlo[w], h->d_thi[w], h->d_mlo[w], h->d_mhi[w], k, d_knn);
}

// development aids of the selection kernel (MRS_DEV=1): MRS_KNN_DBG=1 counters, MRS_KNN_REC=0 = pass 2 walks the hierarchy again
int knn_dev_switches(hipStream_t s)
{
    if (mrs::dev_env("MRS_KNN_DBG")) {
        const int on = 1;
        MRS_HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_knn_dbg_on), &on, sizeof(on)));
    }
    const char* const e = mrs::dev_env("MRS_KNN_REC");      // read per call (the tests flip it)
    const int off = (e && atoi(e) == 0) ? 1 : 0;
    static int cur = 0;
    if (off != cur) {
        MRS_HIP_TRY(hipStreamSynchronize(s));
        MRS_HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_knn_norec), &off, sizeof(off)));
        cur = off;
    }
    return MRS_OK;
}

int knn_dbg_report(hipStream_t s)
{
    static const bool knn_dbg = mrs::dev_env("MRS_KNN_DBG") != nullptr;
    if (!knn_dbg) return MRS_OK;
    unsigned long long c[8];
    MRS_HIP_TRY(hipStreamSynchronize(s));
    MRS_HIP_TRY(hipMemcpyFromSymbol(c, HIP_SYMBOL(g_knn_dbg), sizeof(c)));
    const double w = (double)(c[6] ? c[6] : 1);
    fprintf(stderr, "[knn dbg] per query wave: pass 1 groups of 8 candidates %.1f (some lane noted one in %.1f), chain passes after the seed %.1f, pass 2 groups %.1f, "
                    "entries ranked %.1f; %llu waves; the busiest wave: %llu groups in pass 1, %llu chain passes\n", c[1] / w, c[2] / w, c[3] / w, c[4] / w, c[5] / w, c[6], c[7], c[0]);
    unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    MRS_HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_knn_dbg), z, sizeof(z)));
    unsigned long long ck[4];
    MRS_HIP_TRY(hipMemcpyFromSymbol(ck, HIP_SYMBOL(g_knn_clk), sizeof(ck)));
    fprintf(stderr, "[knn dbg] per query wave: clocks in the seed %.0f, the pass-1 walk %.0f, pass 2 %.0f; tiles visited in pass 1 %.1f\n", ck[0] / w, ck[1] / w, ck[2] / w, ck[3] / w);
    MRS_HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_knn_clk), z, sizeof(ck)));
    if (const char* path = getenv("MRS_KNN_TRACE_FILE")) {       // (start, end) of every workgroup of the launch just finished, raw uint64 pairs
        std::vector<unsigned long long> tr(2 * 65536);
        MRS_HIP_TRY(hipMemcpyFromSymbol(tr.data(), HIP_SYMBOL(g_knn_trace), tr.size() * sizeof(unsigned long long)));
        if (FILE* f = fopen(path, "wb")) { fwrite(tr.data(), sizeof(unsigned long long), tr.size(), f); fclose(f); }
        std::fill(tr.begin(), tr.end(), 0ull);
        MRS_HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_knn_trace), tr.data(), tr.size() * sizeof(unsigned long long)));
    }
    return MRS_OK;
}

}  // namespace

// Everything of set_clouds that does not look at the points: checks, (re)allocation of the side's buffers, offsets and tile bases, reset of
// the warm-start seeds.  longest / longest_tiles: the largest cloud of the side.
static int prepare_side(mrs_gicp_batch* h, int32_t which, const int64_t* h_offsets, hipStream_t s, int64_t& longest, int& longest_tiles)
{
    MRS_REQUIRE(h && h_offsets, "null pointer");
    MRS_REQUIRE(which == 0 || which == 1, "which must be 0 (source) or 1 (target)");
    MRS_REQUIRE(h_offsets[0] == 0, "offsets[0] must be 0");
    for (int i = 0; i < h->n_pairs; ++i) {
        MRS_REQUIRE(h_offsets[i + 1] > h_offsets[i], "every cloud needs at least one point");
        MRS_REQUIRE(h_offsets[i + 1] - h_offsets[i] < (1ll << 28), "cloud too large (2^28 points at most)");
    }
    MRS_HIP_TRY(hipSetDevice(h->ctx->device));
    const int64_t total = h_offsets[h->n_pairs];
    MRS_REQUIRE(total < (1ll << 31), "more than 2^31 points in one batch");
    MRS_REQUIRE(h->n_pairs < (1 << 21), "too many pairs for the 64-bit sort key");
    h->offs[which].assign(h_offsets, h_offsets + h->n_pairs + 1);
    std::vector<int> tile_base(h->n_pairs);
    int tiles = 0;
    longest_tiles = 0;
    longest = 0;
    for (int i = 0; i < h->n_pairs; ++i) {
        const int64_t n = h_offsets[i + 1] - h_offsets[i];
        const int nt = (int)((n + kTile - 1) / kTile);
        tile_base[i] = tiles;
        tiles += nt;
        longest_tiles = std::max(longest_tiles, nt);
        longest = std::max(longest, n);
    }
    h->max_tiles[which] = longest_tiles;
    h->cov_valid[which] = false;
    // a registration object is fed a new cloud per loop candidate (ICPCheck, global_manager.cpp:2018-2019): keep the device
    // buffers and only grow them (each hipFree synchronises the device, each hipMalloc costs tens of microseconds)
    if (total > h->cap_points[which] || tiles > h->cap_tiles[which] || !h->d_offs[which]) {
        free_cloud(h, which);
        const int64_t cap = total + total / 8;
        const int capt = tiles + tiles / 8 + 1;
        MRS_HIP_TRY(hipMalloc(&h->d_offs[which], (h->n_pairs + 1) * sizeof(int64_t)));
        MRS_HIP_TRY(hipMalloc(&h->d_pts[which], ((size_t)cap + 16) * sizeof(float4)));      // + 16: a mini's 16 candidates are read whole (cand_request)
        if (!h->no_cov) MRS_HIP_TRY(hipMalloc(&h->d_cov[which], (size_t)cap * kCovDoubles * sizeof(double)));
        MRS_HIP_TRY(hipMalloc(&h->d_tile_base[which], h->n_pairs * sizeof(int)));
        MRS_HIP_TRY(hipMalloc(&h->d_tlo[which], (size_t)capt * sizeof(float4)));
        MRS_HIP_TRY(hipMalloc(&h->d_thi[which], (size_t)capt * sizeof(float4)));
        MRS_HIP_TRY(hipMalloc(&h->d_mlo[which], (size_t)capt * 64 * sizeof(float4)));
        MRS_HIP_TRY(hipMalloc(&h->d_mhi[which], (size_t)capt * 64 * sizeof(float4)));
        MRS_HIP_TRY(hipMalloc(&h->d_bbox[which], (size_t)h->n_pairs * 6 * sizeof(int)));
        h->cap_points[which] = cap; h->cap_tiles[which] = capt;
        if (which == 0 && !h->no_cov) {       // correspondences, seeds and certificates belong to alignments: the RING++ front end's containers (no_cov) never read them
            if (h->d_corr) (void)hipFree(h->d_corr);
            if (h->d_seed) (void)hipFree(h->d_seed);
            h->d_corr = nullptr; h->d_seed = nullptr;
            MRS_HIP_TRY(hipMalloc(&h->d_corr, (size_t)cap * sizeof(int)));
            MRS_HIP_TRY(hipMalloc(&h->d_seed, (size_t)cap * sizeof(int)));
            if (h->cert.lb) (void)hipFree(h->cert.lb);
            if (h->cert.work) (void)hipFree(h->cert.work);
            h->cert.lb = nullptr; h->cert.work = nullptr;
            MRS_HIP_TRY(hipMalloc(&h->cert.lb, (size_t)cap * sizeof(float)));
            MRS_HIP_TRY(hipMalloc(&h->cert.work, ((size_t)cap + kCertBlock) * sizeof(int)));
            if (!h->cert.t_prev) {
                MRS_HIP_TRY(hipMalloc(&h->cert.t_prev, (size_t)h->n_pairs * 12 * sizeof(float)));

                MRS_HIP_TRY(hipMalloc(&h->cert.searched, (size_t)h->n_pairs * kStatStride * sizeof(unsigned long long)));
            }
        }
    }
    if (which == 0) h->n_seed = (size_t)total;
    if (h->d_seed) MRS_HIP_TRY(hipMemsetAsync(h->d_seed, 0xff, h->n_seed * sizeof(int), s));  // -1: no warm start across clouds
    MRS_HIP_TRY(hipMemcpyAsync(h->d_offs[which], h_offsets, (h->n_pairs + 1) * sizeof(int64_t), hipMemcpyHostToDevice, s));
    MRS_HIP_TRY(hipMemcpyAsync(h->d_tile_base[which], tile_base.data(), h->n_pairs * sizeof(int), hipMemcpyHostToDevice, s));
    // per-cloud bounding boxes armed for k_cloud_bbox (ordered ints: min = +max, max = -max)
    std::vector<int> box_init((size_t)h->n_pairs * 6);
    for (int i = 0; i < h->n_pairs; ++i)
        for (int a = 0; a < 3; ++a) { box_init[6 * i + a] = INT32_MAX; box_init[6 * i + 3 + a] = INT32_MIN; }
    MRS_HIP_TRY(hipMemcpyAsync(h->d_bbox[which], box_init.data(), box_init.size() * sizeof(int), hipMemcpyHostToDevice, s));
    MRS_HIP_TRY(hipStreamSynchronize(s));   // tile_base, box_init and h_offsets are temporaries
    return MRS_OK;
}

extern "C" {

int mrs_gicp_batch_set_search(mrs_gicp_batch* h, int32_t core)
{
    MRS_REQUIRE(h, "null handle");
    MRS_REQUIRE(core >= 0 && core <= 3, "core must be 0 .. 3");
    const int base = core == 0 ? 0 : 1;
    if ((core == 3) != (h->search_core == 1 && h->cold_core == 1)) h->cov_valid[0] = h->cov_valid[1] = false;   // the k-NN kernel changes
    h->search_core = base;
    h->use_certificates = core == 1 || core == 3;
    h->cold_core = core == 3 ? 1 : 0;
    return MRS_OK;
}

double mrs_gicp_batch_last_searched_fraction(const mrs_gicp_batch* h) { return h ? h->last_searched : 1.0; }

void mrs_gicp_default_params(mrs_gicp_params* p)
{
    if (!p) return;
    p->k_correspondences = 20;          // fast_gicp default; Mapping sets 15 (global_manager.cpp:2442)
    p->max_correspondence_distance = DBL_MAX;
    p->max_iterations = 64;
    p->rotation_epsilon = 2e-3;
    p->transformation_epsilon = 5e-4;
    p->lm_max_iterations = 10;
    p->lm_init_lambda_factor = 1e-9;
    p->convergence_factor = 10.0;       // upstream LsqRegistration::is_converged scales both deltas by 10
    p->force_iterations = 0;
    p->voxel_resolution = 0.0;          // FastVGICP(Cuda) default is 1.0; Mapping sets 0.5 (global_manager.cpp:2450)
    p->voxel_neighbors = 1;             // DIRECT1 (global_manager.cpp:2452)
}

int mrs_gicp_batch_create(mrs_ctx* ctx, int32_t n_pairs, mrs_gicp_batch** out)
{
    MRS_REQUIRE(ctx && out, "null pointer");
    MRS_REQUIRE(n_pairs > 0, "n_pairs must be positive");
    MRS_REQUIRE(n_pairs <= mrs::kMaxGridY, "at most 65535 pairs per batch (create several batches)");
    *out = nullptr;
    MRS_HIP_TRY(hipSetDevice(ctx->device));
    mrs_gicp_batch* h = new mrs_gicp_batch();
    h->ctx = ctx;
    h->n_pairs = n_pairs;
    mrs_gicp_params d;
    mrs_gicp_default_params(&d);
    *out = h;
    return mrs_gicp_batch_set_params(h, &d);
}

int mrs_gicp_batch_destroy(mrs_gicp_batch* h)
{
    if (!h) return MRS_OK;
    (void)hipSetDevice(h->ctx->device);
    free_cloud(h, 0);
    free_cloud(h, 1);
    if (h->d_state) (void)hipFree(h->d_state);
    if (h->d_partial) (void)hipFree(h->d_partial);
    if (h->d_nblocks) (void)hipFree(h->d_nblocks);
    if (h->d_nactive) (void)hipFree(h->d_nactive);
    if (h->d_corr) (void)hipFree(h->d_corr);
    if (h->d_seed) (void)hipFree(h->d_seed);
    for (void* p : {(void*)h->cert.lb, (void*)h->cert.work, (void*)h->cert.t_prev, (void*)h->cert.bcount, (void*)h->cert.searched})
        if (p) (void)hipFree(p);
    if (h->d_vkeys) (void)hipFree(h->d_vkeys);
    if (h->d_vmean) (void)hipFree(h->d_vmean);
    if (h->d_vcov) (void)hipFree(h->d_vcov);
    delete h;
    return MRS_OK;
}

int mrs_gicp_batch_set_params(mrs_gicp_batch* h, const mrs_gicp_params* p)
{
    MRS_REQUIRE(h && p, "null pointer");
    MRS_REQUIRE(p->k_correspondences >= 3 && p->k_correspondences <= 32, "k_correspondences must be in [3, 32]");
    MRS_REQUIRE(p->max_iterations > 0 && p->lm_max_iterations > 0, "iteration limits must be positive");
    MRS_REQUIRE(p->max_correspondence_distance > 0, "max_correspondence_distance must be positive");
    if (p->k_correspondences != h->prm.k) { h->cov_valid[0] = h->cov_valid[1] = false; }
    h->prm.k = p->k_correspondences;
    h->prm.max_corr2 = p->max_correspondence_distance >= 1e150 ? INFINITY
                                                               : p->max_correspondence_distance * p->max_correspondence_distance;
    h->prm.max_iter = p->max_iterations;
    h->prm.rot_eps = p->rotation_epsilon;
    h->prm.trans_eps = p->transformation_epsilon;
    h->prm.lm_max_iter = p->lm_max_iterations;
    h->prm.lm_init_factor = p->lm_init_lambda_factor;
    MRS_REQUIRE(p->convergence_factor >= 0.0, "convergence_factor must be >= 0 (0 selects upstream's 10)");
    h->prm.conv_factor = p->convergence_factor > 0.0 ? p->convergence_factor : 10.0;
    h->prm.force_iters = p->force_iterations;
    MRS_REQUIRE(p->voxel_resolution >= 0.0, "voxel_resolution must be >= 0");
    MRS_REQUIRE(p->voxel_neighbors == 1 || p->voxel_neighbors == 7 || p->voxel_neighbors == 27, "voxel_neighbors must be 1, 7 or 27");
    h->prm.cert_margin = 0.004f;
    h->prm.motion_switch = 0.02f;
    { const char* e = mrs::dev_env("MRS_MOTION_SWITCH"); if (e) h->prm.motion_switch = (float)atof(e); }
    { const char* e = mrs::dev_env("MRS_CERT_MARGIN"); if (e) h->prm.cert_margin = (float)atof(e); }
    h->prm.voxel_res = p->voxel_resolution;
    h->prm.voxel_neighbors = p->voxel_neighbors;
    return MRS_OK;
}

int mrs_gicp_batch_set_clouds(mrs_gicp_batch* h, int32_t which, const float* d_points, int32_t stride_floats,
                              const int64_t* h_offsets, mrs_stream stream)
{
    MRS_REQUIRE(h && d_points && h_offsets, "null pointer");
    MRS_REQUIRE(stride_floats >= 3, "stride_floats must be >= 3");
    hipStream_t s = (hipStream_t)stream;
    int64_t longest = 0;
    int longest_tiles = 0;
    int st = prepare_side(h, which, h_offsets, s, longest, longest_tiles);
    if (st != MRS_OK) return st;
    const int64_t total = h_offsets[h->n_pairs];

    // Morton order: per-cloud bounding box -> 64-bit keys (cloud id | Morton code) -> stable radix sort
    mrs::Scratch keys_in, keys_out, vals_in, vals_out, tmp;
    int* const bbox_p = h->d_bbox[which];
    if ((st = keys_in.alloc((size_t)total * 8, s)) != MRS_OK) return st;
    if ((st = keys_out.alloc((size_t)total * 8, s)) != MRS_OK) return st;
    if ((st = vals_in.alloc((size_t)total * 4, s)) != MRS_OK) return st;
    if ((st = vals_out.alloc((size_t)total * 4, s)) != MRS_OK) return st;
    // (the boxes were armed by prepare_side, in the same host synchronisation as the offsets: one round trip less per cloud)
    const dim3 pg((unsigned)std::min<int64_t>((longest + 255) / 256, 1024), h->n_pairs);
    hipLaunchKernelGGL(k_cloud_bbox, dim3(std::min(pg.x, 64u), pg.y), dim3(256), 0, s, d_points, stride_floats, h->d_offs[which],
                       bbox_p);
    hipLaunchKernelGGL(k_morton_keys, pg, dim3(256), 0, s, d_points, stride_floats, h->d_offs[which], bbox_p,
                       keys_in.as<unsigned long long>(), vals_in.as<int>());
    int key_bits = 42;                       // 42-bit Morton code + the bits of the cloud id: fewer radix passes than 64
    while ((1ll << (key_bits - 42)) < h->n_pairs) ++key_bits;
    size_t tmp_bytes = 0;
    MRS_HIP_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, keys_in.as<unsigned long long>(),
                                                   keys_out.as<unsigned long long>(), vals_in.as<int>(), vals_out.as<int>(),
                                                   (int)total, 0, key_bits, s));
    if ((st = tmp.alloc(tmp_bytes, s)) != MRS_OK) return st;
    MRS_HIP_TRY(hipcub::DeviceRadixSort::SortPairs(tmp.p, tmp_bytes, keys_in.as<unsigned long long>(),
                                                   keys_out.as<unsigned long long>(), vals_in.as<int>(), vals_out.as<int>(),
                                                   (int)total, 0, key_bits, s));
    hipLaunchKernelGGL(k_gather_sorted, pg, dim3(256), 0, s, d_points, stride_floats, h->d_offs[which], vals_out.as<int>(),
                       h->d_pts[which]);
    hipLaunchKernelGGL(k_boxes, dim3(longest_tiles, h->n_pairs), dim3(256), 0, s, h->d_pts[which], h->d_offs[which],
                       h->d_tile_base[which], h->d_tlo[which], h->d_thi[which], h->d_mlo[which], h->d_mhi[which]);
    MRS_HIP_TRY(hipGetLastError());
    // the octree-cell hierarchy serves the round-4 searches: correspondences search the TARGETS (which == 1); the sources need it only
    // for the round-4 k-NN kernel (setting 3); the RING++ front end (want_leaf_hier = false) not at all
    h->hier_valid[which] = false;
    if (h->want_leaf_hier && (which == 1 || (h->search_core == 1 && h->cold_core == 1))) {
        if ((st = build_leaf_hier(h, which, keys_out.as<unsigned long long>(), total, s)) != MRS_OK) return st;
        h->hier_valid[which] = true;
    }
    MRS_HIP_TRY(hipStreamSynchronize(s));
    return MRS_OK;
}

/* Pair i's cloud of side `which` := cloud h_ids[i] of side `store_which` of `store` (a batch used as a container of unique submaps):
 * sorted points, covariances, tile / mini boxes and the octree-cell hierarchy are COPIED on the device (a few MB per cloud) instead of
 * being rebuilt -- a submap that takes part in several pairs (the node checks a new scan against several stored candidates,
 * main_RING.py:81-104, global_manager.cpp:2016-2021) pays for its Morton sort and its covariances once. */
int mrs_gicp_batch_set_clouds_from(mrs_gicp_batch* h, int32_t which, mrs_gicp_batch* store, int32_t store_which, const int32_t* h_ids,
                                   mrs_stream stream)
{
    MRS_REQUIRE(h && store && h_ids, "null pointer");
    MRS_REQUIRE(h != store, "a batch cannot be its own store");
    MRS_REQUIRE(which == 0 || which == 1, "which must be 0 (source) or 1 (target)");
    MRS_REQUIRE(store_which == 0 || store_which == 1, "store_which must be 0 or 1");
    MRS_REQUIRE(h->ctx->device == store->ctx->device, "batch and store live on different devices");
    MRS_REQUIRE(store->d_pts[store_which] != nullptr, "the store holds no clouds on that side");
    MRS_REQUIRE(!h->no_cov && !store->no_cov, "covariance-free containers cannot take part");
    MRS_REQUIRE(h->prm.k == store->prm.k, "batch and store use different k_correspondences");
    const int sw = store_which, P = h->n_pairs, U = store->n_pairs;
    for (int i = 0; i < P; ++i) MRS_REQUIRE(h_ids[i] >= 0 && h_ids[i] < U, "cloud id outside the store");
    const bool need_hier = h->want_leaf_hier && (which == 1 || (h->search_core == 1 && h->cold_core == 1));
    MRS_REQUIRE(!need_hier || store->hier_valid[sw], "the store side has no octree-cell hierarchy (store the clouds as targets)");
    MRS_HIP_TRY(hipSetDevice(h->ctx->device));
    hipStream_t s = (hipStream_t)stream;
    int st;
    if (!store->cov_valid[sw] && (st = mrs_gicp_batch_compute_covariances(store, sw, nullptr, stream)) != MRS_OK) return st;
    const std::vector<int64_t>& so = store->offs[sw];
    std::vector<int64_t> offs(P + 1, 0);
    for (int i = 0; i < P; ++i) offs[i + 1] = offs[i] + (so[h_ids[i] + 1] - so[h_ids[i]]);
    int64_t longest = 0;
    int longest_tiles = 0;
    if ((st = prepare_side(h, which, offs.data(), s, longest, longest_tiles)) != MRS_OK) return st;
    // segment tables {source offset, destination offset, count}: points, 1024-point tiles, and the three levels of the hierarchy
    std::vector<int> stile(U + 1, 0);
    for (int u = 0; u < U; ++u) stile[u + 1] = stile[u] + (int)((so[u + 1] - so[u] + kTile - 1) / kTile);
    std::vector<int64_t> seg((size_t)5 * 3 * P);
    std::vector<int> lf(P + 1, 0), tf(P + 1, 0), sf(P + 1, 0);
    int dtile = 0;
    int64_t most[5] = {0, 0, 0, 0, 0};
    for (int i = 0; i < P; ++i) {
        const int u = h_ids[i];
        const int64_t cnt[5] = {so[u + 1] - so[u], stile[u + 1] - stile[u],
                                need_hier ? store->h_leaf_first[sw][u + 1] - store->h_leaf_first[sw][u] : 0,
                                need_hier ? store->h_tile_first[sw][u + 1] - store->h_tile_first[sw][u] : 0,
                                need_hier ? store->h_super_first[sw][u + 1] - store->h_super_first[sw][u] : 0};
        const int64_t from[5] = {so[u], stile[u], need_hier ? store->h_leaf_first[sw][u] : 0, need_hier ? store->h_tile_first[sw][u] : 0,
                                 need_hier ? store->h_super_first[sw][u] : 0};
        const int64_t to[5] = {offs[i], dtile, lf[i], tf[i], sf[i]};
        for (int a = 0; a < 5; ++a) {
            seg[((size_t)a * P + i) * 3] = from[a]; seg[((size_t)a * P + i) * 3 + 1] = to[a]; seg[((size_t)a * P + i) * 3 + 2] = cnt[a];
            most[a] = std::max(most[a], cnt[a]);
        }
        dtile += (int)cnt[1];
        lf[i + 1] = lf[i] + (int)cnt[2]; tf[i + 1] = tf[i] + (int)cnt[3]; sf[i + 1] = sf[i] + (int)cnt[4];
    }
    mrs::Scratch dseg;
    if ((st = dseg.alloc(seg.size() * sizeof(int64_t), s)) != MRS_OK) return st;
    MRS_HIP_TRY(hipMemcpyAsync(dseg.p, seg.data(), seg.size() * sizeof(int64_t), hipMemcpyHostToDevice, s));
    auto table = [&](int a) { return dseg.as<int64_t>() + (size_t)a * P * 3; };
    auto copy4 = [&](const float4* src, float4* dst, int a, int64_t scale) {
        const unsigned bx = (unsigned)std::max<int64_t>(1, std::min<int64_t>((most[a] * scale + 1023) / 1024, 256));
        hipLaunchKernelGGL(k_copy_segments<float4>, dim3(bx, P), dim3(256), 0, s, src, dst, (const int64_t*)table(a), scale);
    };
    copy4(store->d_pts[sw], h->d_pts[which], 0, 1);
    {   // the normals: 3 doubles per point
        const unsigned bx = (unsigned)std::max<int64_t>(1, std::min<int64_t>((most[0] * kCovDoubles + 1023) / 1024, 256));
        hipLaunchKernelGGL(k_copy_segments<double>, dim3(bx, P), dim3(256), 0, s, (const double*)store->d_cov[sw], h->d_cov[which], (const int64_t*)table(0),
                           (int64_t)kCovDoubles);
    }
    copy4(store->d_tlo[sw], h->d_tlo[which], 1, 1);
    copy4(store->d_thi[sw], h->d_thi[which], 1, 1);
    copy4(store->d_mlo[sw], h->d_mlo[which], 1, 64);
    copy4(store->d_mhi[sw], h->d_mhi[which], 1, 64);
    {   // the clouds' bounding boxes (the Morton grids): 6 ints per cloud
        std::vector<int64_t> bseg((size_t)3 * P);
        for (int i = 0; i < P; ++i) { bseg[3 * i] = 6 * (int64_t)h_ids[i]; bseg[3 * i + 1] = 6 * (int64_t)i; bseg[3 * i + 2] = 6; }
        mrs::Scratch dbs;
        if ((st = dbs.alloc(bseg.size() * sizeof(int64_t), s)) != MRS_OK) return st;
        MRS_HIP_TRY(hipMemcpyAsync(dbs.p, bseg.data(), bseg.size() * sizeof(int64_t), hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(k_copy_segments<int>, dim3(1, P), dim3(64), 0, s, (const int*)store->d_bbox[sw], h->d_bbox[which], (const int64_t*)dbs.p, (int64_t)1);
        MRS_HIP_TRY(hipStreamSynchronize(s));     // bseg / dbs are temporaries
    }
    h->hier_valid[which] = false;
    if (need_hier) {
        if (!h->d_leaf_first[which]) {
            MRS_HIP_TRY(hipMalloc(&h->d_leaf_first[which], (size_t)(P + 1) * sizeof(int)));
            MRS_HIP_TRY(hipMalloc(&h->d_tile_first[which], (size_t)(P + 1) * sizeof(int)));
            MRS_HIP_TRY(hipMalloc(&h->d_super_first[which], (size_t)(P + 1) * sizeof(int)));
        }
        auto grow = [](float4*& a, float4*& b, int& cap, int need) -> hipError_t {
            if (need <= cap && a) return hipSuccess;
            if (a) (void)hipFree(a);
            if (b) (void)hipFree(b);
            a = b = nullptr;
            cap = need + need / 8 + 1;
            hipError_t e = hipMalloc(&a, (size_t)cap * sizeof(float4));
            return e != hipSuccess ? e : hipMalloc(&b, (size_t)cap * sizeof(float4));
        };
        MRS_HIP_TRY(grow(h->d_llo[which], h->d_lhi[which], h->cap_leaves[which], lf[P]));
        MRS_HIP_TRY(grow(h->d_t2lo[which], h->d_t2hi[which], h->cap_tiles2[which], tf[P]));
        MRS_HIP_TRY(grow(h->d_slo[which], h->d_shi[which], h->cap_supers[which], sf[P]));
        MRS_HIP_TRY(hipMemcpyAsync(h->d_leaf_first[which], lf.data(), lf.size() * sizeof(int), hipMemcpyHostToDevice, s));
        MRS_HIP_TRY(hipMemcpyAsync(h->d_tile_first[which], tf.data(), tf.size() * sizeof(int), hipMemcpyHostToDevice, s));
        MRS_HIP_TRY(hipMemcpyAsync(h->d_super_first[which], sf.data(), sf.size() * sizeof(int), hipMemcpyHostToDevice, s));
        copy4(store->d_llo[sw], h->d_llo[which], 2, 1);
        copy4(store->d_lhi[sw], h->d_lhi[which], 2, 1);
        copy4(store->d_t2lo[sw], h->d_t2lo[which], 3, 1);
        copy4(store->d_t2hi[sw], h->d_t2hi[which], 3, 1);
        copy4(store->d_slo[sw], h->d_slo[which], 4, 1);
        copy4(store->d_shi[sw], h->d_shi[which], 4, 1);
        h->n_leaves[which] = lf[P];
        h->h_leaf_first[which] = lf; h->h_tile_first[which] = tf; h->h_super_first[which] = sf;
        h->hier_valid[which] = true;
    }
    MRS_HIP_TRY(hipGetLastError());
    MRS_HIP_TRY(hipStreamSynchronize(s));         // the tables are temporaries
    h->cov_valid[which] = true;
    return MRS_OK;
}

/* host-array form of set_clouds (what a pcl::PointCloud / numpy caller holds): staged through a scratch buffer */
int mrs_gicp_batch_set_clouds_host(mrs_gicp_batch* h, int32_t which, const float* h_points, int32_t stride_floats, const int64_t* h_offsets)
{
    MRS_REQUIRE(h && h_points && h_offsets, "null pointer");
    MRS_REQUIRE(stride_floats >= 3, "stride_floats must be >= 3");
    MRS_REQUIRE(h_offsets[0] == 0 && h_offsets[h->n_pairs] > 0, "offsets must start at 0 and hold points");
    MRS_HIP_TRY(hipSetDevice(h->ctx->device));
    const size_t bytes = (size_t)h_offsets[h->n_pairs] * stride_floats * sizeof(float);
    mrs::Scratch stage;
    int st = stage.alloc(bytes, nullptr);
    if (st != MRS_OK) return st;
    MRS_HIP_TRY(hipMemcpy(stage.p, h_points, bytes, hipMemcpyHostToDevice));
    st = mrs_gicp_batch_set_clouds(h, which, stage.as<float>(), stride_floats, h_offsets, nullptr);
    if (st != MRS_OK) return st;
    MRS_HIP_TRY(hipStreamSynchronize(nullptr));     // the staging buffer goes back to the cache only after set_clouds has read it
    return MRS_OK;
}

int mrs_gicp_batch_compute_covariances(mrs_gicp_batch* h, int32_t which, int32_t* d_knn_out, mrs_stream stream)
{
    MRS_REQUIRE(h, "null handle");
    MRS_REQUIRE(which == 0 || which == 1, "which must be 0 or 1");
    MRS_REQUIRE(h->d_pts[which] != nullptr, "set_clouds first");
    MRS_HIP_TRY(hipSetDevice(h->ctx->device));
    hipStream_t s = (hipStream_t)stream;
    int64_t longest = 0;
    for (int i = 0; i < h->n_pairs; ++i) longest = std::max(longest, h->offs[which][i + 1] - h->offs[which][i]);
    const dim3 grid((unsigned)((longest + kNNThreads - 1) / kNNThreads), h->n_pairs);
    const int k = h->prm.k;
    if (h->search_core == 1 && h->cold_core == 1) {     // setting 3: the round-4 k-NN kernel (slower than the round-3 one on the bench's scans)
        MRS_REQUIRE(h->hier_valid[which], "search setting 3 was selected after set_clouds: set the clouds again");
        mrs::Scratch knn;
        int st = knn.alloc(knn_ints(h->offs[which][h->n_pairs], h->n_pairs, k) * sizeof(int), s);
        if (st != MRS_OK) return st;
        if ((st = launch_knn_select(h, which, k, knn.as<int>(), s)) != MRS_OK) return st;
        hipLaunchKernelGGL(k_cov_from_knn, dim3((unsigned)((longest + 255) / 256), h->n_pairs), dim3(256), 0, s, (const float4*)h->d_pts[which],
                           (const int64_t*)h->d_offs[which], k, (const int*)knn.as<int>(), h->d_cov[which], d_knn_out);
        MRS_HIP_TRY(hipGetLastError());
        h->cov_valid[which] = true;
        if (which == 1) h->vox_res_built = 0.0;
        return MRS_OK;
    }
    int st = knn_dev_switches(s);
    if (st != MRS_OK) return st;
    // the neighbour indices pass from the selection to the covariance tail through scratch memory: clouds are processed in chunks so that
    // it stays below ~512 MB (256 clouds x 120 k points x k = 15 would be 1.8 GB held by the scratch cache for the life of the process)
    const int64_t per_cloud = (int64_t)(knn_ints(longest, 1, k) * sizeof(int));
    const int chunk = (int)std::max<int64_t>(1, std::min<int64_t>(h->n_pairs, (512ll << 20) / per_cloud));
    mrs::Scratch knn;
    if ((st = knn.alloc((size_t)chunk * per_cloud, s)) != MRS_OK) return st;
    for (int c0 = 0; c0 < h->n_pairs; c0 += chunk) {
        const int nc = std::min(chunk, h->n_pairs - c0);
        // the kernels index knn by GLOBAL point number (knn_at: (offs[c] + 64 c) k with c counted from the chunk's first cloud): shift the
        // chunk's buffer so that the chunk's first point lands on its start
        int* const kn = knn.as<int>() - (size_t)h->offs[which][c0] * k;
        const int64_t* const offs_c = h->d_offs[which] + c0;
        launch_knn_cov(h, which, c0, nc, longest, k, kn, s);
        hipLaunchKernelGGL(k_cov_from_knn, dim3((unsigned)((longest + 255) / 256), nc), dim3(256), 0, s, (const float4*)h->d_pts[which],
                           offs_c, k, (const int*)kn, h->d_cov[which], d_knn_out);
    }
    MRS_HIP_TRY(hipGetLastError());
    if ((st = knn_dbg_report(s)) != MRS_OK) return st;
    h->cov_valid[which] = true;
    if (which == 1) h->vox_res_built = 0.0;
    return MRS_OK;
}

int mrs_gicp_batch_get_covariances(mrs_gicp_batch* h, int32_t which, double* h_cov6)
{
    MRS_REQUIRE(h && h_cov6, "null pointer");
    MRS_REQUIRE(which == 0 || which == 1, "which must be 0 or 1");
    MRS_REQUIRE(h->cov_valid[which], "covariances not computed");
    MRS_HIP_TRY(hipSetDevice(h->ctx->device));
    MRS_HIP_TRY(hipDeviceSynchronize());
    const size_t total = (size_t)h->offs[which][h->n_pairs];
    std::vector<double> sorted(total * kCovDoubles);
    std::vector<float4> pts(total);
    MRS_HIP_TRY(hipMemcpy(sorted.data(), h->d_cov[which], total * kCovDoubles * sizeof(double), hipMemcpyDeviceToHost));
    MRS_HIP_TRY(hipMemcpy(pts.data(), h->d_pts[which], total * sizeof(float4), hipMemcpyDeviceToHost));
    for (int c = 0; c < h->n_pairs; ++c) {  // the library stores clouds in Morton order; .w = original index
        const int64_t o = h->offs[which][c];
        for (int64_t i = o; i < h->offs[which][c + 1]; ++i) {
            int orig;
            memcpy(&orig, &pts[i].w, sizeof(int));
            double c6[6];
            cov6_from_normal(&sorted[(size_t)i * kCovDoubles], c6);      // the doubles the device kernels work with
            memcpy(h_cov6 + (size_t)(o + orig) * 6, c6, 6 * sizeof(double));
        }
    }
    return MRS_OK;
}

static int ensure_state(mrs_gicp_batch* h)
{
    int64_t longest = 0;
    for (int i = 0; i < h->n_pairs; ++i) longest = std::max(longest, h->offs[0][i + 1] - h->offs[0][i]);
    // workgroups of the reduction kernels (k_linearize, k_linearize_voxel, k_fitness) per pair: every workgroup walks kLinChunks blocks of 1024
    // points before its 28 wave reductions + LDS round (one per 1024 points, the ds_bpermute butterflies were 38 % of the LDS pipe's time and
    // a third of the kernel's instructions: profiles/r04_pmc.json).  The partial sums are added in workgroup order (k_lm_update): a fixed order
    // for a given cloud size and batch size, the same for every search setting.
    // A small batch (the node's one pair at a time: 39 blocks of 1024 points) cannot afford that: 10 workgroups on 256 compute units; below four
    // workgroups per compute unit every block of 1024 points gets its own workgroup (k_linearize 25.7 -> 11 us per launch for one pair of 39 k points).
    static const char* const ch_s = mrs::dev_env("MRS_LIN_CHUNKS");
    const int cus = h->ctx->num_cu > 0 ? h->ctx->num_cu : 256;
    const int chunks = ch_s ? std::max(1, atoi(ch_s)) : ((int64_t)h->n_pairs * blocks_for_points((int)longest) >= (int64_t)4 * kLinChunks * cus ? kLinChunks : 1);
    const int mb = (blocks_for_points((int)longest) + chunks - 1) / chunks;
    h->longest_src = (int)longest;
    if (!h->d_state) {
        MRS_HIP_TRY(hipMalloc(&h->d_state, h->n_pairs * sizeof(LmState)));
        MRS_HIP_TRY(hipMalloc(&h->d_nblocks, h->n_pairs * sizeof(int)));
        MRS_HIP_TRY(hipMalloc(&h->d_nactive, kLmWindowMax * 4 * sizeof(int)));
    }
    if (mb > h->cap_blocks) {
        if (h->d_partial) (void)hipFree(h->d_partial);
        MRS_HIP_TRY(hipMalloc(&h->d_partial, (size_t)h->n_pairs * mb * kTerms * sizeof(double)));
        h->cap_blocks = mb;
    }
    h->max_blocks = mb;      // grid AND row stride of d_partial: a function of the clouds at hand only (the block -> points mapping, hence the order
                             // of the sums, must not depend on what the object held before)
    const int cb = (int)((longest + kCertBlock - 1) / kCertBlock);
    if (cb > h->cert.nb || !h->cert.bcount) {
        if (h->cert.bcount) (void)hipFree(h->cert.bcount);
        h->cert.bcount = nullptr;
        MRS_HIP_TRY(hipMalloc(&h->cert.bcount, (size_t)h->n_pairs * cb * sizeof(int)));
        h->cert.nb = cb;
    }
    std::vector<int> nb(h->n_pairs);
    for (int i = 0; i < h->n_pairs; ++i) nb[i] = std::min(blocks_for_points((int)(h->offs[0][i + 1] - h->offs[0][i])), h->max_blocks);
    MRS_HIP_TRY(hipMemcpy(h->d_nblocks, nb.data(), nb.size() * sizeof(int), hipMemcpyHostToDevice));
    return MRS_OK;
}


static int build_voxel_map(mrs_gicp_batch* h, hipStream_t s)
{
    if (h->vox_res_built == h->prm.voxel_res && h->d_vkeys) return MRS_OK;
    MRS_REQUIRE(h->n_pairs < 65536, "VGICP supports at most 65535 pairs per batch");
    const size_t total = (size_t)h->offs[1][h->n_pairs];
    mrs::Scratch keys_in, keys_out, vals_in, vals_out, head, slot, tmp;
    int st;
    if ((st = keys_in.alloc(total * 8, s)) != MRS_OK) return st;
    if ((st = keys_out.alloc(total * 8, s)) != MRS_OK) return st;
    if ((st = vals_in.alloc(total * 4, s)) != MRS_OK) return st;
    if ((st = vals_out.alloc(total * 4, s)) != MRS_OK) return st;
    if ((st = head.alloc(total * 4, s)) != MRS_OK) return st;
    if ((st = slot.alloc(total * 4, s)) != MRS_OK) return st;
    int64_t longest = 0;
    for (int i = 0; i < h->n_pairs; ++i) longest = std::max(longest, h->offs[1][i + 1] - h->offs[1][i]);
    hipLaunchKernelGGL(k_vox_keys, dim3((unsigned)std::min<int64_t>((longest + 255) / 256, 1024), h->n_pairs), dim3(256), 0, s,
                       h->d_pts[1], h->d_offs[1], h->prm.voxel_res, keys_in.as<unsigned long long>(), vals_in.as<int>());
    size_t b1 = 0, b2 = 0;
    MRS_HIP_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, b1, keys_in.as<unsigned long long>(), keys_out.as<unsigned long long>(),
                                                   vals_in.as<int>(), vals_out.as<int>(), (int)total, 0, 64, s));
    MRS_HIP_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, b2, head.as<int>(), slot.as<int>(), (int)total, s));
    if ((st = tmp.alloc(std::max(b1, b2), s)) != MRS_OK) return st;
    MRS_HIP_TRY(hipcub::DeviceRadixSort::SortPairs(tmp.p, b1, keys_in.as<unsigned long long>(), keys_out.as<unsigned long long>(),
                                                   vals_in.as<int>(), vals_out.as<int>(), (int)total, 0, 64, s));
    const int fb = (int)std::min<size_t>((total + 255) / 256, 4096);
    hipLaunchKernelGGL(k_vox_heads, dim3(fb), dim3(256), 0, s, keys_out.as<unsigned long long>(), total, head.as<int>());
    MRS_HIP_TRY(hipcub::DeviceScan::ExclusiveSum(tmp.p, b2, head.as<int>(), slot.as<int>(), (int)total, s));
    int last_head = 0, last_slot = 0;
    MRS_HIP_TRY(hipMemcpyAsync(&last_head, head.as<int>() + (total - 1), 4, hipMemcpyDeviceToHost, s));
    MRS_HIP_TRY(hipMemcpyAsync(&last_slot, slot.as<int>() + (total - 1), 4, hipMemcpyDeviceToHost, s));
    MRS_HIP_TRY(hipStreamSynchronize(s));
    h->n_voxels = last_head + last_slot;
    if (h->d_vkeys) (void)hipFree(h->d_vkeys);
    if (h->d_vmean) (void)hipFree(h->d_vmean);
    if (h->d_vcov) (void)hipFree(h->d_vcov);
    h->d_vkeys = nullptr; h->d_vmean = nullptr; h->d_vcov = nullptr;
    MRS_HIP_TRY(hipMalloc(&h->d_vkeys, (size_t)h->n_voxels * 8));
    MRS_HIP_TRY(hipMalloc(&h->d_vmean, (size_t)h->n_voxels * sizeof(float4)));
    MRS_HIP_TRY(hipMalloc(&h->d_vcov, (size_t)h->n_voxels * 6 * sizeof(double)));
    hipLaunchKernelGGL(k_vox_build, dim3(fb), dim3(256), 0, s, h->d_pts[1], h->d_cov[1], keys_out.as<unsigned long long>(),
                       vals_out.as<int>(), head.as<int>(), slot.as<int>(), total, h->d_vkeys, h->d_vmean, h->d_vcov);
    MRS_HIP_TRY(hipGetLastError());
    MRS_HIP_TRY(hipStreamSynchronize(s));
    h->vox_res_built = h->prm.voxel_res;
    return MRS_OK;
}

int mrs_gicp_batch_align(mrs_gicp_batch* h, const double* h_guess, double* h_final, int32_t* h_converged,
                         int32_t* h_iterations, double* h_hessian, mrs_stream stream)
{
    MRS_REQUIRE(h && h_final, "null pointer");
    MRS_REQUIRE(h->d_pts[0] && h->d_pts[1], "set source and target clouds first");
    MRS_HIP_TRY(hipSetDevice(h->ctx->device));
    hipStream_t s = (hipStream_t)stream;
    int st;
    // device-side state first: its (blocking) upload of the per-pair block counts would otherwise wait for the covariance kernels enqueued below
    // and keep the host from enqueuing the first ticks behind them
    st = ensure_state(h);
    if (st != MRS_OK) return st;
    const bool small = h->n_pairs <= kLmWindowPairs;
    // a second stream, two events and a pinned buffer from the context's pool for the duration of this call (small batches only)
    struct Side {
        mrs_ctx* ctx; mrs::SideSlot sl;
        ~Side() { mrs::side_release(ctx, sl); }
    } side{h->ctx, {}};
    if (small && (st = mrs::side_acquire(h->ctx, &side.sl)) != MRS_OK) return st;
    if (small && !h->cov_valid[0] && !h->cov_valid[1] && !mrs::dev_env("MRS_GICP_SERIAL_COV")) {
        // both clouds are new (every registration of the nodes): a cloud of 30-40 k points fills 150 of the 256 compute units with one wave per
        // SIMD, so the two k-NN + covariance passes run side by side on two streams instead of back to back
        MRS_HIP_TRY(hipEventRecord(side.sl.fork, s));
        MRS_HIP_TRY(hipStreamWaitEvent(side.sl.stream, side.sl.fork, 0));
        st = mrs_gicp_batch_compute_covariances(h, 1, nullptr, (mrs_stream)side.sl.stream);
        if (st != MRS_OK) return st;
        MRS_HIP_TRY(hipEventRecord(side.sl.join, side.sl.stream));
        st = mrs_gicp_batch_compute_covariances(h, 0, nullptr, stream);
        if (st != MRS_OK) return st;
        MRS_HIP_TRY(hipStreamWaitEvent(s, side.sl.join, 0));
    }
    for (int w = 0; w < 2; ++w)
        if (!h->cov_valid[w]) { st = mrs_gicp_batch_compute_covariances(h, w, nullptr, stream); if (st != MRS_OK) return st; }
    if (h->prm.voxel_res > 0.0) {
        st = build_voxel_map(h, s);
        if (st != MRS_OK) return st;
    }
    std::vector<LmState> init(h->n_pairs);
    for (int p = 0; p < h->n_pairs; ++p) {
        LmState& S = init[p];
        memset(&S, 0, sizeof(S));
        for (int i = 0; i < 16; ++i) {
            const double g = h_guess ? h_guess[(size_t)p * 16 + i] : (i % 5 == 0 ? 1.0 : 0.0);
            S.x[i] = S.xi[i] = (double)(float)g;  // the reference passes an Eigen::Matrix4f guess
        }
        S.lambda = -1.0; S.nu = 2.0; S.active = 1;
    }
    MRS_HIP_TRY(hipMemcpyAsync(h->d_state, init.data(), init.size() * sizeof(LmState), hipMemcpyHostToDevice, s));
    if (h->cert.searched) MRS_HIP_TRY(hipMemsetAsync(h->cert.searched, 0, (size_t)h->n_pairs * kStatStride * sizeof(unsigned long long), s));
    const dim3 grid(h->max_blocks, h->n_pairs);
    const int limit = h->prm.force_iters > 0 ? h->prm.force_iters : h->prm.max_iter;
    const long max_ticks = (long)limit * (h->prm.lm_max_iter + 1) + 1;
    long ticks = 0, nn_ticks = 0;
    int next[3] = {h->n_pairs, 0, h->n_pairs};   // pairs to linearise (phase 0), pairs in an LM trial (phase 1), pairs of [0] that moved far
    int window = kLmWindow;
    if (const char* v = mrs::dev_env("MRS_GICP_WINDOW")) window = std::max(0, std::min(kLmWindowMax, atoi(v)));
    if (small && window > 1 && h->prm.voxel_res <= 0.0) {
        static_assert(kLmWindowMax * 4 <= mrs::kSidePinnedInts, "the slot's pinned buffer holds a window's counters");
        int* const h_win = side.sl.pinned;
        const bool alternate = h->n_pairs == 1 && !mrs::dev_env("MRS_GICP_FULL_TICKS");
        const long max_ticks_w = alternate ? 2 * max_ticks : max_ticks;      // a sat-out tick does no work: the bound counts work ticks
        while (next[0] + next[1] > 0 && ticks < max_ticks_w) {
            MRS_HIP_TRY(hipMemsetAsync(h->d_nactive, 0, (size_t)window * 4 * sizeof(int), s));
            for (int t = 0; t < window; ++t) {
                // ONE pair alternates between a linearisation and (at least) one LM trial: every second tick is enqueued without its four
                // search kernels (an empty launch still costs ~5 us on the stream: 20 us per trial tick, ~160 us per registration).  If the
                // pair needs a linearisation on such a tick after all (it only does after a rejected trial shifted the rhythm) it sits the
                // tick out -- k_linearize / k_lm_update leave it alone -- and takes the next one: same transitions, same bits.
                const int trial_only = (alternate && ((ticks + t) & 1)) ? 1 : 0;
                h->big_movers = 1;                      // the broad search gates itself on the pair's motion (k_nn_scan: gate)
                if (!trial_only && (st = nn_pass(h, (ticks == 0 && t == 0) ? 0 : 1, s)) != MRS_OK) return st;
                launch_linearize(grid, s, h->d_pts[0], h->d_offs[0], h->d_cov[0],
                                   h->d_pts[1], h->d_offs[1], h->d_cov[1], h->d_state, h->d_corr, h->d_partial, h->max_blocks, trial_only);
                hipLaunchKernelGGL(k_lm_update, dim3(h->n_pairs), dim3(kLmThreads), 0, s, h->d_state, h->d_partial, h->d_nblocks,
                                   h->max_blocks, h->prm, h->d_nactive + 4 * t, trial_only);
            }
            MRS_HIP_TRY(hipGetLastError());
            MRS_HIP_TRY(hipMemcpyAsync(h_win, h->d_nactive, (size_t)window * 4 * sizeof(int), hipMemcpyDeviceToHost, s));
            MRS_HIP_TRY(hipStreamSynchronize(s));
            for (int t = 0; t < window; ++t) nn_ticks += h_win[4 * t + 3];
            for (int i = 0; i < 3; ++i) next[i] = h_win[4 * (window - 1) + i];
            ticks += window;
        }
    }
    while (next[0] + next[1] > 0 && ticks < max_ticks) {
        MRS_HIP_TRY(hipMemsetAsync(h->d_nactive, 0, 4 * sizeof(int), s));
        if (h->prm.voxel_res > 0.0) {
            hipLaunchKernelGGL(k_linearize_voxel, grid, dim3(kNNThreads), 0, s, h->d_pts[0], h->d_offs[0], h->d_cov[0], h->d_vkeys,
                               h->d_vmean, h->d_vcov, h->n_voxels, h->d_state, h->prm, h->d_partial, h->max_blocks);
        } else {
            if (next[0] > 0) {   // only linearisations search; LM trials score the cached correspondences
                h->big_movers = next[2];
                if ((st = nn_pass(h, nn_ticks == 0 ? 0 : 1, s)) != MRS_OK) return st;
            }
            launch_linearize(grid, s, h->d_pts[0], h->d_offs[0], h->d_cov[0],
                               h->d_pts[1], h->d_offs[1], h->d_cov[1], h->d_state, h->d_corr, h->d_partial, h->max_blocks, 0);
        }
        if (next[0] > 0) ++nn_ticks;
        hipLaunchKernelGGL(k_lm_update, dim3(h->n_pairs), dim3(kLmThreads), 0, s, h->d_state, h->d_partial, h->d_nblocks,
                           h->max_blocks, h->prm, h->d_nactive, 0);
        MRS_HIP_TRY(hipGetLastError());
        MRS_HIP_TRY(hipMemcpyAsync(next, h->d_nactive, 3 * sizeof(int), hipMemcpyDeviceToHost, s));
        MRS_HIP_TRY(hipStreamSynchronize(s));
        ++ticks;
    }
    h->last_nn_passes = (double)nn_ticks;
    h->last_searched = 1.0;
    if (h->search_core == 1 && h->cert.searched && h->prm.voxel_res <= 0.0 && nn_ticks > 0) {
        std::vector<unsigned long long> stat((size_t)h->n_pairs * kStatStride);
        MRS_HIP_TRY(hipMemcpy(stat.data(), h->cert.searched, stat.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        unsigned long long q[2] = {0, 0};
        for (int p = 0; p < h->n_pairs; ++p) { q[0] += stat[(size_t)p * kStatStride]; q[1] += stat[(size_t)p * kStatStride + 1]; }
        if (q[1]) h->last_searched = (double)q[0] / (double)q[1];
    }
    MRS_HIP_TRY(hipMemcpy(init.data(), h->d_state, init.size() * sizeof(LmState), hipMemcpyDeviceToHost));
    for (int p = 0; p < h->n_pairs; ++p) {
        const LmState& S = init[p];
        for (int i = 0; i < 16; ++i) h_final[(size_t)p * 16 + i] = (double)(float)S.x[i];  // final_transformation_ is float
        if (h_converged) h_converged[p] = S.converged;
        if (h_iterations) h_iterations[p] = S.outer;
        if (h_hessian) memcpy(h_hessian + (size_t)p * 36, S.final_H, sizeof(S.final_H));
    }
    return MRS_OK;
}

int mrs_gicp_batch_linearize(mrs_gicp_batch* h, const double* h_poses, double* h_H, double* h_b, double* h_err,
                             int32_t* d_corr, mrs_stream stream)
{
    MRS_REQUIRE(h && h_poses && h_H && h_b && h_err, "null pointer");
    MRS_REQUIRE(h->d_pts[0] && h->d_pts[1], "set source and target clouds first");
    MRS_HIP_TRY(hipSetDevice(h->ctx->device));
    hipStream_t s = (hipStream_t)stream;
    int st;
    for (int w = 0; w < 2; ++w)
        if (!h->cov_valid[w]) { st = mrs_gicp_batch_compute_covariances(h, w, nullptr, stream); if (st != MRS_OK) return st; }
    st = ensure_state(h);
    if (st != MRS_OK) return st;
    std::vector<LmState> init(h->n_pairs);
    for (int p = 0; p < h->n_pairs; ++p) {
        memset(&init[p], 0, sizeof(LmState));
        for (int i = 0; i < 16; ++i) init[p].x[i] = init[p].xi[i] = h_poses[(size_t)p * 16 + i];
        init[p].active = 1;
    }
    MRS_HIP_TRY(hipMemcpyAsync(h->d_state, init.data(), init.size() * sizeof(LmState), hipMemcpyHostToDevice, s));
    if (h->prm.voxel_res > 0.0) {
        st = build_voxel_map(h, s);
        if (st != MRS_OK) return st;
        MRS_REQUIRE(d_corr == nullptr, "per-point correspondences are not defined for the voxelised variant");
        hipLaunchKernelGGL(k_linearize_voxel, dim3(h->max_blocks, h->n_pairs), dim3(kNNThreads), 0, s, h->d_pts[0], h->d_offs[0],
                           h->d_cov[0], h->d_vkeys, h->d_vmean, h->d_vcov, h->n_voxels, h->d_state, h->prm, h->d_partial,
                           h->max_blocks);
    } else {
        if ((st = nn_pass(h, 2, s)) != MRS_OK) return st;
        launch_linearize(dim3(h->max_blocks, h->n_pairs), s, h->d_pts[0], h->d_offs[0],
                           h->d_cov[0], h->d_pts[1], h->d_offs[1], h->d_cov[1], h->d_state, h->d_corr, h->d_partial,
                           h->max_blocks, 0);
    }
    if (d_corr)
        hipLaunchKernelGGL(k_corr_to_original, dim3(64, h->n_pairs), dim3(256), 0, s, h->d_pts[0], h->d_offs[0], h->d_pts[1],
                           h->d_offs[1], h->d_corr, d_corr);
    MRS_HIP_TRY(hipGetLastError());
    std::vector<double> part((size_t)h->n_pairs * h->max_blocks * kTerms);
    MRS_HIP_TRY(hipMemcpyAsync(part.data(), h->d_partial, part.size() * sizeof(double), hipMemcpyDeviceToHost, s));
    MRS_HIP_TRY(hipStreamSynchronize(s));
    for (int p = 0; p < h->n_pairs; ++p) {
        double sum[kTerms] = {0};
        const int nb = std::min(blocks_for_points((int)(h->offs[0][p + 1] - h->offs[0][p])), h->max_blocks);
        for (int b = 0; b < nb; ++b)
            for (int t = 0; t < kTerms; ++t) sum[t] += part[((size_t)p * h->max_blocks + b) * kTerms + t];
        int t = 0;
        for (int r = 0; r < 6; ++r)
            for (int c = r; c < 6; ++c) { h_H[(size_t)p * 36 + 6 * r + c] = sum[t]; h_H[(size_t)p * 36 + 6 * c + r] = sum[t]; ++t; }
        for (int r = 0; r < 6; ++r) h_b[(size_t)p * 6 + r] = sum[21 + r];
        h_err[p] = sum[27];
    }
    return MRS_OK;
}

/* Measurement hook (bench.py's GICP rooflines): every kernel of one outer iteration launched ALONE between HIP events on `stream`, at
 * the given poses and with the batch's clouds / covariances, `reps` times each (the average goes to out_ms):
 *   [0] k_linearize, all 28 sums (phase 0)            [1] k_linearize, error only (an LM trial)
 *   [2] round-3 search of every point, warm           [3] k_nn_certify at an unchanged pose (everything certified)
 *   [4] round-4 search of every point, warm           [5] k-NN selection of the sources (k_knn_cov)
 *   [6] k_cov_from_knn of the sources                 [7] k_nn_certify + work-list search at a pose moved by 1 mm along x
 * out_counts: [0] source points, [1] correspondences (d^2 < max_corr^2) at the poses, [2] queries on the work lists of [7]. */
int mrs_gicp_batch_profile(mrs_gicp_batch* h, const double* h_poses, int32_t reps, float* out_ms, int64_t* out_counts, mrs_stream stream)
{
    MRS_REQUIRE(h && h_poses && out_ms && out_counts, "null pointer");
    MRS_REQUIRE(h->d_pts[0] && h->d_pts[1], "set source and target clouds first");
    MRS_REQUIRE(h->prm.voxel_res <= 0.0, "GICP only");
    MRS_REQUIRE(reps >= 1, "reps must be >= 1");
    MRS_HIP_TRY(hipSetDevice(h->ctx->device));
    hipStream_t s = (hipStream_t)stream;
    int st;
    for (int w = 0; w < 2; ++w)
        if (!h->cov_valid[w]) { st = mrs_gicp_batch_compute_covariances(h, w, nullptr, stream); if (st != MRS_OK) return st; }
    if ((st = ensure_state(h)) != MRS_OK) return st;
    const int P = h->n_pairs;
    std::vector<LmState> init(P);
    auto upload = [&](int phase, double dx) -> int {
        for (int p = 0; p < P; ++p) {
            memset(&init[p], 0, sizeof(LmState));
            for (int i = 0; i < 16; ++i) init[p].x[i] = init[p].xi[i] = init[p].delta[i] = h_poses[(size_t)p * 16 + i];
            for (int i = 0; i < 16; ++i) init[p].delta[i] = (i % 5 == 0) ? 1.0 : 0.0;      // last step: none
            init[p].x[3] += dx; init[p].xi[3] += dx;
            init[p].active = 1; init[p].phase = phase;
        }
        MRS_HIP_TRY(hipMemcpyAsync(h->d_state, init.data(), init.size() * sizeof(LmState), hipMemcpyHostToDevice, s));
        MRS_HIP_TRY(hipStreamSynchronize(s));
        return MRS_OK;
    };
    // everything this function borrows goes back on EVERY way out (the MRS_HIP_TRY returns included): the two events and the search settings
    struct Guard {
        mrs_gicp_batch* h; int core, cold; bool cert; hipEvent_t e0 = nullptr, e1 = nullptr;
        ~Guard() {
            h->search_core = core; h->cold_core = cold; h->use_certificates = cert;
            if (e0) (void)hipEventDestroy(e0);
            if (e1) (void)hipEventDestroy(e1);
        }
    } guard{h, h->search_core, h->cold_core, h->use_certificates};
    MRS_HIP_TRY(hipEventCreate(&guard.e0)); MRS_HIP_TRY(hipEventCreate(&guard.e1));
    const hipEvent_t e0 = guard.e0, e1 = guard.e1;
    auto timed = [&](float& ms, auto&& launch) -> int {
        launch();                                     // warm
        MRS_HIP_TRY(hipEventRecord(e0, s));
        for (int r = 0; r < reps; ++r) launch();
        MRS_HIP_TRY(hipEventRecord(e1, s));
        MRS_HIP_TRY(hipEventSynchronize(e1));
        MRS_HIP_TRY(hipGetLastError());
        MRS_HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
        ms /= (float)reps;
        return MRS_OK;
    };
    const dim3 lin_grid(h->max_blocks, P), wg((unsigned)((h->longest_src + kCertBlock - 1) / kCertBlock), P);
    h->search_core = 1; h->cold_core = 0; h->use_certificates = true;      // restored by `guard`
    auto round3 = [&]() {
        launch_nn_scan(h->longest_src, P, h->ctx->num_cu, s, h->d_pts[0], h->d_offs[0], h->d_pts[1], h->d_offs[1], h->d_tile_base[1], h->d_tlo[1],
                       h->d_thi[1], h->d_mlo[1], h->d_mhi[1], h->d_state, h->prm, h->d_corr, h->d_seed, (const int*)h->d_bbox[1], (float*)nullptr, 0);
    };
    auto round4_all = [&]() {
        hipLaunchKernelGGL((k_nn_scan_g<false, false>), wg, dim3(kNNThreads), 0, s, (const float4*)h->d_pts[0], (const int64_t*)h->d_offs[0],
                           (const float4*)h->d_pts[1], (const int64_t*)h->d_offs[1], h->hier(1), (const LmState*)h->d_state, h->prm, h->d_corr, h->d_seed,
                           (const int*)h->d_bbox[1], h->cert);
    };
    auto certify = [&]() {
        hipLaunchKernelGGL(k_nn_certify, wg, dim3(256), 0, s, (const float4*)h->d_pts[0], (const int64_t*)h->d_offs[0], (const float4*)h->d_pts[1],
                           (const int64_t*)h->d_offs[1], (const LmState*)h->d_state, h->prm, h->d_corr, (const int*)h->d_seed, h->cert);
    };
    auto listed = [&]() {
        hipLaunchKernelGGL((k_nn_scan_g<false, true>), wg, dim3(kNNThreads), 0, s, (const float4*)h->d_pts[0], (const int64_t*)h->d_offs[0],
                           (const float4*)h->d_pts[1], (const int64_t*)h->d_offs[1], h->hier(1), (const LmState*)h->d_state, h->prm, h->d_corr, h->d_seed,
                           (const int*)h->d_bbox[1], h->cert);
    };
    auto store_pose = [&]() {
        hipLaunchKernelGGL(k_nn_store_pose, dim3((P + 255) / 256), dim3(256), 0, s, (const LmState*)h->d_state, P, h->cert, 0, (const int64_t*)h->d_offs[0],
                           h->prm.motion_switch);
    };
    auto fail = [&](int code) { return code; };      // `guard` cleans up
    if ((st = upload(0, 0.0)) != MRS_OK) return fail(st);
    round3();                                         // seeds + correspondences at the poses
    if ((st = knn_dev_switches(s)) != MRS_OK) return fail(st);
    if ((st = timed(out_ms[2], round3)) != MRS_OK) return fail(st);
    // MRS_NN_TRACE_FILE: (start, end) of every workgroup of the last k_nn_scan launch (MRS_NN_TRACE_KERNEL=4: of k_nn_scan_g over every point), raw uint64 pairs
    auto dump_trace = [&](int which) -> int {
        const char* path = mrs::dev_env("MRS_NN_TRACE_FILE");
        const char* k = mrs::dev_env("MRS_NN_TRACE_KERNEL");
        if (!path || (k ? atoi(k) : 3) != which) return MRS_OK;
        std::vector<unsigned long long> tr(2 * 65536);
        MRS_HIP_TRY(hipStreamSynchronize(s));
        MRS_HIP_TRY(hipMemcpyFromSymbol(tr.data(), HIP_SYMBOL(g_nn_trace), tr.size() * sizeof(unsigned long long)));
        if (FILE* f = fopen(path, "wb")) { fwrite(tr.data(), sizeof(unsigned long long), tr.size(), f); fclose(f); }
        return MRS_OK;
    };
    if ((st = dump_trace(3)) != MRS_OK) return fail(st);
    if ((st = timed(out_ms[0], [&]() {
             launch_linearize(lin_grid, s, h->d_pts[0], h->d_offs[0], h->d_cov[0], h->d_pts[1], h->d_offs[1], h->d_cov[1],
                                h->d_state, h->d_corr, h->d_partial, h->max_blocks, 0);
         })) != MRS_OK) return fail(st);
    {   // correspondences at the poses
        std::vector<int> corr(h->n_seed);
        MRS_HIP_TRY(hipMemcpy(corr.data(), h->d_corr, corr.size() * sizeof(int), hipMemcpyDeviceToHost));
        int64_t c = 0;
        for (int v : corr) c += v >= 0;
        out_counts[0] = (int64_t)h->n_seed; out_counts[1] = c;
    }
    if ((st = timed(out_ms[4], round4_all)) != MRS_OK) return fail(st);      // leaves certificates at the poses
    if ((st = dump_trace(4)) != MRS_OK) return fail(st);
    store_pose();
    if ((st = timed(out_ms[3], certify)) != MRS_OK) return fail(st);
    if ((st = upload(1, 0.0)) != MRS_OK) return fail(st);
    if ((st = timed(out_ms[1], [&]() {
             launch_linearize(lin_grid, s, h->d_pts[0], h->d_offs[0], h->d_cov[0], h->d_pts[1], h->d_offs[1], h->d_cov[1],
                                h->d_state, h->d_corr, h->d_partial, h->max_blocks, 0);
         })) != MRS_OK) return fail(st);
    // a pass after a 1 mm step: certify + search the work lists (the certificates are those of the unmoved poses: t_prev stays)
    if ((st = upload(0, 1e-3)) != MRS_OK) return fail(st);
    {
        mrs::Scratch lb_save;
        if ((st = lb_save.alloc(h->n_seed * sizeof(float), s)) != MRS_OK) return fail(st);
        MRS_HIP_TRY(hipMemcpyAsync(lb_save.p, h->cert.lb, h->n_seed * sizeof(float), hipMemcpyDeviceToDevice, s));
        float acc = 0.0f;
        for (int r = 0; r < reps + 1; ++r) {
            MRS_HIP_TRY(hipMemcpyAsync(h->cert.lb, lb_save.p, h->n_seed * sizeof(float), hipMemcpyDeviceToDevice, s));
            MRS_HIP_TRY(hipEventRecord(e0, s));
            certify(); listed();
            MRS_HIP_TRY(hipEventRecord(e1, s));
            MRS_HIP_TRY(hipEventSynchronize(e1));
            float ms = 0.0f;
            MRS_HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
            if (r > 0) acc += ms;
        }
        out_ms[7] = acc / (float)reps;
        std::vector<int> bc((size_t)P * h->cert.nb);
        MRS_HIP_TRY(hipMemcpy(bc.data(), h->cert.bcount, bc.size() * sizeof(int), hipMemcpyDeviceToHost));
        int64_t q = 0;
        for (int p = 0; p < P; ++p) {
            const int64_t n = h->offs[0][p + 1] - h->offs[0][p];
            for (int b = 0; b * (int64_t)kCertBlock < n; ++b) {
                const int64_t members = std::min<int64_t>(kCertBlock, n - b * (int64_t)kCertBlock), c = bc[(size_t)p * h->cert.nb + b];
                q += 2 * c > members ? members : c;
            }
        }
        out_counts[2] = q;
    }
    {   // k-NN selection + covariances of the sources
        const int k = h->prm.k;
        mrs::Scratch knn;
        if ((st = knn.alloc(knn_ints(h->n_seed, P, k) * sizeof(int), s)) != MRS_OK) return fail(st);
        auto select = [&]() { launch_knn_cov(h, 0, 0, P, h->longest_src, k, knn.as<int>(), s); };
        if ((st = timed(out_ms[5], select)) != MRS_OK) return fail(st);
        if ((st = timed(out_ms[6], [&]() {
                 hipLaunchKernelGGL(k_cov_from_knn, dim3((unsigned)((h->longest_src + 255) / 256), P), dim3(256), 0, s, (const float4*)h->d_pts[0],
                                    (const int64_t*)h->d_offs[0], k, (const int*)knn.as<int>(), h->d_cov[0], (int*)nullptr);
             })) != MRS_OK) return fail(st);
        MRS_HIP_TRY(hipStreamSynchronize(s));
    }
    return MRS_OK;       // `guard` restores the search settings and destroys the events
}

int mrs_gicp_batch_fitness(mrs_gicp_batch* h, const double* h_poses, double max_range, double* h_scores,
                           mrs_stream stream)
{
    MRS_REQUIRE(h && h_poses && h_scores, "null pointer");
    MRS_REQUIRE(h->d_pts[0] && h->d_pts[1], "set source and target clouds first");
    MRS_HIP_TRY(hipSetDevice(h->ctx->device));
    hipStream_t s = (hipStream_t)stream;
    int st = ensure_state(h);
    if (st != MRS_OK) return st;
    mrs::Scratch poses, part;
    st = poses.alloc((size_t)h->n_pairs * 16 * sizeof(double), s);
    if (st != MRS_OK) return st;
    // one round of the search (512 source points) per workgroup: the reduction kernels' grid (max_blocks: up to 4096 points per workgroup) left one
    // pair of 39 k points to 10 workgroups, 337 us for a search that takes 90
    const int fb = std::max(1, (h->longest_src + 2 * kNNThreads - 1) / (2 * kNNThreads));
    st = part.alloc((size_t)h->n_pairs * fb * 2 * sizeof(double), s);
    if (st != MRS_OK) return st;
    // pageable host memory: a blocking copy (hipMemcpyAsync from a caller's stack array may be deferred past the launch)
    MRS_HIP_TRY(hipStreamSynchronize(s));
    MRS_HIP_TRY(hipMemcpy(poses.p, h_poses, (size_t)h->n_pairs * 16 * sizeof(double), hipMemcpyHostToDevice));
    MRS_HIP_TRY(hipMemsetAsync(part.p, 0, (size_t)h->n_pairs * fb * 2 * sizeof(double), s));
    hipLaunchKernelGGL(k_fitness, dim3(fb, h->n_pairs), dim3(kNNThreads), 0, s, h->d_pts[0], h->d_offs[0],
                       h->d_pts[1], h->d_offs[1], h->d_tile_base[1], h->d_tlo[1], h->d_thi[1], h->d_mlo[1], h->d_mhi[1], poses.as<double>(), max_range,
                       part.as<double>(), fb, (const int*)h->d_seed);
    MRS_HIP_TRY(hipGetLastError());
    std::vector<double> hp((size_t)h->n_pairs * fb * 2);
    MRS_HIP_TRY(hipMemcpyAsync(hp.data(), part.p, hp.size() * sizeof(double), hipMemcpyDeviceToHost, s));
    MRS_HIP_TRY(hipStreamSynchronize(s));
    for (int p = 0; p < h->n_pairs; ++p) {
        double sum = 0, cnt = 0;
        for (int b = 0; b < fb; ++b) { sum += hp[((size_t)p * fb + b) * 2]; cnt += hp[((size_t)p * fb + b) * 2 + 1]; }
        h_scores[p] = cnt > 0 ? sum / cnt : DBL_MAX;  // pcl: std::numeric_limits<double>::max() when empty
    }
    return MRS_OK;
}


/* ---- RING++ point-feature front-end (row N1) ---- */
int mrs_pointfeat_from_neighbors(mrs_ctx* ctx, const float* d_points, int32_t n, int32_t k, const int32_t* d_knn,
                                 const float* d_eigens, float* d_features, mrs_stream stream)
{
    MRS_REQUIRE(ctx && d_points && d_knn && d_eigens && d_features, "null pointer");
    MRS_REQUIRE(n >= 0, "n must be >= 0");
    MRS_REQUIRE(k >= 1 && k <= 32, "k must be in [1, 32]");
    MRS_HIP_TRY(hipSetDevice(ctx->device));
    if (n == 0) return MRS_OK;
    hipLaunchKernelGGL(k_features_from_neighbors, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, d_points, n, k,
                       d_knn, d_eigens, d_features);
    MRS_HIP_TRY(hipGetLastError());
    return MRS_OK;
}

int mrs_pointfeat_from_neighbors_host(mrs_ctx* ctx, const float* h_points, int32_t n, int32_t k, const int32_t* h_knn,
                                      const float* h_eigens, float* h_features)
{
    MRS_REQUIRE(ctx && h_points && h_knn && h_eigens && h_features, "null pointer");
    MRS_REQUIRE(n > 0, "n must be positive");
    MRS_HIP_TRY(hipSetDevice(ctx->device));
    float *dp = nullptr, *de = nullptr, *df = nullptr;
    int* dk = nullptr;
    int st = MRS_OK;
    if (hipMalloc(&dp, (size_t)n * 3 * 4) != hipSuccess || hipMalloc(&dk, (size_t)n * k * 4) != hipSuccess ||
        hipMalloc(&de, (size_t)n * 5 * 4) != hipSuccess || hipMalloc(&df, (size_t)n * 13 * 4) != hipSuccess) {
        mrs::set_error("hipMalloc failed in pointfeat host path");
        st = MRS_ERR_HIP;
    } else if (hipMemcpy(dp, h_points, (size_t)n * 3 * 4, hipMemcpyHostToDevice) != hipSuccess ||
               hipMemcpy(dk, h_knn, (size_t)n * k * 4, hipMemcpyHostToDevice) != hipSuccess ||
               hipMemcpy(de, h_eigens, (size_t)n * 5 * 4, hipMemcpyHostToDevice) != hipSuccess) {
        mrs::set_error("H2D copy failed");
        st = MRS_ERR_HIP;
    } else {
        st = mrs_pointfeat_from_neighbors(ctx, dp, n, k, dk, de, df, nullptr);
        if (st == MRS_OK && hipMemcpy(h_features, df, (size_t)n * 13 * 4, hipMemcpyDeviceToHost) != hipSuccess) {
            mrs::set_error("D2H copy failed");
            st = MRS_ERR_HIP;
        }
    }
    if (dp) (void)hipFree(dp);
    if (dk) (void)hipFree(dk);
    if (de) (void)hipFree(de);
    if (df) (void)hipFree(df);
    return st;
}

int mrs_pointfeat_batch(mrs_ctx* ctx, const float* d_points, int32_t stride_floats, const int64_t* h_offsets,
                        int32_t batch, int32_t k, int32_t* d_knn, float* d_eigens, float* d_features,
                        float* d_feat_planes, mrs_stream stream)
{
    MRS_REQUIRE(ctx && d_points && h_offsets, "null pointer");
    MRS_REQUIRE(batch > 0, "batch must be positive");
    MRS_REQUIRE(k >= 2 && k <= 32, "k must be in [2, 32]");
    MRS_REQUIRE(d_knn || d_eigens || d_features || d_feat_planes, "no output requested");
    // the Morton-ordered cloud container of the GICP front end, kept per context and batch size between calls (see mrs_ctx::pointfeat_cache)
    mrs_gicp_batch* h = nullptr;
    bool cached = false;
    int st = MRS_OK;
    std::unique_lock<std::mutex> lk(ctx->pointfeat_mu, std::try_to_lock);
    if (lk.owns_lock()) {
        auto it = ctx->pointfeat_cache.find(batch);
        if (it != ctx->pointfeat_cache.end()) { h = static_cast<mrs_gicp_batch*>(it->second); cached = true; }
    }
    if (!h) {
        st = mrs_gicp_batch_create(ctx, batch, &h);
        if (st != MRS_OK) return st;
        h->no_cov = true;          // no covariance buffer (48 B per point) for the feature front end
        if (lk.owns_lock() && ctx->pointfeat_cache.size() < 4) {
            ctx->pointfeat_cache[batch] = h;
            ctx->pointfeat_free = [](void* p) { (void)mrs_gicp_batch_destroy(static_cast<mrs_gicp_batch*>(p)); };
            cached = true;
        }
    }
    {
        const char* cs = mrs::dev_env("MRS_NN_CORE");
        h->want_leaf_hier = cs && atoi(cs) == 1;
        if (h->want_leaf_hier) { h->search_core = 1; h->cold_core = 1; }
    }
    st = mrs_gicp_batch_set_clouds(h, 0, d_points, stride_floats, h_offsets, stream);
    if (st == MRS_OK) {
        hipStream_t s = (hipStream_t)stream;
        int64_t longest = 0;
        for (int i = 0; i < batch; ++i) longest = std::max(longest, h_offsets[i + 1] - h_offsets[i]);
        static const char* const core_s = mrs::dev_env("MRS_NN_CORE");      // development aid: 1 = the round-4 k-NN kernel (slower here)
        const bool core4 = core_s && atoi(core_s) == 1;
        // the selection (k_knn_cov<30>: 5 waves per SIMD, no fp64 state) hands the neighbour indices to k_feat_from_knn through a scratch buffer
        // in blocks of 64 points, slot-major (knn_at); clouds go through in chunks that keep the buffer below 1 GiB (64 scans of 120 k points
        // at k = 30 are 0.92 GB: one chunk)
        const int64_t per_cloud = (int64_t)(knn_ints(longest, 1, k) * sizeof(int));
        const char* const lim_s = mrs::dev_env("MRS_FEAT_CHUNK_MB");        // development aid (the tests): a small limit forces several chunks
        const int64_t limit = lim_s ? std::max<int64_t>(1, atoll(lim_s)) << 20 : (1ll << 30);
        const int chunk = core4 ? batch : (int)std::max<int64_t>(1, std::min<int64_t>(batch, limit / per_cloud));
        mrs::Scratch knn;
        st = knn.alloc(core4 ? knn_ints(h_offsets[batch], batch, k) * sizeof(int) : (size_t)chunk * per_cloud, s);
        if (st == MRS_OK && !core4) st = knn_dev_switches(s);
        for (int c0 = 0; st == MRS_OK && c0 < batch; c0 += chunk) {
            const int nc = std::min(chunk, batch - c0);
            int* const kn = knn.as<int>() - (size_t)h_offsets[c0] * k;       // the kernels index by global point number (see compute_covariances)
            if (core4) st = launch_knn_select(h, 0, k, kn, s);
            else launch_knn_cov(h, 0, c0, nc, longest, k, kn, s);
            if (st != MRS_OK) break;
            hipLaunchKernelGGL(k_feat_from_knn, dim3((unsigned)((longest + 255) / 256), nc), dim3(256), 0, s, (const float4*)h->d_pts[0],
                               (const int64_t*)h->d_offs[0] + c0, k, (const int*)kn, d_knn, d_eigens, d_features, d_feat_planes);
        }
        if (st == MRS_OK && !core4) st = knn_dbg_report(s);
        if (st == MRS_OK && (hipGetLastError() != hipSuccess || hipStreamSynchronize(s) != hipSuccess)) {
            mrs::set_error("point-feature kernels failed: %s", hipGetErrorString(hipGetLastError()));
            st = MRS_ERR_HIP;
        }
    }
    if (!cached) mrs_gicp_batch_destroy(h);
    return st;
}

double mrs_gicp_batch_last_nn_passes(const mrs_gicp_batch* h) { return h ? h->last_nn_passes : 0.0; }

}  // extern "C"
