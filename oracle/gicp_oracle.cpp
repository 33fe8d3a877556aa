/*
 * oracle/gicp_oracle.cpp -- TEST INFRASTRUCTURE ONLY (not product code).
 *
 * CPU restatement of fast_gicp's FastGICP (GICP with PLANE-regularised covariances and a
 * Levenberg-Marquardt optimiser) as MR_SLAM configures it.
 *
 * PARITY UNPINNED.  fast_gicp is an un-vendored git submodule of the reference
 * (/root/reference/.gitmodules:1-6 -> https://github.com/SMRT-AIST/fast_gicp, no pinned
 * commit; Mapping/src/fast_gicp and LoopDetection/src/fast_gicp are empty directories) and the
 * reference holds no test or golden vector for it.  This file restates the published algorithm
 * of upstream include/fast_gicp/gicp/impl/{fast_gicp_impl.hpp,lsq_registration_impl.hpp} and
 * include/fast_gicp/so3/so3.hpp as recorded in SURVEY.md Appendix A.2, and is anchored on the
 * reference's call sites:
 *   Mapping/src/global_manager/src/global_manager.cpp:2435-2443 (threads 8, transEps 1e-3,
 *       maxIter icp_iters, maxCorrDist 100, k = 15), :2016-2021 (align), :2058-2071 (fitness)
 *   LoopDetection/src/RING_ros/main_RING.py:81-104 (pygicp: defaults k = 20, maxCorrDist 5.0)
 * Acceptance (tests/test_oracle_gicp.py): recovers known SE(3) perturbations of synthetic
 * clouds; an independent float64 numpy/scipy formulation of the same equations agrees.
 *
 * Nearest neighbours come from an exact kd-tree (the reference uses pcl::search::KdTree, also
 * exact), so results do not depend on the search structure except for exact distance ties.
 *
 * Upstream semantics this file follows function by function (recalled from upstream master; there
 * is no network and no vendored copy, so no tag could be checked -- see DESIGN.md section 2):
 *   FastGICP::update_correspondences  (fast_gicp_impl.hpp): float-transformed source point, 1-NN,
 *       reject d^2 >= max^2, mahalanobis_[i] = (C_B + T C_A T^T)^-1 at the LINEARISATION pose,
 *       both cached in correspondences_ / mahalanobis_.
 *   FastGICP::linearize               : update_correspondences(trans), then H, b, sum of e^T M e.
 *   FastGICP::compute_error           : NO new search: iterates the cached correspondences_ and
 *       mahalanobis_ left by the last linearize, only e = mu_B - trans mu_A is re-evaluated.
 *   LsqRegistration::step_lm          : y0 = linearize(x0); trials yi = compute_error(delta * x0).
 *   LsqRegistration::is_converged     : max(10 |R - I| / rotation_epsilon, 10 |t| / transformation_epsilon) < 1
 *       (the factor 10 is upstream's; exposed as conv_factor).
 *   GaussianVoxelMap / calc_voxel_coord (FastVGICPCuda): coord = floor(x / resolution - 0.5) on the
 *       float-transformed point in float arithmetic; voxel mean = mean of points, voxel covariance =
 *       mean of the points' regularised covariances; correspondences = voxel index per source point
 *       (DIRECT1/7/27 offsets), cached by linearize and reused by compute_error with the matrices of
 *       the linearisation pose (x_linearized / x_eval in the CUDA kernels); weight sqrt(points in voxel).
 */
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <map>
#include <numeric>
#include <tuple>
#include <vector>

#ifdef _OPENMP
#include <omp.h>
#endif

namespace {

// ------------------------------------------------------------------------------- kd-tree
struct KdTree {
    const float* pts = nullptr;  // [n][3]
    int n = 0;
    std::vector<int> idx;        // permutation
    struct Node { int lo, hi, axis; float split; int left, right; };
    std::vector<Node> nodes;
    static constexpr int kLeaf = 12;

    void build(const float* p, int count)
    {
        pts = p; n = count;
        idx.resize(n);
        std::iota(idx.begin(), idx.end(), 0);
        nodes.clear();
        nodes.reserve(2 * (n / kLeaf + 2));
        if (n > 0) build_rec(0, n);
    }
    int build_rec(int lo, int hi)
    {
        const int id = (int)nodes.size();
        nodes.push_back({lo, hi, -1, 0.f, -1, -1});
        if (hi - lo <= kLeaf) return id;
        float mn[3] = {1e30f, 1e30f, 1e30f}, mx[3] = {-1e30f, -1e30f, -1e30f};
        for (int i = lo; i < hi; ++i)
            for (int a = 0; a < 3; ++a) {
                const float v = pts[3 * (size_t)idx[i] + a];
                mn[a] = std::min(mn[a], v); mx[a] = std::max(mx[a], v);
            }
        int axis = 0;
        for (int a = 1; a < 3; ++a) if (mx[a] - mn[a] > mx[axis] - mn[axis]) axis = a;
        const int mid = (lo + hi) / 2;
        std::nth_element(idx.begin() + lo, idx.begin() + mid, idx.begin() + hi,
                         [&](int u, int v) { return pts[3 * (size_t)u + axis] < pts[3 * (size_t)v + axis]; });
        nodes[id].axis = axis;
        nodes[id].split = pts[3 * (size_t)idx[mid] + axis];
        const int l = build_rec(lo, mid);
        const int r = build_rec(mid, hi);
        nodes[id].left = l; nodes[id].right = r;
        return id;
    }
    // k nearest (sorted ascending by squared distance, ties by smaller index)
    void knn(const float* q, int k, int* out_idx, float* out_d2) const
    {
        for (int i = 0; i < k; ++i) { out_idx[i] = -1; out_d2[i] = std::numeric_limits<float>::infinity(); }
        if (n > 0) search(0, q, k, out_idx, out_d2);
    }
    static inline bool better(float d, int i, float d0, int i0) { return d < d0 || (d == d0 && i < i0); }
    void search(int id, const float* q, int k, int* oi, float* od) const
    {
        const Node& nd = nodes[id];
        if (nd.axis < 0) {
            for (int i = nd.lo; i < nd.hi; ++i) {
                const int p = idx[i];
                const float dx = q[0] - pts[3 * (size_t)p], dy = q[1] - pts[3 * (size_t)p + 1], dz = q[2] - pts[3 * (size_t)p + 2];
                const float d = fmaf(dz, dz, fmaf(dy, dy, dx * dx));  // same op order as the HIP scan
                if (better(d, p, od[k - 1], oi[k - 1] < 0 ? INT32_MAX : oi[k - 1])) {
                    int s = k - 1;
                    while (s > 0 && better(d, p, od[s - 1], oi[s - 1] < 0 ? INT32_MAX : oi[s - 1])) { od[s] = od[s - 1]; oi[s] = oi[s - 1]; --s; }
                    od[s] = d; oi[s] = p;
                }
            }
            return;
        }
        const float diff = q[nd.axis] - nd.split;
        const int first = diff < 0 ? nd.left : nd.right, second = diff < 0 ? nd.right : nd.left;
        search(first, q, k, oi, od);
        if (diff * diff <= od[k - 1]) search(second, q, k, oi, od);
    }
};

// ------------------------------------------------------------------------ small linear algebra
struct M3 { double m[9]; };  // row-major

inline void mul3(const double* a, const double* b, double* c)
{
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) c[3 * i + j] = a[3 * i] * b[j] + a[3 * i + 1] * b[3 + j] + a[3 * i + 2] * b[6 + j];
}

// eigenvector of the smallest eigenvalue of a symmetric 3x3 (cyclic Jacobi, double)
void smallest_eigvec(const double* c, double* n_out)
{
    double a[9], v[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    std::memcpy(a, c, sizeof(a));
    for (int sweep = 0; sweep < 30; ++sweep) {
        const double off = a[1] * a[1] + a[2] * a[2] + a[5] * a[5];
        if (off < 1e-300) break;
        for (int p = 0; p < 2; ++p)
            for (int q = p + 1; q < 3; ++q) {
                const double apq = a[3 * p + q];
                if (apq == 0.0) continue;
                const double theta = (a[3 * q + q] - a[3 * p + p]) / (2.0 * apq);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
                const double cs = 1.0 / std::sqrt(t * t + 1.0), sn = t * cs;
                for (int k = 0; k < 3; ++k) {  // A <- A J
                    const double akp = a[3 * k + p], akq = a[3 * k + q];
                    a[3 * k + p] = cs * akp - sn * akq;
                    a[3 * k + q] = sn * akp + cs * akq;
                }
                for (int k = 0; k < 3; ++k) {  // A <- J^T A
                    const double apk = a[3 * p + k], aqk = a[3 * q + k];
                    a[3 * p + k] = cs * apk - sn * aqk;
                    a[3 * q + k] = sn * apk + cs * aqk;
                }
                for (int k = 0; k < 3; ++k) {
                    const double vkp = v[3 * k + p], vkq = v[3 * k + q];
                    v[3 * k + p] = cs * vkp - sn * vkq;
                    v[3 * k + q] = sn * vkp + cs * vkq;
                }
            }
    }
    int s = 0;
    if (a[4] < a[0]) s = 1;
    if (a[8] < a[4 * s]) s = 2;
    n_out[0] = v[s]; n_out[1] = v[3 + s]; n_out[2] = v[6 + s];
}

bool inv3(const double* a, double* r)
{
    const double c0 = a[4] * a[8] - a[5] * a[7], c1 = a[5] * a[6] - a[3] * a[8], c2 = a[3] * a[7] - a[4] * a[6];
    const double det = a[0] * c0 + a[1] * c1 + a[2] * c2;
    if (det == 0.0) return false;
    const double id = 1.0 / det;
    r[0] = c0 * id; r[1] = (a[2] * a[7] - a[1] * a[8]) * id; r[2] = (a[1] * a[5] - a[2] * a[4]) * id;
    r[3] = c1 * id; r[4] = (a[0] * a[8] - a[2] * a[6]) * id; r[5] = (a[2] * a[3] - a[0] * a[5]) * id;
    r[6] = c2 * id; r[7] = (a[1] * a[6] - a[0] * a[7]) * id; r[8] = (a[0] * a[4] - a[1] * a[3]) * id;
    return true;
}

// LDL^T solve of a symmetric positive definite 6x6 (Eigen::LDLT in the reference)
bool solve6(const double* Hin, const double* rhs, double* x)
{
    double L[36] = {0}, D[6];
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

// so3_exp / se3_exp (upstream include/fast_gicp/so3/so3.hpp)
void se3_exp(const double* a, double* T /*4x4 row-major*/)
{
    const double wx = a[0], wy = a[1], wz = a[2];
    const double theta_sq = wx * wx + wy * wy + wz * wz;
    double imag, real, theta = 0;
    if (theta_sq < 1e-10) {
        const double t4 = theta_sq * theta_sq;
        imag = 0.5 - (1.0 / 48.0) * theta_sq + (1.0 / 3840.0) * t4;
        real = 1.0 - (1.0 / 8.0) * theta_sq + (1.0 / 384.0) * t4;
    } else {
        theta = std::sqrt(theta_sq);
        const double half = 0.5 * theta;
        imag = std::sin(half) / theta;
        real = std::cos(half);
    }
    double qw = real, qx = imag * wx, qy = imag * wy, qz = imag * wz;
    const double nq = std::sqrt(qw * qw + qx * qx + qy * qy + qz * qz);
    qw /= nq; qx /= nq; qy /= nq; qz /= nq;
    double R[9] = {1 - 2 * (qy * qy + qz * qz), 2 * (qx * qy - qz * qw), 2 * (qx * qz + qy * qw),
                   2 * (qx * qy + qz * qw), 1 - 2 * (qx * qx + qz * qz), 2 * (qy * qz - qx * qw),
                   2 * (qx * qz - qy * qw), 2 * (qy * qz + qx * qw), 1 - 2 * (qx * qx + qy * qy)};
    double V[9];
    if (theta < 1e-10) {
        std::memcpy(V, R, sizeof(V));
    } else {
        const double O[9] = {0, -wz, wy, wz, 0, -wx, -wy, wx, 0};
        double O2[9];
        mul3(O, O, O2);
        const double c1 = (1.0 - std::cos(theta)) / theta_sq, c2 = (theta - std::sin(theta)) / (theta_sq * theta);
        for (int i = 0; i < 9; ++i) V[i] = (i % 4 == 0 ? 1.0 : 0.0) + c1 * O[i] + c2 * O2[i];
    }
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) T[4 * i + j] = R[3 * i + j];
        T[4 * i + 3] = V[3 * i] * a[3] + V[3 * i + 1] * a[4] + V[3 * i + 2] * a[5];
    }
    T[12] = T[13] = T[14] = 0; T[15] = 1;
}

void mul4(const double* a, const double* b, double* c)
{
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            double s = 0;
            for (int k = 0; k < 4; ++k) s += a[4 * i + k] * b[4 * k + j];
            c[4 * i + j] = s;
        }
}

struct Cloud {
    std::vector<float> pts;   // [n][3]
    std::vector<double> cov;  // [n][9] 3x3 block of the regularised covariance
    KdTree tree;
    int n = 0;
};

struct Gicp {
    Cloud src, tgt;
    int k = 20;
    double max_corr = std::numeric_limits<double>::max();
    int max_iter = 64;
    double rot_eps = 2e-3, trans_eps = 5e-4;
    double conv_factor = 10.0;   // upstream is_converged scales both deltas by 10 before the test
    int lm_max_iter = 10;
    double lm_init_factor = 1e-9;
    double lm_lambda = -1.0;
    int threads = 1;
    // per-source-point state
    std::vector<int> corr;
    std::vector<double> mahal;  // [n][9]
    double final_T[16];
    double final_H[36];
    int converged = 0, iterations = 0;
    int lm_trials = 0;
    // row G7: voxelised target (FastVGICP); 0 = plain GICP
    double voxel_res = 0.0;
    int voxel_neighbors = 1;
    struct Voxel { double mean[3] = {0, 0, 0}; double cov[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}; int n = 0; };
    std::map<std::tuple<int, int, int>, Voxel> voxels;
    double voxels_built_for = 0.0;
    // cached by linearize_voxel (upstream voxel_correspondences_ + matrices of the linearisation pose)
    struct VoxCorr { int src; const Voxel* vox; double M[9]; };
    std::vector<VoxCorr> vox_corr;
    int nn_passes = 0;           // update_correspondences calls of the last align
};

// calc_voxel_coord of upstream's CUDA voxel map: float arithmetic, floor(x / res - 0.5)
inline int voxel_coord_f(float v, float res) { return (int)std::floor(v / res - 0.5f); }

// Gaussian voxel map of the target (ADDITIVE accumulation): mean of the points, mean of their covariances
void build_voxels(Gicp& g)
{
    if (g.voxels_built_for == g.voxel_res && !g.voxels.empty()) return;
    g.voxels.clear();
    for (int i = 0; i < g.tgt.n; ++i) {
        const float* p = &g.tgt.pts[3 * (size_t)i];
        const float resf = (float)g.voxel_res;
        auto key = std::make_tuple(voxel_coord_f(p[0], resf), voxel_coord_f(p[1], resf), voxel_coord_f(p[2], resf));
        Gicp::Voxel& v = g.voxels[key];
        for (int a = 0; a < 3; ++a) v.mean[a] += (double)p[a];
        for (int a = 0; a < 9; ++a) v.cov[a] += g.tgt.cov[9 * (size_t)i + a];
        ++v.n;
    }
    for (auto& kv : g.voxels) {
        for (int a = 0; a < 3; ++a) kv.second.mean[a] = (double)(float)(kv.second.mean[a] / kv.second.n);
        for (int a = 0; a < 9; ++a) kv.second.cov[a] /= kv.second.n;
    }
    g.voxels_built_for = g.voxel_res;
}

// row G7, upstream FastVGICPCuda::update_correspondences: the voxel containing the float-transformed source
// point (+ 6 / 26 neighbours) and the matrix (C_voxel + R C_A R^T)^-1 of THIS pose, cached per correspondence
void update_voxel_correspondences(Gicp& g, const double* T)
{
    build_voxels(g);
    ++g.nn_passes;
    g.vox_corr.clear();
    const double R[9] = {T[0], T[1], T[2], T[4], T[5], T[6], T[8], T[9], T[10]};
    double Rt[9];
    for (int a = 0; a < 3; ++a) for (int bb = 0; bb < 3; ++bb) Rt[3 * a + bb] = R[3 * bb + a];
    float Tf[16];
    for (int i = 0; i < 16; ++i) Tf[i] = (float)T[i];
    const float resf = (float)g.voxel_res;
    for (int i = 0; i < g.src.n; ++i) {
        const float* pa = &g.src.pts[3 * (size_t)i];
        float q[3];
        for (int a = 0; a < 3; ++a) q[a] = Tf[4 * a] * pa[0] + Tf[4 * a + 1] * pa[1] + Tf[4 * a + 2] * pa[2] + Tf[4 * a + 3];
        const int c[3] = {voxel_coord_f(q[0], resf), voxel_coord_f(q[1], resf), voxel_coord_f(q[2], resf)};
        double RC[9], RCRa[9];
        mul3(R, &g.src.cov[9 * (size_t)i], RC);
        mul3(RC, Rt, RCRa);
        for (int o = 0; o < 27; ++o) {
            const int dx = o % 3 - 1, dy = (o / 3) % 3 - 1, dz = o / 9 - 1;
            const int man = std::abs(dx) + std::abs(dy) + std::abs(dz);
            if ((g.voxel_neighbors == 1 && man != 0) || (g.voxel_neighbors == 7 && man > 1)) continue;
            auto it = g.voxels.find(std::make_tuple(c[0] + dx, c[1] + dy, c[2] + dz));
            if (it == g.voxels.end()) continue;
            Gicp::VoxCorr vc;
            vc.src = i; vc.vox = &it->second;
            double RCR[9];
            for (int a = 0; a < 9; ++a) RCR[a] = RCRa[a] + it->second.cov[a];
            if (!inv3(RCR, vc.M)) continue;
            g.vox_corr.push_back(vc);
        }
    }
}

// sums over the cached voxel correspondences at the evaluation pose T (upstream compute_error(trans, H, b):
// correspondences and matrices of the linearisation pose, residuals of the evaluation pose)
double evaluate_voxel(const Gicp& g, const double* T, double* H, double* b)
{
    double Hs[36] = {0}, bs[6] = {0}, err = 0;
    for (const Gicp::VoxCorr& vc : g.vox_corr) {
        const float* pa = &g.src.pts[3 * (size_t)vc.src];
        double ta[3];
        for (int a = 0; a < 3; ++a) ta[a] = T[4 * a] * (double)pa[0] + T[4 * a + 1] * (double)pa[1] + T[4 * a + 2] * (double)pa[2] + T[4 * a + 3];
        const Gicp::Voxel& v = *vc.vox;
        const double* M = vc.M;
        const double w = std::sqrt((double)v.n);
        double e[3], Me[3];
        for (int a = 0; a < 3; ++a) e[a] = v.mean[a] - ta[a];
        for (int a = 0; a < 3; ++a) Me[a] = M[3 * a] * e[0] + M[3 * a + 1] * e[1] + M[3 * a + 2] * e[2];
        err += w * (e[0] * Me[0] + e[1] * Me[1] + e[2] * Me[2]);
        if (!H) continue;
        double J[18] = {0, -ta[2], ta[1], -1, 0, 0, ta[2], 0, -ta[0], 0, -1, 0, -ta[1], ta[0], 0, 0, 0, -1};
        double MJ[18];
        for (int a = 0; a < 3; ++a)
            for (int cc = 0; cc < 6; ++cc) MJ[6 * a + cc] = M[3 * a] * J[cc] + M[3 * a + 1] * J[6 + cc] + M[3 * a + 2] * J[12 + cc];
        for (int r = 0; r < 6; ++r) {
            for (int cc = 0; cc < 6; ++cc) Hs[6 * r + cc] += w * (J[r] * MJ[cc] + J[6 + r] * MJ[6 + cc] + J[12 + r] * MJ[12 + cc]);
            bs[r] += w * (J[r] * Me[0] + J[6 + r] * Me[1] + J[12 + r] * Me[2]);
        }
    }
    if (H) { std::memcpy(H, Hs, sizeof(Hs)); std::memcpy(b, bs, sizeof(bs)); }
    return err;
}

double linearize_voxel(Gicp& g, const double* T, double* H, double* b)
{
    update_voxel_correspondences(g, T);
    return evaluate_voxel(g, T, H, b);
}

void compute_covariances(Cloud& c, int k, int threads)
{
    c.cov.assign((size_t)c.n * 9, 0.0);
#pragma omp parallel for num_threads(threads) schedule(guided, 8)
    for (int i = 0; i < c.n; ++i) {
        std::vector<int> ki(k);
        std::vector<float> kd(k);
        c.tree.knn(&c.pts[3 * (size_t)i], k, ki.data(), kd.data());
        double mean[3] = {0, 0, 0};
        int cnt = 0;
        for (int j = 0; j < k; ++j) if (ki[j] >= 0) { for (int a = 0; a < 3; ++a) mean[a] += (double)c.pts[3 * (size_t)ki[j] + a]; ++cnt; }
        for (int a = 0; a < 3; ++a) mean[a] /= cnt;
        double cov[9] = {0};
        for (int j = 0; j < k; ++j) if (ki[j] >= 0) {
            double d[3];
            for (int a = 0; a < 3; ++a) d[a] = (double)c.pts[3 * (size_t)ki[j] + a] - mean[a];
            for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) cov[3 * a + b] += d[a] * d[b];
        }
        for (int a = 0; a < 9; ++a) cov[a] /= cnt;
        // PLANE regularisation: U diag(1,1,1e-3) V^T  ==  I - (1 - 1e-3) n n^T
        double nrm[3];
        smallest_eigvec(cov, nrm);
        double* o = &c.cov[9 * (size_t)i];
        for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) o[3 * a + b] = (a == b ? 1.0 : 0.0) - 0.999 * nrm[a] * nrm[b];
    }
}

// update_correspondences: NN of the float-transformed source point, Mahalanobis matrices
void update_correspondences(Gicp& g, const double* T)
{
    const int n = g.src.n;
    ++g.nn_passes;
    g.corr.assign(n, -1);
    g.mahal.assign((size_t)n * 9, 0.0);
    float Tf[16];
    for (int i = 0; i < 16; ++i) Tf[i] = (float)T[i];
    const double R[9] = {T[0], T[1], T[2], T[4], T[5], T[6], T[8], T[9], T[10]};
    const double max2 = g.max_corr >= 1e150 ? std::numeric_limits<double>::infinity() : g.max_corr * g.max_corr;
#pragma omp parallel for num_threads(g.threads) schedule(guided, 8)
    for (int i = 0; i < n; ++i) {
        const float* p = &g.src.pts[3 * (size_t)i];
        float q[3];
        for (int a = 0; a < 3; ++a) q[a] = Tf[4 * a] * p[0] + Tf[4 * a + 1] * p[1] + Tf[4 * a + 2] * p[2] + Tf[4 * a + 3];
        int j; float d2;
        g.tgt.tree.knn(q, 1, &j, &d2);
        if (j < 0 || !((double)d2 < max2)) continue;
        g.corr[i] = j;
        double RC[9], RCR[9], Rt[9];
        for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) Rt[3 * a + b] = R[3 * b + a];
        mul3(R, &g.src.cov[9 * (size_t)i], RC);
        mul3(RC, Rt, RCR);
        for (int a = 0; a < 9; ++a) RCR[a] += g.tgt.cov[9 * (size_t)j + a];
        inv3(RCR, &g.mahal[9 * (size_t)i]);
    }
}

// linearize (H, b optional) -> sum of e^T M e
double linearize_voxel(Gicp& g, const double* T, double* H, double* b);

// sums over the cached correspondences_ / mahalanobis_ at the evaluation pose T
double evaluate(const Gicp& g, const double* T, double* H, double* b)
{
    const int n = g.src.n;
    const int nt = g.threads;
    std::vector<double> Hs((size_t)nt * 36, 0.0), bs((size_t)nt * 6, 0.0), es(nt, 0.0);
#pragma omp parallel for num_threads(nt) schedule(guided, 8)
    for (int i = 0; i < n; ++i) {
        const int j = g.corr[i];
        if (j < 0) continue;
#ifdef _OPENMP
        const int t = omp_get_thread_num();
#else
        const int t = 0;
#endif
        const float* pa = &g.src.pts[3 * (size_t)i];
        const float* pb = &g.tgt.pts[3 * (size_t)j];
        double ta[3], e[3];
        for (int a = 0; a < 3; ++a) ta[a] = T[4 * a] * (double)pa[0] + T[4 * a + 1] * (double)pa[1] + T[4 * a + 2] * (double)pa[2] + T[4 * a + 3];
        for (int a = 0; a < 3; ++a) e[a] = (double)pb[a] - ta[a];
        const double* M = &g.mahal[9 * (size_t)i];
        double Me[3];
        for (int a = 0; a < 3; ++a) Me[a] = M[3 * a] * e[0] + M[3 * a + 1] * e[1] + M[3 * a + 2] * e[2];
        es[t] += e[0] * Me[0] + e[1] * Me[1] + e[2] * Me[2];
        if (!H) continue;
        // J = [ skew(T a)  -I ]  (3x6)
        double J[18] = {0, -ta[2], ta[1], -1, 0, 0,
                        ta[2], 0, -ta[0], 0, -1, 0,
                        -ta[1], ta[0], 0, 0, 0, -1};
        double MJ[18];
        for (int a = 0; a < 3; ++a)
            for (int c = 0; c < 6; ++c) MJ[6 * a + c] = M[3 * a] * J[c] + M[3 * a + 1] * J[6 + c] + M[3 * a + 2] * J[12 + c];
        double* Ht = &Hs[(size_t)t * 36];
        double* bt = &bs[(size_t)t * 6];
        for (int r = 0; r < 6; ++r) {
            for (int c = 0; c < 6; ++c) Ht[6 * r + c] += J[r] * MJ[c] + J[6 + r] * MJ[6 + c] + J[12 + r] * MJ[12 + c];
            bt[r] += J[r] * Me[0] + J[6 + r] * Me[1] + J[12 + r] * Me[2];
        }
    }
    double err = 0;
    for (int t = 0; t < nt; ++t) err += es[t];
    if (H) {
        std::fill(H, H + 36, 0.0); std::fill(b, b + 6, 0.0);
        for (int t = 0; t < nt; ++t) { for (int a = 0; a < 36; ++a) H[a] += Hs[(size_t)t * 36 + a]; for (int a = 0; a < 6; ++a) b[a] += bs[(size_t)t * 6 + a]; }
    }
    return err;
}

// upstream FastGICP::linearize: the only place that searches
double linearize(Gicp& g, const double* T, double* H, double* b)
{
    if (g.voxel_res > 0.0) return linearize_voxel(g, T, H, b);
    update_correspondences(g, T);
    return evaluate(g, T, H, b);
}

// upstream FastGICP::compute_error / FastVGICPCuda::compute_error: cached correspondences and matrices
double compute_error(const Gicp& g, const double* T)
{
    return g.voxel_res > 0.0 ? evaluate_voxel(g, T, nullptr, nullptr) : evaluate(g, T, nullptr, nullptr);
}

bool is_converged(const Gicp& g, const double* delta)
{
    double mr = 0, mt = 0;
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) mr = std::max(mr, g.conv_factor * std::fabs(delta[4 * i + j] - (i == j ? 1.0 : 0.0)) / g.rot_eps);
        mt = std::max(mt, g.conv_factor * std::fabs(delta[4 * i + 3]) / g.trans_eps);
    }
    return std::max(mr, mt) < 1.0;
}

// one LM outer iteration (upstream LsqRegistration::step_lm)
bool step_lm(Gicp& g, double* x0, double* delta)
{
    double H[36], b[6];
    const double y0 = linearize(g, x0, H, b);
    if (g.lm_lambda < 0.0) {
        double mx = 0;
        for (int i = 0; i < 6; ++i) mx = std::max(mx, std::fabs(H[7 * i]));
        g.lm_lambda = g.lm_init_factor * mx;
    }
    double nu = 2.0;
    for (int it = 0; it < g.lm_max_iter; ++it) {
        ++g.lm_trials;
        double Hl[36], rhs[6], d[6];
        std::memcpy(Hl, H, sizeof(H));
        for (int i = 0; i < 6; ++i) { Hl[7 * i] += g.lm_lambda; rhs[i] = -b[i]; }
        if (!solve6(Hl, rhs, d)) return false;
        se3_exp(d, delta);
        double xi[16];
        mul4(delta, x0, xi);
        const double yi = compute_error(g, xi);
        double denom = 0;
        for (int i = 0; i < 6; ++i) denom += d[i] * (g.lm_lambda * d[i] - b[i]);
        const double rho = (y0 - yi) / denom;
        if (rho < 0) {
            if (is_converged(g, delta)) return true;
            g.lm_lambda = nu * g.lm_lambda;
            nu = 2 * nu;
            continue;
        }
        std::memcpy(x0, xi, sizeof(xi));
        g.lm_lambda = g.lm_lambda * std::max(1.0 / 3.0, 1.0 - std::pow(2 * rho - 1, 3));
        std::memcpy(g.final_H, H, sizeof(H));
        return true;
    }
    return false;
}

}  // namespace

extern "C" {

void* orc_gicp_create() { return new Gicp(); }
void orc_gicp_destroy(void* h) { delete static_cast<Gicp*>(h); }

static void set_cloud(Cloud& c, const float* xyz, int n)
{
    c.n = n;
    c.pts.assign(xyz, xyz + 3 * (size_t)n);
    c.cov.clear();
    c.tree.build(c.pts.data(), n);
}
void orc_gicp_set_source(void* h, const float* xyz, int n) { set_cloud(static_cast<Gicp*>(h)->src, xyz, n); }
void orc_gicp_set_target(void* h, const float* xyz, int n) { set_cloud(static_cast<Gicp*>(h)->tgt, xyz, n); static_cast<Gicp*>(h)->voxels.clear(); }

void orc_gicp_set_params(void* h, int k, double max_corr, int max_iter, double rot_eps, double trans_eps, int threads)
{
    Gicp* g = static_cast<Gicp*>(h);
    g->k = k; g->max_corr = max_corr; g->max_iter = max_iter; g->rot_eps = rot_eps; g->trans_eps = trans_eps;
    g->threads = threads > 0 ? threads : 1;
    g->src.cov.clear(); g->tgt.cov.clear();
}

void orc_gicp_set_voxel(void* h, double resolution, int neighbors)
{
    Gicp* g = static_cast<Gicp*>(h);
    g->voxel_res = resolution; g->voxel_neighbors = neighbors; g->voxels.clear();
}

void orc_gicp_covariances(void* h, int which, double* out /* [n][9] or null */)
{
    Gicp* g = static_cast<Gicp*>(h);
    Cloud& c = which ? g->tgt : g->src;
    compute_covariances(c, g->k, g->threads);
    if (out) std::memcpy(out, c.cov.data(), c.cov.size() * sizeof(double));
}

/* computeTransformation: guess / final are row-major 4x4 doubles; the result is narrowed to
 * float like final_transformation_ = x0.cast<float>().  force_iters > 0 runs exactly that many
 * outer iterations (convergence test disabled, for timing). */
int orc_gicp_align(void* h, const double* guess, double* final_T, int force_iters)
{
    Gicp* g = static_cast<Gicp*>(h);
    if (g->src.cov.empty()) compute_covariances(g->src, g->k, g->threads);
    if (g->tgt.cov.empty()) compute_covariances(g->tgt, g->k, g->threads);
    double x0[16];
    for (int i = 0; i < 16; ++i) x0[i] = (double)(float)guess[i];  // guess.cast<double>() of a Matrix4f
    g->lm_lambda = -1.0;
    g->converged = 0;
    g->iterations = 0;
    g->lm_trials = 0;
    g->nn_passes = 0;
    const int limit = force_iters > 0 ? force_iters : g->max_iter;
    for (int i = 0; i < limit && !g->converged; ++i) {
        double delta[16];
        if (!step_lm(*g, x0, delta)) break;  // "lm not converged" / solver failure
        g->iterations = i + 1;
        if (force_iters <= 0) g->converged = is_converged(*g, delta);
    }
    for (int i = 0; i < 16; ++i) g->final_T[i] = (double)(float)x0[i];
    std::memcpy(final_T, g->final_T, sizeof(g->final_T));
    return g->converged;
}

int orc_gicp_iterations(void* h) { return static_cast<Gicp*>(h)->iterations; }
int orc_gicp_lm_trials(void* h) { return static_cast<Gicp*>(h)->lm_trials; }
int orc_gicp_nn_passes(void* h) { return static_cast<Gicp*>(h)->nn_passes; }
void orc_gicp_set_conv_factor(void* h, double f) { static_cast<Gicp*>(h)->conv_factor = f; }

/* one linearisation at T (for kernel-level parity): returns the error, fills H[36], b[6] and,
 * if non-null, the per-source correspondences */
double orc_gicp_linearize(void* h, const double* T, double* H, double* b, int* corr)
{
    Gicp* g = static_cast<Gicp*>(h);
    if (g->src.cov.empty()) compute_covariances(g->src, g->k, g->threads);
    if (g->tgt.cov.empty()) compute_covariances(g->tgt, g->k, g->threads);
    const double e = linearize(*g, T, H, b);
    if (corr) std::memcpy(corr, g->corr.data(), g->corr.size() * sizeof(int));
    return e;
}

/* pcl::Registration::getFitnessScore(max_range) (global_manager.cpp:2058): mean squared NN
 * distance of the transformed source over correspondences whose SQUARED distance <= max_range */
double orc_gicp_fitness(void* h, const double* T, double max_range)
{
    Gicp* g = static_cast<Gicp*>(h);
    float Tf[16];
    for (int i = 0; i < 16; ++i) Tf[i] = (float)T[i];
    double sum = 0; long nr = 0;
#pragma omp parallel for num_threads(g->threads) reduction(+ : sum, nr)
    for (int i = 0; i < g->src.n; ++i) {
        const float* p = &g->src.pts[3 * (size_t)i];
        float q[3];
        for (int a = 0; a < 3; ++a) q[a] = Tf[4 * a] * p[0] + Tf[4 * a + 1] * p[1] + Tf[4 * a + 2] * p[2] + Tf[4 * a + 3];
        int j; float d2;
        g->tgt.tree.knn(q, 1, &j, &d2);
        if (j >= 0 && (double)d2 <= max_range) { sum += (double)d2; ++nr; }
    }
    return nr > 0 ? sum / (double)nr : std::numeric_limits<double>::max();
}

/* exact kNN (for kernel-level parity of the covariance front-end) */
void orc_knn(const float* xyz, int n, int k, int* out_idx)
{
    KdTree t;
    t.build(xyz, n);
#pragma omp parallel for schedule(guided, 8)
    for (int i = 0; i < n; ++i) {
        std::vector<float> d(k);
        t.knn(&xyz[3 * (size_t)i], k, &out_idx[(size_t)i * k], d.data());
    }
}

/* d^2 (float, the searches' operation chain) of every float-transformed source point to tgt[idx[i]] */
void orc_pair_d2(const float* src, int n, const double* T, const float* tgt, const int* idx, float* out)
{
    float Tf[16];
    for (int i = 0; i < 16; ++i) Tf[i] = (float)T[i];
    for (int i = 0; i < n; ++i) {
        if (idx[i] < 0) { out[i] = std::numeric_limits<float>::infinity(); continue; }
        const float* p = &src[3 * (size_t)i];
        float q[3];
        for (int a = 0; a < 3; ++a) q[a] = Tf[4 * a] * p[0] + Tf[4 * a + 1] * p[1] + Tf[4 * a + 2] * p[2] + Tf[4 * a + 3];
        const float* t = &tgt[3 * (size_t)idx[i]];
        const float dx = q[0] - t[0], dy = q[1] - t[1], dz = q[2] - t[2];
        out[i] = fmaf(dz, dz, fmaf(dy, dy, dx * dx));
    }
}

void orc_se3_exp(const double* a, double* T) { se3_exp(a, T); }

}  // extern "C"
