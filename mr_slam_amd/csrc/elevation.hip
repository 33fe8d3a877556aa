// elevation.hip -- 2.5-D elevation mapping on the GPU: the nine entry
// points of the reference's libgpu.so (Mapping/src/elevation_mapping_periodical/elevation_mapping/cuda/
// gpu_process.cu:938-1312), declared by hand in src/ElevationMapping.cpp:44-50 and
// src/sensor_processors/SensorProcessorBase.cpp:34.
//
// Behaviour reproduced (never copied) kernel by kernel; file:line cites are in the matching functions.
// Differences in HOW (not what):
//   * G_fuse lets every cell thread walk ALL points (O(cells * N)); here the points are bucketed per
//     cell with a stable radix sort and each cell walks only its own points, in input order: identical
//     result, O(N log N);
//   * the racy `map_lowest` update of G_pointsprocess is given its sequential-in-input-order reading with
//     the same bucketing (see oracle/elev_oracle.cpp header);
//   * state lives in a handle instead of __device__ globals, per-call buffers come from the library's scratch cache
//     (capi.hip) instead of cudaMalloc/cudaFree per call, every call returns a status.
#include <hipcub/hipcub.hpp>

#include <cmath>

#include "common.hpp"

struct mrs_elev_map {
    mrs_ctx* ctx = nullptr;
    int L = 0;
    float res = 0, mahal_thr = 0, obstacle_thr = 0;
    float* lowest = nullptr; float* elevation = nullptr; float* variance = nullptr; float* intensity = nullptr; float* traver = nullptr;
    int* cr = nullptr; int* cg = nullptr; int* cb = nullptr;
    float central[2] = {0, 0};
    int start[2] = {0, 0};
    float sensor_z = 0;
};

namespace {

struct Frame { int L; float res; float cx, cy; int sx, sy; };

// PointsToIndex (:308-331) / PointsToMapIndex (:333-359)
__device__ __forceinline__ int points_to_index(const Frame& f, float px, float py, bool storage)
{
    const float sx = px - f.cx, sy = py - f.cy;
    int ix, iy;
    if (f.L % 2 == 0) {
        ix = (int)((float)(f.L / 2) - sx / f.res);
        iy = (int)((float)(f.L / 2) - sy / f.res);
    } else {
        ix = f.L / 2 - (int)((double)(sx / f.res) + 0.5 * (sx > 0 ? 1 : -1));
        iy = f.L / 2 - (int)((double)(sy / f.res) + 0.5 * (sy > 0 ? 1 : -1));
    }
    if (!(ix >= 0 && ix < f.L && iy >= 0 && iy < f.L)) return -1;
    if (!storage) return ix * f.L + iy;
    return ((ix + f.sx) % f.L) * f.L + (iy + f.sy) % f.L;
}

__global__ void k_elev_fill(float* lowest, float* elevation, float* variance, float* intensity, float* traver, int* cr, int* cg,
                            int* cb, int cells, int mode /* 0 init (:198-214), 1 clear all (:216-230) */)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= cells) return;
    intensity[i] = 0; elevation[i] = -10; variance[i] = -10; traver[i] = -10;
    cr[i] = 0; cg[i] = 0; cb[i] = 0;
    if (mode == 0) lowest[i] = 100;
}

// G_Clear_map (:255-279)
__global__ void k_elev_clear_region(float* elevation, float* variance, float* intensity, int* cr, int* cg, int* cb, int L, int start,
                                    int shift, int row)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= L * shift) return;
    const int c = row ? start * L + i : i / shift * L + i % shift + start;
    intensity[c] = 0; elevation[c] = -10; variance[c] = -10; cr[c] = 0; cg[c] = 0; cb[c] = 0;
}

struct PointParams {
    float T[12];
    double lower, upper;
    float min_r, beam_a, beam_c;
    float sensorJacobian[3], rotationVariance[9], C_SB_t[9], P_mul[3], B_skew[9];
};

// G_pointsprocess (:384-454) without the map_lowest update (done per cell afterwards)
__global__ void k_elev_points(Frame f, PointParams p, int n, float* px, float* py, float* pz, int* map_index, int* grid_index,
                              float* var, float* xts, float* yts, float* zts)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float x = px[i], y = py[i], z = pz[i];
    const float height = p.T[8] * x + p.T[9] * y + p.T[10] * z + p.T[11];
    int flag = 0;
    if (((double)x > -1.5 && (double)x < 1.5 && (double)y > -1.5 && (double)y < 1.5) || (y > -1 && y < 1) || y > 0) flag = 1;
    if (((double)height > p.lower && (double)height < p.upper) && flag == 0) {
        const float tx = p.T[0] * x + p.T[1] * y + p.T[2] * z + p.T[3];
        const float ty = p.T[4] * x + p.T[5] * y + p.T[6] * z + p.T[7];
        xts[i] = tx; yts[i] = ty; zts[i] = height;
        const float dist = sqrtf(x * x + y * y + z * z);
        const float vn = p.min_r * p.min_r;
        const float bl = p.beam_c + p.beam_a * dist;
        const float vl = bl * bl;
        float q[3];
#pragma unroll
        for (int r = 0; r < 3; ++r) q[r] = p.C_SB_t[3 * r] * x + p.C_SB_t[3 * r + 1] * y + p.C_SB_t[3 * r + 2] * z;
        float S[9] = {0, -q[2], q[1], q[2], 0, -q[0], -q[1], q[0], 0};
#pragma unroll
        for (int k = 0; k < 9; ++k) S[k] += p.B_skew[k];
        float J[3], A1[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) J[c] = p.P_mul[0] * S[c] + p.P_mul[1] * S[3 + c] + p.P_mul[2] * S[6 + c];
#pragma unroll
        for (int c = 0; c < 3; ++c) A1[c] = J[0] * p.rotationVariance[c] + J[1] * p.rotationVariance[3 + c] + J[2] * p.rotationVariance[6 + c];
        float hv = A1[0] * J[0] + A1[1] * J[1] + A1[2] * J[2];
        const float sv[3] = {vl, vl, vn};
        float B1[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) B1[c] = p.sensorJacobian[c] * sv[c];
        hv += B1[0] * p.sensorJacobian[0] + B1[1] * p.sensorJacobian[1] + B1[2] * p.sensorJacobian[2];
        var[i] = hv;
        grid_index[i] = points_to_index(f, tx, ty, false);
        map_index[i] = points_to_index(f, tx, ty, true);
    } else {
        map_index[i] = -1; grid_index[i] = -1;
        px[i] = -1; py[i] = -1; pz[i] = -1;
        xts[i] = -1; yts[i] = -1; zts[i] = -1;
        var[i] = -1;
    }
}

__global__ void k_keys_from_index(const int* idx, int n, int cells, unsigned* keys, int* vals)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int c = idx[i];
    keys[i] = (c >= 0 && c < cells) ? (unsigned)c : 0xffffffffu;
    vals[i] = i;
}

// one lane per bucket head: lowest = (h <= lowest) ? h + 3 var : lowest, points in input order (:441-448)
__global__ void k_elev_lowest(const unsigned* keys, const int* perm, int n, const float* zts, const float* var, float* lowest)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const unsigned k = keys[i];
    if (k == 0xffffffffu || (i > 0 && keys[i - 1] == k)) return;
    float cur = lowest[k];
    for (int j = i; j < n && keys[j] == k; ++j) {
        const int pi = perm[j];
        if (zts[pi] <= cur) cur = zts[pi] + 3 * var[pi];
    }
    lowest[k] = cur;
}

// G_fuse (:477-535) per bucket
__global__ void k_elev_fuse(const unsigned* keys, const int* perm, int n, const int* cR, const int* cG, const int* cB, const float* inten,
                            const float* ph, const float* pv, float* elevation, float* variance, float* intensity, int* cr, int* cg, int* cb)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const unsigned k = keys[i];
    if (k == 0xffffffffu || (i > 0 && keys[i - 1] == k)) return;
    float e = elevation[k], v = variance[k], it = intensity[k];
    int r = cr[k], g = cg[k], b = cb[k];
    for (int j = i; j < n && keys[j] == k; ++j) {
        const int pi = perm[j];
        const float h = ph[pi], hv = pv[pi];
        if (h == -1) continue;
        const bool colored = cR[pi] != 0 && cG[pi] != 0 && cB[pi] != 0 && inten[pi] != 0;
        bool take = false;
        if (e == -10) { e = h; v = hv; take = true; }
        else {
            const float md = fabsf(h - e) / sqrtf(v);
            if (md > 5) { if (e < h) { e = h; v = hv; take = true; } }
            else { e = (v * h + hv * e) / (v + hv); v = (hv * v) / (hv + v); take = true; }
        }
        if (take && colored) { it = inten[pi]; r = cR[pi]; g = cG[pi]; b = cB[pi]; }
    }
    elevation[k] = e; variance[k] = v; intensity[k] = it; cr[k] = r; cg[k] = g; cb[k] = b;
}

__global__ void k_elev_var_floor(float* variance, int cells)  // :530-531
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < cells && (double)variance[i] < 0.0001) variance[i] = 0.0001f;
}

__global__ void k_elev_var_add(float* variance, int cells, float v)  // G_Mapvar_update :538-545
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < cells && variance[i] != -10) variance[i] += v;
}

__global__ void k_elev_height_add(float* elevation, int cells, float v)  // G_update_mapheight :1189-1197
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < cells && elevation[i] != -10) elevation[i] += v;
}

// computerEigenvalue (:64-186)
__device__ void smallest_eigvec_f(float* a, float* out)
{
    float v[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    int count = 0;
    while (true) {
        float mx = a[1];
        int row = 0, col = 1;
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) {
                const float d = fabsf(a[i * 3 + j]);
                if (i != j && d > mx) { mx = d; row = i; col = j; }
            }
        if (mx < 0.01f) break;
        if (count > 30) break;
        ++count;
        const float app = a[row * 3 + row], apq = a[row * 3 + col], aqq = a[col * 3 + col];
        const float ang = 0.5f * atan2f(-2 * apq, aqq - app);
        const float sn = sinf(ang), cs = cosf(ang), s2 = sinf(2 * ang), c2 = cosf(2 * ang);
        a[row * 3 + row] = app * cs * cs + aqq * sn * sn + 2 * apq * cs * sn;
        a[col * 3 + col] = app * sn * sn + aqq * cs * cs - 2 * apq * cs * sn;
        a[row * 3 + col] = 0.5f * (aqq - app) * s2 + apq * c2;
        a[col * 3 + row] = a[row * 3 + col];
        for (int i = 0; i < 3; ++i)
            if (i != col && i != row) {
                const int u = i * 3 + row, w = i * 3 + col;
                const float t = a[u];
                a[u] = a[w] * sn + t * cs;
                a[w] = a[w] * cs - t * sn;
            }
        for (int j = 0; j < 3; ++j)
            if (j != col && j != row) {
                const int u = row * 3 + j, w = col * 3 + j;
                const float t = a[u];
                a[u] = a[w] * sn + t * cs;
                a[w] = a[w] * cs - t * sn;
            }
        for (int i = 0; i < 3; ++i) {
            const int u = i * 3 + row, w = i * 3 + col;
            const float t = v[u];
            v[u] = v[w] * sn + t * cs;
            v[w] = v[w] * cs - t * sn;
        }
    }
    int mn = 0;
    float mv = a[0];
    for (int i = 1; i < 3; ++i)
        if (mv > a[i * 3 + i]) { mv = a[i * 3 + i]; mn = i; }
    for (int i = 0; i < 3; ++i) out[i] = v[mn + 3 * i];
}

// G_Mapfeature (:547-668).  rough / slope / traver of empty cells are written as 0 / 0 / -10 here
// (the reference leaves freshly cudaMalloc'ed memory there).
__global__ void k_elev_feature(Frame f, const float* elevation, const float* variance, const float* intensity, const int* cr,
                               const int* cg, const int* cb, float* traver_map, float* o_elev, float* o_var, int* o_r, int* o_g,
                               int* o_b, float* o_rough, float* o_slope, float* o_traver, float* o_int)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int L = f.L;
    if (idx >= L * L) return;
    o_elev[idx] = elevation[idx]; o_r[idx] = cr[idx]; o_g[idx] = cg[idx]; o_b[idx] = cb[idx];
    o_int[idx] = intensity[idx]; o_var[idx] = variance[idx];
    if (elevation[idx] == -10) { o_rough[idx] = 0; o_slope[idx] = 0; o_traver[idx] = -10; return; }
    const int cx = idx / L, cy = idx % L;
    float X[25], Y[25], Z[25], mxs = 0, mys = 0, mzs = 0;
    int pn = 0;
    for (int i = -2; i < 3; ++i)
        for (int j = -2; j < 3; ++j) {
            const int ex = (cx + L - f.sx) % L + i, ey = (cy + L - f.sy) % L + j;
            if (ex >= 0 && ex < L && ey >= 0 && ey < L) {
                const int px = (cx + i + L) % L, py = (cy + j + L) % L;
                const float sz = elevation[px * L + py];
                if (sz != -10) {
                    X[pn] = px * f.res; Y[pn] = py * f.res; Z[pn] = sz;
                    mxs += X[pn]; mys += Y[pn]; mzs += Z[pn];
                    ++pn;
                }
            }
        }
    if (pn > 7) {
        mxs /= pn; mys /= pn; mzs /= pn;
        float P[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        for (int i = 0; i < pn; ++i) {
            P[0] += (X[i] - mxs) * (X[i] - mxs); P[4] += (Y[i] - mys) * (Y[i] - mys); P[8] += (Z[i] - mzs) * (Z[i] - mzs);
            P[1] += (X[i] - mxs) * (Y[i] - mys); P[2] += (X[i] - mxs) * (Z[i] - mzs); P[5] += (Y[i] - mys) * (Z[i] - mzs);
        }
        P[3] = P[1]; P[6] = P[2]; P[7] = P[5];
        float nv[3];
        smallest_eigvec_f(P, nv);
        const float sl = nv[2] > 0 ? acosf(nv[2]) : acosf(-nv[2]);
        const float ro = fabsf(elevation[idx] - mzs);
        const float tr = (float)(0.5 * (1.0 - (double)sl / 0.6) + 0.5 * (1.0 - ((double)ro / 0.2)));
        o_slope[idx] = sl; o_rough[idx] = ro; o_traver[idx] = tr; traver_map[idx] = tr;
    } else {
        o_slope[idx] = 0; o_rough[idx] = 0; o_traver[idx] = -10; traver_map[idx] = -10;
    }
}

// G_Raytracing (:706-893) + G_Clear_maplowest (:232-239, separate launch)
__global__ void k_elev_raytrace(Frame f, float obstacle_thr, float sensor_z, const float* traver, const float* lowest,
                                const float* variance, float* elevation)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int L = f.L;
    if (i >= L * L) return;
    if (!(traver[i] < obstacle_thr && elevation[i] != -10)) return;
    const int cx = i / L, cy = i % L;
    const int ob0 = (cx + L - f.sx) % L, ob1 = (cy + L - f.sy) % L;
    const float oe = elevation[i];
    int c0 = ob0, c1 = ob1;
    const int robot = L % 2 == 0 ? (int)(float)((double)(L / 2) - 0.5) : (int)(float)(L / 2);
    const float inc0 = (float)(ob0 - robot), inc1 = (float)(ob1 - robot);
    const int ix = inc0 > 0 ? 1 : (inc0 == 0 ? 0 : -1), iy = inc1 > 0 ? 1 : (inc1 == 0 ? 0 : -1);
    if (ix == 0 || iy == 0) return;  // reference: early return on the robot's row / column
    float restrict_e = oe;
    const float dis = sqrtf(inc0 * inc0 + inc1 * inc1);
    const float d0 = inc0 / dis, d1 = inc1 / dis;
    float thr;
    if (fabsf(inc0) > fabsf(inc1)) thr = (float)sqrt(0.5 * 0.5 + pow(0.5 / (double)inc0 * (double)inc1, 2.0));
    else thr = (float)sqrt(0.5 * 0.5 + pow(0.5 / (double)inc1 * (double)inc0, 2.0));
    float bx = (float)ix / 2, by = (float)iy / 2;
    float dnx = bx / d0, dny = by / d1, later = 0;
    while (c0 >= 0 && c0 < L && c1 >= 0 && c1 < L) {
        const float dn = dnx > dny ? dny : dnx;
        if (dn - later > thr && c0 != ob0 && c1 != ob1 && lowest[c0 * L + c1] != 10) {
            const float x1 = (float)(c0 - ob0), x2 = (float)c0 - (float)robot;
            const float low = lowest[c0 * L + c1];
            const float e = low + (sensor_z - low) / x2 * x1;
            if (e < restrict_e) restrict_e = e;
        }
        if (dnx > dny) { c1 += iy; by += (float)iy; later = dny; dny = by / d1; }
        else if (dnx < dny) { c0 += ix; bx += (float)ix; later = dnx; dnx = bx / d0; }
        else { c0 += ix; c1 += iy; bx += (float)ix; by += (float)iy; later = dnx; dnx = bx / d0; dny = by / d1; }
    }
    if (oe - 3 * sqrtf(variance[i]) > restrict_e) elevation[i] = -10;
}

__global__ void k_fill_f(float* p, int n, float v)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

inline int nb(int n) { return (n + 255) / 256; }

Frame frame_of(const mrs_elev_map* m) { return Frame{m->L, m->res, m->central[0], m->central[1], m->start[0], m->start[1]}; }

int index_to_range(int index, int L)  // :915-920
{
    if (index < 0) index += ((-index / L) + 1) * L;
    return index % L;
}
float position_to_range(float p, float shift, float res)  // :992-998
{
    const int pi = (int)std::round(p / res), si = (int)std::round(shift / res);
    return (pi + si) * res;
}

// stable bucket sort of point indices by cell
int bucket(const int* d_idx, int n, int cells, hipStream_t s, mrs::Scratch& keys_out, mrs::Scratch& vals_out)
{
    mrs::Scratch keys_in, vals_in, tmp;
    int st;
    if ((st = keys_in.alloc((size_t)n * 4, s)) != MRS_OK) return st;
    if ((st = vals_in.alloc((size_t)n * 4, s)) != MRS_OK) return st;
    if ((st = keys_out.alloc((size_t)n * 4, s)) != MRS_OK) return st;
    if ((st = vals_out.alloc((size_t)n * 4, s)) != MRS_OK) return st;
    hipLaunchKernelGGL(k_keys_from_index, dim3(nb(n)), dim3(256), 0, s, d_idx, n, cells, keys_in.as<unsigned>(), vals_in.as<int>());
    size_t bytes = 0;
    MRS_HIP_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, keys_in.as<unsigned>(), keys_out.as<unsigned>(), vals_in.as<int>(),
                                                   vals_out.as<int>(), n, 0, 32, s));
    if ((st = tmp.alloc(bytes, s)) != MRS_OK) return st;
    MRS_HIP_TRY(hipcub::DeviceRadixSort::SortPairs(tmp.p, bytes, keys_in.as<unsigned>(), keys_out.as<unsigned>(), vals_in.as<int>(),
                                                   vals_out.as<int>(), n, 0, 32, s));
    return MRS_OK;
}

}  // namespace

extern "C" {

// Init_GPU_elevationmap (:938-990)
int mrs_elev_create(mrs_ctx* ctx, int32_t length, float resolution, float mahalanobis_threshold, float obstacle_threshold,
                    mrs_elev_map** out)
{
    MRS_REQUIRE(ctx && out, "null pointer");
    MRS_REQUIRE(length > 0 && length <= 8192 && resolution > 0.0f, "bad map geometry");
    *out = nullptr;
    MRS_HIP_TRY(hipSetDevice(ctx->device));
    mrs_elev_map* m = new mrs_elev_map();
    m->ctx = ctx; m->L = length; m->res = resolution; m->mahal_thr = mahalanobis_threshold; m->obstacle_thr = obstacle_threshold;
    const size_t cells = (size_t)length * length;
    float** fp[5] = {&m->lowest, &m->elevation, &m->variance, &m->intensity, &m->traver};
    int** ip[3] = {&m->cr, &m->cg, &m->cb};
    for (auto p : fp) MRS_HIP_TRY(hipMalloc(p, cells * 4));
    for (auto p : ip) MRS_HIP_TRY(hipMalloc(p, cells * 4));
    hipLaunchKernelGGL(k_elev_fill, dim3(nb((int)cells)), dim3(256), 0, nullptr, m->lowest, m->elevation, m->variance, m->intensity,
                       m->traver, m->cr, m->cg, m->cb, (int)cells, 0);
    MRS_HIP_TRY(hipDeviceSynchronize());
    *out = m;
    return MRS_OK;
}

int mrs_elev_destroy(mrs_elev_map* m)
{
    if (!m) return MRS_OK;
    (void)hipSetDevice(m->ctx->device);
    float* fp[5] = {m->lowest, m->elevation, m->variance, m->intensity, m->traver};
    int* ip[3] = {m->cr, m->cg, m->cb};
    for (auto p : fp) if (p) (void)hipFree(p);
    for (auto p : ip) if (p) (void)hipFree(p);
    delete m;
    return MRS_OK;
}

// Move (:1000-1074)
int mrs_elev_move(mrs_elev_map* m, const float* h_position3, float* h_central2, int32_t* h_start2, float* h_aligned_shift2)
{
    MRS_REQUIRE(m && h_position3 && h_central2 && h_start2 && h_aligned_shift2, "null pointer");
    MRS_HIP_TRY(hipSetDevice(m->ctx->device));
    m->sensor_z = h_position3[2];
    const float pshift[2] = {h_position3[0] - m->central[0], h_position3[1] - m->central[1]};
    int ishift[2];
    for (int i = 0; i < 2; ++i) {
        ishift[i] = static_cast<int>(pshift[i] / m->res + 0.5 * (pshift[i] > 0 ? 1 : -1));
        h_aligned_shift2[i] = (float)ishift[i] * m->res;
    }
    const int L = m->L, cells = L * L;
    auto clear = [&](int start, int shift, bool row) {
        hipLaunchKernelGGL(k_elev_clear_region, dim3(nb(L * shift)), dim3(256), 0, nullptr, m->elevation, m->variance, m->intensity,
                           m->cr, m->cg, m->cb, L, start, shift, row ? 1 : 0);
    };
    for (int i = 0; i < 2; ++i) {
        if (ishift[i] != 0) {
            if (ishift[i] >= L) {
                hipLaunchKernelGGL(k_elev_fill, dim3(nb(cells)), dim3(256), 0, nullptr, m->lowest, m->elevation, m->variance,
                                   m->intensity, m->traver, m->cr, m->cg, m->cb, cells, 1);
            } else {
                const int sign = ishift[i] > 0 ? 1 : -1;
                const int s0 = m->start[i] - (sign > 0 ? 1 : 0);
                const int e0 = s0 + sign - ishift[i];
                const int nc = std::abs(ishift[i]);
                const int idx = index_to_range(sign < 0 ? s0 : e0, L);
                if (idx + nc <= L) clear(idx, nc, i == 0);
                else {
                    const int first = L - idx;
                    clear(idx, first, i == 0);
                    clear(0, nc - first, i == 0);
                }
            }
        }
        m->start[i] = index_to_range(m->start[i] - ishift[i], L);
        m->central[i] = position_to_range(m->central[i], h_aligned_shift2[i], m->res);
    }
    MRS_HIP_TRY(hipGetLastError());
    MRS_HIP_TRY(hipDeviceSynchronize());
    h_central2[0] = m->central[0]; h_central2[1] = m->central[1];
    h_start2[0] = m->start[0]; h_start2[1] = m->start[1];
    return MRS_OK;
}

// Process_points (:1076-1137): host arrays in / out like the reference
int mrs_elev_process_points(mrs_elev_map* m, int32_t n, const float* h_x, const float* h_y, const float* h_z, const float* h_transform16,
                            double lower, double upper, float min_r, float beam_a, float beam_c, const float* h_sensorJacobian3,
                            const float* h_rotationVariance9, const float* h_C_SB_transpose9, const float* h_P_mul_C_BM_transpose3,
                            const float* h_B_r_BS_skew9, int32_t* h_map_index, float* h_var, float* h_x_ts, float* h_y_ts, float* h_z_ts)
{
    MRS_REQUIRE(m && h_x && h_y && h_z && h_transform16 && h_map_index && h_var && h_x_ts && h_y_ts && h_z_ts, "null pointer");
    MRS_REQUIRE(h_sensorJacobian3 && h_rotationVariance9 && h_C_SB_transpose9 && h_P_mul_C_BM_transpose3 && h_B_r_BS_skew9, "null pointer");
    MRS_REQUIRE(n > 0, "n must be positive");
    MRS_HIP_TRY(hipSetDevice(m->ctx->device));
    hipStream_t s = nullptr;
    PointParams p;
    memcpy(p.T, h_transform16, sizeof(p.T));
    p.lower = lower; p.upper = upper; p.min_r = min_r; p.beam_a = beam_a; p.beam_c = beam_c;
    memcpy(p.sensorJacobian, h_sensorJacobian3, 12); memcpy(p.rotationVariance, h_rotationVariance9, 36);
    memcpy(p.C_SB_t, h_C_SB_transpose9, 36); memcpy(p.P_mul, h_P_mul_C_BM_transpose3, 12); memcpy(p.B_skew, h_B_r_BS_skew9, 36);
    mrs::Scratch buf;
    int st = buf.alloc((size_t)n * 4 * 9, s);
    if (st != MRS_OK) return st;
    float* dx = buf.as<float>(); float* dy = dx + n; float* dz = dy + n; float* dv = dz + n; float* dxt = dv + n; float* dyt = dxt + n; float* dzt = dyt + n;
    int* dmi = reinterpret_cast<int*>(dzt + n); int* dgi = dmi + n;
    MRS_HIP_TRY(hipMemcpyAsync(dx, h_x, (size_t)n * 4, hipMemcpyHostToDevice, s));
    MRS_HIP_TRY(hipMemcpyAsync(dy, h_y, (size_t)n * 4, hipMemcpyHostToDevice, s));
    MRS_HIP_TRY(hipMemcpyAsync(dz, h_z, (size_t)n * 4, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_elev_points, dim3(nb(n)), dim3(256), 0, s, frame_of(m), p, n, dx, dy, dz, dmi, dgi, dv, dxt, dyt, dzt);
    mrs::Scratch keys, perm;
    st = bucket(dgi, n, m->L * m->L, s, keys, perm);
    if (st != MRS_OK) return st;
    hipLaunchKernelGGL(k_elev_lowest, dim3(nb(n)), dim3(256), 0, s, keys.as<unsigned>(), perm.as<int>(), n, dzt, dv, m->lowest);
    MRS_HIP_TRY(hipGetLastError());
    MRS_HIP_TRY(hipMemcpyAsync(h_var, dv, (size_t)n * 4, hipMemcpyDeviceToHost, s));
    MRS_HIP_TRY(hipMemcpyAsync(h_x_ts, dxt, (size_t)n * 4, hipMemcpyDeviceToHost, s));
    MRS_HIP_TRY(hipMemcpyAsync(h_y_ts, dyt, (size_t)n * 4, hipMemcpyDeviceToHost, s));
    MRS_HIP_TRY(hipMemcpyAsync(h_z_ts, dzt, (size_t)n * 4, hipMemcpyDeviceToHost, s));
    MRS_HIP_TRY(hipMemcpyAsync(h_map_index, dmi, (size_t)n * 4, hipMemcpyDeviceToHost, s));
    MRS_HIP_TRY(hipStreamSynchronize(s));
    return MRS_OK;
}

// Fuse (:1148-1187)
int mrs_elev_fuse(mrs_elev_map* m, int32_t n, const int32_t* h_index, const int32_t* h_colorR, const int32_t* h_colorG,
                  const int32_t* h_colorB, const float* h_intensity, const float* h_height, const float* h_var)
{
    MRS_REQUIRE(m && h_index && h_colorR && h_colorG && h_colorB && h_intensity && h_height && h_var, "null pointer");
    MRS_REQUIRE(n > 0, "n must be positive");
    MRS_HIP_TRY(hipSetDevice(m->ctx->device));
    hipStream_t s = nullptr;
    mrs::Scratch buf;
    int st = buf.alloc((size_t)n * 4 * 7, s);
    if (st != MRS_OK) return st;
    int* di = buf.as<int>(); int* dr = di + n; int* dg = dr + n; int* db = dg + n;
    float* dit = reinterpret_cast<float*>(db + n); float* dh = dit + n; float* dv = dh + n;
    const void* src[7] = {h_index, h_colorR, h_colorG, h_colorB, h_intensity, h_height, h_var};
    void* dst[7] = {di, dr, dg, db, dit, dh, dv};
    for (int k = 0; k < 7; ++k) MRS_HIP_TRY(hipMemcpyAsync(dst[k], src[k], (size_t)n * 4, hipMemcpyHostToDevice, s));
    mrs::Scratch keys, perm;
    st = bucket(di, n, m->L * m->L, s, keys, perm);
    if (st != MRS_OK) return st;
    hipLaunchKernelGGL(k_elev_fuse, dim3(nb(n)), dim3(256), 0, s, keys.as<unsigned>(), perm.as<int>(), n, dr, dg, db, dit, dh, dv,
                       m->elevation, m->variance, m->intensity, m->cr, m->cg, m->cb);
    hipLaunchKernelGGL(k_elev_var_floor, dim3(nb(m->L * m->L)), dim3(256), 0, s, m->variance, m->L * m->L);
    MRS_HIP_TRY(hipGetLastError());
    MRS_HIP_TRY(hipStreamSynchronize(s));
    return MRS_OK;
}

int mrs_elev_mapvar_update(mrs_elev_map* m, float var_update)  // Mapvar_update (:1139-1146)
{
    MRS_REQUIRE(m, "null handle");
    MRS_HIP_TRY(hipSetDevice(m->ctx->device));
    hipLaunchKernelGGL(k_elev_var_add, dim3(nb(m->L * m->L)), dim3(256), 0, nullptr, m->variance, m->L * m->L, var_update);
    MRS_HIP_TRY(hipGetLastError());
    return MRS_OK;
}

// Map_feature (:1248-1296)
int mrs_elev_map_feature(mrs_elev_map* m, float* h_elevation, float* h_var, int32_t* h_colorR, int32_t* h_colorG, int32_t* h_colorB,
                         float* h_rough, float* h_slope, float* h_traver, float* h_intensity)
{
    MRS_REQUIRE(m && h_elevation && h_var && h_colorR && h_colorG && h_colorB && h_rough && h_slope && h_traver && h_intensity, "null pointer");
    MRS_HIP_TRY(hipSetDevice(m->ctx->device));
    hipStream_t s = nullptr;
    const int cells = m->L * m->L;
    mrs::Scratch buf;
    int st = buf.alloc((size_t)cells * 4 * 9, s);
    if (st != MRS_OK) return st;
    float* oe = buf.as<float>(); float* ov = oe + cells; float* oro = ov + cells; float* osl = oro + cells; float* otr = osl + cells; float* oin = otr + cells;
    int* orr = reinterpret_cast<int*>(oin + cells); int* og = orr + cells; int* ob = og + cells;
    hipLaunchKernelGGL(k_elev_feature, dim3(nb(cells)), dim3(256), 0, s, frame_of(m), m->elevation, m->variance, m->intensity, m->cr, m->cg,
                       m->cb, m->traver, oe, ov, orr, og, ob, oro, osl, otr, oin);
    MRS_HIP_TRY(hipGetLastError());
    void* dst[9] = {h_elevation, h_var, h_rough, h_slope, h_traver, h_intensity, h_colorR, h_colorG, h_colorB};
    const void* src[9] = {oe, ov, oro, osl, otr, oin, orr, og, ob};
    for (int k = 0; k < 9; ++k) MRS_HIP_TRY(hipMemcpyAsync(dst[k], src[k], (size_t)cells * 4, hipMemcpyDeviceToHost, s));
    MRS_HIP_TRY(hipStreamSynchronize(s));
    return MRS_OK;
}

// Raytracing (:1298-1312)
int mrs_elev_raytracing(mrs_elev_map* m)
{
    MRS_REQUIRE(m, "null handle");
    MRS_HIP_TRY(hipSetDevice(m->ctx->device));
    const int cells = m->L * m->L;
    hipLaunchKernelGGL(k_elev_raytrace, dim3(nb(cells)), dim3(256), 0, nullptr, frame_of(m), m->obstacle_thr, m->sensor_z, m->traver,
                       m->lowest, m->variance, m->elevation);
    hipLaunchKernelGGL(k_fill_f, dim3(nb(cells)), dim3(256), 0, nullptr, m->lowest, cells, 10.0f);
    MRS_HIP_TRY(hipGetLastError());
    MRS_HIP_TRY(hipDeviceSynchronize());
    return MRS_OK;
}

// Map_optmove (:1210-1227)
int mrs_elev_map_optmove(mrs_elev_map* m, const float* h_opt_p2, float height_update, float* h_aligned2)
{
    MRS_REQUIRE(m && h_opt_p2 && h_aligned2, "null pointer");
    MRS_HIP_TRY(hipSetDevice(m->ctx->device));
    for (int i = 0; i < 2; ++i) {
        const float ps = h_opt_p2[i] - m->central[i];
        const int is = static_cast<int>(ps / m->res + 0.5 * (ps > 0 ? 1 : -1));
        h_aligned2[i] = m->central[i] + m->res * is;
    }
    m->central[0] = h_aligned2[0]; m->central[1] = h_aligned2[1];
    hipLaunchKernelGGL(k_elev_height_add, dim3(nb(m->L * m->L)), dim3(256), 0, nullptr, m->elevation, m->L * m->L, height_update);
    MRS_HIP_TRY(hipGetLastError());
    return MRS_OK;
}

// Map_closeloop (:1229-1246)
int mrs_elev_map_closeloop(mrs_elev_map* m, const float* h_update_position2, float height_update)
{
    MRS_REQUIRE(m && h_update_position2, "null pointer");
    MRS_HIP_TRY(hipSetDevice(m->ctx->device));
    for (int i = 0; i < 2; ++i) {
        const float ps = h_update_position2[i] - m->central[i];
        const int is = static_cast<int>(ps / m->res + 0.5 * (ps > 0 ? 1 : -1));
        m->central[i] = position_to_range(m->central[i], (float)is * m->res, m->res);
    }
    hipLaunchKernelGGL(k_elev_height_add, dim3(nb(m->L * m->L)), dim3(256), 0, nullptr, m->elevation, m->L * m->L, height_update);
    MRS_HIP_TRY(hipGetLastError());
    return MRS_OK;
}

// state readback (tests / debugging): which = 0 lowest, 1 elevation, 2 variance, 3 intensity, 4 traver
int mrs_elev_get_layer(mrs_elev_map* m, int32_t which, float* h_out)
{
    MRS_REQUIRE(m && h_out, "null pointer");
    MRS_REQUIRE(which >= 0 && which < 5, "which must be in [0, 4]");
    MRS_HIP_TRY(hipSetDevice(m->ctx->device));
    const float* src[5] = {m->lowest, m->elevation, m->variance, m->intensity, m->traver};
    MRS_HIP_TRY(hipDeviceSynchronize());
    MRS_HIP_TRY(hipMemcpy(h_out, src[which], (size_t)m->L * m->L * 4, hipMemcpyDeviceToHost));
    return MRS_OK;
}

int mrs_elev_get_frame(mrs_elev_map* m, float* h_central2, int32_t* h_start2)
{
    MRS_REQUIRE(m && h_central2 && h_start2, "null pointer");
    h_central2[0] = m->central[0]; h_central2[1] = m->central[1];
    h_start2[0] = m->start[0]; h_start2[1] = m->start[1];
    return MRS_OK;
}

}  // extern "C"
