/*
 * oracle/elev_oracle.cpp -- TEST INFRASTRUCTURE ONLY (not product code).
 *
 * Sequential CPU restatement of the elevation-mapping GPU library of the Mapping workspace
 * (SURVEY.md section 8(f) row N3):
 *   Mapping/src/elevation_mapping_periodical/elevation_mapping/cuda/gpu_process.cu
 *     Init_GPU_elevationmap :938-990   G_Init_map :198-214
 *     Move                  :1000-1074 G_Clear_map :255-279, G_Clear_allmap :216-230
 *     Process_points        :1076-1137 G_pointsprocess :384-454 (PointsToIndex :308-331, PointsToMapIndex :333-359)
 *     Fuse                  :1148-1187 G_fuse :477-535
 *     Mapvar_update         :1139-1146 G_Mapvar_update :538-545
 *     Map_feature           :1248-1296 G_Mapfeature :547-668, computerEigenvalue :64-186
 *     Raytracing            :1298-1312 G_Raytracing :706-893, G_Clear_maplowest :232-239
 *     Map_optmove :1210-1227, Map_closeloop :1229-1246, G_update_mapheight :1189-1197
 * PINNED to the reference source itself: oracle/Makefile builds gpu_process.cu for the host (oracle/ref_elev_shim.cpp,
 * oracle/_ref/libref_elev.so: launches run thread by thread in gid order, stand-ins for the CUDA runtime and the few Eigen
 * operations), and tests/test_oracle_elev.py replays one multi-frame session on both: indices, layers, variances bit-identical,
 * slope / traversability 6e-7.  Two reference behaviours are racy on a GPU and are given their sequential-in-point-order
 * reading here, in the host build and in the HIP code:
 *   - the `map_lowest` update in G_pointsprocess (atomicMin followed by a non-atomic "+3 sigma" bump):
 *     read as  lowest = (h <= lowest) ? h + 3*var : lowest, points in input order;
 *   - G_fuse is already sequential per cell (each cell thread walks all points in order).
 * Quirks reproduced on purpose: the point filter keeps only points with y <= -1 outside the 1.5 m box
 * (:394-397), Raytracing returns before its final test for cells on the robot's row/column (:770-804),
 * `robot_index` is an int (:730,:742-751), d_min_elevation only uses x indices (:690-704).
 */
#include <cmath>
#include <cstring>
#include <vector>

namespace {

struct ElevMap {
    int L = 0;
    float res = 0, mahal_thr = 0, obstacle_thr = 0;
    std::vector<float> lowest, elevation, variance, intensity, traver;
    std::vector<int> cr, cg, cb;
    float central[2] = {0, 0};
    int start[2] = {0, 0};
    float sensor_z = 0;
};

int points_to_index(const ElevMap& m, float px, float py, bool storage)
{
    const float sx = px - m.central[0], sy = py - m.central[1];
    int ix, iy;
    if (m.L % 2 == 0) {
        ix = (int)((float)(m.L / 2) - sx / m.res);
        iy = (int)((float)(m.L / 2) - sy / m.res);
    } else {
        ix = m.L / 2 - static_cast<int>(sx / m.res + 0.5 * (sx > 0 ? 1 : -1));
        iy = m.L / 2 - static_cast<int>(sy / m.res + 0.5 * (sy > 0 ? 1 : -1));
    }
    if (!(ix >= 0 && ix < m.L && iy >= 0 && iy < m.L)) return -1;
    if (!storage) return ix * m.L + iy;
    return ((ix + m.start[0]) % m.L) * m.L + (iy + m.start[1]) % m.L;
}

int index_to_range(int index, int L)
{
    if (index < 0) index += ((-index / L) + 1) * L;
    return index % L;
}

float position_to_range(float p, float shift, float res)
{
    const int pi = (int)std::round(p / res), si = (int)std::round(shift / res);
    return (pi + si) * res;
}

void clear_region(ElevMap& m, int start, int shift, bool row)
{
    for (int i = 0; i < m.L * shift; ++i) {
        const int c = row ? start * m.L + i : i / shift * m.L + i % shift + start;
        m.intensity[c] = 0; m.elevation[c] = -10; m.variance[c] = -10;
        m.cr[c] = m.cg[c] = m.cb[c] = 0;
    }
}

// computerEigenvalue (:64-186): classical Jacobi in float, returns the eigenvector of the smallest eigenvalue
void smallest_eigvec_f(float* a, float* out)
{
    float v[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    const float eps = 0.01f;
    int count = 0;
    while (true) {
        float mx = a[1];
        int row = 0, col = 1;
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) {
                const float d = std::fabs(a[i * 3 + j]);
                if (i != j && d > mx) { mx = d; row = i; col = j; }
            }
        if (mx < eps) break;
        if (count > 30) break;
        ++count;
        const float app = a[row * 3 + row], apq = a[row * 3 + col], aqq = a[col * 3 + col];
        const float ang = 0.5f * std::atan2(-2 * apq, aqq - app);
        const float sn = std::sin(ang), cs = std::cos(ang), s2 = std::sin(2 * ang), c2 = std::cos(2 * ang);
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

}  // namespace

extern "C" {

void* orc_elev_create(int length, float resolution, float mahal_thr, float obstacle_thr)
{
    ElevMap* m = new ElevMap();
    m->L = length; m->res = resolution; m->mahal_thr = mahal_thr; m->obstacle_thr = obstacle_thr;
    const size_t n = (size_t)length * length;
    m->intensity.assign(n, 0); m->elevation.assign(n, -10); m->variance.assign(n, -10);
    m->lowest.assign(n, 100); m->traver.assign(n, -10);
    m->cr.assign(n, 0); m->cg.assign(n, 0); m->cb.assign(n, 0);
    return m;
}
void orc_elev_destroy(void* h) { delete static_cast<ElevMap*>(h); }

void orc_elev_move(void* h, const float* pos3, float* central, int* start, float* aligned_shift)
{
    ElevMap& m = *static_cast<ElevMap*>(h);
    m.sensor_z = pos3[2];
    float pshift[2] = {pos3[0] - m.central[0], pos3[1] - m.central[1]};
    int ishift[2];
    for (int i = 0; i < 2; ++i) {
        ishift[i] = static_cast<int>(pshift[i] / m.res + 0.5 * (pshift[i] > 0 ? 1 : -1));
        aligned_shift[i] = (float)ishift[i] * m.res;
    }
    for (int i = 0; i < 2; ++i) {
        if (ishift[i] != 0) {
            if (ishift[i] >= m.L) {
                for (size_t c = 0; c < m.elevation.size(); ++c) {
                    m.intensity[c] = 0; m.elevation[c] = -10; m.variance[c] = -10; m.traver[c] = -10;
                    m.cr[c] = m.cg[c] = m.cb[c] = 0;
                }
            } else {
                const int sign = ishift[i] > 0 ? 1 : -1;
                const int s0 = m.start[i] - (sign > 0 ? 1 : 0);
                const int e0 = s0 + sign - ishift[i];
                const int nc = std::abs(ishift[i]);
                int idx = index_to_range(sign < 0 ? s0 : e0, m.L);
                if (idx + nc <= m.L) clear_region(m, idx, nc, i == 0);
                else {
                    const int first = m.L - idx;
                    clear_region(m, idx, first, i == 0);
                    clear_region(m, 0, nc - first, i == 0);
                }
            }
        }
        m.start[i] = index_to_range(m.start[i] - ishift[i], m.L);
        m.central[i] = position_to_range(m.central[i], aligned_shift[i], m.res);
    }
    central[0] = m.central[0]; central[1] = m.central[1];
    start[0] = m.start[0]; start[1] = m.start[1];
}

/* T: row-major 4x4; 3-vectors and row-major 3x3 matrices as plain floats */
void orc_elev_process_points(void* h, int n, const float* px, const float* py, const float* pz, const float* T, double lower, double upper,
                             float min_r, float beam_a, float beam_c, const float* sensorJacobian, const float* rotationVariance,
                             const float* C_SB_transpose, const float* P_mul_C_BM_transpose, const float* B_r_BS_skew,
                             int* map_index, float* var, float* xts, float* yts, float* zts)
{
    ElevMap& m = *static_cast<ElevMap*>(h);
    for (int i = 0; i < n; ++i) {
        const float x = px[i], y = py[i], z = pz[i];
        const float height = T[8] * x + T[9] * y + T[10] * z + T[11];
        int flag = 0;
        if ((x > -1.5 && x < 1.5 && y > -1.5 && y < 1.5) || (y > -1 && y < 1) || y > 0) flag = 1;
        if ((height > lower && height < upper) && flag == 0) {
            xts[i] = T[0] * x + T[1] * y + T[2] * z + T[3];
            yts[i] = T[4] * x + T[5] * y + T[6] * z + T[7];
            zts[i] = height;
            const float dist = std::sqrt(x * x + y * y + z * z);
            const float vn = std::pow(min_r, 2.0f);
            const float vl = std::pow(beam_c + beam_a * dist, 2.0f);
            // skew(C_SB^T p) + B_r_BS_skew
            float q[3];
            for (int r = 0; r < 3; ++r) q[r] = C_SB_transpose[3 * r] * x + C_SB_transpose[3 * r + 1] * y + C_SB_transpose[3 * r + 2] * z;
            float S[9] = {0, -q[2], q[1], q[2], 0, -q[0], -q[1], q[0], 0};
            for (int k = 0; k < 9; ++k) S[k] += B_r_BS_skew[k];
            float J[3];
            for (int c = 0; c < 3; ++c) J[c] = P_mul_C_BM_transpose[0] * S[c] + P_mul_C_BM_transpose[1] * S[3 + c] + P_mul_C_BM_transpose[2] * S[6 + c];
            float A1[3];
            for (int c = 0; c < 3; ++c) A1[c] = J[0] * rotationVariance[c] + J[1] * rotationVariance[3 + c] + J[2] * rotationVariance[6 + c];
            float hv = A1[0] * J[0] + A1[1] * J[1] + A1[2] * J[2];
            const float sv[3] = {vl, vl, vn};
            float B1[3];
            for (int c = 0; c < 3; ++c) B1[c] = sensorJacobian[c] * sv[c];
            hv += B1[0] * sensorJacobian[0] + B1[1] * sensorJacobian[1] + B1[2] * sensorJacobian[2];
            var[i] = hv;
            const int gi = points_to_index(m, xts[i], yts[i], false);
            map_index[i] = points_to_index(m, xts[i], yts[i], true);
            if (gi != -1 && height <= m.lowest[gi]) m.lowest[gi] = height + 3 * hv;
        } else {
            map_index[i] = -1;   // the kernel also overwrites its DEVICE copy of x, y, z with -1; Process_points never copies
                                 // that back (gpu_process.cu:1119-1123), so the caller's arrays stay as they were
            xts[i] = yts[i] = zts[i] = -1;
            var[i] = -1;
        }
    }
}

void orc_elev_fuse(void* h, int n, const int* index, const int* cR, const int* cG, const int* cB, const float* inten,
                   const float* ph, const float* pv)
{
    ElevMap& m = *static_cast<ElevMap*>(h);
    const int cells = m.L * m.L;
    for (int i = 0; i < n; ++i) {
        const int c = index[i];
        if (c < 0 || c >= cells || ph[i] == -1) continue;
        const bool colored = cR[i] != 0 && cG[i] != 0 && cB[i] != 0 && inten[i] != 0;
        auto take_color = [&]() { if (colored) { m.intensity[c] = inten[i]; m.cr[c] = cR[i]; m.cg[c] = cG[i]; m.cb[c] = cB[i]; } };
        if (m.elevation[c] == -10) {
            m.elevation[c] = ph[i]; m.variance[c] = pv[i]; take_color();
        } else {
            const float md = std::fabs(ph[i] - m.elevation[c]) / std::sqrt(m.variance[c]);
            if (md > 5) {
                if (m.elevation[c] < ph[i]) { m.elevation[c] = ph[i]; m.variance[c] = pv[i]; take_color(); }
            } else {
                m.elevation[c] = (m.variance[c] * ph[i] + pv[i] * m.elevation[c]) / (m.variance[c] + pv[i]);
                m.variance[c] = (pv[i] * m.variance[c]) / (pv[i] + m.variance[c]);
                take_color();
            }
        }
    }
    for (int c = 0; c < cells; ++c)
        if (m.variance[c] < 0.0001) m.variance[c] = 0.0001f;
}

void orc_elev_mapvar_update(void* h, float v)
{
    ElevMap& m = *static_cast<ElevMap*>(h);
    for (auto& x : m.variance) if (x != -10) x += v;
}

void orc_elev_map_feature(void* h, float* elevation, float* var, int* cR, int* cG, int* cB, float* rough, float* slope,
                          float* traver, float* intensity)
{
    ElevMap& m = *static_cast<ElevMap*>(h);
    const int L = m.L;
    for (int idx = 0; idx < L * L; ++idx) {
        elevation[idx] = m.elevation[idx]; cR[idx] = m.cr[idx]; cG[idx] = m.cg[idx]; cB[idx] = m.cb[idx];
        intensity[idx] = m.intensity[idx]; var[idx] = m.variance[idx];
        if (m.elevation[idx] == -10) continue;   // rough/slope/traver left untouched like the reference (uninitialised there)
        const int cx = idx / L, cy = idx % L;
        float X[25], Y[25], Z[25], mxs = 0, mys = 0, mzs = 0;
        int pn = 0;
        for (int i = -2; i < 3; ++i)
            for (int j = -2; j < 3; ++j) {
                const int ex = (cx + L - m.start[0]) % L + i, ey = (cy + L - m.start[1]) % L + j;
                if (ex >= 0 && ex < L && ey >= 0 && ey < L) {
                    const int px = (cx + i + L) % L, py = (cy + j + L) % L;
                    const float sz = m.elevation[px * L + py];
                    if (sz != -10) {
                        X[pn] = px * m.res; Y[pn] = py * m.res; Z[pn] = sz;
                        mxs += X[pn]; mys += Y[pn]; mzs += Z[pn];
                        ++pn;
                    }
                }
            }
        if (pn > 7) {
            mxs /= pn; mys /= pn; mzs /= pn;
            float P[9] = {0};
            for (int i = 0; i < pn; ++i) {
                P[0] += (X[i] - mxs) * (X[i] - mxs); P[4] += (Y[i] - mys) * (Y[i] - mys); P[8] += (Z[i] - mzs) * (Z[i] - mzs);
                P[1] += (X[i] - mxs) * (Y[i] - mys); P[2] += (X[i] - mxs) * (Z[i] - mzs); P[5] += (Y[i] - mys) * (Z[i] - mzs);
                P[3] = P[1]; P[6] = P[2]; P[7] = P[5];
            }
            float nv[3];
            smallest_eigvec_f(P, nv);
            const float sl = nv[2] > 0 ? std::acos(nv[2]) : std::acos(-nv[2]);
            const float ro = std::fabs(m.elevation[idx] - mzs);
            const float tr = 0.5 * (1.0 - sl / 0.6) + 0.5 * (1.0 - (ro / 0.2));
            slope[idx] = sl; rough[idx] = ro; traver[idx] = tr; m.traver[idx] = tr;
        } else {
            slope[idx] = 0; rough[idx] = 0; traver[idx] = -10; m.traver[idx] = -10;
        }
    }
}

void orc_elev_raytracing(void* h)
{
    ElevMap& m = *static_cast<ElevMap*>(h);
    const int L = m.L;
    auto valid = [&](int x, int y) { return m.lowest[x * L + y] != 10; };
    auto min_ele = [&](int x, int y, int ox, float rx) {
        const float x1 = (float)(x - ox), x2 = (float)x - rx;
        const float low = m.lowest[x * L + y];
        return low + (m.sensor_z - low) / x2 * x1;
    };
    std::vector<float> elev = m.elevation;  // cells only write their own entry: a snapshot is equivalent
    for (int i = 0; i < L * L; ++i) {
        if (!(m.traver[i] < m.obstacle_thr && m.elevation[i] != -10)) continue;
        const int cx = i / L, cy = i % L;
        const int ob[2] = {(cx + L - m.start[0]) % L, (cy + L - m.start[1]) % L};
        const float oe = m.elevation[i];
        int cur[2] = {ob[0], ob[1]};
        int robot = L % 2 == 0 ? (int)(float)(L / 2 - 0.5) : (int)(float)(L / 2);
        float inc[2] = {(float)(ob[0] - robot), (float)(ob[1] - robot)};
        const int ix = inc[0] > 0 ? 1 : (inc[0] == 0 ? 0 : -1), iy = inc[1] > 0 ? 1 : (inc[1] == 0 ? 0 : -1);
        float restrict_e = oe;
        if (ix == 0 || iy == 0) continue;  // the reference returns before its final test on these branches
        const float dis = std::sqrt(inc[0] * inc[0] + inc[1] * inc[1]);
        const float dir[2] = {inc[0] / dis, inc[1] / dis};
        float thr;
        if (std::fabs(inc[0]) > std::fabs(inc[1])) thr = std::sqrt(0.5 * 0.5 + std::pow(0.5 / inc[0] * inc[1], 2));
        else thr = std::sqrt(0.5 * 0.5 + std::pow(0.5 / inc[1] * inc[0], 2));
        float bx = (float)ix / 2, by = (float)iy / 2;
        float dnx = bx / dir[0], dny = by / dir[1], later = 0;
        while (cur[0] >= 0 && cur[0] < L && cur[1] >= 0 && cur[1] < L) {
            const float dn = dnx > dny ? dny : dnx;   // equal -> dnx
            if (dn - later > thr && cur[0] != ob[0] && cur[1] != ob[1] && valid(cur[0], cur[1])) {
                const float e = min_ele(cur[0], cur[1], ob[0], (float)robot);
                if (e < restrict_e) restrict_e = e;
            }
            if (dnx > dny) { cur[1] += iy; by += (float)iy; later = dny; dny = by / dir[1]; }
            else if (dnx < dny) { cur[0] += ix; bx += (float)ix; later = dnx; dnx = bx / dir[0]; }
            else { cur[0] += ix; cur[1] += iy; bx += (float)ix; by += (float)iy; later = dnx; dnx = bx / dir[0]; dny = by / dir[1]; }
        }
        if (oe - 3 * std::sqrt(m.variance[i]) > restrict_e) elev[i] = -10;
    }
    m.elevation = elev;
    for (auto& x : m.lowest) x = 10;
}

void orc_elev_map_optmove(void* h, const float* opt_p, float height_update, float* aligned)
{
    ElevMap& m = *static_cast<ElevMap*>(h);
    for (int i = 0; i < 2; ++i) {
        const float ps = opt_p[i] - m.central[i];
        const int is = static_cast<int>(ps / m.res + 0.5 * (ps > 0 ? 1 : -1));
        aligned[i] = m.central[i] + m.res * is;
    }
    m.central[0] = aligned[0]; m.central[1] = aligned[1];
    for (auto& e : m.elevation) if (e != -10) e += height_update;
}

void orc_elev_map_closeloop(void* h, const float* update_pos, float height_update)
{
    ElevMap& m = *static_cast<ElevMap*>(h);
    for (int i = 0; i < 2; ++i) {
        const float ps = update_pos[i] - m.central[i];
        const int is = static_cast<int>(ps / m.res + 0.5 * (ps > 0 ? 1 : -1));
        m.central[i] = position_to_range(m.central[i], (float)is * m.res, m.res);
    }
    for (auto& e : m.elevation) if (e != -10) e += height_update;
}

/* state readback for the tests: which = 0 lowest, 1 elevation, 2 variance, 3 intensity, 4 traver */
void orc_elev_get(void* h, int which, float* out)
{
    ElevMap& m = *static_cast<ElevMap*>(h);
    const std::vector<float>* v[5] = {&m.lowest, &m.elevation, &m.variance, &m.intensity, &m.traver};
    std::memcpy(out, v[which]->data(), v[which]->size() * sizeof(float));
}
void orc_elev_get_frame(void* h, float* central, int* start)
{
    ElevMap& m = *static_cast<ElevMap*>(h);
    central[0] = m.central[0]; central[1] = m.central[1]; start[0] = m.start[0]; start[1] = m.start[1];
}

}  // extern "C"
