// Drives libgpu.so (bindings/elevation/libgpu_shim.cpp) through the declarations the reference's elevation_mapping package
// writes by hand (src/ElevationMapping.cpp:44-50, src/sensor_processors/SensorProcessorBase.cpp:34,
// src/RobotMotionMapUpdater.cpp:18 -- transcribed here, test scaffolding) and compares every output with the C ABI called
// directly on a second map with the same inputs.  Catches marshalling errors (Eigen column-major vs the ABI's row-major).
#include "../../bindings/elevation/eigen_min.hpp"
#include "mrslam_hip.h"

#include <cmath>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>

// GPU functions api (ElevationMapping.cpp:44-50)
void Move(float *current_Position, float resolution, int length, float *h_central_coordinate, int *h_start_indice, float *position_shift);
void Init_GPU_elevationmap(int length, float resolution, float h_mahalanobisDistanceThreshold_, float h_obstacle_threshold);
void Map_closeloop(float *update_position, float height_update, int length, float resolution);
void Raytracing(int length_);
void Fuse(int length, int point_num, int *point_index, int *point_colorR, int *point_colorG, int *point_colorB, float *point_intensity, float *point_height, float *point_var);
void Map_feature(int length, float *elevation, float *var, int *point_colorR, int *point_colorG, int *point_colorB, float *rough, float *slope, float *traver, float *intensity);
void Map_optmove(float *opt_p, float height_update, float resolution,  int length, float *opt_alignedPosition);
// SensorProcessorBase.cpp:34
int Process_points(int *mapindex, float *point_x, float *point_y, float *point_z, float *point_var, float *point_x_ts, float *point_y_ts, float *point_z_ts, Eigen::Matrix4f Transform, int point_num, double relativeLowerThreshold, double relativeUpperThreshold, float min_r, float beam_a, float beam_c, Eigen::RowVector3f sensorJacobian, Eigen::Matrix3f rotationVariance, Eigen::Matrix3f C_SB_transpose, Eigen::RowVector3f P_mul_C_BM_transpose, Eigen::Matrix3f B_r_BS_skew);
// RobotMotionMapUpdater.cpp:18
int Mapvar_update(int length, float update_var);

template <class T>
static bool same(const std::vector<T>& a, const std::vector<T>& b, const char* what)
{
    if (a.size() == b.size() && std::memcmp(a.data(), b.data(), a.size() * sizeof(T)) == 0) return true;
    std::printf("MISMATCH %s\n", what);
    return false;
}

int main()
{
    const int L = 120; const float res = 0.1f;
    const int n = 30000;
    std::mt19937 rng(3);
    std::uniform_real_distribution<float> ux(-5.f, 5.f), uy(-5.f, -0.8f), u01(0.f, 1.f);
    std::normal_distribution<float> nz(0.f, 0.02f);
    std::vector<float> x(n), y(n), z(n), inten(n);
    std::vector<int> cr(n), cg(n), cb(n);
    for (int i = 0; i < n; ++i) { x[i] = ux(rng); y[i] = uy(rng); z[i] = 0.1f * std::sin(x[i]) - 0.6f + nz(rng); inten[i] = u01(rng); cr[i] = i % 256; cg[i] = (3 * i) % 256; cb[i] = (7 * i) % 256; }
    // deliberately non-symmetric matrices
    Eigen::Matrix4f T; const float yaw = 0.3f;
    T(0,0)=std::cos(yaw); T(0,1)=-std::sin(yaw); T(0,2)=0.02f; T(0,3)=0.4f;
    T(1,0)=std::sin(yaw); T(1,1)=std::cos(yaw);  T(1,2)=-0.01f; T(1,3)=-0.2f;
    T(2,0)=-0.02f; T(2,1)=0.01f; T(2,2)=1.f; T(2,3)=0.9f; T(3,0)=0; T(3,1)=0; T(3,2)=0; T(3,3)=1.f;
    Eigen::Matrix3f rv, csb, bs; Eigen::RowVector3f sj, pm;
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) { rv(r,c) = 1e-4f * (1 + r + 2 * c); csb(r,c) = (r == c) + 0.01f * (r - 2 * c); bs(r,c) = 0.03f * (r * 3 + c) * ((r + c) % 2 ? -1 : 1); }
    sj(0,0)=0.1f; sj(0,1)=-0.2f; sj(0,2)=1.f; pm(0,0)=0.05f; pm(0,1)=0.02f; pm(0,2)=1.f;
    float Tr[16], rvr[9], csbr[9], bsr[9], sjr[3] = {sj(0,0), sj(0,1), sj(0,2)}, pmr[3] = {pm(0,0), pm(0,1), pm(0,2)};
    for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) Tr[4 * r + c] = T(r,c);
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) { rvr[3*r+c] = rv(r,c); csbr[3*r+c] = csb(r,c); bsr[3*r+c] = bs(r,c); }

    // --- through libgpu.so
    Init_GPU_elevationmap(L, res, 2.0f, 0.6f);
    float pos[3] = {0.35f, -0.12f, 0.9f}, cc[2], sh[2]; int si[2];
    Move(pos, res, L, cc, si, sh);
    std::vector<int> mi(n); std::vector<float> var(n), xt(n), yt(n), zt(n);
    std::vector<float> x1 = x, y1 = y, z1 = z;
    Process_points(mi.data(), x1.data(), y1.data(), z1.data(), var.data(), xt.data(), yt.data(), zt.data(), T, n, -2.0, 3.0, 0.02f, 0.003f, 0.01f, sj, rv, csb, pm, bs);
    Fuse(L, n, mi.data(), cr.data(), cg.data(), cb.data(), inten.data(), zt.data(), var.data());
    Mapvar_update(L, 1e-4f);
    const int cells = L * L;
    std::vector<float> el(cells), va(cells), ro(cells), sl(cells), tr(cells), in2(cells); std::vector<int> r2(cells), g2(cells), b2(cells);
    Map_feature(L, el.data(), va.data(), r2.data(), g2.data(), b2.data(), ro.data(), sl.data(), tr.data(), in2.data());
    Raytracing(L);
    float optp[2] = {0.45f, -0.05f}, al[2]; Map_optmove(optp, 0.02f, res, L, al);
    float up[2] = {0.1f, 0.1f}; Map_closeloop(up, -0.01f, L, res);

    // --- the C ABI directly, same inputs
    mrs_ctx* ctx = nullptr; mrs_elev_map* m = nullptr;
    if (mrs_ctx_create(0, &ctx) != MRS_OK || mrs_elev_create(ctx, L, res, 2.0f, 0.6f, &m) != MRS_OK) { std::printf("create failed: %s\n", mrs_last_error()); return 2; }
    float cc_[2], sh_[2]; int si_[2];
    mrs_elev_move(m, pos, cc_, si_, sh_);
    std::vector<int> mi_(n); std::vector<float> var_(n), xt_(n), yt_(n), zt_(n), x2 = x, y2 = y, z2 = z;
    mrs_elev_process_points(m, n, x2.data(), y2.data(), z2.data(), Tr, -2.0, 3.0, 0.02f, 0.003f, 0.01f, sjr, rvr, csbr, pmr, bsr, mi_.data(), var_.data(), xt_.data(), yt_.data(), zt_.data());
    mrs_elev_fuse(m, n, mi_.data(), cr.data(), cg.data(), cb.data(), inten.data(), zt_.data(), var_.data());
    mrs_elev_mapvar_update(m, 1e-4f);
    std::vector<float> el_(cells), va_(cells), ro_(cells), sl_(cells), tr_(cells), in_(cells); std::vector<int> r_(cells), g_(cells), b_(cells);
    mrs_elev_map_feature(m, el_.data(), va_.data(), r_.data(), g_.data(), b_.data(), ro_.data(), sl_.data(), tr_.data(), in_.data());
    mrs_elev_raytracing(m);
    float al_[2]; mrs_elev_map_optmove(m, optp, 0.02f, al_);
    mrs_elev_map_closeloop(m, up, -0.01f);

    bool ok = cc[0] == cc_[0] && cc[1] == cc_[1] && si[0] == si_[0] && si[1] == si_[1] && sh[0] == sh_[0] && sh[1] == sh_[1] && al[0] == al_[0] && al[1] == al_[1];
    if (!ok) std::printf("MISMATCH move/optmove\n");
    ok &= same(mi, mi_, "map_index") & same(var, var_, "var") & same(xt, xt_, "x_ts") & same(yt, yt_, "y_ts") & same(zt, zt_, "z_ts") & same(x1, x2, "x");
    ok &= same(el, el_, "elevation") & same(va, va_, "variance") & same(ro, ro_, "rough") & same(sl, sl_, "slope") & same(tr, tr_, "traver") &
          same(in2, in_, "intensity") & same(r2, r_, "colorR") & same(g2, g_, "colorG") & same(b2, b_, "colorB");
    long hit = 0; for (int v : mi) hit += v >= 0;
    long filled = 0; for (float v : el) filled += v > -9.f;
    std::printf("libgpu.so == C ABI: %s (points in map %ld / %d, cells with elevation %ld / %d)\n", ok ? "yes" : "NO", hit, n, filled, cells);
    return (ok && hit > n / 4 && filled > 100) ? 0 : 1;
}
