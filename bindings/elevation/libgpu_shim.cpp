// libgpu_shim.cpp -> libgpu.so: the ten functions the Mapping node's elevation_mapping package declares by hand and links
// from the reference's CUDA library (Mapping/src/elevation_mapping_periodical/elevation_mapping/src/ElevationMapping.cpp:44-50,
// src/sensor_processors/SensorProcessorBase.cpp:34, src/RobotMotionMapUpdater.cpp:18; defined in cuda/gpu_process.cu:938-1312),
// implemented over the C ABI (mrs_elev_*, include/mrslam_hip.h).  C++ linkage and the reference's exact parameter lists,
// so the mangled symbols are the ones ElevationMapping.cpp / SensorProcessorBase.cpp reference.  Like the reference's
// library (which keeps the map in __device__ globals) it holds ONE map per process.
//
//   g++ -O2 -std=c++17 -shared -fPIC -I<repo>/include [-I<eigen3>] bindings/elevation/libgpu_shim.cpp -o libgpu.so \
//       -L<repo>/mr_slam_amd -lmrslam_hip -Wl,-rpath,<repo>/mr_slam_amd
#if defined(__has_include) && __has_include(<Eigen/Core>)
#include <Eigen/Core>
#else
#include "eigen_min.hpp"
#endif

#include <cstdio>
#include <cstdlib>

#include "mrslam_hip.h"

namespace {
mrs_ctx* g_ctx = nullptr;
mrs_elev_map* g_map = nullptr;

void check(int st, const char* what)
{
    if (st != MRS_OK) {   // the reference's library reports CUDA errors on stderr and carries on; so does this one
        std::fprintf(stderr, "[libgpu(mrslam_hip)] %s: %s: %s\n", what, mrs_status_str(st), mrs_last_error());
    }
}

template <class M>
void row_major(const M& m, int rows, int cols, float* out)
{
    for (int r = 0; r < rows; ++r)
        for (int c = 0; c < cols; ++c) out[r * cols + c] = m(r, c);
}
}  // namespace

void Init_GPU_elevationmap(int length, float resolution, float h_mahalanobisDistanceThreshold_, float h_obstacle_threshold)
{
    if (!g_ctx) check(mrs_ctx_create(0, &g_ctx), "mrs_ctx_create");
    if (g_map) { mrs_elev_destroy(g_map); g_map = nullptr; }
    check(mrs_elev_create(g_ctx, length, resolution, h_mahalanobisDistanceThreshold_, h_obstacle_threshold, &g_map), "Init_GPU_elevationmap");
}

void Move(float* current_Position, float /*resolution*/, int /*length*/, float* h_central_coordinate, int* h_start_indice, float* position_shift)
{
    check(mrs_elev_move(g_map, current_Position, h_central_coordinate, h_start_indice, position_shift), "Move");
}

int Process_points(int* mapindex, float* point_x, float* point_y, float* point_z, float* point_var, float* point_x_ts, float* point_y_ts,
                   float* point_z_ts, Eigen::Matrix4f Transform, int point_num, double relativeLowerThreshold, double relativeUpperThreshold,
                   float min_r, float beam_a, float beam_c, Eigen::RowVector3f sensorJacobian, Eigen::Matrix3f rotationVariance,
                   Eigen::Matrix3f C_SB_transpose, Eigen::RowVector3f P_mul_C_BM_transpose, Eigen::Matrix3f B_r_BS_skew)
{
    float T[16], sj[3], rv[9], csb[9], pm[3], bs[9];   // the C ABI takes row-major floats
    row_major(Transform, 4, 4, T);
    row_major(sensorJacobian, 1, 3, sj);
    row_major(rotationVariance, 3, 3, rv);
    row_major(C_SB_transpose, 3, 3, csb);
    row_major(P_mul_C_BM_transpose, 1, 3, pm);
    row_major(B_r_BS_skew, 3, 3, bs);
    const int st = mrs_elev_process_points(g_map, point_num, point_x, point_y, point_z, T, relativeLowerThreshold, relativeUpperThreshold,
                                           min_r, beam_a, beam_c, sj, rv, csb, pm, bs, mapindex, point_var, point_x_ts, point_y_ts, point_z_ts);
    check(st, "Process_points");
    return 0;   // gpu_process.cu:1141
}

void Fuse(int /*length*/, int point_num, int* point_index, int* point_colorR, int* point_colorG, int* point_colorB, float* point_intensity,
          float* point_height, float* point_var)
{
    check(mrs_elev_fuse(g_map, point_num, point_index, point_colorR, point_colorG, point_colorB, point_intensity, point_height, point_var), "Fuse");
}

void Mapvar_update(int /*length*/, float var_update) { check(mrs_elev_mapvar_update(g_map, var_update), "Mapvar_update"); }

void Map_feature(int /*length*/, float* elevation, float* var, int* point_colorR, int* point_colorG, int* point_colorB, float* rough,
                 float* slope, float* traver, float* intensity)
{
    check(mrs_elev_map_feature(g_map, elevation, var, point_colorR, point_colorG, point_colorB, rough, slope, traver, intensity), "Map_feature");
}

void Raytracing(int /*length_*/) { check(mrs_elev_raytracing(g_map), "Raytracing"); }

void Map_optmove(float* opt_p, float height_update, float /*resolution*/, int /*length*/, float* opt_alignedPosition)
{
    check(mrs_elev_map_optmove(g_map, opt_p, height_update, opt_alignedPosition), "Map_optmove");
}

void Map_closeloop(float* update_position, float height_update, int /*length*/, float /*resolution*/)
{
    check(mrs_elev_map_closeloop(g_map, update_position, height_update), "Map_closeloop");
}
