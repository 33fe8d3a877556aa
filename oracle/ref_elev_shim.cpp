// oracle/ref_elev_shim.cpp -- TEST INFRASTRUCTURE ONLY.
// Builds the reference's elevation_mapping/cuda/gpu_process.cu (Mapping/src/elevation_mapping_periodical/elevation_mapping/
// cuda/gpu_process.cu, the source of its libgpu.so: SURVEY.md 8(f) row N3) for the HOST, from where it lies under
// /root/reference, and exposes its nine functions with the calling convention of oracle/elev_oracle.cpp so that one test
// session can be replayed on both.  oracle/Makefile rewrites every `kernel<<<grid, block>>>(args)` of a scratch copy into
// REF_LAUNCH (threads one after another in gid order = the sequential reading of the kernels' racy updates);
// ref_cuda_host/cuda_runtime.h supplies cudaMalloc / cudaMemcpy[To|From]Symbol / atomics, ref_cuda_host/Eigen/Core the few
// fixed-size Eigen operations used (neither CUDA nor Eigen is in this image).  Nothing of the reference is copied into the
// repository: the scratch copy lives under oracle/_ref/build/ (git-ignored, deleted again after linking).
// The reference keeps ONE map in module-scope variables: handles are not supported, create() re-initialises it.
#include "gpu_process_host.cpp"

namespace {
int g_length = 0;
float g_resolution = 0.0f;

Eigen::Matrix3f mat3(const float* a)
{
    Eigen::Matrix3f m;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) m(i, j) = a[3 * i + j];   // row-major input
    return m;
}
Eigen::RowVector3f row3(const float* a) { Eigen::RowVector3f r; r(0, 0) = a[0]; r(0, 1) = a[1]; r(0, 2) = a[2]; return r; }
}  // namespace

extern "C" {

void ref_elev_create(int length, float resolution, float mahal_thr, float obstacle_thr)
{
    g_length = length;
    g_resolution = resolution;
    Init_GPU_elevationmap(length, resolution, mahal_thr, obstacle_thr);
}

void ref_elev_move(const float* pos3, float* central, int* start, float* aligned_shift)
{
    float p[3] = {pos3[0], pos3[1], pos3[2]};
    Move(p, g_resolution, g_length, central, start, aligned_shift);
}

void ref_elev_process_points(int n, float* px, float* py, float* pz, const float* T, double lower, double upper, float min_r,
                             float beam_a, float beam_c, const float* sensor_jacobian, const float* rotation_variance,
                             const float* c_sb_transpose, const float* p_mul_c_bm_transpose, const float* b_r_bs_skew,
                             int* map_index, float* var, float* x_ts, float* y_ts, float* z_ts)
{
    Eigen::Matrix4f tf;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) tf(i, j) = T[4 * i + j];
    Process_points(map_index, px, py, pz, var, x_ts, y_ts, z_ts, tf, n, lower, upper, min_r, beam_a, beam_c, row3(sensor_jacobian),
                   mat3(rotation_variance), mat3(c_sb_transpose), row3(p_mul_c_bm_transpose), mat3(b_r_bs_skew));
}

void ref_elev_fuse(int n, const int* index, const int* cR, const int* cG, const int* cB, const float* inten, const float* h,
                   const float* v)
{
    Fuse(g_length, n, const_cast<int*>(index), const_cast<int*>(cR), const_cast<int*>(cG), const_cast<int*>(cB),
         const_cast<float*>(inten), const_cast<float*>(h), const_cast<float*>(v));
}

void ref_elev_mapvar_update(float v) { Mapvar_update(g_length, v); }

void ref_elev_map_feature(float* elevation, float* var, int* cR, int* cG, int* cB, float* rough, float* slope, float* traver,
                          float* intensity)
{
    Map_feature(g_length, elevation, var, cR, cG, cB, rough, slope, traver, intensity);
}

void ref_elev_raytracing() { Raytracing(g_length); }

void ref_elev_map_optmove(const float* opt_p, float height_update, float* aligned)
{
    float p[2] = {opt_p[0], opt_p[1]};
    Map_optmove(p, height_update, g_resolution, g_length, aligned);
}

void ref_elev_map_closeloop(const float* update_pos, float height_update)
{
    float p[2] = {update_pos[0], update_pos[1]};
    Map_closeloop(p, height_update, g_length, g_resolution);
}

// state readback: which = 0 lowest, 1 elevation, 2 variance, 3 intensity, 4 traversability (the module-scope maps)
void ref_elev_get(int which, float* out)
{
    const float* src = which == 0 ? map_lowest : which == 1 ? map_elevation : which == 2 ? map_variance : which == 3 ? map_intensity : map_traver;
    std::memcpy(out, src, sizeof(float) * g_length * g_length);
}

void ref_elev_get_frame(float* central, int* start)
{
    central[0] = central_coordinate[0]; central[1] = central_coordinate[1];
    start[0] = start_indice[0]; start[1] = start_indice[1];
}

}  // extern "C"
