// oracle/ref_symbolic_shim.cpp -- TEST INFRASTRUCTURE ONLY.
// extern "C" handle around the reference's analytic Radon checker, compiled in place from
//   /root/reference/LoopDetection/torch-radon/src/{symbolic.cpp,log.cpp}, include/*.h
// (never copied).  parameter_classes.cu is a CUDA translation unit whose ProjectionCfg
// constructor is plain host code; the stub below supplies that one constructor
// (field order per include/parameter_classes.h:37-67).
#include "symbolic.h"

ProjectionCfg::ProjectionCfg(int dc_u, float ds_u, int dc_v, float ds_v, float sd, float dd, float pi,
                             float iz, int pt)
    : det_count_u(dc_u), det_spacing_u(ds_u), det_count_v(dc_v), det_spacing_v(ds_v), s_dist(sd),
      d_dist(dd), pitch(pi), initial_z(iz), projection_type(pt) {}

extern "C" {
void* ref_sym_create(float h, float w) { return new SymbolicFunction(h, w); }
void ref_sym_destroy(void* f) { delete static_cast<SymbolicFunction*>(f); }
void ref_sym_add_gaussian(void* f, float k, float cx, float cy, float a, float b)
{ static_cast<SymbolicFunction*>(f)->add_gaussian(k, cx, cy, a, b); }
void ref_sym_add_ellipse(void* f, float k, float cx, float cy, float r, float a)
{ static_cast<SymbolicFunction*>(f)->add_ellipse(k, cx, cy, r, a); }
void ref_sym_discretize(void* f, float* data, int h, int w)
{ static_cast<SymbolicFunction*>(f)->discretize(data, h, w); }
void ref_sym_forward(void* f, int det_count, float det_spacing, const float* angles, int n_angles, float* sino)
{
    ProjectionCfg proj(det_count, det_spacing);
    proj.n_angles = n_angles;
    symbolic_forward(*static_cast<SymbolicFunction*>(f), proj, angles, n_angles, sino);
}
}
