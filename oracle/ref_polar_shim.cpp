// oracle/ref_polar_shim.cpp -- TEST INFRASTRUCTURE ONLY.
// extern "C" handle around the reference's own CPU polar rasteriser, compiled from the
// sources where they lie (never copied):
//   /root/reference/LoopDetection/src/disco_ros/tools/multi-layer-polar-cpu/cython/src/
//   {kernel.cpp, manager.cpp, manager.hh}
// Built by oracle/Makefile into oracle/_ref/libref_polar.so (git-ignored, travels to the
// GPU box with the snapshot).  Used to pin oracle/bev_oracle.c and as the "reference"
// kind of CPU baseline in bench.py.
#include <manager.cpp>   // pulls in kernel.cpp and manager.hh via -I<reference src dir>

extern "C" {

// transform() + retreive() in one call; out must hold 3*R*S*H*enough_large zeroed floats.
void ref_polar_bev(float* xyz_soa, int n, int max_length, int max_height, int num_ring,
                   int num_sector, int num_height, int enough_large, float* out)
{
    GPUTransformer t(xyz_soa, n, nullptr, nullptr, nullptr, max_length, max_height,
                     num_ring, num_sector, num_height, enough_large);
    t.transform();
    t.retreive(out);
}

// index math only (kernel.cpp:40-77), for per-point parity checks.
void ref_polar_indices(float* xyz_soa, int n, int max_length, int max_height, int num_ring,
                       int num_sector, int num_height, int* ring, int* sector, int* height)
{
    point2gridmap(xyz_soa, ring, sector, height, n, max_length, max_height, num_ring,
                  num_sector, num_height);
}

}  // extern "C"
