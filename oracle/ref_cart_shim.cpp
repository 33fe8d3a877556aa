// oracle/ref_cart_shim.cpp -- TEST INFRASTRUCTURE ONLY.
// extern "C" handle around the reference's own Cartesian max-z rasteriser (module `voxelocc`), compiled from
//   /root/reference/LoopDetection/generate_bev_cython_binary/src/{kernel.cu, manager.cu, manager.hh}
// where they lie: oracle/Makefile rewrites the single <<<>>> launch of manager.cu into a host loop in a scratch
// copy (oracle/_ref/build/cart/manager_host.cpp, git-ignored, deleted again after linking) and supplies oracle/ref_cuda_host/cuda_runtime.h.
// Built into oracle/_ref/libref_cart.so.  Pins oracle/bev_oracle.c rows A3 / A4.
#include <manager_host.cpp>   // = the reference's manager.cu with the launch rewritten; pulls in kernel.cu, manager.hh

extern "C" {

// the Cython wrapper's sequence (generate_bev_cython_binary/wrapper.pyx:19-39): construct with three zeroed index
// arrays, transform(), retreive() into a zeroed output of 3*num_x*num_y*num_height*enough_large floats
void ref_cart_bev(float* xyz_soa, int n, int max_length, int max_height, int num_x, int num_y, int num_height,
                  int enough_large, float* out)
{
    int* zero = (int*)std::calloc((size_t)(n > 0 ? n : 1) * 3, sizeof(int));
    {
        GPUTransformer t(xyz_soa, n, zero, zero + n, zero + 2 * n, max_length, max_height, num_x, num_y, num_height, enough_large);
        t.transform();
        t.retreive(out);
    }   // ~GPUTransformer frees the four buffers a second time (tolerated by the stand-in, an error code on CUDA)
    std::free(zero);
}

// index math only (kernel.cu:14-61)
void ref_cart_indices(float* xyz_soa, int n, int max_length, int max_height, int num_x, int num_y, int num_height,
                      int* ix, int* iy, int* ih)
{
    REF_LAUNCH(point2gridmap, dim3((n + 255) / 256), dim3(256), xyz_soa, ix, iy, ih, n, max_length, max_height, num_x, num_y, num_height);
}

}  // extern "C"
