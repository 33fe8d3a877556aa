// oracle/ref_feat_shim.cpp -- TEST INFRASTRUCTURE ONLY.
// extern "C" handle around the reference's own feature BEV + point-feature kernel (module `voxelfeat`), compiled from
//   /root/reference/LoopDetection/generate_bev_pointfeat_cython/src/{kernel.cu, manager.cu, manager.hh}
// where they lie (same recipe as ref_cart_shim.cpp: launch rewritten into a host loop in oracle/_ref/build/feat/).
// The kernels run their "threads" one after another in gid order, so the racy per-cell max of kernel.cu:151-158
// becomes its sequential reading (= the true per-cell maximum of the positive values).
// Built into oracle/_ref/libref_feat.so.  Pins oracle/bev_oracle.c row A5 and oracle/pointfeat_oracle.py (N1).
#include <manager_host.cpp>

extern "C" {

// wrapper.pyx:22-39: GPUTransformer(point[F*n] channel-major, ...), transform(), retreive()
void ref_feat_bev(float* pts_cm, int n, int max_length, int max_height, int num_x, int num_y, int num_height, int featsize,
                  float* out)
{
    int* zero = (int*)std::calloc((size_t)(n > 0 ? n : 1) * 3, sizeof(int));
    GPUTransformer t(pts_cm, n, zero, zero + n, zero + 2 * n, max_length, max_height, num_x, num_y, num_height, featsize);
    t.transform();
    t.retreive(out);
    std::free(zero);
}

// wrapper.pyx:46-59: GPUFeatureExtractor(point[3n] row-major, n, 13, k, neighbours[n*k], eigens[n*5]).get_features()
void ref_point_features(float* pts, int n, int featsize, int k, int* neighbors, float* eigens, float* out)
{
    GPUFeatureExtractor e(pts, n, featsize, k, neighbors, eigens);
    e.get_features(out);
}

}  // extern "C"
