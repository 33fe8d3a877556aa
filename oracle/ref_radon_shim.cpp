// oracle/ref_radon_shim.cpp -- TEST INFRASTRUCTURE ONLY.
// The reference's own Radon forward kernel (LoopDetection/torch-radon/src/forward.cu:12-124, `radon_forward_kernel`), run on the
// host: oracle/Makefile cuts the kernel (and the two configuration constructors, src/parameter_classes.cu:6-23) out of the
// reference files into scratch includes under oracle/_ref/build/radon/ (git-ignored, deleted again after linking), this file supplies the CUDA names around
// it (ref_cuda_host/cuda_runtime.h: threadIdx / blockIdx ...; ref_cuda_host/ref_texture.h: the texture fetch as the CUDA guide
// documents it) and runs one "thread" per (ray, angle, image).  The configuration classes are the reference's own header.
#include <stdlib.h>
#include <math.h>

#include "cuda_runtime.h"
#include "ref_texture.h"
#include "parameter_classes.h"      // the reference's include/parameter_classes.h (+ defines.h)

#include "param_ctors.inc"
#include "forward_kernel.inc"

extern "C" {

void ref_radon_set_weight_bits(int bits) { ref_tex_weight_bits = bits; }

// img [B][H][W] -> sino [B][n_angles][det]; Volume2D defaults (centre 0, voxel size 1: torch_radon/volumes.py:6-21),
// parallel beam, one channel (ParallelBeam.forward: torch_radon/radon.py:62-87)
void ref_radon_parallel(const float* img, int B, int H, int W, const float* angles, int n_angles, int det, float spacing, float* sino)
{
    const ref_texture tex{img, B, H, W};
    VolumeCfg vol(0, H, W, 0.0f, 0.0f, 0.0f, 1.0f, 1.0f, 1.0f, false);
    ProjectionCfg proj(det, spacing);
    proj.n_angles = n_angles;
    blockDim = dim3(1, 1, 1);
    gridDim = dim3((unsigned)det, (unsigned)n_angles, (unsigned)B);
    threadIdx.x = threadIdx.y = threadIdx.z = 0;
    for (unsigned z = 0; z < gridDim.z; ++z)
        for (unsigned y = 0; y < gridDim.y; ++y)
            for (unsigned x = 0; x < gridDim.x; ++x) {
                blockIdx.x = x; blockIdx.y = y; blockIdx.z = z;
                radon_forward_kernel<true, 1, float>(sino, &tex, angles, vol, proj);
            }
}

}  // extern "C"
