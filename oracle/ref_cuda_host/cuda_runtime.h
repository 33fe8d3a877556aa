// oracle/ref_cuda_host/cuda_runtime.h -- TEST INFRASTRUCTURE ONLY.
// Host stand-in for the handful of CUDA runtime names the reference's generate_bev_* sources (and, further down, its
// elevation_mapping/cuda/gpu_process.cu) use
// (cudaMalloc / cudaMemcpy / cudaFree / cudaDeviceSynchronize, __global__, threadIdx / blockIdx /
// blockDim, dim3), so that their kernel.cu + manager.cu -- host-only logic apart from one <<<>>>
// launch -- compile with g++ from where they lie under /root/reference.  oracle/Makefile rewrites the
// launch `kernel<<<grid, block>>>(args)` into REF_LAUNCH(kernel, grid, block, args) in a scratch copy
// under oracle/_ref/build/ (git-ignored, deleted again after linking); REF_LAUNCH runs the "threads" one after another in gid order,
// i.e. the sequential reading of the kernel.  cudaFree tolerates the reference's double free
// (generate_bev_cython_binary/src/manager.cu:87-90 and :94-99 free the same four buffers).
#pragma once
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <set>

#define __global__
#define __device__
#define __host__
#define __constant__

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct ref_uint3 { unsigned x = 0, y = 0, z = 0; };
inline ref_uint3 threadIdx, blockIdx;
inline dim3 blockDim, gridDim;
inline void __syncthreads() {}

typedef int cudaError_t;
enum { cudaSuccess = 0 };
enum cudaMemcpyKind { cudaMemcpyHostToHost, cudaMemcpyHostToDevice, cudaMemcpyDeviceToHost, cudaMemcpyDeviceToDevice };

inline std::set<void*>& ref_live_allocations() { static std::set<void*> s; return s; }
inline cudaError_t cudaMalloc(void** p, size_t n) { *p = std::malloc(n ? n : 1); ref_live_allocations().insert(*p); return cudaSuccess; }
inline cudaError_t cudaFree(void* p)
{
    auto it = ref_live_allocations().find(p);
    if (it != ref_live_allocations().end()) { ref_live_allocations().erase(it); std::free(p); }
    return cudaSuccess;   // a second cudaFree of the same pointer is an error code on CUDA, not a crash
}
inline cudaError_t cudaMemcpy(void* d, const void* s, size_t n, cudaMemcpyKind) { std::memcpy(d, s, n); return cudaSuccess; }
inline cudaError_t cudaDeviceSynchronize() { return cudaSuccess; }
inline cudaError_t cudaGetLastError() { return cudaSuccess; }
inline const char* cudaGetErrorString(cudaError_t) { return "no error (host stand-in)"; }

#define REF_LAUNCH(kernel, grid, block, ...)                                   \
    do {                                                                       \
        gridDim = (grid); blockDim = (block);                                  \
        for (unsigned bx__ = 0; bx__ < gridDim.x; ++bx__)                      \
            for (unsigned tx__ = 0; tx__ < blockDim.x; ++tx__) {               \
                blockIdx.x = bx__; threadIdx.x = tx__;                         \
                kernel(__VA_ARGS__);                                           \
            }                                                                  \
    } while (0)

// ---- additions for elevation_mapping/cuda/gpu_process.cu: module-scope __device__ / __constant__ variables become plain
// globals, so the "symbol" copies are memcpy's on the variable itself; atomics are their sequential reading.
template <class T>
inline cudaError_t cudaMemcpyToSymbol(T& symbol, const void* src, size_t n, size_t offset = 0, cudaMemcpyKind = cudaMemcpyHostToDevice)
{
    std::memcpy(reinterpret_cast<char*>(&symbol) + offset, src, n);
    return cudaSuccess;
}
template <class T>
inline cudaError_t cudaMemcpyFromSymbol(void* dst, const T& symbol, size_t n, size_t offset = 0, cudaMemcpyKind = cudaMemcpyDeviceToHost)
{
    std::memcpy(dst, reinterpret_cast<const char*>(&symbol) + offset, n);
    return cudaSuccess;
}
inline cudaError_t cudaMemset(void* p, int v, size_t n) { std::memset(p, v, n); return cudaSuccess; }
inline int __float_as_int(float f) { int i; std::memcpy(&i, &f, 4); return i; }
inline float __int_as_float(int i) { float f; std::memcpy(&f, &i, 4); return f; }
inline int atomicCAS(int* a, int compare, int val) { const int old = *a; if (old == compare) *a = val; return old; }
inline int atomicAdd(int* a, int v) { const int old = *a; *a = old + v; return old; }
inline float atomicAdd(float* a, float v) { const float old = *a; *a = old + v; return old; }
inline int atomicMax(int* a, int v) { const int old = *a; if (v > old) *a = v; return old; }
inline int atomicMin(int* a, int v) { const int old = *a; if (v < old) *a = v; return old; }
inline int atomicExch(int* a, int v) { const int old = *a; *a = v; return old; }
