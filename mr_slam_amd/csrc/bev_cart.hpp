// bev_cart.hpp -- per-point cell arithmetic of the Cartesian rasterisers (generate_bev_cython_binary/src/kernel.cu:14-61)
// and the streaming 16-byte accessors, shared by bev.hip and fused.hip.  See bev.hip for the design notes.
#pragma once
#include "common.hpp"

namespace {

struct CartP {
    float gap_x, gap_y, gap_h;
    float inv_x, inv_y, inv_h;
    float eps_x, eps_y, eps_h;
    int NX, NY, H, F;
    float eps_fast;   // fused.hip: edge distance below which a point leaves the fma fast path (see make_cart)
};

__host__ __device__ inline float eps_for(int bins)
{
    float e = (float)(bins + 2) * 2e-6f;
    return e < 2e-4f ? 2e-4f : e;
}

__device__ __forceinline__ float cart_prep(float v)
{
    if (v == 0.0f) v = 0.0001f;
    if (v > 1.0f) v = 0.9999f;
    if (v < -1.0f) v = -0.9999f;
    return v;
}

// floor(((double)v + 1.0) / (double)gap) for v in [-1,1]
__device__ __forceinline__ int cart_axis(float v, float gap, float inv, float eps)
{
    const float g = (v + 1.0f) * inv;
    const float f = floorf(g);
    const float fr = g - f;
    if (fr >= eps && fr <= 1.0f - eps) return (int)f;
    return (int)floor(((double)v + 1.0) / (double)gap);
}

// kernel.cu:14-61 for one point
__device__ __forceinline__ bool cart_cell(const CartP& p, float x, float y, float z, int& ix,
                                          int& iy, int& ih)
{
    x = cart_prep(x);
    y = cart_prep(y);
    z = cart_prep(z);
    if (!(x == x) || !(y == y) || !(z == z)) return false;
    ix = cart_axis(x, p.gap_x, p.inv_x, p.eps_x);
    iy = cart_axis(y, p.gap_y, p.inv_y, p.eps_y);
    ih = cart_axis(z, p.gap_h, p.inv_h, p.eps_h);
    return true;
}

// returns lin (cell incl. height layer) or -1; col = lin without the height layer
__device__ __forceinline__ int cart_lin(const CartP& p, float x, float y, float z, int& col)
{
    int ix, iy, ih;
    if (!cart_cell(p, x, y, z, ix, iy, ih)) return -1;
    const long long cols = (long long)p.NX * p.NY;
    const long long c = (long long)iy + (long long)ix * p.NY;
    const long long lin = c + (long long)ih * cols;
    if (c < 0 || c >= cols || lin < 0 || lin >= cols * p.H) return -1;
    col = (int)c;
    return (int)lin;
}

__device__ __forceinline__ bool aligned16(const void* a) { return (((uintptr_t)a) & 15) == 0; }

// streaming 16-byte access: every point is read once and every output cell written once, so keep them out
// of the way of the L2 / MALL replacement policy (non-temporal hint; +10-15 % HBM throughput measured)
typedef float f4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 stream_load4(const float4* p)
{
    const f4v v = __builtin_nontemporal_load(reinterpret_cast<const f4v*>(p));
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void stream_store4(float4* p, float4 v)
{
    f4v w = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(w, reinterpret_cast<f4v*>(p));
}

// grid parameters exactly as the reference computes them (kernel.cu:22-24: float gaps from double quotients)
inline int make_cart(const mrs_bev_cfg* c, bool feat, CartP& p)
{
    MRS_REQUIRE(c->n0 > 0 && c->n1 > 0 && c->num_height > 0, "grid sizes must be positive");
    MRS_REQUIRE(c->max_length > 0 && c->max_height > 0, "max_length / max_height must be positive");
    MRS_REQUIRE(!feat || c->enough_large >= 3, "featsize must be >= 3 (x,y,z planes)");
    MRS_REQUIRE((int64_t)c->n0 * c->n1 * c->num_height * (feat ? c->enough_large : 3) < (1ll << 28), "grid too large");
    MRS_REQUIRE(c->num_height <= 1024, "num_height > 1024 unsupported");
    p.NX = c->n0; p.NY = c->n1; p.H = c->num_height; p.F = feat ? c->enough_large : 3;
    p.gap_x = (float)(2.0 * (float)c->max_length / (float)c->n0);           // kernel.cu:22-24
    p.gap_y = (float)(2.0 * (float)c->max_length / (float)c->n1);
    p.gap_h = (float)(2.0 * (float)c->max_height / (float)c->num_height);
    p.inv_x = 1.0f / p.gap_x; p.inv_y = 1.0f / p.gap_y; p.inv_h = 1.0f / p.gap_h;
    p.eps_x = eps_for(p.NX); p.eps_y = eps_for(p.NY); p.eps_h = eps_for(p.H);
    // fast-path quotient g = fma(v, inv, inv) with inv = fl32(1 / gap): two roundings, |g - (v + 1) / gap| <= bins * 2^-23 * (1 + 2^-24);
    // the fraction / edge-distance arithmetic adds at most 2^-26.  (bins + 2) * 4e-7 is 3.4 times that bound.
    p.eps_fast = (float)((p.NX > p.NY ? p.NX : p.NY) + 2) * 4e-7f;
    return MRS_OK;
}

}  // namespace
