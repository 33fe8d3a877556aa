// oracle/ref_cuda_host/ref_texture.h -- TEST INFRASTRUCTURE ONLY.
// Host stand-in for the texture fetch of the reference's Radon kernel (torch-radon/src/forward.cu:103-113, texture set up in
// src/texture.cu:133-143: layered 2-D float texture, unnormalised coordinates, cudaAddressModeBorder, cudaFilterModeLinear).
// Linear filtering as the CUDA programming guide documents it ("Texture Fetching", linear filtering):
//   tex(x, y) = (1-a)(1-b) T[i][j] + a(1-b) T[i+1][j] + (1-a) b T[i][j+1] + a b T[i+1][j+1],
//   i = floor(x - 0.5), a = frac(x - 0.5), j = floor(y - 0.5), b = frac(y - 0.5),
//   a and b stored in 9-bit fixed point with 8 fractional bits; texels outside the image read 0 (border mode).
// ref_tex_weight_bits = 8 reproduces that; 0 keeps the fp32 fractions (what the HIP kernel and oracle/radon_oracle.c use),
// which separates "same rays, same samples" from the texture unit's weight quantisation.
#pragma once
#include <cmath>

struct ref_texture { const float* data; int layers, height, width; };
typedef const ref_texture* cudaTextureObject_t;
inline int ref_tex_weight_bits = 8;

inline float ref_tex_texel(const ref_texture* t, int layer, int ix, int iy)
{
    if (ix < 0 || iy < 0 || ix >= t->width || iy >= t->height) return 0.0f;
    return t->data[((size_t)layer * t->height + iy) * t->width + ix];
}
inline float ref_tex_frac(float f)
{
    if (ref_tex_weight_bits <= 0) return f;
    const float s = (float)(1 << ref_tex_weight_bits);
    return std::floor(f * s + 0.5f) / s;
}
template <typename T>
inline T tex2DLayered(cudaTextureObject_t t, float x, float y, int layer)
{
    const float xb = x - 0.5f, yb = y - 0.5f;
    const float fi = std::floor(xb), fj = std::floor(yb);
    const float a = ref_tex_frac(xb - fi), b = ref_tex_frac(yb - fj);
    const int i = (int)fi, j = (int)fj;
    return (T)((1.0f - a) * (1.0f - b) * ref_tex_texel(t, layer, i, j) + a * (1.0f - b) * ref_tex_texel(t, layer, i + 1, j) +
               (1.0f - a) * b * ref_tex_texel(t, layer, i, j + 1) + a * b * ref_tex_texel(t, layer, i + 1, j + 1));
}
struct float4 { float x, y, z, w; };
template <>
inline float4 tex2DLayered<float4>(cudaTextureObject_t, float, float, int) { return float4{0, 0, 0, 0}; }   // 4-channel path: not driven here

// device intrinsics of the kernel: the fast-math cosine / sine are hardware approximations (2^-21.4 absolute error per the
// CUDA guide); libm's take their place
inline float __cosf(float x) { return std::cos(x); }
inline float __sinf(float x) { return std::sin(x); }
inline int __float2int_rn(float x) { return (int)std::nearbyint(x); }
template <class A, class B> inline auto max(A a, B b) -> decltype(a + b) { return a > b ? a : b; }
template <class A, class B> inline auto min(A a, B b) -> decltype(a + b) { return a < b ? a : b; }
