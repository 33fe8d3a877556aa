/*
 * oracle/radon_oracle.c -- TEST INFRASTRUCTURE ONLY (not product code).
 *
 * CPU restatement of the reference parallel-beam Radon forward projector:
 *   LoopDetection/torch-radon/src/forward.cu:12-124 (radon_forward_kernel<true,1,float>)
 *   with the texture semantics of src/texture.cu:133-143 (unnormalised coordinates,
 *   linear filter, border address mode -> 0 outside).
 *
 * Pinned against the reference's own analytic checker (src/symbolic.cpp, compiled in place
 * into oracle/_ref/libref_symbolic.so) with the bound of tests/test_parallel_beam.py:70:
 * see tests/test_oracle_radon.py.  Bit parity with the CUDA path is impossible by
 * construction (hardware 8-bit bilinear weights, __sinf/__cosf); the deviations chosen here
 * are stated in DESIGN.md and shared with the HIP kernel so that HIP == oracle bit for bit:
 *   - cos/sin of the angle: (float)cos((double)a), (float)sin((double)a)   [ref: __cosf/__sinf]
 *   - interpolation weights in full fp32, lerp form with fused multiply-add [ref: 1.8 fixed point]
 *   - the reference aligns the samples of a ray to pixel centres along its dominant axis
 *     (forward.cu:100-112), so the bilinear weight along that axis is 0 up to float drift
 *     (1e-6, far below the texture unit's 1/256 weight resolution).  The alignment is made
 *     exact here: the dominant-axis texel index is an integer that steps by +-1, and each sample
 *     is a 2-tap linear interpolation along the minor axis.                [ref: 4-tap HW fetch]
 *   - hypot(a,b) -> sqrtf(a*a + b*b)                                        [ref: CUDA hypot]
 */
#include <math.h>
#include <stdlib.h>

/* zero outside [0,W) x [0,H) : cudaAddressModeBorder */
static inline float orc_texel(const float *img, int W, int H, int i, int j)
{
    return (i < 0 || j < 0 || i >= W || j >= H) ? 0.0f : img[(size_t)j * W + i];
}

/* linear interpolation along the minor axis at minor coordinate m (texel centres at +0.5),
 * on the texel line `major` of the dominant axis; ydom selects which image axis is which */
static inline float orc_tex1d(const float *img, int W, int H, int ydom, int major, float m)
{
    const float mb = m - 0.5f;
    const float fl = floorf(mb);
    const float fr = mb - fl;
    const int i = (int)fl;
    const float t0 = ydom ? orc_texel(img, W, H, i, major) : orc_texel(img, W, H, major, i);
    const float t1 = ydom ? orc_texel(img, W, H, i + 1, major) : orc_texel(img, W, H, major, i + 1);
    return fmaf(fr, t1 - t0, t0);
}

/* One sinogram: img [H][W] -> sino [n_angles][det].  Volume centre 0, voxel size 1
 * (torch_radon/volumes.py:13-21 defaults), as every MR_SLAM call site uses it. */
void orc_radon_parallel(const float *img, int H, int W, const float *angles, int n_angles,
                        int det, float spacing, float *sino)
{
    const float L = sqrtf((W * 0.5f) * (W * 0.5f) + (H * 0.5f) * (H * 0.5f)); /* forward.cu:33 */
    const float ox = -0.5f * (float)W, oy = -0.5f * (float)H;                 /* forward.cu:56-57 */
    for (int a = 0; a < n_angles; ++a) {
        const float cs = (float)cos((double)angles[a]);
        const float sn = (float)sin((double)angles[a]);
        for (int r = 0; r < det; ++r) {
            const float sx = ((float)r - (float)det * 0.5f + 0.5f) * spacing; /* forward.cu:32 */
            const float sy = L, ex = sx, ey = -L;
            float rsx = sx * cs + sy * sn;                                    /* forward.cu:50-53 */
            float rsy = -sx * sn + sy * cs;
            float rdx = ex * cs + ey * sn - rsx;
            float rdy = -ex * sn + ey * cs - rsy;
            rsx = rsx - ox;                                                   /* forward.cu:58-61, scale 1 */
            rsy = rsy - oy;
            const float dx = rdx >= 0 ? fmaxf(rdx, 1e-6f) : fminf(rdx, -1e-6f);
            const float dy = rdy >= 0 ? fmaxf(rdy, 1e-6f) : fminf(rdy, -1e-6f);
            const float axm = (-rsx) / dx, axp = ((float)W - rsx) / dx;
            const float aym = (-rsy) / dy, ayp = ((float)H - rsy) / dy;
            const float as = fmaxf(fminf(axp, axm), fminf(ayp, aym));
            const float ae = fminf(fmaxf(axp, axm), fmaxf(ayp, aym));
            float *dst = sino + (size_t)a * det + r;
            if ((double)as > (double)ae - 1e-6) { *dst = 0.0f; continue; }     /* forward.cu:75 */
            rsx += rdx * as;
            rsy += rdy * as;
            rdx *= (ae - as);
            rdy *= (ae - as);
            const float m = fmaxf(fabsf(rdx), fabsf(rdy));
            const int n_steps = (int)rintf(m);                                /* __float2int_rn */
            const float vx = rdx / m, vy = rdy / m;
            const float n = sqrtf(vx * vx + vy * vy);
            float step;
            if (fabsf(rdy) >= fabsf(rdx)) {
                const float inc = 0.5f - rsy + rintf(rsy);
                step = inc / vy;
                step += (vy < 0) ? 1.0f : 0.0f;
            } else {
                const float inc = 0.5f - rsx + rintf(rsx);
                step = inc / vx;
                step += (vx < 0) ? 1.0f : 0.0f;
            }
            rsx += step * vx;
            rsy += step * vy;
            /* dominant axis: integer texel line stepping by +-1; minor axis: cumulative float */
            const int ydom = fabsf(rdy) >= fabsf(rdx);
            int major = (int)floorf(ydom ? rsy : rsx);
            const int mstep = (ydom ? vy : vx) < 0 ? -1 : 1;
            float mc = ydom ? rsx : rsy;      /* minor-axis coordinate */
            const float vm = ydom ? vx : vy;
            float acc = 0.0f;
            for (int j = 0; j < n_steps; ++j) {
                acc += orc_tex1d(img, W, H, ydom, major, mc);
                mc += vm;
                major += mstep;
            }
            *dst = acc * n;
        }
    }
}

void orc_radon_parallel_batch(const float *img, int B, int H, int W, const float *angles,
                              int n_angles, int det, float spacing, float *sino)
{
#pragma omp parallel for schedule(dynamic)
    for (int b = 0; b < B; ++b)
        orc_radon_parallel(img + (size_t)b * H * W, H, W, angles, n_angles, det, spacing,
                           sino + (size_t)b * n_angles * det);
}
