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
 *   - a ray is summed along increasing dominant-axis index, one fp32 running sum per tap
 *     (t0*(1-fr) and t1*fr), added at the end                              [ref: one running sum]
 *   - hypot(a,b) -> sqrtf(a*a + b*b)                                        [ref: CUDA hypot]
 */
#include <math.h>
#include <stdlib.h>

/* zero outside [0,W) x [0,H) : cudaAddressModeBorder */
static inline float orc_texel(const float *img, int W, int H, int i, int j)
{
    return (i < 0 || j < 0 || i >= W || j >= H) ? 0.0f : img[(size_t)j * W + i];
}

/* Geometry of one ray (forward.cu:32-112), independent of the image.  The samples of a ray are
 * summed along INCREASING dominant-axis texel index (a ray that runs the other way is entered at
 * its last sample: same sample set, the order of an fp32 sum is not part of the reference's
 * contract -- the CUDA kernel itself tree-reduces nothing but accumulates in a hardware-dependent
 * FMA order).  q is the minor-axis coordinate shifted by +1.5 (= -0.5 texel-centre offset + 2 border
 * texels), which keeps it positive so that floor == truncation. */
typedef struct {
    int n_steps; /* 0: the ray misses the image */
    int ydom;    /* dominant axis is y */
    int major;   /* dominant-axis texel index of the first sample */
    float q;     /* shifted minor-axis coordinate of the first sample */
    float vm;    /* its increment per sample */
    float n;     /* length of one step */
} orc_ray;

static orc_ray orc_ray_setup(int H, int W, float cs, float sn, int r, int det, float spacing, float L)
{
    orc_ray o = {0, 0, 0, 0.0f, 0.0f, 0.0f};
    const float ox = -0.5f * (float)W, oy = -0.5f * (float)H;             /* forward.cu:56-57 */
    const float sx = ((float)r - (float)det * 0.5f + 0.5f) * spacing;     /* forward.cu:32 */
    const float sy = L, ex = sx, ey = -L;
    float rsx = sx * cs + sy * sn;                                        /* forward.cu:50-53 */
    float rsy = -sx * sn + sy * cs;
    float rdx = ex * cs + ey * sn - rsx;
    float rdy = -ex * sn + ey * cs - rsy;
    rsx = rsx - ox;                                                       /* forward.cu:58-61, scale 1 */
    rsy = rsy - oy;
    const float dx = rdx >= 0 ? fmaxf(rdx, 1e-6f) : fminf(rdx, -1e-6f);
    const float dy = rdy >= 0 ? fmaxf(rdy, 1e-6f) : fminf(rdy, -1e-6f);
    const float axm = (-rsx) / dx, axp = ((float)W - rsx) / dx;
    const float aym = (-rsy) / dy, ayp = ((float)H - rsy) / dy;
    const float as = fmaxf(fminf(axp, axm), fminf(ayp, aym));
    const float ae = fminf(fmaxf(axp, axm), fmaxf(ayp, aym));
    if ((double)as > (double)ae - 1e-6) return o;                          /* forward.cu:75 */
    rsx += rdx * as;
    rsy += rdy * as;
    rdx *= (ae - as);
    rdy *= (ae - as);
    const float m = fmaxf(fabsf(rdx), fabsf(rdy));
    const int n_steps = (int)rintf(m);                                    /* __float2int_rn */
    const float vx = rdx / m, vy = rdy / m;
    o.n = sqrtf(vx * vx + vy * vy);
    float step;
    if (fabsf(rdy) >= fabsf(rdx)) {                                       /* forward.cu:100-112 */
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
    o.ydom = fabsf(rdy) >= fabsf(rdx);
    o.major = (int)floorf(o.ydom ? rsy : rsx);
    o.q = (o.ydom ? rsx : rsy) + 1.5f;
    o.vm = o.ydom ? vx : vy;
    o.n_steps = n_steps;
    if (n_steps > 0 && (o.ydom ? vy : vx) < 0) { /* enter at the last sample, walk back */
        o.major -= n_steps - 1;
        o.q = fmaf((float)(n_steps - 1), o.vm, o.q);
        o.vm = -o.vm;
    }
    return o;
}

/* One sinogram: img [H][W] -> sino [n_angles][det].  Volume centre 0, voxel size 1
 * (torch_radon/volumes.py:13-21 defaults), as every MR_SLAM call site uses it. */
void orc_radon_parallel(const float *img, int H, int W, const float *angles, int n_angles,
                        int det, float spacing, float *sino)
{
    const float L = sqrtf((W * 0.5f) * (W * 0.5f) + (H * 0.5f) * (H * 0.5f)); /* forward.cu:33 */
    for (int a = 0; a < n_angles; ++a) {
        const float cs = (float)cos((double)angles[a]);
        const float sn = (float)sin((double)angles[a]);
        for (int r = 0; r < det; ++r) {
            const orc_ray g = orc_ray_setup(H, W, cs, sn, r, det, spacing, L);
            /* dominant axis: integer texel line stepping by +1; minor axis: cumulative float, 2-tap
             * interpolation  t0*(1-fr) + t1*fr  with one running sum per tap */
            float q = g.q, a0 = 0.0f, a1 = 0.0f;
            int major = g.major;
            for (int j = 0; j < g.n_steps; ++j) {
                const float fl = floorf(q);
                const float fr = q - fl;
                const int i = (int)fl - 2;
                const float t0 = g.ydom ? orc_texel(img, W, H, i, major) : orc_texel(img, W, H, major, i);
                const float t1 = g.ydom ? orc_texel(img, W, H, i + 1, major) : orc_texel(img, W, H, major, i + 1);
                a0 = fmaf(t0, 1.0f - fr, a0);
                a1 = fmaf(t1, fr, a1);
                q += g.vm;
                major += 1;
            }
            sino[(size_t)a * det + r] = (a0 + a1) * g.n;
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
