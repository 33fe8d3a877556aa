/* fastpath_oracle.c -- TEST INFRASTRUCTURE ONLY (never linked into the product).
 *
 * Exhaustive host check of the fp32 fast path of the Cartesian rasterisers (mr_slam_amd/csrc/bev.hip k_cart_lds,
 * fused.hip rasterise_scan) against the reference formula of generate_bev_cython_binary/src/kernel.cu:52-54,
 *     idx = (int)floor(((double)v + 1.0) / (double)gap),   gap = (float)(2.0 * max_length / num)  (kernel.cu:22-24).
 * The device fast path is built from correctly rounded IEEE operations only (v_fma_f32, v_floor_f32, subtractions, compares),
 * so the same expression evaluated here with fmaf / floorf (-ffp-contract=off, no fast-math) gives the device's bits:
 *     g = fmaf(v, inv, inv), inv = 1.0f / gap;  f = floorf(g);  e = 0.5f - fabsf((g - f) - 0.5f);
 *     the point stays on the fast path iff 0 < |v| <= 1 and e >= eps, and then uses (int)f.
 * mrs_fastpath_check walks EVERY float bit pattern of the requested range (both signs), and reports how many values the fast path
 * accepts, how many of those disagree with the reference (must be 0), and the largest |g - q| seen in units of bins
 * (the bound eps has to stay above: bins * 2^-23).
 * mrs_axispath_check does the same for cart_axis() (bev_cart.hpp; index kernels, reference-layout kernels and the exact path's
 * first try): g = (v + 1.0f) * inv (three roundings), accepted iff eps <= g - floorf(g) <= 1 - eps.
 * mrs_polar_height_check: the height layer of the polar rasterisers (bev.hip k_polar_lds / polar_cell) against
 * multi-layer-polar-cpu/cython/src/kernel.cpp:48,66: idx = floor((z + (float)max_height) / gap_height) in float,
 * gap_height = (float)(2.0 * max_height / num_height), for EVERY finite float z.  Fast path: sh = z + mh, g = sh * inv;
 * when fract(g) is within eps of a bin edge g is replaced by the IEEE quotient sh / gap (what the reference evaluates);
 * accepted iff 0 <= g < 2 * H + 16, then (int)g.  (Ring and sector use v_sqrt / v_rcp / a polynomial atan, which are not
 * correctly rounded: those paths are checked on the device against the reference build, not here.)
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

static inline float bits_to_float(uint32_t b)
{
    float f;
    memcpy(&f, &b, sizeof f);
    return f;
}

/* bits_lo..bits_hi: magnitude bit patterns (0x00000001 .. 0x3f800000 covers every non-zero |v| <= 1). */
int mrs_fastpath_check(int bins, int max_length, float eps, uint32_t bits_lo, uint32_t bits_hi, uint64_t* accepted,
                       uint64_t* mismatches, double* max_err_bins, float* first_bad)
{
    const float gap = (float)(2.0 * (float)max_length / (float)bins);
    const float inv = 1.0f / gap;
    uint64_t acc = 0, bad = 0;
    double worst = 0.0;
    float bad_v = 0.0f;
#pragma omp parallel for schedule(static) reduction(+ : acc, bad) reduction(max : worst)
    for (int64_t b = (int64_t)bits_lo; b <= (int64_t)bits_hi; ++b) {
        for (int sign = 0; sign < 2; ++sign) {
            const float v = bits_to_float((uint32_t)b | (sign ? 0x80000000u : 0u));
            if (!(fabsf(v) <= 1.0f) || v == 0.0f) continue;
            const float g = fmaf(v, inv, inv);
            const float f = floorf(g);
            const float e = 0.5f - fabsf((g - f) - 0.5f);
            const double q = ((double)v + 1.0) / (double)gap;
            const double err = fabs((double)g - q);
            if (err > worst) worst = err;
            if (!(e >= eps)) continue;
            ++acc;
            if ((int)f != (int)floor(q)) {
                ++bad;
#pragma omp critical
                bad_v = v;
            }
        }
    }
    *accepted = acc;
    *mismatches = bad;
    *max_err_bins = worst;
    *first_bad = bad_v;
    return 0;
}

int mrs_axispath_check(int bins, int max_length, float eps, uint32_t bits_lo, uint32_t bits_hi, uint64_t* accepted,
                       uint64_t* mismatches, double* max_err_bins, float* first_bad)
{
    const float gap = (float)(2.0 * (float)max_length / (float)bins);
    const float inv = 1.0f / gap;
    uint64_t acc = 0, bad = 0;
    double worst = 0.0;
    float bad_v = 0.0f;
#pragma omp parallel for schedule(static) reduction(+ : acc, bad) reduction(max : worst)
    for (int64_t b = (int64_t)bits_lo; b <= (int64_t)bits_hi; ++b) {
        for (int sign = 0; sign < 2; ++sign) {
            const float v = bits_to_float((uint32_t)b | (sign ? 0x80000000u : 0u));
            if (!(fabsf(v) <= 1.0f) || v == 0.0f) continue;   /* what cart_prep() leaves: non-zero, within [-1, 1] */
            const float g = (v + 1.0f) * inv;
            const float f = floorf(g);
            const float fr = g - f;
            const double q = ((double)v + 1.0) / (double)gap;
            const double err = fabs((double)g - q);
            if (err > worst) worst = err;
            if (!(fr >= eps && fr <= 1.0f - eps)) continue;
            ++acc;
            if ((int)f != (int)floor(q)) {
                ++bad;
#pragma omp critical
                bad_v = v;
            }
        }
    }
    *accepted = acc;
    *mismatches = bad;
    *max_err_bins = worst;
    *first_bad = bad_v;
    return 0;
}

int mrs_polar_height_check(int num_height, int max_height, float eps, uint64_t* accepted, uint64_t* mismatches, float* first_bad)
{
    const float gap = (float)(2.0 * (float)max_height / (float)num_height);
    const float inv = 1.0f / gap;
    const float mh = (float)max_height;
    const float hi = 1.0f - eps, hmax = (float)(2 * num_height + 16);
    uint64_t acc = 0, bad = 0;
    float bad_v = 0.0f;
#pragma omp parallel for schedule(static) reduction(+ : acc, bad)
    for (int64_t b = 1; b < 0x7f800000ll; ++b) {          /* every finite non-zero magnitude (z == 0 is substituted before) */
        for (int sign = 0; sign < 2; ++sign) {
            const float z = bits_to_float((uint32_t)b | (sign ? 0x80000000u : 0u));
            const float sh = z + mh;
            float g = sh * inv;
            float fr = g - floorf(g);                      /* v_fract_f32 */
            if (fr >= 1.0f) fr = 0x1.fffffep-1f;
            if (!(fr >= eps) || !(fr <= hi)) g = sh / gap;
            if (!(g >= 0.0f) || !(g < hmax)) continue;     /* leaves the fast path (exact evaluation elsewhere) */
            ++acc;
            const float ref = floorf(sh / gap);
            if ((float)(int)g != ref) {
                ++bad;
#pragma omp critical
                bad_v = z;
            }
        }
    }
    *accepted = acc;
    *mismatches = bad;
    *first_bad = bad_v;
    return 0;
}
