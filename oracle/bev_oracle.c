/*
 * oracle/bev_oracle.c -- TEST INFRASTRUCTURE ONLY (not product code).
 *
 * CPU restatement of the reference BEV rasterisers.  Only tests/, the
 * __graft_entry__.smoke() check and bench.py's cpu_baseline leg may load this.
 *
 *   polar  (A1/A2): LoopDetection/src/disco_ros/tools/multi-layer-polar-cpu/cython/src/
 *                   kernel.cpp:23-36 (xy2theta), kernel.cpp:40-77 (point2gridmap),
 *                   manager.cpp:41-58 (retreive)
 *   cart   (A3/A4): LoopDetection/generate_bev_cython_binary/src/kernel.cu:14-61,
 *                   manager.cu:53-91
 *   feat   (A5)   : LoopDetection/generate_bev_pointfeat_cython/src/kernel.cu:106-164
 *
 * Pinned against: oracle/_ref/libref_polar.so (the reference's own kernel.cpp +
 * manager.cpp compiled unmodified by oracle/Makefile) on the reference fixtures
 * 1.bin / 2.bin and on seeded random clouds (tests/test_oracle_bev.py), and against
 * the golden index arrays in tests/golden/.  The Cartesian / feature rasterisers are
 * CUDA-only in the reference (no nvcc here); their index math is IEEE double
 * add/div/floor and therefore reproducible bit for bit on any host.
 *
 * Where the reference has undefined behaviour this restatement DROPS the point and
 * says so (DESIGN.md "defined subset"):
 *   - NaN coordinates (xy2theta falls off its last branch)
 *   - quotients outside int range ((int)floor(huge))
 *   - a linear cell index outside [0, cells): the reference writes out of bounds.
 * A per-axis index that is out of its own range but whose LINEAR index stays inside
 * the grid (e.g. sector == num_sector when theta rounds to 360.0f) aliases to the
 * neighbouring cell exactly as the reference's pointer arithmetic does.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_DROP INT32_MIN

static int orc_floor_to_int(double q, int *ok)
{
    double f = floor(q);
    if (!(f > -1073741824.0 && f < 1073741824.0)) { *ok = 0; return 0; } /* NaN fails too */
    return (int)f;
}

/* kernel.cpp:23-36.  `atan` resolves to ::atan(double) there; the (180/M_PI) factor and
 * the subtraction are double, the return narrows to float. */
static float orc_theta_deg(float x, float y, int *ok)
{
    const double k = 180 / M_PI;
    if (x >= 0 && y >= 0) return (float)(k * atan((double)(y / x)));
    if (x < 0 && y >= 0)  return (float)(180 - k * atan((double)(y / (-x))));
    if (x < 0 && y < 0)   return (float)(180 + k * atan((double)(y / x)));
    if (x >= 0 && y < 0)  return (float)(360 - k * atan((double)((-y) / x)));
    *ok = 0; /* NaN */
    return 0.f;
}

/* A1: per point ring / sector / height exactly as kernel.cpp:40-77 computes them.
 * xyz is the reference SoA [x0..xn-1, y0.., z0..].  valid[i]=0 marks dropped points. */
void orc_bev_polar_indices(const float *xyz, int n, int max_length, int max_height,
                           int num_ring, int num_sector, int num_height,
                           int *ring, int *sector, int *height, unsigned char *valid)
{
    const float gap_ring = (float)max_length / (float)num_ring;
    const float gap_sector = (float)(360.0 / (float)num_sector);
    const float gap_height = (float)(2.0 * (float)max_height / (float)num_height);
    for (int i = 0; i < n; ++i) {
        float x = xyz[i], y = xyz[i + n], z = xyz[i + 2 * (size_t)n];
        int ok = 1;
        if (x == 0.0) x = 0.0001;
        if (y == 0.0) y = 0.0001;
        if (z == 0.0) z = 0.0001;
        float theta = orc_theta_deg(x, y, &ok);
        float far = (float)sqrt((double)x * (double)x + (double)y * (double)y);
        int r = orc_floor_to_int((double)(far / gap_ring), &ok);
        int s = orc_floor_to_int((double)(theta / gap_sector), &ok);
        int h = orc_floor_to_int((double)((z + (float)max_height) / gap_height), &ok);
        if (ok && r >= num_ring) r = num_ring - 1;
        ring[i] = ok ? r : ORC_DROP;
        sector[i] = ok ? s : ORC_DROP;
        height[i] = ok ? h : ORC_DROP;
        if (valid) valid[i] = (unsigned char)ok;
    }
}

/* A2: manager.cpp:41-58.  out has 3*cells*enough_large floats, zero-filled by caller. */
void orc_bev_polar_scatter(const float *xyz, int n, const int *ring, const int *sector,
                           const int *height, int num_ring, int num_sector, int num_height,
                           int enough_large, float *out)
{
    const int64_t cells = (int64_t)num_ring * num_sector * num_height;
    int *counter = (int *)calloc((size_t)cells, sizeof(int));
    for (int i = 0; i < n; ++i) {
        if (ring[i] == ORC_DROP) continue;
        int64_t lin = (int64_t)sector[i] + (int64_t)ring[i] * num_sector +
                      (int64_t)height[i] * num_sector * num_ring;
        if (lin < 0 || lin >= cells) continue; /* reference: out-of-bounds write */
        int k = counter[lin];
        if (k < enough_large) {
            float *o = out + 3 * (lin + (int64_t)k * cells);
            o[0] = xyz[i];
            o[1] = xyz[i + n];
            o[2] = 1.f;
            counter[lin] = k + 1;
        }
    }
    free(counter);
}

/* substitution + clamp shared by the Cartesian rasterisers (kernel.cu:31-50) */
static float orc_cart_prep(float v)
{
    if (v == 0.0) v = 0.0001;
    if (v > 1.0) v = 0.9999;
    if (v < -1.0) v = -0.9999;
    return v;
}

/* A3: kernel.cu:14-61.  `(x + 1.0) / gap` is evaluated in double, gap is a float. */
void orc_bev_cart_indices(const float *xyz, int n, int max_length, int max_height,
                          int num_x, int num_y, int num_height,
                          int *ix, int *iy, int *ih, unsigned char *valid)
{
    const float gap_x = (float)(2.0 * (float)max_length / (float)num_x);
    const float gap_y = (float)(2.0 * (float)max_length / (float)num_y);
    const float gap_h = (float)(2.0 * (float)max_height / (float)num_height);
    for (int i = 0; i < n; ++i) {
        float x = orc_cart_prep(xyz[i]);
        float y = orc_cart_prep(xyz[i + n]);
        float z = orc_cart_prep(xyz[i + 2 * (size_t)n]);
        int ok = 1;
        int a = orc_floor_to_int(((double)x + 1.0) / (double)gap_x, &ok);
        int b = orc_floor_to_int(((double)y + 1.0) / (double)gap_y, &ok);
        int c = orc_floor_to_int(((double)z + 1.0) / (double)gap_h, &ok);
        ix[i] = ok ? a : ORC_DROP;
        iy[i] = ok ? b : ORC_DROP;
        ih[i] = ok ? c : ORC_DROP;
        if (valid) valid[i] = (unsigned char)ok;
    }
}

/* A4: manager.cu:53-91.  Sequential semantics: ch0/ch1 = x,y of the last point of the
 * cell in input order; ch2 = z of the last point that raised the running (double)
 * maximum of its COLUMN (max_h is indexed without the height layer, initialised 0).
 * out has 3*num_x*num_y*num_height floats, zero-filled by caller. */
void orc_bev_cart_scatter(const float *xyz, int n, const int *ix, const int *iy, const int *ih,
                          int num_x, int num_y, int num_height, float *out)
{
    const int64_t cols = (int64_t)num_x * num_y;
    const int64_t cells = cols * num_height;
    double *max_h = (double *)calloc((size_t)cols, sizeof(double));
    for (int i = 0; i < n; ++i) {
        if (ix[i] == ORC_DROP) continue;
        int64_t col = (int64_t)iy[i] + (int64_t)ix[i] * num_y;
        int64_t lin = col + (int64_t)ih[i] * cols;
        if (col < 0 || col >= cols || lin < 0 || lin >= cells) continue; /* reference: OOB */
        float z = xyz[i + 2 * (size_t)n];
        out[3 * lin + 0] = xyz[i];
        out[3 * lin + 1] = xyz[i + n];
        if (max_h[col] < z) {
            out[3 * lin + 2] = z;
            max_h[col] = z;
        }
    }
    free(max_h);
}

/* A5: kernel.cu:106-164 with the per-cell, per-channel maximum taken as a TRUE maximum
 * (the reference's plain load/compare/store races).  pts is
 * channel-major [F*n]: planes 0..2 are x,y,z.  out has num_x*num_y*num_height*F floats,
 * zero-filled by caller; like the reference, values <= 0 never replace the initial 0.
 * For num_height > 1 the reference semantics are order dependent and racy; this
 * restatement applies the sequential reading (running max per column and channel). */
void orc_bev_feat(const float *pts, int n, int F, int max_length, int max_height,
                  int num_x, int num_y, int num_height, float *out)
{
    const float gap_x = (float)(2.0 * (float)max_length / (float)num_x);
    const float gap_y = (float)(2.0 * (float)max_length / (float)num_y);
    const float gap_h = (float)(2.0 * (float)max_height / (float)num_height);
    const int64_t cols = (int64_t)num_x * num_y;
    const int64_t cells = cols * num_height;
    float *max_h = (float *)calloc((size_t)(cols * F), sizeof(float));
    for (int i = 0; i < n; ++i) {
        float x = orc_cart_prep(pts[i]);
        float y = orc_cart_prep(pts[i + n]);
        float z = orc_cart_prep(pts[i + 2 * (size_t)n]);
        int ok = 1;
        int a = orc_floor_to_int(((double)x + 1.0) / (double)gap_x, &ok);
        int b = orc_floor_to_int(((double)y + 1.0) / (double)gap_y, &ok);
        int c = orc_floor_to_int(((double)z + 1.0) / (double)gap_h, &ok);
        if (!ok) continue;
        int64_t col = (int64_t)b + (int64_t)a * num_y;
        int64_t lin = col + (int64_t)c * cols;
        if (col < 0 || col >= cols || lin < 0 || lin >= cells) continue;
        for (int j = 0; j < F; ++j) {
            float v = pts[i + (size_t)j * n];
            if (max_h[F * col + j] < v) {
                out[F * lin + j] = v;
                max_h[F * col + j] = v;
            }
        }
    }
    free(max_h);
}

/* FNV-1a style fingerprint over the ascending linear indices of cells whose channel-2
 * value is non-zero (SURVEY.md section 8(c) session fingerprints). */
uint64_t orc_occupied_fingerprint(const float *out3, int64_t cells, int64_t *count)
{
    uint64_t h = 1469598103934665603ull;
    int64_t c = 0;
    for (int64_t i = 0; i < cells; ++i)
        if (out3[3 * i + 2] != 0.f) { h = (h ^ (uint64_t)i) * 1099511628211ull; ++c; }
    if (count) *count = c;
    return h;
}
