/*
 * oracle/voxel_oracle.c -- TEST INFRASTRUCTURE ONLY (not product code).
 *
 * CPU restatement of pcl::ApproximateVoxelGrid<pcl::PointXYZ>::applyFilter as pygicp.downsample(points, resolution)
 * drives it (call sites LoopDetection/src/RING_ros/main_RING.py:84-85,
 * disco_ros/main.py:177-178, main_SC.py:111-112).
 *
 * PARITY UNPINNED.  Neither fast_gicp's Python binding (un-vendored submodule, .gitmodules:1-6; upstream
 * src/python/main.cpp: `downsample` = eigen2pcl -> ApproximateVoxelGrid::setLeafSize(r, r, r) -> filter -> pcl2eigen) nor
 * PCL (system dependency of the reference's Docker image, docker/Dockerfile; PCL 1.10 on its Ubuntu 20.04 base) is in
 * /root/reference, and the reference holds no test for it.  Restated from PCL's published implementation
 * (filters/include/pcl/filters/impl/approximate_voxel_grid.hpp, identical in 1.8 - 1.12):
 *   - histsize_ = 512 direct-mapped history entries {ix, iy, iz, count, centroid};
 *   - per input point, in input order: ix = (int) floor(x * inverse_leaf_size) (float arithmetic; likewise iy, iz);
 *     hash = (ix * 7171 + iy * 3079 + iz * 4231) & (histsize_ - 1); if the entry holds another voxel its centroid
 *     (float sum / float count) is flushed to the output and the entry restarted; the point is added to the entry;
 *   - finally every non-empty entry is flushed in table order.
 * The filter is an APPROXIMATION of a voxel-grid centroid filter: a voxel evicted by a colliding one and revisited later
 * yields several output points; output order = flush order.
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

#define HISTSIZE 512

typedef struct { int ix, iy, iz, count; float c[3]; } he_t;

/* xyz: [n][3] float (pygicp casts its float64 input to pcl::PointXYZ); out: [<= n][3] float; returns the output count */
int orc_approx_voxel_grid(const float* xyz, int n, float leaf, float* out)
{
    he_t hist[HISTSIZE];
    memset(hist, 0, sizeof(hist));
    const float inv = 1.0f / leaf;
    int op = 0;
    for (int cp = 0; cp < n; ++cp) {
        const float* p = xyz + 3 * (size_t)cp;
        const int ix = (int)floorf(p[0] * inv), iy = (int)floorf(p[1] * inv), iz = (int)floorf(p[2] * inv);
        const unsigned hash = (unsigned)((ix * 7171 + iy * 3079 + iz * 4231) & (HISTSIZE - 1));
        he_t* h = &hist[hash];
        if (h->count && (ix != h->ix || iy != h->iy || iz != h->iz)) {
            const float c = (float)h->count;
            out[3 * (size_t)op] = h->c[0] / c; out[3 * (size_t)op + 1] = h->c[1] / c; out[3 * (size_t)op + 2] = h->c[2] / c;
            ++op;
            h->count = 0;
            h->c[0] = h->c[1] = h->c[2] = 0.0f;
        }
        h->ix = ix; h->iy = iy; h->iz = iz;
        h->count++;
        h->c[0] += p[0]; h->c[1] += p[1]; h->c[2] += p[2];
    }
    for (int i = 0; i < HISTSIZE; ++i) {
        he_t* h = &hist[i];
        if (!h->count) continue;
        const float c = (float)h->count;
        out[3 * (size_t)op] = h->c[0] / c; out[3 * (size_t)op + 1] = h->c[1] / c; out[3 * (size_t)op + 2] = h->c[2] / c;
        ++op;
    }
    return op;
}
