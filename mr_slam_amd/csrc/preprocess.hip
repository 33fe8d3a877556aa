// preprocess.hip -- scan pre-processing on the GPU: the steps between the
// ROS message and the BEV rasteriser in the LoopDetection nodes.
//
// Reference behaviour reproduced (never copied):
//   * voxel_down_sample(0.2): LoopDetection/src/RING_ros/main_RING.py:257-259 (open3d): voxel index =
//     floor((p - (min_bound - voxel/2)) / voxel) in double, output = mean of the points of each voxel
//     (double).  open3d's output ORDER is the iteration order of an unordered_map (unspecified);
//     here voxels come out sorted by (ix, iy, iz), points of a voxel summed in input order.
//   * pygicp.downsample(points, 0.2) (main_RING.py:84-85) = pcl::ApproximateVoxelGrid<PointXYZ>: a 512-entry
//     direct-mapped history streamed over the points in input order (row G1).  The stream is a state machine, but
//     its output has a closed form that parallelises exactly: the points of one hash bucket, in input order, split
//     into maximal runs of equal voxel coordinates; every run is one output centroid (float sum in input order /
//     float count); a run is flushed when the next run of its bucket starts (flush time = input index of that
//     run's first point) and the last run of every bucket at the end, in bucket order.  So: stable radix sort by
//     bucket, run heads, one thread per run, then a sort of the runs by flush time.  Bit-identical to the stream.
//   * load_pc_infer: LoopDetection/src/RING_ros/util.py:91-112: float32 cast, keep |x|,|y| < 70 and
//     0 < z < 30, divide by 70/70/30 (float32), order preserved.  The output is written straight in the
//     ragged SoA layout the BEV kernels consume, with device-side offsets: no host round trip.
#include <hipcub/hipcub.hpp>

#include <cmath>
#include <cstddef>

#include "common.hpp"

namespace {

__device__ __forceinline__ unsigned long long d2ord(double v)  // order-preserving double -> uint64
{
    const unsigned long long u = (unsigned long long)__double_as_longlong(v);
    return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
}
__device__ __forceinline__ double ord2d(unsigned long long u)
{
    return __longlong_as_double((long long)((u >> 63) ? (u & 0x7fffffffffffffffull) : ~u));
}

template <class T>
__global__ void k_min_bound(const T* __restrict__ pts, int stride, int n, unsigned long long* __restrict__ mn)
{
    double lo[3] = {INFINITY, INFINITY, INFINITY};
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
#pragma unroll
        for (int a = 0; a < 3; ++a) lo[a] = fmin(lo[a], (double)pts[(size_t)i * stride + a]);
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        for (int s = 32; s > 0; s >>= 1) lo[a] = fmin(lo[a], __shfl_xor(lo[a], s, 64));
        if ((threadIdx.x & 63) == 0) atomicMin(&mn[a], d2ord(lo[a]));
    }
}

// key = 21 bits per axis of floor((p - (min - voxel/2)) / voxel)
template <class T>
__global__ void k_voxel_keys(const T* __restrict__ pts, int stride, int n, double voxel,
                             const unsigned long long* __restrict__ mn, unsigned long long* __restrict__ keys,
                             int* __restrict__ vals, int* __restrict__ overflow)
{
    const double o0 = ord2d(mn[0]) - 0.5 * voxel, o1 = ord2d(mn[1]) - 0.5 * voxel, o2 = ord2d(mn[2]) - 0.5 * voxel;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const double x = (double)pts[(size_t)i * stride], y = (double)pts[(size_t)i * stride + 1], z = (double)pts[(size_t)i * stride + 2];
        const double fx = floor((x - o0) / voxel), fy = floor((y - o1) / voxel), fz = floor((z - o2) / voxel);
        if (!(fx >= 0 && fx < 2097152.0 && fy >= 0 && fy < 2097152.0 && fz >= 0 && fz < 2097152.0)) {
            atomicAdd(overflow, 1);  // NaN or an extent beyond 2^21 voxels
            keys[i] = ~0ull;
        } else {
            keys[i] = ((unsigned long long)fx << 42) | ((unsigned long long)fy << 21) | (unsigned long long)fz;
        }
        vals[i] = i;
    }
}

__global__ void k_segment_heads(const unsigned long long* __restrict__ keys, int n, int* __restrict__ head)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
        head[i] = (keys[i] != ~0ull && (i == 0 || keys[i] != keys[i - 1])) ? 1 : 0;
}

// one thread per voxel head: sequential mean over the voxel's (stable-sorted) points
template <class T>
__global__ void k_voxel_means(const T* __restrict__ pts, int stride, const unsigned long long* __restrict__ keys,
                              const int* __restrict__ perm, const int* __restrict__ head, const int* __restrict__ slot,
                              int n, double* __restrict__ out)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        if (!head[i]) continue;
        double s0 = 0, s1 = 0, s2 = 0;
        int c = 0;
        const unsigned long long k = keys[i];
        for (int j = i; j < n && keys[j] == k; ++j) {
            const T* p = pts + (size_t)perm[j] * stride;
            s0 += (double)p[0]; s1 += (double)p[1]; s2 += (double)p[2];
            ++c;
        }
        double* o = out + (size_t)slot[i] * 3;
        o[0] = s0 / c; o[1] = s1 / c; o[2] = s2 / c;
    }
}

// ---- pcl::ApproximateVoxelGrid (row G1) -----------------------------------------------------------------------
constexpr int kHist = 512;   // histsize_ of pcl::ApproximateVoxelGrid

template <class T>
__global__ void k_avg_keys(const T* __restrict__ pts, int stride, int n, float inv_leaf, unsigned* __restrict__ bucket,
                           int3* __restrict__ vox, int* __restrict__ vals)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const float x = (float)pts[(size_t)i * stride], y = (float)pts[(size_t)i * stride + 1], z = (float)pts[(size_t)i * stride + 2];
        const int ix = (int)floorf(x * inv_leaf), iy = (int)floorf(y * inv_leaf), iz = (int)floorf(z * inv_leaf);
        bucket[i] = (unsigned)((ix * 7171 + iy * 3079 + iz * 4231) & (kHist - 1));
        vox[i] = make_int3(ix, iy, iz);
        vals[i] = i;
    }
}

// sorted position j (by bucket, then input index): head of a run?
__global__ void k_avg_heads(const unsigned* __restrict__ sbucket, const int* __restrict__ perm, const int3* __restrict__ vox, int n,
                            int* __restrict__ head)
{
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < n; j += gridDim.x * blockDim.x) {
        bool h = j == 0 || sbucket[j] != sbucket[j - 1];
        if (!h) {
            const int3 a = vox[perm[j]], b = vox[perm[j - 1]];
            h = a.x != b.x || a.y != b.y || a.z != b.z;
        }
        head[j] = h ? 1 : 0;
    }
}

// one thread per run: float sum in input order, centroid, flush time
template <class T>
__global__ void k_avg_runs(const T* __restrict__ pts, int stride, const unsigned* __restrict__ sbucket, const int* __restrict__ perm,
                           const int* __restrict__ head, const int* __restrict__ slot, int n, float* __restrict__ centroid,
                           unsigned* __restrict__ flush_key, int* __restrict__ run_id)
{
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < n; j += gridDim.x * blockDim.x) {
        if (!head[j]) continue;
        float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f;
        int c = 0, e = j;
        do {
            const T* p = pts + (size_t)perm[e] * stride;
            s0 += (float)p[0]; s1 += (float)p[1]; s2 += (float)p[2];
            ++c; ++e;
        } while (e < n && !head[e]);
        const int r = slot[j];
        const float fc = (float)c;
        centroid[3 * (size_t)r] = s0 / fc; centroid[3 * (size_t)r + 1] = s1 / fc; centroid[3 * (size_t)r + 2] = s2 / fc;
        // evicted by the first point of the next run of the same bucket; the last run of a bucket is flushed at the end
        flush_key[r] = (e < n && sbucket[e] == sbucket[j]) ? (unsigned)perm[e] : (unsigned)n + sbucket[j];
        run_id[r] = r;
    }
}

__global__ void k_avg_emit(const float* __restrict__ centroid, const int* __restrict__ order, int m, double* __restrict__ out)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < m; i += gridDim.x * blockDim.x) {
        const int r = order[i];
        out[3 * (size_t)i] = (double)centroid[3 * (size_t)r];
        out[3 * (size_t)i + 1] = (double)centroid[3 * (size_t)r + 1];
        out[3 * (size_t)i + 2] = (double)centroid[3 * (size_t)r + 2];
    }
}

// load_pc_infer flags; pts may be float or double ([n][stride], xyz first)
template <class T>
__global__ void k_crop_flags(const T* __restrict__ pts, int stride, size_t n, int* __restrict__ flag)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float x = (float)pts[i * stride], y = (float)pts[i * stride + 1], z = (float)pts[i * stride + 2];
        flag[i] = (fabsf(x) < 70.0f && fabsf(y) < 70.0f && z < 30.0f && z > 0.0f) ? 1 : 0;
    }
}

// offsets_out[b] = pos[raw_offs[b]] (b < batch), offsets_out[batch] = total
__global__ void k_out_offsets(const int* __restrict__ pos, const int* __restrict__ flag, const int64_t* __restrict__ raw_offs,
                              int batch, int64_t* __restrict__ out_offs)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b > batch) return;
    const int64_t total_raw = raw_offs[batch];
    if (b < batch) out_offs[b] = raw_offs[b] < total_raw ? pos[raw_offs[b]] : (total_raw ? pos[total_raw - 1] + flag[total_raw - 1] : 0);
    else out_offs[b] = total_raw ? pos[total_raw - 1] + flag[total_raw - 1] : 0;
}

template <class T>
__global__ void k_crop_scatter(const T* __restrict__ pts, int stride, const int64_t* __restrict__ raw_offs,
                               const int* __restrict__ flag, const int* __restrict__ pos, const int64_t* __restrict__ out_offs,
                               float* __restrict__ out)
{
    const int b = blockIdx.y;
    const int64_t r0 = raw_offs[b], r1 = raw_offs[b + 1];
    const int64_t o0 = out_offs[b];
    const int64_t nb = out_offs[b + 1] - o0;
    float* px = out + 3 * o0;
    for (int64_t i = r0 + blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < r1; i += (int64_t)gridDim.x * blockDim.x) {
        if (!flag[i]) continue;
        const int64_t l = pos[i] - o0;
        px[l] = (float)pts[i * stride] / 70.0f;
        px[nb + l] = (float)pts[i * stride + 1] / 70.0f;
        px[2 * nb + l] = (float)pts[i * stride + 2] / 30.0f;
    }
}

inline int grid_for(size_t n) { size_t b = (n + 255) / 256; return (int)(b > 4096 ? 4096 : (b ? b : 1)); }

template <class T>
int voxel_downsample_impl(mrs_ctx* ctx, const T* d_pts, int stride, int n, double voxel, double* d_out, int32_t* h_count,
                          hipStream_t s)
{
    mrs::Scratch mn, keys_in, keys_out, vals_in, vals_out, head, slot, tmp, ovf;
    int st;
    if ((st = mn.alloc(3 * 8, s)) != MRS_OK) return st;
    if ((st = ovf.alloc(4, s)) != MRS_OK) return st;
    if ((st = keys_in.alloc((size_t)n * 8, s)) != MRS_OK) return st;
    if ((st = keys_out.alloc((size_t)n * 8, s)) != MRS_OK) return st;
    if ((st = vals_in.alloc((size_t)n * 4, s)) != MRS_OK) return st;
    if ((st = vals_out.alloc((size_t)n * 4, s)) != MRS_OK) return st;
    if ((st = head.alloc((size_t)n * 4, s)) != MRS_OK) return st;
    if ((st = slot.alloc((size_t)n * 4, s)) != MRS_OK) return st;
    MRS_HIP_TRY(hipMemsetAsync(mn.p, 0xff, 3 * 8, s));
    MRS_HIP_TRY(hipMemsetAsync(ovf.p, 0, 4, s));
    hipLaunchKernelGGL(k_min_bound<T>, dim3(grid_for(n)), dim3(256), 0, s, d_pts, stride, n, mn.as<unsigned long long>());
    hipLaunchKernelGGL(k_voxel_keys<T>, dim3(grid_for(n)), dim3(256), 0, s, d_pts, stride, n, voxel, mn.as<unsigned long long>(),
                       keys_in.as<unsigned long long>(), vals_in.as<int>(), ovf.as<int>());
    size_t bytes = 0;
    MRS_HIP_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, keys_in.as<unsigned long long>(), keys_out.as<unsigned long long>(),
                                                   vals_in.as<int>(), vals_out.as<int>(), n, 0, 64, s));
    size_t bytes2 = 0;
    MRS_HIP_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, bytes2, head.as<int>(), slot.as<int>(), n, s));
    if ((st = tmp.alloc(std::max(bytes, bytes2), s)) != MRS_OK) return st;
    MRS_HIP_TRY(hipcub::DeviceRadixSort::SortPairs(tmp.p, bytes, keys_in.as<unsigned long long>(), keys_out.as<unsigned long long>(),
                                                   vals_in.as<int>(), vals_out.as<int>(), n, 0, 64, s));
    hipLaunchKernelGGL(k_segment_heads, dim3(grid_for(n)), dim3(256), 0, s, keys_out.as<unsigned long long>(), n, head.as<int>());
    MRS_HIP_TRY(hipcub::DeviceScan::ExclusiveSum(tmp.p, bytes2, head.as<int>(), slot.as<int>(), n, s));
    hipLaunchKernelGGL(k_voxel_means<T>, dim3(grid_for(n)), dim3(256), 0, s, d_pts, stride, keys_out.as<unsigned long long>(),
                       vals_out.as<int>(), head.as<int>(), slot.as<int>(), n, d_out);
    MRS_HIP_TRY(hipGetLastError());
    int last_head = 0, last_slot = 0, overflow = 0;
    MRS_HIP_TRY(hipMemcpyAsync(&last_head, head.as<int>() + (n - 1), 4, hipMemcpyDeviceToHost, s));
    MRS_HIP_TRY(hipMemcpyAsync(&last_slot, slot.as<int>() + (n - 1), 4, hipMemcpyDeviceToHost, s));
    MRS_HIP_TRY(hipMemcpyAsync(&overflow, ovf.p, 4, hipMemcpyDeviceToHost, s));
    MRS_HIP_TRY(hipStreamSynchronize(s));
    if (overflow) {
        mrs::set_error("voxel_downsample: %d points are NaN or more than 2^21 voxels from the minimum bound", overflow);
        return MRS_ERR_UNSUPPORTED;
    }
    *h_count = last_slot + last_head;
    return MRS_OK;
}

// ---- batched voxel_down_sample: hash grid, no sort -----------------------------------------------------------------------------------
// The single-scan form above sorts (voxel key, point) pairs: deterministic, but a radix sort and a host synchronisation per scan.  A batch
// goes through one hash table instead (region [tab_off[b], tab_off[b + 1]) of it per scan, 2 slots per point): a point finds or claims its
// voxel's slot (atomicCAS on the key, linear probing), records the smallest point index seen there (atomicMin: the voxel's "head"), counts
// itself and adds its offset from the voxel's lower corner as 64-bit FIXED-POINT numbers (atomicAdd: integer sums do not depend on the order,
// so the result is deterministic whichever slot order the probing produced; the offsets are < one voxel, so scale 2^46 / voxel keeps 2^17
// points per voxel inside 63 bits at a resolution of voxel x 1.4e-14; scans with more points get a coarser scale, see the launch code).  Heads are then numbered in point order (prefix sum): the output
// lists the voxels of a scan in order of first occurrence, mean = corner + sum / (scale x count) -- within 1e-13 m of the double sums of
// the sorted form.
struct VoxSlot {
    unsigned long long key;        // voxel key + 1 (0 = empty)
    long long sum[3];
    int first_inv;                 // INT_MAX - (index of the first point of the voxel), 0 = none yet: grows under atomicMax, so the table's one
                                   // zeroing memset initialises it (rounds 2-5: `first` under atomicMin and a strided 2-D memset of 0x7f, 0.45 ms per batch)
    int count;
};

template <class T>
__global__ __launch_bounds__(256) void k_min_bound_batch(const T* __restrict__ pts, int stride, const int64_t* __restrict__ offs, unsigned long long* __restrict__ mn)
{
    // one atomic per workgroup and axis (rounds 2-5: one per WAVE of up to 512 x 4 waves per scan -- 390 k atomics on 192 words per batch of 64
    // scans, 1.14 ms for a 133 MB read)
    __shared__ double red[4][3];
    const int b = blockIdx.y;
    const int64_t o = offs[b];
    const int n = (int)(offs[b + 1] - o);
    double lo[3] = {INFINITY, INFINITY, INFINITY};
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
#pragma unroll
        for (int a = 0; a < 3; ++a) lo[a] = fmin(lo[a], (double)pts[(size_t)(o + i) * stride + a]);
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        for (int s = 32; s > 0; s >>= 1) lo[a] = fmin(lo[a], __shfl_xor(lo[a], s, 64));
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][a] = lo[a];
    }
    __syncthreads();
    if (threadIdx.x < 3 && n > 0) {
        const int a = threadIdx.x;
        atomicMin(&mn[3 * b + a], d2ord(fmin(fmin(red[0][a], red[1][a]), fmin(red[2][a], red[3][a]))));
    }
}

// Lidar points arrive in firing order, so consecutive points often share a voxel (0.2 m voxels, centimetres between neighbouring returns).
// A wave therefore merges RUNS of consecutive lanes with the same key before touching the table: the run's first lane claims / finds the
// slot (one CAS chain per run), a segmented scan over the wave adds the run's count and fixed-point sums and takes its smallest index, and the
// run's last lane issues the five atomics for all of them -- 6 device-scope atomics per run instead of per point.  Counts, integer sums and
// the minimum are order-free, so the table holds exactly what one-atomic-per-point insertion left there.
template <class T>
__global__ __launch_bounds__(256) void k_vox_insert(const T* __restrict__ pts, int stride, const int64_t* __restrict__ offs, double voxel, double scale,
                             const unsigned long long* __restrict__ mn, VoxSlot* __restrict__ tab, int* __restrict__ slot_of,
                             int* __restrict__ overflow)
{
    const int b = blockIdx.y;
    const int64_t o = offs[b];
    const int n = (int)(offs[b + 1] - o);
    const double o0 = ord2d(mn[3 * b]) - 0.5 * voxel, o1 = ord2d(mn[3 * b + 1]) - 0.5 * voxel, o2 = ord2d(mn[3 * b + 2]) - 0.5 * voxel;
    VoxSlot* t = tab + 2 * o;
    const unsigned size = 2u * (unsigned)n;
    const int lane = threadIdx.x & 63;
    const int rounds = (n + (int)(gridDim.x * blockDim.x) - 1) / (int)(gridDim.x * blockDim.x);      // every lane of a wave takes part in the shuffles
    for (int r = 0; r < rounds; ++r) {
        const int i = r * (int)(gridDim.x * blockDim.x) + blockIdx.x * blockDim.x + threadIdx.x;
        unsigned long long key = 0ull;                          // 0 = nothing to insert (past the end, or outside the grid)
        long long sx = 0, sy = 0, sz = 0;
        if (i < n) {
            const double x = (double)pts[(size_t)(o + i) * stride], y = (double)pts[(size_t)(o + i) * stride + 1], z = (double)pts[(size_t)(o + i) * stride + 2];
            const double fx = floor((x - o0) / voxel), fy = floor((y - o1) / voxel), fz = floor((z - o2) / voxel);
            if (!(fx >= 0 && fx < 2097152.0 && fy >= 0 && fy < 2097152.0 && fz >= 0 && fz < 2097152.0)) {
                atomicAdd(overflow, 1);  // NaN or an extent beyond 2^21 voxels
                slot_of[o + i] = -1;
            } else {
                key = (((unsigned long long)fx << 42) | ((unsigned long long)fy << 21) | (unsigned long long)fz) + 1ull;
                // offset from the voxel's lower corner (the same expression in k_vox_emit), as fixed point
                const double cx = o0 + fx * voxel, cy = o1 + fy * voxel, cz = o2 + fz * voxel;
                sx = llrint((x - cx) * scale); sy = llrint((y - cy) * scale); sz = llrint((z - cz) * scale);
            }
        }
        const unsigned long long prev = __shfl_up(key, 1, 64), next = __shfl_down(key, 1, 64);
        const bool head = lane == 0 || prev != key;
        const bool tail = lane == 63 || next != key;
        int sl = -1;
        if (head && key != 0ull) {
            const unsigned long long h = key * 0x9E3779B97F4A7C15ull;
            unsigned q = (unsigned)(((h >> 32) * (unsigned long long)size) >> 32);
            while (true) {
                const unsigned long long seen = atomicCAS(&t[q].key, 0ull, key);
                if (seen == 0ull || seen == key) break;
                q = q + 1 == size ? 0 : q + 1;
            }
            sl = (int)q;
        }
        // segmented inclusive scan over the wave: (flag, values) (+) (flag', values') = (flag | flag', flag' ? values' : values + values')
        int f = head ? 1 : 0, cnt = key != 0ull ? 1 : 0, first = i;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int f_up = __shfl_up(f, d, 64), c_up = __shfl_up(cnt, d, 64), i_up = __shfl_up(first, d, 64), s_up = __shfl_up(sl, d, 64);
            const long long x_up = __shfl_up(sx, d, 64), y_up = __shfl_up(sy, d, 64), z_up = __shfl_up(sz, d, 64);
            if (lane >= d && !f) {
                cnt += c_up; sx += x_up; sy += y_up; sz += z_up;
                first = min(first, i_up);
                sl = s_up;                                       // the head's slot travels down its run
                f = f_up;
            }
        }
        if (key != 0ull) {
            slot_of[o + i] = sl;
            if (tail) {
                atomicMax(&t[sl].first_inv, INT_MAX - first);
                atomicAdd(&t[sl].count, cnt);
                atomicAdd((unsigned long long*)&t[sl].sum[0], (unsigned long long)sx);
                atomicAdd((unsigned long long*)&t[sl].sum[1], (unsigned long long)sy);
                atomicAdd((unsigned long long*)&t[sl].sum[2], (unsigned long long)sz);
            }
        }
    }
}

__global__ void k_vox_heads(const int64_t* __restrict__ offs, const VoxSlot* __restrict__ tab, const int* __restrict__ slot_of, int* __restrict__ head)
{
    const int b = blockIdx.y;
    const int64_t o = offs[b];
    const int n = (int)(offs[b + 1] - o);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int sl = slot_of[o + i];
        head[o + i] = (sl >= 0 && tab[2 * o + sl].first_inv == INT_MAX - i) ? 1 : 0;
    }
}

__global__ void k_vox_emit(const int64_t* __restrict__ offs, int batch, double voxel, double scale, const unsigned long long* __restrict__ mn,
                           const VoxSlot* __restrict__ tab, const int* __restrict__ slot_of, const int* __restrict__ head, const int* __restrict__ rank,
                           int64_t total, double* __restrict__ out, int64_t* __restrict__ out_offs)
{
    const int b = blockIdx.y;
    const int64_t o = offs[b];
    const int n = (int)(offs[b + 1] - o);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        out_offs[b] = o < total ? (int64_t)rank[o] : (int64_t)(rank[total - 1] + head[total - 1]);     // an empty scan at the end
        if (b == batch - 1) out_offs[batch] = total > 0 ? (int64_t)(rank[total - 1] + head[total - 1]) : 0;
    }
    const double o0 = ord2d(mn[3 * b]) - 0.5 * voxel, o1 = ord2d(mn[3 * b + 1]) - 0.5 * voxel, o2 = ord2d(mn[3 * b + 2]) - 0.5 * voxel;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        if (!head[o + i]) continue;
        const VoxSlot v = tab[2 * o + slot_of[o + i]];
        const unsigned long long key = v.key - 1ull;
        const double fx = (double)(key >> 42), fy = (double)((key >> 21) & 0x1FFFFFull), fz = (double)(key & 0x1FFFFFull);
        const double c = (double)v.count;
        double* dst = out + 3 * (size_t)rank[o + i];
        dst[0] = (o0 + fx * voxel) + ((double)v.sum[0] / scale) / c;
        dst[1] = (o1 + fy * voxel) + ((double)v.sum[1] / scale) / c;
        dst[2] = (o2 + fz * voxel) + ((double)v.sum[2] / scale) / c;
    }
}

template <class T>
int voxel_downsample_batch_impl(mrs_ctx* ctx, const T* d_pts, int stride, const int64_t* d_offs, int64_t total, int longest, int batch, double voxel,
                                double* d_out, int64_t* d_out_offs, hipStream_t s)
{
    mrs::Scratch mn, tab, slot_of, head, rank, tmp, ovf;
    int st;
    if ((st = mn.alloc((size_t)batch * 3 * 8, s)) != MRS_OK) return st;
    if ((st = ovf.alloc(4, s)) != MRS_OK) return st;
    if ((st = tab.alloc((size_t)total * 2 * sizeof(VoxSlot), s)) != MRS_OK) return st;
    if ((st = slot_of.alloc((size_t)total * 4, s)) != MRS_OK) return st;
    if ((st = head.alloc((size_t)total * 4, s)) != MRS_OK) return st;
    if ((st = rank.alloc((size_t)total * 4, s)) != MRS_OK) return st;
    MRS_HIP_TRY(hipMemsetAsync(mn.p, 0xff, (size_t)batch * 3 * 8, s));
    MRS_HIP_TRY(hipMemsetAsync(ovf.p, 0, 4, s));
    MRS_HIP_TRY(hipMemsetAsync(tab.p, 0, (size_t)total * 2 * sizeof(VoxSlot), s));
    const dim3 g((unsigned)std::min<int64_t>((longest + 255) / 256, 512), batch);
    const dim3 gmin((unsigned)std::min<int64_t>((longest + 4095) / 4096, 64), batch);        // 16 points per lane: one atomic per workgroup and axis
    // offsets < voxel <= 2^ceil(log2 voxel): |offset x scale| < 2^bits, and a voxel holds at most `longest` points: bits = 46 up to 2^17 points per
    // scan, fewer beyond (a million-point scan: 42 bits = voxel x 2e-13), so that no sum can leave 63 bits whatever the distribution of the points
    int pbits = 1;
    while (pbits < 31 && (1ll << pbits) < (long long)longest) ++pbits;
    const int bits = pbits <= 17 ? 46 : 62 - pbits;
    const double scale = ldexp(1.0, bits - (int)ceil(log2(voxel)));
    hipLaunchKernelGGL(k_min_bound_batch<T>, gmin, dim3(256), 0, s, d_pts, stride, d_offs, mn.as<unsigned long long>());
    hipLaunchKernelGGL(k_vox_insert<T>, g, dim3(256), 0, s, d_pts, stride, d_offs, voxel, scale, mn.as<unsigned long long>(), tab.as<VoxSlot>(),
                       slot_of.as<int>(), ovf.as<int>());
    hipLaunchKernelGGL(k_vox_heads, g, dim3(256), 0, s, d_offs, tab.as<VoxSlot>(), slot_of.as<int>(), head.as<int>());
    size_t bytes = 0;
    MRS_HIP_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, bytes, head.as<int>(), rank.as<int>(), (int)total, s));
    if ((st = tmp.alloc(bytes, s)) != MRS_OK) return st;
    MRS_HIP_TRY(hipcub::DeviceScan::ExclusiveSum(tmp.p, bytes, head.as<int>(), rank.as<int>(), (int)total, s));
    hipLaunchKernelGGL(k_vox_emit, g, dim3(256), 0, s, d_offs, batch, voxel, scale, mn.as<unsigned long long>(), tab.as<VoxSlot>(), slot_of.as<int>(),
                       head.as<int>(), rank.as<int>(), total, d_out, d_out_offs);
    MRS_HIP_TRY(hipGetLastError());
    int overflow = 0;
    MRS_HIP_TRY(hipMemcpyAsync(&overflow, ovf.p, 4, hipMemcpyDeviceToHost, s));
    MRS_HIP_TRY(hipStreamSynchronize(s));
    if (overflow) {
        mrs::set_error("voxel_downsample_batch: %d points are NaN or more than 2^21 voxels from their scan's minimum bound", overflow);
        return MRS_ERR_UNSUPPORTED;
    }
    return MRS_OK;
}

template <class T>
int approx_voxel_grid_impl(mrs_ctx* ctx, const T* d_pts, int stride, int n, float leaf, double* d_out, int32_t* h_count, hipStream_t s)
{
    mrs::Scratch bucket, sbucket, vox, vals_in, perm, head, slot, cent, fkey, fkey_s, rid, order, tmp;
    int st;
    if ((st = bucket.alloc((size_t)n * 4, s)) != MRS_OK) return st;
    if ((st = sbucket.alloc((size_t)n * 4, s)) != MRS_OK) return st;
    if ((st = vox.alloc((size_t)n * sizeof(int3), s)) != MRS_OK) return st;
    if ((st = vals_in.alloc((size_t)n * 4, s)) != MRS_OK) return st;
    if ((st = perm.alloc((size_t)n * 4, s)) != MRS_OK) return st;
    if ((st = head.alloc((size_t)n * 4, s)) != MRS_OK) return st;
    if ((st = slot.alloc((size_t)n * 4, s)) != MRS_OK) return st;
    if ((st = cent.alloc((size_t)n * 12, s)) != MRS_OK) return st;
    if ((st = fkey.alloc((size_t)n * 4, s)) != MRS_OK) return st;
    if ((st = fkey_s.alloc((size_t)n * 4, s)) != MRS_OK) return st;
    if ((st = rid.alloc((size_t)n * 4, s)) != MRS_OK) return st;
    if ((st = order.alloc((size_t)n * 4, s)) != MRS_OK) return st;
    const float inv_leaf = 1.0f / leaf;   // inverse_leaf_size_ = Ones / leaf_size_ (float)
    hipLaunchKernelGGL(k_avg_keys<T>, dim3(grid_for(n)), dim3(256), 0, s, d_pts, stride, n, inv_leaf, bucket.as<unsigned>(),
                       vox.as<int3>(), vals_in.as<int>());
    size_t b1 = 0, b2 = 0, b3 = 0;
    MRS_HIP_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, b1, bucket.as<unsigned>(), sbucket.as<unsigned>(), vals_in.as<int>(),
                                                   perm.as<int>(), n, 0, 9, s));
    MRS_HIP_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, b2, head.as<int>(), slot.as<int>(), n, s));
    MRS_HIP_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, b3, fkey.as<unsigned>(), fkey_s.as<unsigned>(), rid.as<int>(), order.as<int>(),
                                                   n, 0, 32, s));
    if ((st = tmp.alloc(std::max(b1, std::max(b2, b3)), s)) != MRS_OK) return st;
    // 9-bit keys, stable: bucket-major, input order inside a bucket
    MRS_HIP_TRY(hipcub::DeviceRadixSort::SortPairs(tmp.p, b1, bucket.as<unsigned>(), sbucket.as<unsigned>(), vals_in.as<int>(),
                                                   perm.as<int>(), n, 0, 9, s));
    hipLaunchKernelGGL(k_avg_heads, dim3(grid_for(n)), dim3(256), 0, s, sbucket.as<unsigned>(), perm.as<int>(), vox.as<int3>(), n,
                       head.as<int>());
    MRS_HIP_TRY(hipcub::DeviceScan::ExclusiveSum(tmp.p, b2, head.as<int>(), slot.as<int>(), n, s));
    int last_head = 0, last_slot = 0;
    MRS_HIP_TRY(hipMemcpyAsync(&last_head, head.as<int>() + (n - 1), 4, hipMemcpyDeviceToHost, s));
    MRS_HIP_TRY(hipMemcpyAsync(&last_slot, slot.as<int>() + (n - 1), 4, hipMemcpyDeviceToHost, s));
    hipLaunchKernelGGL(k_avg_runs<T>, dim3(grid_for(n)), dim3(256), 0, s, d_pts, stride, sbucket.as<unsigned>(), perm.as<int>(),
                       head.as<int>(), slot.as<int>(), n, cent.as<float>(), fkey.as<unsigned>(), rid.as<int>());
    MRS_HIP_TRY(hipGetLastError());
    MRS_HIP_TRY(hipStreamSynchronize(s));
    const int m = last_head + last_slot;
    MRS_HIP_TRY(hipcub::DeviceRadixSort::SortPairs(tmp.p, b3, fkey.as<unsigned>(), fkey_s.as<unsigned>(), rid.as<int>(), order.as<int>(),
                                                   m, 0, 32, s));
    hipLaunchKernelGGL(k_avg_emit, dim3(grid_for(m)), dim3(256), 0, s, cent.as<float>(), order.as<int>(), m, d_out);
    MRS_HIP_TRY(hipGetLastError());
    MRS_HIP_TRY(hipStreamSynchronize(s));
    *h_count = m;
    return MRS_OK;
}

template <class T>
int crop_scale_impl(mrs_ctx* ctx, const T* d_pts, int stride, const int64_t* d_raw_offs, int64_t total_raw, int longest,
                    int batch, float* d_xyz_soa, int64_t* d_out_offs, hipStream_t s)
{
    mrs::Scratch flag, pos, tmp;
    int st;
    if ((st = flag.alloc((size_t)(total_raw ? total_raw : 1) * 4, s)) != MRS_OK) return st;
    if ((st = pos.alloc((size_t)(total_raw ? total_raw : 1) * 4, s)) != MRS_OK) return st;
    if (total_raw) {
        hipLaunchKernelGGL(k_crop_flags<T>, dim3(grid_for((size_t)total_raw)), dim3(256), 0, s, d_pts, stride, (size_t)total_raw, flag.as<int>());
        size_t bytes = 0;
        MRS_HIP_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, bytes, flag.as<int>(), pos.as<int>(), (int)total_raw, s));
        if ((st = tmp.alloc(bytes, s)) != MRS_OK) return st;
        MRS_HIP_TRY(hipcub::DeviceScan::ExclusiveSum(tmp.p, bytes, flag.as<int>(), pos.as<int>(), (int)total_raw, s));
    }
    hipLaunchKernelGGL(k_out_offsets, dim3((batch + 256) / 256), dim3(256), 0, s, pos.as<int>(), flag.as<int>(), d_raw_offs, batch, d_out_offs);
    if (total_raw)
        hipLaunchKernelGGL(k_crop_scatter<T>, dim3(grid_for((size_t)longest), batch), dim3(256), 0, s, d_pts, stride, d_raw_offs,
                           flag.as<int>(), pos.as<int>(), d_out_offs, d_xyz_soa);
    MRS_HIP_TRY(hipGetLastError());
    return MRS_OK;
}

}  // namespace

extern "C" {

int mrs_voxel_downsample(mrs_ctx* ctx, const void* d_points, int32_t is_double, int32_t stride, int32_t n,
                         double voxel_size, double* d_out, int32_t* h_count, mrs_stream stream)
{
    MRS_REQUIRE(ctx && d_points && d_out && h_count, "null pointer");
    MRS_REQUIRE(n > 0 && stride >= 3, "n must be positive and stride >= 3");
    MRS_REQUIRE(voxel_size > 0.0, "voxel_size must be positive");
    MRS_HIP_TRY(hipSetDevice(ctx->device));
    return is_double ? voxel_downsample_impl<double>(ctx, (const double*)d_points, stride, n, voxel_size, d_out, h_count, (hipStream_t)stream)
                     : voxel_downsample_impl<float>(ctx, (const float*)d_points, stride, n, voxel_size, d_out, h_count, (hipStream_t)stream);
}

int mrs_voxel_downsample_batch(mrs_ctx* ctx, const void* d_points, int32_t is_double, int32_t stride, const int64_t* d_raw_offsets,
                               const int64_t* h_raw_offsets, int32_t batch, double voxel_size, double* d_out, int64_t* d_out_offsets, mrs_stream stream)
{
    MRS_REQUIRE(ctx && d_points && d_raw_offsets && h_raw_offsets && d_out && d_out_offsets, "null pointer");
    MRS_REQUIRE(batch > 0 && batch <= mrs::kMaxGridY && stride >= 3, "batch must be within [1, 65535] and stride >= 3");
    MRS_REQUIRE(voxel_size > 0.0, "voxel_size must be positive");
    MRS_REQUIRE(h_raw_offsets[0] == 0 && h_raw_offsets[batch] > 0 && h_raw_offsets[batch] < (1ll << 30), "between 1 and 2^30 points per call");
    int64_t longest = 0;
    for (int b = 0; b < batch; ++b) {
        MRS_REQUIRE(h_raw_offsets[b + 1] >= h_raw_offsets[b], "offsets must be non-decreasing");
        longest = std::max(longest, h_raw_offsets[b + 1] - h_raw_offsets[b]);
    }
    MRS_HIP_TRY(hipSetDevice(ctx->device));
    return is_double ? voxel_downsample_batch_impl<double>(ctx, (const double*)d_points, stride, d_raw_offsets, h_raw_offsets[batch], (int)longest, batch,
                                                           voxel_size, d_out, d_out_offsets, (hipStream_t)stream)
                     : voxel_downsample_batch_impl<float>(ctx, (const float*)d_points, stride, d_raw_offsets, h_raw_offsets[batch], (int)longest, batch,
                                                          voxel_size, d_out, d_out_offsets, (hipStream_t)stream);
}

int mrs_voxel_downsample_approx(mrs_ctx* ctx, const void* d_points, int32_t is_double, int32_t stride, int32_t n,
                                double leaf_size, double* d_out, int32_t* h_count, mrs_stream stream)
{
    MRS_REQUIRE(ctx && d_points && d_out && h_count, "null pointer");
    MRS_REQUIRE(n > 0 && stride >= 3, "n must be positive and stride >= 3");
    MRS_REQUIRE(leaf_size > 0.0, "leaf_size must be positive");
    MRS_HIP_TRY(hipSetDevice(ctx->device));
    return is_double ? approx_voxel_grid_impl<double>(ctx, (const double*)d_points, stride, n, (float)leaf_size, d_out, h_count, (hipStream_t)stream)
                     : approx_voxel_grid_impl<float>(ctx, (const float*)d_points, stride, n, (float)leaf_size, d_out, h_count, (hipStream_t)stream);
}

/* host-array form (pygicp.downsample hands over a numpy array): h_out holds up to n x 3 doubles */
int mrs_voxel_downsample_approx_host(mrs_ctx* ctx, const void* h_points, int32_t is_double, int32_t stride, int32_t n, double leaf_size,
                                     double* h_out, int32_t* h_count)
{
    MRS_REQUIRE(ctx && h_points && h_out && h_count, "null pointer");
    MRS_REQUIRE(n > 0 && stride >= 3, "n must be positive and stride >= 3");
    MRS_HIP_TRY(hipSetDevice(ctx->device));
    const size_t in_bytes = (size_t)n * stride * (is_double ? 8 : 4);
    mrs::Scratch in, out;
    int st = in.alloc(in_bytes, nullptr);
    if (st != MRS_OK) return st;
    if ((st = out.alloc((size_t)n * 3 * sizeof(double), nullptr)) != MRS_OK) return st;
    MRS_HIP_TRY(hipMemcpy(in.p, h_points, in_bytes, hipMemcpyHostToDevice));
    st = mrs_voxel_downsample_approx(ctx, in.p, is_double, stride, n, leaf_size, out.as<double>(), h_count, nullptr);
    if (st != MRS_OK) return st;
    if (*h_count > 0) MRS_HIP_TRY(hipMemcpy(h_out, out.p, (size_t)*h_count * 3 * sizeof(double), hipMemcpyDeviceToHost));
    return MRS_OK;
}

int mrs_crop_scale_batch(mrs_ctx* ctx, const void* d_points, int32_t is_double, int32_t stride, const int64_t* d_raw_offsets,
                         const int64_t* h_raw_offsets, int32_t batch, float* d_xyz_soa, int64_t* d_out_offsets, mrs_stream stream)
{
    MRS_REQUIRE(ctx && d_points && d_raw_offsets && h_raw_offsets && d_xyz_soa && d_out_offsets, "null pointer");
    MRS_REQUIRE(batch > 0 && stride >= 3, "batch must be positive and stride >= 3");
    MRS_REQUIRE(h_raw_offsets[batch] < (1ll << 31), "more than 2^31 points");
    int64_t longest = 0;
    for (int b = 0; b < batch; ++b) {
        MRS_REQUIRE(h_raw_offsets[b + 1] >= h_raw_offsets[b], "offsets must be non-decreasing");
        longest = std::max(longest, h_raw_offsets[b + 1] - h_raw_offsets[b]);
    }
    MRS_HIP_TRY(hipSetDevice(ctx->device));
    return is_double ? crop_scale_impl<double>(ctx, (const double*)d_points, stride, d_raw_offsets, h_raw_offsets[batch], (int)longest, batch,
                                               d_xyz_soa, d_out_offsets, (hipStream_t)stream)
                     : crop_scale_impl<float>(ctx, (const float*)d_points, stride, d_raw_offsets, h_raw_offsets[batch], (int)longest, batch,
                                              d_xyz_soa, d_out_offsets, (hipStream_t)stream);
}

}  // extern "C"
