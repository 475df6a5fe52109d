// K1: hard voxelisation (+ per-voxel mean), bit-identical to the reference's serial first-touch loop.
// Restates det3d/ops/point_cloud/point_cloud_ops.py:7-55 (_points_to_voxel_reverse_kernel), :112-184
// (points_to_voxel) and det3d/models/readers/voxel_encoder.py:18-28 (VoxelFeatureExtractorV3.forward).
//
// Serial semantics to reproduce on a parallel machine:
//   * a point is dropped if any floor((p_j - lo_j) / vs_j) falls outside the grid (fp32 arithmetic);
//   * voxels are numbered in order of their FIRST point (input order); once max_voxels exist, points that
//     would open a new voxel are dropped, points of existing voxels are still accepted;
//   * each voxel keeps its first max_points points, in input order.
// Parallel formulation (all integer, no float atomics, order independent -> bitwise reproducible):
//   1. key[i] = z*gy*gx + y*gx + x ; cell_map[key] = min(point index)              (atomicMin)
//   2. creator(i) = cell_map[key[i]] == i ; voxel id = exclusive prefix count of creators   (scan)
//   3. rounds r = 1..max_points-1: every still unplaced point atomicMin's its index into slot[vid][r];
//      the winner takes slot r.  Slot 0 is the creator.  After max_points rounds the rest is dropped,
//      exactly the points the serial loop would have skipped.
//   4. per voxel: count = number of filled slots, zero-fill the others, mean = (sum over slots) / count.
// The dense cell map (40x1440x1440 int32 = 332 MB for the nuScenes grid) is allocated once by the caller and
// restored to its all-empty state before the call returns; the reference re-allocates it on every call.
#include "common.hpp"
#include <math.h>
#include <limits.h>

namespace shasta {

constexpr int kEmpty = 0x7fffffff;
constexpr int kSlotEmpty = 0x7f7f7f7f;  // what hipMemsetAsync(0x7f) leaves

struct VoxGrid {
    float lo[3], vs[3];
    int g[3];  // x, y, z cells
};

static VoxGrid make_grid(const float* r, const float* v) {
    VoxGrid G;
    for (int j = 0; j < 3; ++j) {
        G.lo[j] = r[j];
        G.vs[j] = v[j];
        const float span = r[3 + j] - r[j];  // fp32, as numpy float32
        G.g[j] = (int)lrintf(span / v[j]);   // np.round: half to even
    }
    return G;
}

__global__ void vox_key_kernel(const float* __restrict__ pts, int P, int ndim, VoxGrid G, int* __restrict__ keys,
                               int* __restrict__ cell_map) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    int c[3];
    bool ok = true;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const float f = floorf(__fdiv_rn(__fsub_rn(pts[(size_t)i * ndim + j], G.lo[j]), G.vs[j]));
        if (f < 0.0f || f >= (float)G.g[j]) ok = false;
        c[j] = ok ? (int)f : 0;
    }
    int key = -1;
    if (ok) {
        key = (c[2] * G.g[1] + c[1]) * G.g[0] + c[0];
        atomicMin(&cell_map[key], i);
    }
    keys[i] = key;
}

__global__ __launch_bounds__(256) void vox_count_kernel(const int* __restrict__ keys, const int* __restrict__ cell_map,
                                                        int P, int* __restrict__ block_sums) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    const bool flag = i < P && keys[i] >= 0 && cell_map[keys[i]] == i;
    const int n = __syncthreads_count(flag);
    if (threadIdx.x == 0) block_sums[blockIdx.x] = n;
}

// single block: exclusive scan of block_sums[0..n) in place; block_sums[n] = total
__global__ __launch_bounds__(1024) void vox_scan_kernel(int* __restrict__ block_sums, int n) {
    __shared__ int part[1024];
    const int t = threadIdx.x;
    const int chunk = (n + 1023) / 1024;
    const int beg = t * chunk, end = min(n, beg + chunk);
    int s = 0;
    for (int i = beg; i < end; ++i) s += block_sums[i];
    part[t] = s;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        const int v = t >= off ? part[t - off] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    int run = part[t] - s;  // exclusive prefix of this thread's chunk
    for (int i = beg; i < end; ++i) {
        const int v = block_sums[i];
        block_sums[i] = run;
        run += v;
    }
    if (t == 1023) block_sums[n] = part[1023];
}

__global__ __launch_bounds__(256) void vox_assign_kernel(const int* __restrict__ keys, const int* __restrict__ cell_map,
                                                         int P, const int* __restrict__ block_sums, VoxGrid G,
                                                         int max_voxels, int* __restrict__ vid_of_point,
                                                         int* __restrict__ coors, int* __restrict__ num_voxels,
                                                         int nblocks) {
    __shared__ int wave_cnt[4];
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int key = i < P ? keys[i] : -1;
    const bool flag = key >= 0 && cell_map[key] == i;
    const unsigned long long m = __ballot(flag);
    if (lane == 0) wave_cnt[wid] = __popcll(m);
    __syncthreads();
    int off = block_sums[blockIdx.x];
    for (int w = 0; w < wid; ++w) off += wave_cnt[w];
    const int vid = off + __popcll(m & ((1ull << lane) - 1ull));
    if (flag) {
        vid_of_point[i] = vid;
        if (vid < max_voxels) {
            const int x = key % G.g[0], y = (key / G.g[0]) % G.g[1], z = key / (G.g[0] * G.g[1]);
            coors[vid * 3 + 0] = z;
            coors[vid * 3 + 1] = y;
            coors[vid * 3 + 2] = x;
        }
    }
    if (i == 0) *num_voxels = min(block_sums[nblocks], max_voxels);
}

// pvid[i] = voxel of point i (or -1 when the point is dropped)
__global__ void vox_pvid_kernel(const int* __restrict__ keys, const int* __restrict__ cell_map,
                                const int* __restrict__ vid_of_point, int P, int max_voxels, int* __restrict__ pvid) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    const int key = keys[i];
    int v = -1;
    if (key >= 0) {
        v = vid_of_point[cell_map[key]];
        if (v >= max_voxels) v = -1;
    }
    pvid[i] = v;
}

// round r: a point that won slot r (slot_idx == i; the creator for r == 0) copies itself into voxels[vid][r] and
// retires; every other live point bids for slot r+1.  Round 0 also restores the cell map.
__global__ void vox_round_kernel(const float* __restrict__ pts, int P, int ndim, int r, int max_points,
                                 int* __restrict__ keys, int* __restrict__ pvid, int* __restrict__ cell_map,
                                 int* __restrict__ slot_idx, float* __restrict__ voxels) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    if (r == 0) {
        const int key = keys[i];
        if (key >= 0) cell_map[key] = kEmpty;  // same value from every point of the cell
    }
    const int v = pvid[i];
    if (v < 0) return;
    int* slots = slot_idx + (size_t)v * max_points;
    const bool won = (r == 0) ? (slots[0] == i) : (slots[r] == i);
    if (won) {
        float* dst = voxels + ((size_t)v * max_points + r) * ndim;
        for (int c = 0; c < ndim; ++c) dst[c] = pts[(size_t)i * ndim + c];
        pvid[i] = -1;
    } else if (r + 1 < max_points) {
        atomicMin(&slots[r + 1], i);
    }
}

// slot 0 of every voxel belongs to its creator
__global__ void vox_seed_kernel(const int* __restrict__ keys, const int* __restrict__ pvid, int P, int max_points,
                                int* __restrict__ slot_idx) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    const int v = pvid[i];
    if (v >= 0) atomicMin(&slot_idx[(size_t)v * max_points], i);
}

__global__ void vox_finalize_kernel(const int* __restrict__ slot_idx, const int* __restrict__ num_voxels,
                                    int max_points, int ndim, float* __restrict__ voxels, int* __restrict__ num_points,
                                    float* __restrict__ mean) {
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= *num_voxels) return;
    int cnt = 0;
    for (int r = 0; r < max_points; ++r) {
        if (slot_idx[(size_t)v * max_points + r] != kSlotEmpty) {
            ++cnt;  // filled slots are contiguous from 0: a point only bids for r+1 after losing r
        } else {
            float* dst = voxels + ((size_t)v * max_points + r) * ndim;
            for (int c = 0; c < ndim; ++c) dst[c] = 0.0f;
        }
    }
    num_points[v] = cnt;
    if (mean) {
        for (int c = 0; c < ndim; ++c) {
            float s = 0.0f;
            for (int r = 0; r < cnt; ++r) s += voxels[((size_t)v * max_points + r) * ndim + c];
            mean[(size_t)v * ndim + c] = s / (float)cnt;
        }
    }
}

// VoxelFeatureExtractorV3.forward on already voxelised input: one thread per (voxel, channel), slots summed in order
__global__ void voxel_mean_kernel(const float* __restrict__ voxels, const float* __restrict__ counts, int V, int max_points,
                                  int ndim, int c_used, float* __restrict__ out) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)V * c_used) return;
    const int v = (int)(i / c_used), c = (int)(i % c_used);
    const float* src = voxels + (size_t)v * max_points * ndim + c;
    float s = 0.0f;
    for (int r = 0; r < max_points; ++r) s += src[(size_t)r * ndim];  // padded slots are zero (voxel_encoder.py:24)
    out[i] = s / counts[v];
}

struct VoxWs {
    size_t keys, pvid, vidp, bsum, slots, total;
    VoxWs(int P, int max_voxels, int max_points) {
        size_t o = 0;
        const size_t p = align_up((size_t)(P > 0 ? P : 1) * sizeof(int), 256);
        keys = o; o += p;
        pvid = o; o += p;
        vidp = o; o += p;
        bsum = o; o += align_up((size_t)(cdiv(P > 0 ? P : 1, 256) + 1) * sizeof(int), 256);
        slots = o; o += align_up((size_t)max_voxels * max_points * sizeof(int), 256);
        total = o;
    }
};

}  // namespace shasta

using namespace shasta;

extern "C" size_t shasta_voxelize_cell_map_bytes(const float* h_range6, const float* h_voxel3) {
    if (!h_range6 || !h_voxel3) return 0;
    const VoxGrid G = make_grid(h_range6, h_voxel3);
    if (G.g[0] <= 0 || G.g[1] <= 0 || G.g[2] <= 0) return 0;
    return (size_t)G.g[0] * G.g[1] * G.g[2] * sizeof(int32_t);
}

extern "C" int shasta_voxelize_cell_map_init(int32_t* cell_map, size_t bytes, shasta_stream_t stream) {
    SHASTA_REQUIRE(cell_map, "cell_map_init: null pointer");
    // 0x7fffffff is not a byte pattern: fill with a kernel-free 32-bit memset
    hipError_t e = hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(cell_map), kEmpty, bytes / 4, as_stream(stream));
    if (e != hipSuccess) {
        set_error("cell_map_init", e);
        return SHASTA_E_LAUNCH;
    }
    return SHASTA_OK;
}

extern "C" size_t shasta_voxelize_workspace_bytes(int num_points, int max_voxels, int max_points) {
    if (num_points < 0 || max_voxels < 0 || max_points < 1) return 0;
    return VoxWs(num_points, max_voxels, max_points).total;
}

extern "C" int shasta_voxelize_mean_f32(const float* points, int P, int ndim, const float* h_range6,
                                        const float* h_voxel3, int max_points, int max_voxels, float* voxels,
                                        int32_t* coors, int32_t* num_points_per_voxel, float* mean,
                                        int32_t* num_voxels, int32_t* cell_map, void* workspace, size_t workspace_bytes,
                                        shasta_stream_t stream) {
    SHASTA_REQUIRE(h_range6 && h_voxel3 && num_voxels && cell_map && workspace, "voxelize: null pointer");
    SHASTA_REQUIRE(P >= 0 && ndim >= 3 && max_points >= 1 && max_voxels >= 0, "voxelize: bad size");
    SHASTA_REQUIRE(P == 0 || points, "voxelize: null points");
    SHASTA_REQUIRE(max_voxels == 0 || (voxels && coors && num_points_per_voxel), "voxelize: null outputs");
    const VoxGrid G = make_grid(h_range6, h_voxel3);
    SHASTA_REQUIRE(G.g[0] > 0 && G.g[1] > 0 && G.g[2] > 0, "voxelize: empty grid");
    SHASTA_REQUIRE((double)G.g[0] * G.g[1] * G.g[2] < 2147483647.0, "voxelize: grid too large for int32 keys");
    const VoxWs L(P, max_voxels, max_points);
    if (workspace_bytes < L.total) {
        set_error_msg("voxelize: workspace too small");
        return SHASTA_E_WORKSPACE;
    }
    hipStream_t st = as_stream(stream);
    char* ws = static_cast<char*>(workspace);
    int* keys = reinterpret_cast<int*>(ws + L.keys);
    int* pvid = reinterpret_cast<int*>(ws + L.pvid);
    int* vidp = reinterpret_cast<int*>(ws + L.vidp);
    int* bsum = reinterpret_cast<int*>(ws + L.bsum);
    int* slots = reinterpret_cast<int*>(ws + L.slots);
    if (P == 0 || max_voxels == 0) {
        hipError_t e = hipMemsetAsync(num_voxels, 0, sizeof(int32_t), st);
        if (e != hipSuccess) {
            set_error("voxelize: memset", e);
            return SHASTA_E_LAUNCH;
        }
        return SHASTA_OK;
    }
    hipError_t e = hipMemsetAsync(slots, 0x7f, (size_t)max_voxels * max_points * sizeof(int), st);
    if (e != hipSuccess) {
        set_error("voxelize: memset", e);
        return SHASTA_E_LAUNCH;
    }
    const int nb = cdiv(P, 256);
    int rc;
    hipLaunchKernelGGL(vox_key_kernel, dim3(nb), dim3(256), 0, st, points, P, ndim, G, keys, cell_map);
    if ((rc = check_launch("vox_key"))) return rc;
    hipLaunchKernelGGL(vox_count_kernel, dim3(nb), dim3(256), 0, st, keys, cell_map, P, bsum);
    if ((rc = check_launch("vox_count"))) return rc;
    hipLaunchKernelGGL(vox_scan_kernel, dim3(1), dim3(1024), 0, st, bsum, nb);
    if ((rc = check_launch("vox_scan"))) return rc;
    hipLaunchKernelGGL(vox_assign_kernel, dim3(nb), dim3(256), 0, st, keys, cell_map, P, bsum, G, max_voxels, vidp, coors,
                       num_voxels, nb);
    if ((rc = check_launch("vox_assign"))) return rc;
    hipLaunchKernelGGL(vox_pvid_kernel, dim3(nb), dim3(256), 0, st, keys, cell_map, vidp, P, max_voxels, pvid);
    if ((rc = check_launch("vox_pvid"))) return rc;
    hipLaunchKernelGGL(vox_seed_kernel, dim3(nb), dim3(256), 0, st, keys, pvid, P, max_points, slots);
    if ((rc = check_launch("vox_seed"))) return rc;
    for (int r = 0; r < max_points; ++r) {
        hipLaunchKernelGGL(vox_round_kernel, dim3(nb), dim3(256), 0, st, points, P, ndim, r, max_points, keys, pvid,
                           cell_map, slots, voxels);
        if ((rc = check_launch("vox_round"))) return rc;
    }
    hipLaunchKernelGGL(vox_finalize_kernel, dim3(cdiv(max_voxels, 256)), dim3(256), 0, st, slots, num_voxels, max_points,
                       ndim, voxels, num_points_per_voxel, mean);
    return check_launch("vox_finalize");
}

extern "C" int shasta_voxel_mean_f32(const float* voxels, const float* num_points_f32, int num_voxels, int max_points, int ndim,
                                     int num_features, float* out, shasta_stream_t stream) {
    SHASTA_REQUIRE(num_voxels >= 0 && max_points > 0 && ndim > 0 && num_features > 0 && num_features <= ndim, "voxel_mean: bad size");
    if (num_voxels == 0) return SHASTA_OK;
    SHASTA_REQUIRE(voxels && num_points_f32 && out, "voxel_mean: null pointer");
    const long total = (long)num_voxels * num_features;
    hipLaunchKernelGGL(voxel_mean_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, as_stream(stream), voxels, num_points_f32,
                       num_voxels, max_points, ndim, num_features, out);
    return check_launch("voxel_mean");
}
