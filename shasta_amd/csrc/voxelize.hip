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
//   2. owner(i) = cell_map[key[i]] (one random read per point); creator(i) = owner(i) == i ; voxel id = exclusive prefix count of
//      creators (scan); the creator takes slot 0
//   3. rounds r = 1..max_points-1: every still unplaced point atomicMin's its index into slot[vid][r];
//      the winner takes slot r.  After max_points rounds the rest is dropped, exactly the points the serial loop would have skipped.
//   4. per (voxel, slot): zero-fill the empty slots, the last filled one names the count; mean = (sum over slots) / count.
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

// A batch of clouds in one chain of launches (the reference voxelises the current and the previous cloud of every sample:
// datasets/pipelines/preprocess.py:179-208).  The clouds lie back to back in `pts`; a workgroup of 256 points never straddles two
// clouds (block b of the launch belongs to cloud c with boff[c] <= b < boff[c + 1] and covers points off[c] + 256 (b - boff[c]) ...),
// every cloud has its own dense cell map (cell_map + c * cells) and its own rows of the outputs.
constexpr int kMaxClouds = 32;
struct VoxBatch {
    int n;
    int off[kMaxClouds + 1];   // first point of every cloud (+ the total)
    int boff[kMaxClouds + 1];  // first block of every cloud (+ the total)
};
// (cloud, point index inside the cloud or -1) of thread `t` of block `b`
__device__ __forceinline__ void vox_locate(const VoxBatch& B, int b, int t, int& c, int& i) {
    c = 0;
    while (c + 1 < B.n && b >= B.boff[c + 1]) ++c;  // uniform
    i = (b - B.boff[c]) * 256 + t;
    if (i >= B.off[c + 1] - B.off[c]) i = -1;
}

__global__ __launch_bounds__(256) void vox_key_kernel(const float* __restrict__ pts, VoxBatch B, int ndim, VoxGrid G, long cells,
                                                      int* __restrict__ keys, int* __restrict__ cell_map) {
    int c, i;
    vox_locate(B, blockIdx.x, threadIdx.x, c, i);
    if (i < 0) return;
    const size_t gi = (size_t)B.off[c] + i;
    int cc[3];
    bool ok = true;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const float f = floorf(__fdiv_rn(__fsub_rn(pts[gi * ndim + j], G.lo[j]), G.vs[j]));
        if (f < 0.0f || f >= (float)G.g[j]) ok = false;
        cc[j] = ok ? (int)f : 0;
    }
    int key = -1;
    if (ok) {
        key = (cc[2] * G.g[1] + cc[1]) * G.g[0] + cc[0];
        atomicMin(&cell_map[(size_t)c * cells + key], i);
    }
    keys[gi] = key;
}

// owner[i] = the first point of point i's cell (the point that creates the voxel) or -1 for a dropped point - the ONE random read of
// the cell map per point; the block's number of creators for the scan
__global__ __launch_bounds__(256) void vox_owner_kernel(const int* __restrict__ keys, const int* __restrict__ cell_map, VoxBatch B, long cells,
                                                        int* __restrict__ owner, int* __restrict__ block_sums) {
    int c, i;
    vox_locate(B, blockIdx.x, threadIdx.x, c, i);
    bool flag = false;
    if (i >= 0) {
        const size_t gi = (size_t)B.off[c] + i;
        const int key = keys[gi];
        const int o = key >= 0 ? cell_map[(size_t)c * cells + key] : -1;
        owner[gi] = o;
        flag = o == i;
    }
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

// voxel ids count from 0 inside every cloud: the scan runs over all blocks, a cloud's ids start at the prefix of its first block.
// The creator also takes slot 0 of its voxel.
__global__ __launch_bounds__(256) void vox_assign_kernel(const int* __restrict__ keys, const int* __restrict__ owner, VoxBatch B,
                                                         const int* __restrict__ block_sums, VoxGrid G, int max_voxels, int max_points,
                                                         int* __restrict__ vid_of_point, int* __restrict__ coors,
                                                         int* __restrict__ num_voxels, int* __restrict__ slot_idx) {
    __shared__ int wave_cnt[4];
    int c, i;
    vox_locate(B, blockIdx.x, threadIdx.x, c, i);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const size_t gi = (size_t)B.off[c] + max(i, 0);
    const bool flag = i >= 0 && owner[gi] == i;
    const unsigned long long m = __ballot(flag);
    if (lane == 0) wave_cnt[wid] = __popcll(m);
    __syncthreads();
    int off = block_sums[blockIdx.x] - block_sums[B.boff[c]];
    for (int w = 0; w < wid; ++w) off += wave_cnt[w];
    const int vid = off + __popcll(m & ((1ull << lane) - 1ull));
    if (flag) {
        vid_of_point[gi] = vid;
        if (vid < max_voxels) {
            const int key = keys[gi];
            const int x = key % G.g[0], y = (key / G.g[0]) % G.g[1], z = key / (G.g[0] * G.g[1]);
            int* co = coors + ((size_t)c * max_voxels + vid) * 3;
            co[0] = z;
            co[1] = y;
            co[2] = x;
            slot_idx[((size_t)c * max_voxels + vid) * max_points] = i;
        }
    }
    if ((int)blockIdx.x == B.boff[c] && threadIdx.x == 0) num_voxels[c] = min(block_sums[B.boff[c + 1]] - block_sums[B.boff[c]], max_voxels);
}

// Every point learns its voxel (pvid, in place over `owner`; -1 = dropped or placed); creators copy themselves into slot 0 and retire,
// the others bid for slot 1.  Every cell-map read is done by now: the map is restored to all-empty here.
__global__ __launch_bounds__(256) void vox_place_kernel(const float* __restrict__ pts, VoxBatch B, int ndim, int max_voxels, int max_points,
                                                        long cells, const int* __restrict__ keys, const int* __restrict__ vid_of_point,
                                                        int* __restrict__ pvid, int* __restrict__ cell_map, int* __restrict__ slot_idx,
                                                        float* __restrict__ voxels) {
    int c, i;
    vox_locate(B, blockIdx.x, threadIdx.x, c, i);
    if (i < 0) return;
    const size_t base = (size_t)B.off[c], gi = base + i;
    const int o = pvid[gi];
    if (o < 0) return;
    if (o == i) cell_map[(size_t)c * cells + keys[gi]] = kEmpty;  // every touched cell has exactly one owner: one random write per cell
    const int v = vid_of_point[base + o];
    if (v >= max_voxels) {
        pvid[gi] = -1;
        return;
    }
    if (o == i) {
        float* dst = voxels + ((size_t)c * max_voxels + v) * max_points * ndim;
        for (int k = 0; k < ndim; ++k) dst[k] = pts[gi * ndim + k];
        pvid[gi] = -1;
    } else {
        pvid[gi] = v;
        if (max_points > 1) atomicMin(&slot_idx[((size_t)c * max_voxels + v) * max_points + 1], i);
    }
}

// round r >= 1: a point that won slot r (slot_idx == i) copies itself into voxels[vid][r] and retires; every other live point bids
// for slot r + 1
__global__ __launch_bounds__(256) void vox_round_kernel(const float* __restrict__ pts, VoxBatch B, int ndim, int r, int max_voxels, int max_points,
                                                        int nblocks, int* __restrict__ pvid, int* __restrict__ slot_idx, float* __restrict__ voxels) {
    // four 256-point blocks per workgroup: most points have retired after the first rounds, a workgroup then only reads 4 KB of pvid
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int b = blockIdx.x * 4 + j;
        if (b >= nblocks) return;
        int c, i;
        vox_locate(B, b, threadIdx.x, c, i);
        if (i < 0) continue;
        const size_t gi = (size_t)B.off[c] + i;
        const int v = pvid[gi];
        if (v < 0) continue;
        int* slots = slot_idx + ((size_t)c * max_voxels + v) * max_points;
        if (slots[r] == i) {
            float* dst = voxels + (((size_t)c * max_voxels + v) * max_points + r) * ndim;
            for (int k = 0; k < ndim; ++k) dst[k] = pts[gi * ndim + k];
            pvid[gi] = -1;
        } else if (r + 1 < max_points) {
            atomicMin(&slots[r + 1], i);
        }
    }
}

// one thread per (voxel, slot): empty slots are zero-filled (consecutive threads = consecutive slots: whole lines), the last filled slot
// names the count (filled slots are contiguous from 0: a point only bids for r + 1 after losing r).  grid.y = cloud
__global__ __launch_bounds__(256) void vox_finalize_kernel(const int* __restrict__ slot_idx, const int* __restrict__ num_voxels, int max_voxels,
                                                           int max_points, int ndim, float* __restrict__ voxels, int* __restrict__ num_points) {
    const int c = blockIdx.y;
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    const int v = (int)(t / max_points), r = (int)(t - (long)v * max_points);
    if (v >= num_voxels[c]) return;
    const size_t row = (size_t)c * max_voxels + v;
    const int* sl = slot_idx + row * max_points;
    if (sl[r] == kSlotEmpty) {
        float* dst = voxels + (row * max_points + r) * ndim;
        for (int k = 0; k < ndim; ++k) dst[k] = 0.0f;
    } else if (r + 1 == max_points || sl[r + 1] == kSlotEmpty) {
        num_points[row] = r + 1;
    }
}

// one thread per (voxel, channel), eight channel lanes per voxel (a loop beyond 8 channels): the filled slots summed in order / their
// number.  (Folded into the finalize pass - the thread of the last filled slot summing all channels - it ran 140 us slower per 16
// clouds: one lane per voxel walks the slots while its 63 neighbours wait.)
__global__ __launch_bounds__(256) void vox_mean_kernel(const float* __restrict__ voxels, const int* __restrict__ num_points,
                                                       const int* __restrict__ num_voxels, int max_voxels, int max_points, int ndim,
                                                       float* __restrict__ mean) {
    const int c = blockIdx.y;
    const int v = blockIdx.x * 32 + (threadIdx.x >> 3);
    if (v >= num_voxels[c]) return;
    const size_t row = (size_t)c * max_voxels + v;
    const int cnt = num_points[row];
    const float* src = voxels + row * max_points * ndim;
    for (int k = threadIdx.x & 7; k < ndim; k += 8) {
        float s = 0.0f;
        for (int r = 0; r < cnt; ++r) s += src[r * ndim + k];
        mean[row * ndim + k] = s / (float)cnt;
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
    VoxWs(long P, int nblocks, int nclouds, int max_voxels, int max_points) {
        size_t o = 0;
        const size_t p = align_up((size_t)(P > 0 ? P : 1) * sizeof(int), 256);
        keys = o; o += p;
        pvid = o; o += p;
        vidp = o; o += p;
        bsum = o; o += align_up((size_t)(nblocks + 1) * sizeof(int), 256);
        slots = o; o += align_up((size_t)nclouds * max_voxels * max_points * sizeof(int), 256);
        total = o;
    }
};

static int vox_batch(const float* points, const int* h_offsets, int n, int ndim, const VoxGrid& G, int max_points, int max_voxels,
                     float* voxels, int32_t* coors, int32_t* num_points_per_voxel, float* mean, int32_t* num_voxels, int32_t* cell_map,
                     void* workspace, size_t workspace_bytes, hipStream_t st) {
    VoxBatch B;
    B.n = n;
    B.off[0] = B.boff[0] = 0;
    for (int c = 0; c < n; ++c) {
        const int pc = h_offsets[c + 1] - h_offsets[c];
        if (pc < 0) {
            set_error_msg("voxelize: offsets must not decrease");
            return SHASTA_E_ARG;
        }
        B.off[c + 1] = B.off[c] + pc;
        B.boff[c + 1] = B.boff[c] + cdiv(pc, 256);
    }
    for (int c = n + 1; c <= kMaxClouds; ++c) B.off[c] = B.off[n], B.boff[c] = B.boff[n];
    const long P = B.off[n];
    const int nb = B.boff[n];
    const long cells = (long)G.g[0] * G.g[1] * G.g[2];
    const VoxWs L(P, nb, n, max_voxels, max_points);
    if (workspace_bytes < L.total) {
        set_error_msg("voxelize: workspace too small");
        return SHASTA_E_WORKSPACE;
    }
    char* ws = static_cast<char*>(workspace);
    int* keys = reinterpret_cast<int*>(ws + L.keys);
    int* pvid = reinterpret_cast<int*>(ws + L.pvid);
    int* vidp = reinterpret_cast<int*>(ws + L.vidp);
    int* bsum = reinterpret_cast<int*>(ws + L.bsum);
    int* slots = reinterpret_cast<int*>(ws + L.slots);
    hipError_t e = hipMemsetAsync(num_voxels, 0, (size_t)n * sizeof(int32_t), st);  // (clouds without points never write theirs)
    if (e == hipSuccess && nb > 0 && max_voxels > 0) e = hipMemsetAsync(slots, 0x7f, (size_t)n * max_voxels * max_points * sizeof(int), st);
    if (e != hipSuccess) {
        set_error("voxelize: memset", e);
        return SHASTA_E_LAUNCH;
    }
    if (nb == 0 || max_voxels == 0) return SHASTA_OK;
    int rc;
    hipLaunchKernelGGL(vox_key_kernel, dim3(nb), dim3(256), 0, st, points, B, ndim, G, cells, keys, cell_map);
    if ((rc = check_launch("vox_key"))) return rc;
    hipLaunchKernelGGL(vox_owner_kernel, dim3(nb), dim3(256), 0, st, keys, cell_map, B, cells, pvid, bsum);
    if ((rc = check_launch("vox_owner"))) return rc;
    hipLaunchKernelGGL(vox_scan_kernel, dim3(1), dim3(1024), 0, st, bsum, nb);
    if ((rc = check_launch("vox_scan"))) return rc;
    hipLaunchKernelGGL(vox_assign_kernel, dim3(nb), dim3(256), 0, st, keys, pvid, B, bsum, G, max_voxels, max_points, vidp, coors, num_voxels, slots);
    if ((rc = check_launch("vox_assign"))) return rc;
    hipLaunchKernelGGL(vox_place_kernel, dim3(nb), dim3(256), 0, st, points, B, ndim, max_voxels, max_points, cells, keys, vidp, pvid, cell_map,
                       slots, voxels);
    if ((rc = check_launch("vox_place"))) return rc;
    for (int r = 1; r < max_points; ++r) {
        hipLaunchKernelGGL(vox_round_kernel, dim3(cdiv(nb, 4)), dim3(256), 0, st, points, B, ndim, r, max_voxels, max_points, nb, pvid, slots, voxels);
        if ((rc = check_launch("vox_round"))) return rc;
    }
    hipLaunchKernelGGL(vox_finalize_kernel, dim3((unsigned)(((long)max_voxels * max_points + 255) / 256), n), dim3(256), 0, st, slots, num_voxels,
                       max_voxels, max_points, ndim, voxels, num_points_per_voxel);
    if ((rc = check_launch("vox_finalize"))) return rc;
    if (mean)
        hipLaunchKernelGGL(vox_mean_kernel, dim3((unsigned)cdiv(max_voxels, 32), n), dim3(256), 0, st, voxels, num_points_per_voxel,
                           num_voxels, max_voxels, max_points, ndim, mean);
    return check_launch("vox_finalize");
}

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
    return VoxWs(num_points, cdiv(num_points > 0 ? num_points : 1, 256), 1, max_voxels, max_points).total;
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
    const int off[2] = {0, P};
    return vox_batch(points, off, 1, ndim, G, max_points, max_voxels, voxels, coors, num_points_per_voxel, mean, num_voxels, cell_map, workspace,
                     workspace_bytes, as_stream(stream));
}

extern "C" size_t shasta_voxelize_batch_workspace_bytes(const int* h_offsets, int num_clouds, int max_voxels, int max_points) {
    if (!h_offsets || num_clouds < 1 || num_clouds > kMaxClouds || max_voxels < 0 || max_points < 1) return 0;
    int nb = 0;
    for (int c = 0; c < num_clouds; ++c) {
        if (h_offsets[c + 1] < h_offsets[c]) return 0;
        nb += cdiv(h_offsets[c + 1] - h_offsets[c], 256);
    }
    return VoxWs((long)h_offsets[num_clouds] - h_offsets[0], nb, num_clouds, max_voxels, max_points).total;
}

extern "C" int shasta_voxelize_mean_batch_f32(const float* points, const int* h_offsets, int num_clouds, int ndim, const float* h_range6,
                                              const float* h_voxel3, int max_points, int max_voxels, float* voxels, int32_t* coors,
                                              int32_t* num_points_per_voxel, float* mean, int32_t* num_voxels, int32_t* cell_maps,
                                              void* workspace, size_t workspace_bytes, shasta_stream_t stream) {
    SHASTA_REQUIRE(h_offsets && h_range6 && h_voxel3 && num_voxels && cell_maps && workspace, "voxelize_batch: null pointer");
    SHASTA_REQUIRE(num_clouds >= 1 && num_clouds <= kMaxClouds, "voxelize_batch: 1 to 32 clouds per call");
    SHASTA_REQUIRE(ndim >= 3 && max_points >= 1 && max_voxels >= 0 && h_offsets[0] >= 0, "voxelize_batch: bad size");
    SHASTA_REQUIRE(h_offsets[num_clouds] == h_offsets[0] || points, "voxelize_batch: null points");
    SHASTA_REQUIRE(max_voxels == 0 || (voxels && coors && num_points_per_voxel), "voxelize_batch: null outputs");
    const VoxGrid G = make_grid(h_range6, h_voxel3);
    SHASTA_REQUIRE(G.g[0] > 0 && G.g[1] > 0 && G.g[2] > 0, "voxelize_batch: empty grid");
    SHASTA_REQUIRE((double)G.g[0] * G.g[1] * G.g[2] < 2147483647.0, "voxelize_batch: grid too large for int32 keys");
    return vox_batch(points ? points + (size_t)h_offsets[0] * ndim : nullptr, h_offsets, num_clouds, ndim, G, max_points, max_voxels, voxels, coors,
                     num_points_per_voxel, mean, num_voxels, cell_maps, workspace, workspace_bytes, as_stream(stream));
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
