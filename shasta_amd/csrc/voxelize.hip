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
//   1. key[i] = z*gy*gx + y*gx + x ; cell[key] = min(point index)                  (atomicMin)
//   2. owner(i) = cell[key[i]] (one random read per point); creator(i) = owner(i) == i ; voxel id = exclusive prefix count of
//      creators (scan); the creator takes slot 0
//   3. every other point inserts itself into the voxel's slots 1..max_points-1, kept sorted by point index (atomicMin leaves the
//      smaller value in the slot, the larger one moves on): the slots end up with the first max_points points, exactly the ones the
//      serial loop keeps
//   4. per voxel the filled slots are counted, their points copied in and the other slots zeroed (whole rows, put together in LDS),
//      mean = (sum over slots) / count.
// cell[.]: the reference keeps a dense (40, 1440, 1440) int32 map of the grid (332 MB for nuScenes, allocated on every call,
// point_cloud_ops.py:150).  A cloud of P points touches at most P cells: here the map is an open-addressing hash table of
// 2^ceil(log2(2 P)) (key, first point) entries per cloud (8 MB for 3e5 points), cleared by one memset per call - random accesses over
// 5.3 GB of dense maps (16 clouds) were most of the chain's time (TLB and DRAM page misses on every point).
#include "common.hpp"
#include <math.h>
#include <limits.h>

namespace shasta {

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
// every cloud has its own hash table (table + c * 2^tbits entries) and its own rows of the outputs.
constexpr int kMaxClouds = 32;
struct VoxBatch {
    int n, tbits;              // clouds; log2 of the entries of a cloud's hash table
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

// cell -> first point: one 64-bit word per entry, (key << 32) | point index, empty = all ones (what the clearing memset leaves).
// ONE atomic per probe: atomicMin of the packed pair.  Same key in the slot -> the smaller point index stays (the map's purpose).  Another
// key in the slot: the numerically smaller pair stays and the larger one - ours, or the one we just displaced - moves on to the next
// slot (linear probing; an entry only ever moves forward along its own probe path, so a look-up that walks from the key's home slot
// finds it).  Tables are at most half full.
typedef unsigned long long VoxCell;
__device__ __forceinline__ unsigned vox_hash(unsigned key, int bits) { return (key * 2654435761u) >> (32 - bits); }
__device__ __forceinline__ void vox_insert(VoxCell* __restrict__ t, int bits, unsigned key, unsigned i) {
    const unsigned mask = (1u << bits) - 1u;
    unsigned h = vox_hash(key, bits);
    VoxCell mine = ((VoxCell)key << 32) | i;
    for (;;) {
        const VoxCell old = atomicMin(&t[h], mine);
        if (old == ~0ull || (unsigned)(old >> 32) == (unsigned)(mine >> 32)) return;  // took an empty slot / merged with our key
        if (old > mine) mine = old;  // we displaced another key's entry: carry it on
        h = (h + 1) & mask;
    }
}
__device__ __forceinline__ int vox_lookup(const VoxCell* __restrict__ t, int bits, unsigned key) {
    const unsigned mask = (1u << bits) - 1u;
    unsigned h = vox_hash(key, bits);
    for (;;) {
        const VoxCell e = t[h];
        if ((unsigned)(e >> 32) == key) return (int)(unsigned)e;
        h = (h + 1) & mask;  // (the key is in the table: every point was inserted by the pass before)
    }
}

__global__ __launch_bounds__(256) void vox_key_kernel(const float* __restrict__ pts, VoxBatch B, int ndim, VoxGrid G,
                                                      int* __restrict__ keys, VoxCell* __restrict__ table) {
    int c, i;
    vox_locate(B, blockIdx.x, threadIdx.x, c, i);
    if (i < 0) return;
    const size_t gi = (size_t)B.off[c] + i;
    int cc[3];
    bool ok = true;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const float f = floorf(__fdiv_rn(__fsub_rn(pts[gi * ndim + j], G.lo[j]), G.vs[j]));
        if (!(f >= 0.0f && f < (float)G.g[j])) ok = false;  // a NaN coordinate drops the point (undefined in the reference: its int cast indexes the map)
        cc[j] = ok ? (int)f : 0;
    }
    int key = -1;
    if (ok) {
        key = (cc[2] * G.g[1] + cc[1]) * G.g[0] + cc[0];
        vox_insert(table + ((size_t)c << B.tbits), B.tbits, (unsigned)key, (unsigned)i);
    }
    keys[gi] = key;
}

// owner[i] = the first point of point i's cell (the point that creates the voxel) or -1 for a dropped point - the ONE look-up of the
// table per point; the block's number of creators for the scan
__global__ __launch_bounds__(256) void vox_owner_kernel(const int* __restrict__ keys, const VoxCell* __restrict__ table, VoxBatch B,
                                                        int* __restrict__ owner, int* __restrict__ block_sums) {
    int c, i;
    vox_locate(B, blockIdx.x, threadIdx.x, c, i);
    bool flag = false;
    if (i >= 0) {
        const size_t gi = (size_t)B.off[c] + i;
        const int key = keys[gi];
        const int o = key >= 0 ? vox_lookup(table + ((size_t)c << B.tbits), B.tbits, (unsigned)key) : -1;
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

// Every point that did not create its voxel inserts itself into the voxel's slots 1 .. max_points - 1, kept SORTED by point index:
// at slot r it leaves the smaller of (itself, the slot's content) there - one atomicMin - and carries the larger one on to slot r + 1;
// a value that meets an empty slot stays, one that runs off the end is dropped.  Whatever the arrival order, slot r ends up with the
// (r + 1)-th smallest point index of the voxel: every value but the smallest passes on from slot 0, so slot 1 collects the minimum of
// the rest, and so on - exactly the points, in the order, the serial loop keeps.  A point whose index is already above the last slot's
// content can never get in (slots only decrease) and leaves after one read: crowded voxels stop costing atomics once they hold small
// indices.  This one pass replaces the placement pass and max_points - 1 bidding rounds of rounds 1 - 5 (an atomic per live point and
// round).  The creator sits in slot 0 since vox_assign.
__global__ __launch_bounds__(256) void vox_insert_kernel(VoxBatch B, int max_voxels, int max_points, const int* __restrict__ owner,
                                                         const int* __restrict__ vid_of_point, int* __restrict__ slot_idx) {
    int c, i;
    vox_locate(B, blockIdx.x, threadIdx.x, c, i);
    if (i < 0) return;
    const size_t base = (size_t)B.off[c];
    const int o = owner[base + i];
    if (o < 0 || o == i) return;
    const int v = vid_of_point[base + o];
    if (v >= max_voxels) return;
    int* slots = slot_idx + ((size_t)c * max_voxels + v) * max_points;
    int p = i;
    if (__hip_atomic_load(&slots[max_points - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < p) return;
    for (int r = 1; r < max_points; ++r) {
        const int old = atomicMin(&slots[r], p);
        if (old == kSlotEmpty) return;
        if (old > p) p = old;
    }
}

// The outputs of 32 voxels per workgroup.  The filled slots of a voxel (contiguous from 0) name its points, the other slots are the
// zero padding of the reference's np.zeros output; for the reader (voxel_encoder.py:18-28) all max_points slots are summed in slot
// order - padding included, as points.sum(dim=1) does - and divided by the count.  The 32 complete rows (max_points * ndim floats each,
// back to back in the output) are put together in LDS: one thread per (voxel, slot) - the slot words of the 32 voxels are one linear
// run, and every point fetch of the workgroup is in flight at once (a loop over the slots per voxel is a chain of max_points dependent
// slot -> point loads: 0.25 ms per 16 clouds, latency bound) - then eight lanes per voxel count and sum, and the rows leave as one
// linear run of 16-byte stores.  Rounds 1 - 6a cleared the whole (max_voxels, max_points, ndim) output with a memset in front of the
// chain (512 MB per 16 clouds) and stored the kept points over it in 20-byte pieces.  Rows >= V are not written.  grid.y = cloud
template <int NDIM>
__device__ __forceinline__ void vox_copy_point(const float* __restrict__ src, float* dst, int ndim, bool filled) {
    if (NDIM > 0) {
        float v[NDIM > 0 ? NDIM : 1];
#pragma unroll
        for (int k = 0; k < NDIM; ++k) v[k] = filled ? src[k] : 0.0f;
#pragma unroll
        for (int k = 0; k < NDIM; ++k) dst[k] = v[k];
    } else {
        for (int k = 0; k < ndim; ++k) dst[k] = filled ? src[k] : 0.0f;
    }
}
template <int NDIM>
__global__ __launch_bounds__(256) void vox_finalize_kernel(const float* __restrict__ pts, VoxBatch B, const int* __restrict__ slot_idx,
                                                           const int* __restrict__ num_voxels, int max_voxels, int max_points, int ndim_rt,
                                                           float* __restrict__ voxels, int* __restrict__ num_points, float* __restrict__ mean) {
    extern __shared__ __attribute__((aligned(16))) float vox_tile[];  // [32][max_points][ndim]
    const int ndim = NDIM > 0 ? NDIM : ndim_rt;
    const int c = blockIdx.y, V = num_voxels[c], v0 = blockIdx.x * 32;
    if (v0 >= V) return;  // the whole workgroup
    const int nv = min(32, V - v0), rowlen = max_points * ndim;
    const size_t row0 = (size_t)c * max_voxels + v0;
    const int* sl = slot_idx + row0 * max_points;
    const float* cloud = pts + (size_t)B.off[c] * ndim;
    for (int p = threadIdx.x; p < nv * max_points; p += 256) {
        const int idx = sl[p];
        const bool filled = idx != kSlotEmpty;
        vox_copy_point<NDIM>(cloud + (size_t)(filled ? idx : 0) * ndim, vox_tile + (size_t)p * ndim, ndim, filled);
    }
    __syncthreads();
    const int g = threadIdx.x >> 3, k0 = threadIdx.x & 7;
    if (g < nv) {
        int cnt = 0;
        for (int r = k0; r < max_points; r += 8) cnt += sl[g * max_points + r] != kSlotEmpty;
        cnt += __shfl_xor(cnt, 1, 64);
        cnt += __shfl_xor(cnt, 2, 64);
        cnt += __shfl_xor(cnt, 4, 64);
        if (k0 == 0) num_points[row0 + g] = cnt;
        if (mean) {
            for (int k = k0; k < ndim; k += 8) {
                float s = 0.0f;
                for (int r = 0; r < max_points; ++r) s += vox_tile[g * rowlen + r * ndim + k];
                mean[(row0 + g) * ndim + k] = s / (float)cnt;
            }
        }
    }
    const int nfl = nv * rowlen;
    float* out = voxels + row0 * rowlen;
    int done = 0;
    if (((uintptr_t)out & 15) == 0) {
        for (int i = threadIdx.x; i < nfl / 4; i += 256)
            __builtin_nontemporal_store(reinterpret_cast<const f32x4*>(vox_tile)[i], reinterpret_cast<f32x4*>(out) + i);
        done = nfl & ~3;
    }
    for (int i = done + threadIdx.x; i < nfl; i += 256) out[i] = vox_tile[i];
}

// The same for rows too long for LDS (32 x max_points x ndim floats above 64 KB): eight lanes per voxel (lane = channel; a loop beyond
// 8 channels) walk the slots and store straight into the output.
__global__ __launch_bounds__(256) void vox_finalize_direct_kernel(const float* __restrict__ pts, VoxBatch B, const int* __restrict__ slot_idx,
                                                                  const int* __restrict__ num_voxels, int max_voxels, int max_points, int ndim,
                                                                  float* __restrict__ voxels, int* __restrict__ num_points, float* __restrict__ mean) {
    const int c = blockIdx.y;
    const int v = blockIdx.x * 32 + (threadIdx.x >> 3), k0 = threadIdx.x & 7;
    if (v >= num_voxels[c]) return;
    const size_t row = (size_t)c * max_voxels + v;
    const int* sl = slot_idx + row * max_points;
    const float* cloud = pts + (size_t)B.off[c] * ndim;
    float* dst = voxels + row * max_points * ndim;
    int cnt = 0;
    bool live = true;
    float s[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    for (int r = 0; r < max_points; ++r) {
        const int idx = live ? sl[r] : kSlotEmpty;
        if (idx == kSlotEmpty) live = false;
        else ++cnt;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (k0 + 8 * j < ndim) {
                const float val = live ? cloud[(size_t)idx * ndim + k0 + 8 * j] : 0.0f;
                dst[r * ndim + k0 + 8 * j] = val;
                s[j] += val;
            }
        }
    }
    if (k0 == 0) num_points[row] = cnt;
    if (mean) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (k0 + 8 * j < ndim) mean[row * ndim + k0 + 8 * j] = s[j] / (float)cnt;
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

static int vox_table_bits(int max_cloud_points) {
    int bits = 10;
    while (bits < 30 && (1L << bits) < 2L * max_cloud_points) ++bits;
    return bits;
}

struct VoxWs {
    size_t keys, pvid, vidp, bsum, slots, table, total;
    VoxWs(long P, int nblocks, int nclouds, int max_voxels, int max_points, int tbits) {
        size_t o = 0;
        const size_t p = align_up((size_t)(P > 0 ? P : 1) * sizeof(int), 256);
        keys = o; o += p;
        pvid = o; o += p;
        vidp = o; o += p;
        bsum = o; o += align_up((size_t)(nblocks + 1) * sizeof(int), 256);
        slots = o; o += align_up((size_t)nclouds * max_voxels * max_points * sizeof(int), 256);
        table = o; o += ((size_t)nclouds << tbits) * sizeof(VoxCell);
        total = o;
    }
};

static int vox_batch(const float* points, const int* h_offsets, int n, int ndim, const VoxGrid& G, int max_points, int max_voxels,
                     float* voxels, int32_t* coors, int32_t* num_points_per_voxel, float* mean, int32_t* num_voxels,
                     void* workspace, size_t workspace_bytes, hipStream_t st) {
    VoxBatch B;
    B.n = n;
    B.off[0] = B.boff[0] = 0;
    int pmax = 0;
    for (int c = 0; c < n; ++c) {
        const int pc = h_offsets[c + 1] - h_offsets[c];
        pmax = pc > pmax ? pc : pmax;
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
    B.tbits = vox_table_bits(pmax);
    const VoxWs L(P, nb, n, max_voxels, max_points, B.tbits);
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
    VoxCell* table = reinterpret_cast<VoxCell*>(ws + L.table);
    hipError_t e = hipMemsetAsync(num_voxels, 0, (size_t)n * sizeof(int32_t), st);  // (clouds without points never write theirs)
    if (e == hipSuccess && nb > 0 && max_voxels > 0) e = hipMemsetAsync(table, 0xff, ((size_t)n << B.tbits) * sizeof(VoxCell), st);
    if (e == hipSuccess && nb > 0 && max_voxels > 0) e = hipMemsetAsync(slots, 0x7f, (size_t)n * max_voxels * max_points * sizeof(int), st);
    if (e != hipSuccess) {
        set_error("voxelize: memset", e);
        return SHASTA_E_LAUNCH;
    }
    if (nb == 0 || max_voxels == 0) return SHASTA_OK;
    int rc;
    hipLaunchKernelGGL(vox_key_kernel, dim3(nb), dim3(256), 0, st, points, B, ndim, G, keys, table);
    if ((rc = check_launch("vox_key"))) return rc;
    hipLaunchKernelGGL(vox_owner_kernel, dim3(nb), dim3(256), 0, st, keys, table, B, pvid, bsum);
    if ((rc = check_launch("vox_owner"))) return rc;
    hipLaunchKernelGGL(vox_scan_kernel, dim3(1), dim3(1024), 0, st, bsum, nb);
    if ((rc = check_launch("vox_scan"))) return rc;
    hipLaunchKernelGGL(vox_assign_kernel, dim3(nb), dim3(256), 0, st, keys, pvid, B, bsum, G, max_voxels, max_points, vidp, coors, num_voxels, slots);
    if ((rc = check_launch("vox_assign"))) return rc;
    if (max_points > 1) {
        hipLaunchKernelGGL(vox_insert_kernel, dim3(nb), dim3(256), 0, st, B, max_voxels, max_points, pvid, vidp, slots);
        if ((rc = check_launch("vox_insert"))) return rc;
    }
    const size_t tile = (size_t)32 * max_points * ndim * sizeof(float);  // 6.4 KB for 10 x 5
    const dim3 fgrid((unsigned)cdiv(max_voxels, 32), n);
    if (tile > 64 * 1024)
        hipLaunchKernelGGL(vox_finalize_direct_kernel, fgrid, dim3(256), 0, st, points, B, slots, num_voxels, max_voxels, max_points, ndim, voxels,
                           num_points_per_voxel, mean);
    else if (ndim == 5)
        hipLaunchKernelGGL(vox_finalize_kernel<5>, fgrid, dim3(256), tile, st, points, B, slots, num_voxels, max_voxels, max_points, ndim, voxels,
                           num_points_per_voxel, mean);
    else if (ndim == 4)
        hipLaunchKernelGGL(vox_finalize_kernel<4>, fgrid, dim3(256), tile, st, points, B, slots, num_voxels, max_voxels, max_points, ndim, voxels,
                           num_points_per_voxel, mean);
    else
        hipLaunchKernelGGL(vox_finalize_kernel<0>, fgrid, dim3(256), tile, st, points, B, slots, num_voxels, max_voxels, max_points, ndim, voxels,
                           num_points_per_voxel, mean);
    return check_launch("vox_finalize");
}

}  // namespace shasta

using namespace shasta;

extern "C" size_t shasta_voxelize_workspace_bytes(int num_points, int max_voxels, int max_points) {
    if (num_points < 0 || max_voxels < 0 || max_points < 1) return 0;
    return VoxWs(num_points, cdiv(num_points > 0 ? num_points : 1, 256), 1, max_voxels, max_points, vox_table_bits(num_points)).total;
}

static int vox_check_grid(const VoxGrid& G) {
    SHASTA_REQUIRE(G.g[0] > 0 && G.g[1] > 0 && G.g[2] > 0, "voxelize: empty grid");
    SHASTA_REQUIRE((double)G.g[0] * G.g[1] * G.g[2] < 2147483647.0, "voxelize: grid too large for int32 keys");
    return SHASTA_OK;
}

extern "C" int shasta_voxelize_mean_f32(const float* points, int P, int ndim, const float* h_range6,
                                        const float* h_voxel3, int max_points, int max_voxels, float* voxels,
                                        int32_t* coors, int32_t* num_points_per_voxel, float* mean,
                                        int32_t* num_voxels, void* workspace, size_t workspace_bytes,
                                        shasta_stream_t stream) {
    SHASTA_REQUIRE(h_range6 && h_voxel3 && num_voxels && workspace, "voxelize: null pointer");
    SHASTA_REQUIRE(P >= 0 && ndim >= 3 && ndim <= 32 && max_points >= 1 && max_voxels >= 0, "voxelize: bad size");
    SHASTA_REQUIRE(P == 0 || points, "voxelize: null points");
    SHASTA_REQUIRE(max_voxels == 0 || (voxels && coors && num_points_per_voxel), "voxelize: null outputs");
    const VoxGrid G = make_grid(h_range6, h_voxel3);
    int rc = vox_check_grid(G);
    if (rc) return rc;
    const int off[2] = {0, P};
    return vox_batch(points, off, 1, ndim, G, max_points, max_voxels, voxels, coors, num_points_per_voxel, mean, num_voxels, workspace,
                     workspace_bytes, as_stream(stream));
}

extern "C" size_t shasta_voxelize_batch_workspace_bytes(const int* h_offsets, int num_clouds, int max_voxels, int max_points) {
    if (!h_offsets || num_clouds < 1 || num_clouds > kMaxClouds || max_voxels < 0 || max_points < 1) return 0;
    int nb = 0, pmax = 0;
    for (int c = 0; c < num_clouds; ++c) {
        const int pc = h_offsets[c + 1] - h_offsets[c];
        if (pc < 0) return 0;
        nb += cdiv(pc, 256);
        pmax = pc > pmax ? pc : pmax;
    }
    return VoxWs((long)h_offsets[num_clouds] - h_offsets[0], nb, num_clouds, max_voxels, max_points, vox_table_bits(pmax)).total;
}

extern "C" int shasta_voxelize_mean_batch_f32(const float* points, const int* h_offsets, int num_clouds, int ndim, const float* h_range6,
                                              const float* h_voxel3, int max_points, int max_voxels, float* voxels, int32_t* coors,
                                              int32_t* num_points_per_voxel, float* mean, int32_t* num_voxels, void* workspace,
                                              size_t workspace_bytes, shasta_stream_t stream) {
    SHASTA_REQUIRE(h_offsets && h_range6 && h_voxel3 && num_voxels && workspace, "voxelize_batch: null pointer");
    SHASTA_REQUIRE(num_clouds >= 1 && num_clouds <= kMaxClouds, "voxelize_batch: 1 to 32 clouds per call");
    SHASTA_REQUIRE(ndim >= 3 && ndim <= 32 && max_points >= 1 && max_voxels >= 0 && h_offsets[0] >= 0, "voxelize_batch: bad size");
    SHASTA_REQUIRE(h_offsets[num_clouds] == h_offsets[0] || points, "voxelize_batch: null points");
    SHASTA_REQUIRE(max_voxels == 0 || (voxels && coors && num_points_per_voxel), "voxelize_batch: null outputs");
    const VoxGrid G = make_grid(h_range6, h_voxel3);
    int rc = vox_check_grid(G);
    if (rc) return rc;
    return vox_batch(points ? points + (size_t)h_offsets[0] * ndim : nullptr, h_offsets, num_clouds, ndim, G, max_points, max_voxels, voxels, coors,
                     num_points_per_voxel, mean, num_voxels, workspace, workspace_bytes, as_stream(stream));
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
