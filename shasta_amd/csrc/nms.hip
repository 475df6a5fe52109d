// Next row f-2 (second half): rotated BEV NMS.
// Replaces det3d/ops/iou3d_nms: nms_kernel (src/iou3d_nms_kernel.cu:267-311, pairwise rotated BEV IoU > thresh packed into
// 64-bit masks), the host-side mask reduction of nms_gpu (src/iou3d_nms.cpp:100-143) and iou_bev (:226-233,
// IoU = overlap / max(area_a + area_b - overlap, 1e-8)).  Boxes are [x, y, z, dx, dy, dz, heading] sorted by descending
// score (the caller sorts, like iou3d_nms_utils.py:74-89).  Two launches, nothing returns to the host in between:
//   1. mask[i][w] bit b = IoU_bev(box i, box 64w + b) > thresh for 64w + b > i   (one lane per pair, the wave ballot is the
//      mask word; boxes staged in LDS; blocks below the diagonal are skipped - the reduction never needs them)
//   2. one wavefront walks the boxes in score order: lane l owns words l, l+64, ... of the `removed` bitmap; a box that is
//      not removed is kept and ORs its mask row in (mask rows are fetched a chunk of boxes ahead).
// The overlap is the float64 Sutherland-Hodgman clip of the two rectangles (geom2d.hpp) = the exact intersection area.  The
// reference's float32 edge-intersection + angular-sort routine (iou3d_nms_kernel.cu:104-225) yields the same area up to its
// rounding EXCEPT that it counts a corner up to 1e-2 outside the other box as inside (check_in_box2d, :51-61): for nearly
// touching or nearly coincident edges its area is larger by up to about 1e-2 x edge length.  Decisions can therefore differ for
// IoUs within that distance of the threshold (oracle/nms_oracle.py restates both; tests/test_nms.py measures the gap).
// Also here: nms_normal (axis-aligned IoU, bit-identical decisions) and the pairwise matrices boxes_overlap_bev / boxes_iou_bev /
// boxes_iou3d of the same reference module.
#include "common.hpp"
#include "geom2d.hpp"

namespace shasta {

__device__ __forceinline__ void bev_corners(const float* b, P2* c) {
    const double cx = b[0], cy = b[1], hx = 0.5 * (double)b[3], hy = 0.5 * (double)b[4];
    const double cs = cos((double)b[6]), sn = sin((double)b[6]);
    const double ux[4] = {-hx, hx, hx, -hx}, uy[4] = {-hy, -hy, hy, hy};
#pragma unroll
    for (int k = 0; k < 4; ++k) c[k] = {cx + ux[k] * cs - uy[k] * sn, cy + ux[k] * sn + uy[k] * cs};
}

__device__ __forceinline__ double iou_bev(const float* a, const float* b) {
    // cheap reject: centres further apart than the two half-diagonals
    const double dx = (double)a[0] - b[0], dy = (double)a[1] - b[1];
    const double ra = 0.5 * sqrt((double)a[3] * a[3] + (double)a[4] * a[4]), rb = 0.5 * sqrt((double)b[3] * b[3] + (double)b[4] * b[4]);
    if (dx * dx + dy * dy > (ra + rb) * (ra + rb)) return 0.0;
    P2 ca[4], cb[4];
    bev_corners(a, ca);
    bev_corners(b, cb);
    const double ov = clip_area(ca, cb);
    const double sa = (double)a[3] * a[4], sb = (double)b[3] * b[4];
    return ov / fmax(sa + sb - ov, 1e-8);
}

// iou_normal (iou3d_nms_kernel.cu:313-323): the axis-aligned IoU of the BEV footprints, heading ignored; plain fp32 in the
// reference's operation order (no contraction: -ffp-contract=off), so the decisions are the reference's bit for bit
__device__ __forceinline__ float iou_axis_aligned(const float* a, const float* b) {
    const float left = fmaxf(a[0] - a[3] / 2, b[0] - b[3] / 2), right = fminf(a[0] + a[3] / 2, b[0] + b[3] / 2);
    const float top = fmaxf(a[1] - a[4] / 2, b[1] - b[4] / 2), bottom = fminf(a[1] + a[4] / 2, b[1] + b[4] / 2);
    const float width = fmaxf(right - left, 0.f), height = fmaxf(bottom - top, 0.f);
    const float inter = width * height;
    const float sa = a[3] * a[4], sb = b[3] * b[4];
    return inter / fmaxf(sa + sb - inter, 1e-8f);
}

// block = (64-column block cb, 64-row block rb), 4 waves: a wave takes rows r = wave, wave + 4, ...; lane j evaluates the pair
// (row i, column 64 cb + j) and the wave's ballot IS the 64-bit mask word of that row (one thread per row with a 64-step
// loop left most of the chip idle: 136 waves for 1000 boxes)
template <bool AXIS_ALIGNED>
__global__ __launch_bounds__(256) void nms_mask_kernel(const float* __restrict__ boxes, int n, float thresh, int words,
                                                       unsigned long long* __restrict__ mask) {
    const int cb = blockIdx.x, rb = blockIdx.y;
    if (cb < rb) return;  // below the diagonal: never read
    __shared__ float col[64 * 7], row[64 * 7];
    const int ncol = min(64, n - cb * 64), nrow = min(64, n - rb * 64);
    for (int e = threadIdx.x; e < ncol * 7; e += 256) col[e] = boxes[(size_t)cb * 64 * 7 + e];
    for (int e = threadIdx.x; e < nrow * 7; e += 256) row[e] = boxes[(size_t)rb * 64 * 7 + e];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = cb * 64 + lane;
    for (int r = wave; r < nrow; r += 4) {
        const int i = rb * 64 + r;
        bool hit = lane < ncol && j > i;
        if (hit) {
            if constexpr (AXIS_ALIGNED) hit = iou_axis_aligned(row + r * 7, col + lane * 7) > thresh;
            else hit = iou_bev(row + r * 7, col + lane * 7) > (double)thresh;
        }
        const unsigned long long bits = __ballot(hit);
        if (lane == 0) mask[(size_t)i * words + cb] = bits;
    }
}

constexpr int NMS_MAX_WORDS_PER_LANE = 8;  // n <= 64 * 64 * 8 = 32768

// SLOTS = mask words per lane (lane l owns words l, l + 64, ...).  The walk over the boxes is sequential, so the latency of
// the mask rows must not be: rows are fetched CH = 32 / SLOTS at a time (all loads of a chunk in flight together) and then
// consumed from registers; a row-at-a-time version paid one global-load latency per box (0.57 us, 0.86 ms for 1000 boxes).
template <int SLOTS>
__global__ __launch_bounds__(64) void nms_reduce_kernel(const unsigned long long* __restrict__ mask, int n, int words,
                                                        int* __restrict__ keep, int* __restrict__ num_keep) {
    constexpr int CH = 32 / SLOTS;
    const int lane = threadIdx.x;
    unsigned long long removed[SLOTS];
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) removed[s] = 0ull;
    int count = 0;
    for (int base = 0; base < n; base += CH) {
        unsigned long long buf[CH][SLOTS];
#pragma unroll
        for (int r = 0; r < CH; ++r) {
            const int i = base + r;
#pragma unroll
            for (int s = 0; s < SLOTS; ++s) {
                const int w = lane + 64 * s;
                // words left of the diagonal block were never written: treat as zero
                buf[r][s] = (i < n && w < words && w >= (i >> 6)) ? mask[(size_t)i * words + w] : 0ull;
            }
        }
#pragma unroll
        for (int r = 0; r < CH; ++r) {
            const int i = base + r;
            const int w = i >> 6, owner = w & 63, slot = w >> 6;
            unsigned long long word = 0ull;
#pragma unroll
            for (int s = 0; s < SLOTS; ++s)
                if (s == slot) word = removed[s];
            word = __shfl(word, owner, 64);
            if (i < n && !((word >> (i & 63)) & 1ull)) {  // wave-uniform
                if (lane == 0) keep[count] = i;
                ++count;
#pragma unroll
                for (int s = 0; s < SLOTS; ++s) removed[s] |= buf[r][s];
            }
        }
    }
    if (lane == 0) *num_keep = count;
}

// Pairwise matrices of det3d/ops/iou3d_nms (boxes_overlap_kernel / boxes_iou_bev_kernel, iou3d_nms_kernel.cu:236-265, and the
// 3-D IoU that iou3d_nms_utils.py:35-72 builds on the overlap): one thread per (a, b) pair, boxes of the column block in LDS.
// mode 0: BEV overlap area; 1: BEV IoU = overlap / max(sa + sb - overlap, 1e-8); 2: 3-D IoU = overlap * overlap_h /
// max(vol_a + vol_b - overlap_3d, 1e-6) with the reference's fp32 operation order behind the (float64 -> fp32) overlap.
__global__ __launch_bounds__(256) void boxes_bev_kernel(const float* __restrict__ A, int na, const float* __restrict__ Bx, int nb, int mode,
                                                        float* __restrict__ out) {
    __shared__ float colb[64 * 7];
    const int b0 = blockIdx.x * 64, nbl = min(64, nb - b0);
    for (int e = threadIdx.x; e < nbl * 7; e += 256) colb[e] = Bx[(size_t)b0 * 7 + e];
    __syncthreads();
    const int lane = threadIdx.x & 63, i = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (i >= na || lane >= nbl) return;
    float a[7];
#pragma unroll
    for (int k = 0; k < 7; ++k) a[k] = A[(size_t)i * 7 + k];
    const float* b = colb + lane * 7;
    P2 ca[4], cb[4];
    double ovd = 0.0;
    {
        const double dx = (double)a[0] - b[0], dy = (double)a[1] - b[1];
        const double ra = 0.5 * sqrt((double)a[3] * a[3] + (double)a[4] * a[4]), rb = 0.5 * sqrt((double)b[3] * b[3] + (double)b[4] * b[4]);
        if (dx * dx + dy * dy <= (ra + rb) * (ra + rb)) {
            bev_corners(a, ca);
            bev_corners(b, cb);
            ovd = clip_area(ca, cb);
        }
    }
    const float ov = (float)ovd;
    float r = ov;
    if (mode == 1) {
        const float sa = a[3] * a[4], sb = b[3] * b[4];
        r = ov / fmaxf(sa + sb - ov, 1e-8f);
    } else if (mode == 2) {
        const float amax = a[2] + a[5] / 2, amin = a[2] - a[5] / 2, bmax = b[2] + b[5] / 2, bmin = b[2] - b[5] / 2;
        const float oh = fmaxf(fminf(amax, bmax) - fmaxf(amin, bmin), 0.f);
        const float o3 = ov * oh;
        const float va = a[3] * a[4] * a[5], vb = b[3] * b[4] * b[5];
        r = o3 / fmaxf(va + vb - o3, 1e-6f);
    }
    out[(size_t)i * nb + b0 + lane] = r;
}

}  // namespace shasta

using namespace shasta;

extern "C" int shasta_boxes_bev_f32(const float* boxes_a, int num_a, const float* boxes_b, int num_b, int mode, float* out,
                                    shasta_stream_t stream) {
    SHASTA_REQUIRE(num_a >= 0 && num_b >= 0 && mode >= 0 && mode <= 2, "boxes_bev: bad argument");
    if (num_a == 0 || num_b == 0) return SHASTA_OK;
    SHASTA_REQUIRE(boxes_a && boxes_b && out, "boxes_bev: null pointer");
    SHASTA_REQUIRE(num_a <= 4 * 65535, "boxes_bev: at most 262140 rows (boxes_a) per call");
    hipLaunchKernelGGL(boxes_bev_kernel, dim3(cdiv(num_b, 64), cdiv(num_a, 4)), dim3(256), 0, as_stream(stream), boxes_a, num_a, boxes_b,
                       num_b, mode, out);
    return check_launch("boxes_bev");
}

extern "C" size_t shasta_nms_workspace_bytes(int num_boxes) {
    const size_t words = (size_t)cdiv(std::max(num_boxes, 1), 64);
    return (size_t)std::max(num_boxes, 1) * words * sizeof(unsigned long long);
}

static int nms_impl(bool axis_aligned, const float* boxes_sorted, int num_boxes, float thresh, void* workspace, size_t workspace_bytes,
                    int32_t* keep, int32_t* num_keep, shasta_stream_t stream) {
    SHASTA_REQUIRE(keep && num_keep && num_boxes >= 0, axis_aligned ? "nms_normal: bad argument" : "nms_rotated: bad argument");
    SHASTA_REQUIRE(num_boxes <= 64 * 64 * NMS_MAX_WORDS_PER_LANE, axis_aligned ? "nms_normal: at most 32768 boxes" : "nms_rotated: at most 32768 boxes");
    hipStream_t st = as_stream(stream);
    if (num_boxes == 0) {
        hipError_t e = hipMemsetAsync(num_keep, 0, sizeof(int32_t), st);
        if (e != hipSuccess) {
            set_error(axis_aligned ? "nms_normal: memset" : "nms_rotated: memset", e);
            return SHASTA_E_LAUNCH;
        }
        return SHASTA_OK;
    }
    SHASTA_REQUIRE(boxes_sorted && workspace, axis_aligned ? "nms_normal: null pointer" : "nms_rotated: null pointer");
    if (workspace_bytes < shasta_nms_workspace_bytes(num_boxes)) {
        set_error_msg(axis_aligned ? "nms_normal: workspace too small" : "nms_rotated: workspace too small");
        return SHASTA_E_WORKSPACE;
    }
    const int words = cdiv(num_boxes, 64);
    unsigned long long* mask = static_cast<unsigned long long*>(workspace);
    if (axis_aligned) hipLaunchKernelGGL(nms_mask_kernel<true>, dim3(words, words), dim3(256), 0, st, boxes_sorted, num_boxes, thresh, words, mask);
    else hipLaunchKernelGGL(nms_mask_kernel<false>, dim3(words, words), dim3(256), 0, st, boxes_sorted, num_boxes, thresh, words, mask);
    int rc = check_launch("nms_mask");
    if (rc) return rc;
    const int slots = cdiv(words, 64);
    if (slots <= 1) hipLaunchKernelGGL(nms_reduce_kernel<1>, dim3(1), dim3(64), 0, st, mask, num_boxes, words, keep, num_keep);
    else if (slots <= 2) hipLaunchKernelGGL(nms_reduce_kernel<2>, dim3(1), dim3(64), 0, st, mask, num_boxes, words, keep, num_keep);
    else if (slots <= 4) hipLaunchKernelGGL(nms_reduce_kernel<4>, dim3(1), dim3(64), 0, st, mask, num_boxes, words, keep, num_keep);
    else hipLaunchKernelGGL(nms_reduce_kernel<8>, dim3(1), dim3(64), 0, st, mask, num_boxes, words, keep, num_keep);
    return check_launch("nms_reduce");
}

extern "C" int shasta_nms_rotated_f32(const float* boxes_sorted, int num_boxes, float thresh, void* workspace, size_t workspace_bytes,
                                      int32_t* keep, int32_t* num_keep, shasta_stream_t stream) {
    return nms_impl(false, boxes_sorted, num_boxes, thresh, workspace, workspace_bytes, keep, num_keep, stream);
}

extern "C" int shasta_nms_normal_f32(const float* boxes_sorted, int num_boxes, float thresh, void* workspace, size_t workspace_bytes,
                                     int32_t* keep, int32_t* num_keep, shasta_stream_t stream) {
    return nms_impl(true, boxes_sorted, num_boxes, thresh, workspace, workspace_bytes, keep, num_keep, stream);
}
