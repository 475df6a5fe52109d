// Next row f-2 (second half): rotated BEV NMS.
// Replaces det3d/ops/iou3d_nms: nms_kernel (src/iou3d_nms_kernel.cu:267-311, pairwise rotated BEV IoU > thresh packed into
// 64-bit masks), the host-side mask reduction of nms_gpu (src/iou3d_nms.cpp:100-143) and iou_bev (:226-233,
// IoU = overlap / max(area_a + area_b - overlap, 1e-8)).  Boxes are [x, y, z, dx, dy, dz, heading] sorted by descending
// score (the caller sorts, like iou3d_nms_utils.py:74-89).  Two launches, nothing returns to the host in between:
//   1. mask[i][w] bit b = IoU_bev(box i, box 64w + b) > thresh for 64w + b > i   (one lane per pair, the wave ballot is the
//      mask word; boxes staged in LDS; blocks below the diagonal are skipped - the reduction never needs them)
//   2. one wavefront walks the boxes in score order: lane l owns words l, l+64, ... of the `removed` bitmap; a box that is
//      not removed is kept and ORs its mask row in (mask rows are fetched a chunk of boxes ahead).
// The overlap is the float64 Sutherland-Hodgman clip of the two rectangles (geom2d.hpp); the reference's float32 edge-
// intersection + angular-sort routine (iou3d_nms_kernel.cu:104-225) yields the same area up to its rounding, so decisions
// can differ only for IoUs within ~1e-6 of the threshold.
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

// block = (64-column block cb, 64-row block rb), 4 waves: a wave takes rows r = wave, wave + 4, ...; lane j evaluates the pair
// (row i, column 64 cb + j) and the wave's ballot IS the 64-bit mask word of that row (one thread per row with a 64-step
// loop left most of the chip idle: 136 waves for 1000 boxes)
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
        const bool hit = lane < ncol && j > i && iou_bev(row + r * 7, col + lane * 7) > (double)thresh;
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

}  // namespace shasta

using namespace shasta;

extern "C" size_t shasta_nms_workspace_bytes(int num_boxes) {
    const size_t words = (size_t)cdiv(std::max(num_boxes, 1), 64);
    return (size_t)std::max(num_boxes, 1) * words * sizeof(unsigned long long);
}

extern "C" int shasta_nms_rotated_f32(const float* boxes_sorted, int num_boxes, float thresh, void* workspace, size_t workspace_bytes,
                                      int32_t* keep, int32_t* num_keep, shasta_stream_t stream) {
    SHASTA_REQUIRE(keep && num_keep && num_boxes >= 0, "nms_rotated: bad argument");
    SHASTA_REQUIRE(num_boxes <= 64 * 64 * NMS_MAX_WORDS_PER_LANE, "nms_rotated: at most 32768 boxes");
    hipStream_t st = as_stream(stream);
    if (num_boxes == 0) {
        hipError_t e = hipMemsetAsync(num_keep, 0, sizeof(int32_t), st);
        if (e != hipSuccess) {
            set_error("nms_rotated: memset", e);
            return SHASTA_E_LAUNCH;
        }
        return SHASTA_OK;
    }
    SHASTA_REQUIRE(boxes_sorted && workspace, "nms_rotated: null pointer");
    if (workspace_bytes < shasta_nms_workspace_bytes(num_boxes)) {
        set_error_msg("nms_rotated: workspace too small");
        return SHASTA_E_WORKSPACE;
    }
    const int words = cdiv(num_boxes, 64);
    unsigned long long* mask = static_cast<unsigned long long*>(workspace);
    hipLaunchKernelGGL(nms_mask_kernel, dim3(words, words), dim3(256), 0, st, boxes_sorted, num_boxes, thresh, words, mask);
    int rc = check_launch("nms_mask");
    if (rc) return rc;
    const int slots = cdiv(words, 64);
    if (slots <= 1) hipLaunchKernelGGL(nms_reduce_kernel<1>, dim3(1), dim3(64), 0, st, mask, num_boxes, words, keep, num_keep);
    else if (slots <= 2) hipLaunchKernelGGL(nms_reduce_kernel<2>, dim3(1), dim3(64), 0, st, mask, num_boxes, words, keep, num_keep);
    else if (slots <= 4) hipLaunchKernelGGL(nms_reduce_kernel<4>, dim3(1), dim3(64), 0, st, mask, num_boxes, words, keep, num_keep);
    else hipLaunchKernelGGL(nms_reduce_kernel<8>, dim3(1), dim3(64), 0, st, mask, num_boxes, words, keep, num_keep);
    return check_launch("nms_reduce");
}
