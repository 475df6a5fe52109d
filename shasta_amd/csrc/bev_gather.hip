// K2: BEV bilinear gather at box centre / edge mid-points.
// Restates (in one kernel) Shasta.get_box_center (det3d/models/tracker/shasta.py:121-161),
// center_to_corner_box2d (det3d/core/bbox/box_torch_ops.py:184-203), BEVFeatureExtractor.forward
// (det3d/models/second_stage/bird_eye_view.py:18-41) and bilinear_interpolate_torch
// (det3d/core/utils/center_utils.py:92-121).
//
// Layout: 16 lanes per (batch, object, point); lane c owns channels 4c..4c+3 (+64, ...), so each of the four corner
// fetches is one coalesced C*4-byte row of the NHWC map (256 B at C=64) read with 16-byte loads; 4 points per wave.  The
// output row [pt0 C | pt1 C | ...] is written directly in the packed (N, num_point*C) order the
// reference builds with a cat of sections (bird_eye_view.py:35-37).
// HBM-bound: 4*C*4 B read + C*4 B written per point.
#include "common.hpp"

namespace shasta {

// Arithmetic notes (parity): every product/sum below is a separately rounded fp32 op
// (__fmul_rn/__fadd_rn/__fsub_rn, never contracted) because the reference evaluates them as
// separate ATen ops; the metric->pixel map keeps the reference's two successive divisions
// (bird_eye_view.py:19-20) -- a fused reciprocal changes floor() for ~1e-6 of coordinates.
// ABSMAX: additionally the largest magnitude written per batch item (bit pattern of a non-negative float, atomicMax: order
// independent) into the ABSMAX_SLOTS x 128-byte lines absmax[b][slot][0] - the range exponent of the fp16 form of the aug_shape
// weight stream (anchor_split.hip), which otherwise costs a pass of its own over the tables (row_max_kernel: 0.12 ms at 512
// frame-pairs).  The lines must be zero on entry.
constexpr int ABSMAX_SLOTS = 8;
template <bool ABSMAX>
__global__ __launch_bounds__(256) void bev_gather_kernel(
    const float* __restrict__ bev, int H, int W, int C, const float* __restrict__ boxes, int N,
    int box_stride, int box_batch_stride, int num_point, float pc_x0, float pc_y0, float vs_x,
    float vs_y, float out_stride_px, float* __restrict__ out, int out_row_stride,
    int out_batch_stride, int total_points, unsigned* __restrict__ absmax, const float* __restrict__ bev2,
    const float* __restrict__ boxes2, float* __restrict__ out2, unsigned* __restrict__ absmax2) {
    // grid.y == 2: both frames of the pair in one launch (second map / box table / feature table / maxima)
    if (blockIdx.y == 1) {
        bev = bev2;
        boxes = boxes2;
        out = out2;
        absmax = absmax2;
    }
    // 16 lanes per point: each lane owns 4 consecutive channels (one 16-byte load per corner), 4 points per wave
    const int wave_raw = (blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    const int lane = threadIdx.x & 15;
    if (!ABSMAX && wave_raw >= total_points) return;
    const bool live = wave_raw < total_points;
    const int wave = live ? wave_raw : total_points - 1;  // ABSMAX: a spare group repeats the last point (no store) and joins the shuffles
    const int pt = wave % num_point;
    const int n = (wave / num_point) % N;
    const int b = wave / (num_point * N);

    const float* box = boxes + (size_t)b * box_batch_stride + (size_t)n * box_stride;
    const float cx = box[0], cy = box[1];
    float px = cx, py = cy;
    // point type: num_point==5 -> [centre, front, back, left, right]; 4 -> no centre; 1 -> centre
    const int edge = (num_point == 5) ? pt - 1 : (num_point == 4 ? pt : -1);
    if (edge >= 0) {
        const float w = box[3], l = box[4], yaw = box[6];
        const float s = sinf(yaw), c = cosf(yaw);
        // unit corners (clockwise from the minimum point): (-.5,-.5) (-.5,.5) (.5,.5) (.5,-.5)
        // edge mid-points: front=(c0+c1)/2, back=(c2+c3)/2, left=(c0+c3)/2, right=(c1+c2)/2
        const int ia = (edge == 0) ? 0 : (edge == 1) ? 2 : (edge == 2) ? 0 : 1;
        const int ib = (edge == 0) ? 1 : (edge == 1) ? 3 : (edge == 2) ? 3 : 2;
        float qx[2], qy[2];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int ci = k ? ib : ia;
            const float ux = (ci < 2) ? -0.5f : 0.5f;
            const float uy = (ci == 1 || ci == 2) ? 0.5f : -0.5f;
            const float dx = __fmul_rn(w, ux), dy = __fmul_rn(l, uy);
            // rotation_2d: x' = x*cos + y*sin ; y' = -x*sin + y*cos   (box_torch_ops.py:145-158)
            const float rx = __fadd_rn(__fmul_rn(dx, c), __fmul_rn(dy, s));
            const float ry = __fadd_rn(__fmul_rn(-dx, s), __fmul_rn(dy, c));
            qx[k] = __fadd_rn(rx, cx);
            qy[k] = __fadd_rn(ry, cy);
        }
        px = __fdiv_rn(__fadd_rn(qx[0], qx[1]), 2.0f);
        py = __fdiv_rn(__fadd_rn(qy[0], qy[1]), 2.0f);
    }
    const float x = __fdiv_rn(__fdiv_rn(__fsub_rn(px, pc_x0), vs_x), out_stride_px);
    const float y = __fdiv_rn(__fdiv_rn(__fsub_rn(py, pc_y0), vs_y), out_stride_px);
    // floor -> int with clamping done in float first so that wild coordinates (inf/NaN/1e30)
    // cannot overflow the conversion; the reference clamps the int64 indices.
    const float fx = floorf(x), fy = floorf(y);
    auto clampi = [](float f, int hi) -> int {
        if (!(f > -2.0f)) return 0 - 1;  // also NaN -> behaves like far-left (weights become NaN anyway)
        if (f > (float)(hi + 1)) return hi + 1;
        return (int)f;
    };
    int x0 = clampi(fx, W), y0 = clampi(fy, H);
    int x1 = x0 + 1, y1 = y0 + 1;
    x0 = min(max(x0, 0), W - 1);
    x1 = min(max(x1, 0), W - 1);
    y0 = min(max(y0, 0), H - 1);
    y1 = min(max(y1, 0), H - 1);
    const float fx0 = (float)x0, fx1 = (float)x1, fy0 = (float)y0, fy1 = (float)y1;
    const float wa = __fmul_rn(__fsub_rn(fx1, x), __fsub_rn(fy1, y));
    const float wb = __fmul_rn(__fsub_rn(fx1, x), __fsub_rn(y, fy0));
    const float wc = __fmul_rn(__fsub_rn(x, fx0), __fsub_rn(fy1, y));
    const float wd = __fmul_rn(__fsub_rn(x, fx0), __fsub_rn(y, fy0));

    const float* im = bev + (size_t)b * H * W * C;
    const float* Ia = im + ((size_t)y0 * W + x0) * C;
    const float* Ib = im + ((size_t)y1 * W + x0) * C;
    const float* Ic = im + ((size_t)y0 * W + x1) * C;
    const float* Id = im + ((size_t)y1 * W + x1) * C;
    float* o = out + (size_t)b * out_batch_stride + (size_t)n * out_row_stride + (size_t)pt * C;
    float amax = 0.0f;
    if ((C & 3) == 0) {
        // (not unrolled: C = 64 is one trip per lane; unrolled by the compiler's choice - which depends on the form of the maximum below -
        // the kernel took 178 registers instead of 44 and, at two waves per SIMD, 1.47 instead of 0.90 ms per 1024 frame-pairs)
#pragma unroll 1
        for (int c4 = lane; c4 < C / 4; c4 += 16) {
            const f32x4 a = reinterpret_cast<const f32x4*>(Ia)[c4], b4 = reinterpret_cast<const f32x4*>(Ib)[c4];
            const f32x4 c = reinterpret_cast<const f32x4*>(Ic)[c4], d = reinterpret_cast<const f32x4*>(Id)[c4];
            f32x4 r;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float v = __fmul_rn(a[q], wa);
                v = __fadd_rn(v, __fmul_rn(b4[q], wb));
                v = __fadd_rn(v, __fmul_rn(c[q], wc));
                v = __fadd_rn(v, __fmul_rn(d[q], wd));
                r[q] = v;
                if (ABSMAX) amax = __builtin_elementwise_maximum(amax, fabsf(v));  // v_maximum3_f32: NaN-propagating, |v| as a source modifier
            }
            if (live) reinterpret_cast<f32x4*>(o)[c4] = r;
        }
    } else {
#pragma unroll 1
        for (int ch = lane; ch < C; ch += 16) {
            float v = __fmul_rn(Ia[ch], wa);
            v = __fadd_rn(v, __fmul_rn(Ib[ch], wb));
            v = __fadd_rn(v, __fmul_rn(Ic[ch], wc));
            v = __fadd_rn(v, __fmul_rn(Id[ch], wd));
            if (ABSMAX) amax = __builtin_elementwise_maximum(amax, fabsf(v));
            if (live) o[ch] = v;
        }
    }
    if constexpr (ABSMAX) {
        // 16 lanes -> one value per point -> (normally) one per wave, posted into one of ABSMAX_SLOTS cache lines of the batch item
        // (consecutive waves take consecutive lines): every wave of a batch item hitting ONE address serialises on a single L2
        // channel - measured: one atomic per 16-point block 211 -> 267 us per call, a read-then-atomic per wave 2.0 ms.  No LDS, no
        // barrier.  The consumer reduces the lines (absmax_finalize_kernel).  A NaN / inf survives in the maximum as in the stand-alone
        // row_max / row_prep passes: the IEEE-754-2019 maximum in the loop (one instruction, |v| as a source modifier), the bit-pattern
        // maximum in the few shuffles and the atomic.
#pragma unroll
        for (int off = 8; off > 0; off >>= 1) amax = absmax_keep_nan(amax, __shfl_xor(amax, off, 16));
        if (!live) amax = 0.0f;
        const int slot = ((blockIdx.x * blockDim.x + threadIdx.x) >> 6) & (ABSMAX_SLOTS - 1);
        const int b_first = __shfl(b, 0, 64), b_last = __shfl(b, 48, 64);  // batch items are non-decreasing along the wave's four points
        if (b_first == b_last) {
            amax = absmax_keep_nan(amax, __shfl_xor(amax, 16, 64));
            amax = absmax_keep_nan(amax, __shfl_xor(amax, 32, 64));
            if ((threadIdx.x & 63) == 0) atomicMax(absmax + ((size_t)b * ABSMAX_SLOTS + slot) * 32, __float_as_uint(amax));
        } else if (lane == 0) {
            atomicMax(absmax + ((size_t)b * ABSMAX_SLOTS + slot) * 32, __float_as_uint(amax));
        }
    }
}

// out[i] = max over the ABSMAX_SLOTS lines of item i (i < n)
__global__ __launch_bounds__(256) void absmax_finalize_kernel(const unsigned* __restrict__ slots, unsigned* __restrict__ out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    unsigned m = 0;
#pragma unroll
    for (int s = 0; s < ABSMAX_SLOTS; ++s) m = max(m, slots[((size_t)i * ABSMAX_SLOTS + s) * 32]);
    out[i] = m;
}

size_t bev_absmax_slot_bytes(int items) { return (size_t)items * ABSMAX_SLOTS * 128; }
int launch_absmax_finalize(const unsigned* slots, unsigned* out, int items, hipStream_t st) {
    hipLaunchKernelGGL(absmax_finalize_kernel, dim3(cdiv(items, 256)), dim3(256), 0, st, slots, out, items);
    return check_launch("absmax_finalize");
}

// bev2 / boxes2 / out2 / absmax2: null, or the pair's other frame (same shapes and strides), gathered by the same launch
int launch_bev_gather(const float* bev, int B, int H, int W, int C, const float* boxes, int N, int box_stride, int box_batch_stride,
                      int num_point, float pc_x0, float pc_y0, float vs_x, float vs_y, float out_stride, float* out, int out_row_stride,
                      int out_batch_stride, unsigned* absmax, hipStream_t st, const float* bev2 = nullptr, const float* boxes2 = nullptr,
                      float* out2 = nullptr, unsigned* absmax2 = nullptr) {
    const long total = (long)B * N * num_point;
    if (total == 0) return SHASTA_OK;
    SHASTA_REQUIRE(total < (1L << 30), "bev_gather: too many points");
    const int points_per_block = 16;  // 256 threads, 16 lanes per point
    const dim3 grid(cdiv((int)total, points_per_block), bev2 ? 2 : 1);
    if (absmax)
        hipLaunchKernelGGL(bev_gather_kernel<true>, grid, dim3(16 * points_per_block), 0, st, bev, H, W, C, boxes, N, box_stride,
                           box_batch_stride, num_point, pc_x0, pc_y0, vs_x, vs_y, out_stride, out, out_row_stride, out_batch_stride, (int)total,
                           absmax, bev2, boxes2, out2, absmax2);
    else
        hipLaunchKernelGGL(bev_gather_kernel<false>, grid, dim3(16 * points_per_block), 0, st, bev, H, W, C, boxes, N, box_stride,
                           box_batch_stride, num_point, pc_x0, pc_y0, vs_x, vs_y, out_stride, out, out_row_stride, out_batch_stride, (int)total,
                           nullptr, bev2, boxes2, out2, nullptr);
    return check_launch("bev_gather");
}

}  // namespace shasta

extern "C" int shasta_bev_gather_f32(const float* bev, int B, int H, int W, int C, const float* boxes,
                                     int N, int box_stride, int box_batch_stride, int num_point,
                                     float pc_x0, float pc_y0, float vs_x, float vs_y, float out_stride,
                                     float* out, int out_row_stride, int out_batch_stride,
                                     shasta_stream_t stream) {
    using namespace shasta;
    SHASTA_REQUIRE(bev && boxes && out, "bev_gather: null pointer");
    SHASTA_REQUIRE(B >= 0 && N >= 0 && H > 0 && W > 0 && C > 0, "bev_gather: bad size");
    SHASTA_REQUIRE(num_point == 1 || num_point == 4 || num_point == 5, "bev_gather: num_point must be 1, 4 or 5");
    SHASTA_REQUIRE(box_stride >= (num_point == 1 ? 2 : 7) && out_row_stride >= num_point * C, "bev_gather: bad stride");
    return launch_bev_gather(bev, B, H, W, C, boxes, N, box_stride, box_batch_stride, num_point, pc_x0, pc_y0, vs_x, vs_y, out_stride, out,
                             out_row_stride, out_batch_stride, nullptr, as_stream(stream));
}
