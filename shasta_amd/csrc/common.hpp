// Shared helpers for the gfx950 kernels of the ShaSTA affinity path.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/shasta_hip.h"

namespace shasta {

constexpr int kWave = 64;  // CDNA wavefront

void set_error(const char* what, hipError_t e);
void set_error_msg(const char* what);

inline int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error(what, e);
        return SHASTA_E_LAUNCH;
    }
    return SHASTA_OK;
}

inline hipStream_t as_stream(shasta_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }
inline int cdiv(int a, int b) { return (a + b - 1) / b; }

#define SHASTA_REQUIRE(cond, msg)            \
    do {                                     \
        if (!(cond)) {                       \
            shasta::set_error_msg(msg);      \
            return SHASTA_E_ARG;             \
        }                                    \
    } while (0)

// wave-wide sum, result in every lane (fixed butterfly order -> deterministic)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// Range exponent of a row whose largest magnitude is mx (given as the bit pattern of the non-negative float, the form
// row_max_kernel accumulates with atomicMax): the e with mx * 2^e in (2^13, 2^14], so that the scaled row uses fp16's normal range
// with room below 65504 (two-piece fp16 form of the weight stream, anchor_split.hip; undone exactly in anchor_hidden_kernel).
__device__ __forceinline__ int range_exponent_bits(unsigned bits) {
    const float mx = __uint_as_float(bits);
    if (!(mx > 0.0f) || !(mx < INFINITY)) return 0;
    int ex;
    (void)frexpf(mx, &ex);  // mx = m 2^ex, m in [0.5, 1)
    return max(-60, min(60, 14 - ex));
}

// Largest of two magnitudes (both non-negative or NaN with the sign cleared, i.e. results of fabsf) by their bit patterns: a NaN
// (0x7fc00000 ...) beats an infinity (0x7f800000) beats every finite value, so - unlike fmaxf, which drops a NaN - a non-finite
// entry of a row survives in the row's maximum and the kernels that scale by it can see it (pair_f16.hip).
__device__ __forceinline__ float absmax_keep_nan(float a, float b) { return __uint_as_float(max(__float_as_uint(a), __float_as_uint(b))); }

// ReLU as torch computes it: a NaN stays a NaN (v_maximum3_f32, the IEEE-754-2019 maximum; fmaxf / v_max_f32 return the other
// operand, i.e. turn a NaN into 0 and every later layer into finite numbers)
__device__ __forceinline__ float relu_nan(float x) { return __builtin_elementwise_maximum(x, 0.0f); }

// Layout of the companion buffer of the four aug_shape first-layer matrices (shasta_aug_shape_aux_f32), H = rows per matrix:
//   [0, A)      4 H uint32: bit patterns of the row maxima                       A = align256(16 H)
//   [A, A + S)  4 H fp32: sum of |w| of every row, then 16 floats of summary (aux_ratio_kernel: [0] = largest max / mean|w| over all
//               rows, [1] = its row)                                             S = align256(16 H + 64)
//   [A + S, ..) the pre-cut fp16 piece image (only with SHASTA_OPT_PRECUT_WEIGHT_STREAM)
inline size_t aux_maxima_bytes(size_t H) { return align_up(4 * H * sizeof(unsigned), 256); }
inline size_t aux_stats_bytes(size_t H) { return align_up(4 * H * sizeof(float) + 64, 256); }
inline size_t aux_image_offset(size_t H) { return aux_maxima_bytes(H) + aux_stats_bytes(H); }

// XCD-aware tile order.  Workgroups go to the 8 XCDs (each with its own L2) round-robin by their linear id, x fastest: with the tiles
// of one frame along x, every XCD sees every frame and fetches that frame's shared rows into its own L2 (the pair kernel at 1024
// frame-pairs: 2.7 GB fetched for 0.6 GB of tables).  This returns the block index of the LOGICAL tile for the calling workgroup such
// that the workgroups of one XCD take consecutive logical tiles - the tiles of a frame share an L2.  (A pure relabelling: which
// workgroup computes a tile has no influence on the tile's result.)
__device__ __forceinline__ void xcd_logical_block(int& bx, int& by, int& bz) {
    const unsigned nx = gridDim.x, ny = gridDim.y, total = nx * ny * gridDim.z;
    unsigned L = blockIdx.x + nx * (blockIdx.y + ny * blockIdx.z);
    const unsigned per = total >> 3;
    if (L < per * 8) L = (L & 7) * per + (L >> 3);
    bx = (int)(L % nx);
    by = (int)((L / nx) % ny);
    bz = (int)(L / (nx * ny));
}

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

}  // namespace shasta
