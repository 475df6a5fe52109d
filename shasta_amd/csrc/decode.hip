// Next row f-4: the decode of the affinity matrices (tools/nusc_shasta/eval.py:127-173 == validate.py:68-114) as two
// batched kernels that return compact per-row / per-column decisions instead of one .item() per element on the host.
//   prev rows  n < n_prev : A = [m1[n, :n_cur] | m1[n, N], m1[n, N+1]] ; (val, k) = first maximum
//                           val > 0.5 and k == dead column -> DEAD ; val > 0.5 and k == FN column -> FN (score 1 - A[n,-2])
//   detections k < n_cur  : Bm = [m2[kept prev rows, k] ; m2[N, k] ; m2[N+1, k]] ; (val, r) = first maximum
//                           val > 0.7 and r == FP row -> dropped ; val > 0.5 and r == newborn row -> newborn flag ;
//                           score = 1 - m2[N+1, k]
// "First maximum" reproduces torch.max / argmax tie-breaking (lowest index).
#include "common.hpp"

namespace shasta {

__device__ __forceinline__ void argmax_merge(float& v, int& i, float ov, int oi) {
    if (ov > v || (ov == v && oi < i)) {
        v = ov;
        i = oi;
    }
}

// one wave per (b, n)
__global__ __launch_bounds__(256) void decode_rows_kernel(const float* __restrict__ m1, const int* __restrict__ n_prev,
                                                          const int* __restrict__ n_cur, int B, int N,
                                                          int* __restrict__ prev_class, float* __restrict__ prev_score) {
    const int lane = threadIdx.x & 63;
    const int item = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (item >= B * N) return;
    const int b = item / N, n = item - b * N;
    const int np = n_prev[b], nc = n_cur[b];
    if (n >= np) {
        if (lane == 0) {
            prev_class[item] = -1;
            prev_score[item] = 0.0f;
        }
        return;
    }
    const float* row = m1 + ((size_t)b * N + n) * (N + 2);
    float v = -INFINITY;
    int idx = 0x7fffffff;
    for (int k = lane; k < nc + 2; k += 64) {
        const float x = k < nc ? row[k] : row[N + (k - nc)];
        argmax_merge(v, idx, x, k);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const float ov = __shfl_xor(v, off, 64);
        const int oi = __shfl_xor(idx, off, 64);
        argmax_merge(v, idx, ov, oi);
    }
    if (lane == 0) {
        int cls = 0;
        if (v > 0.5f && idx == nc) cls = 1;           // dead track
        else if (v > 0.5f && idx == nc + 1) cls = 2;  // false negative
        prev_class[item] = cls;
        prev_score[item] = row[N];  // A[n, -2]; the host forms 1 - x in double like the reference
    }
}

// one wave per (b, k)
__global__ __launch_bounds__(256) void decode_cols_kernel(const float* __restrict__ m2, const int* __restrict__ n_prev,
                                                          const int* __restrict__ n_cur, const int* __restrict__ prev_class,
                                                          int B, int N, int* __restrict__ det_flags,
                                                          float* __restrict__ det_score) {
    const int lane = threadIdx.x & 63;
    const int item = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (item >= B * N) return;
    const int b = item / N, k = item - b * N;
    const int np = n_prev[b], nc = n_cur[b];
    if (k >= nc) {
        if (lane == 0) {
            det_flags[item] = -1;
            det_score[item] = 0.0f;
        }
        return;
    }
    const float* col = m2 + (size_t)b * (N + 2) * N + k;
    const int* pc = prev_class + (size_t)b * N;
    // row order of Bm: kept previous rows by increasing n, then newborn (N), then FP (N+1): the original row index is
    // monotone in that order, so "first maximum" = lowest original index among equal values
    float v = -INFINITY;
    int idx = 0x7fffffff;
    for (int r = lane; r < np; r += 64)
        if (pc[r] == 0) argmax_merge(v, idx, col[(size_t)r * N], r);
    if (lane == 0) argmax_merge(v, idx, col[(size_t)N * N], N);
    if (lane == 1) argmax_merge(v, idx, col[(size_t)(N + 1) * N], N + 1);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const float ov = __shfl_xor(v, off, 64);
        const int oi = __shfl_xor(idx, off, 64);
        argmax_merge(v, idx, ov, oi);
    }
    if (lane == 0) {
        int flags = 1;                                   // bit 0: kept
        if (v > 0.7f && idx == N + 1) flags = 0;         // false positive: dropped
        else if (v > 0.5f && idx == N) flags |= 2;       // bit 1: newborn
        det_flags[item] = flags;
        det_score[item] = col[(size_t)(N + 1) * N];  // Bm[-1, k]
    }
}

}  // namespace shasta

extern "C" int shasta_decode_flags_f32(const float* matched1, const float* matched2, const int32_t* n_prev, const int32_t* n_cur,
                                       int B, int max_obj, int32_t* prev_class, float* prev_score, int32_t* det_flags,
                                       float* det_score, shasta_stream_t stream) {
    using namespace shasta;
    SHASTA_REQUIRE(B >= 0 && max_obj >= 1, "decode_flags: bad size");
    if (B == 0) return SHASTA_OK;
    SHASTA_REQUIRE(matched1 && matched2 && n_prev && n_cur && prev_class && prev_score && det_flags && det_score,
                   "decode_flags: null pointer");
    const int grid = cdiv(B * max_obj, 4);
    hipLaunchKernelGGL(decode_rows_kernel, dim3(grid), dim3(256), 0, as_stream(stream), matched1, n_prev, n_cur, B, max_obj,
                       prev_class, prev_score);
    int rc = check_launch("decode_rows");
    if (rc) return rc;
    hipLaunchKernelGGL(decode_cols_kernel, dim3(grid), dim3(256), 0, as_stream(stream), matched2, n_prev, n_cur, prev_class, B,
                       max_obj, det_flags, det_score);
    return check_launch("decode_cols");
}
