// K6: aff row-MLP + the two softmaxes.  Restates det3d/models/tracker/shasta.py:94-109 and :323-325:
//   matched  = aff(residual)                     six nn.Linear over the D axis, applied to each of the T rows
//   matched1 = softmax(matched[:, :-2, :], dim=2)   (B, N, N+2): each previous detection over {dets, dead, FN}
//   matched2 = softmax(matched[:, :, :-2], dim=1)   (B, N+2, N): each detection over {prev dets, newborn, FP}
// The six layers are six launches of the generic matrix-core GEMM over the B*T rows (bias + ReLU fused).
#include "common.hpp"
#include "pair_layout.hpp"

namespace shasta {

int launch_gemm_nt(const float* A, int lda, const float* W, int ldw, const float* bias, float* C, int ldc, int M,
                   int N, int K, int act, hipStream_t st);

// one wave per (b, t < N): softmax over the D entries of the row
__global__ __launch_bounds__(256) void softmax_rows_kernel(const float* __restrict__ matched, float* __restrict__ m1,
                                                           int B, int N, int T, int D, int ld) {
    const int lane = threadIdx.x & 63;
    const int item = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (item >= B * N) return;
    const int b = item / N, t = item % N;
    const float* x = matched + ((size_t)b * T + t) * ld;
    float mx = -INFINITY;
    for (int d = lane; d < D; d += 64) mx = fmaxf(mx, x[d]);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
    float s = 0.0f;
    for (int d = lane; d < D; d += 64) s += expf(x[d] - mx);
    s = wave_sum(s);
    float* o = m1 + ((size_t)b * N + t) * D;
    for (int d = lane; d < D; d += 64) o[d] = expf(x[d] - mx) / s;
}

// block = 64 detections x 4 track groups: softmax over the T rows of each column d < N
__global__ __launch_bounds__(256) void softmax_cols_kernel(const float* __restrict__ matched, float* __restrict__ m2,
                                                           int N, int T, int ld) {
    __shared__ float red[4][64];
    const int b = blockIdx.y, dl = threadIdx.x & 63, tq = threadIdx.x >> 6;
    const int d = blockIdx.x * 64 + dl;
    const int dcl = min(d, N - 1);
    const float* x = matched + (size_t)b * T * ld + dcl;
    float mx = -INFINITY;
    for (int t = tq; t < T; t += 4) mx = fmaxf(mx, x[(size_t)t * ld]);
    red[tq][dl] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0][dl], red[1][dl]), fmaxf(red[2][dl], red[3][dl]));
    __syncthreads();
    float s = 0.0f;
    for (int t = tq; t < T; t += 4) s += expf(x[(size_t)t * ld] - mx);
    red[tq][dl] = s;
    __syncthreads();
    s = (red[0][dl] + red[1][dl]) + (red[2][dl] + red[3][dl]);
    if (d >= N) return;
    float* o = m2 + (size_t)b * T * N + d;
    for (int t = tq; t < T; t += 4) o[(size_t)t * N] = expf(x[(size_t)t * ld] - mx) / s;
}

size_t aff_workspace_bytes(int B, int N) {
    const int T = N + 2, Dp = (T + 3) / 4 * 4;
    return 2 * align_up((size_t)B * T * 128 * sizeof(float), 256) + align_up((size_t)B * T * Dp * sizeof(float), 256);
}

int aff_softmax(const shasta_weights* w, const float* packed, int B, const float* residual, int ld, float* m1,
                float* m2, float* matched_out, void* ws, size_t ws_bytes, hipStream_t st) {
    const int N = w->max_obj, T = N + 2, D = N + 2, Dp = (T + 3) / 4 * 4;
    const PackedLayout P(N, w->num_feats, w->feat_dim);
    if (ws_bytes < aff_workspace_bytes(B, N)) {
        set_error_msg("aff_softmax: workspace too small");
        return SHASTA_E_WORKSPACE;
    }
    if (B == 0) return SHASTA_OK;
    char* base = static_cast<char*>(ws);
    float* h0 = reinterpret_cast<float*>(base);
    base += align_up((size_t)B * T * 128 * sizeof(float), 256);
    float* h1 = reinterpret_cast<float*>(base);
    base += align_up((size_t)B * T * 128 * sizeof(float), 256);
    float* matched = reinterpret_cast<float*>(base);
    const int M = B * T;
    int rc;
    // aff.0 uses the zero-padded copy when the caller's residual rows are Dp-strided (16-byte aligned rows)
    if (ld % 4 == 0)
        rc = launch_gemm_nt(residual, ld, packed + P.aff0, Dp, w->aff[0].bias, h0, 128, M, 128, D, 1, st);
    else
        rc = launch_gemm_nt(residual, ld, w->aff[0].weight, D, w->aff[0].bias, h0, 128, M, 128, D, 1, st);
    if (rc) return rc;
    if ((rc = launch_gemm_nt(h0, 128, w->aff[1].weight, 128, w->aff[1].bias, h1, 128, M, 64, 128, 1, st))) return rc;
    if ((rc = launch_gemm_nt(h1, 128, w->aff[2].weight, 64, w->aff[2].bias, h0, 128, M, 32, 64, 1, st))) return rc;
    if ((rc = launch_gemm_nt(h0, 128, w->aff[3].weight, 32, w->aff[3].bias, h1, 128, M, 64, 32, 1, st))) return rc;
    if ((rc = launch_gemm_nt(h1, 128, w->aff[4].weight, 64, w->aff[4].bias, h0, 128, M, 128, 64, 1, st))) return rc;
    if ((rc = launch_gemm_nt(h0, 128, w->aff[5].weight, 128, w->aff[5].bias, matched, Dp, M, D, 128, 0, st))) return rc;
    hipLaunchKernelGGL(softmax_rows_kernel, dim3(cdiv(B * N, 4)), dim3(256), 0, st, matched, m1, B, N, T, D, Dp);
    if ((rc = check_launch("softmax_rows"))) return rc;
    hipLaunchKernelGGL(softmax_cols_kernel, dim3(cdiv(N, 64), B), dim3(256), 0, st, matched, m2, N, T, Dp);
    if ((rc = check_launch("softmax_cols"))) return rc;
    if (matched_out) {
        hipError_t e = hipMemcpy2DAsync(matched_out, (size_t)D * sizeof(float), matched, (size_t)Dp * sizeof(float),
                                        (size_t)D * sizeof(float), (size_t)M, hipMemcpyDeviceToDevice, st);
        if (e != hipSuccess) {
            set_error("aff_softmax: copy matched", e);
            return SHASTA_E_LAUNCH;
        }
    }
    return SHASTA_OK;
}

}  // namespace shasta
