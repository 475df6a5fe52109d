// Generic fp32 "NT" GEMM on the CDNA4 matrix cores:  C[m][n] = act(sum_k A[m][k] * W[n][k] + bias[n])
// Both operands are K-contiguous (activations row major, nn.Linear weight (out,in) row major), so this
// is the one dense kernel behind every nn.Linear of the path that is applied to many rows:
//   * the factorised first layers of fuse_shape / res_coeff (shasta.py:59-67,86-92) on the T+D table rows,
//   * the six aff layers (shasta.py:94-106) on the B*T residual rows.
// v_mfma_f32_32x32x2_f32: f32 in / f32 accumulate, bit-for-bit a k-ordered fmaf chain (exact fp32; gfx950 has
// no xf32), 64 FLOP/clk/SIMD.  Workgroup = 4 waves = 64x64 output tile, each wave one 32x32 accumulator
// (16 VGPRs); K is walked in 32-wide slices staged through LDS (row stride 33 floats: the one-float-per-lane
// fragment reads A[i=l&31][k=l>>5] then hit 32 distinct banks), next slice prefetched into registers while
// the current one is multiplied.
#include "common.hpp"

namespace shasta {

constexpr int BM = 64, BN = 64, BK = 32, LDS_LD = BK + 1;

template <bool VEC>
__device__ __forceinline__ void load_slice(const float* __restrict__ P, int ld, int rows, int K, int r0, int k0,
                                           int tid, float (&reg)[2][4]) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int idx = tid + 256 * i;
        const int row = idx >> 3, c4 = idx & 7;
        const int gr = r0 + row, gk = k0 + 4 * c4;
        if (VEC && gr < rows && gk + 3 < K) {
            const float4 v = *reinterpret_cast<const float4*>(P + (size_t)gr * ld + gk);
            reg[i][0] = v.x; reg[i][1] = v.y; reg[i][2] = v.z; reg[i][3] = v.w;
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                reg[i][j] = (gr < rows && gk + j < K) ? P[(size_t)gr * ld + gk + j] : 0.0f;
        }
    }
}

__device__ __forceinline__ void store_slice(float* S, int tid, const float (&reg)[2][4]) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int idx = tid + 256 * i;
        const int row = idx >> 3, c4 = idx & 7;
#pragma unroll
        for (int j = 0; j < 4; ++j) S[row * LDS_LD + 4 * c4 + j] = reg[i][j];
    }
}

template <bool VEC>
__global__ __launch_bounds__(256) void gemm_nt_f32_kernel(const float* __restrict__ A, int lda,
                                                          const float* __restrict__ W, int ldw,
                                                          const float* __restrict__ bias, float* __restrict__ C,
                                                          int ldc, int M, int N, int K, int act) {
    __shared__ float As[BM * LDS_LD];
    __shared__ float Ws[BN * LDS_LD];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 1, wn = wid & 1;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    float ra[2][4], rw[2][4];
    f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const int nk = (K + BK - 1) / BK;
    load_slice<VEC>(A, lda, M, K, m0, 0, tid, ra);
    load_slice<VEC>(W, ldw, N, K, n0, 0, tid, rw);
    const float* af = As + (wm * 32 + (lane & 31)) * LDS_LD + (lane >> 5);
    const float* wf = Ws + (wn * 32 + (lane & 31)) * LDS_LD + (lane >> 5);
    for (int kt = 0; kt < nk; ++kt) {
        __syncthreads();  // previous slice fully consumed
        store_slice(As, tid, ra);
        store_slice(Ws, tid, rw);
        __syncthreads();
        if (kt + 1 < nk) {
            load_slice<VEC>(A, lda, M, K, m0, (kt + 1) * BK, tid, ra);
            load_slice<VEC>(W, ldw, N, K, n0, (kt + 1) * BK, tid, rw);
        }
#pragma unroll
        for (int s = 0; s < BK / 2; ++s)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[2 * s], wf[2 * s], acc, 0, 0, 0);
    }
    // C/D map of the 32x32 accumulator: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
    const int col = n0 + wn * 32 + (lane & 31);
    if (col < N) {
        const float bv = bias ? bias[col] : 0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            if (row < M) {
                float v = acc[r] + bv;
                if (act == 1) v = fmaxf(v, 0.0f);
                else if (act == 2) v = fabsf(v);
                C[(size_t)row * ldc + col] = v;
            }
        }
    }
}

int launch_gemm_nt(const float* A, int lda, const float* W, int ldw, const float* bias, float* C, int ldc, int M,
                   int N, int K, int act, hipStream_t st) {
    if (M == 0 || N == 0) return SHASTA_OK;
    dim3 grid(cdiv(N, BN), cdiv(M, BM));
    const bool vec = (lda % 4 == 0) && (ldw % 4 == 0) && (((uintptr_t)A | (uintptr_t)W) % 16 == 0);
    if (vec)
        hipLaunchKernelGGL(gemm_nt_f32_kernel<true>, grid, dim3(256), 0, st, A, lda, W, ldw, bias, C, ldc, M, N, K, act);
    else
        hipLaunchKernelGGL(gemm_nt_f32_kernel<false>, grid, dim3(256), 0, st, A, lda, W, ldw, bias, C, ldc, M, N, K, act);
    return check_launch("gemm_nt_f32");
}

}  // namespace shasta

extern "C" int shasta_gemm_nt_f32(const float* A, int lda, const float* W, int ldw, const float* bias, float* C,
                                  int ldc, int M, int N, int K, int act, shasta_stream_t stream) {
    using namespace shasta;
    SHASTA_REQUIRE(A && W && C, "gemm_nt: null pointer");
    SHASTA_REQUIRE(M >= 0 && N >= 0 && K > 0 && lda >= K && ldw >= K && ldc >= N, "gemm_nt: bad size");
    SHASTA_REQUIRE(act >= 0 && act <= 2, "gemm_nt: bad activation");
    return launch_gemm_nt(A, lda, W, ldw, bias, C, ldc, M, N, K, act, as_stream(stream));
}
