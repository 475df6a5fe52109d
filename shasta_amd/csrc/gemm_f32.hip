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
__device__ __forceinline__ void gemm_nt_tile(float* As, float* Ws, const float* __restrict__ A, int lda, const float* __restrict__ W,
                                             int ldw, const float* __restrict__ bias, float* __restrict__ C, int ldc, int M, int N,
                                             int K, int act) {
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
                if (act == 1) v = relu_nan(v);
                else if (act == 2) v = fabsf(v);
                C[(size_t)row * ldc + col] = v;
            }
        }
    }
}

template <bool VEC>
__global__ __launch_bounds__(256) void gemm_nt_f32_kernel(const float* __restrict__ A, int lda,
                                                          const float* __restrict__ W, int ldw,
                                                          const float* __restrict__ bias, float* __restrict__ C,
                                                          int ldc, int M, int N, int K, int act) {
    __shared__ float As[BM * LDS_LD];
    __shared__ float Ws[BN * LDS_LD];
    gemm_nt_tile<VEC>(As, Ws, A, lda, W, ldw, bias, C, ldc, M, N, K, act);
}

// two independent problems of the same shape in one launch (blockIdx.z picks the problem): the two row-embedding GEMMs of
// the pair stage are small (16 workgroups each at one frame pair), one launch runs them side by side
struct GemmNtDual {
    const float* A[2];
    const float* W[2];
    const float* bias[2];
    float* C[2];
};

__global__ __launch_bounds__(256) void gemm_nt_f32_dual_kernel(GemmNtDual p, int lda, int ldw, int ldc, int M, int N, int K, int act) {
    __shared__ float As[BM * LDS_LD];
    __shared__ float Ws[BN * LDS_LD];
    const int z = blockIdx.z;
    gemm_nt_tile<true>(As, Ws, z ? p.A[1] : p.A[0], lda, z ? p.W[1] : p.W[0], ldw, z ? p.bias[1] : p.bias[0], z ? p.C[1] : p.C[0], ldc, M,
                       N, K, act);
}

// Few rows (the row embeddings of one or a few small frame-pairs: 12 workgroups of the kernel above walk ten dependent 32-wide slices,
// 13 us at max_obj 90, batch 1): the whole K in ONE round of loads.  A workgroup owns a 32 x 32 output tile; its four waves take a
// quarter of K each - every operand element of the wave straight from global memory into the MFMA layout, all loads in flight at
// once (lane (r, h) holds row r, elements [KQ/2 * h, KQ/2 * (h + 1)) of the wave's quarter for both operands, so MFMA step s pairs the
// same k on both sides) - and the four partial tiles are added in wave order through LDS.  K % 32 == 0 and K <= 512 (F = 64 .. 512).
template <int KQ>  // K / 4: elements per wave
__global__ __launch_bounds__(256) void gemm_nt_f32_small_dual_kernel(GemmNtDual p, int lda, int ldw, int ldc, int M, int N, int act) {
    __shared__ float red[4][32][33];
    constexpr int KH = KQ / 2;  // elements per lane and operand
    static_assert(KH % 4 == 0, "16-byte loads");
    const int z = blockIdx.z;
    const float* A = z ? p.A[1] : p.A[0];
    const float* W = z ? p.W[1] : p.W[0];
    const float* bias = z ? p.bias[1] : p.bias[0];
    float* C = z ? p.C[1] : p.C[0];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int m0 = blockIdx.y * 32, n0 = blockIdx.x * 32;
    const f32x4* ap = reinterpret_cast<const f32x4*>(A + (size_t)min(m0 + r, M - 1) * lda + wid * KQ + h * KH);
    const f32x4* wp = reinterpret_cast<const f32x4*>(W + (size_t)min(n0 + r, N - 1) * ldw + wid * KQ + h * KH);
    f32x4 av[KH / 4], wv[KH / 4];
#pragma unroll
    for (int i = 0; i < KH / 4; ++i) {
        av[i] = ap[i];
        wv[i] = wp[i];
    }
    f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < KH / 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i][j], wv[i][j], acc, 0, 0, 0);
    // accumulator map: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
#pragma unroll
    for (int q = 0; q < 16; ++q) red[wid][(q & 3) + 8 * (q >> 2) + 4 * h][r] = acc[q];
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int e = tid + 256 * i, row = e >> 5, col = e & 31;
        if (m0 + row < M && n0 + col < N) {
            float v = ((red[0][row][col] + red[1][row][col]) + red[2][row][col]) + red[3][row][col];
            if (bias) v += bias[n0 + col];
            if (act == 1) v = relu_nan(v);
            else if (act == 2) v = fabsf(v);
            C[(size_t)(m0 + row) * ldc + n0 + col] = v;
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

// up to four independent problems of one shape in one launch (blockIdx.z picks the problem): the second layers of the four
// aug_shape / first layers of the four aug_dets anchor MLPs at batch >= 16 (anchor.hip)
struct GemmNtQuad {
    const float* A[4];
    const float* W[4];
    const float* bias[4];
    float* C[4];
};

template <bool VEC>
__global__ __launch_bounds__(256) void gemm_nt_f32_quad_kernel(GemmNtQuad p, int lda, int ldw, int ldc, int M, int N, int K, int act) {
    __shared__ float As[BM * LDS_LD];
    __shared__ float Ws[BN * LDS_LD];
    const int z = blockIdx.z;
    gemm_nt_tile<VEC>(As, Ws, p.A[z], lda, p.W[z], ldw, p.bias[z], p.C[z], ldc, M, N, K, act);
}

// Skinny problems with a long K (the second aug_shape layers: M = batch, N = F, K = N F / 64; the first aug_dets layers: K = 7 N): the
// 64 x 64 tiles above are 16 .. 128 workgroups that walk K in 60 .. 110 dependent slices, 80 - 100 us whatever the batch.  Here a
// workgroup owns a 32 x 32 tile and walks K in chunks of 128: its four waves take 32 consecutive k of every chunk each, operands go
// straight from global memory (L2) into the MFMA layout - lane (r, h) holds row r, 16 consecutive k, for both operands -, the next
// chunk's loads are in flight under the 16 MFMAs of this one, no LDS and no barrier until the four partial tiles are added in wave
// order at the end.  Preconditions (launch_gemm_nt_quad checks them): K % 4 == 0, 16-byte aligned operand rows.  Measured at
// N = 500: 50 us against 94 us for the 64 x 64 tiles at 512 rows; at 64 rows 36 - 41 us, no better than the VALU kernels of the small
// batches (anchor.hip: 37 - 40 us), which therefore stay below 256 rows.
#ifndef SHASTA_GEMM_DIRECT
#define SHASTA_GEMM_DIRECT 1
#endif
__global__ __launch_bounds__(256) void gemm_nt_f32_direct_quad_kernel(GemmNtQuad p, int lda, int ldw, int ldc, int M, int N, int K, int act) {
    __shared__ float red[4][32][33];
    const int z = blockIdx.z;
    const float* A = p.A[z];
    const float* W = p.W[z];
    const float* bias = p.bias[z];
    float* C = p.C[z];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int m0 = blockIdx.y * 32, n0 = blockIdx.x * 32;
    const float* ap = A + (size_t)min(m0 + r, M - 1) * lda + wid * 32 + h * 16;
    const float* wp = W + (size_t)min(n0 + r, N - 1) * ldw + wid * 32 + h * 16;
    const int kl = wid * 32 + h * 16;  // this lane's first k inside a chunk
    const f32x4 zero = {0.0f, 0.0f, 0.0f, 0.0f};
    // (two chunks per wave in flight; four measured the same: 50.1 against 51.5 us at 512 rows, K = 2000 / 3500)
    constexpr int NB = 2;
    f32x4 av[NB][4], wv[NB][4];
    auto load = [&](int c, f32x4 (&a4)[4], f32x4 (&w4)[4]) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bool in = c * 128 + kl + 4 * i < K;
            a4[i] = in ? *reinterpret_cast<const f32x4*>(ap + (size_t)c * 128 + 4 * i) : zero;
            w4[i] = in ? *reinterpret_cast<const f32x4*>(wp + (size_t)c * 128 + 4 * i) : zero;
        }
    };
    const int nc = (K + 127) / 128;
    f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int b = 0; b < NB - 1; ++b)
        if (b < nc) load(b, av[b], wv[b]);
    for (int c = 0; c < nc; c += NB) {
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            if (c + b >= nc) break;
            if (c + b + NB - 1 < nc) load(c + b + NB - 1, av[(b + NB - 1) % NB], wv[(b + NB - 1) % NB]);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[b][i][j], wv[b][i][j], acc, 0, 0, 0);
        }
    }
#pragma unroll
    for (int q = 0; q < 16; ++q) red[wid][(q & 3) + 8 * (q >> 2) + 4 * h][r] = acc[q];
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int e = tid + 256 * i, row = e >> 5, col = e & 31;
        if (m0 + row < M && n0 + col < N) {
            float v = ((red[0][row][col] + red[1][row][col]) + red[2][row][col]) + red[3][row][col];
            if (bias) v += bias[n0 + col];
            if (act == 1) v = relu_nan(v);
            else if (act == 2) v = fabsf(v);
            C[(size_t)(m0 + row) * ldc + n0 + col] = v;
        }
    }
}

// true when launch_gemm_nt_quad will take the direct form for these operands
bool gemm_nt_quad_direct_ok(const float* const A[4], const float* const W[4], int lda, int ldw, int K) {
    bool ok = SHASTA_GEMM_DIRECT && lda % 4 == 0 && ldw % 4 == 0 && K % 4 == 0 && K >= 512;
    for (int i = 0; i < 4; ++i) ok = ok && (((uintptr_t)A[i] | (uintptr_t)W[i]) % 16 == 0);
    return ok;
}
int launch_gemm_nt_quad(const float* const A[4], const float* const W[4], const float* const bias[4], float* const C[4], int lda,
                        int ldw, int ldc, int M, int N, int K, int act, hipStream_t st) {
    if (M == 0 || N == 0) return SHASTA_OK;
    GemmNtQuad p;
    bool vec = (lda % 4 == 0) && (ldw % 4 == 0);
    for (int i = 0; i < 4; ++i) {
        p.A[i] = A[i];
        p.W[i] = W[i];
        p.bias[i] = bias[i];
        p.C[i] = C[i];
        vec = vec && (((uintptr_t)A[i] | (uintptr_t)W[i]) % 16 == 0);
    }
    if (SHASTA_GEMM_DIRECT && vec && K % 4 == 0 && K >= 512) {  // skinny, long K: the direct form
        hipLaunchKernelGGL(gemm_nt_f32_direct_quad_kernel, dim3(cdiv(N, 32), cdiv(M, 32), 4), dim3(256), 0, st, p, lda, ldw, ldc, M, N, K, act);
        return check_launch("gemm_nt_f32_direct_quad");
    }
    dim3 grid(cdiv(N, BN), cdiv(M, BM), 4);
    if (vec) hipLaunchKernelGGL(gemm_nt_f32_quad_kernel<true>, grid, dim3(256), 0, st, p, lda, ldw, ldc, M, N, K, act);
    else hipLaunchKernelGGL(gemm_nt_f32_quad_kernel<false>, grid, dim3(256), 0, st, p, lda, ldw, ldc, M, N, K, act);
    return check_launch("gemm_nt_f32_quad");
}

// C0 = A0 W0^T (+bias0), C1 = A1 W1^T (+bias1), same shapes and leading dimensions; falls back to two launches when the
// vector-load preconditions do not hold
int launch_gemm_nt_dual(const float* A0, const float* W0, const float* bias0, float* C0, const float* A1, const float* W1,
                        const float* bias1, float* C1, int lda, int ldw, int ldc, int M, int N, int K, int act, hipStream_t st) {
    if (M == 0 || N == 0) return SHASTA_OK;
    const bool vec = (lda % 4 == 0) && (ldw % 4 == 0) && (((uintptr_t)A0 | (uintptr_t)W0 | (uintptr_t)A1 | (uintptr_t)W1) % 16 == 0);
    if (!vec) {
        int rc = launch_gemm_nt(A0, lda, W0, ldw, bias0, C0, ldc, M, N, K, act, st);
        if (rc) return rc;
        return launch_gemm_nt(A1, lda, W1, ldw, bias1, C1, ldc, M, N, K, act, st);
    }
    GemmNtDual p{{A0, A1}, {W0, W1}, {bias0, bias1}, {C0, C1}};
    // while the 64 x 64 tiles leave most of the chip idle: the one-shot kernel (four times the workgroups, one load latency)
    if (2 * cdiv(N, BN) * cdiv(M, BM) <= 128 && (K == 64 || K == 256 || K == 320 || K == 512)) {
        const dim3 grid(cdiv(N, 32), cdiv(M, 32), 2);
        if (K == 64) hipLaunchKernelGGL(gemm_nt_f32_small_dual_kernel<16>, grid, dim3(256), 0, st, p, lda, ldw, ldc, M, N, act);
        else if (K == 256) hipLaunchKernelGGL(gemm_nt_f32_small_dual_kernel<64>, grid, dim3(256), 0, st, p, lda, ldw, ldc, M, N, act);
        else if (K == 320) hipLaunchKernelGGL(gemm_nt_f32_small_dual_kernel<80>, grid, dim3(256), 0, st, p, lda, ldw, ldc, M, N, act);
        else hipLaunchKernelGGL(gemm_nt_f32_small_dual_kernel<128>, grid, dim3(256), 0, st, p, lda, ldw, ldc, M, N, act);
        return check_launch("gemm_nt_f32_small_dual");
    }
    hipLaunchKernelGGL(gemm_nt_f32_dual_kernel, dim3(cdiv(N, BN), cdiv(M, BM), 2), dim3(256), 0, st, p, lda, ldw, ldc, M, N, K, act);
    return check_launch("gemm_nt_f32_dual");
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

namespace shasta {

// ------------------------------------------------------------------------------------------------------------------
// Strided form for the training path (backward of every nn.Linear):
//   C[m][n] (+)= sum_k A(m, k) * W(n, k),  A(m, k) = A[m*sa_m + k*sa_k], W(n, k) = W[n*sw_n + k*sw_k]
//   forward  Y  = X . W^T        : sa = (ldx, 1),  sw = (ldw, 1)           (the NT kernel above)
//   dX = dY . W                  : A = dY (ldy, 1), "W" = W^T  -> sw = (1, ldw)          reduction over out features
//   dW = dY^T . X                : A = dY^T -> sa = (1, ldy), "W" = X^T -> sw = (1, ldx)  reduction over rows
// Optional epilogue: bias, activation, and a ReLU mask (C *= mask > 0) for the backward through a ReLU.
// Split-K (grid.z slices of the reduction) writes partial tiles to `C + z*slice_stride`; the caller sums them in a fixed
// order (gemm_reduce_kernel) -> deterministic, no float atomics.
// ------------------------------------------------------------------------------------------------------------------
struct GemmS {
    const float* A;
    const float* W;
    const float* bias;
    const float* mask;  // (M, ldmask) or nullptr
    float* C;
    long sa_m, sa_k, sw_n, sw_k;
    int vec_a, vec_w;  // operand rows are k-contiguous and 16-byte aligned: vector loads
    int ldc, ldmask, M, N, K, act, kslice;  // kslice: reduction elements per grid.z slice (multiple of BK); act & 4: C += result
    long slice_stride;
};

__device__ __forceinline__ void load_slice_strided(const float* __restrict__ P, long s_r, long s_k, int rows, int K, int r0,
                                                   int k0, int tid, bool vec, float (&reg)[2][4]) {
    if (s_k == 1 || s_r != 1) {  // k-contiguous (or general): thread -> (row, 4 consecutive k)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int idx = tid + 256 * i;
            const int row = idx >> 3, c4 = idx & 7;
            const int gr = r0 + row, gk = k0 + 4 * c4;
            if (vec && gr < rows && gk + 3 < K) {  // k-contiguous, 16-byte aligned rows: one dwordx4 load
                const f32x4 v = *reinterpret_cast<const f32x4*>(P + (long)gr * s_r + gk);
#pragma unroll
                for (int j = 0; j < 4; ++j) reg[i][j] = v[j];
                continue;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) reg[i][j] = (gr < rows && gk + j < K) ? P[(long)gr * s_r + (long)(gk + j) * s_k] : 0.0f;
        }
    } else {  // row-contiguous: adjacent threads read adjacent rows of the same k (coalesced)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                // element e = (i*4 + j)*256 + tid of the 64 x 32 tile: row = e & 63, k = e >> 6
                const int e = (i * 4 + j) * 256 + tid;
                const int row = e & 63, kk = e >> 6;
                const int gr = r0 + row, gk = k0 + kk;
                reg[i][j] = (gr < rows && gk < K) ? P[(long)gr + (long)gk * s_k] : 0.0f;
            }
    }
}

__device__ __forceinline__ void store_slice_strided(float* S, long s_r, long s_k, int tid, const float (&reg)[2][4]) {
    if (s_k == 1 || s_r != 1) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int idx = tid + 256 * i;
            const int row = idx >> 3, c4 = idx & 7;
#pragma unroll
            for (int j = 0; j < 4; ++j) S[row * LDS_LD + 4 * c4 + j] = reg[i][j];
        }
    } else {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int e = (i * 4 + j) * 256 + tid;
                S[(e & 63) * LDS_LD + (e >> 6)] = reg[i][j];
            }
    }
}

// shared epilogue of the strided kernels: bias, activation, ReLU mask, accumulate
__device__ __forceinline__ void strided_epilogue(const GemmS& g, const f32x16& acc, int m0, int n0, int wm, int wn, int lane) {
    float* C = g.C + (long)blockIdx.z * g.slice_stride;
    const int col = n0 + wn * 32 + (lane & 31);
    if (col < g.N) {
        const float bv = (g.bias && blockIdx.z == 0) ? g.bias[col] : 0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            if (row < g.M) {
                float v = acc[r] + bv;
                if ((g.act & 3) == 1) v = relu_nan(v);
                else if ((g.act & 3) == 2) v = fabsf(v);
                if (g.mask && !(g.mask[(long)row * g.ldmask + col] > 0.0f)) v = 0.0f;
                if (g.act & 4) v += C[(long)row * g.ldc + col];
                C[(long)row * g.ldc + col] = v;
            }
        }
    }
}

// A GROUP of up to 8 products of one shape (the four aug_shape / aug_dets MLPs of the anchor backward, the two sides of a pair MLP's
// first layer): operands, bias, mask and result per member, everything else shared; grid.y = members x row tiles.  Each member is
// computed exactly as a launch of its own would (same tiles, same reduction slices): the group saves launches, not arithmetic.
constexpr int GEMM_GROUP_MAX = 8;
struct GemmGroup {
    const float* A[GEMM_GROUP_MAX];
    const float* W[GEMM_GROUP_MAX];
    const float* bias[GEMM_GROUP_MAX];
    const float* mask[GEMM_GROUP_MAX];
    float* C[GEMM_GROUP_MAX];
    int mtiles;  // row tiles per member
};
__device__ __forceinline__ GemmS group_member(const GemmS& g0, const GemmGroup& gg, int& by) {
    GemmS g = g0;
    const int m = blockIdx.y / gg.mtiles;
    by = blockIdx.y - m * gg.mtiles;
    g.A = gg.A[m];
    g.W = gg.W[m];
    g.bias = gg.bias[m];
    g.mask = gg.mask[m];
    g.C = gg.C[m];
    return g;
}

__device__ __forceinline__ void gemm_strided_f32_body(const GemmS& g, int by) {
    __shared__ float As[BM * LDS_LD];
    __shared__ float Ws[BN * LDS_LD];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 1, wn = wid & 1;
    const int m0 = by * BM, n0 = blockIdx.x * BN;
    const int kbeg = blockIdx.z * g.kslice, kend = min(g.K, kbeg + g.kslice);
    float ra[2][4], rw[2][4];
    f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const int nk = (kend - kbeg + BK - 1) / BK;
    if (nk > 0) {
        load_slice_strided(g.A, g.sa_m, g.sa_k, g.M, kend, m0, kbeg, tid, g.vec_a != 0, ra);
        load_slice_strided(g.W, g.sw_n, g.sw_k, g.N, kend, n0, kbeg, tid, g.vec_w != 0, rw);
    }
    const float* af = As + (wm * 32 + (lane & 31)) * LDS_LD + (lane >> 5);
    const float* wf = Ws + (wn * 32 + (lane & 31)) * LDS_LD + (lane >> 5);
    for (int kt = 0; kt < nk; ++kt) {
        __syncthreads();
        store_slice_strided(As, g.sa_m, g.sa_k, tid, ra);
        store_slice_strided(Ws, g.sw_n, g.sw_k, tid, rw);
        __syncthreads();
        if (kt + 1 < nk) {
            load_slice_strided(g.A, g.sa_m, g.sa_k, g.M, kend, m0, kbeg + (kt + 1) * BK, tid, g.vec_a != 0, ra);
            load_slice_strided(g.W, g.sw_n, g.sw_k, g.N, kend, n0, kbeg + (kt + 1) * BK, tid, g.vec_w != 0, rw);
        }
#pragma unroll
        for (int s = 0; s < BK / 2; ++s)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[2 * s], wf[2 * s], acc, 0, 0, 0);
    }
    strided_epilogue(g, acc, m0, n0, wm, wn, lane);
}
__global__ __launch_bounds__(256) void gemm_strided_f32_kernel(GemmS g) { gemm_strided_f32_body(g, blockIdx.y); }
__global__ __launch_bounds__(256) void gemm_strided_group_f32_kernel(GemmS g0, GemmGroup gg) {
    int by;
    const GemmS g = group_member(g0, gg, by);
    gemm_strided_f32_body(g, by);
}

// bf16 operands (act & 8; BASELINE config 5's reduced-precision option for the training GEMMs): the fp32 operands in HBM are
// rounded to bf16 (nearest even) when a K slice is stored to LDS and multiplied on v_mfma_f32_32x32x16_bf16 with fp32
// accumulation - master weights, activations in HBM, bias / activation epilogue and the split-K sums all stay fp32.
// LDS rows are 32 + 8 bf16 = 80 bytes: the 16 lanes of a ds_read_b128 phase start 20 banks apart and cover all 64 banks.
constexpr int LDH = BK + 8;
typedef __bf16 gbf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t gu32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint16_t to_bf16_bits(float v) { return __builtin_bit_cast(uint16_t, (__bf16)v); }

__device__ __forceinline__ void store_slice_strided_bf16(uint16_t* S, long s_r, long s_k, int tid, const float (&reg)[2][4]) {
    if (s_k == 1 || s_r != 1) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int idx = tid + 256 * i;
            const int row = idx >> 3, c4 = idx & 7;
            uint32_t* d = reinterpret_cast<uint32_t*>(S + row * LDH + 4 * c4);
            d[0] = (uint32_t)to_bf16_bits(reg[i][0]) | ((uint32_t)to_bf16_bits(reg[i][1]) << 16);
            d[1] = (uint32_t)to_bf16_bits(reg[i][2]) | ((uint32_t)to_bf16_bits(reg[i][3]) << 16);
        }
    } else {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int e = (i * 4 + j) * 256 + tid;
                S[(e & 63) * LDH + (e >> 6)] = to_bf16_bits(reg[i][j]);
            }
    }
}

__device__ __forceinline__ void gemm_strided_bf16_body(const GemmS& g, int by) {
    __shared__ __attribute__((aligned(16))) uint16_t As[BM * LDH];
    __shared__ __attribute__((aligned(16))) uint16_t Ws[BN * LDH];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 1, wn = wid & 1;
    const int m0 = by * BM, n0 = blockIdx.x * BN;
    const int kbeg = blockIdx.z * g.kslice, kend = min(g.K, kbeg + g.kslice);
    float ra[2][4], rw[2][4];
    f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const int nk = (kend - kbeg + BK - 1) / BK;
    if (nk > 0) {
        load_slice_strided(g.A, g.sa_m, g.sa_k, g.M, kend, m0, kbeg, tid, g.vec_a != 0, ra);
        load_slice_strided(g.W, g.sw_n, g.sw_k, g.N, kend, n0, kbeg, tid, g.vec_w != 0, rw);
    }
    const uint16_t* af = As + (wm * 32 + (lane & 31)) * LDH + (lane >> 5) * 8;
    const uint16_t* wf = Ws + (wn * 32 + (lane & 31)) * LDH + (lane >> 5) * 8;
    for (int kt = 0; kt < nk; ++kt) {
        __syncthreads();
        store_slice_strided_bf16(As, g.sa_m, g.sa_k, tid, ra);
        store_slice_strided_bf16(Ws, g.sw_n, g.sw_k, tid, rw);
        __syncthreads();
        if (kt + 1 < nk) {
            load_slice_strided(g.A, g.sa_m, g.sa_k, g.M, kend, m0, kbeg + (kt + 1) * BK, tid, g.vec_a != 0, ra);
            load_slice_strided(g.W, g.sw_n, g.sw_k, g.N, kend, n0, kbeg + (kt + 1) * BK, tid, g.vec_w != 0, rw);
        }
#pragma unroll
        for (int s = 0; s < BK / 16; ++s) {
            const gu32x4 a = *reinterpret_cast<const gu32x4*>(af + 16 * s), w = *reinterpret_cast<const gu32x4*>(wf + 16 * s);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(gbf16x8, a), __builtin_bit_cast(gbf16x8, w), acc, 0, 0, 0);
        }
    }
    strided_epilogue(g, acc, m0, n0, wm, wn, lane);
}
__global__ __launch_bounds__(256) void gemm_strided_bf16_kernel(GemmS g) { gemm_strided_bf16_body(g, blockIdx.y); }
__global__ __launch_bounds__(256) void gemm_strided_group_bf16_kernel(GemmS g0, GemmGroup gg) {
    int by;
    const GemmS g = group_member(g0, gg, by);
    gemm_strided_bf16_body(g, by);
}

// C[row][col] = sum_z part[z*stride + i], i = row*N + col: 16 elements x 16 z-groups per block, each group sums its
// slices in order and the groups are combined in order (fixed association -> deterministic)
__device__ __forceinline__ void gemm_reduce_body(const float* __restrict__ part, long stride, int nz, float* __restrict__ C, long n, int N,
                                                 int ldc, const float* __restrict__ bias, int act) {
    __shared__ float red[16][17];
    const int e = threadIdx.x & 15, q = threadIdx.x >> 4;
    const long i = (long)blockIdx.x * 16 + e;
    float s = 0.0f;
    if (i < n)
        for (int z = q; z < nz; z += 16) s += part[(long)z * stride + i];
    red[q][e] = s;
    __syncthreads();
    if (q == 0 && i < n) {
        float t = red[0][e];
#pragma unroll
        for (int k = 1; k < 16; ++k) t += red[k][e];
        if (bias) t += bias[i % N];
        if ((act & 3) == 1) t = relu_nan(t);
        else if ((act & 3) == 2) t = fabsf(t);
        C[(i / N) * ldc + (i % N)] = t;
    }
}
__global__ __launch_bounds__(256) void gemm_reduce_kernel(const float* __restrict__ part, long stride, int nz, float* __restrict__ C,
                                                          long n, int N, int ldc, const float* __restrict__ bias, int act) {
    gemm_reduce_body(part, stride, nz, C, n, N, ldc, bias, act);
}
// grid.y = member: its slices lie nz * stride apart in `part`
__global__ __launch_bounds__(256) void gemm_reduce_group_kernel(const float* __restrict__ part, long stride, int nz, GemmGroup gg, long n, int N,
                                                                int ldc, int act) {
    gemm_reduce_body(part + (size_t)blockIdx.y * nz * stride, stride, nz, gg.C[blockIdx.y], n, N, ldc, gg.bias[blockIdx.y], act);
}

int launch_gemm_strided(const float* A, long sa_m, long sa_k, const float* W, long sw_n, long sw_k, const float* bias,
                        const float* mask, int ldmask, float* C, int ldc, int M, int N, int K, int act, float* splitk_ws,
                        size_t splitk_ws_bytes, hipStream_t st) {
    if (M == 0 || N == 0) return SHASTA_OK;
    GemmS g;
    g.A = A; g.W = W; g.bias = bias; g.mask = mask; g.C = C;
    g.sa_m = sa_m; g.sa_k = sa_k; g.sw_n = sw_n; g.sw_k = sw_k;
    g.ldc = ldc; g.ldmask = ldmask; g.M = M; g.N = N; g.K = K; g.act = act;
    g.vec_a = (sa_k == 1 && sa_m % 4 == 0 && ((uintptr_t)A & 15) == 0) ? 1 : 0;
    g.vec_w = (sw_k == 1 && sw_n % 4 == 0 && ((uintptr_t)W & 15) == 0) ? 1 : 0;
    const int tiles = cdiv(M, BM) * cdiv(N, BN);
    int nz = 1;
    // long reductions with few output tiles (weight gradients): split the reduction over grid.z
    // (bias and activation then belong to the sum of the slices: gemm_reduce_kernel applies them)
    // (from K = 256 in slices of at least 128: a few-row GEMM with K = 450 ... 630 in ONE workgroup took 30 - 36 us of the car
    // configuration's training step, four times per step)
    if (splitk_ws && K >= 256 && tiles < 128 && !mask && (act & 4) == 0) {
        nz = min(min(512, cdiv(1024, tiles)), cdiv(K, 128));
        while (nz > 1 && (size_t)nz * M * N * sizeof(float) > splitk_ws_bytes) --nz;
    }
    if (nz <= 1) {
        g.kslice = cdiv(max(K, 1), BK) * BK;
        g.slice_stride = 0;
        if (act & 8) hipLaunchKernelGGL(gemm_strided_bf16_kernel, dim3(cdiv(N, BN), cdiv(M, BM), 1), dim3(256), 0, st, g);
        else hipLaunchKernelGGL(gemm_strided_f32_kernel, dim3(cdiv(N, BN), cdiv(M, BM), 1), dim3(256), 0, st, g);
        return check_launch("gemm_strided");
    }
    g.kslice = cdiv(cdiv(K, nz), BK) * BK;
    nz = cdiv(K, g.kslice);
    g.C = splitk_ws;
    g.bias = nullptr;
    g.act = act & 8;
    g.ldc = N;
    g.slice_stride = (long)M * N;
    if (act & 8) hipLaunchKernelGGL(gemm_strided_bf16_kernel, dim3(cdiv(N, BN), cdiv(M, BM), nz), dim3(256), 0, st, g);
    else hipLaunchKernelGGL(gemm_strided_f32_kernel, dim3(cdiv(N, BN), cdiv(M, BM), nz), dim3(256), 0, st, g);
    int rc = check_launch("gemm_strided(split-K)");
    if (rc) return rc;
    const long n = (long)M * N;
    hipLaunchKernelGGL(gemm_reduce_kernel, dim3((unsigned)((n + 15) / 16)), dim3(256), 0, st, splitk_ws, g.slice_stride, nz, C, n, N, ldc, bias, act);
    return check_launch("gemm_reduce");
}

// `count` products of one shape in one launch (two with a split reduction); every member as launch_gemm_strided would compute it alone
// when given count-th of the scratch
int launch_gemm_strided_group(int count, const float* const* A, const float* const* W, const float* const* bias, const float* const* mask,
                              float* const* C, long sa_m, long sa_k, long sw_n, long sw_k, int ldmask, int ldc, int M, int N, int K, int act,
                              float* splitk_ws, size_t splitk_ws_bytes, hipStream_t st) {
    if (M == 0 || N == 0 || count == 0) return SHASTA_OK;
    GemmS g;
    GemmGroup gg;
    bool va = sa_k == 1 && sa_m % 4 == 0, vw = sw_k == 1 && sw_n % 4 == 0, any_mask = false;
    for (int i = 0; i < GEMM_GROUP_MAX; ++i) {
        const int j = i < count ? i : 0;
        gg.A[i] = A[j];
        gg.W[i] = W[j];
        gg.bias[i] = bias ? bias[j] : nullptr;
        gg.mask[i] = mask ? mask[j] : nullptr;
        gg.C[i] = C[j];
        va = va && ((uintptr_t)A[j] & 15) == 0;
        vw = vw && ((uintptr_t)W[j] & 15) == 0;
        any_mask = any_mask || gg.mask[i] != nullptr;
    }
    g.A = gg.A[0]; g.W = gg.W[0]; g.bias = gg.bias[0]; g.mask = gg.mask[0]; g.C = gg.C[0];
    g.sa_m = sa_m; g.sa_k = sa_k; g.sw_n = sw_n; g.sw_k = sw_k;
    g.ldc = ldc; g.ldmask = ldmask; g.M = M; g.N = N; g.K = K; g.act = act;
    g.vec_a = va ? 1 : 0;
    g.vec_w = vw ? 1 : 0;
    gg.mtiles = cdiv(M, BM);
    const int tiles = cdiv(M, BM) * cdiv(N, BN);
    int nz = 1;
    if (splitk_ws && K >= 256 && tiles < 128 && !any_mask && (act & 4) == 0) {
        nz = min(min(512, cdiv(1024, tiles)), cdiv(K, 128));
        while (nz > 1 && (size_t)nz * M * N * sizeof(float) > splitk_ws_bytes / count) --nz;
    }
    const dim3 grid(cdiv(N, BN), gg.mtiles * count, 1);
    if (nz <= 1) {
        g.kslice = cdiv(max(K, 1), BK) * BK;
        g.slice_stride = 0;
        if (act & 8) hipLaunchKernelGGL(gemm_strided_group_bf16_kernel, grid, dim3(256), 0, st, g, gg);
        else hipLaunchKernelGGL(gemm_strided_group_f32_kernel, grid, dim3(256), 0, st, g, gg);
        return check_launch("gemm_strided_group");
    }
    g.kslice = cdiv(cdiv(K, nz), BK) * BK;
    nz = cdiv(K, g.kslice);
    GemmGroup gs = gg;  // the slices of member i: splitk_ws + i * nz * M * N
    for (int i = 0; i < GEMM_GROUP_MAX; ++i) {
        gs.C[i] = splitk_ws + (size_t)(i < count ? i : 0) * nz * M * N;
        gs.bias[i] = nullptr;
    }
    g.act = act & 8;
    g.ldc = N;
    g.slice_stride = (long)M * N;
    const dim3 gridz(cdiv(N, BN), gg.mtiles * count, nz);
    if (act & 8) hipLaunchKernelGGL(gemm_strided_group_bf16_kernel, gridz, dim3(256), 0, st, g, gs);
    else hipLaunchKernelGGL(gemm_strided_group_f32_kernel, gridz, dim3(256), 0, st, g, gs);
    int rc = check_launch("gemm_strided_group(split-K)");
    if (rc) return rc;
    const long n = (long)M * N;
    hipLaunchKernelGGL(gemm_reduce_group_kernel, dim3((unsigned)((n + 15) / 16), count), dim3(256), 0, st, splitk_ws, g.slice_stride, nz, gg, n, N, ldc, act);
    return check_launch("gemm_reduce_group");
}

}  // namespace shasta

extern "C" int shasta_gemm_strided_group_f32(int count, const float* const* A, const float* const* W, const float* const* bias,
                                             const float* const* relu_mask, float* const* C, long sa_m, long sa_k, long sw_n, long sw_k,
                                             int ldmask, int ldc, int M, int N, int K, int act, void* splitk_ws, size_t splitk_ws_bytes,
                                             shasta_stream_t stream) {
    using namespace shasta;
    SHASTA_REQUIRE(count >= 0 && count <= GEMM_GROUP_MAX, "gemm_strided_group: 0 to 8 members");
    SHASTA_REQUIRE(count == 0 || (A && W && C), "gemm_strided_group: null pointer");
    for (int i = 0; i < count; ++i) SHASTA_REQUIRE(A[i] && W[i] && C[i], "gemm_strided_group: null member pointer");
    SHASTA_REQUIRE(M >= 0 && N >= 0 && K >= 0 && ldc >= N, "gemm_strided_group: bad size");
    SHASTA_REQUIRE(act >= 0 && act <= 14 && (act & 3) != 3,
                   "gemm_strided_group: bad activation (0 none, 1 relu, 2 abs, +4 accumulate into C, +8 bf16 operands)");
    return launch_gemm_strided_group(count, A, W, bias, relu_mask, C, sa_m, sa_k, sw_n, sw_k, ldmask, ldc, M, N, K, act,
                                     static_cast<float*>(splitk_ws), splitk_ws_bytes, as_stream(stream));
}

extern "C" int shasta_gemm_strided_f32(const float* A, long sa_m, long sa_k, const float* W, long sw_n, long sw_k, const float* bias,
                                       const float* relu_mask, int ldmask, float* C, int ldc, int M, int N, int K, int act,
                                       void* splitk_ws, size_t splitk_ws_bytes, shasta_stream_t stream) {
    using namespace shasta;
    SHASTA_REQUIRE(A && W && C, "gemm_strided: null pointer");
    SHASTA_REQUIRE(M >= 0 && N >= 0 && K >= 0 && ldc >= N, "gemm_strided: bad size");
    SHASTA_REQUIRE(act >= 0 && act <= 14 && (act & 3) != 3,
                   "gemm_strided: bad activation (0 none, 1 relu, 2 abs, +4 accumulate into C, +8 bf16 operands)");
    return launch_gemm_strided(A, sa_m, sa_k, W, sw_n, sw_k, bias, relu_mask, ldmask, C, ldc, M, N, K, act,
                               static_cast<float*>(splitk_ws), splitk_ws_bytes, as_stream(stream));
}
