// fp32 "NT" GEMM on the bf16 matrix path:  C[m][n] = act(sum_k A[m][k] * W[n][k] + bias[n]),  A, W, C fp32.
// Same contract as gemm_nt_f32_kernel (gemm_f32.hip) for the nn.Linear layers that are applied to many table rows
// (det3d/models/tracker/shasta.py:59-67,86-92,94-106), but every fp32 product is formed from exact bf16 pieces like in
// anchor_split.hip: a = a_hi + a_mid + a_lo (8 significand bits each, exact), w * a = the six piece products of weight
// 2^0 .. 2^-16 accumulated in the fp32 accumulator of v_mfma_f32_32x32x16_bf16 (the three dropped products are below
// 2^-24 |w a|, i.e. below the rounding of the fp32 FMA they replace).  6 bf16 MFMAs of K=16 replace 8 f32 MFMAs of K=2:
// 2.7 x fewer matrix cycles per fp32 product.
// Workgroup = 4 waves = 128 x 128 output tile (each wave 64 x 64 = 2 x 2 accumulators of 32 x 32); K is walked in 32-wide
// slices: fp32 slices are prefetched into registers, cut on the VALU when they are stored to LDS (three bf16 images per
// operand, row stride 80 B so that the 16 lanes of a ds_read_b128 phase cover all 64 banks), and read back as MFMA
// fragments (8 consecutive k per lane).  Two workgroups per CU (60 KB of LDS each): one cuts while the other multiplies.
// (A one-workgroup-per-CU form with double-buffered LDS, one barrier per slice and the cut interleaved behind the MFMA chains
// of the same wave measured 61 us instead of 39.5 us on 64 256 x 128 x 256: the second workgroup hides more than the pipeline.)
#include "common.hpp"

namespace shasta {

typedef __bf16 pbf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t pu32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t pu32x2 __attribute__((ext_vector_type(2)));

constexpr int PM = 128, PN = 128, PK = 32;
constexpr int PROW = 80;                   // bytes per LDS row of one piece image: 32 bf16 + 16 B pad
constexpr int PIMG = PM * PROW;            // one piece image of one operand
static_assert(PM == PN, "both operands use the same staging code");

__device__ __forceinline__ void cut3(float a, float& h, float& m, float& l) {
    h = __uint_as_float(__float_as_uint(a) & 0xffff0000u);
    const float r = a - h;
    m = __uint_as_float(__float_as_uint(r) & 0xffff0000u);
    l = r - m;
}
__device__ __forceinline__ uint32_t top2(float even, float odd) {
    return __builtin_amdgcn_perm(__float_as_uint(odd), __float_as_uint(even), 0x07060302u);
}

// 128 rows x 32 k of fp32 -> registers (4 float4 per thread); rows >= `rows` and k >= K read as zero
template <bool VEC>
__device__ __forceinline__ void pieces_load(const float* __restrict__ P, int ld, int rows, int K, int r0, int k0, int tid,
                                            f32x4 (&reg)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int idx = tid + 256 * i;
        const int gr = r0 + (idx >> 3), gk = k0 + 4 * (idx & 7);
        if (VEC && gr < rows && gk + 3 < K) {
            reg[i] = *reinterpret_cast<const f32x4*>(P + (size_t)gr * ld + gk);
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) reg[i][j] = (gr < rows && gk + j < K) ? P[(size_t)gr * ld + gk + j] : 0.0f;
        }
    }
}

// registers -> three bf16 images in LDS
__device__ __forceinline__ void pieces_store(char* img, int tid, const f32x4 (&reg)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int idx = tid + 256 * i;
        char* dst = img + (idx >> 3) * PROW + (idx & 7) * 8;
        float h[4], m[4], l[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) cut3(reg[i][j], h[j], m[j], l[j]);
        *reinterpret_cast<pu32x2*>(dst) = pu32x2{top2(h[0], h[1]), top2(h[2], h[3])};
        *reinterpret_cast<pu32x2*>(dst + PIMG) = pu32x2{top2(m[0], m[1]), top2(m[2], m[3])};
        *reinterpret_cast<pu32x2*>(dst + 2 * PIMG) = pu32x2{top2(l[0], l[1]), top2(l[2], l[3])};
    }
}

struct GemmPieces {
    const float* A[2];
    const float* W[2];
    const float* bias[2];
    float* C[2];
    int lda, ldw, ldc, M, N, K, act;
};

template <bool VEC>
__global__ __launch_bounds__(256) void gemm_nt_pieces_kernel(GemmPieces g) {
    __shared__ __attribute__((aligned(16))) char s_a[3 * PIMG];
    __shared__ __attribute__((aligned(16))) char s_w[3 * PIMG];
    const int z = blockIdx.z;
    const float* __restrict__ A = z ? g.A[1] : g.A[0];
    const float* __restrict__ W = z ? g.W[1] : g.W[0];
    const float* __restrict__ bias = z ? g.bias[1] : g.bias[0];
    float* __restrict__ C = z ? g.C[1] : g.C[0];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 1, wn = wid & 1;
    const int m0 = blockIdx.y * PM, n0 = blockIdx.x * PN;
    f32x4 ra[4], rw[4];
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f32x16{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const int nk = (g.K + PK - 1) / PK;
    pieces_load<VEC>(A, g.lda, g.M, g.K, m0, 0, tid, ra);
    pieces_load<VEC>(W, g.ldw, g.N, g.K, n0, 0, tid, rw);
    // fragment of row-block i, k-step s, piece p: 8 consecutive k of row 64*w + 32*i + (lane & 31), k = 16 s + 8 (lane >> 5)
    const char* af = s_a + (wm * 64 + (lane & 31)) * PROW + (lane >> 5) * 16;
    const char* wf = s_w + (wn * 64 + (lane & 31)) * PROW + (lane >> 5) * 16;
    constexpr int PA[6] = {2, 0, 1, 1, 0, 0};  // piece products, small to large
    constexpr int PB[6] = {0, 2, 1, 0, 1, 0};
    for (int kt = 0; kt < nk; ++kt) {
        __syncthreads();  // previous slice fully consumed
        pieces_store(s_a, tid, ra);
        pieces_store(s_w, tid, rw);
        __syncthreads();
        if (kt + 1 < nk) {
            pieces_load<VEC>(A, g.lda, g.M, g.K, m0, (kt + 1) * PK, tid, ra);
            pieces_load<VEC>(W, g.ldw, g.N, g.K, n0, (kt + 1) * PK, tid, rw);
        }
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            pu32x4 fa[2][3], fw[2][3];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int p = 0; p < 3; ++p) {
                    fa[i][p] = *reinterpret_cast<const pu32x4*>(af + p * PIMG + i * 32 * PROW + s * 32);
                    fw[i][p] = *reinterpret_cast<const pu32x4*>(wf + p * PIMG + i * 32 * PROW + s * 32);
                }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int q = 0; q < 6; ++q)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(pbf16x8, fa[i][PA[q]]),
                                                                            __builtin_bit_cast(pbf16x8, fw[j][PB[q]]), acc[i][j], 0, 0, 0);
        }
    }
    // C/D map of a 32x32 accumulator: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int col = n0 + wn * 64 + 32 * j + (lane & 31);
        if (col >= g.N) continue;
        const float bv = bias ? bias[col] : 0.0f;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * 64 + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (row < g.M) {
                    float v = acc[i][j][r] + bv;
                    if (g.act == 1) v = relu_nan(v);
                    else if (g.act == 2) v = fabsf(v);
                    C[(size_t)row * g.ldc + col] = v;
                }
            }
    }
}

// One problem (A1 == nullptr) or two independent problems of the same shape in one launch.
int launch_gemm_nt_pieces(const float* A0, const float* W0, const float* bias0, float* C0, const float* A1, const float* W1,
                          const float* bias1, float* C1, int lda, int ldw, int ldc, int M, int N, int K, int act, hipStream_t st) {
    if (M == 0 || N == 0) return SHASTA_OK;
    GemmPieces g{{A0, A1}, {W0, W1}, {bias0, bias1}, {C0, C1}, lda, ldw, ldc, M, N, K, act};
    uintptr_t al = (uintptr_t)A0 | (uintptr_t)W0;
    if (A1) al |= (uintptr_t)A1 | (uintptr_t)W1;
    const bool vec = (lda % 4 == 0) && (ldw % 4 == 0) && (al % 16 == 0);
    dim3 grid(cdiv(N, PN), cdiv(M, PM), A1 ? 2 : 1);
    if (vec) hipLaunchKernelGGL(gemm_nt_pieces_kernel<true>, grid, dim3(256), 0, st, g);
    else hipLaunchKernelGGL(gemm_nt_pieces_kernel<false>, grid, dim3(256), 0, st, g);
    return check_launch("gemm_nt_pieces");
}

}  // namespace shasta

extern "C" int shasta_gemm_nt_pieces_f32(const float* A, int lda, const float* W, int ldw, const float* bias, float* C,
                                         int ldc, int M, int N, int K, int act, shasta_stream_t stream) {
    using namespace shasta;
    SHASTA_REQUIRE(A && W && C, "gemm_nt_pieces: null pointer");
    SHASTA_REQUIRE(M >= 0 && N >= 0 && K > 0 && lda >= K && ldw >= K && ldc >= N, "gemm_nt_pieces: bad size");
    SHASTA_REQUIRE(act >= 0 && act <= 2, "gemm_nt_pieces: bad activation");
    return launch_gemm_nt_pieces(A, W, bias, C, nullptr, nullptr, nullptr, nullptr, lda, ldw, ldc, M, N, K, act, as_stream(stream));
}
