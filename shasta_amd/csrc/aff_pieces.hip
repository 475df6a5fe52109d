// K6 for large batches: the six aff layers (det3d/models/tracker/shasta.py:94-106, applied :323) and the row softmax (:324)
// with every fp32 product formed from exact bf16 pieces on the bf16 matrix path - the arithmetic of anchor_split.hip /
// gemm_pieces.hip:  a = a_hi + a_mid + a_lo (8 significand bits each, cut by truncation, exact),  w * a = the six piece products
// of weight 2^0 .. 2^-16 accumulated in the fp32 accumulator of v_mfma_f32_32x32x16_bf16.  Six bf16 MFMAs of K = 16 replace
// 32 f32 MFMAs of 16x16x4: 2.7 x fewer matrix cycles per fp32 product than aff_fused_kernel (aff.hip), which stays the kernel
// of small batches and of tables wider than 512 columns.
//
// Orientation: out^T[feature][row] = W[feature][k] . h^T[k][row].  First MFMA operand = a 32-feature block of W, second =
// the 32 residual rows of the workgroup; a lane's 16 result registers are features (r&3) + 8 (r>>2) + 4 (lane>>5) of row
// lane&31, i.e. four groups of four CONSECUTIVE features: 8-byte piece writes into the next layer's image, 16-byte fp32 writes
// into the output staging.
//  * Weights are cut ONCE, at pack time, into fragment order ([layer][feature block][k step][piece][lane] x 16 B = 1 KB per
//    fragment): a wave fetches its first operand with one coalesced global_load_dwordx4 per piece, no LDS, no VALU.
//    What bounds this stage is the L2 -> CU weight traffic (0.9 MB of fragments per pass over the layers), so a workgroup owns
//    128 residual rows = 4 row blocks and every fragment feeds 4 x 6 MFMAs (with 32 rows per workgroup the kernel measured
//    1.44 ms for 257 k rows, slower than the f32 kernel's 1.33 ms).
//  * Layer 1 (K = N+2) reads the residual rows straight from global memory (8 consecutive floats per lane and k step, the
//    lines are shared by the four k steps that cover them) and cuts them in registers, in the shadow of the 12 MFMAs of the
//    step: wave = (row block, two feature blocks).
//  * Layers 2-5 (128 -> 64 -> 32 -> 64 -> 128) keep their activations in LDS as piece images [piece][row][k], alternating
//    between image A (128 wide, row stride 272 B) and image B (64 wide, 144 B): every 16-lane phase of a ds_read_b128 covers
//    all 64 banks; each value is cut once, by the producing wave.
//  * Layer 6 (128 -> N+2): wave = 2 feature blocks x 4 row blocks = 8 accumulators.  The row softmax statistics are taken
//    from the registers (per-wave partial max / sum through LDS, fixed order); the rows then pass through an fp32 staging
//    [128][256] in two column passes and leave as whole row segments: `matched` raw, `matched1` = exp(x - max) / sum.
#include "common.hpp"
#include "pair_layout.hpp"

namespace shasta {

typedef __bf16 qbf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t qu32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t qu32x2 __attribute__((ext_vector_type(2)));

constexpr int AP_WAVES = 8, AP_ROWS = 128;    // 4 row blocks of 32 per workgroup: every weight fragment feeds 4 x 6 MFMAs
constexpr int AP_AROW = 272;                  // bytes per row of image A (128 bf16 + 16 B pad: 68 dwords, conflict-free b128 reads)
constexpr int AP_BROW = 144;                  // bytes per row of image B (64 bf16 + 16 B pad: 36 dwords, conflict-free)
constexpr int AP_AIMG = AP_ROWS * AP_AROW;    // one piece of image A
constexpr int AP_BIMG = AP_ROWS * AP_BROW;
constexpr int AP_ABYTES = 3 * AP_AIMG, AP_BBYTES = 3 * AP_BIMG;  // 104448 + 55296 = 159744 B
constexpr int AP_SCOLS = 256, AP_SROW = AP_SCOLS + 4;           // output staging: 128 rows x 256 features per pass (133120 B)
constexpr int AP_STAT = AP_ROWS * AP_SROW * 4;                   // byte offset of the softmax scratch behind the staging
static_assert(AP_STAT + (2 * AP_WAVES + 2) * AP_ROWS * 4 <= AP_ABYTES + AP_BBYTES, "staging + softmax scratch must fit the two images");

__device__ __forceinline__ void ap_cut3(float a, float& h, float& m, float& l) {
    h = __uint_as_float(__float_as_uint(a) & 0xffff0000u);
    const float r = a - h;
    m = __uint_as_float(__float_as_uint(r) & 0xffff0000u);
    l = r - m;
}
__device__ __forceinline__ uint32_t ap_top2(float even, float odd) {
    return __builtin_amdgcn_perm(__float_as_uint(odd), __float_as_uint(even), 0x07060302u);
}

struct AffPackArgs {
    shasta_linear aff[6];
    uint32_t* out;
    int D;
};

// one thread per (fragment, lane): 8 weights -> three 16-byte piece vectors
__global__ __launch_bounds__(256) void aff_pieces_pack_kernel(AffPackArgs a) {
    const int D = a.D;
    for (int layer = 0; layer < 6; ++layer) {
        const int nks = ap_ksteps(layer, D), nfb = ap_fblocks(layer, D), kin = ap_kin(layer, D), nout = ap_out(layer, D);
        const float* W = a.aff[layer].weight;
        uint32_t* o = a.out + ap_layer_offset(layer, D) * 256;
        for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < nfb * nks * 64; e += gridDim.x * blockDim.x) {
            const int lane = e & 63, ks = (e >> 6) % nks, fb = (e >> 6) / nks;
            const int f = fb * 32 + (lane & 31), k0 = ks * 16 + (lane >> 5) * 8;
            float h[8], m[8], l[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float w = (f < nout && k0 + j < kin) ? W[(size_t)f * kin + k0 + j] : 0.0f;
                ap_cut3(w, h[j], m[j], l[j]);
            }
            qu32x4* dst = reinterpret_cast<qu32x4*>(o) + ((size_t)(fb * nks + ks) * 3) * 64 + lane;
            dst[0] = qu32x4{ap_top2(h[0], h[1]), ap_top2(h[2], h[3]), ap_top2(h[4], h[5]), ap_top2(h[6], h[7])};
            dst[64] = qu32x4{ap_top2(m[0], m[1]), ap_top2(m[2], m[3]), ap_top2(m[4], m[5]), ap_top2(m[6], m[7])};
            dst[128] = qu32x4{ap_top2(l[0], l[1]), ap_top2(l[2], l[3]), ap_top2(l[4], l[5]), ap_top2(l[6], l[7])};
        }
    }
}

int aff_pieces_pack(const shasta_weights* w, float* out, hipStream_t st) {
    AffPackArgs a;
    for (int i = 0; i < 6; ++i) a.aff[i] = w->aff[i];
    a.out = reinterpret_cast<uint32_t*>(out);
    a.D = w->max_obj + 2;
    hipLaunchKernelGGL(aff_pieces_pack_kernel, dim3(128), dim3(256), 0, st, a);
    return check_launch("aff_pieces_pack");
}

struct AffPiecesArgs {
    const uint32_t* wp;  // piece fragments of the six layers
    const float* bias[6];
    const float* residual;
    float* matched;  // (M, ldm) pre-softmax, for the column softmax
    float* m1;       // (B, N, D)
    int M, T, N, D, Dp, ld, ldm;
};

#define AP_MFMA(a, b, c) \
    __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(qbf16x8, (a)), __builtin_bit_cast(qbf16x8, (b)), (c), 0, 0, 0)

// the six piece products of one k step, small to large (first operand = weight pieces, second = activation pieces)
__device__ __forceinline__ void ap_step(const qu32x4 (&w)[3], const qu32x4 (&x)[3], f32x16& acc) {
    acc = AP_MFMA(w[2], x[0], acc);
    acc = AP_MFMA(w[0], x[2], acc);
    acc = AP_MFMA(w[1], x[1], acc);
    acc = AP_MFMA(w[1], x[0], acc);
    acc = AP_MFMA(w[0], x[1], acc);
    acc = AP_MFMA(w[0], x[0], acc);
}

__device__ __forceinline__ void ap_load_w(const qu32x4* frag, int lane, qu32x4 (&w)[3]) {
    w[0] = frag[lane];
    w[1] = frag[64 + lane];
    w[2] = frag[128 + lane];
}

// activation fragment of k step `ks`, row block rb, from a hidden piece image (ROW bytes per row, IMG bytes per piece)
template <int ROW, int IMG>
__device__ __forceinline__ void ap_load_h(const char* H, int rb, int ks, int lane, qu32x4 (&x)[3]) {
    const char* p = H + (rb * 32 + (lane & 31)) * ROW + (ks * 16 + (lane >> 5) * 8) * 2;
    x[0] = *reinterpret_cast<const qu32x4*>(p);
    x[1] = *reinterpret_cast<const qu32x4*>(p + IMG);
    x[2] = *reinterpret_cast<const qu32x4*>(p + 2 * IMG);
}

// acc (+ bias, ReLU) of (feature block fb, row block rb) -> the piece image of the next layer
template <int ROW, int IMG>
__device__ __forceinline__ void ap_store_h(char* H, int fb, int rb, int lane, const f32x16& acc, const float* __restrict__ bias) {
    const int n = rb * 32 + (lane & 31), hh = lane >> 5;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int f0 = fb * 32 + 8 * g + 4 * hh;
        const float b[4] = {bias[f0], bias[f0 + 1], bias[f0 + 2], bias[f0 + 3]};
        float h[4], m[4], l[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) ap_cut3(fmaxf(acc[4 * g + j] + b[j], 0.0f), h[j], m[j], l[j]);
        char* dst = H + n * ROW + f0 * 2;
        *reinterpret_cast<qu32x2*>(dst) = qu32x2{ap_top2(h[0], h[1]), ap_top2(h[2], h[3])};
        *reinterpret_cast<qu32x2*>(dst + IMG) = qu32x2{ap_top2(m[0], m[1]), ap_top2(m[2], m[3])};
        *reinterpret_cast<qu32x2*>(dst + 2 * IMG) = qu32x2{ap_top2(l[0], l[1]), ap_top2(l[2], l[3])};
    }
}

// one (feature block, row block) task of a hidden layer with KS k steps: weights from global fragments, activations from LDS
template <int KS, int ROW, int IMG>
__device__ __forceinline__ void ap_hidden_task(const qu32x4* wl, int fb, int rb, const char* Hin, int lane, f32x16& acc) {
    const qu32x4* frag = wl + (size_t)fb * KS * 3 * 64;
    qu32x4 w[KS][3];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) ap_load_w(frag + ks * 3 * 64, lane, w[ks]);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        qu32x4 x[3];
        ap_load_h<ROW, IMG>(Hin, rb, ks, lane, x);
        ap_step(w[ks], x, acc);
    }
}

__global__ __launch_bounds__(64 * AP_WAVES) void aff_pieces_kernel(AffPiecesArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* HA = smem;              // [3][128][272 B]: layer outputs of width 128 / 32
    char* HB = smem + AP_ABYTES;  // [3][128][144 B]: layer outputs of width 64
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g0 = blockIdx.x * AP_ROWS;
    const int D = a.D;
    const qu32x4* wp = reinterpret_cast<const qu32x4*>(a.wp);
    const f32x16 zero16 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};

    // ---- layer 1 (K = D -> 128): wave = (row block wid >> 1, feature blocks 2 (wid & 1) + {0, 1}); the residual rows come
    // straight from global memory, 8 consecutive floats per lane and k step, and are cut in registers ----
    {
        const int nks = ap_ksteps(0, D), rb = wid >> 1, fb0 = 2 * (wid & 1);
        const qu32x4* frag0 = wp + (size_t)fb0 * nks * 3 * 64;
        const qu32x4* frag1 = frag0 + (size_t)nks * 3 * 64;
        const int row = min(g0 + rb * 32 + (lane & 31), a.M - 1);
        const float* xr = a.residual + (size_t)row * a.ld + (lane >> 5) * 8;
        f32x16 acc0 = zero16, acc1 = zero16;
        auto load_x = [&](int ks, float (&v)[8]) {
            const int k0 = ks * 16 + (lane >> 5) * 8;
            if (k0 + 8 <= D) {  // ld = Dp is a multiple of 4 and the rows are 16-byte aligned
                const f32x4 p = *reinterpret_cast<const f32x4*>(xr + ks * 16), q = *reinterpret_cast<const f32x4*>(xr + ks * 16 + 4);
                v[0] = p[0]; v[1] = p[1]; v[2] = p[2]; v[3] = p[3]; v[4] = q[0]; v[5] = q[1]; v[6] = q[2]; v[7] = q[3];
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = k0 + j < D ? xr[ks * 16 + j] : 0.0f;
            }
        };
        qu32x4 w0c[3], w1c[3], w0n[3], w1n[3];
        float xc[8], xn[8];
        ap_load_w(frag0, lane, w0c);
        ap_load_w(frag1, lane, w1c);
        load_x(0, xc);
        for (int ks = 0; ks < nks; ++ks) {
            if (ks + 1 < nks) {
                ap_load_w(frag0 + (size_t)(ks + 1) * 3 * 64, lane, w0n);
                ap_load_w(frag1 + (size_t)(ks + 1) * 3 * 64, lane, w1n);
                load_x(ks + 1, xn);
            }
            float h[8], m[8], l[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) ap_cut3(xc[j], h[j], m[j], l[j]);
            qu32x4 x[3];
            x[0] = qu32x4{ap_top2(h[0], h[1]), ap_top2(h[2], h[3]), ap_top2(h[4], h[5]), ap_top2(h[6], h[7])};
            x[1] = qu32x4{ap_top2(m[0], m[1]), ap_top2(m[2], m[3]), ap_top2(m[4], m[5]), ap_top2(m[6], m[7])};
            x[2] = qu32x4{ap_top2(l[0], l[1]), ap_top2(l[2], l[3]), ap_top2(l[4], l[5]), ap_top2(l[6], l[7])};
            ap_step(w0c, x, acc0);
            ap_step(w1c, x, acc1);
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                w0c[p] = w0n[p];
                w1c[p] = w1n[p];
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) xc[j] = xn[j];
        }
        ap_store_h<AP_AROW, AP_AIMG>(HA, fb0, rb, lane, acc0, a.bias[0]);
        ap_store_h<AP_AROW, AP_AIMG>(HA, fb0 + 1, rb, lane, acc1, a.bias[0]);
    }
    __syncthreads();
    const qu32x4* w2 = wp + ap_layer_offset(1, D) * 64;
    const qu32x4* w3 = wp + ap_layer_offset(2, D) * 64;
    const qu32x4* w4 = wp + ap_layer_offset(3, D) * 64;
    const qu32x4* w5 = wp + ap_layer_offset(4, D) * 64;
    const qu32x4* w6 = wp + ap_layer_offset(5, D) * 64;
    {  // 128 -> 64: 2 feature blocks x 4 row blocks = one task per wave; A -> B
        f32x16 acc = zero16;
        ap_hidden_task<8, AP_AROW, AP_AIMG>(w2, wid & 1, wid >> 1, HA, lane, acc);
        ap_store_h<AP_BROW, AP_BIMG>(HB, wid & 1, wid >> 1, lane, acc, a.bias[1]);
    }
    __syncthreads();
    if (wid < 4) {  // 64 -> 32: 1 x 4 tasks; B -> A
        f32x16 acc = zero16;
        ap_hidden_task<4, AP_BROW, AP_BIMG>(w3, 0, wid, HB, lane, acc);
        ap_store_h<AP_AROW, AP_AIMG>(HA, 0, wid, lane, acc, a.bias[2]);
    }
    __syncthreads();
    {  // 32 -> 64: 2 x 4 tasks; A -> B
        f32x16 acc = zero16;
        ap_hidden_task<2, AP_AROW, AP_AIMG>(w4, wid & 1, wid >> 1, HA, lane, acc);
        ap_store_h<AP_BROW, AP_BIMG>(HB, wid & 1, wid >> 1, lane, acc, a.bias[3]);
    }
    __syncthreads();
    {  // 64 -> 128: 4 x 4 tasks, two per wave; B -> A
        const int rb = wid >> 1;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            f32x16 acc = zero16;
            ap_hidden_task<4, AP_BROW, AP_BIMG>(w5, 2 * (wid & 1) + i, rb, HB, lane, acc);
            ap_store_h<AP_AROW, AP_AIMG>(HA, 2 * (wid & 1) + i, rb, lane, acc, a.bias[4]);
        }
    }
    __syncthreads();
    // ---- layer 6 (128 -> D): wave = feature blocks {wid, wid + 8} x the 4 row blocks: 8 accumulators; every weight fragment
    // feeds 4 x 6 MFMAs ----
    const int nfb = ap_fblocks(5, D);
    f32x16 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) acc[i][rb] = zero16;
#pragma unroll 1
    for (int ks = 0; ks < 8; ++ks) {
        qu32x4 x[4][3];
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) ap_load_h<AP_AROW, AP_AIMG>(HA, rb, ks, lane, x[rb]);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int fb = wid + 8 * i;
            if (fb < nfb) {  // wave-uniform
                qu32x4 w[3];
                ap_load_w(w6 + ((size_t)fb * 8 + ks) * 3 * 64, lane, w);
#pragma unroll
                for (int rb = 0; rb < 4; ++rb) ap_step(w, x[rb], acc[i][rb]);
            }
        }
    }
    // bias; per-row maximum over this wave's features (features >= D do not take part)
    const int n = lane & 31, hh = lane >> 5;
    float mx[4];
#pragma unroll
    for (int rb = 0; rb < 4; ++rb) mx[rb] = -INFINITY;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int fb = wid + 8 * i;
        if (fb >= nfb) continue;
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int f = fb * 32 + 8 * g + 4 * hh + j;
                const float bv = f < D ? a.bias[5][f] : 0.0f;
#pragma unroll
                for (int rb = 0; rb < 4; ++rb) {
                    const float v = acc[i][rb][4 * g + j] + bv;
                    acc[i][rb][4 * g + j] = v;
                    if (f < D) mx[rb] = fmaxf(mx[rb], v);
                }
            }
    }
    __syncthreads();  // every wave is done reading image A: the staging and the softmax scratch may overwrite it
    float* xs = reinterpret_cast<float*>(smem);                       // [128][AP_SROW] staging
    float* pmax = reinterpret_cast<float*>(smem + AP_STAT);           // [8 waves][128 rows]
    float* psum = pmax + AP_WAVES * AP_ROWS;                          // [8 waves][128 rows]
    float* rmax = psum + AP_WAVES * AP_ROWS;                          // [128]
    float* rinv = rmax + AP_ROWS;                                     // [128]
#pragma unroll
    for (int rb = 0; rb < 4; ++rb) {
        const float o = fmaxf(mx[rb], __shfl_xor(mx[rb], 32, 64));
        if (hh == 0) pmax[wid * AP_ROWS + rb * 32 + n] = o;
    }
    __syncthreads();
    float rm[4], se[4];
#pragma unroll
    for (int rb = 0; rb < 4; ++rb) {
        float m = pmax[rb * 32 + n];
#pragma unroll
        for (int w = 1; w < AP_WAVES; ++w) m = fmaxf(m, pmax[w * AP_ROWS + rb * 32 + n]);
        rm[rb] = m;
        se[rb] = 0.0f;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int fb = wid + 8 * i;
        if (fb >= nfb) continue;
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int f = fb * 32 + 8 * g + 4 * hh + j;
                if (f < D) {
#pragma unroll
                    for (int rb = 0; rb < 4; ++rb) se[rb] += expf(acc[i][rb][4 * g + j] - rm[rb]);
                }
            }
    }
#pragma unroll
    for (int rb = 0; rb < 4; ++rb) {
        const float o = se[rb] + __shfl_xor(se[rb], 32, 64);
        if (hh == 0) psum[wid * AP_ROWS + rb * 32 + n] = o;
    }
    __syncthreads();
    if (tid < AP_ROWS) {
        float s = psum[tid];
#pragma unroll
        for (int w = 1; w < AP_WAVES; ++w) s += psum[w * AP_ROWS + tid];  // fixed order
        rmax[tid] = pmax[tid];
        float m = pmax[tid];
#pragma unroll
        for (int w = 1; w < AP_WAVES; ++w) m = fmaxf(m, pmax[w * AP_ROWS + tid]);
        rmax[tid] = m;
        rinv[tid] = 1.0f / s;
    }
    // ---- output: two passes of 8 feature blocks (256 columns) through the staging, whole row segments to global memory ----
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        __syncthreads();  // pass 0: rmax / rinv visible; pass 1: the read-out of pass 0 is finished
        const int fb = wid + 8 * i;
        if (fb < nfb) {
#pragma unroll
            for (int rb = 0; rb < 4; ++rb)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    *reinterpret_cast<f32x4*>(xs + (rb * 32 + n) * AP_SROW + wid * 32 + 8 * g + 4 * hh) =
                        f32x4{acc[i][rb][4 * g], acc[i][rb][4 * g + 1], acc[i][rb][4 * g + 2], acc[i][rb][4 * g + 3]};
        }
        __syncthreads();
        const int c0 = i * AP_SCOLS, ncol = min(AP_SCOLS, D - c0);
        if (ncol <= 0) continue;
        // a wave writes one row at a time: `matched` (raw) and, for t < N, matched1 = exp(x - max) / sum
        for (int r = wid; r < AP_ROWS; r += AP_WAVES) {
            const int gr = g0 + r;
            if (gr >= a.M) break;
            const float* x = xs + r * AP_SROW;
            float* mo = a.matched + (size_t)gr * a.ldm + c0;
            const int b = gr / a.T, t = gr - b * a.T;
            float* o = a.m1 + ((size_t)b * a.N + t) * D + c0;
            const float m = rmax[r], inv = rinv[r];
            for (int d = lane; d < ncol; d += 64) {
                const float v = x[d];
                mo[d] = v;
                if (t < a.N) o[d] = expf(v - m) * inv;
            }
        }
    }
}

// LDS bytes of aff_pieces_kernel (independent of the table width, which must not exceed 512 columns)
size_t aff_pieces_lds_bytes(int Dp) {
    (void)Dp;
    return (size_t)AP_ABYTES + AP_BBYTES;
}
bool aff_pieces_serves(int D) { return D <= 2 * AP_SCOLS; }

int launch_aff_pieces(const shasta_weights* w, const float* packed_pieces, const float* residual, int ld, float* matched, int ldm,
                      float* m1, int M, hipStream_t st) {
    const int N = w->max_obj, T = N + 2, D = N + 2, Dp = (T + 3) / 4 * 4;
    AffPiecesArgs a;
    a.wp = reinterpret_cast<const uint32_t*>(packed_pieces);
    for (int i = 0; i < 6; ++i) a.bias[i] = w->aff[i].bias;
    a.residual = residual;
    a.matched = matched;
    a.m1 = m1;
    a.M = M;
    a.T = T;
    a.N = N;
    a.D = D;
    a.Dp = Dp;
    a.ld = ld;
    a.ldm = ldm;
    const size_t lds = aff_pieces_lds_bytes(Dp);
    (void)hipFuncSetAttribute((const void*)aff_pieces_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(aff_pieces_kernel, dim3(cdiv(M, AP_ROWS)), dim3(64 * AP_WAVES), lds, st, a);
    return check_launch("aff_pieces");
}

}  // namespace shasta
