// K4c, fp16-piece form on 32-wide tiles (SHASTA_OPT_F16X2_PAIR at F = 320, the width of every shipped class configuration:
// configs/nusc/car.py:22-39 num_point = 5): the per-pair MLP tails of det3d/models/tracker/shasta.py:286-319 with their SECOND layers
// (fuse_shape 40 -> 20, res_coeff 72 -> 18, fuse_det 32 -> 8: 2480 of the 2743 multiply-adds per pair) on v_mfma_f32_32x32x16_f16,
// same arithmetic as pair_f16.hip (two fp16 pieces per value, w a = w_l a_h + w_h a_l + w_h a_h, fp32 accumulation, exact descaling).
//
// Why another tiling than pair_f16_kernel (16x16x32, F = 256): at F = 320 the first-layer widths 40 | 72 | 32 are no multiples of 32.
//  * K axis: the 144 columns of a UP / UC row as they lie - [fuse_shape 40 | res_coeff 72 | fuse_det 32] - are nine 16-wide k steps; the
//    weight fragments carry zeros where a k step holds another MLP's columns, so no activation is padded or moved (k step 2 holds
//    fuse_shape 32..39 and res_coeff 0..7: it feeds both output blocks).
//  * M axis: two 32-row output blocks: block 1 = [fuse_shape.2 rows 0..19 | fuse_det.2 rows 0..7 | 4 zero rows], k steps 0 1 2 | 7 8;
//    block 2 = [res_coeff.2 rows 0..17 | 14 zero rows], k steps 2..6: 10 fragments, 30 MFMAs per 32 pairs (16x16x32 tiles would need
//    33 per 16 pairs at twice the cycles per 32 pairs... 9 % more matrix time and 17 % more values to cut, its k padded to 64 | 96 | 32).
//  * N axis: 32 pairs per MFMA.  A wave owns 32 detections and walks the tracks two at a time: sub-step A = (track t, the 32 detections),
//    sub-step B = (track t + 1, the same detections).  Lane (n = lane & 31, kb = lane >> 5) cuts h1[16 ks + 8 kb + j], j < 8, of
//    detection n for both: its 72 UC values never change - they are loaded ONCE into registers (no detection tile in LDS, no LDS read
//    per pair); the UP values of a track are LDS broadcasts.
//  * Transposition: a 32x32 result leaves lane (n, hb) with rows 8 g + 4 hb + r of pair n.  v_permlane32_swap of the registers of
//    sub-step A with those of sub-step B gives every lane ALL rows of ONE pair - lanes 0..31: (t, n), lanes 32..63: (t + 1, n) - in
//    quads of four consecutive rows, which is what the lane-per-pair phase (layers 3 - 4 on v_mfma_f32_4x4x1, hand-designed residual,
//    combine: as pair_mfma4_kernel) reads.  No transposition tile in LDS (106 KB at 8 waves with the pair_f16_kernel scheme).
// F = 256 through this kernel (8 fragments, 48 MFMAs of 32 cycles per 64 pairs instead of pair_f16_kernel's 48 of 16) was measured:
// 4.71 - 4.84 ms against 4.10 ms at 512 frame-pairs of N = 500 - at that width the 32-row blocks are a third empty and the kernel turns
// from vector-bound into matrix-bound; F = 256 stays on pair_f16_kernel.
// Registers: 72 for the UC values + 64 accumulators + the pipeline's operands = 249: TWO waves per SIMD (two 4-wave workgroups per CU);
// the weight pieces (80 registers) stay in LDS and are read per k step, 20 reads per 64 pairs.  (With them in registers - 256 + 74
// registers, one wave per SIMD - the vector and matrix work of a wave did not overlap: 4.37 ms for the stage at N = 500 x 256 against
// 3.51 ms.)  Where the time goes per 64 pairs and wave: 60 MFMAs of 32 cycles + 102 v_mfma_f32_4x4x1 of 8 = 2736 matrix cycles, ~780
// vector instructions (the cut is 3 per value: v_pk_fma_f32 with clamp = ReLU, v_pk_mul_f32 2^14, v_cvt_pk_f16_f32, two v_fma_mix_f32,
// v_cvt_pk_f16_f32 at 5.3 / 5.3 / 8.1 / 7 / 7 / 8.1 cycles per two values) = ~4200: with two waves per SIMD the matrix pipe is busy
// ~85 % of the time at the ~1.7 GHz the power cap leaves under this load - removing vector work (the in-place UP scaling below:
// -68 instructions) moved the time by 1.6 %.
#include "common.hpp"
#include "pair_layout.hpp"

#include <type_traits>

namespace shasta {

typedef _Float16 wh16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 wh16x2 __attribute__((ext_vector_type(2)));
typedef uint32_t wu4 __attribute__((ext_vector_type(4)));
typedef float wf2 __attribute__((ext_vector_type(2)));

#define MFMA4W(a, b, c) __builtin_amdgcn_mfma_f32_4x4x1f32((a), (b), (c), 0, 0, 0)
#define MFMA32H(a, b, c) \
    __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(wh16x8, (a)), __builtin_bit_cast(wh16x8, (b)), (c), 0, 0, 0)

template <int F>
struct PW {
    static constexpr PairDims dm{F};
    static constexpr int ET = dm.ET, H1 = dm.H1, R1 = dm.R1, H2 = dm.H2, R2 = dm.R2, NKS = ET / 16;
    static_assert(ET % 16 == 0 && H2 % 4 == 0 && H2 + 8 <= 32 && R2 <= 32, "wide pair tiles: 16-wide k steps, two 32-row output blocks");
    // does k step ks hold columns of block 1 (fuse_shape: [0, H1), fuse_det: [H1 + R1, ET)) / block 2 (res_coeff: [H1, H1 + R1))?
    static constexpr bool has1(int ks) { return 16 * ks < H1 || 16 * ks + 16 > H1 + R1; }
    static constexpr bool has2(int ks) { return 16 * ks + 16 > H1 && 16 * ks < H1 + R1; }
    static constexpr int frag1(int ks) {  // fragment index: k steps in order, block 1 before block 2
        int f = 0;
        for (int i = 0; i < ks; ++i) f += (has1(i) ? 1 : 0) + (has2(i) ? 1 : 0);
        return f;
    }
    static constexpr int frag2(int ks) { return frag1(ks) + (has1(ks) ? 1 : 0); }
    static constexpr int NFRAG = frag1(NKS);
    static constexpr int FRAG_DW = 2 * NFRAG * 64 * 4;  // [piece][fragment][lane][4 dwords]
};
constexpr int P16W_MAX_DW = 2 * 16 * 64 * 4 + 4;
static_assert(PW<320>::FRAG_DW + 4 <= P16W_MAX_DW && PW<256>::FRAG_DW + 4 <= P16W_MAX_DW, "PackedLayout::p16w");

template <int F, int L>
struct A4w {
    static constexpr LayerDesc D = layer_desc(F, L);
    static constexpr int NOB = a4_nob(F, L), KG = a4_kg(F, L), OFF = a4_offset(F, L), KIN = D.kin, BIAS = NOB * KG * 16;
};

// (the three asm helpers and their hazard rule: pair_f16.hip)
__device__ __forceinline__ wf2 w_fma2_relu01(wf2 a, wf2 c, wf2 b) {
    wf2 r;
    asm("v_pk_fma_f32 %0, %1, %2, %3 clamp" : "=v"(r) : "v"(a), "v"(c), "v"(b));
    return r;
}
__device__ __forceinline__ wf2 w_mul2(wf2 a, wf2 c) {  // (the compiler splits a wf2 product of an asm result into two v_mul_f32)
    wf2 r;
    asm("v_pk_mul_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(c));
    return r;
}
__device__ __forceinline__ float w_res_lo(float x, uint32_t hpk) {
    float r;
    asm("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel_hi:[0,0,1]" : "=v"(r) : "v"(x), "v"(hpk));
    return r;
}
__device__ __forceinline__ float w_res_hi(float x, uint32_t hpk) {
    float r;
    asm("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(r) : "v"(x), "v"(hpk));
    return r;
}
__device__ __forceinline__ uint32_t w_cvt2(float a, float b) {
    const wh16x2 v = {(_Float16)a, (_Float16)b};
    return __builtin_bit_cast(uint32_t, v);
}

// ---- pack: the three second layers as A operands of v_mfma_f32_32x32x16_f16 -------------------------------------------------
// fragment (block, k step): lane (m = lane & 31, kb = lane >> 5) holds Wblock[m][16 ks + 8 kb + j] * 2^e_mlp, j < 8, where row m of
// block 1 is fuse_shape.2 row m (m < H2, columns [0, H1)) or fuse_det.2 row m - H2 (m < H2 + 8, columns [H1 + R1, ET)) and row m of
// block 2 is res_coeff.2 row m (m < R2, columns [H1, H1 + R1)); zero elsewhere.  High pieces of all fragments, then low pieces, then
// the three exponents (fs, rc, fd).
struct PairF16WPackArgs {
    const float* w_fs2;  // fuse_shape.2.weight (H2, H1)
    const float* w_rc2;  // res_coeff.2.weight (R2, R1)
    const float* w_fd2;  // fuse_det.2.weight (8, 32)
    uint32_t* out;
};

template <int F>
__global__ __launch_bounds__(256) void pair_f16w_pack_kernel(PairF16WPackArgs a) {
    using W = PW<F>;
    __shared__ float red[3][4];
    __shared__ int ex[3];
    const int tid = threadIdx.x;
    const float* Wm[3] = {a.w_fs2, a.w_rc2, a.w_fd2};
    const int cnt[3] = {W::H2 * W::H1, W::R2 * W::R1, 8 * 32};
    for (int m = 0; m < 3; ++m) {
        float mx = 0.0f;
        for (int i = tid; i < cnt[m]; i += 256) mx = fmaxf(mx, fabsf(Wm[m][i]));
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
        if ((tid & 63) == 0) red[m][tid >> 6] = mx;
    }
    __syncthreads();
    if (tid < 3) {
        const float mx = fmaxf(fmaxf(red[tid][0], red[tid][1]), fmaxf(red[tid][2], red[tid][3]));
        ex[tid] = range_exponent_bits(__float_as_uint(mx));
        reinterpret_cast<int*>(a.out)[W::FRAG_DW + tid] = ex[tid];
    }
    if (tid == 3) a.out[W::FRAG_DW + 3] = 0;
    __syncthreads();
    const int lane = tid & 63, m = lane & 31, kb = lane >> 5;
    for (int item = tid >> 6; item < 2 * W::NKS; item += 4) {
        const int ks = item >> 1, blk = item & 1;
        if (!(blk == 0 ? W::has1(ks) : W::has2(ks))) continue;
        const int frag = blk == 0 ? W::frag1(ks) : W::frag2(ks);
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int c = 16 * ks + 8 * kb + j;
            float x = 0.0f;
            if (blk == 0) {
                if (m < W::H2 && c < W::H1) x = __builtin_ldexpf(a.w_fs2[m * W::H1 + c], ex[0]);
                else if (m >= W::H2 && m < W::H2 + 8 && c >= W::H1 + W::R1) x = __builtin_ldexpf(a.w_fd2[(m - W::H2) * 32 + (c - W::H1 - W::R1)], ex[2]);
            } else if (m < W::R2 && c >= W::H1 && c < W::H1 + W::R1) {
                x = __builtin_ldexpf(a.w_rc2[m * W::R1 + (c - W::H1)], ex[1]);
            }
            v[j] = x;
        }
        wu4 hi, lo;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const _Float16 h0 = (_Float16)v[2 * j], h1 = (_Float16)v[2 * j + 1];
            const wh16x2 hh = {h0, h1};
            hi[j] = __builtin_bit_cast(uint32_t, hh);
            lo[j] = w_cvt2(v[2 * j] - (float)h0, v[2 * j + 1] - (float)h1);
        }
        reinterpret_cast<wu4*>(a.out)[(0 * W::NFRAG + frag) * 64 + lane] = hi;
        reinterpret_cast<wu4*>(a.out)[(1 * W::NFRAG + frag) * 64 + lane] = lo;
    }
}

bool pair_f16w_serves(int F) { return F == 320; }

int pair_f16w_pack(const shasta_weights* w, float* out, hipStream_t st) {
    PairF16WPackArgs a;
    a.w_fs2 = w->fuse_shape[1].weight;
    a.w_rc2 = w->res_coeff[1].weight;
    a.w_fd2 = w->fuse_det[1].weight;
    a.out = reinterpret_cast<uint32_t*>(out);
    hipLaunchKernelGGL(pair_f16w_pack_kernel<320>, dim3(1), dim3(256), 0, st, a);
    return check_launch("pair_f16w_pack");
}

// ---- the kernel -------------------------------------------------------------------------------------------------------------
constexpr int PWK_WPB = 4;       // waves per workgroup (two workgroups per CU: two waves per SIMD)
constexpr int PWK_SLOT = 256;    // floats per UP row slot (a row = ET + 16 hand floats; one 1 KB LDS-DMA per row)

template <int F>
__global__ __launch_bounds__(64 * PWK_WPB) __attribute__((amdgpu_waves_per_eu(2, 2))) void pair_f16w_kernel(const float* __restrict__ packed, const uint32_t* __restrict__ p16,
                                                               const float* __restrict__ UP, const float* __restrict__ UC,
                                                               const float* __restrict__ hand_prev, const float* __restrict__ hand_det,
                                                               const float* __restrict__ denom, float* __restrict__ residual, int T,
                                                               int D, int ld, int TWG) {
    using W = PW<F>;
    constexpr int ET = W::ET, NKS = W::NKS, NFRAG = W::NFRAG;
    constexpr int NA4 = a4_total(F);
    extern __shared__ __attribute__((aligned(16))) float s_dynw[];
    float* s_a4 = s_dynw;                          // [NA4] 4x4x1 operand table (layers 3-4 and the layer-2 biases)
    float* s_up = s_dynw + ((NA4 + 3) & ~3);       // [WPB][3 steps][2 rows][PWK_SLOT]
    wu4* s_w = reinterpret_cast<wu4*>(s_up + PWK_WPB * 6 * PWK_SLOT);  // [piece][fragment][lane] second-layer weight pieces
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    int lbx, by, b;
    xcd_logical_block(lbx, by, b);  // the detection tiles of a frame on one XCD (common.hpp)
    const int d0 = lbx * 32;
    const int n = lane & 31, kb = lane >> 5;
    const int d = d0 + n, dcl = min(d, D - 1);
    const PackedLayout P(0, 0, F);
    {
        const f32x4* asrc = reinterpret_cast<const f32x4*>(packed + P.a4);
#pragma unroll 2
        for (int e = tid; e < NA4 / 4; e += 64 * PWK_WPB) reinterpret_cast<f32x4*>(s_a4)[e] = asrc[e];
    }
    // this lane's UC values, for the whole kernel: columns 16 ks + 8 kb .. + 7 of detection n
    f32x4 uc[NKS][2];
    {
        const f32x4* src = reinterpret_cast<const f32x4*>(UC + ((size_t)b * D + dcl) * ET);
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            uc[ks][0] = src[4 * ks + 2 * kb];
            uc[ks][1] = src[4 * ks + 2 * kb + 1];
        }
    }
    float hd[12];
    float mc;  // largest |UC| of this lane's detection row (row_prep), then of the whole 32-detection tile
    {
        const f32x4* h = reinterpret_cast<const f32x4*>(hand_det + ((size_t)b * D + dcl) * 16);
        const f32x4 a = h[0], c = h[1], e = h[2], g = h[3];
        hd[0] = a[0]; hd[1] = a[1]; hd[2] = a[2]; hd[3] = a[3]; hd[4] = c[0]; hd[5] = c[1]; hd[6] = c[2];
        hd[7] = e[0]; hd[8] = e[1]; hd[9] = e[2]; hd[10] = e[3]; hd[11] = g[0];
        mc = g[1];
    }
#pragma unroll
    for (int off = 16; off > 0; off >>= 1) mc = absmax_keep_nan(mc, __shfl_xor(mc, off, 64));  // (both halves hold the same 32 rows)
    const float dnm = denom[(size_t)b * D + dcl], rdn = 1.0f / dnm;
    // second-layer weight pieces: registers for the whole kernel
    for (int e = tid; e < 2 * NFRAG * 64; e += 64 * PWK_WPB) s_w[e] = reinterpret_cast<const wu4*>(p16)[e];
    const wu4* my_w = s_w + lane;
    const int ew_fs = reinterpret_cast<const int*>(p16)[W::FRAG_DW + 0], ew_rc = reinterpret_cast<const int*>(p16)[W::FRAG_DW + 1],
              ew_fd = reinterpret_cast<const int*>(p16)[W::FRAG_DW + 2];
    __syncthreads();
    typedef __attribute__((address_space(3))) float lfloat;
    typedef __attribute__((address_space(3))) f32x4 lf32x4;
    const unsigned arow_base = (unsigned)(unsigned long long)(s_a4 + (lane & 3) * 4);
    const unsigned abias_base = (unsigned)(unsigned long long)(s_a4 + (lane & 3));
    const f32x4 zero4 = {0, 0, 0, 0};
    const f32x16 zero16 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};

    // TWG tracks per workgroup (a multiple of 2 WPB), the same number to every wave, two per step
    const int tw = TWG / PWK_WPB;
    const int t_beg = by * TWG + wid * tw;
    const int t_end = min(min(T, (by + 1) * TWG), t_beg + tw);
    float* my_up = s_up + wid * (6 * PWK_SLOT);
    const bool hp_lane = lane >= ET / 4 && lane < ET / 4 + 4;
    const int up_lane = 4 * min(lane, ET / 4 - 1), hp_off = 4 * (lane - ET / 4);
    auto dma_up = [&](int row, int slot) __attribute__((always_inline)) {
        const size_t r = (size_t)b * T + min(row, T - 1);
        const float* src = hp_lane ? hand_prev + r * 16 + hp_off : UP + r * ET + up_lane;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(my_up + slot * PWK_SLOT), 16, 0, 0);
    };
    if (t_beg < t_end) {
        dma_up(t_beg, 0);
        dma_up(t_beg + 1, 1);
        dma_up(t_beg + 2, 2);
        dma_up(t_beg + 3, 3);
    }
    const wf2 c14 = {16384.0f, 16384.0f};
    int step = 0;
    for (int t = t_beg; t < t_end; t += 2, ++step) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        dma_up(t + 4, ((step + 2) % 3) * 2);
        dma_up(t + 5, ((step + 2) % 3) * 2 + 1);
        unsigned upo = (unsigned)(unsigned long long)(my_up + (step % 3) * 2 * PWK_SLOT);
        asm volatile("" : "+v"(upo));
        const lfloat* upA = (const lfloat*)(unsigned long long)upo;
        const lfloat* upB = upA + PWK_SLOT;
        // the two tracks' scales (uniform): every h1 of a track and this tile is at most max |UP[t]| + max |UC| (pair_f16.hip)
        const float boundA = upA[ET + 13] + mc, boundB = upB[ET + 13] + mc;
        const int e1A = range_exponent_bits(__float_as_uint(boundA)), e1B = range_exponent_bits(__float_as_uint(boundB));
        const float csA = __builtin_ldexpf(1.0f, e1A - 14), csB = __builtin_ldexpf(1.0f, e1B - 14);
        const wf2 cs2A = {csA, csA}, cs2B = {csB, csB};
        // the two UP rows are scaled ONCE, in place (lane i takes float4 i of both; the lanes behind the row repeat its last float4):
        // 4 packed multiplies per step instead of the 72 that every lane would spend on the values it reads back below
        {
            typedef __attribute__((address_space(3))) f32x4 wlf32x4;
            const int q = 4 * min(lane, ET / 4 - 1);
            wlf32x4* pa = (wlf32x4*)(unsigned long long)(upo + 4 * q);
            wlf32x4* pb = (wlf32x4*)(unsigned long long)(upo + 4 * (PWK_SLOT + q));
            const f32x4 va = *pa, vb = *pb;
            const wf2 a0 = w_mul2(wf2{va[0], va[1]}, cs2A), a1 = w_mul2(wf2{va[2], va[3]}, cs2A);
            const wf2 b0 = w_mul2(wf2{vb[0], vb[1]}, cs2B), b1 = w_mul2(wf2{vb[2], vb[3]}, cs2B);
            *pa = f32x4{a0[0], a0[1], a1[0], a1[1]};
            *pb = f32x4{b0[0], b0[1], b1[0], b1[1]};
        }

        f32x16 acc1A = zero16, acc2A = zero16, acc1B = zero16, acc2B = zero16;
        // pieces of h1 for one k step of one sub-step: relu(UP + UC) scaled into [0, 2^14], cut in two
        auto cut = [&](const f32x4& a, const f32x4& c, const wf2 cs2, int ks, wu4& xh, wu4& xl) __attribute__((always_inline)) {
            const wf2 upv[4] = {wf2{a[0], a[1]}, wf2{a[2], a[3]}, wf2{c[0], c[1]}, wf2{c[2], c[3]}};  // (scaled in place above)
#pragma unroll
            for (int j2 = 0; j2 < 2; ++j2) {
                const f32x4 uu = uc[ks][j2];
                const wf2 sa = w_mul2(w_fma2_relu01(wf2{uu[0], uu[1]}, cs2, upv[2 * j2]), c14);
                const wf2 sb = w_mul2(w_fma2_relu01(wf2{uu[2], uu[3]}, cs2, upv[2 * j2 + 1]), c14);
                const uint32_t hA = w_cvt2(sa[0], sa[1]), hB = w_cvt2(sb[0], sb[1]);
                xh[2 * j2] = hA;
                xh[2 * j2 + 1] = hB;
                xl[2 * j2] = w_cvt2(w_res_lo(sa[0], hA), w_res_hi(sa[1], hA));
                xl[2 * j2 + 1] = w_cvt2(w_res_lo(sb[0], hB), w_res_hi(sb[1], hB));
            }
        };
        // the LDS reads of k step ks + 1 (UP values of both tracks, weight pieces) are issued before the arithmetic of k step ks
        struct KIn {
            f32x4 a0, a1, b0, b1;
            wu4 w1l, w1h, w2l, w2h;
        };
        auto fetch = [&](int ks, KIn& k) __attribute__((always_inline)) {
            k.a0 = *reinterpret_cast<const lf32x4*>(upA + 16 * ks + 8 * kb);
            k.a1 = *reinterpret_cast<const lf32x4*>(upA + 16 * ks + 8 * kb + 4);
            k.b0 = *reinterpret_cast<const lf32x4*>(upB + 16 * ks + 8 * kb);
            k.b1 = *reinterpret_cast<const lf32x4*>(upB + 16 * ks + 8 * kb + 4);
            if (W::has1(ks)) {
                k.w1l = my_w[(NFRAG + W::frag1(ks)) * 64];
                k.w1h = my_w[W::frag1(ks) * 64];
            }
            if (W::has2(ks)) {
                k.w2l = my_w[(NFRAG + W::frag2(ks)) * 64];
                k.w2h = my_w[W::frag2(ks) * 64];
            }
        };
        KIn kin[2];
        fetch(0, kin[0]);
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            if (ks + 1 < NKS) fetch(ks + 1, kin[(ks + 1) & 1]);
            const KIn& k = kin[ks & 1];
            wu4 xhA, xlA, xhB, xlB;
            cut(k.a0, k.a1, cs2A, ks, xhA, xlA);
            cut(k.b0, k.b1, cs2B, ks, xhB, xlB);
            if (W::has1(ks)) {  // (compile-time after the unrolling)
                acc1A = MFMA32H(k.w1l, xhA, acc1A);
                acc1B = MFMA32H(k.w1l, xhB, acc1B);
                acc1A = MFMA32H(k.w1h, xlA, acc1A);
                acc1B = MFMA32H(k.w1h, xlB, acc1B);
                acc1A = MFMA32H(k.w1h, xhA, acc1A);
                acc1B = MFMA32H(k.w1h, xhB, acc1B);
            }
            if (W::has2(ks)) {
                acc2A = MFMA32H(k.w2l, xhA, acc2A);
                acc2B = MFMA32H(k.w2l, xhB, acc2B);
                acc2A = MFMA32H(k.w2h, xlA, acc2A);
                acc2B = MFMA32H(k.w2h, xlB, acc2B);
                acc2A = MFMA32H(k.w2h, xhA, acc2A);
                acc2B = MFMA32H(k.w2h, xhB, acc2B);
            }
        }
        // ---- transposition: lanes 0..31 take pair (t, n), lanes 32..63 pair (t + 1, n) --------------------------------------
        // after the swap X[4 g + r] = row 8 g + r, Y[4 g + r] = row 8 g + 4 + r of the lane's own pair
        f32x4 q1[8], q2[8];  // quads of four consecutive rows: q[2 g] = rows 8 g .. 8 g + 3, q[2 g + 1] = rows 8 g + 4 .. 8 g + 7
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const auto s1 = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc1A[4 * g + r]), __float_as_uint(acc1B[4 * g + r]), false, false);
                q1[2 * g][r] = __uint_as_float(s1[0]);
                q1[2 * g + 1][r] = __uint_as_float(s1[1]);
                if (8 * g < W::R2) {
                    const auto s2 = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc2A[4 * g + r]), __float_as_uint(acc2B[4 * g + r]), false, false);
                    q2[2 * g][r] = __uint_as_float(s2[0]);
                    q2[2 * g + 1][r] = __uint_as_float(s2[1]);
                }
            }
        // ---- lane = pair from here on ----------------------------------------------------------------------------------------
        const int my_t = t + kb;
        const lfloat* upL = upA + kb * PWK_SLOT;
        float hp[16];
        {
            const f32x4 h0 = *reinterpret_cast<const lf32x4*>(upL + ET), h1 = *reinterpret_cast<const lf32x4*>(upL + ET + 4),
                        h2 = *reinterpret_cast<const lf32x4*>(upL + ET + 8), h3 = *reinterpret_cast<const lf32x4*>(upL + ET + 12);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                hp[k] = h0[k];
                hp[4 + k] = h1[k];
                hp[8 + k] = h2[k];
                hp[12 + k] = h3[k];
            }
        }
        const float bound = kb ? boundB : boundA;
        const bool finite_bound = bound < INFINITY;  // false for NaN and +inf (pair_f16.hip)
        const int e1 = kb ? e1B : e1A;
        unsigned ao = arow_base, bo = abias_base;
        asm volatile("" : "+v"(ao), "+v"(bo));
        const lfloat* arow = (const lfloat*)(unsigned long long)ao;
        const lfloat* abias = (const lfloat*)(unsigned long long)bo;
        const float i_rc = __builtin_ldexpf(1.0f, -(e1 + ew_rc)), i_fs = __builtin_ldexpf(1.0f, -(e1 + ew_fs)), i_fd = __builtin_ldexpf(1.0f, -(e1 + ew_fd));
        constexpr int NQ_FS = W::H2 / 4, NQ_RC = (W::R2 + 3) / 4;
        f32x4 a_fs2[NQ_FS], a_rc2[NQ_RC], a_fd2[2];
        {
            const float* b_rc = s_a4 + A4w<F, L_RC2>::OFF + A4w<F, L_RC2>::BIAS;
            const float* b_fs = s_a4 + A4w<F, L_FS2>::OFF + A4w<F, L_FS2>::BIAS;
            const float* b_fd = s_a4 + A4w<F, L_FD2>::OFF + A4w<F, L_FD2>::BIAS;
            auto fma4 = [&](const f32x4& v, float sc, const f32x4& bb) __attribute__((always_inline)) {
                const wf2 s2 = {sc, sc};
                const wf2 lo = __builtin_elementwise_fma(wf2{v[0], v[1]}, s2, wf2{bb[0], bb[1]});
                const wf2 hi = __builtin_elementwise_fma(wf2{v[2], v[3]}, s2, wf2{bb[2], bb[3]});
                return f32x4{fmaxf(lo[0], 0.0f), fmaxf(lo[1], 0.0f), fmaxf(hi[0], 0.0f), fmaxf(hi[1], 0.0f)};
            };
#pragma unroll
            for (int g = 0; g < NQ_FS; ++g) a_fs2[g] = fma4(q1[g], i_fs, *reinterpret_cast<const f32x4*>(b_fs + 4 * g));
#pragma unroll
            for (int g = 0; g < 2; ++g) a_fd2[g] = fma4(q1[NQ_FS + g], i_fd, *reinterpret_cast<const f32x4*>(b_fd + 4 * g));
#pragma unroll
            for (int g = 0; g < NQ_RC; ++g) a_rc2[g] = fma4(q2[g], i_rc, *reinterpret_cast<const f32x4*>(b_rc + 4 * g));
        }
        auto init = [&](auto tag, f32x4* acc) __attribute__((always_inline)) {
            using AL = decltype(tag);
#pragma unroll
            for (int ob = 0; ob < AL::NOB; ++ob) acc[ob] = MFMA4W(abias[AL::OFF + AL::BIAS + ob * 4], 1.0f, zero4);
        };
        auto layer = [&](auto tag, const f32x4* in, f32x4* acc, auto relu_done) __attribute__((always_inline)) {
            using AL = decltype(tag);
            init(tag, acc);
#pragma unroll
            for (int kg = 0; kg < AL::KG; ++kg) {
                f32x4 a4[AL::NOB];
#pragma unroll
                for (int ob = 0; ob < AL::NOB; ++ob) a4[ob] = *reinterpret_cast<const lf32x4*>(arow + AL::OFF + (ob * AL::KG + kg) * 16);
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    if (4 * kg + kk < AL::KIN) {
                        const float h = decltype(relu_done)::value ? in[kg][kk] : fmaxf(in[kg][kk], 0.0f);
#pragma unroll
                        for (int ob = 0; ob < AL::NOB; ++ob) acc[ob] = MFMA4W(a4[ob][kk], h, acc[ob]);
                    }
                }
            }
        };
        f32x4 a_rc3[A4w<F, L_RC3>::NOB], a_fs3[A4w<F, L_FS3>::NOB], a_fs4[A4w<F, L_FS4>::NOB], a_fd3[A4w<F, L_FD3>::NOB];
        {
            // the three third layers side by side: every accumulator is touched once per round, so no 4x4x1 waits for the one before it
            using FS = A4w<F, L_FS3>;
            using RC = A4w<F, L_RC3>;
            using FD = A4w<F, L_FD3>;
            static_assert(RC::NOB == 1 && FD::NOB == 1 && FD::KG == 2, "third layers: res_coeff -> 3, fuse_det -> 1");
            init(FS{}, a_fs3);
            init(RC{}, a_rc3);
            init(FD{}, a_fd3);
            constexpr int KGM = FS::KG > RC::KG ? FS::KG : RC::KG;
#pragma unroll
            for (int kg = 0; kg < KGM; ++kg) {
                f32x4 w_fs[FS::NOB], w_rc = zero4, w_fd = zero4;
#pragma unroll
                for (int ob = 0; ob < FS::NOB; ++ob)
                    if (kg < FS::KG) w_fs[ob] = *reinterpret_cast<const lf32x4*>(arow + FS::OFF + (ob * FS::KG + kg) * 16);
                if (kg < RC::KG) w_rc = *reinterpret_cast<const lf32x4*>(arow + RC::OFF + kg * 16);
                if (kg < FD::KG) w_fd = *reinterpret_cast<const lf32x4*>(arow + FD::OFF + kg * 16);
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    if (kg < FS::KG && 4 * kg + kk < FS::KIN)
#pragma unroll
                        for (int ob = 0; ob < FS::NOB; ++ob) a_fs3[ob] = MFMA4W(w_fs[ob][kk], a_fs2[kg][kk], a_fs3[ob]);
                    if (kg < RC::KG && 4 * kg + kk < RC::KIN) a_rc3[0] = MFMA4W(w_rc[kk], a_rc2[kg][kk], a_rc3[0]);
                    if (kg < FD::KG) a_fd3[0] = MFMA4W(w_fd[kk], a_fd2[kg][kk], a_fd3[0]);
                }
            }
        }
        layer(A4w<F, L_FS4>{}, a_fs3, a_fs4, std::false_type{});
        // ---- hand-designed residual (shasta.py:277-283) and combine (shasta.py:316-319) ----
        const float dist = hand_dist(hp, hd, dnm, rdn);
        const float res = (a_rc3[0][0] * a_fd3[0][0] + a_rc3[0][1] * dist) + a_rc3[0][2] * a_fs4[0][0];
        if (d < D && my_t < t_end) residual[((size_t)b * T + my_t) * ld + d] = finite_bound ? res : __builtin_nanf("");
    }
}

size_t pair_f16w_lds_bytes(int F) {
    return ((size_t)((a4_total(F) + 3) & ~3) + (size_t)PWK_WPB * 6 * PWK_SLOT) * sizeof(float) + (size_t)PW<320>::FRAG_DW * 4;
}

int launch_pair_f16w(const float* packed, const float* p16, const float* UP, const float* UC, const float* hand_prev,
                     const float* hand_det, const float* denom, float* residual, int B, int T, int D, int ld, int F, hipStream_t st) {
    if (F != 320) {
        set_error_msg("launch_pair_f16w: feat_dim must be 320");
        return SHASTA_E_ARG;
    }
    // tracks per workgroup: the T tracks dealt evenly to the ny workgroups of a detection tile (two per wave and step), ny the smallest
    // power of two that leaves two rounds of one-wave-per-SIMD workgroups on the 256 CUs (see launch_pair_f16)
    constexpr int unit = 2 * PWK_WPB;
    int ny = 1;
    while ((long)B * cdiv(D, 32) * ny < 512 && cdiv(T, unit * ny * 2) >= 2) ny *= 2;
    while ((long)B * cdiv(D, 32) * ny < 256 && cdiv(T, unit * ny * 2) >= 1) ny *= 2;
    const int twg = cdiv(cdiv(T, ny), unit) * unit;
    const size_t lds = pair_f16w_lds_bytes(F);
    dim3 grd(cdiv(D, 32), cdiv(T, twg), B);
    if (hipFuncSetAttribute((const void*)pair_f16w_kernel<320>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
        (void)hipGetLastError();
        set_error_msg("pair_f16w: the device does not grant the kernel's LDS per workgroup");
        return SHASTA_E_UNSUPPORTED;
    }
    hipLaunchKernelGGL((pair_f16w_kernel<320>), grd, dim3(64 * PWK_WPB), lds, st, packed, reinterpret_cast<const uint32_t*>(p16), UP, UC,
                       hand_prev, hand_det, denom, residual, T, D, ld, twg);
    return check_launch("pair_f16w");
}

}  // namespace shasta
