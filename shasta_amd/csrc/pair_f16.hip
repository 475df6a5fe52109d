// K4c, fp16-piece form (SHASTA_OPT_F16X2_PAIR, F = 256): the per-pair MLP tails of det3d/models/tracker/shasta.py:286-319 with
// their SECOND layers - 1792 of the 1984 multiply-adds per pair - on the f16 matrix path, everything else as in pair_mfma4_kernel.
//
// Why only the second layers: the activations h1 = relu(UP[t] + UC[d]) exist per pair, so cutting them into pieces is VALU work
// per pair; it pays where one cut value feeds 16 multiply-adds (layer 2: 64 -> 16, 32 -> 16, 32 -> 8) and not for the narrow
// layers behind it.  Arithmetic (the two-piece fp16 form of anchor_split.hip): a * 2^e = h + l with h = fp16(a 2^e),
// l = fp16(a 2^e - h), both rounded to nearest (|err| <= 2^-24 |a 2^e|); w * a = w_l a_h + w_h a_l + w_h a_h, three
// v_mfma_f32_16x16x32_f16 instead of sixteen f32 4x4x1 MFMA slots per 16 x 16 x 32 block; fp32 accumulation.
//
// Layouts.  A wave owns 64 detections x a range of tracks (as before).  Per track:
//  (1) four sub-steps of 16 detections in the MFMA layout lane = (pair p = lane & 15, k block kb = lane >> 4): the lane forms
//      h1[32 s + 8 kb + j], j < 8, for the four 32-wide k steps s (fuse_shape | res_coeff lower half | upper half | fuse_det) from
//      8 UP values (LDS broadcast, read once per track) and 8 UC values (LDS) each - ReLU and range scaling are one packed
//      v_pk_fma_f32 with clamp per two values (fma2_relu01) -, multiplies by 2^14 (packed), converts (v_cvt_pk_f16_f32), takes the
//      residual (v_fma_mix_f32: x - h with h as an f16 operand, exact) and converts it: 2.5 issue slots of 5 - 8 cycles per value;
//      12 MFMAs; the three result blocks (4 consecutive output features of pair p per lane) go to a wave-private LDS tile
//      [64 pairs][40 features].
//  (2) lane = pair: the lane reads its 40 layer-2 pre-activations, descales (exact power of two) and adds the bias with one
//      fma each, and runs layers 3-4, the hand-designed residual and the combine exactly as pair_mfma4_kernel does.
// Range: the scale 2^e of a track is the one that puts (max |UP[t]| + max over the tile's detections of max |UC[d]|) - an upper
// bound of every h1 of the sub-steps - into (2^13, 2^14]; the row maxima come from embed_rows / row_prep (slot 13 of the hand rows).  The
// second-layer weights are cut once at pack time with one exponent per MLP (pair_f16_pack_kernel).
#include "common.hpp"
#include "pair_layout.hpp"

#include <type_traits>

namespace shasta {

typedef _Float16 ph16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 ph16x2 __attribute__((ext_vector_type(2)));
typedef uint32_t pu4 __attribute__((ext_vector_type(4)));

#define MFMA4(a, b, c) __builtin_amdgcn_mfma_f32_4x4x1f32((a), (b), (c), 0, 0, 0)
#define MFMA16H(a, b, c) \
    __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(ph16x8, (a)), __builtin_bit_cast(ph16x8, (b)), (c), 0, 0, 0)

template <int F, int L>
struct A4h {
    static constexpr LayerDesc D = layer_desc(F, L);
    static constexpr int NOB = a4_nob(F, L), KG = a4_kg(F, L), OFF = a4_offset(F, L), KIN = D.kin, BIAS = NOB * KG * 16;
};

// HAZARD RULE for the three asm helpers below: the compiler inserts the wait states a VALU result needs before an MFMA or a
// half-register reader consumes it only for its OWN instructions, not around inline asm.  Every result of these helpers must
// therefore pass through a compiler-generated VALU instruction (here: the packed multiply, v_cvt_pk_f16_f32) before it reaches an
// MFMA operand; feeding one straight into an MFMA needs an explicit "s_nop 1" (the clamp-fma tried in pair_mfma4).
// {clamp01(a0 * c + b0), clamp01(a1 * c + b1)}: with a, b pre-scaled so that every sum is at most 1, the clamp IS the ReLU
typedef float pf2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ pf2 fma2_relu01(pf2 a, pf2 c, pf2 b) {
    pf2 r;
    asm("v_pk_fma_f32 %0, %1, %2, %3 clamp" : "=v"(r) : "v"(a), "v"(c), "v"(b));
    return r;
}
// x - h (exact in fp32) with h = the low / high half of a packed f16 pair read as an f16 operand
__device__ __forceinline__ float res_lo(float x, uint32_t hpk) {
    float r;
    asm("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel_hi:[0,0,1]" : "=v"(r) : "v"(x), "v"(hpk));
    return r;
}
__device__ __forceinline__ float res_hi(float x, uint32_t hpk) {
    float r;
    asm("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(r) : "v"(x), "v"(hpk));
    return r;
}
__device__ __forceinline__ uint32_t cvt2(float a, float b) {
    const ph16x2 v = {(_Float16)a, (_Float16)b};
    return __builtin_bit_cast(uint32_t, v);
}

// ---- pack: the three second layers as A operands of v_mfma_f32_16x16x32_f16 -----------------------------------------------
// fragments (1 KB each = [64 lanes][8 fp16]): 0 fuse_shape.2 (16 x 32) | 1, 2 res_coeff.2 (16 x 64, two k steps) | 3 fuse_det.2
// (8 x 32, rows 8..15 zero); lane (i = lane & 15, kb = lane >> 4) holds W[i][32 ks + 8 kb + j] * 2^e_mlp; high pieces then low pieces.
// layout (dwords): [piece 2][fragment 4][lane 64][4], then 3 int exponents (fs, rc, fd), padded to 4.
constexpr int P16_FRAG_DW = 2 * 4 * 64 * 4;
constexpr int P16_DW = P16_FRAG_DW + 4;
size_t pair_f16_packed_floats() { return P16_DW; }

struct PairF16PackArgs {
    const float* w_fs2;  // fuse_shape.2.weight (16, 32)
    const float* w_rc2;  // res_coeff.2.weight (16, 64)
    const float* w_fd2;  // fuse_det.2.weight (8, 32)
    uint32_t* out;
};

__global__ __launch_bounds__(256) void pair_f16_pack_kernel(PairF16PackArgs a) {
    __shared__ float red[3][4];
    __shared__ int ex[3];
    const int tid = threadIdx.x;
    const float* W[3] = {a.w_fs2, a.w_rc2, a.w_fd2};
    const int cnt[3] = {16 * 32, 16 * 64, 8 * 32};
    for (int m = 0; m < 3; ++m) {
        float mx = 0.0f;
        for (int i = tid; i < cnt[m]; i += 256) mx = fmaxf(mx, fabsf(W[m][i]));
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
        if ((tid & 63) == 0) red[m][tid >> 6] = mx;
    }
    __syncthreads();
    if (tid < 3) {
        const float mx = fmaxf(fmaxf(red[tid][0], red[tid][1]), fmaxf(red[tid][2], red[tid][3]));
        ex[tid] = range_exponent_bits(__float_as_uint(mx));
        reinterpret_cast<int*>(a.out)[P16_FRAG_DW + tid] = ex[tid];
    }
    if (tid == 3) a.out[P16_FRAG_DW + 3] = 0;
    __syncthreads();
    const int lane = tid & 63, frag = tid >> 6;  // 4 fragments, one wave each
    const int i = lane & 15, kb = lane >> 4;
    const int mlp = frag == 0 ? 0 : frag == 3 ? 2 : 1;
    const int rows = mlp == 2 ? 8 : 16, kin = mlp == 1 ? 64 : 32, ks = frag == 2 ? 1 : 0;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = i < rows ? __builtin_ldexpf(W[mlp][i * kin + 32 * ks + 8 * kb + j], ex[mlp]) : 0.0f;
    pu4 hi, lo;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const _Float16 h0 = (_Float16)v[2 * j], h1 = (_Float16)v[2 * j + 1];
        const ph16x2 hh = {h0, h1};
        hi[j] = __builtin_bit_cast(uint32_t, hh);
        lo[j] = cvt2(v[2 * j] - (float)h0, v[2 * j + 1] - (float)h1);
    }
    reinterpret_cast<pu4*>(a.out)[(0 * 4 + frag) * 64 + lane] = hi;
    reinterpret_cast<pu4*>(a.out)[(1 * 4 + frag) * 64 + lane] = lo;
}

int pair_f16_pack(const shasta_weights* w, float* out, hipStream_t st) {
    PairF16PackArgs a;
    a.w_fs2 = w->fuse_shape[1].weight;
    a.w_rc2 = w->res_coeff[1].weight;
    a.w_fd2 = w->fuse_det[1].weight;
    a.out = reinterpret_cast<uint32_t*>(out);
    hipLaunchKernelGGL(pair_f16_pack_kernel, dim3(1), dim3(256), 0, st, a);
    return check_launch("pair_f16_pack");
}

// ---- the kernel (F = 256: H1 = 32, R1 = 64, fuse_det 32; H2 = 16, R2 = 16, 8) ---------------------------------------------
constexpr int PF_TS = 44;  // floats per pair in the transposition tile: 40 + 4 pad (conflict-free b128 reads at stride 44)

// GRID (SHASTA_OPT_F16GRID_PAIR, opt-in): the pieces of h1 are not cut per pair.  UP[t] and UC[d] are cut ONCE per row into two
// fp16 pieces on a common fixed grid per MLP - high piece an integer, low piece a multiple of 2^-11, after scaling the largest
// possible |UP| + |UC| of the (workgroup's tracks, detection tile) to at most 2^11 -, so that per pair the pieces of UP + UC are
// two exact packed adds and its ReLU two packed maxima (h' = max(h, -1), l' = max(l, -h'): for h >= 1 both pieces stay, for h = 0
// the low piece is clamped at 0, for h <= -1 the two cancel): 256 packed fp16 instructions per track and wave instead of 384
// conversions / mixes / packed fp32 ops.  The price is accuracy: a fixed grid spends its 22 bits on the LARGEST sum of the tile
// (tools/pair_quant_sim.py: max / rms error of `residual` 2x / 5x the fp32 kernels', 1e-6 of its range), which is why this is
// not the default arithmetic.
#ifdef PAIR_STAMP  // diagnostic build only (tools/pair_clock.py): in-kernel clock = d(s_memtime) / d(s_memrealtime) * 100 MHz per workgroup,
// and where / when each of its waves ran: per wave {shader cycles, s_memrealtime at its start, at its end, XCC_ID << 32 | HW_ID}
__device__ unsigned long long g_pair_stamp[4096][8][4];
#endif

template <int WPB, bool GRID>
__global__ __launch_bounds__(64 * WPB) void pair_f16_kernel(const float* __restrict__ packed, const uint32_t* __restrict__ p16,
                                                       const float* __restrict__ UP, const float* __restrict__ UC,
                                                       const float* __restrict__ hand_prev, const float* __restrict__ hand_det,
                                                       const float* __restrict__ denom, float* __restrict__ residual, int T,
                                                       int D, int ld, int nf, int TWG, int TA, int TB) {
    // TWG tracks per workgroup: TA to each of the waves 0 .. 3, TB to each of the waves 4 .. 7 (launch_pair_f16: the two waves of a
    // SIMD do not advance at the same rate)
    constexpr int F = 256;
    constexpr PairDims dm(F);
    constexpr int ET = dm.ET, US = ET + 4;
    static_assert(dm.H1 == 32 && dm.R1 == 64 && ET == 128 && dm.H2 == 16 && dm.R2 == 16, "pair_f16_kernel is laid out for F = 256");
    constexpr int NA4 = a4_total(F);
    extern __shared__ __attribute__((aligned(16))) float s_dynh[];
    float* s_uc = s_dynh;                                  // [64][US]
    float* s_a4 = s_dynh + 64 * US;                        // [NA4] 4x4x1 operand table (layers 3-4 and the layer-2 biases)
    float* s_up = s_a4 + ((NA4 + 3) & ~3);                 // [WPB][3 slots][256 floats]
    float* s_tr = s_up + WPB * 3 * 256;                    // [WPB][64 pairs][PF_TS]
    // GRID: the tile holds fp16 pieces instead: [64][GR_ROW] high pieces, then the same of low pieces (in place of s_uc, 1 KB more:
    // everything behind it moves by GR_EXTRA floats), and every wave has a row of UP pieces: [WPB][2][128 fp16] behind the tiles
    constexpr int GR_ROW = 136;                            // fp16 per tile row: 128 + 8 pad (272 B)
    constexpr int GR_EXTRA = GRID ? (2 * 64 * GR_ROW / 2 - 64 * US) : 0;
    if constexpr (GRID) {
        s_a4 += GR_EXTRA;
        s_up += GR_EXTRA;
        s_tr += GR_EXTRA;
    }
    // (all piece storage is written and read as 32-bit words = fp16 pairs: one access type, no type punning through memory)
    uint32_t* s_uch = reinterpret_cast<uint32_t*>(s_dynh);            // [64][GR_ROW / 2]
    uint32_t* s_ucl = s_uch + 64 * (GR_ROW / 2);
    uint32_t* s_upp = reinterpret_cast<uint32_t*>(s_tr + WPB * 64 * PF_TS);  // [WPB][2][64]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef PAIR_STAMP
    const unsigned long long st0 = __builtin_amdgcn_s_memtime(), sr0 = __builtin_amdgcn_s_memrealtime();
#endif
    // (Placing the 8 detection tiles of a frame-pair on ONE XCD - they all read that frame-pair's UP / hand tables, 265 KB, which with
    // the plain order is fetched into eight L2s - was measured: 1.5 % less energy per step, pair kernel 4.21 - 4.30 -> 4.35 ms; not kept.)
#ifdef PAIR_NO_XCD_ORDER
    const int b = blockIdx.z, d0 = blockIdx.x * 64;
    const int by = blockIdx.y;
#else
    int lbx, by, b;
    xcd_logical_block(lbx, by, b);  // the detection tiles of a frame on one XCD: its UP / UC rows come from HBM once
    const int d0 = lbx * 64;
#endif
    const int d = d0 + lane, dcl = min(d, D - 1);
    const PackedLayout P(0, 0, F);
    {
        if constexpr (!GRID) {
            const f32x4* src = reinterpret_cast<const f32x4*>(UC);
#pragma unroll 4
            for (int e = tid; e < 64 * (ET / 4); e += 64 * WPB) {
                const int r = e / (ET / 4), c = e - r * (ET / 4);
                // float4 c of the row = k step c >> 3, k block (c >> 1) & 3, half c & 1: stored k-block-major, the blocks in the order
                // 0, 2, 1, 3 (see load_uc: blocks 0 / 1 and 2 / 3 must sit 64 floats apart)
                const int cp = (((c >> 1) & 1) << 4) | (((c >> 2) & 1) << 3) | ((c >> 3) << 1) | (c & 1);
                *reinterpret_cast<f32x4*>(&s_uc[r * US + 4 * cp]) = src[((size_t)b * D + min(d0 + r, D - 1)) * (ET / 4) + c];
            }
        }
        const f32x4* asrc = reinterpret_cast<const f32x4*>(packed + P.a4);
#pragma unroll 2
        for (int e = tid; e < NA4 / 4; e += 64 * WPB) reinterpret_cast<f32x4*>(s_a4)[e] = asrc[e];
    }
    float hd[12];
    float mc;  // largest |UC| of this lane's detection row (row_prep), then of the whole 64-detection tile
    {
        const f32x4* h = reinterpret_cast<const f32x4*>(hand_det + ((size_t)b * D + dcl) * 16);
        const f32x4 a = h[0], c = h[1], e = h[2], g = h[3];
        hd[0] = a[0]; hd[1] = a[1]; hd[2] = a[2]; hd[3] = a[3]; hd[4] = c[0]; hd[5] = c[1]; hd[6] = c[2];
        hd[7] = e[0]; hd[8] = e[1]; hd[9] = e[2]; hd[10] = e[3]; hd[11] = g[0];
        mc = g[1];
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mc = absmax_keep_nan(mc, __shfl_xor(mc, off, 64));  // a NaN / inf row maximum survives
    const float dnm = denom[(size_t)b * D + dcl], rdn = 1.0f / dnm;
    // GRID: one grid per MLP for all tracks of this workgroup and the detections of this tile.  Largest magnitudes per column range
    // (hand slots: 14 fuse_shape, 15 res_coeff, 13 = all columns, the bound used for fuse_det) of the tile's UC rows and of the
    // workgroup's UP rows; ge[r] = the exponent that puts their sum into (2^10, 2^11].
    int ge[3] = {0, 0, 0};
    bool grid_finite = true;
    if constexpr (GRID) {
        float mcr[3], mur[3] = {0.0f, 0.0f, 0.0f};
        {
            const float* h = hand_det + ((size_t)b * D + dcl) * 16;
            mcr[0] = h[14];
            mcr[1] = h[15];
            mcr[2] = h[13];
        }
        const int tw0 = by * TWG, tw1 = min(T, tw0 + TWG);
        for (int t = tw0 + tid; t < tw1; t += 64 * WPB) {
            const float* h = hand_prev + ((size_t)b * T + t) * 16;
            mur[0] = absmax_keep_nan(mur[0], h[14]);
            mur[1] = absmax_keep_nan(mur[1], h[15]);
            mur[2] = absmax_keep_nan(mur[2], h[13]);
        }
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                mcr[r] = absmax_keep_nan(mcr[r], __shfl_xor(mcr[r], off, 64));
                mur[r] = absmax_keep_nan(mur[r], __shfl_xor(mur[r], off, 64));
            }
        float* red = s_tr;  // scratch: the transposition tiles are not in use yet
        if (lane == 0) {
            red[wid * 3 + 0] = mur[0];
            red[wid * 3 + 1] = mur[1];
            red[wid * 3 + 2] = mur[2];
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            float m = red[r];
            for (int w = 1; w < WPB; ++w) m = absmax_keep_nan(m, red[w * 3 + r]);
            // (+ 0.2 %: the two high pieces are rounded to integers separately, their sum must stay below 2048, where fp16 still
            // holds every integer)
            const float bound = (m + mcr[r]) * 1.002f;
            grid_finite = grid_finite && (bound < INFINITY);
            ge[r] = range_exponent_bits(__float_as_uint(bound)) - 3;
        }
        __syncthreads();  // red is read by everyone before the tiles are written
        // the tile: UC rows cut on the grids
        const f32x4* src = reinterpret_cast<const f32x4*>(UC);
        for (int e = tid; e < 64 * (ET / 4); e += 64 * WPB) {
            const int r = e / (ET / 4), c = e - r * (ET / 4);  // float4 c = columns 4 c .. 4 c + 3: one MLP range
            const f32x4 v = src[((size_t)b * D + min(d0 + r, D - 1)) * (ET / 4) + c];
            const int ex = c < 8 ? ge[0] : c < 24 ? ge[1] : ge[2];
            _Float16 hh[4], ll[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float sv = __builtin_ldexpf(v[q], ex), hi = rintf(sv);
                hh[q] = (_Float16)hi;
                ll[q] = (_Float16)(rintf((sv - hi) * 2048.0f) * (1.0f / 2048.0f));
            }
            typedef uint32_t pu2 __attribute__((ext_vector_type(2)));
            typedef _Float16 ph2t __attribute__((ext_vector_type(2)));
            *reinterpret_cast<pu2*>(s_uch + r * (GR_ROW / 2) + 2 * c) =
                pu2{__builtin_bit_cast(uint32_t, ph2t{hh[0], hh[1]}), __builtin_bit_cast(uint32_t, ph2t{hh[2], hh[3]})};
            *reinterpret_cast<pu2*>(s_ucl + r * (GR_ROW / 2) + 2 * c) =
                pu2{__builtin_bit_cast(uint32_t, ph2t{ll[0], ll[1]}), __builtin_bit_cast(uint32_t, ph2t{ll[2], ll[3]})};
        }
    }
    // second-layer weight pieces: 4 fragments x {high, low}, registers for the whole kernel
    pu4 wh[4], wl[4];
#pragma unroll
    for (int f = 0; f < 4; ++f) {
        wh[f] = reinterpret_cast<const pu4*>(p16)[(0 * 4 + f) * 64 + lane];
        wl[f] = reinterpret_cast<const pu4*>(p16)[(1 * 4 + f) * 64 + lane];
    }
    const int ew_fs = reinterpret_cast<const int*>(p16)[P16_FRAG_DW + 0], ew_rc = reinterpret_cast<const int*>(p16)[P16_FRAG_DW + 1],
              ew_fd = reinterpret_cast<const int*>(p16)[P16_FRAG_DW + 2];
    __syncthreads();
    typedef __attribute__((address_space(3))) float lfloat;
    typedef __attribute__((address_space(3))) f32x4 lf32x4;
    const unsigned arow_base = (unsigned)(unsigned long long)(s_a4 + (lane & 3) * 4);
    const unsigned abias_base = (unsigned)(unsigned long long)(s_a4 + (lane & 3));
    const f32x4 zero4 = {0, 0, 0, 0};
    const int p = lane & 15, kb = lane >> 4;
    float* my_tr = s_tr + wid * (64 * PF_TS);

    const int t_beg = by * TWG + (wid < 4 ? wid * TA : 4 * TA + (wid - 4) * TB);
    const int t_end = min(min(T, (by + 1) * TWG), t_beg + (wid < 4 ? TA : TB));
    float* my_up = s_up + wid * (3 * 256);
    const bool hp_lane = lane >= ET / 4 && lane < ET / 4 + 4;
    const int up_lane = 4 * min(lane, ET / 4 - 1), hp_off = 4 * (lane - ET / 4);
    auto dma_up = [&](int row, int slot) {
        const size_t r = (size_t)b * T + min(row, T - 1);
        const float* src = hp_lane ? hand_prev + r * 16 + hp_off : UP + r * ET + up_lane;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(my_up + slot * 256), 16, 0, 0);
    };
    if (t_beg < t_end) {
        dma_up(t_beg, 0);
        dma_up(t_beg + 1, 1);
    }
    for (int t = t_beg; t < t_end; ++t) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        dma_up(t + 2, (t - t_beg + 2) % 3);
        unsigned upo = (unsigned)(unsigned long long)(my_up + ((t - t_beg) % 3) * 256);
        asm volatile("" : "+v"(upo));
        const lfloat* up = (const lfloat*)(unsigned long long)upo;
        float hp[16];
        {
            const f32x4 h0 = *reinterpret_cast<const lf32x4*>(up + ET), h1 = *reinterpret_cast<const lf32x4*>(up + ET + 4),
                        h2 = *reinterpret_cast<const lf32x4*>(up + ET + 8), h3 = *reinterpret_cast<const lf32x4*>(up + ET + 12);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                hp[k] = h0[k];
                hp[4 + k] = h1[k];
                hp[8 + k] = h2[k];
                hp[12 + k] = h3[k];
            }
        }
        typedef _Float16 ph2 __attribute__((ext_vector_type(2)));
        bool finite_bound;
        int e1 = 0;
        pf2 cs2 = {0.0f, 0.0f};
        const pf2 c14 = {16384.0f, 16384.0f};
        pf2 upv[GRID ? 1 : 16];   // !GRID: this lane's UP values, scaled
        pu4 uph[GRID ? 4 : 1], upl[GRID ? 4 : 1];  // GRID: this lane's UP pieces, 8 per k step
        if constexpr (!GRID) {
            // the track's scale: every h1 of this track and tile is at most max |UP[t]| + max |UC|
            // Non-finite embeddings: the clamp of the packed fma below turns a NaN into 0 and saturates an infinity at 1, so they would
            // come out as finite residuals; the row maxima keep them (absmax_keep_nan), and a track whose bound is not finite gets NaN
            // for the whole tile - more NaNs than the reference's element-wise propagation, never a finite number in their place.
            const float bound = hp[13] + mc;
            finite_bound = bound < INFINITY;  // false for NaN and +inf
            e1 = range_exponent_bits(__float_as_uint(bound));
            // h1 = relu(UP + UC) is formed as clamp01(UC cs + UP cs) with cs = 2^(e1 - 14): every sum is at most 1 after the scaling
            // (exact, a power of two), so the clamp of one packed fma is the ReLU of two values; the conversion multiplies by 2^14.
            const float cs = __builtin_ldexpf(1.0f, e1 - 14);
            cs2 = pf2{cs, cs};
            // this lane's UP values, scaled: 8 per k step (the same address in the 16 lanes of a k block: LDS broadcast).  (Scaling the row
            // ONCE in place in LDS - 2 packed multiplies per track instead of 16, as pair_f16w.hip does - measured slower here: 4.06 - 4.13 ->
            // 4.18 - 4.21 ms at 512 frame-pairs: the write-read round trip through LDS sits on this kernel's critical path.)
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const f32x4 a = *reinterpret_cast<const lf32x4*>(up + 32 * s + 8 * kb), c = *reinterpret_cast<const lf32x4*>(up + 32 * s + 8 * kb + 4);
                upv[4 * s] = pf2{a[0], a[1]} * cs2;
                upv[4 * s + 1] = pf2{a[2], a[3]} * cs2;
                upv[4 * s + 2] = pf2{c[0], c[1]} * cs2;
                upv[4 * s + 3] = pf2{c[2], c[3]} * cs2;
            }
        } else {
            finite_bound = grid_finite;
            // the track's UP row cut on the grids: lane L cuts columns 2 L, 2 L + 1 (L < 16: fuse_shape, < 48: res_coeff, else fuse_det)
            // into the wave's piece rows; every lane then reads its 8 values per k step back (LDS broadcast within a k block)
            uint32_t* my_p = s_upp + wid * 128;  // [high 64 words | low 64 words]
            {
                typedef __attribute__((address_space(3))) pf2 lpf2;
                const pf2 v = *reinterpret_cast<const lpf2*>(up + 2 * lane);
                const int ex = lane < 16 ? ge[0] : lane < 48 ? ge[1] : ge[2];
                const float s0 = __builtin_ldexpf(v[0], ex), s1 = __builtin_ldexpf(v[1], ex);
                const float h0 = rintf(s0), h1v = rintf(s1);
                const ph2 hh = {(_Float16)h0, (_Float16)h1v};
                const ph2 ll = {(_Float16)(rintf((s0 - h0) * 2048.0f) * (1.0f / 2048.0f)), (_Float16)(rintf((s1 - h1v) * 2048.0f) * (1.0f / 2048.0f))};
                my_p[lane] = __builtin_bit_cast(uint32_t, hh);
                my_p[64 + lane] = __builtin_bit_cast(uint32_t, ll);
            }
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                uph[s] = *reinterpret_cast<const pu4*>(my_p + 16 * s + 4 * kb);
                upl[s] = *reinterpret_cast<const pu4*>(my_p + 64 + 16 * s + 4 * kb);
            }
        }
        // Software pipeline over the four sub-steps: the UC reads of sub-step s+1 are issued before the arithmetic of sub-step s,
        // and the pieces of sub-step s+1 are cut before the MFMAs of sub-step s are issued (they run under the cut of s+1).
        auto load_uc = [&](int sub, f32x4 (&u)[8]) {
            if constexpr (GRID) {
                // u[0..3] = the high pieces of the four k steps (8 fp16 each), u[4..7] = the low pieces
                const uint32_t* hr = s_uch + (16 * sub + p) * (GR_ROW / 2) + 4 * kb;
                const uint32_t* lr = s_ucl + (16 * sub + p) * (GR_ROW / 2) + 4 * kb;
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    u[s] = __builtin_bit_cast(f32x4, *reinterpret_cast<const pu4*>(hr + 16 * s));
                    u[4 + s] = __builtin_bit_cast(f32x4, *reinterpret_cast<const pu4*>(lr + 16 * s));
                }
                return;
            }
            // A ds_read_b128 serves lanes {0-3, 12-15, 20-23, 24-27}, {4-11, 16-19, 28-31} (and the same + 32) together - not 16
            // consecutive lanes (tools/probes/lds_pattern_probe.hip).  In this layout lane = (p, kb) a group mixes rows p of k blocks
            // kb and kb + 1; with the four k blocks of a step 8 floats apart those fell on each other's banks (every read took 8
            // cycles instead of 4, the kernel's LDS busy 57 % of the time).  The tile therefore keeps a row k-block-major with the
            // blocks of a pair 64 floats = one bank round apart: 16 rows x 2 blocks then cover the 64 banks exactly once.
            const float* ucr = s_uc + (16 * sub + p) * US + ((kb & 1) << 6) + ((kb >> 1) << 5);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                u[2 * s] = *reinterpret_cast<const f32x4*>(ucr + 8 * s);
                u[2 * s + 1] = *reinterpret_cast<const f32x4*>(ucr + 8 * s + 4);
            }
        };
        auto cut = [&](const f32x4 (&u)[8], pu4 (&xh)[4], pu4 (&xl)[4]) {
            if constexpr (GRID) {
                // pieces of relu(UP + UC): exact packed adds, then h' = max(h, -1), l' = max(l, -h'); on whole 8-halves vectors (the
                // compiler lowers them to v_pk_add_f16 / v_pk_max_f16 with the negation as an operand modifier; written dword by dword
                // through bit casts of vector elements hipcc 7.2 computed element 0 only and replicated it)
                const _Float16 m1 = (_Float16)-1.0f;
                const ph16x8 neg1 = {m1, m1, m1, m1, m1, m1, m1, m1};
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const ph16x8 h = __builtin_bit_cast(ph16x8, u[s]) + __builtin_bit_cast(ph16x8, uph[s]);
                    const ph16x8 l = __builtin_bit_cast(ph16x8, u[4 + s]) + __builtin_bit_cast(ph16x8, upl[s]);
                    const ph16x8 h2 = __builtin_elementwise_max(h, neg1);
                    const ph16x8 l2 = __builtin_elementwise_max(l, -h2);
                    xh[s] = __builtin_bit_cast(pu4, h2);
                    xl[s] = __builtin_bit_cast(pu4, l2);
                }
                return;
            }
#pragma unroll
            for (int s = 0; s < 4; ++s) {
#pragma unroll
                for (int j2 = 0; j2 < 2; ++j2) {  // the float4 u[2 s + j2] = four values
                    const f32x4 uu = u[2 * s + j2];
                    // scaled to [0, 2^14]: high piece = the packed conversion (round to nearest even), residual exact, low piece =
                    // its conversion.  Only full-register writes: the v_fma_mixlo / mixhi_f16 pair this replaces cost 17.8 cycles per
                    // two values against 13.4 for multiply + convert (tools/probes/valu_cost_probe.hip) and needed a hand-placed wait
                    // state between a half-register write and its reader that the compiler does not insert around inline asm.
                    const pf2 sa = fma2_relu01(pf2{uu[0], uu[1]}, cs2, upv[4 * s + 2 * j2]) * c14;
                    const pf2 sb = fma2_relu01(pf2{uu[2], uu[3]}, cs2, upv[4 * s + 2 * j2 + 1]) * c14;
                    const uint32_t hA = cvt2(sa[0], sa[1]), hB = cvt2(sb[0], sb[1]);
                    xh[s][2 * j2] = hA;
                    xh[s][2 * j2 + 1] = hB;
                    xl[s][2 * j2] = cvt2(res_lo(sa[0], hA), res_hi(sa[1], hA));
                    xl[s][2 * j2 + 1] = cvt2(res_lo(sb[0], hB), res_hi(sb[1], hB));
                }
            }
        };
        auto mma = [&](const pu4 (&xh)[4], const pu4 (&xl)[4], f32x4& a_fs, f32x4& a_rc, f32x4& a_fd) {
            a_fs = a_rc = a_fd = zero4;
            a_fs = MFMA16H(wl[0], xh[0], a_fs);
            a_rc = MFMA16H(wl[1], xh[1], a_rc);
            a_fd = MFMA16H(wl[3], xh[3], a_fd);
            a_fs = MFMA16H(wh[0], xl[0], a_fs);
            a_rc = MFMA16H(wh[1], xl[1], a_rc);
            a_fd = MFMA16H(wh[3], xl[3], a_fd);
            a_rc = MFMA16H(wl[2], xh[2], a_rc);
            a_fs = MFMA16H(wh[0], xh[0], a_fs);
            a_fd = MFMA16H(wh[3], xh[3], a_fd);
            a_rc = MFMA16H(wh[2], xl[2], a_rc);
            a_rc = MFMA16H(wh[1], xh[1], a_rc);
            a_rc = MFMA16H(wh[2], xh[2], a_rc);
        };
        // (storing a sub-step's results behind the NEXT sub-step's MFMAs instead of right behind their own - where each store waits
        // 5 - 7 idle states for its accumulator - measured slower: 4.20 -> 4.30 ms; the compiler fills those states elsewhere)
        auto store = [&](int sub, const f32x4& a_fs, const f32x4& a_rc, const f32x4& a_fd) {
            // lane (p, kb) holds output features 4 kb .. 4 kb + 3 of pair 16 sub + p: [rc 16 | fs 16 | fd 8]
            float* row = my_tr + (16 * sub + p) * PF_TS + 4 * kb;
            *reinterpret_cast<f32x4*>(row) = a_rc;
            *reinterpret_cast<f32x4*>(row + 16) = a_fs;
            if (kb < 2) *reinterpret_cast<f32x4*>(row + 32) = a_fd;
        };
        {
            f32x4 ua[8], ub[8];
            pu4 xha[4], xla[4], xhb[4], xlb[4];
            f32x4 fsA, rcA, fdA, fsB, rcB, fdB;
            load_uc(0, ua);
            load_uc(1, ub);
            cut(ua, xha, xla);
            load_uc(2, ua);
            cut(ub, xhb, xlb);
            mma(xha, xla, fsA, rcA, fdA);
            store(0, fsA, rcA, fdA);
            load_uc(3, ub);
            cut(ua, xha, xla);
            mma(xhb, xlb, fsB, rcB, fdB);
            store(1, fsB, rcB, fdB);
            cut(ub, xhb, xlb);
            mma(xha, xla, fsA, rcA, fdA);
            store(2, fsA, rcA, fdA);
            mma(xhb, xlb, fsB, rcB, fdB);
            store(3, fsB, rcB, fdB);
        }
        // ---- lane = pair from here on: descale + bias, layers 3-4, hand residual, combine (as pair_mfma4_kernel) ----
        unsigned ao = arow_base, bo = abias_base;
        asm volatile("" : "+v"(ao), "+v"(bo));
        const lfloat* arow = (const lfloat*)(unsigned long long)ao;
        const lfloat* abias = (const lfloat*)(unsigned long long)bo;
        (void)abias;
        // exact descaling: one exponent per track (default form) or one per MLP and workgroup (GRID)
        const float i_rc = __builtin_ldexpf(1.0f, -((GRID ? ge[1] : e1) + ew_rc)), i_fs = __builtin_ldexpf(1.0f, -((GRID ? ge[0] : e1) + ew_fs)),
                    i_fd = __builtin_ldexpf(1.0f, -((GRID ? ge[2] : e1) + ew_fd));
        f32x4 a_rc2[4], a_fs2[4], a_fd2[2];
        {
            const float* mine = my_tr + lane * PF_TS;
            // the layer-2 biases sit behind the 4x4x1 operands of their layers in the LDS table ([ob][i], A4h<..>::BIAS)
            const float* b_rc = s_a4 + A4h<F, L_RC2>::OFF + A4h<F, L_RC2>::BIAS;
            const float* b_fs = s_a4 + A4h<F, L_FS2>::OFF + A4h<F, L_FS2>::BIAS;
            const float* b_fd = s_a4 + A4h<F, L_FD2>::OFF + A4h<F, L_FD2>::BIAS;
            // descale + bias as packed fmas (two values per 5-cycle slot instead of one per 6), ReLU right behind them
            auto fma4 = [&](const f32x4& v, float sc, const f32x4& bb) {
                const pf2 s2 = {sc, sc};
                const pf2 lo = __builtin_elementwise_fma(pf2{v[0], v[1]}, s2, pf2{bb[0], bb[1]});
                const pf2 hi = __builtin_elementwise_fma(pf2{v[2], v[3]}, s2, pf2{bb[2], bb[3]});
                // (fmaxf, not the NaN-propagating relu_nan of the other kernels: non-finite inputs never get here - finite_bound above -
                // and v_maximum3_f32 in this loop measured 1 - 3 % of the kernel: 4.39 - 4.43 -> 4.45 - 4.56 ms)
                return f32x4{fmaxf(lo[0], 0.0f), fmaxf(lo[1], 0.0f), fmaxf(hi[0], 0.0f), fmaxf(hi[1], 0.0f)};
            };
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                a_rc2[g] = fma4(*reinterpret_cast<const f32x4*>(mine + 4 * g), i_rc, *reinterpret_cast<const f32x4*>(b_rc + 4 * g));
                a_fs2[g] = fma4(*reinterpret_cast<const f32x4*>(mine + 16 + 4 * g), i_fs, *reinterpret_cast<const f32x4*>(b_fs + 4 * g));
            }
#pragma unroll
            for (int g = 0; g < 2; ++g)
                a_fd2[g] = fma4(*reinterpret_cast<const f32x4*>(mine + 32 + 4 * g), i_fd, *reinterpret_cast<const f32x4*>(b_fd + 4 * g));
        }
        // All forty ReLUs of the layer-2 outputs are done before the first MFMA of layers 3-4 (the empty asm pins them there): a VALU
        // result read by the MFMA right behind it costs two wait states, and the compiler had put one v_max + s_nop 1 in front of
        // every 4x4x1.  With the third layers interleaved below: 53 -> 36 s_nop per track, pair kernel 4.73 - 4.95 -> 4.56 - 4.62 ms.
        asm volatile("" : "+v"(a_rc2[0]), "+v"(a_rc2[1]), "+v"(a_rc2[2]), "+v"(a_rc2[3]), "+v"(a_fs2[0]), "+v"(a_fs2[1]), "+v"(a_fs2[2]),
                     "+v"(a_fs2[3]), "+v"(a_fd2[0]), "+v"(a_fd2[1]));
        auto init = [&](auto tag, f32x4* acc) {
            using AL = decltype(tag);
#pragma unroll
            for (int ob = 0; ob < AL::NOB; ++ob) acc[ob] = MFMA4(abias[AL::OFF + AL::BIAS + ob * 4], 1.0f, zero4);
        };
        auto layer = [&](auto tag, const f32x4* in, f32x4* acc, auto relu_done) {
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
                        for (int ob = 0; ob < AL::NOB; ++ob) acc[ob] = MFMA4(a4[ob][kk], h, acc[ob]);
                    }
                }
            }
        };
        f32x4 a_rc3[A4h<F, L_RC3>::NOB], a_fs3[A4h<F, L_FS3>::NOB], a_fs4[A4h<F, L_FS4>::NOB], a_fd3[A4h<F, L_FD3>::NOB];
        {
            // the three third layers side by side: every accumulator is touched once per round, so no 4x4x1 waits for the one before it
            using RC = A4h<F, L_RC3>;
            using FS = A4h<F, L_FS3>;
            using FD = A4h<F, L_FD3>;
            static_assert(RC::NOB == 1 && FS::NOB == 2 && FD::NOB == 1 && RC::KG == 4 && FS::KG == 4 && FD::KG == 2, "F = 256");
            init(RC{}, a_rc3);
            init(FS{}, a_fs3);
            init(FD{}, a_fd3);
#pragma unroll
            for (int kg = 0; kg < 4; ++kg) {
                const f32x4 w_rc = *reinterpret_cast<const lf32x4*>(arow + RC::OFF + kg * 16);
                const f32x4 w_f0 = *reinterpret_cast<const lf32x4*>(arow + FS::OFF + kg * 16);
                const f32x4 w_f1 = *reinterpret_cast<const lf32x4*>(arow + FS::OFF + (FS::KG + kg) * 16);
                f32x4 w_fd = zero4;
                if (kg < 2) w_fd = *reinterpret_cast<const lf32x4*>(arow + FD::OFF + kg * 16);
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    a_rc3[0] = MFMA4(w_rc[kk], a_rc2[kg][kk], a_rc3[0]);
                    a_fs3[0] = MFMA4(w_f0[kk], a_fs2[kg][kk], a_fs3[0]);
                    a_fs3[1] = MFMA4(w_f1[kk], a_fs2[kg][kk], a_fs3[1]);
                    if (kg < 2) a_fd3[0] = MFMA4(w_fd[kk], a_fd2[kg][kk], a_fd3[0]);
                }
            }
        }
        layer(A4h<F, L_FS4>{}, a_fs3, a_fs4, std::false_type{});

        // ---- hand-designed residual (shasta.py:277-283) ----
        const float dist = hand_dist(hp, hd, dnm, rdn);
        // ---- combine (shasta.py:316-319) ----
        const float res = (a_rc3[0][0] * a_fd3[0][0] + a_rc3[0][1] * dist) + a_rc3[0][2] * a_fs4[0][0];
        if (d < D) residual[((size_t)b * T + t) * ld + d] = finite_bound ? res : __builtin_nanf("");
    }
#ifdef PAIR_STAMP
    if (lane == 0) {
        const unsigned slot = ((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) & 4095;
        g_pair_stamp[slot][wid & 7][0] = __builtin_amdgcn_s_memtime() - st0;
        g_pair_stamp[slot][wid & 7][1] = sr0;
        g_pair_stamp[slot][wid & 7][2] = __builtin_amdgcn_s_memrealtime();
        g_pair_stamp[slot][wid & 7][3] = ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32) | __builtin_amdgcn_s_getreg((31 << 11) | 4);
    }
#endif
}

size_t pair_f16_lds_bytes(int wpb, bool grid) {
    constexpr PairDims dm(256);
    const size_t base = ((size_t)64 * (dm.ET + 4) + ((a4_total(256) + 3) & ~3) + (size_t)wpb * 3 * 256 + (size_t)wpb * 64 * PF_TS) * sizeof(float);
    // GRID: the fp16 piece tile is 1 KB larger than the fp32 tile, + 512 bytes of UP pieces per wave
    return base + (grid ? (size_t)(2 * 64 * 136 * 2 - 64 * (dm.ET + 4) * 4) + (size_t)wpb * 512 : 0);
}

int launch_pair_f16(const float* packed, const float* p16, const float* UP, const float* UC, const float* hand_prev,
                    const float* hand_det, const float* denom, float* residual, int B, int T, int D, int ld, int nf, bool grid,
                    hipStream_t st) {
#ifdef PAIR_WPB  // probe builds only: waves per workgroup = waves per CU (the LDS allows one workgroup per CU)
    constexpr int wpb = PAIR_WPB;
#else
    constexpr int wpb = 8;
#endif
    // tracks per workgroup: the T tracks dealt evenly to the ny workgroups of a detection tile, ny the smallest power of two that leaves
    // 512 workgroups (two rounds of the CU array).  A workgroup's prologue - the detection tile and the operand table into LDS behind a
    // barrier - is paid once per workgroup: 5.42 / 5.23 / 5.11 / 5.05 ms for 8 / 16 / 32 / 64 tracks per wave at 512 frame-pairs.
    // Tracks per WAVE: the LDS allows one workgroup per CU, so its waves w and w + 4 share a SIMD for the whole kernel, and the SIMD
    // issues from the older wave first: stamped per wave (tools/pair_clock.py), waves 0 .. 3 ran their tracks at the pace of a wave
    // that has the SIMD to itself (6 870 cycles per track) while waves 4 .. 7 advanced 0.625 tracks per track of theirs, then finished
    // alone - the slow way, one wave per SIMD - for the last quarter of the workgroup's time, with the CU's LDS held.  Dealing the
    // tracks 78 : 48 instead of 63 : 63 lets both finish together (within 7 us of 234): pair stage 4.41 -> 4.16 ms at 512 frame-pairs;
    // 55 : 100 and 68 : 100 measured 1 - 2 % behind 61.5 : 100.  (s_setprio by phase - the wave in its lane-per-pair phase first, or
    // last - with even or uneven tracks: at best equal, 7.71 against 7.65 ms at 1024 frame-pairs; not kept.)
    int ny = 1;
    while ((long)B * cdiv(D, 64) * ny < 512 && cdiv(T, wpb * ny * 2) >= 2) ny *= 2;
    const int twg = cdiv(cdiv(T, ny), wpb) * wpb;  // a multiple of the waves
#ifndef PAIR_SPLIT_PERMILLE
#define PAIR_SPLIT_PERMILLE 615  // tracks of a late wave per 1000 of an early one
#endif
    int ta = twg / wpb, tb = ta;
    if (wpb == 8) {
        const int per_simd = twg / 4;
        ta = (per_simd * 1000 + (1000 + PAIR_SPLIT_PERMILLE) / 2) / (1000 + PAIR_SPLIT_PERMILLE);  // to nearest
        tb = per_simd - ta;
    }
    const size_t lds = pair_f16_lds_bytes(wpb, grid);
    dim3 grd(cdiv(D, 64), cdiv(T, twg), B);
    if (grid) {
        (void)hipFuncSetAttribute((const void*)pair_f16_kernel<wpb, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((pair_f16_kernel<wpb, true>), grd, dim3(64 * wpb), lds, st, packed, reinterpret_cast<const uint32_t*>(p16), UP, UC,
                           hand_prev, hand_det, denom, residual, T, D, ld, nf, twg, ta, tb);
    } else {
        (void)hipFuncSetAttribute((const void*)pair_f16_kernel<wpb, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((pair_f16_kernel<wpb, false>), grd, dim3(64 * wpb), lds, st, packed, reinterpret_cast<const uint32_t*>(p16), UP, UC,
                           hand_prev, hand_det, denom, residual, T, D, ld, nf, twg, ta, tb);
    }
    return check_launch("pair_f16");
}

}  // namespace shasta

#ifdef PAIR_STAMP
extern "C" __attribute__((visibility("default"))) int shasta_debug_pair_stamp(unsigned long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(shasta::g_pair_stamp), sizeof(shasta::g_pair_stamp));
}
#endif
