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
//    fragment): an operand is one coalesced 16-byte load per lane and piece, no VALU.  A workgroup owns 128 residual rows = 4 row
//    blocks (64 = 2 row blocks while there are too few rows to fill the chip, ApShape) so that every fragment feeds 4 x 6 MFMAs.
//  * Layer 1 (K = N+2): residual rows and weight fragments reach LDS by LDS-DMA, each byte once per workgroup, through a ring of
//    32-column chunks laid over the piece images (see the kernel); wave = (row block, two feature blocks), the residual values
//    are cut in registers.
//  * Layers 2-5 (128 -> 64 -> 32 -> 64 -> 128) keep their activations in LDS as piece images [piece][row][k], alternating
//    between image A (128 wide, row stride 272 B) and image B (64 wide, 144 B): every 16-lane phase of a ds_read_b128 covers
//    all 64 banks; each value is cut once, by the producing wave.
//  * Layer 6 (128 -> N+2): wave = 2 feature blocks x 4 row blocks = 8 accumulators.  The row softmax statistics are taken
//    from the registers (per-wave partial max / sum through LDS, fixed order); the rows then pass through an fp32 staging
//    [rows][256] in two column passes and leave as whole row segments (16-byte stores for `matched`, 8-byte stores for
//    `matched1` = exp(x - max) / sum, four rows in flight per wave).
//  Where a workgroup's time goes (tools/probes/aff_probe.hip, 128 rows, shader cycles): layer 1 56 k (MFMA 25 k, LDS reads 16 k
//  and the cuts 15 k do not overlap: one barrier per chunk keeps the two waves of a SIMD in step), layers 2-5 31 k (latency of
//  their weight loads and four barriers), layer 6 30 k (MFMA-bound), softmax statistics 30 k, output 61 k (20 k without the stores).
#include "aff_frame.hpp"

// the LDS-DMA asm below names m0 in its clobber list on purpose (it writes it)
#pragma clang diagnostic ignored "-Winline-asm"

namespace shasta {

typedef __bf16 qbf16x8 __attribute__((ext_vector_type(8)));
#ifndef AP_PACE
#define AP_PACE 4  // s_sleep units between the MFMA groups of layer 1 (see there)
#endif

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
        for (int j = 0; j < 4; ++j) ap_cut3(relu_nan(acc[4 * g + j] + b[j]), h[j], m[j], l[j]);
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

// The six layers for the ROWS residual rows [g0, g0 + ROWS) (rows beyond glast, the last row this workgroup may read, repeat row
// glast): on return acc[i][rb] holds, WITHOUT the bias, features (wid + WAVES i) 32 + 8 (r >> 2) + 4 (lane >> 5) + (r & 3) of row
// 32 rb + (lane & 31), and every wave has passed the barrier behind layer 5 (image A is still being read by layer 6 of other waves).
template <int ROWS, int WAVES>
__device__ __forceinline__ void ap_mlp(const AffPiecesArgs& a, char* smem, int g0, int glast, int tid, int lane, int wid,
                                       f32x16 (&acc)[ApShape<ROWS, WAVES>::NFW][ApShape<ROWS, WAVES>::RB]) {
    using S = ApShape<ROWS, WAVES>;
    constexpr int RB = S::RB, NFW = S::NFW, AIMG = S::AIMG, BIMG = S::BIMG;
    char* HA = smem;              // [3][ROWS][272 B]: layer outputs of width 128 / 32
    char* HB = smem + S::ABYTES;  // [3][ROWS][144 B]: layer outputs of width 64
    const int D = a.D;
    const qu32x4* wp = reinterpret_cast<const qu32x4*>(a.wp);
    const f32x16 zero16 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    AP_STAMP(0);

    // ---- layer 1 (K = D -> 128): wave = (row block wid >> 1, feature blocks 2 (wid & 1) + {0, 1}).  Residual rows and weight
    // fragments reach LDS by LDS-DMA, each byte once per workgroup, in chunks of two k steps = 32 columns through a ring of NS
    // slots laid over the (still unused) piece images: x [ROWS][8 x 16 B], the 16-byte piece c of row r at position
    // c ^ ((r >> 1) & 7) (swizzle on the source address; ds_read_b128 of 16 rows then covers all banks), followed by the 24
    // lane-linear weight fragments [feature block 4][k step 2][piece 3].  One barrier per chunk: a wave waits (counted vmcnt) for
    // its own share of chunk c, the barrier certifies everybody's share and that chunk c-1 has been consumed, whose slot is then
    // refilled with chunk c+NS-1.  (Fetching every operand from global memory per wave - each weight fragment 4 x, each residual
    // line 8 x per workgroup - kept the vector memory pipe busy for 3 700 cycles per k step against 770 of MFMA.)
    {
        constexpr int NS = S::NS, WPW = S::WPW, PER = S::PER;
        const int nks = ap_ksteps(0, D), NC = (nks + 1) / 2, rb = wid >> 1, fb0 = 2 * (wid & 1);
        const uint32_t lds0 = (uint32_t)(size_t)((__attribute__((address_space(3))) char*)smem);
        // x share of this wave: rows 16 wid + 8 j + (lane >> 3), j = 0, 1; position lane & 7
        uint32_t xoff[2], xoff_last[2];
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            const int r = 16 * wid + 8 * jj + (lane >> 3), c = (lane & 7) ^ ((r >> 1) & 7);
            const int rowoff = (min(g0 + r, glast) - g0) * a.ld;
            xoff[jj] = (uint32_t)((rowoff + 4 * c) * 4);
            // last chunk: a float4 beyond the row is fetched from the row's last float4 (zeroed when the step is cut)
            xoff_last[jj] = (uint32_t)((rowoff + min(4 * c, a.ld - 4 - 32 * (NC - 1))) * 4);
        }
        const char* xbase = reinterpret_cast<const char*>(a.residual + (size_t)g0 * a.ld);
        const char* wbase = reinterpret_cast<const char*>(wp);
        const uint32_t woff = (uint32_t)(lane * 16);
        auto issue = [&](int c, int slot) {
            const uint32_t sl = lds0 + (uint32_t)(slot * S::L1SLOT);
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                const char* base = xbase + (size_t)c * 128;
                const uint32_t dst = sl + (uint32_t)((16 * wid + 8 * jj) * 128);
                const uint32_t vo = c == NC - 1 ? xoff_last[jj] : xoff[jj];
                asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 nt" ::"v"(vo), "s"(base), "s"(dst) : "memory", "m0");
            }
#pragma unroll
            for (int jj = 0; jj < WPW; ++jj) {
                const int f24 = WPW * wid + jj, fb = f24 / 6, within = f24 % 6;  // within = 3 (k step) + piece
                const char* base = wbase + ((size_t)(fb * nks + 2 * c) * 3 + within) * 1024;
                const uint32_t dst = sl + (uint32_t)(S::L1X + f24 * 1024);
                asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(woff), "s"(base), "s"(dst) : "memory", "m0");
            }
        };
        f32x16 acc0 = zero16, acc1 = zero16;
        const int xrow = rb * 32 + (lane & 31), xsw = (xrow >> 1) & 7, hh2 = (lane >> 5) * 2;
        auto compute = [&](int c, int slot) {
            const char* sl = smem + slot * S::L1SLOT;
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                const f32x4 p = *reinterpret_cast<const f32x4*>(sl + xrow * 128 + ((4 * st + hh2) ^ xsw) * 16);
                const f32x4 q = *reinterpret_cast<const f32x4*>(sl + xrow * 128 + ((4 * st + hh2 + 1) ^ xsw) * 16);
                float v[8] = {p[0], p[1], p[2], p[3], q[0], q[1], q[2], q[3]};
                const int ks = 2 * c + st;
                if (ks * 16 + 16 > D) {  // wave-uniform: the step that holds column D (and a phantom step behind it)
                    const int k0 = ks * 16 + (lane >> 5) * 8;
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = k0 + e < D ? v[e] : 0.0f;
                }
                qu32x4 w0[3], w1[3];
                const qu32x4* wf = reinterpret_cast<const qu32x4*>(sl + S::L1X) + lane;
#pragma unroll
                for (int pc = 0; pc < 3; ++pc) {
                    w0[pc] = wf[(fb0 * 6 + st * 3 + pc) * 64];
                    w1[pc] = wf[((fb0 + 1) * 6 + st * 3 + pc) * 64];
                }
                float h[8], m[8], l[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) ap_cut3(v[e], h[e], m[e], l[e]);
                qu32x4 x[3];
                x[0] = qu32x4{ap_top2(h[0], h[1]), ap_top2(h[2], h[3]), ap_top2(h[4], h[5]), ap_top2(h[6], h[7])};
                x[1] = qu32x4{ap_top2(m[0], m[1]), ap_top2(m[2], m[3]), ap_top2(m[4], m[5]), ap_top2(m[6], m[7])};
                x[2] = qu32x4{ap_top2(l[0], l[1]), ap_top2(l[2], l[3]), ap_top2(l[4], l[5]), ap_top2(l[6], l[7])};
                // 256 idle cycles behind each group of six MFMAs.  Without them this phase (eight waves issuing MFMAs back to back
                // next to the LDS-DMA) makes the chip fall into a lower clock state for the WHOLE step: measured on three boxes
                // (tools/gpu_ab.sh, alternating): 12.00 - 12.06 ms per step and 1075 W without the pauses against 11.52 - 11.65 ms
                // and 1140 W with them (2, 4, 6 or 8 units of 64 cycles alike), although the kernel itself is 8 % shorter without.
                ap_step(w0, x, acc0);
                __builtin_amdgcn_s_sleep(AP_PACE);
                ap_step(w1, x, acc1);
                __builtin_amdgcn_s_sleep(AP_PACE);
            }
        };
#pragma unroll
        for (int c = 0; c < NS - 1; ++c)
            if (c < NC) issue(c, c);
        int slot = 0;
#pragma unroll 1
        for (int c = 0; c < NC; ++c) {
            // chunks issued after chunk c: min(NS - 2, NC - 1 - c)
            if (NS > 2 && c + NS - 2 < NC) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER * (NS - 2)) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (c + NS - 1 < NC) issue(c + NS - 1, slot == 0 ? NS - 1 : slot - 1);
            compute(c, slot);
            slot = slot == NS - 1 ? 0 : slot + 1;
        }
        __syncthreads();  // the ring lies over image A: every wave must be done reading it
        ap_store_h<AP_AROW, AIMG>(HA, fb0, rb, lane, acc0, a.bias[0]);
        ap_store_h<AP_AROW, AIMG>(HA, fb0 + 1, rb, lane, acc1, a.bias[0]);
    }
    __syncthreads();
    AP_STAMP(1);
    const qu32x4* w2 = wp + ap_layer_offset(1, D) * 64;
    const qu32x4* w3 = wp + ap_layer_offset(2, D) * 64;
    const qu32x4* w4 = wp + ap_layer_offset(3, D) * 64;
    const qu32x4* w5 = wp + ap_layer_offset(4, D) * 64;
    const qu32x4* w6 = wp + ap_layer_offset(5, D) * 64;
    {  // 128 -> 64: 2 feature blocks x RB row blocks = one task per wave; A -> B
        f32x16 acc = zero16;
        ap_hidden_task<8, AP_AROW, AIMG>(w2, wid & 1, wid >> 1, HA, lane, acc);
        ap_store_h<AP_BROW, BIMG>(HB, wid & 1, wid >> 1, lane, acc, a.bias[1]);
    }
    __syncthreads();
    if (wid < RB) {  // 64 -> 32: 1 x RB tasks; B -> A
        f32x16 acc = zero16;
        ap_hidden_task<4, AP_BROW, BIMG>(w3, 0, wid, HB, lane, acc);
        ap_store_h<AP_AROW, AIMG>(HA, 0, wid, lane, acc, a.bias[2]);
    }
    __syncthreads();
    {  // 32 -> 64: 2 x RB tasks; A -> B
        f32x16 acc = zero16;
        ap_hidden_task<2, AP_AROW, AIMG>(w4, wid & 1, wid >> 1, HA, lane, acc);
        ap_store_h<AP_BROW, BIMG>(HB, wid & 1, wid >> 1, lane, acc, a.bias[3]);
    }
    __syncthreads();
    {  // 64 -> 128: 4 x RB tasks, two per wave; B -> A
        const int rb = wid >> 1;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            f32x16 acc = zero16;
            ap_hidden_task<4, AP_BROW, BIMG>(w5, 2 * (wid & 1) + i, rb, HB, lane, acc);
            ap_store_h<AP_AROW, AIMG>(HA, 2 * (wid & 1) + i, rb, lane, acc, a.bias[4]);
        }
    }
    __syncthreads();
    AP_STAMP(2);
    // ---- layer 6 (128 -> D): wave = feature blocks {wid + WAVES i} x the RB row blocks: 8 accumulators; every weight fragment
    // feeds RB x 6 MFMAs ----
    const int nfb = ap_fblocks(5, D);
#pragma unroll
    for (int i = 0; i < NFW; ++i)
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) acc[i][rb] = zero16;
#pragma unroll 1
    for (int ks = 0; ks < 8; ++ks) {
        qu32x4 x[RB][3];
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) ap_load_h<AP_AROW, AIMG>(HA, rb, ks, lane, x[rb]);
#pragma unroll
        for (int i = 0; i < NFW; ++i) {
            const int fb = wid + WAVES * i;
            if (fb < nfb) {  // wave-uniform
                qu32x4 w[3];
                ap_load_w(w6 + ((size_t)fb * 8 + ks) * 3 * 64, lane, w);
#pragma unroll
                for (int rb = 0; rb < RB; ++rb) ap_step(w, x[rb], acc[i][rb]);
            }
        }
    }
    AP_STAMP(3);
}

template <int ROWS, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void aff_pieces_kernel(AffPiecesArgs a) {
    using S = ApShape<ROWS, WAVES>;
    constexpr int RB = S::RB, NFW = S::NFW;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g0 = blockIdx.x * ROWS;
    const int D = a.D, nfb = ap_fblocks(5, D);
    f32x16 acc[NFW][RB];
    ap_mlp<ROWS, WAVES>(a, smem, g0, a.M - 1, tid, lane, wid, acc);
    // bias; per-row maximum over this wave's features (features >= D do not take part)
    const int n = lane & 31, hh = lane >> 5;
    float mx[RB];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) mx[rb] = -INFINITY;
#pragma unroll
    for (int i = 0; i < NFW; ++i) {
        const int fb = wid + WAVES * i;
        if (fb >= nfb) continue;
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int f = fb * 32 + 8 * g + 4 * hh + j;
                const float bv = f < D ? a.bias[5][f] : 0.0f;
#pragma unroll
                for (int rb = 0; rb < RB; ++rb) {
                    const float v = acc[i][rb][4 * g + j] + bv;
                    acc[i][rb][4 * g + j] = v;
                    if (f < D) mx[rb] = fmaxf(mx[rb], v);
                }
            }
    }
    __syncthreads();  // every wave is done reading image A: the staging and the softmax scratch may overwrite it
    float* xs = reinterpret_cast<float*>(smem);                  // [ROWS][AP_SROW] staging
    float* pmax = reinterpret_cast<float*>(smem + S::STAT);      // [WAVES][ROWS]
    float* psum = pmax + WAVES * ROWS;                           // [WAVES][ROWS]
    float* rmax = psum + WAVES * ROWS;                           // [ROWS]
    float* rinv = rmax + ROWS;                                   // [ROWS]
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
        const float o = fmaxf(mx[rb], __shfl_xor(mx[rb], 32, 64));
        if (hh == 0) pmax[wid * ROWS + rb * 32 + n] = o;
    }
    __syncthreads();
    float rm[RB], se[RB];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
        float m = pmax[rb * 32 + n];
#pragma unroll
        for (int w = 1; w < WAVES; ++w) m = fmaxf(m, pmax[w * ROWS + rb * 32 + n]);
        rm[rb] = m;
        se[rb] = 0.0f;
    }
#pragma unroll
    for (int i = 0; i < NFW; ++i) {
        const int fb = wid + WAVES * i;
        if (fb >= nfb) continue;
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int f = fb * 32 + 8 * g + 4 * hh + j;
                if (f < D) {
#pragma unroll
                    for (int rb = 0; rb < RB; ++rb) se[rb] += expf(acc[i][rb][4 * g + j] - rm[rb]);
                }
            }
    }
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
        const float o = se[rb] + __shfl_xor(se[rb], 32, 64);
        if (hh == 0) psum[wid * ROWS + rb * 32 + n] = o;
    }
    __syncthreads();
    if (tid < ROWS) {
        float s = psum[tid];
#pragma unroll
        for (int w = 1; w < WAVES; ++w) s += psum[w * ROWS + tid];  // fixed order
        float m = pmax[tid];
#pragma unroll
        for (int w = 1; w < WAVES; ++w) m = fmaxf(m, pmax[w * ROWS + tid]);
        rmax[tid] = m;
        rinv[tid] = 1.0f / s;
    }
    AP_STAMP(4);
    // ---- output: two passes of 8 feature blocks (256 columns) through the staging, whole row segments to global memory ----
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        __syncthreads();  // pass 0: rmax / rinv visible; pass 1: the read-out of pass 0 is finished
#pragma unroll
        for (int k = i * NFW / 2; k < (i + 1) * NFW / 2; ++k) {
            const int fb = wid + WAVES * k;
            if (fb < nfb) {
#pragma unroll
                for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                    for (int g = 0; g < 4; ++g)
                        *reinterpret_cast<f32x4*>(xs + (rb * 32 + n) * AP_SROW + (fb - 8 * i) * 32 + 8 * g + 4 * hh) =
                            f32x4{acc[k][rb][4 * g], acc[k][rb][4 * g + 1], acc[k][rb][4 * g + 2], acc[k][rb][4 * g + 3]};
            }
        }
        __syncthreads();
        const int c0 = i * AP_SCOLS;
        if (c0 >= D) continue;
        // a wave writes whole 256-column row segments: one ds_read_b128 per lane and row, `matched` (raw) as one 16-byte store,
        // matched1 = exp(x - max) / sum (rows t < N) as 8-byte stores (its rows are D floats apart)
        const int col = c0 + 4 * lane;
        const bool even = (D & 1) == 0;  // wave-uniform
#pragma unroll 4
        for (int rr = 0; rr < ROWS / WAVES; ++rr) {
            const int r = wid + rr * WAVES, gr = g0 + r;
            if (gr < a.M && col < a.ldm) {
                f32x4 v = *reinterpret_cast<const f32x4*>(xs + r * AP_SROW + 4 * lane);
                const float m = rmax[r], inv = rinv[r];
                const int b = gr / a.T, t = gr - b * a.T;
                float e[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    e[q] = expf(v[q] - m) * inv;
                    if (col + q >= D) v[q] = 0.0f;  // the padding columns of `matched`
                }
                *reinterpret_cast<f32x4*>(a.matched + (size_t)gr * a.ldm + col) = v;
                if (t < a.N) {
                    float* o = a.m1 + ((size_t)b * a.N + t) * D + col;
                    if (even) {
                        typedef float f32x2 __attribute__((ext_vector_type(2)));
                        if (col + 1 < D) *reinterpret_cast<f32x2*>(o) = f32x2{e[0], e[1]};
                        if (col + 3 < D) *reinterpret_cast<f32x2*>(o + 2) = f32x2{e[2], e[3]};
                    } else {
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            if (col + q < D) o[q] = e[q];
                    }
                }
            }
        }
    }
    AP_STAMP(5);
}

// aff_frame_kernel: the bf16-piece layers (ap_mlp) + the one-pass tail of aff_frame.hpp
template <int ROWS, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void aff_frame_kernel(AffFrameArgs fa) {
    using S = ApShape<ROWS, WAVES>;
    constexpr int RB = S::RB, NFW = S::NFW;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const AffPiecesArgs& a = fa.p;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int G = fa.G;
    const int tile = (int)ap_take_ticket(fa, reinterpret_cast<unsigned*>(smem));
    const int b = tile / G, q = tile - b * G;
    const int nrows = min(ROWS, a.T - q * ROWS), g0 = b * a.T + q * ROWS;
    f32x16 acc[NFW][RB];
    ap_mlp<ROWS, WAVES>(a, smem, g0, g0 + nrows - 1, tid, lane, wid, acc);
    const float rs[RB] = {};
    ap_frame_tail<ROWS, WAVES, false>(fa, smem, acc, b, q, nrows, g0, tid, lane, wid, nullptr, rs);
}

// LDS bytes of the workgroup shapes (independent of the table width, which must not exceed 512 columns)
bool aff_pieces_serves(int D) { return D <= 2 * AP_SCOLS; }

template <int ROWS, int WAVES>
static int launch_aff_pieces_shape(const AffPiecesArgs& a, hipStream_t st) {
    using S = ApShape<ROWS, WAVES>;
    const size_t lds = (size_t)S::ABYTES + S::BBYTES;
    (void)hipFuncSetAttribute((const void*)aff_pieces_kernel<ROWS, WAVES>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((aff_pieces_kernel<ROWS, WAVES>), dim3(cdiv(a.M, ROWS)), dim3(64 * WAVES), lds, st, a);
    return check_launch("aff_pieces");
}

size_t aff_frame_workspace_bytes(int B, int N) {
    const int T = N + 2, G = cdiv(T, 64);  // the 64-row shape needs the most partials
    return aff_frame_ctrl_bytes(B) + align_up((size_t)B * G * 1024 * sizeof(float), 256);
}

template <int ROWS, int WAVES>
static int launch_aff_frame_shape(AffFrameArgs& fa, int B, void* ws, hipStream_t st) {
    using S = ApShape<ROWS, WAVES>;
    const size_t lds = (size_t)S::ABYTES + S::BBYTES;
    fa.G = cdiv(fa.p.T, ROWS);
    unsigned* ctrl = static_cast<unsigned*>(ws);  // [status, ticket, arrive[B]]
    fa.status = ctrl;
    fa.ticket = ctrl + 1;
    fa.arrive = ctrl + 2;
    fa.part = reinterpret_cast<float*>(static_cast<char*>(ws) + aff_frame_ctrl_bytes(B));
    if (hipMemsetAsync(ctrl, 0, (size_t)(B + 2) * sizeof(unsigned), st) != hipSuccess) {
        set_error_msg("aff_frame: memset of the control words failed");
        return SHASTA_E_LAUNCH;
    }
    if (hipFuncSetAttribute((const void*)aff_frame_kernel<ROWS, WAVES>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
        (void)hipGetLastError();
        set_error_msg("aff_frame: the device refuses 160 KB of LDS per workgroup");
        return SHASTA_E_LAUNCH;
    }
    hipLaunchKernelGGL((aff_frame_kernel<ROWS, WAVES>), dim3(B * fa.G), dim3(64 * WAVES), lds, st, fa);
    return check_launch("aff_frame");
}

// the six layers and both softmaxes in one launch.  `matched` (B T x ldm, optional) receives the logits; ws: aff_frame_workspace_bytes
int launch_aff_frame(const shasta_weights* w, const float* packed_pieces, const float* residual, int ld, float* matched, int ldm, float* m1,
                     float* m2, int B, void* ws, hipStream_t st) {
    const int N = w->max_obj, T = N + 2, D = N + 2;
    AffFrameArgs fa;
    AffPiecesArgs& a = fa.p;
    a.wp = reinterpret_cast<const uint32_t*>(packed_pieces);
    for (int i = 0; i < 6; ++i) a.bias[i] = w->aff[i].bias;
    a.residual = residual;
    a.matched = matched;
    a.m1 = m1;
    a.M = B * T;
    a.T = T;
    a.N = N;
    a.D = D;
    a.Dp = (T + 3) / 4 * 4;
    a.ld = ld;
    a.ldm = ldm;
    fa.m2 = m2;
#if defined(AP_SHAPE_64)
    return launch_aff_frame_shape<64, 4>(fa, B, ws, st);
#elif defined(AP_SHAPE_128)
    return launch_aff_frame_shape<128, 8>(fa, B, ws, st);
#else
    return B * cdiv(T, 128) >= 256 ? launch_aff_frame_shape<128, 8>(fa, B, ws, st) : launch_aff_frame_shape<64, 4>(fa, B, ws, st);
#endif
}

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
    // 128-row workgroups once they fill the 256 CUs (0.825 ms against 0.837 ms for 257 k rows); below that the 64-row shape puts
    // twice as many workgroups on the chip
#if defined(AP_SHAPE_64) || defined(AP_SHAPE_128)  // probe builds only (tools/probes/aff_probe.hip): force one shape
#ifdef AP_SHAPE_64
    return launch_aff_pieces_shape<64, 4>(a, st);
#else
    return launch_aff_pieces_shape<128, 8>(a, st);
#endif
#else
    return cdiv(M, 128) >= 256 ? launch_aff_pieces_shape<128, 8>(a, st) : launch_aff_pieces_shape<64, 4>(a, st);
#endif
}

}  // namespace shasta
