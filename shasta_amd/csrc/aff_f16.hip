// K6, fp16-piece form (SHASTA_OPT_F16X2_AFF): the six aff layers (det3d/models/tracker/shasta.py:94-106, applied :323) with every fp32
// product formed from two range-scaled, round-to-nearest fp16 pieces per operand - w x = w_l x_h + w_h x_l + w_h x_h on
// v_mfma_f32_32x32x16_f16, the arithmetic of the weight stream (anchor_split.hip, NP = 2) and of the pair kernels (pair_f16*.hip) - and
// both softmaxes (:324-325) in the one-pass tail of aff_frame.hpp.  Three matrix instructions per fp32 product instead of the six of
// the bf16-piece kernel (aff_pieces.hip), two thirds of its fragment bytes.
//
// Scales (all exact powers of two, undone exactly in fp32):
//  * weights: one per output feature, fixed at pack time: the row's largest magnitude in (2^13, 2^14];
//  * activations of layers 2-6: one per residual row and layer, from the row's largest activation - known inside the kernel: a row's
//    features sit in two waves, which exchange their partial maxima through LDS across the barrier that separates the layers anyway;
//  * layer 1 (the residual rows arrive in 32-column chunks through an LDS-DMA ring, their maxima are not known in advance): one per
//    row and CHUNK; a chunk's products are accumulated from zero on the matrix cores and added to the running sums in fp32 with the
//    chunk's factor (16 packed fmas per 12 MFMAs).
// An fp16 piece pair carries 22 significant bits of every element that lies within 2^-17 of its block's largest magnitude and an
// absolute error of 2^-39 of that magnitude below: the block form needs no tight bound, only one that cannot overflow.
//
// Layer 1 is software-pipelined: the pieces of chunk c + 1 are cut (vector ALU) while the MFMAs of chunk c run; the ring holds four
// slots of 32 KB (x chunk + 16 weight fragments), i.e. chunk c + 1 was requested two trips before it is needed.
#include <type_traits>

#include "aff_frame.hpp"

// the LDS-DMA asm below names m0 in its clobber list on purpose (it writes it)
#pragma clang diagnostic ignored "-Winline-asm"

namespace shasta {

typedef _Float16 qh16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 qh16x2 __attribute__((ext_vector_type(2)));

#define AQ_MFMA(a, b, c) \
    __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(qh16x8, (a)), __builtin_bit_cast(qh16x8, (b)), (c), 0, 0, 0)

constexpr int aq_max(int a, int b) { return a > b ? a : b; }

template <int ROWS, int WAVES>
struct AqShape {
    using P = ApShape<ROWS, WAVES>;
    static constexpr int RB = ROWS / 32, NFW = P::NFW;
    static constexpr int AIMG = ROWS * AP_AROW, BIMG = ROWS * AP_BROW;  // one piece of image A / B
    static constexpr int ABYTES = 2 * AIMG, BBYTES = 2 * BIMG;
    static constexpr int L1X = ROWS * 128, L1SLOT = L1X + 16 * 1024;  // layer-1 ring slot: x chunk + 16 weight fragments
    static constexpr int TAIL = P::STAT + (2 * WAVES + 3) * ROWS * 4 + 4 * 512 * 4;  // what ap_frame_tail lays over the images
    static constexpr int SMALL = (WAVES * 32 + 2 * ROWS) * 4;                        // partial row maxima + two row-scale arrays
    static constexpr int BUDGET = (ROWS == 128 ? 160 : 80) * 1024;
    static constexpr int NS = (BUDGET - SMALL) / L1SLOT >= 4 ? 4 : 3;  // ring slots
    static constexpr int BODY = aq_max(aq_max(NS * L1SLOT, ABYTES + BBYTES), TAIL);
    static constexpr int LDS = BODY + SMALL;
    static constexpr int WPW = 16 / WAVES, PER = 2 + WPW;  // LDS-DMA instructions per chunk and wave: weights, total
    static_assert(LDS <= BUDGET, "LDS budget");
    static_assert(PER * (NS - 2) <= 63, "vmcnt is 6 bits");
};

// {a, b} -> packed fp16 pair, round to nearest even (v_cvt_pk_f16_f32)
__device__ __forceinline__ uint32_t aq_cvt2(float a, float b) {
    const qh16x2 v = {(_Float16)a, (_Float16)b};
    return __builtin_bit_cast(uint32_t, v);
}
// x - h (exact in fp32) with h = the low / high half of a packed fp16 pair read as an f16 operand.  The results feed aq_cvt2 (a
// compiler-generated VALU instruction), never an MFMA directly (pair_f16.hip, hazard rule)
__device__ __forceinline__ float aq_res_lo(float x, uint32_t hpk) {
    float r;
    asm("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel_hi:[0,0,1]" : "=v"(r) : "v"(x), "v"(hpk));
    return r;
}
__device__ __forceinline__ float aq_res_hi(float x, uint32_t hpk) {
    float r;
    asm("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(r) : "v"(x), "v"(hpk));
    return r;
}
// two scaled values -> their high and low piece pairs
__device__ __forceinline__ void aq_cut2(float sa, float sb, uint32_t& h, uint32_t& l) {
    h = aq_cvt2(sa, sb);
    l = aq_cvt2(aq_res_lo(sa, h), aq_res_hi(sb, h));
}

// ---- pack: [layer][feature block][k step][piece 2][64 lanes] x 16 B, then one descale factor per output feature and layer ----------
struct AffPack16Args {
    shasta_linear aff[6];
    float* out;
    int D;
};

// one wave per output row: the factor 2^-e with max |W[f][:]| 2^e in (2^13, 2^14]
__global__ __launch_bounds__(256) void aff_f16_scale_kernel(AffPack16Args a) {
    const int D = a.D, lane = threadIdx.x & 63;
    int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    for (int layer = 0; layer < 6; ++layer) {
        const int rows = ap_fblocks(layer, D) * 32;
        if (row < rows) {
            const int kin = ap_kin(layer, D), nout = ap_out(layer, D);
            float m = 0.0f;
            if (row < nout)
                for (int k = lane; k < kin; k += 64) m = absmax_keep_nan(m, fabsf(a.aff[layer].weight[(size_t)row * kin + k]));
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) m = absmax_keep_nan(m, __shfl_xor(m, off, 64));
            if (lane == 0) a.out[ap16_scale_offset(layer, D) + row] = ldexpf(1.0f, -range_exponent_bits(__float_as_uint(m)));
            return;
        }
        row -= rows;
    }
}

// one thread per (fragment pair, lane): 8 weights -> two 16-byte piece vectors
__global__ __launch_bounds__(256) void aff_f16_pack_kernel(AffPack16Args a) {
    const int D = a.D;
    for (int layer = 0; layer < 6; ++layer) {
        const int nks = ap_ksteps(layer, D), nfb = ap_fblocks(layer, D), kin = ap_kin(layer, D), nout = ap_out(layer, D);
        const float* W = a.aff[layer].weight;
        const float* dsc = a.out + ap16_scale_offset(layer, D);
        qu32x4* o = reinterpret_cast<qu32x4*>(a.out) + ap16_frag_offset(layer, D) * 64;
        for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < nfb * nks * 64; e += gridDim.x * blockDim.x) {
            const int lane = e & 63, ks = (e >> 6) % nks, fb = (e >> 6) / nks;
            const int f = fb * 32 + (lane & 31), k0 = ks * 16 + (lane >> 5) * 8;
            const float s = 1.0f / dsc[f];  // exact: a power of two
            uint32_t h[4], l[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float w[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) w[i] = (f < nout && k0 + 2 * j + i < kin) ? W[(size_t)f * kin + k0 + 2 * j + i] * s : 0.0f;
                h[j] = aq_cvt2(w[0], w[1]);
                const qh16x2 hv = __builtin_bit_cast(qh16x2, h[j]);
                l[j] = aq_cvt2(w[0] - (float)hv[0], w[1] - (float)hv[1]);
            }
            qu32x4* dst = o + ((size_t)(fb * nks + ks) * 2) * 64 + lane;
            dst[0] = qu32x4{h[0], h[1], h[2], h[3]};
            dst[64] = qu32x4{l[0], l[1], l[2], l[3]};
        }
    }
    // two spare (zero) fragments behind layer 6
    qu32x4* spare = reinterpret_cast<qu32x4*>(a.out) + ap16_frag_offset(6, D) * 64;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < 128; e += gridDim.x * blockDim.x) spare[e] = qu32x4{0, 0, 0, 0};
}

int aff_f16_pack(const shasta_weights* w, float* out, hipStream_t st) {
    AffPack16Args a;
    for (int i = 0; i < 6; ++i) a.aff[i] = w->aff[i];
    a.out = out;
    a.D = w->max_obj + 2;
    int rows = 0;
    for (int l = 0; l < 6; ++l) rows += ap_fblocks(l, a.D) * 32;
    hipLaunchKernelGGL(aff_f16_scale_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, st, a);
    int rc = check_launch("aff_f16_scale");
    if (rc) return rc;
    hipLaunchKernelGGL(aff_f16_pack_kernel, dim3(128), dim3(256), 0, st, a);
    return check_launch("aff_f16_pack");
}

// ---- layers ------------------------------------------------------------------------------------------------------------------------
template <int KS>
__device__ __forceinline__ void aq_load_w(const qu32x4* wl, int fb, int lane, qu32x4 (&w)[KS][2]) {
    const qu32x4* frag = wl + (size_t)fb * KS * 2 * 64 + lane;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        w[ks][0] = frag[(ks * 2) * 64];
        w[ks][1] = frag[(ks * 2 + 1) * 64];
    }
}

// activation pieces of k step ks, row block rb, from a hidden piece image (ROW bytes per row, IMG bytes per piece)
template <int ROW, int IMG>
__device__ __forceinline__ void aq_load_h(const char* H, int rb, int ks, int lane, qu32x4 (&x)[2]) {
    const char* p = H + (rb * 32 + (lane & 31)) * ROW + (ks * 16 + (lane >> 5) * 8) * 2;
    x[0] = *reinterpret_cast<const qu32x4*>(p);
    x[1] = *reinterpret_cast<const qu32x4*>(p + IMG);
}

// the three piece products of one k step, small to large (first operand = weight pieces, second = activation pieces; [0] = high)
__device__ __forceinline__ void aq_step(const qu32x4 (&w)[2], const qu32x4 (&x)[2], f32x16& acc) {
    acc = AQ_MFMA(w[1], x[0], acc);
    acc = AQ_MFMA(w[0], x[1], acc);
    acc = AQ_MFMA(w[0], x[0], acc);
}

template <int KS, int ROW, int IMG>
__device__ __forceinline__ void aq_hidden(const qu32x4 (&w)[KS][2], int rb, const char* Hin, int lane, f32x16& acc) {
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        qu32x4 x[2];
        aq_load_h<ROW, IMG>(Hin, rb, ks, lane, x);
        aq_step(w[ks], x, acc);
    }
}

// acc (block-scaled sums of feature block fb for the lane's row) -> relu(acc * scale * dsc[f] + bias[f]) in place
__device__ __forceinline__ void aq_activate(f32x16& acc, int fb, int hh, float scale, const float* __restrict__ dsc, const float* __restrict__ bias) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int f0 = fb * 32 + 8 * g + 4 * hh;
        const f32x4 d = *reinterpret_cast<const f32x4*>(dsc + f0);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[4 * g + j] = relu_nan(__builtin_fmaf(acc[4 * g + j], d[j] * scale, bias[f0 + j]));
    }
}

// The NT 32-feature tiles h[] (activations of row rb * 32 + (lane & 31), features fbs[t] * 32 + ...) of a layer's output: row maximum
// (the partner wave wid ^ 1 holds the row's other features when `paired`), the row's scale, pieces into the image Hout, the factor
// that undoes the scale into rsc_out.  Two barriers: every wave of the workgroup calls this, `active` or not.
template <int NT, int ROW, int IMG>
__device__ __forceinline__ void aq_finish(f32x16 (&h)[NT], const int (&fbs)[NT], int rb, bool active, bool paired, char* Hout, float* pm,
                                          float* rsc_out, int lane, int wid) {
    const int n = lane & 31, hh = lane >> 5;
    float m = 0.0f;
    if (active) {
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) m = fmaxf(m, h[t][r]);
        m = fmaxf(m, __shfl_xor(m, 32, 64));
        if (hh == 0) pm[wid * 32 + n] = m;
    }
    __syncthreads();  // partial maxima visible; nobody reads the image behind Hout any more
    if (active) {
        if (paired) m = fmaxf(m, pm[(wid ^ 1) * 32 + n]);
        const int e = range_exponent_bits(__float_as_uint(m));
        const float s = ldexpf(1.0f, e);
        if (hh == 0 && (!paired || (wid & 1) == 0)) rsc_out[rb * 32 + n] = ldexpf(1.0f, -e);
        char* row = Hout + (rb * 32 + n) * ROW;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                uint32_t h0, l0, h1, l1;
                aq_cut2(h[t][4 * g] * s, h[t][4 * g + 1] * s, h0, l0);
                aq_cut2(h[t][4 * g + 2] * s, h[t][4 * g + 3] * s, h1, l1);
                char* dst = row + (fbs[t] * 32 + 8 * g + 4 * hh) * 2;
                *reinterpret_cast<qu32x2*>(dst) = qu32x2{h0, h1};
                *reinterpret_cast<qu32x2*>(dst + IMG) = qu32x2{l0, l1};
            }
    }
    __syncthreads();
}

// The six layers for the ROWS residual rows [g0, g0 + ROWS) (rows beyond glast repeat row glast): on return acc[i][rb] holds the
// block-scaled layer-6 sums of features (wid + WAVES i) 32 + 8 (r >> 2) + 4 (lane >> 5) + (r & 3) of row 32 rb + (lane & 31), and
// rs[rb] the factor of that row (logit = acc * rs * dsc6[feature] + bias).
template <int ROWS, int WAVES>
__device__ __forceinline__ void aq_mlp(const AffPiecesArgs& a, char* smem, int g0, int glast, int tid, int lane, int wid,
                                       f32x16 (&acc)[AqShape<ROWS, WAVES>::NFW][AqShape<ROWS, WAVES>::RB], float (&rs)[AqShape<ROWS, WAVES>::RB]) {
    using S = AqShape<ROWS, WAVES>;
    constexpr int RB = S::RB, NFW = S::NFW, AIMG = S::AIMG, BIMG = S::BIMG;
    char* HA = smem;              // [2][ROWS][272 B]: layer outputs of width 128 / 32
    char* HB = smem + S::ABYTES;  // [2][ROWS][144 B]: layer outputs of width 64
    float* pm = reinterpret_cast<float*>(smem + S::BODY);  // [WAVES][32] partial row maxima
    float* rsc0 = pm + WAVES * 32;                          // [ROWS] row factors, two generations
    float* rsc1 = rsc0 + ROWS;
    const int D = a.D;
    const qu32x4* wq = reinterpret_cast<const qu32x4*>(a.wp);
    const float* sec = reinterpret_cast<const float*>(a.wp);
    const f32x16 zero16 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const int n = lane & 31, hh = lane >> 5, rbw = wid >> 1;
    AP_STAMP(0);

    // ---- layer 1 (K = D -> 128): wave = (row block wid >> 1, feature blocks 2 (wid & 1) + {0, 1}) ----
    f32x16 out0 = zero16, out1 = zero16;
    {
        constexpr int NS = S::NS, WPW = S::WPW, PER = S::PER;
        const int nks = ap_ksteps(0, D), NC = (nks + 1) / 2, fb0 = 2 * (wid & 1);
        const uint32_t lds0 = (uint32_t)(size_t)((__attribute__((address_space(3))) char*)smem);
        // x share of this wave: rows 16 wid + 8 j + (lane >> 3), j = 0, 1; position lane & 7 (swizzled by the row: aff_pieces.hip)
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
        const char* wbase = reinterpret_cast<const char*>(wq);
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
                const int fi = WPW * wid + jj, fb = fi >> 2, within = fi & 3;  // within = 2 (k step) + piece
                const char* base = wbase + ((size_t)(fb * nks + 2 * c) * 2 + within) * 1024;
                const uint32_t dst = sl + (uint32_t)(S::L1X + fi * 1024);
                asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(woff), "s"(base), "s"(dst) : "memory", "m0");
            }
        };
        const int xrow = rbw * 32 + n, xsw = (xrow >> 1) & 7, hh2 = hh * 2;
        // the lane's 2 x 8 values of chunk c -> fp16 pieces under the chunk's row scale; sinv undoes it
        // (tail: the chunk that holds column D - and behind it the padding of the residual rows and a phantom k step - is cut with those
        // positions zeroed: their weights are zero, but what the caller's padding columns hold need not be finite)
        auto prep = [&](auto tail, int c, int slot, qu32x4 (&xh)[2], qu32x4 (&xl)[2], float& sinv) {
            const char* sl = smem + slot * S::L1SLOT;
            float v[2][8];
            float m = 0.0f;
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                const f32x4 p = *reinterpret_cast<const f32x4*>(sl + xrow * 128 + ((4 * st + hh2) ^ xsw) * 16);
                const f32x4 q = *reinterpret_cast<const f32x4*>(sl + xrow * 128 + ((4 * st + hh2 + 1) ^ xsw) * 16);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[st][e] = p[e], v[st][4 + e] = q[e];
                if (decltype(tail)::value) {
                    const int k0 = (2 * c + st) * 16 + hh * 8;
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[st][e] = k0 + e < D ? v[st][e] : 0.0f;
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) m = fmaxf(m, fabsf(v[st][e]));
            }
            m = fmaxf(m, __shfl_xor(m, 32, 64));
            const int ex = range_exponent_bits(__float_as_uint(m));
            const float s = ldexpf(1.0f, ex);
            sinv = ldexpf(1.0f, -ex);
#pragma unroll
            for (int st = 0; st < 2; ++st)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    uint32_t ph, pl;
                    aq_cut2(v[st][2 * j] * s, v[st][2 * j + 1] * s, ph, pl);
                    xh[st][j] = ph;
                    xl[st][j] = pl;
                }
        };
        auto mma = [&](int slot, const qu32x4 (&xh)[2], const qu32x4 (&xl)[2], float sinv) {
            const qu32x4* wf = reinterpret_cast<const qu32x4*>(smem + slot * S::L1SLOT + S::L1X) + lane;
            f32x16 t0 = zero16, t1 = zero16;
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                const qu32x4 w0h = wf[((fb0 * 2 + st) * 2) * 64], w0l = wf[((fb0 * 2 + st) * 2 + 1) * 64];
                const qu32x4 w1h = wf[(((fb0 + 1) * 2 + st) * 2) * 64], w1l = wf[(((fb0 + 1) * 2 + st) * 2 + 1) * 64];
                t0 = AQ_MFMA(w0l, xh[st], t0);
                t1 = AQ_MFMA(w1l, xh[st], t1);
                t0 = AQ_MFMA(w0h, xl[st], t0);
                t1 = AQ_MFMA(w1h, xl[st], t1);
                t0 = AQ_MFMA(w0h, xh[st], t0);
                t1 = AQ_MFMA(w1h, xh[st], t1);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                out0[r] = __builtin_fmaf(t0[r], sinv, out0[r]);
                out1[r] = __builtin_fmaf(t1[r], sinv, out1[r]);
            }
        };
#pragma unroll
        for (int c = 0; c < NS - 1; ++c)
            if (c < NC) issue(c, c);
        // chunk 0 has landed when at most the NS - 2 younger chunks are outstanding
        if (NC >= NS - 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER * (NS - 2)) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        qu32x4 xh[2], xl[2];
        float sinv;
        if (NC == 1) prep(std::true_type{}, 0, 0, xh, xl, sinv);
        else prep(std::false_type{}, 0, 0, xh, xl, sinv);
        int slot = 0;
#pragma unroll 1
        for (int c = 0; c < NC; ++c) {
            // chunk c + 1 must have landed (requested so far: up to chunk c + NS - 2); the barrier also certifies that chunk c - 1
            // has been consumed, whose slot is refilled with chunk c + NS - 1
            if (NS > 3 && c + NS - 2 <= NC - 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER * (NS > 3 ? NS - 3 : 0)) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (c + NS - 1 < NC) issue(c + NS - 1, slot == 0 ? NS - 1 : slot - 1);
            const int nslot = slot == NS - 1 ? 0 : slot + 1;
            qu32x4 yh[2], yl[2];
            float tinv;
            const bool more = c + 1 < NC;  // the last trip cuts its own chunk again (unused): one straight-line body per case
            if (c + 1 >= NC - 1) {
                prep(std::true_type{}, more ? c + 1 : c, more ? nslot : slot, yh, yl, tinv);
                mma(slot, xh, xl, sinv);
            } else {
                prep(std::false_type{}, c + 1, nslot, yh, yl, tinv);
                mma(slot, xh, xl, sinv);
            }
#pragma unroll
            for (int st = 0; st < 2; ++st) xh[st] = yh[st], xl[st] = yl[st];
            sinv = tinv;
            slot = nslot;
        }
    }
    AP_STAMP(1);
    const qu32x4* w2 = wq + ap16_frag_offset(1, D) * 64;
    const qu32x4* w3 = wq + ap16_frag_offset(2, D) * 64;
    const qu32x4* w4 = wq + ap16_frag_offset(3, D) * 64;
    const qu32x4* w5 = wq + ap16_frag_offset(4, D) * 64;
    const qu32x4* w6 = wq + ap16_frag_offset(5, D) * 64;
    {  // layer-1 output: 2 tiles per wave, the partner wave holds the row's other 64 features; -> image A (over the ring)
        qu32x4 wn[8][2];
        aq_load_w<8>(w2, wid & 1, lane, wn);  // next layer's weights: in flight across the barriers
        const int fb0 = 2 * (wid & 1);
        f32x16 h[2] = {out0, out1};
        aq_activate(h[0], fb0, hh, 1.0f, sec + ap16_scale_offset(0, D), a.bias[0]);
        aq_activate(h[1], fb0 + 1, hh, 1.0f, sec + ap16_scale_offset(0, D), a.bias[0]);
        const int fbs[2] = {fb0, fb0 + 1};
        aq_finish<2, AP_AROW, AIMG>(h, fbs, rbw, true, true, HA, pm, rsc0, lane, wid);
        // 128 -> 64: 2 feature blocks x RB row blocks = one task per wave; A -> B
        f32x16 t[1] = {zero16};
        aq_hidden<8, AP_AROW, AIMG>(wn, rbw, HA, lane, t[0]);
        qu32x4 w3n[4][2];
        if (wid < RB) aq_load_w<4>(w3, 0, lane, w3n);
        aq_activate(t[0], wid & 1, hh, rsc0[rbw * 32 + n], sec + ap16_scale_offset(1, D), a.bias[1]);
        const int f1[1] = {wid & 1};
        aq_finish<1, AP_BROW, BIMG>(t, f1, rbw, true, true, HB, pm, rsc1, lane, wid);
        // 64 -> 32: RB tasks; B -> A
        t[0] = zero16;
        const bool act3 = wid < RB;
        if (act3) {
            aq_hidden<4, AP_BROW, BIMG>(w3n, wid, HB, lane, t[0]);
            aq_activate(t[0], 0, hh, rsc1[wid * 32 + n], sec + ap16_scale_offset(2, D), a.bias[2]);
        }
        qu32x4 w4n[2][2];
        aq_load_w<2>(w4, wid & 1, lane, w4n);
        const int f0[1] = {0};
        aq_finish<1, AP_AROW, AIMG>(t, f0, act3 ? wid : 0, act3, false, HA, pm, rsc0, lane, wid);
        // 32 -> 64: 2 x RB tasks; A -> B
        t[0] = zero16;
        aq_hidden<2, AP_AROW, AIMG>(w4n, rbw, HA, lane, t[0]);
        qu32x4 w5n[2][4][2];
        aq_load_w<4>(w5, 2 * (wid & 1), lane, w5n[0]);
        aq_load_w<4>(w5, 2 * (wid & 1) + 1, lane, w5n[1]);
        aq_activate(t[0], wid & 1, hh, rsc0[rbw * 32 + n], sec + ap16_scale_offset(3, D), a.bias[3]);
        aq_finish<1, AP_BROW, BIMG>(t, f1, rbw, true, true, HB, pm, rsc1, lane, wid);
        // 64 -> 128: 4 x RB tasks, two per wave; B -> A
        h[0] = h[1] = zero16;
        aq_hidden<4, AP_BROW, BIMG>(w5n[0], rbw, HB, lane, h[0]);
        aq_hidden<4, AP_BROW, BIMG>(w5n[1], rbw, HB, lane, h[1]);
        const float r5 = rsc1[rbw * 32 + n];
        aq_activate(h[0], fb0, hh, r5, sec + ap16_scale_offset(4, D), a.bias[4]);
        aq_activate(h[1], fb0 + 1, hh, r5, sec + ap16_scale_offset(4, D), a.bias[4]);
        aq_finish<2, AP_AROW, AIMG>(h, fbs, rbw, true, true, HA, pm, rsc0, lane, wid);
    }
    AP_STAMP(2);
    // ---- layer 6 (128 -> D): wave = feature blocks {wid + WAVES i} x the RB row blocks; every weight fragment feeds RB x 3 MFMAs, the
    // fragments of the next k step are requested before the MFMAs of this one ----
    const int nfb = ap_fblocks(5, D);
#pragma unroll
    for (int i = 0; i < NFW; ++i)
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) acc[i][rb] = zero16;
    qu32x4 wc[NFW][2];
    auto load6 = [&](int ks, qu32x4 (&w)[NFW][2]) {
#pragma unroll
        for (int i = 0; i < NFW; ++i) {
            const int fb = wid + WAVES * i;
            if (fb < nfb) {  // wave-uniform
                const qu32x4* frag = w6 + ((size_t)fb * 8 + ks) * 2 * 64 + lane;
                w[i][0] = frag[0];
                w[i][1] = frag[64];
            }
        }
    };
    load6(0, wc);
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
        qu32x4 wnx[NFW][2];
        if (ks + 1 < 8) load6(ks + 1, wnx);
        qu32x4 x[RB][2];
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) aq_load_h<AP_AROW, AIMG>(HA, rb, ks, lane, x[rb]);
#pragma unroll
        for (int i = 0; i < NFW; ++i) {
            if (wid + WAVES * i < nfb) {
#pragma unroll
                for (int rb = 0; rb < RB; ++rb) aq_step(wc[i], x[rb], acc[i][rb]);
            }
        }
        if (ks + 1 < 8) {
#pragma unroll
            for (int i = 0; i < NFW; ++i) wc[i][0] = wnx[i][0], wc[i][1] = wnx[i][1];
        }
    }
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) rs[rb] = rsc0[rb * 32 + n];
    AP_STAMP(3);
}

template <int ROWS, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void aff_frame16_kernel(AffFrameArgs fa) {
    using S = AqShape<ROWS, WAVES>;
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
    float rs[RB];
    aq_mlp<ROWS, WAVES>(a, smem, g0, g0 + nrows - 1, tid, lane, wid, acc, rs);
    ap_frame_tail<ROWS, WAVES, true>(fa, smem, acc, b, q, nrows, g0, tid, lane, wid, reinterpret_cast<const float*>(a.wp) + ap16_scale_offset(5, a.D), rs);
}

size_t aff_frame_workspace_bytes(int B, int N);

template <int ROWS, int WAVES>
static int launch_aff_frame16_shape(AffFrameArgs& fa, int B, void* ws, hipStream_t st) {
    using S = AqShape<ROWS, WAVES>;
    fa.G = cdiv(fa.p.T, ROWS);
    unsigned* ctrl = static_cast<unsigned*>(ws);  // [status, ticket, arrive[B]]
    fa.status = ctrl;
    fa.ticket = ctrl + 1;
    fa.arrive = ctrl + 2;
    fa.part = reinterpret_cast<float*>(static_cast<char*>(ws) + aff_frame_ctrl_bytes(B));
    if (hipMemsetAsync(ctrl, 0, (size_t)(B + 2) * sizeof(unsigned), st) != hipSuccess) {
        set_error_msg("aff_frame16: memset of the control words failed");
        return SHASTA_E_LAUNCH;
    }
    if (hipFuncSetAttribute((const void*)aff_frame16_kernel<ROWS, WAVES>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)S::LDS) != hipSuccess) {
        (void)hipGetLastError();
        set_error_msg("aff_frame16: the device refuses 160 KB of LDS per workgroup");
        return SHASTA_E_LAUNCH;
    }
    hipLaunchKernelGGL((aff_frame16_kernel<ROWS, WAVES>), dim3(B * fa.G), dim3(64 * WAVES), S::LDS, st, fa);
    return check_launch("aff_frame16");
}

// the six layers (fp16 pieces) and both softmaxes in one launch; packed16 = the aff16 section of the packed buffer
int launch_aff_frame16(const shasta_weights* w, const float* packed16, const float* residual, int ld, float* matched, int ldm, float* m1,
                       float* m2, int B, void* ws, hipStream_t st) {
    const int N = w->max_obj, T = N + 2, D = N + 2;
    AffFrameArgs fa;
    AffPiecesArgs& a = fa.p;
    a.wp = reinterpret_cast<const uint32_t*>(packed16);
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
    return launch_aff_frame16_shape<64, 4>(fa, B, ws, st);
#elif defined(AP_SHAPE_128)
    return launch_aff_frame16_shape<128, 8>(fa, B, ws, st);
#else
    return B * cdiv(T, 128) >= 256 ? launch_aff_frame16_shape<128, 8>(fa, B, ws, st) : launch_aff_frame16_shape<64, 4>(fa, B, ws, st);
#endif
}

}  // namespace shasta
