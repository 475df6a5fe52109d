// K4a + K4b for large batches: the row embeddings UP[t] / UC[d] of the factorised pair-MLP first layers
// (det3d/models/tracker/shasta.py:59-60 fuse_shape.0, :86-87 res_coeff.0, :78-79 fuse_det.0, applied :286-316) in ONE kernel
// per side:   E[row] = [ fuse_shape.0 | res_coeff.0 | fuse_det.0 ](row's features, row's box) (+ bias on the current side)
//   columns [0, E12)   feature part  X[row][0:F] . Wemb^T   on the bf16 matrix path, every fp32 product from three exact bf16 pieces
//                      (the arithmetic of gemm_pieces.hip / aff_pieces.hip, six products, fp32 accumulation)
//   columns [H1, ET)   box part      box[row][0:nf] . Wbox^T  (k-ordered fma chain, as row_prep_kernel)
//   hand[row][13]      the row's largest |E| (range scaling of the fp16 pair kernel)
// It replaces gemm_nt_pieces_kernel (128 x 128 tiles, both operands cut per K slice behind two barriers: 0.40 ms at 512
// frame-pairs, four times its matrix time) + the row role of row_prep_kernel (a second pass over the embeddings: 0.23 ms).
//
// Structure = layer 1 of aff_pieces_kernel: a workgroup owns 256 table rows on 8 waves, wave = one 32-row block x all NFB
// feature blocks.  The weights are cut once, at pack time, into MFMA fragment order; rows and fragments reach LDS by LDS-DMA in
// chunks of 32 columns through a ring of slots (a wave fetches its OWN rows; the fragments are shared), one barrier per chunk,
// counted vmcnt.  A wave cuts its 8 row values per k step once and feeds NFB x 6 MFMAs with them.  Epilogue: accumulators ->
// fp32 staging [256][ET + 8] (row values + the row's box) over the ring -> 32 threads per row add bias and box part, take the row maximum and store the row.
#include <algorithm>

#include "common.hpp"
#include "pair_layout.hpp"

// the LDS-DMA asm below names m0 in its clobber list on purpose (it writes it)
#pragma clang diagnostic ignored "-Winline-asm"

namespace shasta {

typedef __bf16 ebf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t eu32x4 __attribute__((ext_vector_type(4)));

// Workgroup shapes (rows = 32 x waves).  <256 rows, 8 waves>: one workgroup per CU, three ring slots.  <128, 4, two slots> at F = 256:
// 72 KB of LDS, TWO workgroups per CU - the epilogue of one (staging, box columns, row maxima, 64 KB of stores: about as long as its
// main loop) runs under the MFMAs of the other: 0.496 -> 0.459 ms per 1024 frame-pairs in an alternating A/B on one box
// (tools/gpu_kernel_ab.sh; <64, 2>: 0.77, <128, 4, three slots> = one workgroup per CU again: 0.67).  Results are the same bits - a
// wavefront's arithmetic does not depend on the shape.

__device__ __forceinline__ void er_cut3(float a, float& h, float& m, float& l) {
    h = __uint_as_float(__float_as_uint(a) & 0xffff0000u);
    const float r = a - h;
    m = __uint_as_float(__float_as_uint(r) & 0xffff0000u);
    l = r - m;
}
__device__ __forceinline__ uint32_t er_top2(float even, float odd) {
    return __builtin_amdgcn_perm(__float_as_uint(odd), __float_as_uint(even), 0x07060302u);
}
#define ER_MFMA(a, b, c) \
    __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(ebf16x8, (a)), __builtin_bit_cast(ebf16x8, (b)), (c), 0, 0, 0)

// fragments per side: [feature block NFB][k step F/16][piece 3][64 lanes] x 16 B
size_t embed_packed_floats(int F) {
    const PairDims d(F);
    const int nfb = (d.H1 + d.R1 + 31) / 32;
    return (size_t)2 * nfb * (F / 16) * 3 * 256;
}

struct EmbedPackArgs {
    const float* w[2];  // [E12][F] row-major: prev / cur feature columns of fuse_shape.0 | res_coeff.0 (PackedLayout wemb_*)
    uint32_t* out;
    int F, E12, nfb;
};

__global__ __launch_bounds__(256) void embed_pack_kernel(EmbedPackArgs a) {
    const int nks = a.F / 16;
    const int per_side = a.nfb * nks * 64;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < 2 * per_side; e += gridDim.x * blockDim.x) {
        const int side = e / per_side, r = e % per_side;
        const int lane = r & 63, ks = (r >> 6) % nks, fb = (r >> 6) / nks;
        const int f = fb * 32 + (lane & 31), k0 = ks * 16 + (lane >> 5) * 8;
        float h[8], m[8], l[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) er_cut3(f < a.E12 ? a.w[side][(size_t)f * a.F + k0 + j] : 0.0f, h[j], m[j], l[j]);
        eu32x4* dst = reinterpret_cast<eu32x4*>(a.out) + ((size_t)((side * a.nfb + fb) * nks + ks) * 3) * 64 + lane;
        dst[0] = eu32x4{er_top2(h[0], h[1]), er_top2(h[2], h[3]), er_top2(h[4], h[5]), er_top2(h[6], h[7])};
        dst[64] = eu32x4{er_top2(m[0], m[1]), er_top2(m[2], m[3]), er_top2(m[4], m[5]), er_top2(m[6], m[7])};
        dst[128] = eu32x4{er_top2(l[0], l[1]), er_top2(l[2], l[3]), er_top2(l[4], l[5]), er_top2(l[6], l[7])};
    }
}

// `packed` = the packed weight buffer (PackedLayout); writes its embp section from its wemb_* sections
int embed_pack(const shasta_weights* w, float* packed, hipStream_t st) {
    const PackedLayout P(w->max_obj, w->num_feats, w->feat_dim);
    EmbedPackArgs a;
    a.w[0] = packed + P.wemb_prev;
    a.w[1] = packed + P.wemb_cur;
    a.out = reinterpret_cast<uint32_t*>(packed + P.embp);
    a.F = w->feat_dim;
    a.E12 = P.E12;
    a.nfb = (P.E12 + 31) / 32;
    hipLaunchKernelGGL(embed_pack_kernel, dim3(64), dim3(256), 0, st, a);
    return check_launch("embed_pack");
}

#ifdef SHASTA_EMBED_STAMP  // diagnostic build only (tools/probes/embed_probe.hip)
__device__ unsigned long long g_embed_stamp[4096][8];
#define ER_STAMP(i) \
    if (threadIdx.x == 0 && blockIdx.y == 0 && blockIdx.x < 4096) g_embed_stamp[blockIdx.x][i] = __builtin_amdgcn_s_memtime()
#else
#define ER_STAMP(i)
#endif

struct EmbedArgs {
    const float* x[2];     // [0] prev feature table, [1] current: (M, F) rows
    const float* tab[2];   // box tables (M, 8)
    float* emb[2];         // UP, UC (M, ET)
    float* hand[2];        // (M, 16): slot 13 is written here
    const uint32_t* wp;    // fragments (embed_pack)
    const float* packed;   // PackedLayout sections: bemb_cur, wbox_*, bbox_cur
    int M, F, nf, N;
};

template <int NFB, int ER_ROWS, int ER_WAVES, int ER_NS>
struct ErShape {
    static_assert(ER_ROWS == 32 * ER_WAVES, "a wave owns 32 rows");
    static constexpr int XB = ER_ROWS * 128;              // x chunk: ER_ROWS rows x 32 floats
    static constexpr int SLOT = XB + NFB * 6 * 1024;      // + NFB x 2 k steps x 3 pieces fragments
    static constexpr int NS = ER_NS ? ER_NS : (3 * SLOT <= 152 * 1024 ? 3 : 2);
    static constexpr int NW = NFB * 6;                    // weight fragments per chunk
    static constexpr int PW = (NW + ER_WAVES - 1) / ER_WAVES, PER = 4 + PW;  // LDS-DMA instructions per chunk and wave
    static_assert(PER * (NS - 1) <= 63, "vmcnt is 6 bits");
};

template <int NFB, int ER_ROWS, int ER_WAVES, int ER_NS>
__global__ __launch_bounds__(64 * ER_WAVES) void embed_rows_kernel(EmbedArgs a) {
    using S = ErShape<NFB, ER_ROWS, ER_WAVES, ER_NS>;
    constexpr int NS = S::NS, PW = S::PW, PER = S::PER;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int side = blockIdx.y, g0 = blockIdx.x * ER_ROWS;
    const int F = a.F, nks = F / 16, NC = F / 32;
    const PairDims d(F);
    const PackedLayout P(a.N, a.nf, F);
    const int ES = d.ET + 8;  // staging row stride (floats): the row's values, then its 8 box floats
    ER_STAMP(0);
    float* wT = reinterpret_cast<float*>(smem + max(NS * S::SLOT, ER_ROWS * ES * 4));  // [7][104] box-column weights, transposed
    {
        const float* wb = a.packed + (side ? P.wbox_cur : P.wbox_prev);
        const int J = d.R1 + 32;
        for (int e = tid; e < J * 8; e += 64 * ER_WAVES) {
            const int j = e >> 3, c = e & 7;
            if (c < 7) wT[c * 104 + j] = wb[e];
        }
    }
    // this thread's share of the workgroup's 256 box rows (32 bytes each): half a row, parked in registers until the staging exists
    const f32x4 mybox = *reinterpret_cast<const f32x4*>(a.tab[side] + (size_t)min(g0 + (tid >> 1), a.M - 1) * 8 + 4 * (tid & 1));
    const f32x16 zero16 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    f32x16 acc[NFB];
#pragma unroll
    for (int i = 0; i < NFB; ++i) acc[i] = zero16;
    {
        const uint32_t lds0 = (uint32_t)(size_t)((__attribute__((address_space(3))) char*)smem);
        // x share of this wave = its own 32 rows: instruction j covers rows 32 wid + 8 j + (lane >> 3), position lane & 7 holds the
        // 16-byte piece (lane & 7) ^ ((row >> 1) & 7) of the chunk (source-side swizzle: conflict-free ds_read_b128)
        uint32_t xoff[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int r = 32 * wid + 8 * j + (lane >> 3), c = (lane & 7) ^ ((r >> 1) & 7);
            xoff[j] = (uint32_t)(((min(g0 + r, a.M - 1) - g0) * F + 4 * c) * 4);
        }
        const char* xbase = reinterpret_cast<const char*>(a.x[side] + (size_t)g0 * F);
        const char* wbase = reinterpret_cast<const char*>(a.wp) + (size_t)side * NFB * nks * 3 * 1024;
        const uint32_t woff = (uint32_t)(lane * 16);
        auto issue = [&](int c, int slot) {
            const uint32_t sl = lds0 + (uint32_t)(slot * S::SLOT);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const char* base = xbase + (size_t)c * 128;
                const uint32_t dst = sl + (uint32_t)((32 * wid + 8 * j) * 128);
                asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 nt" ::"v"(xoff[j]), "s"(base), "s"(dst) : "memory", "m0");
            }
#pragma unroll
            for (int j = 0; j < PW; ++j) {
                const int f = (PW * wid + j) % S::NW, fb = f / 6, within = f % 6;  // within = 3 (k step) + piece; spare slots repeat fragments
                const char* base = wbase + ((size_t)(fb * nks + 2 * c) * 3 + within) * 1024;
                const uint32_t dst = sl + (uint32_t)(S::XB + f * 1024);
                asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(woff), "s"(base), "s"(dst) : "memory", "m0");
            }
        };
        const int xrow = wid * 32 + (lane & 31), xsw = (xrow >> 1) & 7, hh2 = (lane >> 5) * 2;
        auto compute = [&](int slot) {
            const char* sl = smem + slot * S::SLOT;
            // every LDS read of the chunk is issued before the first use (left to itself the compiler fetched the weight fragments
            // one MFMA group at a time, each behind a full wait)
            f32x4 xr[2][2];
            eu32x4 w[2][NFB][3];
            const eu32x4* wf = reinterpret_cast<const eu32x4*>(sl + S::XB) + lane;
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                xr[st][0] = *reinterpret_cast<const f32x4*>(sl + xrow * 128 + ((4 * st + hh2) ^ xsw) * 16);
                xr[st][1] = *reinterpret_cast<const f32x4*>(sl + xrow * 128 + ((4 * st + hh2 + 1) ^ xsw) * 16);
            }
#pragma unroll
            for (int st = 0; st < 2; ++st)
#pragma unroll
                for (int fb = 0; fb < NFB; ++fb)
#pragma unroll
                    for (int pc = 0; pc < 3; ++pc) w[st][fb][pc] = wf[(fb * 6 + st * 3 + pc) * 64];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                const f32x4 p = xr[st][0], q = xr[st][1];
                const float v[8] = {p[0], p[1], p[2], p[3], q[0], q[1], q[2], q[3]};
                float h[8], m[8], l[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) er_cut3(v[e], h[e], m[e], l[e]);
                eu32x4 x[3];
                x[0] = eu32x4{er_top2(h[0], h[1]), er_top2(h[2], h[3]), er_top2(h[4], h[5]), er_top2(h[6], h[7])};
                x[1] = eu32x4{er_top2(m[0], m[1]), er_top2(m[2], m[3]), er_top2(m[4], m[5]), er_top2(m[6], m[7])};
                x[2] = eu32x4{er_top2(l[0], l[1]), er_top2(l[2], l[3]), er_top2(l[4], l[5]), er_top2(l[6], l[7])};
                // the six piece products, small to large (first operand = weight pieces: out^T[feature][row]), product-major so that
                // consecutive MFMAs go to different accumulators
                constexpr int PWI[6] = {2, 0, 1, 1, 0, 0}, PXI[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
                for (int pr = 0; pr < 6; ++pr)
#pragma unroll
                    for (int fb = 0; fb < NFB; ++fb) acc[fb] = ER_MFMA(w[st][fb][PWI[pr]], x[PXI[pr]], acc[fb]);
            }
        };
        ER_STAMP(1);
        // tools/probes/embed_probe.hip: 37 k cycles per workgroup in this loop, 18 k of them MFMA time; launch time with parts removed
        // (timing only): 0.250 ms -> 0.195 without five of the six MFMAs, 0.232 with the rows always from one cached chunk, 0.245
        // with a third of the fragment reads, 0.246 without the barrier.
#pragma unroll
        for (int c = 0; c < NS - 1; ++c)
            if (c < NC) issue(c, c);
        int slot = 0;
#pragma unroll 1
        for (int c = 0; c < NC; ++c) {
            if (NS > 2 && c + NS - 2 < NC) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER * (NS - 2)) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (c + NS - 1 < NC) issue(c + NS - 1, slot == 0 ? NS - 1 : slot - 1);
            compute(slot);
            slot = slot == NS - 1 ? 0 : slot + 1;
        }
    }
    ER_STAMP(2);
    __syncthreads();  // the staging lies over the ring
    float* xs = reinterpret_cast<float*>(smem);
    {
        const int n = lane & 31, hh = lane >> 5;
#pragma unroll
        for (int fb = 0; fb < NFB; ++fb)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                if (fb * 32 + 8 * g + 4 * hh < d.ET)  // F = 320: the last feature block reaches past the staged row
                    *reinterpret_cast<f32x4*>(xs + (wid * 32 + n) * ES + fb * 32 + 8 * g + 4 * hh) =
                        f32x4{acc[fb][4 * g], acc[fb][4 * g + 1], acc[fb][4 * g + 2], acc[fb][4 * g + 3]};
        *reinterpret_cast<f32x4*>(xs + (tid >> 1) * ES + d.ET + 4 * (tid & 1)) = mybox;
    }
    __syncthreads();
    ER_STAMP(3);
    // rows: 32 threads per row, a float4 of outputs per thread and pass (ET <= 128: one pass)
    const int q = tid & 31;
    // this thread's columns 4 q (and 4 q + 128 when ET > 128) are the same for every row: their bias is loaded once
    f32x4 biasv[2] = {{0.0f, 0.0f, 0.0f, 0.0f}, {0.0f, 0.0f, 0.0f, 0.0f}};
    if (side) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int j = 4 * q + 128 * i;
            if (j < P.E12) biasv[i] = *reinterpret_cast<const f32x4*>(a.packed + P.bemb_cur + j);
            else if (j < d.ET) biasv[i] = *reinterpret_cast<const f32x4*>(a.packed + P.bbox_cur + (j - P.E12));
        }
    }
    // ... and so are its box-column weights (7 float4 per column group)
    f32x4 wq[2][7];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int j = 4 * q + 128 * i;
#pragma unroll
        for (int c = 0; c < 7; ++c)
            wq[i][c] = (j >= d.H1 && j < d.ET) ? *reinterpret_cast<const f32x4*>(&wT[c * 104 + (j - d.H1)]) : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    }
    // the 16 rows of a thread, then ONE round of cross-lane maxima for all of them (16 independent shuffles per step: a dependent
    // five-step reduction inside the row loop cost more than the arithmetic of a row)
    constexpr int RPT = ER_ROWS / (2 * ER_WAVES);
    float amax[RPT];
#pragma unroll
    for (int it = 0; it < RPT; ++it) {
        const int r = (tid >> 5) + it * 2 * ER_WAVES, row = g0 + r;
        amax[it] = 0.0f;
        if (row >= a.M) continue;  // the 32 threads of a row decide alike
        const f32x4* bp = reinterpret_cast<const f32x4*>(xs + r * ES + d.ET);  // LDS broadcast
        const f32x4 b0 = bp[0], b1 = bp[1];
        const float bx[7] = {b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2]};
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int j = 4 * q + 128 * i;
            if (j >= d.ET) break;
            f32x4 s = {0.0f, 0.0f, 0.0f, 0.0f};
            if (j >= d.H1) {  // box columns of res_coeff.0 / fuse_det.0 (columns >= nf are packed as 0); k-ordered fmaf chain per output
#pragma unroll
                for (int c = 0; c < 7; ++c) {
                    const f32x4 w4 = wq[i][c];
                    s[0] = fmaf(w4[0], bx[c], s[0]); s[1] = fmaf(w4[1], bx[c], s[1]);
                    s[2] = fmaf(w4[2], bx[c], s[2]); s[3] = fmaf(w4[3], bx[c], s[3]);
                }
            }
            f32x4 v = biasv[i];
            if (j < P.E12) {  // feature part (+ bias on the current side), rounded before it meets the box part
                const f32x4 g = *reinterpret_cast<const f32x4*>(xs + r * ES + j);
                v[0] += g[0]; v[1] += g[1]; v[2] += g[2]; v[3] += g[3];
            }
            s[0] += v[0]; s[1] += v[1]; s[2] += v[2]; s[3] += v[3];
            *reinterpret_cast<f32x4*>(a.emb[side] + (size_t)row * d.ET + j) = s;
            amax[it] = absmax_keep_nan(amax[it], absmax_keep_nan(absmax_keep_nan(fabsf(s[0]), fabsf(s[1])), absmax_keep_nan(fabsf(s[2]), fabsf(s[3]))));
        }
    }
    // maxima over the 8-lane groups first: at F = 256 (ET = 128, one float4 per thread) they are the column ranges of the three MLPs -
    // threads 0-7: fuse_shape, 8-23: res_coeff, 24-31: fuse_det -, which the fixed-grid pair kernel scales separately (slots 14, 15)
#pragma unroll
    for (int off = 4; off > 0; off >>= 1)
#pragma unroll
        for (int it = 0; it < RPT; ++it) amax[it] = absmax_keep_nan(amax[it], __shfl_xor(amax[it], off, 64));
    float m_fs[RPT], m_rc[RPT];
#pragma unroll
    for (int it = 0; it < RPT; ++it) {
        const int base = threadIdx.x & 32;  // first lane of this row's 32 threads inside the wave
        m_fs[it] = __shfl(amax[it], base, 64);
        m_rc[it] = absmax_keep_nan(__shfl(amax[it], base + 8, 64), __shfl(amax[it], base + 16, 64));
    }
#pragma unroll
    for (int off = 16; off > 4; off >>= 1)
#pragma unroll
        for (int it = 0; it < RPT; ++it) amax[it] = absmax_keep_nan(amax[it], __shfl_xor(amax[it], off, 64));
    if (q == 0) {
#pragma unroll
        for (int it = 0; it < RPT; ++it) {
            const int row = g0 + (tid >> 5) + it * 2 * ER_WAVES;
            if (row < a.M) {
                float* h = a.hand[side] + (size_t)row * 16;
                h[13] = amax[it];
                if (d.ET == 128) {  // F = 256
                    h[14] = m_fs[it];
                    h[15] = m_rc[it];
                }
            }
        }
    }
    ER_STAMP(4);
}

bool embed_rows_serves(int F) { return F == 64 || F == 256 || F == 320; }

template <int NFB, int ER_ROWS, int ER_WAVES, int ER_NS>
static int launch_embed_shape(const EmbedArgs& a, hipStream_t st) {
    using S = ErShape<NFB, ER_ROWS, ER_WAVES, ER_NS>;
    const PairDims d(a.F);
    const size_t lds = (size_t)std::max(S::NS * S::SLOT, ER_ROWS * (d.ET + 8) * 4) + 7 * 104 * sizeof(float);
    (void)hipFuncSetAttribute((const void*)embed_rows_kernel<NFB, ER_ROWS, ER_WAVES, ER_NS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((embed_rows_kernel<NFB, ER_ROWS, ER_WAVES, ER_NS>), dim3(cdiv(a.M, ER_ROWS), 2), dim3(64 * ER_WAVES), lds, st, a);
    return check_launch("embed_rows");
}

int launch_embed_rows(const shasta_weights* w, const float* packed, const float* prev_feat, const float* feat, const float* prev_tab,
                      const float* det_tab, float* UP, float* UC, float* hand_prev, float* hand_det, int M, hipStream_t st) {
    const PackedLayout P(w->max_obj, w->num_feats, w->feat_dim);
    EmbedArgs a;
    a.x[0] = prev_feat;
    a.x[1] = feat;
    a.tab[0] = prev_tab;
    a.tab[1] = det_tab;
    a.emb[0] = UP;
    a.emb[1] = UC;
    a.hand[0] = hand_prev;
    a.hand[1] = hand_det;
    a.wp = reinterpret_cast<const uint32_t*>(packed + P.embp);
    a.packed = packed;
    a.M = M;
    a.F = w->feat_dim;
    a.nf = w->num_feats;
    a.N = w->max_obj;
    switch ((P.E12 + 31) / 32) {
        case 2: return launch_embed_shape<2, 256, 8, 0>(a, st);
        case 3: return launch_embed_shape<3, 128, 4, 2>(a, st);  // F = 256: two workgroups per CU
#ifdef SHASTA_ER_F320_128  // experiment (tools/build_variant.py): 128-row workgroups at F = 320 too - 83 KB each, still one per CU: 81 - 82 us against 75 - 77 at the car tables x 512
        case 4: return launch_embed_shape<4, 128, 4, 2>(a, st);
#else
        case 4: return launch_embed_shape<4, 256, 8, 0>(a, st);
#endif
    }
    set_error_msg("embed_rows: unsupported feat_dim");
    return SHASTA_E_ARG;
}

}  // namespace shasta
