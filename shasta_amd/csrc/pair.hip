// K4/K5: pair residual.  Restates det3d/models/tracker/shasta.py:277-319:
//   hand-designed residuals (:277-283), fuse_shape (:286-290), fuse_det (:293-307), res_coeff (:310-316) and the
//   weighted sum residual = alpha*fused + beta*dist + omega*shape (:319), for all T x D (track, detection) pairs.
// See pair_layout.hpp for the factorisation and the MFMA operand chaining.
//
// Kernels (per forward):
//   pack_pair_weights   once per weight load: fragments + factorised first-layer matrices
//   [gemm_nt_f32 x2]    UP = prev_feat . Wemb_prev^T ; UC = feat . Wemb_cur^T + b          (matrix cores)
//   row_prep            per table row: box columns of res_coeff.0 / fuse_det.0, log-dims, cos/sin; per column: the L2 norm
//                       over tracks (F.normalize, dim=1) of the squared box distances (the dim / rot terms and the
//                       normalised distance are evaluated inside the pair kernel)
//   pair_mfma4<F, WPB>  lane = pair: h1 = relu(UP[t]+UC[d]), layers 2-4 on v_mfma_f32_4x4x1_16B_f32 -> residual
// (Two other formulations - a 16x16x4 accumulator-chained MFMA kernel and a packed-VALU kernel with SGPR weights - measured
// within 4 % of this one and were removed from the product library after round 1; they are in the history of this file.)
#include "common.hpp"
#include "pair_layout.hpp"

namespace shasta {

// ------------------------------------------------------------------------------------------------------------
// weight packing
// ------------------------------------------------------------------------------------------------------------
struct PackArgs {
    shasta_linear fs[4], fd[3], rc[3], aff0;
    float* out;
    int N, nf, F;
};

__device__ __forceinline__ void layer_src(const PackArgs& a, int l, const float*& W, const float*& b, int& ldw) {
    const PairDims d(a.F);
    switch (l) {
        case L_FS2: W = a.fs[1].weight; b = a.fs[1].bias; ldw = d.H1; break;
        case L_FS3: W = a.fs[2].weight; b = a.fs[2].bias; ldw = d.H2; break;
        case L_FS4: W = a.fs[3].weight; b = a.fs[3].bias; ldw = d.H3; break;
        case L_RC2: W = a.rc[1].weight; b = a.rc[1].bias; ldw = d.R1; break;
        case L_RC3: W = a.rc[2].weight; b = a.rc[2].bias; ldw = d.R2; break;
        case L_FD2: W = a.fd[1].weight; b = a.fd[1].bias; ldw = 32; break;
        default:    W = a.fd[2].weight; b = a.fd[2].bias; ldw = 8; break;
    }
}

__global__ void pack_pair_weights_kernel(PackArgs a) {
    const PackedLayout P(a.N, a.nf, a.F);
    const PairDims d(a.F);
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    const int nth = gridDim.x * blockDim.x;
    const int F = a.F, nf = a.nf;
    // 4x4x1 A operands: [ob][kg][i][kk], then bias [ob][i]
    for (int l = 0; l < L_COUNT; ++l) {
        const LayerDesc L = layer_desc(F, l);
        const float *W, *b;
        int ldw;
        layer_src(a, l, W, b, ldw);
        const int nob = a4_nob(F, l), kg = a4_kg(F, l);
        float* o = a.out + P.a4 + a4_offset(F, l);
        for (int e = tid; e < nob * kg * 16; e += nth) {
            const int kk = e & 3, i = (e >> 2) & 3, g = (e >> 4) % kg, ob = (e >> 4) / kg;
            const int f = 4 * ob + i, k = 4 * g + kk;
            o[e] = (f < L.hout && k < L.kin) ? W[(size_t)f * ldw + k] : 0.0f;
        }
        for (int e = tid; e < nob * 4; e += nth) o[nob * kg * 16 + e] = e < L.hout ? b[e] : 0.0f;
    }
    // factorised first layers.  Input column order of the reference concatenations:
    //   fuse_shape.0 : [prev_feat F | feat F]                                  (shasta.py:286)
    //   res_coeff.0  : [prev_feat F | prev_box nf | feat F | det_box nf]        (shasta.py:310-312)
    //   fuse_det.0   : [prev_box nf | det_box nf]                               (shasta.py:303)
    const int E12 = P.E12;
    for (int e = tid; e < E12 * F; e += nth) {
        const int j = e / F, k = e % F;
        float wp, wc;
        if (j < d.H1) {
            wp = a.fs[0].weight[(size_t)j * 2 * F + k];
            wc = a.fs[0].weight[(size_t)j * 2 * F + F + k];
        } else {
            const size_t row = (size_t)(j - d.H1) * (2 * F + 2 * nf);
            wp = a.rc[0].weight[row + k];
            wc = a.rc[0].weight[row + F + nf + k];
        }
        a.out[P.wemb_prev + e] = wp;
        a.out[P.wemb_cur + e] = wc;
    }
    for (int e = tid; e < (E12 + 3) / 4 * 4; e += nth)
        a.out[P.bemb_cur + e] = e < d.H1 ? a.fs[0].bias[e] : (e < E12 ? a.rc[0].bias[e - d.H1] : 0.0f);
    for (int e = tid; e < (d.R1 + 32) * 8; e += nth) {
        const int j = e >> 3, c = e & 7;
        float wp = 0.0f, wc = 0.0f;
        if (c < nf) {
            if (j < d.R1) {
                const size_t row = (size_t)j * (2 * F + 2 * nf);
                wp = a.rc[0].weight[row + F + c];
                wc = a.rc[0].weight[row + 2 * F + nf + c];
            } else {
                const size_t row = (size_t)(j - d.R1) * (2 * nf);
                wp = a.fd[0].weight[row + c];
                wc = a.fd[0].weight[row + nf + c];
            }
        }
        a.out[P.wbox_prev + e] = wp;
        a.out[P.wbox_cur + e] = wc;
    }
    for (int e = tid; e < 32; e += nth) a.out[P.bbox_cur + e] = a.fd[0].bias[e];
    // aff.0.weight (128, D) -> (128, Dp) zero padded so that its rows are 16-byte aligned
    for (int e = tid; e < 128 * P.Dp; e += nth) {
        const int j = e / P.Dp, k = e % P.Dp;
        a.out[P.aff0 + e] = k < P.D ? a.aff0.weight[(size_t)j * P.D + k] : 0.0f;
    }
}

// ------------------------------------------------------------------------------------------------------------
// row_prep: everything the pair kernel needs per table row or per column besides the GEMM part of the row embeddings, in ONE
// launch of three block roles.
//  blocks [0, nrow): 32 table rows of one side each (first half: prev rows, second half: cur rows), 32 threads per row, one
//   float4 of outputs per thread:
//   threads 0 .. (R1+32)/4-1:  E[row][H1 + j]       += Wbox[j][:nf] . box[:nf]                 (res_coeff.0 box columns)
//                              E[row][H1 + R1 + j]   = Wbox[R1 + j][:nf] . box[:nf] (+ bias)   (fuse_det.0)
//  blocks [nrow, nrow + nhand): one thread per table row:
//                              hand[row] = [box7 (slots >= num_feats zero), 0, log(w+eps), log(l+eps), log(h+eps), cos(yaw), sin(yaw), max |E[row]|, 0, 0]
//                              (slot 13, the row's largest embedding magnitude, is written by the first role)
//  the remaining blocks: col_norm (shasta.py:278-279): d2[t][d] = sum_{k<nf} (prev_k - det_k)^2,
//   denom[d] = max(||d2[:, d]||_2, 1e-12) (F.normalize acts along dim=1 = tracks); 16 detections x 16 track groups per block,
//   read straight from the (B, T, 8) box tables (4 tracks in flight per thread, fixed-order reduction).
// ------------------------------------------------------------------------------------------------------------
struct RowPrepArgs {
    const float* packed;
    const float* tab[2];  // [0] prev_tab, [1] det_tab   (B, T, 8)
    float* emb[2];        // [0] UP, [1] UC             (B, T, ET)
    float* hand[2];       // (B, T, 16)
    float* denom;         // (B, D)
    int B, T, N, nf, F, nrow_blocks, nhand_blocks, dblocks, dper;  // dper: detections per thread of the column-norm role (1 or 4)
};

// column norms (shasta.py:278-279) of 16 * ND detections of one frame per workgroup: ND per thread - a track read from LDS serves ND
// detections and the frame's previous boxes are staged once per 16 ND columns (ND = 4 from 2048 workgroups on: the staging latency was
// most of a block's time; ND = 1 keeps small batches spread over more workgroups)
template <int ND, int NF>  // NF: num_feats as a compile-time constant (no branch per column in the inner loop), 0 = read it from the arguments
__device__ __forceinline__ void col_norm_block(const RowPrepArgs& a, int cb, int tid) {
    const int T = a.T, D = a.T, nf = NF ? NF : a.nf;
    const int b = cb / a.dblocks, dl = tid & 15, tg = tid >> 4;
    const int d0 = (cb % a.dblocks) * (16 * ND) + dl;  // this thread's detections: d0, d0 + 16, ...
    float db[ND][7];
#pragma unroll
    for (int j = 0; j < ND; ++j) {
        const f32x4* h = reinterpret_cast<const f32x4*>(a.tab[1] + ((size_t)b * D + min(d0 + 16 * j, D - 1)) * 8);
        const f32x4 x = h[0], c = h[1];
        db[j][0] = x[0]; db[j][1] = x[1]; db[j][2] = x[2]; db[j][3] = x[3]; db[j][4] = c[0]; db[j][5] = c[1]; db[j][6] = c[2];
        // columns >= nf take no part: zero on both sides (here and in the staged previous boxes), so that their difference is +0 and
        // the loop below runs over all seven without a mask - a select per column and detection was 40 % of its instructions
#pragma unroll
        for (int k = 0; k < 7; ++k)
            if (k >= nf) db[j][k] = 0.0f;
    }
    // the frame's previous boxes pass through LDS in chunks of 512 rows (one coalesced copy per chunk, then broadcast reads)
    __shared__ __attribute__((aligned(16))) f32x4 sp[2 * 512];
    const f32x4* hp = reinterpret_cast<const f32x4*>(a.tab[0] + (size_t)b * T * 8);
    // Two tracks per step as the halves of packed fp32 operations (v_pk_add / v_pk_mul).  Every product and sum is rounded separately
    // exactly as in the scalar form; a column's sum of squares is the sum of the two halves' running sums (tracks tg, tg + 32, ... and
    // tg + 16, tg + 48, ...), then of the 16 track groups in order.
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 ssq2[ND];
#pragma unroll
    for (int j = 0; j < ND; ++j) ssq2[j] = f2{0.0f, 0.0f};
    for (int t0 = 0; t0 < T; t0 += 512) {
        const int nt = min(512, T - t0);
        if (t0) __syncthreads();
        for (int e = tid; e < 2 * nt; e += 256) {
            f32x4 v = hp[(size_t)t0 * 2 + e];
            const int k0 = (e & 1) * 4;  // box columns k0 .. k0 + 3 of the row
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (k0 + q >= nf) v[q] = 0.0f;
            sp[e] = v;
        }
        __syncthreads();
        for (int t = tg; t < nt; t += 32) {
            const int tb = t + 16;
            const bool two = tb < nt;
            const f32x4 x0 = sp[2 * t], c0 = sp[2 * t + 1];
            const f32x4 x1 = sp[2 * (two ? tb : t)], c1 = sp[2 * (two ? tb : t) + 1];
            const f2 p[7] = {f2{x0[0], x1[0]}, f2{x0[1], x1[1]}, f2{x0[2], x1[2]}, f2{x0[3], x1[3]}, f2{c0[0], c1[0]}, f2{c0[1], c1[1]},
                             f2{c0[2], c1[2]}};
#pragma unroll
            for (int j = 0; j < ND; ++j) {
                f2 d2 = {0.0f, 0.0f};
#pragma unroll
                for (int k = 0; k < (NF ? NF : 7); ++k) {
                    const f2 df = p[k] - f2{db[j][k], db[j][k]};
                    d2 += df * df;
                }
                f2 sq = d2 * d2;
                if (!two) sq[1] = 0.0f;
                ssq2[j] += sq;
            }
        }
    }
    __shared__ float red4[ND][16][17];
#pragma unroll
    for (int j = 0; j < ND; ++j) red4[j][dl][tg] = ssq2[j][0] + ssq2[j][1];
    __syncthreads();
    if (tid < 16 * ND) {
        const int j = tid >> 4, l = tid & 15, dd = (cb % a.dblocks) * (16 * ND) + 16 * j + l;
        if (dd < D) {
            float tot = 0.0f;
#pragma unroll
            for (int i = 0; i < 16; ++i) tot += red4[j][l][i];  // fixed order
            a.denom[(size_t)b * D + dd] = fmaxf(sqrtf(tot), 1e-12f);
        }
    }
}

// the column-norm role alone (large batches: 4 detections per thread): its own register budget and LDS (118 registers, 21 KB: four
// wavefronts per SIMD) - inside row_prep_kernel the three roles share one allocation (130 registers, 41 KB, three wavefronts per SIMD).
// Measured equal to the fused form (0.031 hand rows + 0.135 against 0.166 ms per 1024 frame-pairs): the role is bound by its VALU
// instructions - 124 per 8 pairs, of which the packed subtract / multiply / add of 7 columns need 92 - not by occupancy.
__global__ __launch_bounds__(256) void col_norm_kernel(RowPrepArgs a) { col_norm_block<4, 0>(a, blockIdx.x, threadIdx.x); }

__global__ __launch_bounds__(256) void row_prep_kernel(RowPrepArgs a) {
    const int tid = threadIdx.x;
    if ((int)blockIdx.x < a.nrow_blocks) {
        // 32 table rows per block (4 rounds of 8 rows x 32 threads).  The (R1+32) x 8 box-column weights of this side are staged
        // through LDS TRANSPOSED, wT[c][j]: a thread then reads its 4 consecutive outputs of column c as one conflict-free
        // float4 (straight from the packed [j][8] table the 24 active lanes of a row hit 24 different cache lines per load).
        __shared__ __attribute__((aligned(16))) float wT[7 * 104];
        const PackedLayout P(a.N, a.nf, a.F);
        const PairDims d(a.F);
        const int J = d.R1 + 32;  // <= 104, a multiple of 4 for every supported F
        // a block never straddles the two sides: nrow_blocks = 2 * ceil(B*T / 32), block i < half handles prev rows
        const int half = a.nrow_blocks / 2;
        const int which = (int)blockIdx.x >= half;
        const int side_row0 = (blockIdx.x - which * half) * 32;
        {
            const float* wb = a.packed + (which ? P.wbox_cur : P.wbox_prev);
            for (int e = tid; e < J * 8; e += 256) {
                const int j = e >> 3, c = e & 7;
                if (c < 7) wT[c * 104 + j] = wb[e];
            }
        }
        __syncthreads();
        const int q = tid & 31, nq = J / 4;
        for (int it = 0; it < 4; ++it) {
            const int row = side_row0 + it * 8 + (tid >> 5);
            if (row >= a.B * a.T) continue;  // the 32 threads of a row leave together (the shuffles below stay inside the row)
            float amax = 0.0f;  // largest |E[row][.]| over the columns this thread sees
            if (q < nq) {
                const f32x4* bp = reinterpret_cast<const f32x4*>(a.tab[which] + (size_t)row * 8);
                const f32x4 b0 = bp[0], b1 = bp[1];
                const float bx[7] = {b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2]};
                f32x4 s = {0.0f, 0.0f, 0.0f, 0.0f};  // columns >= nf are packed as 0; k-ordered fmaf chain per output
#pragma unroll
                for (int c = 0; c < 7; ++c) {
                    const f32x4 w4 = *reinterpret_cast<const f32x4*>(&wT[c * 104 + 4 * q]);
                    s[0] = fmaf(w4[0], bx[c], s[0]); s[1] = fmaf(w4[1], bx[c], s[1]);
                    s[2] = fmaf(w4[2], bx[c], s[2]); s[3] = fmaf(w4[3], bx[c], s[3]);
                }
                f32x4* e = reinterpret_cast<f32x4*>(a.emb[which] + (size_t)row * d.ET + d.H1 + 4 * q);
                const int j = 4 * q;
                if (j < d.R1) {
                    f32x4 v = *e;
                    s[0] += v[0]; s[1] += v[1]; s[2] += v[2]; s[3] += v[3];
                } else if (which) {
                    const f32x4 bb = *reinterpret_cast<const f32x4*>(a.packed + P.bbox_cur + (j - d.R1));
                    s[0] += bb[0]; s[1] += bb[1]; s[2] += bb[2]; s[3] += bb[3];
                }
                *e = s;
                amax = absmax_keep_nan(absmax_keep_nan(fabsf(s[0]), fabsf(s[1])), absmax_keep_nan(fabsf(s[2]), fabsf(s[3])));
            } else {  // the spare threads of the row look at the columns the GEMM alone wrote (fuse_shape part, [0, H1))
                const f32x4* e = reinterpret_cast<const f32x4*>(a.emb[which] + (size_t)row * d.ET);
                for (int c4 = q - nq; c4 < d.H1 / 4; c4 += 32 - nq) {
                    const f32x4 v = e[c4];
                    amax = absmax_keep_nan(amax, absmax_keep_nan(absmax_keep_nan(fabsf(v[0]), fabsf(v[1])), absmax_keep_nan(fabsf(v[2]), fabsf(v[3]))));
                }
            }
#pragma unroll
            for (int off = 16; off > 0; off >>= 1) amax = absmax_keep_nan(amax, __shfl_xor(amax, off, 64));
            // slot 13 of the hand row: the row's largest embedding magnitude (range scaling of the fp16 pair kernel, pair_f16.hip)
            if (q == 0) a.hand[which][(size_t)row * 16 + 13] = amax;
        }
        return;
    }
    if ((int)blockIdx.x < a.nrow_blocks + a.nhand_blocks) {
        // hand rows, one THREAD per table row: the five transcendentals are long instruction sequences, so they run with all 64
        // lanes of a wave busy (a wave per row spent its time here with 2-4 active lanes)
        const int item = (blockIdx.x - a.nrow_blocks) * 256 + tid;
        if (item >= 2 * a.B * a.T) return;
        const int which = item / (a.B * a.T), row = item % (a.B * a.T);
        const f32x4* bp = reinterpret_cast<const f32x4*>(a.tab[which] + (size_t)row * 8);
        const f32x4 b0 = bp[0], b1 = bp[1];
        f32x4* h = reinterpret_cast<f32x4*>(a.hand[which] + (size_t)row * 16);
        // (box slots >= num_feats are zero in both tables' hand rows: pair_layout.hpp, hand_dist)
        const int nf = a.nf;
        h[0] = f32x4{b0[0], nf > 1 ? b0[1] : 0.0f, nf > 2 ? b0[2] : 0.0f, nf > 3 ? b0[3] : 0.0f};
        h[1] = f32x4{nf > 4 ? b1[0] : 0.0f, nf > 5 ? b1[1] : 0.0f, nf > 6 ? b1[2] : 0.0f, 0.0f};
        h[2] = f32x4{logf(b0[3] + 1e-10f), logf(b1[0] + 1e-10f), logf(b1[1] + 1e-10f), cosf(b1[2])};
        // slots 13 - 15 belong to the row embeddings (embed_rows.hip / the row role above): the row's largest |E| over all columns and,
        // from the fused kernel at F = 256, over the fuse_shape and res_coeff column ranges
        a.hand[which][(size_t)row * 16 + 12] = sinf(b1[2]);
        return;
    }
    // ---- column norms ----
    const int cb = blockIdx.x - a.nrow_blocks - a.nhand_blocks;
    if (a.dper == 4) {
        // (num_feats stays a run-time value: with it compiled in - no branch per column - the kernel took 0.27 instead of 0.21 ms per
        // 1024 frame-pairs in an alternating A/B on one box, tools/gpu_kernel_ab.sh)
        col_norm_block<4, 0>(a, cb, tid);
    } else {
        col_norm_block<1, 0>(a, cb, tid);
    }
}

// ------------------------------------------------------------------------------------------------------------
// pair_mfma4: lane = pair, layers 2-4 on v_mfma_f32_4x4x1_16B_f32 (16 independent 4x4 blocks per instruction:
// block = 4 consecutive lanes = 4 pairs, 4 output features, K = 1; 8.4 cycles measured, the same MAC rate as 16x16x4).
//   D[i][pair] += A[i] * B[pair] : A = W[4*ob + i][k] (lane l supplies row i = l & 3), B = h[k] of the lane's own pair.
// The result registers of a lane are 4 output features of ITS pair, so the next layer consumes them directly and the
// lane = pair layout of the factorised first layer (UP[t] scalar, UC[d] per lane) is kept end to end.
// Against the 16x16x4 chain: output widths only round up to 4 (not 16): 512 MFMAs x 8.4 = 4.3k cycles per 64 pairs
// instead of 4 x 44 x 32 = 5.6k, and ~330 VALU instructions per 64 pairs instead of ~540.
// ------------------------------------------------------------------------------------------------------------
#define MFMA4(a, b, c) __builtin_amdgcn_mfma_f32_4x4x1f32((a), (b), (c), 0, 0, 0)

template <int F, int L>
struct A4 {
    static constexpr LayerDesc D = layer_desc(F, L);
    static constexpr int NOB = a4_nob(F, L), KG = a4_kg(F, L), OFF = a4_offset(F, L), KIN = D.kin, BIAS = NOB * KG * 16;
};

// DT = detections per wave: 64 (one track per iteration), or 32 with TWO tracks per iteration (lanes 32..63 take track t + 1): the
// same instruction stream covers 2 x 32 pairs, so a table of D = 92 rows (the shipped car configuration: max_obj 90) fills 3 x 32
// lanes-of-work at 96 % instead of 2 x 64 at 72 %, and D = 22 (bus) one tile at 69 % instead of 34 %.
#ifdef PAIR_STAMP  // diagnostic build only (tools/pair_clock.py, as in pair_f16.hip).  Stamped at N = 500, 512 frame-pairs: two workgroups
// per CU; a workgroup's waves 4 .. 7 end 133 us after its waves 0 .. 3 (of 700 us), but the other workgroup of the CU fills the
// SIMDs meanwhile: dealing fewer tracks to the late waves (900 / 850 / 800 per 1000) changed neither this kernel's 5.96 ms nor the car
// configuration's 0.31 ms, so the tracks stay dealt evenly here.
__device__ unsigned long long g_pair_stamp[4096][8][4];
#endif
// (two workgroups per CU = 4 waves per SIMD need at most 128 registers: stated, because <320, 8, 32> sits at the limit - 126, and 130
// with two more live values, which cost 7 % of the car configuration's pair kernel when it happened)
template <int F, int WPB, int DT = 64>
__global__ __launch_bounds__(64 * WPB) __attribute__((amdgpu_waves_per_eu(4))) void pair_mfma4_kernel(const float* __restrict__ packed, const float* __restrict__ UP,
                                                         const float* __restrict__ UC, const float* __restrict__ hand_prev,
                                                         const float* __restrict__ hand_det, const float* __restrict__ denom,
                                                         float* __restrict__ residual, int T, int D, int ld, int nf,
                                                         int TW) {
    constexpr PairDims dm(F);
    constexpr int H1 = dm.H1, R1 = dm.R1, ET = dm.ET, US = ET + 4;
    constexpr int NA4 = a4_total(F);
    static_assert(DT == 64 || DT == 32, "64 detections x 1 track or 32 detections x 2 tracks per wave");
    constexpr int TPI = 64 / DT;       // tracks per iteration
    constexpr int RING = 3 * TPI;      // UP row slots per wave: the rows in use, and two iterations ahead
    extern __shared__ __attribute__((aligned(16))) float s_dyn4[];
    float* s_uc = s_dyn4;              // [DT][US]
    float* s_a4 = s_dyn4 + DT * US;    // [NA4]
    // UP rows (the per-track half of the first layers) of the next two tracks, per wave: [WPB][3 slots][256 floats].  They
    // were scalar loads before; SMEM and LDS share lgkmcnt, so every s_load had to be waited for with lgkmcnt(0) before the
    // next LDS result could be used - nine full scalar-memory latencies per track (measured: 25 % of the wave time parked).
    // An LDS-DMA (counted on vmcnt) fetches row t+2 while row t is in use; the values are then read back as LDS broadcasts.
    float* s_up = s_dyn4 + DT * US + ((NA4 + 3) & ~3);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int dlane = lane & (DT - 1), th = lane / DT;  // detection of the tile, track of the iteration
#ifdef PAIR_STAMP
    const unsigned long long st0 = __builtin_amdgcn_s_memtime(), sr0 = __builtin_amdgcn_s_memrealtime();
#endif
    int lbx, lby, b;
    xcd_logical_block(lbx, lby, b);  // the detection tiles of a frame on one XCD (common.hpp)
    const int d0 = lbx * DT;
    const int d = d0 + dlane, dcl = min(d, D - 1);
    const PackedLayout P(0, 0, F);
    {
        const f32x4* src = reinterpret_cast<const f32x4*>(UC);
#pragma unroll 4
        for (int e = tid; e < DT * (ET / 4); e += 64 * WPB) {
            const int r = e / (ET / 4), c = e - r * (ET / 4);
            *reinterpret_cast<f32x4*>(&s_uc[r * US + 4 * c]) = src[((size_t)b * D + min(d0 + r, D - 1)) * (ET / 4) + c];
        }
        const f32x4* asrc = reinterpret_cast<const f32x4*>(packed + P.a4);
#pragma unroll 2
        for (int e = tid; e < NA4 / 4; e += 64 * WPB) reinterpret_cast<f32x4*>(s_a4)[e] = asrc[e];
    }
    float hd[12];
    {
        const f32x4* h = reinterpret_cast<const f32x4*>(hand_det + ((size_t)b * D + dcl) * 16);
        const f32x4 a = h[0], c = h[1], e = h[2], g = h[3];
        hd[0] = a[0]; hd[1] = a[1]; hd[2] = a[2]; hd[3] = a[3]; hd[4] = c[0]; hd[5] = c[1]; hd[6] = c[2];
        hd[7] = e[0]; hd[8] = e[1]; hd[9] = e[2]; hd[10] = e[3]; hd[11] = g[0];
    }
    const float dnm = denom[(size_t)b * D + dcl], rdn = 1.0f / dnm;
    __syncthreads();
    const float* ucrow = s_uc + dlane * US;
    typedef __attribute__((address_space(3))) float lfloat;
    typedef __attribute__((address_space(3))) f32x4 lf32x4;
    // LDS byte address of this lane's row i = lane & 3 inside every [i][kk] group of the A table
    const unsigned arow_base = (unsigned)(unsigned long long)(s_a4 + (lane & 3) * 4);
    const unsigned abias_base = (unsigned)(unsigned long long)(s_a4 + (lane & 3));
    const f32x4 zero4 = {0, 0, 0, 0};

    const int t_beg = (lby * WPB + wid) * TW;
    const int t_end = min(T, t_beg + TW);
    float* my_up = s_up + wid * (RING * 256);
    // lanes [0, ET/4): the UP row; the next 4 lanes: the 16-float hand row of the same track (lands right behind it); the
    // remaining lanes repeat the last UP chunk (their LDS words are unused)
    const bool hp_lane = lane >= ET / 4 && lane < ET / 4 + 4;
    const int up_lane = 4 * min(lane, ET / 4 - 1), hp_off = 4 * (lane - ET / 4);
    auto dma_up = [&](int row, int slot) {
        const size_t r = (size_t)b * T + min(row, T - 1);
        const float* src = hp_lane ? hand_prev + r * 16 + hp_off : UP + r * ET + up_lane;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(my_up + slot * 256), 16, 0, 0);
    };
    if (t_beg < t_end) {
#pragma unroll
        for (int i = 0; i < 2 * TPI; ++i) dma_up(t_beg + i, i);
    }
    for (int t = t_beg; t < t_end; t += TPI) {
        // the rows of this and of the next iteration were requested at least one whole iteration ago (or in the prologue): nothing
        // younger is in flight
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int i = 0; i < TPI; ++i) dma_up(t + 2 * TPI + i, (t - t_beg + 2 * TPI + i) % RING);
        const int tt = t + th;  // this lane's track
        unsigned upo = (unsigned)(unsigned long long)(my_up + ((t - t_beg + th) % RING) * 256);
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
        // the A table is loop invariant: an opaque copy of its address per track keeps the 128 ds_read_b128 inside the
        // loop instead of 512 hoisted registers
        unsigned ao = arow_base, bo = abias_base;
        asm volatile("" : "+v"(ao), "+v"(bo));
        const lfloat* arow = (const lfloat*)(unsigned long long)ao;
        const lfloat* abias = (const lfloat*)(unsigned long long)bo;

        // bias: acc[ob] = bias[4*ob + i] * 1
        auto init = [&](auto tag, f32x4* acc) {
            using AL = decltype(tag);
#pragma unroll
            for (int ob = 0; ob < AL::NOB; ++ob) acc[ob] = MFMA4(abias[AL::OFF + AL::BIAS + ob * 4], 1.0f, zero4);
        };
        // layer 1 (factorised) feeding layer 2.  (Forming UP[t] + UC[d] on the matrix pipe - one 4x4x1 MFMA with B = 1.0 and
        // C = the UC float4 per 4 features instead of 4 v_add - was measured in round 2: 7.15 - 7.27 ms against 6.31 - 6.51 ms
        // for this form on the same box, 512 frame-pairs per launch; the adds stay on the VALU.)
        auto layer12 = [&](auto tag, int seg, f32x4* acc) {
            using AL = decltype(tag);
            init(tag, acc);
#pragma unroll
            for (int kg = 0; kg < AL::KG; ++kg) {
                const f32x4 u = *reinterpret_cast<const f32x4*>(ucrow + seg + 4 * kg);
                const f32x4 upv = *reinterpret_cast<const lf32x4*>(up + seg + 4 * kg);  // one address per track: LDS broadcast(s)
                f32x4 a4[AL::NOB];
#pragma unroll
                for (int ob = 0; ob < AL::NOB; ++ob) a4[ob] = *reinterpret_cast<const lf32x4*>(arow + AL::OFF + (ob * AL::KG + kg) * 16);
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    if (4 * kg + kk < AL::KIN) {
                        const float h = relu_nan(upv[kk] + u[kk]);
#pragma unroll
                        for (int ob = 0; ob < AL::NOB; ++ob) acc[ob] = MFMA4(a4[ob][kk], h, acc[ob]);
                    }
                }
            }
        };
        // acc = bias + W . relu(in): the k-th input is register k & 3 of the previous layer's block k >> 2
        auto layer = [&](auto tag, const f32x4* in, f32x4* acc) {
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
                        const float h = relu_nan(in[kg][kk]);
#pragma unroll
                        for (int ob = 0; ob < AL::NOB; ++ob) acc[ob] = MFMA4(a4[ob][kk], h, acc[ob]);
                    }
                }
            }
        };
        f32x4 a_rc2[A4<F, L_RC2>::NOB], a_rc3[A4<F, L_RC3>::NOB];
        f32x4 a_fs2[A4<F, L_FS2>::NOB], a_fs3[A4<F, L_FS3>::NOB], a_fs4[A4<F, L_FS4>::NOB];
        f32x4 a_fd2[A4<F, L_FD2>::NOB], a_fd3[A4<F, L_FD3>::NOB];
        layer12(A4<F, L_RC2>{}, H1, a_rc2);
        layer12(A4<F, L_FS2>{}, 0, a_fs2);
        layer12(A4<F, L_FD2>{}, H1 + R1, a_fd2);
        layer(A4<F, L_RC3>{}, a_rc2, a_rc3);
        layer(A4<F, L_FS3>{}, a_fs2, a_fs3);
        layer(A4<F, L_FD3>{}, a_fd2, a_fd3);
        layer(A4<F, L_FS4>{}, a_fs3, a_fs4);

        // ---- hand-designed residual (shasta.py:277-283) ----
        const float dist = hand_dist(hp, hd, dnm, rdn);
        // ---- combine (shasta.py:316-319) ----
        const float res = (a_rc3[0][0] * a_fd3[0][0] + a_rc3[0][1] * dist) + a_rc3[0][2] * a_fs4[0][0];
        if (d < D && tt < t_end) residual[((size_t)b * T + tt) * ld + d] = res;
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

size_t pair_workspace_bytes(int B, int N, int F) {
    const PairDims d(F);
    const int T = N + 2, Dp = (T + 3) / 4 * 4;
    size_t s = 0;
    s += 2 * align_up((size_t)B * T * d.ET * sizeof(float), 256);  // UP, UC
    s += 2 * align_up((size_t)B * T * 16 * sizeof(float), 256);    // hand tables
    s += align_up((size_t)B * T * sizeof(float), 256);             // column norms
    (void)Dp;
    return s;
}

int pair_f16_pack(const shasta_weights* w, float* out, hipStream_t st);
int launch_pair_f16(const float* packed, const float* p16, const float* UP, const float* UC, const float* hand_prev,
                    const float* hand_det, const float* denom, float* residual, int B, int T, int D, int ld, int nf, bool grid,
                    hipStream_t st);
bool pair_f16w_serves(int F);
int pair_f16w_pack(const shasta_weights* w, float* out, hipStream_t st);
int launch_pair_f16w(const float* packed, const float* p16, const float* UP, const float* UC, const float* hand_prev,
                     const float* hand_det, const float* denom, float* residual, int B, int T, int D, int ld, int F, hipStream_t st);
int launch_gemm_nt(const float* A, int lda, const float* W, int ldw, const float* bias, float* C, int ldc, int M,
                   int N, int K, int act, hipStream_t st);
int launch_gemm_nt_dual(const float* A0, const float* W0, const float* bias0, float* C0, const float* A1, const float* W1,
                        const float* bias1, float* C1, int lda, int ldw, int ldc, int M, int N, int K, int act, hipStream_t st);
int launch_gemm_nt_pieces(const float* A0, const float* W0, const float* bias0, float* C0, const float* A1, const float* W1,
                          const float* bias1, float* C1, int lda, int ldw, int ldc, int M, int N, int K, int act, hipStream_t st);

int embed_pack(const shasta_weights* w, float* packed, hipStream_t st);
bool embed_rows_serves(int F);
int launch_embed_rows(const shasta_weights* w, const float* packed, const float* prev_feat, const float* feat, const float* prev_tab,
                      const float* det_tab, float* UP, float* UC, float* hand_prev, float* hand_det, int M, hipStream_t st);

int pair_residual(const shasta_weights* w, const float* packed, int B, const float* feat, const float* prev_feat,
                  const float* det_tab, const float* prev_tab, float* residual, int ld, void* ws, size_t ws_bytes,
                  hipStream_t st, hipEvent_t ev0, hipEvent_t ev1) {
    const int N = w->max_obj, F = w->feat_dim, nf = w->num_feats, T = N + 2, D = N + 2;
    const PairDims d(F);
    const PackedLayout P(N, nf, F);
    if (ws_bytes < pair_workspace_bytes(B, N, F)) {
        set_error_msg("pair_residual: workspace too small");
        return SHASTA_E_WORKSPACE;
    }
    if (B == 0) return SHASTA_OK;
    char* base = static_cast<char*>(ws);
    float* UP = reinterpret_cast<float*>(base);
    base += align_up((size_t)B * T * d.ET * sizeof(float), 256);
    float* UC = reinterpret_cast<float*>(base);
    base += align_up((size_t)B * T * d.ET * sizeof(float), 256);
    float* hand_prev = reinterpret_cast<float*>(base);
    base += align_up((size_t)B * T * 16 * sizeof(float), 256);
    float* hand_det = reinterpret_cast<float*>(base);
    base += align_up((size_t)B * T * 16 * sizeof(float), 256);
    float* denom = reinterpret_cast<float*>(base);

    // row embeddings UP / UC: from 8192 table rows one fused kernel per launch forms the feature part on the bf16-piece matrix path,
    // adds the box columns and takes the row maxima (embed_rows.hip); below that the 64-row f32 GEMM tiles fill the chip better and
    // row_prep's row role adds the box columns.  SHASTA_OPT_F32_EMBED_GEMM keeps the f32 GEMM.
    const bool gemm_f32 = (w->options & SHASTA_OPT_F32_EMBED_GEMM) != 0;
    const bool fused = B * T >= 8192 && !gemm_f32 && embed_rows_serves(F) && ((uintptr_t)feat | (uintptr_t)prev_feat) % 16 == 0;
    int rc;
    if (fused)
        rc = launch_embed_rows(w, packed, prev_feat, feat, prev_tab, det_tab, UP, UC, hand_prev, hand_det, B * T, st);
    else
        rc = launch_gemm_nt_dual(prev_feat, packed + P.wemb_prev, nullptr, UP, feat, packed + P.wemb_cur, packed + P.bemb_cur, UC, F, F,
                                 d.ET, B * T, P.E12, F, 0, st);
    if (rc) return rc;
    RowPrepArgs rp;
    rp.packed = packed;
    rp.tab[0] = prev_tab;
    rp.tab[1] = det_tab;
    rp.emb[0] = UP;
    rp.emb[1] = UC;
    rp.hand[0] = hand_prev;
    rp.hand[1] = hand_det;
    rp.denom = denom;
    rp.B = B;
    rp.T = T;
    rp.N = N;
    rp.nf = nf;
    rp.F = F;
    rp.nrow_blocks = fused ? 0 : 2 * cdiv(B * T, 32);  // the fused kernel has added the box columns already
    rp.nhand_blocks = cdiv(2 * B * T, 256);
    rp.dper = cdiv(D, 16) * B >= 2048 ? 4 : 1;
    rp.dblocks = cdiv(D, 16 * rp.dper);
    if (rp.dper == 4) {  // large batches: the column norms as a kernel of their own
        hipLaunchKernelGGL(row_prep_kernel, dim3(rp.nrow_blocks + rp.nhand_blocks), dim3(256), 0, st, rp);
        rc = check_launch("row_prep");
        if (rc) return rc;
        hipLaunchKernelGGL(col_norm_kernel, dim3(rp.dblocks * B), dim3(256), 0, st, rp);
        rc = check_launch("col_norm");
    } else {
        hipLaunchKernelGGL(row_prep_kernel, dim3(rp.nrow_blocks + rp.nhand_blocks + rp.dblocks * B), dim3(256), 0, st, rp);
        rc = check_launch("row_prep");
    }
    if (rc) return rc;
    if ((w->options & SHASTA_OPT_F16X2_PAIR) && F == 256) {
        // second layers of the three pair MLPs on the f16 matrix path (pair_f16.hip), everything else as below
        if (ev0) (void)hipEventRecord(ev0, st);
        // SHASTA_OPT_F16GRID_PAIR: the fixed-grid form (needs the per-MLP row maxima that only the fused row-embedding kernel writes)
        const bool grid = (w->options & SHASTA_OPT_F16GRID_PAIR) != 0 && fused;
        rc = launch_pair_f16(packed, packed + P.p16, UP, UC, hand_prev, hand_det, denom, residual, B, T, D, ld, nf, grid, st);
        if (ev1) (void)hipEventRecord(ev1, st);
        return rc;
    }
    if ((w->options & SHASTA_OPT_F16X2_PAIR) && pair_f16w_serves(F)) {
        // F = 320 (every shipped class configuration): the same arithmetic on 32-wide tiles (pair_f16w.hip)
        if (ev0) (void)hipEventRecord(ev0, st);
        rc = launch_pair_f16w(packed, packed + P.p16w, UP, UC, hand_prev, hand_det, denom, residual, B, T, D, ld, F, st);
        if (ev1) (void)hipEventRecord(ev1, st);
        if (rc != SHASTA_E_UNSUPPORTED) return rc;  // a device that does not grant its LDS: the f32 kernel below serves the call
    }
    // lane = pair, 4x4x1 MFMA.  The 8 waves of a workgroup share one 64-detection UC tile and take different track ranges
    // (two workgroups per CU by LDS, 113 VGPRs in the VGPR MFMA form: 4 waves per SIMD).
    constexpr int wpb = 8;
    // 32-detection tiles (two tracks per wave iteration) where they waste fewer lanes than 64-detection tiles: D = 92 (car), 22 (bus) ...
    const double fill64 = (double)D / (64.0 * cdiv(D, 64)), fill32 = (double)D / (32.0 * cdiv(D, 32));
    const int dt = fill32 > 1.1 * fill64 ? 32 : 64, tpi = 64 / dt;
    // tracks per wave: the T tracks dealt evenly to the waves of the ny workgroups of a detection tile, ny the smallest power of two
    // that leaves 1024 workgroups (two rounds at two workgroups per CU); see launch_pair_f16
    int ny = 1;
    while ((long)B * cdiv(D, dt) * ny < 1024 && cdiv(T, wpb * ny * 2) >= 2 * tpi) ny *= 2;
    // ... and while the launch does not even hold one workgroup per CU (one small frame-pair: 3 tiles at max_obj 90), down to ONE
    // iteration per wave: the prologue is repeated, but 24 workgroups finish in a third of the time of 3 (31 -> ~10 us at D = 92)
    while ((long)B * cdiv(D, dt) * ny < 256 && cdiv(T, wpb * ny * 2) >= tpi) ny *= 2;
    const int tw = cdiv(cdiv(T, wpb * ny), tpi) * tpi;
    const size_t lds = ((size_t)dt * (d.ET + 4) + ((a4_total(F) + 3) & ~3) + (size_t)wpb * 3 * tpi * 256) * sizeof(float);
    dim3 grid(cdiv(D, dt), cdiv(T, wpb * tw), B);
#define SHASTA_LAUNCH_PAIR4_DT(FF, DD)                                                                                                    \
    do {                                                                                                                                  \
        if (lds > 64 * 1024)                                                                                                              \
            (void)hipFuncSetAttribute((const void*)pair_mfma4_kernel<FF, wpb, DD>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL((pair_mfma4_kernel<FF, wpb, DD>), grid, dim3(64 * wpb), lds, st, packed, UP, UC, hand_prev, hand_det, denom,   \
                           residual, T, D, ld, nf, tw);                                                                                   \
    } while (0)
#define SHASTA_LAUNCH_PAIR4(FF)                 \
    do {                                        \
        if (dt == 32) SHASTA_LAUNCH_PAIR4_DT(FF, 32); \
        else SHASTA_LAUNCH_PAIR4_DT(FF, 64);    \
    } while (0)
    if (ev0) (void)hipEventRecord(ev0, st);  // bench.py: HIP events around the pair kernel alone
    switch (F) {
        case 64: SHASTA_LAUNCH_PAIR4(64); break;
        case 256: SHASTA_LAUNCH_PAIR4(256); break;
        case 320: SHASTA_LAUNCH_PAIR4(320); break;
        default: set_error_msg("pair_residual: feat_dim must be 64, 256 or 320"); return SHASTA_E_ARG;
    }
    if (ev1) (void)hipEventRecord(ev1, st);
#undef SHASTA_LAUNCH_PAIR4_DT
#undef SHASTA_LAUNCH_PAIR4
    return check_launch("pair_mfma4");
}

int aff_pieces_pack(const shasta_weights* w, float* out, hipStream_t st);
int aff_f16_pack(const shasta_weights* w, float* out, hipStream_t st);
int pack_weights(const shasta_weights* w, float* packed, hipStream_t st) {
    PackArgs a;
    for (int i = 0; i < 4; ++i) a.fs[i] = w->fuse_shape[i];
    for (int i = 0; i < 3; ++i) a.fd[i] = w->fuse_det[i];
    for (int i = 0; i < 3; ++i) a.rc[i] = w->res_coeff[i];
    a.aff0 = w->aff[0];
    a.out = packed;
    a.N = w->max_obj;
    a.nf = w->num_feats;
    a.F = w->feat_dim;
    hipLaunchKernelGGL(pack_pair_weights_kernel, dim3(256), dim3(256), 0, st, a);
    int rc = check_launch("pack_pair_weights");
    if (rc) return rc;
    const PackedLayout P(w->max_obj, w->num_feats, w->feat_dim);
    if ((rc = aff_pieces_pack(w, packed + P.affp, st))) return rc;
    if ((rc = aff_f16_pack(w, packed + P.aff16, st))) return rc;
    if (w->feat_dim == 256 && (rc = pair_f16_pack(w, packed + P.p16, st))) return rc;
    if (pair_f16w_serves(w->feat_dim) && (rc = pair_f16w_pack(w, packed + P.p16w, st))) return rc;
    if (embed_rows_serves(w->feat_dim) && (rc = embed_pack(w, packed, st))) return rc;
    return rc;
}

}  // namespace shasta

#ifdef PAIR_STAMP
extern "C" __attribute__((visibility("default"))) int shasta_debug_pair_stamp(unsigned long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(shasta::g_pair_stamp), sizeof(shasta::g_pair_stamp));
}
#endif
