// K4/K5: pair residual.  Restates det3d/models/tracker/shasta.py:277-319:
//   hand-designed residuals (:277-283), fuse_shape (:286-290), fuse_det (:293-307), res_coeff (:310-316) and the
//   weighted sum residual = alpha*fused + beta*dist + omega*shape (:319), for all T x D (track, detection) pairs.
// See pair_layout.hpp for the factorisation and the MFMA operand chaining.
//
// Kernels (per forward):
//   pack_pair_weights   once per weight load: fragments + factorised first-layer matrices
//   [gemm_nt_f32 x2]    UP = prev_feat . Wemb_prev^T ; UC = feat . Wemb_cur^T + b          (matrix cores)
//   row_finish          per table row: box columns of res_coeff.0 / fuse_det.0, log-dims, cos/sin
//   col_norm            column L2 norm over tracks (F.normalize, dim=1) of the squared box distances
//                       (the dim / rot terms and the normalised distance are evaluated inside pair_mfma)
//   pair_mfma<F>        per 16 pairs: h1 = relu(UP[t]+UC[d]) then 44 (F=256) v_mfma_f32_16x16x4_f32 -> residual
#include "common.hpp"
#include "pair_layout.hpp"
#include <stdlib.h>

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

// feature held by output row i of block bo (or -1)
__device__ __forceinline__ int row_feature(const LayerDesc& L, int bo, int i) {
    const int c = blk_count(L.hout, bo);
    const int fl = L.final_ ? i : (i >> 2) + 4 * (i & 3);
    return fl < c ? 16 * bo + fl : -1;
}

// input feature multiplied at k-step s by lanes with k-slot kq (or -1)
__device__ __forceinline__ int step_input(const LayerDesc& L, int s, int kq) {
    if (!L.chained) return kq * (L.kin / 4) + s;
    int bi = 0, r = s;
    while (true) {
        const int c = blk_count(L.kin, bi);
        const int st = (c + 3) / 4;
        if (r < st) {
            const int fl = kq + 4 * r;
            return fl < c ? 16 * bi + fl : -1;
        }
        r -= st;
        ++bi;
    }
}

__global__ void pack_pair_weights_kernel(PackArgs a) {
    const PackedLayout P(a.N, a.nf, a.F);
    const PairDims d(a.F);
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    const int nth = gridDim.x * blockDim.x;
    const int F = a.F, nf = a.nf;
    // MFMA A-operand fragments
    for (int l = 0; l < L_COUNT; ++l) {
        const LayerDesc L = layer_desc(F, l);
        const float *W, *b;
        int ldw;
        layer_src(a, l, W, b, ldw);
        const int steps = L.steps(), nb = nblk(L.hout);
        float* fo = a.out + P.frags + (size_t)frag_offset(F, l) * 64;
        for (int e = tid; e < nb * steps * 64; e += nth) {
            const int lane = e & 63, s = (e >> 6) % steps, bo = (e >> 6) / steps;
            const int f = row_feature(L, bo, lane & 15), k = step_input(L, s, lane >> 4);
            fo[e] = (f >= 0 && k >= 0) ? W[(size_t)f * ldw + k] : 0.0f;
        }
        // bias as the initial accumulator: lane (kq = lane>>4) register r holds output row 4*kq + r
        float* bo_ = a.out + P.biasf + (size_t)bias_offset(F, l) * 256;
        for (int e = tid; e < nb * 256; e += nth) {
            const int r = e & 3, lane = (e >> 2) & 63, bo = e >> 8;
            const int f = row_feature(L, bo, 4 * (lane >> 4) + r);
            bo_[e] = f >= 0 ? b[f] : 0.0f;
        }
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
// row_finish: one wave per table row (prev rows then cur rows of each batch item)
//   E[row][H1 + j]       += Wbox[j][:nf] . box[:nf]                 (res_coeff.0 box columns)
//   E[row][H1 + R1 + j]   = Wbox[R1 + j][:nf] . box[:nf] (+ bias)   (fuse_det.0)
//   hand[row] = [box7, 0, log(w+eps), log(l+eps), log(h+eps), cos(yaw), sin(yaw), 0, 0, 0]
// ------------------------------------------------------------------------------------------------------------
struct RowFinishArgs {
    const float* packed;
    const float* tab[2];  // [0] prev_tab, [1] det_tab   (B, T, 8)
    float* emb[2];        // [0] UP, [1] UC             (B, T, ET)
    float* hand[2];       // (B, T, 16)
    int B, T, N, nf, F;
};

__global__ __launch_bounds__(256) void row_finish_kernel(RowFinishArgs a) {
    const PackedLayout P(a.N, a.nf, a.F);
    const PairDims d(a.F);
    const int lane = threadIdx.x & 63;
    const int item = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (item >= 2 * a.B * a.T) return;
    const int which = item / (a.B * a.T), row = item % (a.B * a.T);
    const float* box = a.tab[which] + (size_t)row * 8;
    float* e = a.emb[which] + (size_t)row * d.ET;
    const float* wb = a.packed + (which ? P.wbox_cur : P.wbox_prev);
    float bx[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) bx[c] = box[c];
    for (int j = lane; j < d.R1 + 32; j += 64) {
        float s = 0.0f;
#pragma unroll
        for (int c = 0; c < 7; ++c) s = fmaf(wb[j * 8 + c], bx[c], s);  // columns >= nf are packed as 0
        if (j < d.R1) {
            e[d.H1 + j] += s;
        } else {
            if (which) s += a.packed[P.bbox_cur + (j - d.R1)];
            e[d.H1 + j] = s;
        }
    }
    if (lane < 16) {
        float v = 0.0f;
        if (lane < 7) v = bx[lane];
        else if (lane >= 8 && lane < 11) v = logf(bx[3 + lane - 8] + 1e-10f);
        else if (lane == 11) v = cosf(bx[6]);
        else if (lane == 12) v = sinf(bx[6]);
        a.hand[which][(size_t)row * 16 + lane] = v;
    }
}

// ------------------------------------------------------------------------------------------------------------
// col_norm (shasta.py:278-279): d2[t][d] = sum_{k<nf} (prev_k - det_k)^2 ; denom[d] = max(||d2[:, d]||_2, 1e-12)
// (F.normalize acts along dim=1 = tracks).  block = 16 detections x 16 track groups; the previous boxes are read as
// one float4 pair per track (first 8 floats of the 16-float hand row), 4 tracks in flight per thread.
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void col_norm_kernel(const float* __restrict__ hand_prev,
                                                       const float* __restrict__ hand_det, float* __restrict__ denom,
                                                       int T, int D, int nf) {
    __shared__ float red[16][17];
    const int b = blockIdx.y, dl = threadIdx.x & 15, tg = threadIdx.x >> 4;
    const int d = blockIdx.x * 16 + dl;
    float db[8];
    {
        const f32x4* h = reinterpret_cast<const f32x4*>(hand_det + ((size_t)b * D + min(d, D - 1)) * 16);
        const f32x4 a = h[0], c = h[1];
        db[0] = a[0]; db[1] = a[1]; db[2] = a[2]; db[3] = a[3]; db[4] = c[0]; db[5] = c[1]; db[6] = c[2]; db[7] = 0.0f;
    }
    // columns >= nf do not take part: zero both sides
    float mask[7];
#pragma unroll
    for (int k = 0; k < 7; ++k) mask[k] = k < nf ? 1.0f : 0.0f;
    const f32x4* hp = reinterpret_cast<const f32x4*>(hand_prev + (size_t)b * T * 16);
    float ssq = 0.0f;
#pragma unroll 4
    for (int t = tg; t < T; t += 16) {
        const f32x4 a = hp[(size_t)t * 4], c = hp[(size_t)t * 4 + 1];
        const float p[7] = {a[0], a[1], a[2], a[3], c[0], c[1], c[2]};
        float d2 = 0.0f;
#pragma unroll
        for (int k = 0; k < 7; ++k) {
            const float df = (p[k] - db[k]) * mask[k];
            d2 += df * df;
        }
        ssq += d2 * d2;
    }
    red[dl][tg] = ssq;
    __syncthreads();
    if (tg == 0 && d < D) {
        float tot = 0.0f;
#pragma unroll
        for (int i = 0; i < 16; ++i) tot += red[dl][i];  // fixed order
        denom[(size_t)b * D + d] = fmaxf(sqrtf(tot), 1e-12f);
    }
}

// ------------------------------------------------------------------------------------------------------------
// pair_mfma
// ------------------------------------------------------------------------------------------------------------
// tracks per workgroup: a runtime argument (8..64).  Larger values amortise the per-workgroup prologue (weight
// fragments, detection-side embeddings, bias fragments) over more pairs; smaller ones give B=1 enough workgroups.

template <int F, int L>
struct Frags {
    static constexpr LayerDesc D = layer_desc(F, L);
    static constexpr int NB = nblk(D.hout), ST = D.steps();
    float w[NB][ST];
    __device__ __forceinline__ void load(const float* packed_frags, int lane) {
#pragma unroll
        for (int bo = 0; bo < NB; ++bo)
#pragma unroll
            for (int s = 0; s < ST; ++s) w[bo][s] = packed_frags[(size_t)(frag_offset(F, L) + bo * ST + s) * 64 + lane];
    }
};

#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

template <int F, int UNROLL>
__global__ __launch_bounds__(256) void pair_mfma_kernel(const float* __restrict__ packed, const float* __restrict__ UP,
                                                        const float* __restrict__ UC, const float* __restrict__ hand_prev,
                                                        const float* __restrict__ hand_det, const float* __restrict__ denom,
                                                        float* __restrict__ residual, int T, int D, int ld, int nf,
                                                        int TT) {
    constexpr PairDims dm(F);
    constexpr int H1 = dm.H1, R1 = dm.R1, ET = dm.ET;
    constexpr int S_FS = H1 / 4, S_RC = R1 / 4, S_FD = 8;
    constexpr int NBIAS = total_bias_blocks(F);
    extern __shared__ __attribute__((aligned(16))) float s_dyn[];
    __shared__ __attribute__((aligned(16))) float s_bias[NBIAS * 256];
    __shared__ float s_hd[64 * 17];
    float* s_up = s_dyn;                // [TT][ET]
    float* s_dist = s_up + TT * ET;     // [TT][64]
    float* s_hp = s_dist + TT * 64;     // [TT][16], followed by 256 floats of scratch

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int p = lane & 15, kq = lane >> 4;
    const int b = blockIdx.z, t0 = blockIdx.y * TT, dblk = blockIdx.x * 64;
    const int d = dblk + wid * 16 + p;
    const int dc = min(d, D - 1);
    const PackedLayout P(0, 0, F);  // only the fragment/bias offsets are used here (independent of N, nf)

    // stage UP rows of this workgroup's tracks, the bias fragments and the dist tile
    const int nt = min(TT, T - t0);
    {
        // ET is a multiple of 4 and the tables are 16-byte aligned: stage with float4, several loads in flight
        const f32x4* src = reinterpret_cast<const f32x4*>(UP + ((size_t)b * T + t0) * ET);
        f32x4* dst = reinterpret_cast<f32x4*>(s_up);
#pragma unroll 4
        for (int e = tid; e < nt * (ET / 4); e += 256) dst[e] = src[e];
        const f32x4* bsrc = reinterpret_cast<const f32x4*>(packed + P.biasf);
        f32x4* bdst = reinterpret_cast<f32x4*>(s_bias);
#pragma unroll
        for (int e = tid; e < NBIAS * 64; e += 256) bdst[e] = bsrc[e];
    }
#pragma unroll 4
    for (int e = tid; e < TT * 16; e += 256) s_hp[e] = hand_prev[((size_t)b * T + min(t0 + (e >> 4), T - 1)) * 16 + (e & 15)];
#pragma unroll
    for (int e = tid; e < 64 * 16; e += 256) {
        const int dd = e >> 4, c = e & 15;
        // slot 7 of a detection row (unused padding in the hand table) carries the column norm of shasta.py:279
        s_hd[dd * 17 + c] = c == 7 ? denom[(size_t)b * D + min(dblk + dd, D - 1)]
                                   : hand_det[((size_t)b * D + min(dblk + dd, D - 1)) * 16 + c];
    }

    Frags<F, L_FS2> w_fs2; Frags<F, L_FS3> w_fs3; Frags<F, L_FS4> w_fs4;
    Frags<F, L_RC2> w_rc2; Frags<F, L_RC3> w_rc3;
    Frags<F, L_FD2> w_fd2; Frags<F, L_FD3> w_fd3;
    const float* pf = packed + P.frags;
    w_fs2.load(pf, lane); w_fs3.load(pf, lane); w_fs4.load(pf, lane);
    w_rc2.load(pf, lane); w_rc3.load(pf, lane);
    w_fd2.load(pf, lane); w_fd3.load(pf, lane);

    // this lane's slice of the detection-side embedding: k = kq*S + s
    float uc_fs[S_FS], uc_rc[S_RC], uc_fd[S_FD];
    {
        const float* u = UC + ((size_t)b * D + dc) * ET;
#pragma unroll
        for (int s = 0; s < S_FS; ++s) uc_fs[s] = u[kq * S_FS + s];
#pragma unroll
        for (int s = 0; s < S_RC; ++s) uc_rc[s] = u[H1 + kq * S_RC + s];
#pragma unroll
        for (int s = 0; s < S_FD; ++s) uc_fd[s] = u[H1 + R1 + kq * S_FD + s];
    }
    __syncthreads();
    // hand-designed residual (shasta.py:277-283) for the TT x 64 pairs of this workgroup, 4 pairs per thread
    for (int e = tid; e < TT * 64; e += 256) {
        const int tt = e >> 6, dd = e & 63;
        const float* hp = s_hp + tt * 16;
        const float* hd = s_hd + dd * 17;
        float d2 = 0.0f;
#pragma unroll
        for (int k = 0; k < 7; ++k)
            if (k < nf) {
                const float df = hp[k] - hd[k];
                d2 += df * df;
            }
        float r = d2 / hd[7];
        const float dim = (fabsf(hp[8] - hd[8]) + fabsf(hp[9] - hd[9])) + fabsf(hp[10] - hd[10]);
        const float dcs = hp[11] - hd[11], dsn = hp[12] - hd[12];
        s_dist[e] = (r + dim) + sqrtf(dcs * dcs + dsn * dsn);
    }
    __syncthreads();

    auto bias = [&](int l, int bo) -> f32x4 {
        return *reinterpret_cast<const f32x4*>(&s_bias[(bias_offset(F, l) + bo) * 256 + lane * 4]);
    };
    auto relu4 = [](f32x4 v) -> f32x4 {
        v[0] = fmaxf(v[0], 0.0f); v[1] = fmaxf(v[1], 0.0f); v[2] = fmaxf(v[2], 0.0f); v[3] = fmaxf(v[3], 0.0f);
        return v;
    };

    // One step = the 16 pairs (track t0+tt) x (this wave's 16 detections).  No branch inside: the result goes to the
    // LDS tile (lanes kq != 0 write to a scratch slot), so two steps can be interleaved by the scheduler (UNROLL = 2)
    // and the tile leaves the workgroup as whole 256-byte rows after the loop.
    auto step = [&](int tt) {
        const float* up = s_up + tt * ET;
        // ---- layer 1 (factorised): h1 = relu(UP[t] + UC[d]) ----
        float h_fs[S_FS], h_rc[S_RC], h_fd[S_FD];
#pragma unroll
        for (int s = 0; s < S_FS; ++s) h_fs[s] = fmaxf(up[kq * S_FS + s] + uc_fs[s], 0.0f);
#pragma unroll
        for (int s = 0; s < S_RC; ++s) h_rc[s] = fmaxf(up[H1 + kq * S_RC + s] + uc_rc[s], 0.0f);
#pragma unroll
        for (int s = 0; s < S_FD; ++s) h_fd[s] = fmaxf(up[H1 + R1 + kq * S_FD + s] + uc_fd[s], 0.0f);

        // ---- layer 2 of the three MLPs (independent accumulators, interleaved) ----
        constexpr int NB_FS2 = Frags<F, L_FS2>::NB, NB_RC2 = Frags<F, L_RC2>::NB;
        f32x4 a_fs2[NB_FS2], a_rc2[NB_RC2], a_fd2;
#pragma unroll
        for (int bo = 0; bo < NB_FS2; ++bo) a_fs2[bo] = bias(L_FS2, bo);
#pragma unroll
        for (int bo = 0; bo < NB_RC2; ++bo) a_rc2[bo] = bias(L_RC2, bo);
        a_fd2 = bias(L_FD2, 0);
        constexpr int SMAX = S_RC > S_FS ? (S_RC > S_FD ? S_RC : S_FD) : (S_FS > S_FD ? S_FS : S_FD);
#pragma unroll
        for (int s = 0; s < SMAX; ++s) {
            if (s < S_RC) {
#pragma unroll
                for (int bo = 0; bo < NB_RC2; ++bo) a_rc2[bo] = MFMA16(w_rc2.w[bo][s], h_rc[s], a_rc2[bo]);
            }
            if (s < S_FS) {
#pragma unroll
                for (int bo = 0; bo < NB_FS2; ++bo) a_fs2[bo] = MFMA16(w_fs2.w[bo][s], h_fs[s], a_fs2[bo]);
            }
            if (s < S_FD) a_fd2 = MFMA16(w_fd2.w[0][s], h_fd[s], a_fd2);
        }
#pragma unroll
        for (int bo = 0; bo < NB_FS2; ++bo) a_fs2[bo] = relu4(a_fs2[bo]);
#pragma unroll
        for (int bo = 0; bo < NB_RC2; ++bo) a_rc2[bo] = relu4(a_rc2[bo]);
        a_fd2 = relu4(a_fd2);

        // ---- layer 3: inputs are the layer-2 accumulators, register r of block bi = k-step ----
        f32x4 a_fs3 = bias(L_FS3, 0), a_rc3 = bias(L_RC3, 0), a_fd3 = bias(L_FD3, 0);
        {
            int st = 0;
#pragma unroll
            for (int bi = 0; bi < NB_FS2; ++bi)
#pragma unroll
                for (int r = 0; r < (blk_count(dm.H2, bi) + 3) / 4; ++r) a_fs3 = MFMA16(w_fs3.w[0][st++], a_fs2[bi][r], a_fs3);
            st = 0;
#pragma unroll
            for (int bi = 0; bi < NB_RC2; ++bi)
#pragma unroll
                for (int r = 0; r < (blk_count(dm.R2, bi) + 3) / 4; ++r) a_rc3 = MFMA16(w_rc3.w[0][st++], a_rc2[bi][r], a_rc3);
#pragma unroll
            for (int r = 0; r < 2; ++r) a_fd3 = MFMA16(w_fd3.w[0][r], a_fd2[r], a_fd3);
        }
        a_fs3 = relu4(a_fs3);
        // ---- layer 4 of fuse_shape ----
        f32x4 a_fs4 = bias(L_FS4, 0);
#pragma unroll
        for (int r = 0; r < (dm.H3 + 3) / 4; ++r) a_fs4 = MFMA16(w_fs4.w[0][r], a_fs3[r], a_fs4);

        // ---- combine (shasta.py:316-319): alpha, beta, omega = res_coeff outputs 0,1,2 (valid in the kq == 0 lanes) ----
        const float alpha = a_rc3[0], beta = a_rc3[1], omega = a_rc3[2];
        const float fused = a_fd3[0], shape = a_fs4[0];
        const int slot = tt * 64 + wid * 16 + p;
        const float dst = s_dist[slot];
        const float r = (alpha * fused + beta * dst) + omega * shape;
        s_dist[kq == 0 ? slot : TT * 64 + TT * 16 + lane + 64 * wid] = r;  // scratch behind s_hp for the other lanes
    };
    if constexpr (UNROLL == 2) {
        int tt = 0;
        for (; tt + 1 < nt; tt += 2) {
            step(tt);
            step(tt + 1);
        }
        if (tt < nt) step(tt);
    } else {
        for (int tt = 0; tt < nt; ++tt) step(tt);
    }
    __syncthreads();
    for (int e = tid; e < nt * 64; e += 256) {
        const int tt = e >> 6, dd = dblk + (e & 63);
        if (dd < D) residual[((size_t)b * T + t0 + tt) * ld + dd] = s_dist[e];
    }
    (void)d;
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

int launch_gemm_nt(const float* A, int lda, const float* W, int ldw, const float* bias, float* C, int ldc, int M,
                   int N, int K, int act, hipStream_t st);

int pair_residual(const shasta_weights* w, const float* packed, int B, const float* feat, const float* prev_feat,
                  const float* det_tab, const float* prev_tab, float* residual, int ld, void* ws, size_t ws_bytes,
                  hipStream_t st) {
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

    int rc = launch_gemm_nt(prev_feat, F, packed + P.wemb_prev, F, nullptr, UP, d.ET, B * T, P.E12, F, 0, st);
    if (rc) return rc;
    rc = launch_gemm_nt(feat, F, packed + P.wemb_cur, F, packed + P.bemb_cur, UC, d.ET, B * D, P.E12, F, 0, st);
    if (rc) return rc;
    RowFinishArgs rf;
    rf.packed = packed;
    rf.tab[0] = prev_tab;
    rf.tab[1] = det_tab;
    rf.emb[0] = UP;
    rf.emb[1] = UC;
    rf.hand[0] = hand_prev;
    rf.hand[1] = hand_det;
    rf.B = B;
    rf.T = T;
    rf.N = N;
    rf.nf = nf;
    rf.F = F;
    hipLaunchKernelGGL(row_finish_kernel, dim3(cdiv(2 * B * T, 4)), dim3(256), 0, st, rf);
    rc = check_launch("row_finish");
    if (rc) return rc;
    hipLaunchKernelGGL(col_norm_kernel, dim3(cdiv(D, 16), B), dim3(256), 0, st, hand_prev, hand_det, denom, T, D, nf);
    rc = check_launch("col_norm");
    if (rc) return rc;
    // tracks per workgroup: the largest of 32/16/8 that still yields >= 2 waves per SIMD on 256 CUs
    int tt = 32;  // measured: 8..32 tie once the chip is full, 64 loses occupancy to its LDS footprint
    while (tt > 8 && (long)B * cdiv(D, 64) * 4 * cdiv(T, tt) < 2048) tt >>= 1;
    if (const char* e = getenv("SHASTA_PAIR_TT")) {  // tuning override
        const int v = atoi(e);
        if (v == 8 || v == 16 || v == 32 || v == 64) tt = v;
    }
    const size_t lds = ((size_t)tt * (d.ET + 64 + 16) + 256) * sizeof(float);
    dim3 grid(cdiv(D, 64), cdiv(T, tt), B);
    static const bool unroll2 = getenv("SHASTA_PAIR_UNROLL2") != nullptr;
    switch (F) {
        case 64:
            if (unroll2) hipLaunchKernelGGL((pair_mfma_kernel<64, 2>), grid, dim3(256), lds, st, packed, UP, UC, hand_prev, hand_det, denom, residual, T, D, ld, nf, tt);
            else hipLaunchKernelGGL((pair_mfma_kernel<64, 1>), grid, dim3(256), lds, st, packed, UP, UC, hand_prev, hand_det, denom, residual, T, D, ld, nf, tt);
            break;
        case 256:
            if (unroll2) hipLaunchKernelGGL((pair_mfma_kernel<256, 2>), grid, dim3(256), lds, st, packed, UP, UC, hand_prev, hand_det, denom, residual, T, D, ld, nf, tt);
            else hipLaunchKernelGGL((pair_mfma_kernel<256, 1>), grid, dim3(256), lds, st, packed, UP, UC, hand_prev, hand_det, denom, residual, T, D, ld, nf, tt);
            break;
        case 320:
            if (unroll2) hipLaunchKernelGGL((pair_mfma_kernel<320, 2>), grid, dim3(256), lds, st, packed, UP, UC, hand_prev, hand_det, denom, residual, T, D, ld, nf, tt);
            else hipLaunchKernelGGL((pair_mfma_kernel<320, 1>), grid, dim3(256), lds, st, packed, UP, UC, hand_prev, hand_det, denom, residual, T, D, ld, nf, tt);
            break;
        default: set_error_msg("pair_residual: feat_dim must be 64, 256 or 320"); return SHASTA_E_ARG;
    }
    return check_launch("pair_mfma");
}

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
    return check_launch("pack_pair_weights");
}

}  // namespace shasta
