// Training path (SURVEY.md 8(a) row 19 / BASELINE config 5): one pair MLP of det3d/models/tracker/shasta.py:59-92 (fuse_shape,
// fuse_det, res_coeff) recomputed and back-propagated per pair in registers.  The dense formulation of round 1-4 wrote the hidden
// activations of every pair - (B T D) x (F/8 | 32 | 32 + F/8) floats, 1 GB at N = 500, B = 8 - and every later layer's, and read them
// back through strided GEMMs (7 ms of a 9.6 ms backward); here nothing per pair leaves the chip but the MLP's output (forward kernel)
// or comes in but its gradient (backward kernel).
//
//   lane = detection d of a 64-wide tile, the wave walks its slice of the tracks t:
//     h1[i] = relu(UP[(b,t)][i] + UC[(b,d)][i])                      (the first layer, factorised over the table rows: pair_layout.hpp)
//     h2 = relu(W2 h1 + b2), [h3 = relu(W3 h2 + b3), out = W4 h3 + b4 | out = W3 h2 + b3]
//   backward per lane: g4 | g3, g2, gz1 = (W2^T g2) . [h1 > 0];  gUC[(b,d)] += gz1 in registers over the wave's tracks.
//   Layers 2 and 3 and their transposes run on v_mfma_f32_4x4x1_f32 with lane = pair (the layout of pair_mfma4_kernel, pair.hip: the
//   4 result registers of a lane are 4 features of ITS pair), the weights as A operands from LDS tables the workgroup builds once.
//   (Weights as scalar operands of plain FMAs - the first form - do not survive the compiler: it hoists a thousand s_loads to the top
//   of the unrolled body and spills the scalar registers into vector lanes; and scalar loads share lgkmcnt with the LDS traffic.)
//   What sums over the LANES goes through a per-wave LDS tile [feature row][lane] (row stride 68 floats = 4 mod 32: the 16 rows x 4 pairs
//   of an MFMA operand cover the banks once, a lane's b128 reads along its row are conflict-free too):
//     weight / bias gradients: v_mfma_f32_16x16x4_f32 with the pair index on the K axis - A = 16 of the rows [g2 | g3 | g4], B = 16 of the
//     rows [h1 | h2 | h3 | 1], only the blocks that hold a wanted product (g2 x h1, g3 x h2, g4 x h3, g x 1 = the biases: 7 of 12 blocks at
//     F = 256's res_coeff) - accumulated over all pairs of the wave (the matrix pipe is otherwise idle);
//     gUP[(b,t)][i] = sum over the tile's lanes of gz1: lane i sums row i of the tile (gz1 overwrites h1 after the MFMAs have read it;
//     LDS operations of one wave execute in order, the tile is private to the wave: no barrier anywhere in the loop).
//   The UP rows of the wave's tracks arrive a chunk (one 1 KB LDS-DMA) ahead of their use.
//   Partial results - gUP per detection tile, gUC per track slice, the weight image per wave - are summed in a fixed order by
//   sum_slices_kernel: every gradient is deterministic.
#include "common.hpp"

namespace shasta {

template <int E1_, int E2_, int E3_, int E4_>
struct PairMlp {
    static constexpr int E1 = E1_, E2 = E2_, E3 = E3_, E4 = E4_;
    static constexpr int E2P = (E2 + 3) & ~3, E3P = (E3 + 3) & ~3;  // the later widths rounded up to the 4 rows of a 4x4x1 block
    static constexpr int NOUT = E4 ? E4 : E3;
    static constexpr int G2 = 0, G3 = E2, G4 = E2 + E3, EG = E2 + E3 + E4;                 // rows of the A tile
    static constexpr int H1 = 0, H2 = E1, H3 = E1 + E2, ONE = E1 + E2 + (E4 ? E3 : 0), EH = ONE + 1;  // rows of the B tile
    // 16 x 16 x 4 tiles of the weight-gradient products: MB blocks of 16 gradient rows x NBH blocks of 16 activation rows, of which only
    // the blocks that hold a wanted product (g2 x h1, g3 x h2, g4 x h3, any g x 1) are computed
    static constexpr int MB = (EG + 15) / 16, NBH = (EH + 15) / 16;
    static constexpr int ROWS = 32 + 16 * NBH, RS = 68;  // row stride = 4 mod 32 floats: the 16 rows x 4 pairs of an operand cover the banks once
    static constexpr int LDS_TILE = ROWS * RS;  // floats per wave
    static constexpr bool overlap(int a0, int a1, int b0, int b1) { return a0 < b1 && b0 < a1; }
    static constexpr bool needed(int mb, int nb) {
        const int g0 = 16 * mb, g1 = g0 + 16, h0 = 16 * nb, h1 = h0 + 16;
        const bool one = overlap(h0, h1, ONE, ONE + 1);
        return (overlap(g0, g1, G2, G2 + E2) && (overlap(h0, h1, H1, H1 + E1) || one)) ||
               (overlap(g0, g1, G3, G3 + E3) && (overlap(h0, h1, H2, H2 + E2) || one)) ||
               (E4 > 0 && overlap(g0, g1, G4, G4 + E4) && (overlap(h0, h1, H3, H3 + E3) || one));
    }
    // weight tables in LDS, shared by the workgroup: a layer W (OUT, IN) as A operands of v_mfma_f32_4x4x1_f32,
    //   forward form  [ob][kg][i][kk] = W[4 ob + i][4 kg + kk]        (4 output rows per block, K walks the inputs)
    //   backward form [ib][og][i][kk] = W[4 og + kk][4 ib + i]        (the same for W^T: 4 input rows per block, K walks the outputs)
    // lane l reads the 16 bytes of its row i = l & 3: four K steps of one block; zero where a width was rounded up
    static constexpr int T2F = 0, T2B = T2F + E2P * E1, T3F = T2B + E2P * E1, T3B = T3F + E3P * E2P, TB2 = T3B + E3P * E2P,
                         TB3 = TB2 + E2P, TW4 = TB3 + E3P, TB4 = TW4 + E3P, NTAB = TB4 + 4;
    static constexpr int CH = 256 / E1;        // UP rows staged per wave at a time: one 1 KB LDS-DMA (64 lanes x 16 bytes)
    static constexpr int LDS_UP = 2 * 256;     // floats per wave: the chunk in use and the next one on its way
    // the flat gradient image: [gW2 (E2, E1) | gb2 | gW3 (E3, E2) | gb3 | gW4 (E4, E3) | gb4], torch's (out, in) layout
    static constexpr int OW2 = 0, OB2 = E2 * E1, OW3 = OB2 + E2, OB3 = OW3 + E3 * E2, OW4 = OB3 + E3, OB4 = OW4 + E4 * E3, NE = OB4 + E4;
    static_assert(EG <= 32, "the gradient rows of one pair must fit one 32-row A tile");
    static_assert(E1 % 4 == 0 && E1 <= 128, "first-layer width");
    static_assert(E4 == 0 || E4 == 1, "the last layer of a four-layer MLP has one output");
};

constexpr int PB_WPB = 4;  // waves per workgroup

template <class M>
constexpr size_t pb_lds_bytes(bool bwd) {
    return (size_t)(M::NTAB + PB_WPB * M::LDS_UP + (bwd ? PB_WPB * M::LDS_TILE : 0)) * sizeof(float);
}

#define PB_MFMA4(a, b, c) __builtin_amdgcn_mfma_f32_4x4x1f32((a), (b), (c), 0, 0, 0)
#ifndef PB_ABL  // ablation builds only (tools/time_pair_mlp.py; wrong results): 1 no weight-gradient MFMAs, 2 no W2^T g2, 4 no sum over the lanes
#define PB_ABL 0
#endif

// W2 (E2, E1), W3 (E3, E2), W4 (E4, E3): torch's nn.Linear layout; b2, b3, b4 their biases (W4, b4 unused when E4 == 0)
template <class M, bool BWD>
__global__ __launch_bounds__(64 * PB_WPB) void pair_mlp_kernel(const float* __restrict__ UP, const float* __restrict__ UC,
                                                               const float* __restrict__ w2, const float* __restrict__ b2,
                                                               const float* __restrict__ w3, const float* __restrict__ b3,
                                                               const float* __restrict__ w4, const float* __restrict__ b4,
                                                               const float* __restrict__ gout, float* __restrict__ out,
                                                               float* __restrict__ gup_part, float* __restrict__ guc_part,
                                                               float* __restrict__ w_part, int B, int T, int D, int tw) {
    constexpr int E1 = M::E1, E2 = M::E2, E3 = M::E3, E4 = M::E4, E2P = M::E2P, E3P = M::E3P, NOUT = M::NOUT, RS = M::RS, MB = M::MB, NBH = M::NBH;
    constexpr int KG1 = E1 / 4, KG2 = E2P / 4, KG3 = E3P / 4;
    extern __shared__ __attribute__((aligned(16))) float s_pb[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    int dt, lby, b;
    xcd_logical_block(dt, lby, b);  // the tiles and track slices of a frame on one XCD: its UP / UC rows come from HBM once (common.hpp)
    const int d = dt * 64 + lane, dcl = min(d, D - 1);
    const int slice = lby * PB_WPB + wid, nslice = gridDim.y * PB_WPB;
    const int t_beg = slice * tw, t_end = min(T, t_beg + tw);
    float* tab = s_pb;
    float* Lup = s_pb + M::NTAB + wid * M::LDS_UP;
    float* L = s_pb + M::NTAB + PB_WPB * M::LDS_UP + (BWD ? wid * M::LDS_TILE : 0);
    // ---- the weight tables ----
    for (int e = tid; e < E2P * E1; e += 64 * PB_WPB) {
        const int kk = e & 3, i = (e >> 2) & 3, g = e >> 4;
        {
            const int ob = g / KG1, kg = g - ob * KG1, o = 4 * ob + i, k = 4 * kg + kk;
            tab[M::T2F + e] = o < E2 ? w2[o * E1 + k] : 0.0f;
        }
        {
            const int ib = g / KG2, og = g - ib * KG2, o = 4 * og + kk, k = 4 * ib + i;
            tab[M::T2B + e] = o < E2 ? w2[o * E1 + k] : 0.0f;
        }
    }
    for (int e = tid; e < E3P * E2P; e += 64 * PB_WPB) {
        const int kk = e & 3, i = (e >> 2) & 3, g = e >> 4;
        {
            const int qb = g / KG2, og = g - qb * KG2, q = 4 * qb + i, o = 4 * og + kk;
            tab[M::T3F + e] = (q < E3 && o < E2) ? w3[q * E2 + o] : 0.0f;
        }
        {
            const int ob = g / KG3, qg = g - ob * KG3, q = 4 * qg + kk, o = 4 * ob + i;
            tab[M::T3B + e] = (q < E3 && o < E2) ? w3[q * E2 + o] : 0.0f;
        }
    }
    for (int e = tid; e < E2P; e += 64 * PB_WPB) tab[M::TB2 + e] = e < E2 ? b2[e] : 0.0f;
    for (int e = tid; e < E3P; e += 64 * PB_WPB) {
        tab[M::TB3 + e] = e < E3 ? b3[e] : 0.0f;
        tab[M::TW4 + e] = (E4 && e < E3) ? w4[e] : 0.0f;
    }
    if (tid < 4) tab[M::TB4 + tid] = (E4 && tid == 0) ? b4[0] : 0.0f;
    float uc[E1], guc[E1];
    {
        const f32x4* src = reinterpret_cast<const f32x4*>(UC + ((size_t)b * D + dcl) * E1);
#pragma unroll
        for (int i = 0; i < E1 / 4; ++i) {
            const f32x4 v = src[i];
            uc[4 * i] = v[0]; uc[4 * i + 1] = v[1]; uc[4 * i + 2] = v[2]; uc[4 * i + 3] = v[3];
        }
    }
#pragma unroll
    for (int i = 0; i < E1; ++i) guc[i] = 0.0f;
    f32x4 acc[MB][NBH];  // (the blocks that are not needed are never touched: no registers)
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int nb = 0; nb < NBH; ++nb) acc[mb][nb] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    if (BWD) L[(32 + M::ONE) * RS + lane] = 1.0f;
    __syncthreads();
    const int ka = (lane & 15) * RS + (lane >> 4);  // this lane's element of a 16x16x4 operand: row lane % 16, pair 4 s + lane / 16
    // the UP rows of CH tracks at a time (contiguous in memory), fetched by one LDS-DMA per chunk into the half of the wave's buffer that is
    // not in use, a chunk ahead: lane e brings floats 4 e .. 4 e + 3 of the chunk (lanes past its end: the chunk's first floats again)
    auto fetch_up = [&](int t0, int half) __attribute__((always_inline)) {
        const int nfl = min(M::CH, T - t0) * E1 / 4;
        const float* src = UP + ((size_t)b * T + t0) * E1 + 4 * (lane < nfl ? lane : 0);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(Lup + 256 * half), 16, 0, 0);
    };
    if (t_beg < t_end) fetch_up(t_beg, 0);
    const float* trow = tab + (lane & 3) * 4;        // this lane's row of every 4x4x1 A block

    for (int t = t_beg; t < t_end; ++t) {
        const int tc = t - t_beg, chunk = tc / M::CH, within = tc - chunk * M::CH;
        if (within == 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this chunk has landed
            if (t + M::CH < t_end) fetch_up(t + M::CH, (chunk + 1) & 1);
        }
        const float* upr = Lup + 256 * (chunk & 1) + within * E1;
        const size_t p = ((size_t)b * T + t) * D + dcl;
        // ---- forward ----
        f32x4 z2[KG2];
#pragma unroll
        for (int ob = 0; ob < KG2; ++ob) z2[ob] = *reinterpret_cast<const f32x4*>(tab + M::TB2 + 4 * ob);
#pragma unroll
        for (int kg = 0; kg < KG1; ++kg) {
            const f32x4 u = *reinterpret_cast<const f32x4*>(upr + 4 * kg);
            float h1[4];
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                h1[kk] = relu_nan(u[kk] + uc[4 * kg + kk]);
                if (BWD) L[(32 + M::H1 + 4 * kg + kk) * RS + lane] = h1[kk];
            }
            // (consecutive MFMAs go to different accumulators: a 4x4x1 result is not back for the next issue slot)
            f32x4 a2[KG2];
#pragma unroll
            for (int ob = 0; ob < KG2; ++ob) a2[ob] = *reinterpret_cast<const f32x4*>(trow + M::T2F + (ob * KG1 + kg) * 16);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int ob = 0; ob < KG2; ++ob) z2[ob] = PB_MFMA4(a2[ob][kk], h1[kk], z2[ob]);
        }
        float h2[E2P];
#pragma unroll
        for (int o = 0; o < E2P; ++o) h2[o] = relu_nan(z2[o / 4][o % 4]);
        f32x4 z3[KG3];
#pragma unroll
        for (int qb = 0; qb < KG3; ++qb) z3[qb] = *reinterpret_cast<const f32x4*>(tab + M::TB3 + 4 * qb);
#pragma unroll
        for (int og = 0; og < KG2; ++og)
#pragma unroll
            for (int qb = 0; qb < KG3; ++qb) {
                const f32x4 a = *reinterpret_cast<const f32x4*>(trow + M::T3F + (qb * KG2 + og) * 16);
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) z3[qb] = PB_MFMA4(a[kk], h2[4 * og + kk], z3[qb]);
            }  // (short chains: KG3 <= 3 accumulators)
        float h3[E3P], res[NOUT];
#pragma unroll
        for (int q = 0; q < E3P; ++q) h3[q] = z3[q / 4][q % 4];
        if (E4) {
            float s = tab[M::TB4];
#pragma unroll
            for (int q = 0; q < E3; ++q) {
                h3[q] = relu_nan(h3[q]);
                s = fmaf(tab[M::TW4 + q], h3[q], s);
            }
            res[0] = s;
        } else {
#pragma unroll
            for (int r = 0; r < NOUT; ++r) res[r] = h3[r];
        }
        if (!BWD) {
            if (d < D) {
#pragma unroll
                for (int r = 0; r < NOUT; ++r) out[p * NOUT + r] = res[r];
            }
            continue;
        }
        // ---- backward of the later layers, per lane (a lane past the last detection carries zero gradients) ----
        float gl[NOUT];
#pragma unroll
        for (int r = 0; r < NOUT; ++r) gl[r] = d < D ? gout[p * NOUT + r] : 0.0f;
        float g3[E3P];
#pragma unroll
        for (int q = 0; q < E3P; ++q) g3[q] = 0.0f;
        if (E4) {
            L[M::G4 * RS + lane] = gl[0];
#pragma unroll
            for (int q = 0; q < E3; ++q) {
                g3[q] = h3[q] > 0.0f ? tab[M::TW4 + q] * gl[0] : 0.0f;
                L[(32 + M::H3 + q) * RS + lane] = h3[q];
            }
        } else {
#pragma unroll
            for (int q = 0; q < E3; ++q) g3[q] = gl[q];
        }
#pragma unroll
        for (int q = 0; q < E3; ++q) L[(M::G3 + q) * RS + lane] = g3[q];
        f32x4 gh2[KG2];
#pragma unroll
        for (int ob = 0; ob < KG2; ++ob) {
            gh2[ob] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int qg = 0; qg < KG3; ++qg) {
                const f32x4 a = *reinterpret_cast<const f32x4*>(trow + M::T3B + (ob * KG3 + qg) * 16);
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) gh2[ob] = PB_MFMA4(a[kk], g3[4 * qg + kk], gh2[ob]);
            }
        }
        float g2[E2P];
#pragma unroll
        for (int o = 0; o < E2P; ++o) {
            g2[o] = h2[o] > 0.0f ? gh2[o / 4][o % 4] : 0.0f;
            if (o < E2) {
                L[(M::G2 + o) * RS + lane] = g2[o];
                L[(32 + M::H2 + o) * RS + lane] = h2[o];
            }
        }
        // ---- weight / bias gradients: sum over the 64 pairs of the tile on the matrix pipe ----
#pragma unroll 4
        for (int s = 0; s < ((PB_ABL & 1) ? 0 : 16); ++s) {
            float av[MB], bv[NBH];
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) av[mb] = L[16 * mb * RS + ka + 4 * s];
#pragma unroll
            for (int nb = 0; nb < NBH; ++nb) bv[nb] = L[(32 + 16 * nb) * RS + ka + 4 * s];
#pragma unroll
            for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                for (int nb = 0; nb < NBH; ++nb)
                    if (M::needed(mb, nb)) acc[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[mb], bv[nb], acc[mb][nb], 0, 0, 0);
        }
        // ---- first layer: gz1 = (W2^T g2) . [h1 > 0]; into this lane's gUC row, and over h1 in the tile for the sum over the lanes ----
        constexpr int IG = (KG1 % 4 == 0) ? 4 : 2;  // input blocks in flight (KG1 = 2, 8, 10, 16, 18)
#pragma unroll
        for (int ib0 = 0; ib0 < KG1; ib0 += IG) {
            f32x4 gh[IG];
#pragma unroll
            for (int j = 0; j < IG; ++j) gh[j] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int og = 0; og < ((PB_ABL & 2) ? 0 : KG2); ++og) {
                f32x4 a[IG];
#pragma unroll
                for (int j = 0; j < IG; ++j) a[j] = *reinterpret_cast<const f32x4*>(trow + M::T2B + ((ib0 + j) * KG2 + og) * 16);
#pragma unroll
                for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                    for (int j = 0; j < IG; ++j) gh[j] = PB_MFMA4(a[j][kk], g2[4 * og + kk], gh[j]);
            }
#pragma unroll
            for (int j = 0; j < IG; ++j) {
                const int ib = ib0 + j;
                const f32x4 u = *reinterpret_cast<const f32x4*>(upr + 4 * ib);
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    const float gz = (u[kk] + uc[4 * ib + kk]) > 0.0f ? gh[j][kk] : 0.0f;
                    guc[4 * ib + kk] += gz;
                    L[(32 + M::H1 + 4 * ib + kk) * RS + lane] = gz;
                }
            }
        }
#pragma unroll
        for (int r0 = 0; r0 < E1; r0 += 64) {
            if (r0 + lane < E1) {
                const f32x4* row = reinterpret_cast<const f32x4*>(L + (32 + M::H1 + r0 + lane) * RS);
                f32x4 s4 = {0.0f, 0.0f, 0.0f, 0.0f};  // four chains, combined in a fixed order
#pragma unroll
                for (int j = 0; j < ((PB_ABL & 4) ? 1 : 16); ++j) s4 += row[j];
                gup_part[(((size_t)dt * B + b) * T + t) * E1 + r0 + lane] = (s4[0] + s4[1]) + (s4[2] + s4[3]);
            }
        }
    }
    if (!BWD) return;
    if (d < D) {
        f32x4* dst = reinterpret_cast<f32x4*>(guc_part + (((size_t)slice * B + b) * D + d) * E1);
#pragma unroll
        for (int i = 0; i < E1 / 4; ++i) {
            const f32x4 v = {guc[4 * i], guc[4 * i + 1], guc[4 * i + 2], guc[4 * i + 3]};
            dst[i] = v;
        }
    }
    // this wave's part of the weight image (D layout of a 16x16 tile: row = 4 (lane / 16) + r, column = lane % 16)
    float* wp = w_part + (((size_t)b * gridDim.x + dt) * nslice + slice) * M::NE;
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
    for (int nb = 0; nb < NBH; ++nb) {
        if (!M::needed(mb, nb)) continue;
        const int hrow = 16 * nb + (lane & 15);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = 16 * mb + 4 * (lane >> 4) + r;
            int idx = -1;
            if (m < E2) {
                if (hrow < E1) idx = M::OW2 + m * E1 + hrow;
                else if (hrow == M::ONE) idx = M::OB2 + m;
            } else if (m < E2 + E3) {
                const int q = m - E2;
                if (hrow >= M::H2 && hrow < M::H2 + E2) idx = M::OW3 + q * E2 + (hrow - M::H2);
                else if (hrow == M::ONE) idx = M::OB3 + q;
            } else if (E4 && m < M::EG) {
                const int r4 = m - E2 - E3;
                if (hrow >= M::H3 && hrow < M::H3 + E3) idx = M::OW4 + r4 * E3 + (hrow - M::H3);
                else if (hrow == M::ONE) idx = M::OB4 + r4;
            }
            if (idx >= 0) wp[idx] = acc[mb][nb][r];
        }
    }
}

// out[i] = part[0][i] + part[1][i] + ... in this order
__global__ __launch_bounds__(256) void sum_slices_kernel(const float* __restrict__ part, int nslices, long n, float* __restrict__ out) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float s = 0.0f;
    for (int k = 0; k < nslices; ++k) s += part[(size_t)k * n + i];
    out[i] = s;
}

// the same for many slices of a short vector (the weight image: one slice per wave of the backward kernel): 16 elements x 16 slice
// groups per block; a group sums its slices g, g + 16, ... in order, the groups are combined in order
__global__ __launch_bounds__(256) void sum_many_slices_kernel(const float* __restrict__ part, int nslices, int n, float* __restrict__ out) {
    __shared__ float red[16][17];
    const int e = threadIdx.x & 15, g = threadIdx.x >> 4;
    const int i = blockIdx.x * 16 + e;
    float s = 0.0f;
    if (i < n) {
#pragma unroll 8
        for (int k = g; k < nslices; k += 16) s += part[(size_t)k * n + i];
    }
    red[g][e] = s;
    __syncthreads();
    if (g == 0 && i < n) {
        float t = red[0][e];
#pragma unroll
        for (int k = 1; k < 16; ++k) t += red[k][e];
        out[i] = t;
    }
}

namespace {

struct PbPlan {
    int nd, ny, tw, nslice;
    size_t gup, guc, wpart;  // floats
};

// the tracks dealt to ny workgroups of four waves each: enough waves for one per SIMD of the chip, no fewer than four tracks per wave
PbPlan pb_plan(int B, int T, int D, int E1, int NE) {
    PbPlan p;
    p.nd = cdiv(D, 64);
    int ny = 1;
    while ((long)B * p.nd * ny * PB_WPB < 1024 && cdiv(T, ny * 2 * PB_WPB) >= 4) ny *= 2;
    p.tw = cdiv(T, ny * PB_WPB);
    p.ny = cdiv(T, p.tw * PB_WPB);
    p.nslice = p.ny * PB_WPB;
    p.gup = (size_t)p.nd * B * T * E1;
    p.guc = (size_t)p.nslice * B * D * E1;
    p.wpart = (size_t)B * p.nd * p.nslice * NE;
    return p;
}

template <class M>
size_t pb_workspace_floats(int B, int T, int D) {
    const PbPlan p = pb_plan(B, T, D, M::E1, M::NE);
    return p.gup + p.guc + p.wpart;
}

template <class M>
int pb_forward(const float* UP, const float* UC, const float* const* w, int B, int T, int D, float* out, hipStream_t st) {
    const PbPlan p = pb_plan(B, T, D, M::E1, M::NE);
    hipLaunchKernelGGL((pair_mlp_kernel<M, false>), dim3(p.nd, p.ny, B), dim3(64 * PB_WPB), pb_lds_bytes<M>(false), st, UP, UC, w[0], w[1], w[2], w[3], w[4], w[5],
                       (const float*)nullptr, out, (float*)nullptr, (float*)nullptr, (float*)nullptr, B, T, D, p.tw);
    return check_launch("pair_mlp_forward");
}

template <class M>
int pb_backward(const float* UP, const float* UC, const float* const* w, const float* gout, int B, int T, int D, float* gUP, float* gUC,
                float* gW, float* ws, size_t ws_bytes, hipStream_t st) {
    const PbPlan p = pb_plan(B, T, D, M::E1, M::NE);
    if (ws_bytes < (p.gup + p.guc + p.wpart) * sizeof(float)) {
        set_error_msg("pair_mlp_backward: workspace too small (shasta_pair_mlp_workspace_bytes)");
        return SHASTA_E_ARG;
    }
    const size_t lds = pb_lds_bytes<M>(true);
    if (hipFuncSetAttribute((const void*)pair_mlp_kernel<M, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
        (void)hipGetLastError();
        set_error_msg("pair_mlp_backward: the device does not grant the kernel's LDS per workgroup");
        return SHASTA_E_UNSUPPORTED;
    }
    float *gup = ws, *guc = ws + p.gup, *wp = guc + p.guc;
    hipLaunchKernelGGL((pair_mlp_kernel<M, true>), dim3(p.nd, p.ny, B), dim3(64 * PB_WPB), lds, st, UP, UC, w[0], w[1], w[2], w[3], w[4], w[5],
                       gout, (float*)nullptr, gup, guc, wp, B, T, D, p.tw);
    int rc = check_launch("pair_mlp_backward");
    if (rc) return rc;
    const long n1 = (long)B * T * M::E1, n2 = (long)B * D * M::E1;
    hipLaunchKernelGGL(sum_slices_kernel, dim3((unsigned)((n1 + 255) / 256)), dim3(256), 0, st, gup, p.nd, n1, gUP);
    hipLaunchKernelGGL(sum_slices_kernel, dim3((unsigned)((n2 + 255) / 256)), dim3(256), 0, st, guc, p.nslice, n2, gUC);
    hipLaunchKernelGGL(sum_many_slices_kernel, dim3(cdiv(M::NE, 16)), dim3(256), 0, st, wp, B * p.nd * p.nslice, (int)M::NE, gW);
    return check_launch("pair_mlp_backward sums");
}

// kind 0 = fuse_shape (2F -> F/8 -> F/16 -> F/32 -> 1), 1 = fuse_det (2 nf -> 32 -> 8 -> 1), 2 = res_coeff (2 nf + 2F -> 32 + F/8 -> 8 + F/32 -> 3)
template <class Fn>
int pb_dispatch(int kind, int F, Fn&& fn) {
    if (kind == 1) return fn(PairMlp<32, 8, 1, 0>());
    if (F == 64) return kind == 0 ? fn(PairMlp<8, 4, 2, 1>()) : fn(PairMlp<40, 10, 3, 0>());
    if (F == 256) return kind == 0 ? fn(PairMlp<32, 16, 8, 1>()) : fn(PairMlp<64, 16, 3, 0>());
    if (F == 320) return kind == 0 ? fn(PairMlp<40, 20, 10, 1>()) : fn(PairMlp<72, 18, 3, 0>());
    set_error_msg("pair_mlp: feat_dim must be 64, 256 or 320 (the dense formulation serves the others)");
    return SHASTA_E_UNSUPPORTED;
}

}  // namespace
}  // namespace shasta

using namespace shasta;

extern "C" int shasta_pair_mlp_supported(int feat_dim) { return feat_dim == 64 || feat_dim == 256 || feat_dim == 320; }

extern "C" int shasta_pair_mlp_grad_floats(int kind, int feat_dim) {
    int n = 0;
    if (kind < 0 || kind > 2) return 0;
    if (pb_dispatch(kind, feat_dim, [&](auto m) { n = decltype(m)::NE; return SHASTA_OK; })) return 0;
    return n;
}

extern "C" size_t shasta_pair_mlp_workspace_bytes(int kind, int feat_dim, int B, int T, int D) {
    size_t n = 0;
    if (kind < 0 || kind > 2 || B <= 0 || T <= 0 || D <= 0) return 0;
    if (pb_dispatch(kind, feat_dim, [&](auto m) { n = pb_workspace_floats<decltype(m)>(B, T, D); return SHASTA_OK; })) return 0;
    return n * sizeof(float);
}

extern "C" int shasta_pair_mlp_forward_f32(int kind, int feat_dim, const float* UP, const float* UC, const float* const* wt, int B, int T,
                                           int D, float* out, shasta_stream_t stream) {
    SHASTA_REQUIRE(kind >= 0 && kind <= 2 && UP && UC && wt && out && B >= 0 && T > 0 && D > 0, "pair_mlp_forward: bad argument");
    if (B == 0) return SHASTA_OK;
    return pb_dispatch(kind, feat_dim, [&](auto m) { return pb_forward<decltype(m)>(UP, UC, wt, B, T, D, out, as_stream(stream)); });
}

extern "C" int shasta_pair_mlp_backward_f32(int kind, int feat_dim, const float* UP, const float* UC, const float* const* wt,
                                            const float* gout, int B, int T, int D, float* gUP, float* gUC, float* gW, float* ws,
                                            size_t ws_bytes, shasta_stream_t stream) {
    SHASTA_REQUIRE(kind >= 0 && kind <= 2 && UP && UC && wt && gout && gUP && gUC && gW && ws && B > 0 && T > 0 && D > 0,
                   "pair_mlp_backward: bad argument");
    return pb_dispatch(kind, feat_dim,
                       [&](auto m) { return pb_backward<decltype(m)>(UP, UC, wt, gout, B, T, D, gUP, gUC, gW, ws, ws_bytes, as_stream(stream)); });
}
