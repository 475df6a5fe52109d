// K0 in train() mode: shared_conv of the affinity network as the reference trains it (det3d/models/tracker/shasta.py:42-47 applied
// :223-228; tools/nusc_shasta/train.py:183-191 freezes children 1, 2 = backbone, neck only and keeps every BatchNorm in train mode)
//   y = Conv2d(Cin -> 64, 3x3, padding 1, bias)(x)          csrc/shared_conv_f16.hip / shared_conv.hip with a RAW pack -> (maps, H, W, 64)
//   out = ReLU(BatchNorm2d(64)(y)) with BATCH statistics     bn_stats / bn_finalize / bn_relu_apply below (one BatchNorm call per map:
//                                                            current, then previous - two updates of the running statistics per step)
// and its backward for a FROZEN producer of x (no dgrad):
//   g' = gout [out > 0];  dgamma = sum g' xhat, dbeta = sum g';  dy = gamma invstd (g' - mean(g') - xhat mean(g' xhat))
//   dW[o][c][ky][kx] = sum_p dy[p][o] x[c][p + (ky-1, kx-1)],  dbias = sum_p dy[p][o]  (zero up to rounding: BatchNorm follows)
// The weight gradient is an implicit GEMM with M = 64 output channels, N = 9 Cin, K = all pixels of all maps (19.1 GFLOP per map at
// 512 x 180 x 180) on the fp16 matrix path in the two-piece form of the forward kernel: every fp32 product from three fp16 products
// of range-scaled pieces (x: one power of two per image - the forward's image maxima; dy: one per output channel and BatchNorm call,
// from a bound the reduction pass computes), fp32 accumulation, K split over workgroups with a fixed-order reduction (deterministic).
//
// conv_wgrad_kernel: a workgroup = 32 input channels x 64 output channels x 9 taps for the rows [r0, r1) of one image.
//  * K runs along image rows: a k-step = 16 consecutive columns of one row (rows padded to a multiple of 16 columns with at least
//    one zero column, so the x-boundary of the convolution needs no masks: a shifted read of x lands on a zero of x or meets a zero
//    of dy); lane half h of an operand fragment holds columns 8h .. 8h+7.
//  * A = dy^T arrives from HBM already in FRAGMENT ORDER (bn_relu_bwd_dy_kernel writes it cut into pieces, one 1 KB block per
//    (row, k-step, channel block, piece)): a coalesced 16-byte load per lane, no LDS.
//  * B = x: the workgroup keeps four padded rows of its 32 channels in LDS as fp16 pieces ([piece][channel][column], three live rows +
//    the incoming one; each image row is read from HBM and cut exactly once per channel block).  The tap's row offset selects the LDS
//    row, its column offset -1 / 0 / +1 is applied in registers: one aligned ds_read_b128 + the two neighbouring dwords, then four
//    v_alignbit_b32 per shifted fragment (a 2-byte shift cannot be had from an aligned 16-byte read).
//  * the eight waves = 2 output-channel blocks x 4 interleaved quarters of the k-steps; a wave holds 9 accumulators of 32 x 32 (one
//    per tap).  Per k-step 27 MFMAs (9 taps x 3 piece products) against 6 ds_read_b128 + 12 ds_read_b32 + 48 v_alignbit: the kernel is
//    bound by the matrix pipe.  The quarters are summed through LDS in a fixed order at the end, scaled back exactly and written as
//    this split's partial; conv_wgrad_reduce_kernel adds the splits in order.
//  * workgroups that read the same dy rows (the 16 channel blocks of a split) are neighbours on one XCD: dy crosses HBM once.
#include "common.hpp"

#include <string.h>

namespace shasta {

typedef _Float16 t16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 t16x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

constexpr int BN_C = 64;          // channels of shared_conv
constexpr int BN_SLICES = 1024;   // workgroups (and partials) of a reduction pass at most

__device__ __forceinline__ uint32_t tr_pack2h(_Float16 even, _Float16 odd) {
    const t16x2 v = {even, odd};
    return __builtin_bit_cast(uint32_t, v);
}
__device__ __forceinline__ void tr_cut2(float a, _Float16& h, _Float16& l) {
    h = (_Float16)a;
    l = (_Float16)(a - (float)h);
}

// ---- batch statistics of y (M, 64): per channel sum and sum of squares in float64 (products of fp32 values are exact in float64),
// partials per workgroup, combined in order; out: mean[64], M2[64] = sum of squared deviations from that mean ------------------------
__global__ __launch_bounds__(256) void bn_stats_kernel(const float* __restrict__ y, long M, double* __restrict__ part) {
    __shared__ double red[2][4][BN_C];
    const int c = threadIdx.x & 63, g = threadIdx.x >> 6;
    const long per = (M + gridDim.x - 1) / gridDim.x, beg = blockIdx.x * per, end = min(M, beg + per);
    double s = 0.0, ss = 0.0;
    // eight rows' loads in flight, then the adds in the rows' order (the same sums as a row at a time)
    for (long p0 = beg + g; p0 < end; p0 += 32) {
        float t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) t[u] = y[min(p0 + 4 * u, end - 1) * BN_C + c];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (p0 + 4 * u < end) {
                const double v = (double)t[u];
                s += v;
                ss += v * v;
            }
        }
    }
    red[0][g][c] = s;
    red[1][g][c] = ss;
    __syncthreads();
    if (g == 0) {
        part[((size_t)blockIdx.x * 2 + 0) * BN_C + c] = (red[0][0][c] + red[0][1][c]) + (red[0][2][c] + red[0][3][c]);
        part[((size_t)blockIdx.x * 2 + 1) * BN_C + c] = (red[1][0][c] + red[1][1][c]) + (red[1][2][c] + red[1][3][c]);
    }
}

// the partials of a pass are combined by ONE workgroup of 16 x 64 threads: thread (g, c) adds the slices i = g, g + 16, ... of channel
// c in order, the 16 group sums are then added in order (fixed summation tree: the same bits on every run)
template <int NV>
__device__ __forceinline__ void combine_slices(const double* __restrict__ part, int nslice, int stride, double (&v)[NV], bool (&is_max)[NV]) {
    __shared__ double red[NV][16][BN_C];
    const int c = threadIdx.x & 63, g = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < NV; ++k) v[k] = 0.0;
    // (eight slices' loads in flight at a time: the adds keep their order, the loop is otherwise one memory latency per slice)
    for (int i0 = g; i0 < nslice; i0 += 16 * 8) {
        double t[8][NV];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = min(i0 + 16 * u, nslice - 1);
#pragma unroll
            for (int k = 0; k < NV; ++k) t[u][k] = part[(size_t)i * stride + k * BN_C + c];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (i0 + 16 * u < nslice) {
#pragma unroll
                for (int k = 0; k < NV; ++k) v[k] = is_max[k] ? fmax(v[k], t[u][k]) : v[k] + t[u][k];
            }
        }
    }
#pragma unroll
    for (int k = 0; k < NV; ++k) red[k][g][c] = v[k];
    __syncthreads();
    if (g == 0) {
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            double t = red[k][0][c];
            for (int j = 1; j < 16; ++j) t = is_max[k] ? fmax(t, red[k][j][c]) : t + red[k][j][c];
            v[k] = t;
        }
    }
}

__global__ __launch_bounds__(1024) void bn_stats_final_kernel(const double* __restrict__ part, int nslice, long M, float* __restrict__ mean_m2) {
    double v[2];
    bool mx[2] = {false, false};
    combine_slices<2>(part, nslice, 2 * BN_C, v, mx);
    if (threadIdx.x < BN_C) {
        const int c = threadIdx.x;
        const double mean = v[0] / (double)M;
        const double m2 = v[1] - v[0] * mean;
        mean_m2[c] = (float)mean;
        mean_m2[BN_C + c] = (float)(m2 > 0.0 ? m2 : 0.0);
    }
}

// mean / M2 of the whole batch (merged over the ranks by the caller when the BatchNorm is synchronised) -> what the normalisation
// uses (mean, 1 / sqrt(biased variance + eps)) and the running statistics as nn.BatchNorm2d keeps them (unbiased variance)
__global__ __launch_bounds__(64) void bn_finalize_kernel(const float* __restrict__ mean_m2, double n, float eps, float momentum,
                                                         float* __restrict__ stat, float* __restrict__ running_mean,
                                                         float* __restrict__ running_var, long* __restrict__ nbt) {
    const int c = threadIdx.x;
    const float mean = mean_m2[c], m2 = mean_m2[BN_C + c];
    const float var = (float)((double)m2 / n);
    stat[c] = mean;
    stat[BN_C + c] = 1.0f / sqrtf(var + eps);
    if (running_mean) running_mean[c] = (1.0f - momentum) * running_mean[c] + momentum * mean;
    if (running_var) running_var[c] = (1.0f - momentum) * running_var[c] + momentum * (float)((double)m2 / (n > 1.0 ? n - 1.0 : 1.0));
    if (nbt && c == 0) nbt[0] += 1;
}

// out = relu((y - mean) invstd gamma + beta), four channels per thread
__global__ __launch_bounds__(256) void bn_relu_apply_kernel(const float* __restrict__ y, long n4, const float* __restrict__ stat,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            float* __restrict__ out) {
    const int c4 = (threadIdx.x & 15) * 4;
    float mean[4], inv[4], ga[4], be[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        mean[j] = stat[c4 + j];
        inv[j] = stat[BN_C + c4 + j];
        ga[j] = gamma[c4 + j];
        be[j] = beta[c4 + j];
    }
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {  // i & 15 == threadIdx.x & 15
        const f32x4 v = reinterpret_cast<const f32x4*>(y)[i];
        f32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = relu_nan(((v[j] - mean[j]) * inv[j]) * ga[j] + be[j]);
        reinterpret_cast<f32x4*>(out)[i] = o;
    }
}

// ---- backward, pass 1: per channel sum g', sum g' xhat (float64), max |g'|, max |xhat| ------------------------------------------------
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const float* __restrict__ y, const float* __restrict__ gout, long M,
                                                            const float* __restrict__ stat, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, double* __restrict__ part) {
    __shared__ double red[4][4][BN_C];
    const int c = threadIdx.x & 63, g = threadIdx.x >> 6;
    const long per = (M + gridDim.x - 1) / gridDim.x, beg = blockIdx.x * per, end = min(M, beg + per);
    const float mean = stat[c], inv = stat[BN_C + c], ga = gamma[c], be = beta[c];
    double s1 = 0.0, s2 = 0.0;
    float mg = 0.0f, mx = 0.0f;
    // four rows' loads of both tensors in flight (the incoming gradient is read whether the ReLU passed or not: a load behind the
    // comparison would wait for y first), then the sums in the rows' order
    for (long p0 = beg + g; p0 < end; p0 += 16) {
        float ty[4], tg[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long p = min(p0 + 4 * u, end - 1);
            ty[u] = y[p * BN_C + c];
            tg[u] = gout[p * BN_C + c];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (p0 + 4 * u < end) {
                const float xh = (ty[u] - mean) * inv;
                const float pre = xh * ga + be;
                const float gp = pre > 0.0f ? tg[u] : 0.0f;
                s1 += (double)gp;
                s2 += (double)gp * (double)xh;
                mg = absmax_keep_nan(mg, fabsf(gp));
                mx = absmax_keep_nan(mx, fabsf(xh));
            }
        }
    }
    red[0][g][c] = s1;
    red[1][g][c] = s2;
    red[2][g][c] = (double)mg;
    red[3][g][c] = (double)mx;
    __syncthreads();
    if (g == 0) {
        double* o = part + (size_t)blockIdx.x * 4 * BN_C + c;
        o[0] = (red[0][0][c] + red[0][1][c]) + (red[0][2][c] + red[0][3][c]);
        o[BN_C] = (red[1][0][c] + red[1][1][c]) + (red[1][2][c] + red[1][3][c]);
        o[2 * BN_C] = fmax(fmax(red[2][0][c], red[2][1][c]), fmax(red[2][2][c], red[2][3][c]));
        o[3 * BN_C] = fmax(fmax(red[3][0][c], red[3][1][c]), fmax(red[3][2][c], red[3][3][c]));
    }
}

__global__ __launch_bounds__(1024) void bn_bwd_reduce_final_kernel(const double* __restrict__ part, int nslice, float* __restrict__ sums) {
    double v[4];
    bool mx[4] = {false, false, true, true};
    combine_slices<4>(part, nslice, 4 * BN_C, v, mx);
    if (threadIdx.x < BN_C) {
#pragma unroll
        for (int k = 0; k < 4; ++k) sums[k * BN_C + threadIdx.x] = (float)v[k];
    }
}

// ---- backward, pass 2: dy of one image row, scaled per channel, cut into fp16 pieces, written in the fragment order of the weight-
// gradient kernel: block ((row id * KS + k) * 2 + channel block) * 2 + piece, 1 KB each: lane (m = channel in the block, h) holds
// columns 16 k + 8 h .. + 7.  grid = rows of all images of this BatchNorm call; row id = img0 * H + blockIdx.x -------------------------
constexpr int DY_MAXPW = 256;
__global__ __launch_bounds__(256) void bn_bwd_dy_kernel(const float* __restrict__ y, const float* __restrict__ gout, int H, int W, int PW,
                                                        long row0, const float* __restrict__ stat, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, const float* __restrict__ sums_global,
                                                        const float* __restrict__ sums_local, double n_global, char* __restrict__ frag,
                                                        float* __restrict__ edy_out, double* __restrict__ dbias_part) {
    extern __shared__ __attribute__((aligned(16))) float tile[];  // [PW][65] scaled dy, then the coefficient arrays
    float* coef = tile + (size_t)PW * 65;                        // a[64], mg[64], mgx[64], scale[64]
    __shared__ double red[4][BN_C];
    const int tid = threadIdx.x, c = tid & 63, g = tid >> 6;
    const long row = blockIdx.x;  // row of this call's images: (image, r) = (row / H, row % H)
    if (tid < BN_C) {
        const float a = gamma[tid] * stat[BN_C + tid];
        const float mg = (float)((double)sums_global[tid] / n_global), mgx = (float)((double)sums_global[BN_C + tid] / n_global);
        // |dy| <= |a| (max |g'| + |mean g'| + max |xhat| |mean g' xhat|): the scale puts that bound into (2^13, 2^14]
        const float bound = fabsf(a) * (sums_local[2 * BN_C + tid] + fabsf(mg) + sums_local[3 * BN_C + tid] * fabsf(mgx));
        const int e = range_exponent_bits(__float_as_uint(bound));
        coef[tid] = a;
        coef[BN_C + tid] = mg;
        coef[2 * BN_C + tid] = mgx;
        coef[3 * BN_C + tid] = __builtin_ldexpf(1.0f, e);
        if (blockIdx.x == 0) edy_out[tid] = (float)e;
    }
    __syncthreads();
    const float mean = stat[c], inv = stat[BN_C + c], ga = gamma[c], be = beta[c];
    const float a = coef[c], mg = coef[BN_C + c], mgx = coef[2 * BN_C + c], sc = coef[3 * BN_C + c];
    const float* yr = y + row * W * BN_C;
    const float* gr = gout + row * W * BN_C;
    double sdy = 0.0;
    for (int px0 = g; px0 < PW; px0 += 16) {  // four pixels' loads of both tensors in flight (see bn_bwd_reduce_kernel)
        float ty[4], tg[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int px = min(px0 + 4 * u, W - 1);
            ty[u] = yr[(size_t)px * BN_C + c];
            tg[u] = gr[(size_t)px * BN_C + c];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int px = px0 + 4 * u;
            if (px < PW) {
                float dy = 0.0f;
                if (px < W) {
                    const float xh = (ty[u] - mean) * inv;
                    const float pre = xh * ga + be;
                    const float gp = pre > 0.0f ? tg[u] : 0.0f;
                    dy = ((gp - mg) - xh * mgx) * a;
                    sdy += (double)dy;
                }
                tile[px * 65 + c] = dy * sc;
            }
        }
    }
    red[g][c] = sdy;
    __syncthreads();
    if (g == 0) dbias_part[(size_t)row * BN_C + c] = (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]);
    const int KS = PW >> 4;
    char* out = frag + (size_t)(row0 + row) * KS * 4096;
    for (int it = tid; it < KS * 128; it += 256) {
        const int lane = it & 63, ob = (it >> 6) & 1, k = it >> 7;
        const int m = lane & 31, h = lane >> 5;
        const float* src = tile + (16 * k + 8 * h) * 65 + 32 * ob + m;
        u32x4 hi, lo;
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            _Float16 h0, l0, h1, l1;
            tr_cut2(src[(2 * jj) * 65], h0, l0);
            tr_cut2(src[(2 * jj + 1) * 65], h1, l1);
            hi[jj] = tr_pack2h(h0, h1);
            lo[jj] = tr_pack2h(l0, l1);
        }
        char* f = out + ((size_t)(k * 2 + ob) * 2) * 1024 + lane * 16;
        *reinterpret_cast<u32x4*>(f) = hi;
        *reinterpret_cast<u32x4*>(f + 1024) = lo;
    }
}

// per channel: sum of the rows' partials (same fixed tree) -> out[c] (+= when accumulate)
__global__ __launch_bounds__(1024) void colsum_f64_kernel(const double* __restrict__ part, int rows, float* __restrict__ out, int accumulate) {
    double v[1];
    bool mx[1] = {false};
    combine_slices<1>(part, rows, BN_C, v, mx);
    if (threadIdx.x < BN_C) out[threadIdx.x] = accumulate ? out[threadIdx.x] + (float)v[0] : (float)v[0];
}

// ---- the weight gradient ------------------------------------------------------------------------------------------------------------
constexpr int WG_CB = 32;      // input channels per workgroup
constexpr int WG_RING = 4;     // padded x rows in LDS: three live + the incoming one
struct WgradArgs {
    const float* x[2];         // current / previous neck outputs (B, Cin, H, W)
    const unsigned* xmax;      // [nimg] bit patterns of the image maxima (the forward's)
    const char* frag;          // dy pieces in fragment order, all images
    const float* edy;          // [2][64] scale exponents of dy per BatchNorm call (current, previous)
    float* part;               // [nsplit][64][Cin * 9]
    int B, nimg, Cin, H, W, PW, KS, ROWB, cblocks, rsplit, rows_per_split, total;
};

template <int MAXKQ>  // k-steps of a wave per image row = padded row width / 64 (rows are padded to whole multiples of 64 columns, so
                      // that every wave runs the same branch-free sequence): 3 for the 180-column maps
__global__ __launch_bounds__(512, 2) void conv_wgrad_kernel(WgradArgs a) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ob = wv & 1, kq = wv >> 1;
    // block -> (split, channel block): the workgroups of one XCD (block ids 8 apart) take consecutive logical ids, channel block fastest
    unsigned L = blockIdx.x;
    {
        const unsigned per = (unsigned)a.total >> 3;
        if (L < per * 8) L = (L & 7) * per + (L >> 3);
    }
    const int split = (int)(L / (unsigned)a.cblocks), cb = (int)(L - (unsigned)split * a.cblocks);
    const int z = split / a.rsplit, rs = split - z * a.rsplit;
    const int r0 = rs * a.rows_per_split, r1 = min(a.H, r0 + a.rows_per_split);
    const int H = a.H, W = a.W, KS = a.KS, ROWB = a.ROWB, Cin = a.Cin;
    const bool second = z >= a.B;
    const float* xin = (second ? a.x[1] + (size_t)(z - a.B) * Cin * H * W : a.x[0] + (size_t)z * Cin * H * W);
    const int PIECE = WG_CB * ROWB, SLOT = 2 * PIECE;  // bytes of one piece plane / one ring slot
    const int ex = range_exponent_bits(a.xmax[z]);
    const float xscale = __builtin_ldexpf(1.0f, ex);

    // zero the ring once: the padding columns are never written afterwards
    {
        const u32x4 zz = {0u, 0u, 0u, 0u};
        for (int i = tid; i < WG_RING * SLOT / 16; i += 512) reinterpret_cast<u32x4*>(lds)[i] = zz;
    }
    // staging role: channel tid >> 4 of the block, column pairs (tid & 15) + 16 i
    const int sc = tid >> 4, sl = tid & 15;
    const bool c_ok = cb * WG_CB + sc < Cin;
    const float* xc = xin + (size_t)(cb * WG_CB + (c_ok ? sc : 0)) * H * W;
    constexpr int MAXNI = 2 * MAXKQ;  // column pairs of a lane
    float sv[MAXNI][2];
    // (straight-line: every load is issued, from a clamped address - a branch per load would make each wait for the one before)
    auto load_row = [&](int row) __attribute__((always_inline)) {
        const bool ok = c_ok && row >= 0 && row < H;
        const float* xr = xc + (size_t)min(max(row, 0), H - 1) * W;
#pragma unroll
        for (int i = 0; i < MAXNI; ++i) {
            const int p = 2 * (sl + 16 * i);
            const float v0 = xr[min(p, W - 1)], v1 = xr[min(p + 1, W - 1)];
            sv[i][0] = (ok && p < W) ? v0 : 0.0f;
            sv[i][1] = (ok && p + 1 < W) ? v1 : 0.0f;
        }
    };
    auto store_row = [&](int row) __attribute__((always_inline)) {
        char* dst = lds + ((row + 1) & 3) * SLOT + sc * ROWB + 16;
#pragma unroll
        for (int i = 0; i < MAXNI; ++i) {
            const int p = 2 * (sl + 16 * i);
            if (p < W) {  // (the odd column beyond an odd W holds a zero: it is a padding column)
                _Float16 h0, l0, h1, l1;
                tr_cut2(sv[i][0] * xscale, h0, l0);
                tr_cut2(sv[i][1] * xscale, h1, l1);
                *reinterpret_cast<uint32_t*>(dst + 2 * p) = tr_pack2h(h0, h1);
                *reinterpret_cast<uint32_t*>(dst + PIECE + 2 * p) = tr_pack2h(l0, l1);
            }
        }
    };
    __syncthreads();
    for (int row = r0 - 1; row <= r0 + 1; ++row) {
        load_row(row);
        store_row(row);
    }

    const int n = lane & 31, h = lane >> 5;
    const f32x16 zero16 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) acc[t] = zero16;
    const char* fragz = a.frag + (size_t)z * H * KS * 4096 + (size_t)ob * 2048 + lane * 16;
    // the dy fragments of a k-step stay in registers for its nine taps; those of the NEXT row's k-step are requested into the same
    // registers as soon as the step is done, and have the rest of the row to arrive
    t16x8 ah[MAXKQ], al[MAXKQ];
    auto load_a = [&](int row, int i) __attribute__((always_inline)) {
        const char* f = fragz + ((size_t)row * KS + (kq + 4 * i)) * 4096;
        ah[i] = *reinterpret_cast<const t16x8*>(f);
        al[i] = *reinterpret_cast<const t16x8*>(f + 1024);
    };
#pragma unroll
    for (int i = 0; i < MAXKQ; ++i) load_a(r0, i);
    __syncthreads();
    const int b_lane = n * ROWB + 16 + 16 * h;
    for (int r = r0; r < r1; ++r) {
        // (unconditional: the last iteration stages a row and requests fragments nobody uses - cheaper than branches around loads)
        const int rnext = min(r + 1, r1 - 1);
        load_row(r + 2);
        __builtin_amdgcn_sched_barrier(0);  // the row's loads are issued HERE (left alone the scheduler sinks them to their first use)
#pragma unroll
        for (int i = 0; i < MAXKQ; ++i) {
            const int k = kq + 4 * i;
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const char* rowp = lds + ((r + ky) & 3) * SLOT + b_lane + 32 * k;  // row r + ky - 1 sits in slot (row + 1) & 3
                t16x8 bh[3], bl[3];
#pragma unroll
                for (int pc = 0; pc < 2; ++pc) {
                    const char* q = rowp + pc * PIECE;
                    const u32x4 w = *reinterpret_cast<const u32x4*>(q);
                    const uint32_t wm = *reinterpret_cast<const uint32_t*>(q - 4), wp = *reinterpret_cast<const uint32_t*>(q + 16);
                    const u32x4 left = {__builtin_amdgcn_alignbit(w[0], wm, 16), __builtin_amdgcn_alignbit(w[1], w[0], 16),
                                        __builtin_amdgcn_alignbit(w[2], w[1], 16), __builtin_amdgcn_alignbit(w[3], w[2], 16)};
                    const u32x4 right = {__builtin_amdgcn_alignbit(w[1], w[0], 16), __builtin_amdgcn_alignbit(w[2], w[1], 16),
                                         __builtin_amdgcn_alignbit(w[3], w[2], 16), __builtin_amdgcn_alignbit(wp, w[3], 16)};
                    t16x8* d = pc ? bl : bh;
                    d[0] = __builtin_bit_cast(t16x8, left);    // kx = 0: columns shifted by -1
                    d[1] = __builtin_bit_cast(t16x8, w);
                    d[2] = __builtin_bit_cast(t16x8, right);   // kx = 2: columns shifted by +1
                }
                // piece products small to large, the three taps of the row interleaved (a tap's accumulator is touched every third MFMA)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) acc[3 * ky + kx] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[kx], acc[3 * ky + kx], 0, 0, 0);
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) acc[3 * ky + kx] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[kx], acc[3 * ky + kx], 0, 0, 0);
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) acc[3 * ky + kx] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[kx], acc[3 * ky + kx], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            load_a(rnext, i);  // next row's fragments of this k-step: requested now, used a row's worth of MFMAs later
            __builtin_amdgcn_sched_barrier(0);
        }
        store_row(r + 2);  // into the slot of row r - 2, which nobody reads any more
        __syncthreads();
    }

    // scale back exactly: D[o][c], lane = column c = n, rows o = 32 ob + (rr & 3) + 8 (rr >> 2) + 4 h
    const float* edy = a.edy + (second ? BN_C : 0);
    float un[16];
#pragma unroll
    for (int rr = 0; rr < 16; ++rr) un[rr] = __builtin_ldexpf(1.0f, -ex - (int)edy[32 * ob + (rr & 3) + 8 * (rr >> 2) + 4 * h]);
    // the four quarters of a channel block's k-steps are summed through LDS in the order 0, 1, 2, 3, one tap at a time
    float* red = reinterpret_cast<float*>(lds);  // [8 waves][16][64]
    const int c = cb * WG_CB + n;
    float* pout = a.part + (size_t)split * BN_C * Cin * 9;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
#pragma unroll
        for (int rr = 0; rr < 16; ++rr) red[(wv * 16 + rr) * 64 + lane] = acc[t][rr] * un[rr];
        __syncthreads();
        if (kq == 0 && c < Cin) {
#pragma unroll
            for (int rr = 0; rr < 16; ++rr) {
                const float s = ((red[((ob + 0) * 16 + rr) * 64 + lane] + red[((ob + 2) * 16 + rr) * 64 + lane]) +
                                 red[((ob + 4) * 16 + rr) * 64 + lane]) + red[((ob + 6) * 16 + rr) * 64 + lane];
                const int o = 32 * ob + (rr & 3) + 8 * (rr >> 2) + 4 * h;
                pout[((size_t)o * Cin + c) * 9 + t] = s;
            }
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void conv_wgrad_reduce_kernel(const float* __restrict__ part, int nsplit, long count, float* __restrict__ dw) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= count) return;
    float s = 0.0f;
    for (int k = 0; k < nsplit; ++k) s += part[(size_t)k * count + i];  // fixed order
    dw[i] = s;
}

static int wgrad_pw(int W) { return (W + 1 + 63) / 64 * 64; }  // whole multiples of 64 columns: 4 waves x 16 columns per round of k-steps
static int wgrad_rowb(int PW) {
    int b = 2 * PW + 32;
    if ((b / 16) % 2 == 0) b += 16;  // an odd number of 16-byte units per channel row: the 16-byte reads of 16 consecutive channels cover all banks
    return b;
}
static size_t frag_bytes(int nimg, int H, int W) { return (size_t)nimg * H * (wgrad_pw(W) / 16) * 4096; }
static void wgrad_split(int nimg, int Cin, int H, int& cblocks, int& rsplit, int& rows_per_split) {
    cblocks = cdiv(Cin, WG_CB);
    rsplit = 1;
    while (cblocks * nimg * rsplit < 256 && cdiv(H, rsplit * 2) >= 8) rsplit *= 2;
    rows_per_split = cdiv(H, rsplit);
    rsplit = cdiv(H, rows_per_split);
}

}  // namespace shasta

using namespace shasta;

extern "C" int shasta_conv_train_supported(int in_channels, int H, int W) {
    if (in_channels <= 0 || H <= 0 || W <= 0 || W > DY_MAXPW - 1) return 0;  // 255 columns: four rounds of k-steps per row
    if ((long)in_channels * H * W >= (1L << 31)) return 0;
    const int PW = wgrad_pw(W);
    return WG_RING * 2 * WG_CB * wgrad_rowb(PW) <= 150 * 1024 && (size_t)(PW * 65 + 4 * BN_C) * sizeof(float) <= 96 * 1024;
}

// bytes of: the reduction partials of the BatchNorm passes; the dy piece image of `nimg` maps; the weight gradient's split partials
extern "C" size_t shasta_bn_workspace_bytes(void) { return (size_t)BN_SLICES * 4 * BN_C * sizeof(double); }
extern "C" size_t shasta_conv_dy_bytes(int nimg, int H, int W) { return nimg <= 0 ? 0 : frag_bytes(nimg, H, W) + (size_t)nimg * H * BN_C * sizeof(double); }
extern "C" size_t shasta_conv_wgrad_workspace_bytes(int nimg, int in_channels, int H, int W) {
    if (nimg <= 0) return 0;
    int cblocks, rsplit, rps;
    wgrad_split(nimg, in_channels, H, cblocks, rsplit, rps);
    return (size_t)nimg * rsplit * BN_C * in_channels * 9 * sizeof(float);
}

static int bn_slices(long M) {
    long s = (M + 255) / 256;
    return (int)(s < 1 ? 1 : s > BN_SLICES ? BN_SLICES : s);
}

extern "C" int shasta_bn_stats_f32(const float* y, long M, float* mean_m2, void* workspace, size_t workspace_bytes, shasta_stream_t stream) {
    SHASTA_REQUIRE(y && mean_m2 && workspace && M > 0, "bn_stats: bad argument");
    if (workspace_bytes < shasta_bn_workspace_bytes()) {
        set_error_msg("bn_stats: workspace too small");
        return SHASTA_E_WORKSPACE;
    }
    const int ns = bn_slices(M);
    hipLaunchKernelGGL(bn_stats_kernel, dim3(ns), dim3(256), 0, as_stream(stream), y, M, static_cast<double*>(workspace));
    hipLaunchKernelGGL(bn_stats_final_kernel, dim3(1), dim3(1024), 0, as_stream(stream), static_cast<const double*>(workspace), ns, M, mean_m2);
    return check_launch("bn_stats");
}

extern "C" int shasta_bn_finalize_f32(const float* mean_m2, double n, float eps, float momentum, float* stat, float* running_mean,
                                      float* running_var, long* num_batches_tracked, shasta_stream_t stream) {
    SHASTA_REQUIRE(mean_m2 && stat && n >= 1.0, "bn_finalize: bad argument");
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(1), dim3(64), 0, as_stream(stream), mean_m2, n, eps, momentum, stat, running_mean, running_var,
                       num_batches_tracked);
    return check_launch("bn_finalize");
}

extern "C" int shasta_bn_relu_apply_f32(const float* y, long M, const float* stat, const float* gamma, const float* beta, float* out,
                                        shasta_stream_t stream) {
    SHASTA_REQUIRE(y && stat && gamma && beta && out && M > 0, "bn_relu_apply: bad argument");
    SHASTA_REQUIRE(((uintptr_t)y | (uintptr_t)out) % 16 == 0, "bn_relu_apply: y / out must be 16-byte aligned");
    const long n4 = M * (BN_C / 4);
    const long blocks = (n4 + 255) / 256;
    hipLaunchKernelGGL(bn_relu_apply_kernel, dim3((unsigned)(blocks > 8192 ? 8192 : blocks)), dim3(256), 0, as_stream(stream), y, n4, stat, gamma,
                       beta, out);
    return check_launch("bn_relu_apply");
}

extern "C" int shasta_bn_relu_bwd_reduce_f32(const float* y, const float* gout, long M, const float* stat, const float* gamma,
                                             const float* beta, float* sums, void* workspace, size_t workspace_bytes, shasta_stream_t stream) {
    SHASTA_REQUIRE(y && gout && stat && gamma && beta && sums && workspace && M > 0, "bn_relu_bwd_reduce: bad argument");
    if (workspace_bytes < shasta_bn_workspace_bytes()) {
        set_error_msg("bn_relu_bwd_reduce: workspace too small");
        return SHASTA_E_WORKSPACE;
    }
    const int ns = bn_slices(M);
    hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3(ns), dim3(256), 0, as_stream(stream), y, gout, M, stat, gamma, beta, static_cast<double*>(workspace));
    hipLaunchKernelGGL(bn_bwd_reduce_final_kernel, dim3(1), dim3(1024), 0, as_stream(stream), static_cast<const double*>(workspace), ns, sums);
    return check_launch("bn_relu_bwd_reduce");
}

extern "C" int shasta_bn_relu_bwd_dy_f16x2(const float* y, const float* gout, int nimg, int img0, int nimg_total, int H, int W, const float* stat,
                                           const float* gamma, const float* beta, const float* sums_global, const float* sums_local,
                                           double n_global, void* dy, size_t dy_bytes, float* edy, float* dbias, int accumulate_dbias,
                                           shasta_stream_t stream) {
    SHASTA_REQUIRE(y && gout && stat && gamma && beta && sums_global && sums_local && dy && edy && dbias, "bn_relu_bwd_dy: null pointer");
    SHASTA_REQUIRE(nimg > 0 && img0 >= 0 && img0 + nimg <= nimg_total && n_global >= 1.0, "bn_relu_bwd_dy: bad image range");
    SHASTA_REQUIRE(shasta_conv_train_supported(16, H, W), "bn_relu_bwd_dy: map too wide");
    if (dy_bytes < shasta_conv_dy_bytes(nimg_total, H, W)) {
        set_error_msg("bn_relu_bwd_dy: dy buffer too small");
        return SHASTA_E_WORKSPACE;
    }
    const int PW = wgrad_pw(W);
    const size_t lds = (size_t)(PW * 65 + 4 * BN_C) * sizeof(float);
    if (hipFuncSetAttribute((const void*)bn_bwd_dy_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
        (void)hipGetLastError();
        set_error_msg("bn_relu_bwd_dy: the device refuses the LDS tile");
        return SHASTA_E_LAUNCH;
    }
    double* part = reinterpret_cast<double*>(static_cast<char*>(dy) + frag_bytes(nimg_total, H, W)) + (size_t)img0 * H * BN_C;
    hipLaunchKernelGGL(bn_bwd_dy_kernel, dim3((unsigned)(nimg * H)), dim3(256), lds, as_stream(stream), y, gout, H, W, PW, (long)img0 * H, stat,
                       gamma, beta, sums_global, sums_local, n_global, static_cast<char*>(dy), edy, part);
    hipLaunchKernelGGL(colsum_f64_kernel, dim3(1), dim3(1024), 0, as_stream(stream), part, nimg * H, dbias, accumulate_dbias);
    return check_launch("bn_relu_bwd_dy");
}

extern "C" int shasta_conv_wgrad_f16x2(const float* x, const float* x_prev, int B, int in_channels, int H, int W, const unsigned* xmax,
                                       const void* dy, const float* edy, float* dweight, void* workspace, size_t workspace_bytes,
                                       shasta_stream_t stream) {
    SHASTA_REQUIRE(x && xmax && dy && edy && dweight && workspace && B > 0, "conv_wgrad: bad argument");
    SHASTA_REQUIRE(shasta_conv_train_supported(in_channels, H, W), "conv_wgrad: shape not served (map width)");
    const int nimg = x_prev ? 2 * B : B;
    if (workspace_bytes < shasta_conv_wgrad_workspace_bytes(nimg, in_channels, H, W)) {
        set_error_msg("conv_wgrad: workspace too small");
        return SHASTA_E_WORKSPACE;
    }
    WgradArgs a;
    a.x[0] = x;
    a.x[1] = x_prev;
    a.xmax = xmax;
    a.frag = static_cast<const char*>(dy);
    a.edy = edy;
    a.part = static_cast<float*>(workspace);
    a.B = B;
    a.nimg = nimg;
    a.Cin = in_channels;
    a.H = H;
    a.W = W;
    a.PW = wgrad_pw(W);
    a.KS = a.PW / 16;
    a.ROWB = wgrad_rowb(a.PW);
    wgrad_split(nimg, in_channels, H, a.cblocks, a.rsplit, a.rows_per_split);
    a.total = a.cblocks * nimg * a.rsplit;
    const int lds_ring = WG_RING * 2 * WG_CB * a.ROWB, lds_red = 8 * 16 * 64 * (int)sizeof(float);
    const int lds = lds_ring > lds_red ? lds_ring : lds_red;
    hipStream_t st = as_stream(stream);
    bool launched = false;
#define SHASTA_WGRAD(NK)                                                                                                             \
    if (a.KS == 4 * NK) {                                                                                                            \
        if (hipFuncSetAttribute((const void*)conv_wgrad_kernel<NK>, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) { \
            (void)hipGetLastError();                                                                                                 \
            set_error_msg("conv_wgrad: the device refuses the LDS ring");                                                            \
            return SHASTA_E_LAUNCH;                                                                                                  \
        }                                                                                                                            \
        hipLaunchKernelGGL(conv_wgrad_kernel<NK>, dim3((unsigned)a.total), dim3(512), lds, st, a);                                   \
        launched = true;                                                                                                             \
    }
    SHASTA_WGRAD(1)
    SHASTA_WGRAD(2)
    SHASTA_WGRAD(3)
    SHASTA_WGRAD(4)
#undef SHASTA_WGRAD
    SHASTA_REQUIRE(launched, "conv_wgrad: map width not served");
    int rc = check_launch("conv_wgrad");
    if (rc) return rc;
    const long count = (long)BN_C * in_channels * 9;
    hipLaunchKernelGGL(conv_wgrad_reduce_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, st, a.part, nimg * a.rsplit, count, dweight);
    return check_launch("conv_wgrad_reduce");
}
