// Training path (SURVEY.md 8(a) row 19 / BASELINE config 5): auxiliary kernels of the backward pass.
// The dense layers run on the strided matrix-core GEMM (gemm_f32.hip: Y = X W^T, dX = dY W, dW = dY^T X); this file holds
// what is not a GEMM: pair-tensor assembly and its transpose (the cat/expand of det3d/models/tracker/shasta.py:286-313
// and their autograd), the hand-designed residual and its gradient (:277-283), the combine (:319), the two softmax
// backward passes (:324-325), bias-gradient column sums, |x| backward and the gather's scatter-add.
// First version of the training path: the pair tensor IS materialised here (reference formulation), so that every saved
// activation is a plain matrix; the factorised backward is future work.  All reductions have a fixed order except the
// scatter-add into the BEV gradient, which uses float atomics (documented, gradient of inputs only).
#include "common.hpp"

namespace shasta {

// ---- pair tensor assembly: X[(b,t,d)][:] ---------------------------------------------------------------------------
// kind 0 (fuse_shape): [prev_feat_t F | feat_d F]
// kind 1 (res_coeff) : [prev_feat_t F | prev_box_t nf | feat_d F | det_box_d nf]
// kind 2 (fuse_det)  : [prev_box_t nf | det_box_d nf]
__device__ __forceinline__ void pair_col(int kind, int F, int nf, int c, int& src, int& off) {
    // src: 0 prev_feat, 1 prev_box, 2 feat, 3 det_box, -1 padding
    if (kind == 0) {
        if (c < F) { src = 0; off = c; } else if (c < 2 * F) { src = 2; off = c - F; } else src = -1;
    } else if (kind == 1) {
        if (c < F) { src = 0; off = c; }
        else if (c < F + nf) { src = 1; off = c - F; }
        else if (c < 2 * F + nf) { src = 2; off = c - F - nf; }
        else if (c < 2 * F + 2 * nf) { src = 3; off = c - 2 * F - nf; }
        else src = -1;
    } else {
        if (c < nf) { src = 1; off = c; } else if (c < 2 * nf) { src = 3; off = c - nf; } else src = -1;
    }
}

__global__ void pair_concat_kernel(const float* __restrict__ prev_feat, const float* __restrict__ feat,
                                   const float* __restrict__ prev_tab, const float* __restrict__ det_tab, int B, int T, int D,
                                   int F, int nf, int kind, int ld, float* __restrict__ X) {
    const long row = blockIdx.x;  // (b, t, d)
    const int d = row % D, t = (row / D) % T, b = row / ((long)T * D);
    for (int c = threadIdx.x; c < ld; c += blockDim.x) {
        int src, off = 0;
        pair_col(kind, F, nf, c, src, off);
        float v = 0.0f;
        if (src == 0) v = prev_feat[((size_t)b * T + t) * F + off];
        else if (src == 1) v = prev_tab[((size_t)b * T + t) * 8 + off];
        else if (src == 2) v = feat[((size_t)b * D + d) * F + off];
        else if (src == 3) v = det_tab[((size_t)b * D + d) * 8 + off];
        X[row * ld + c] = v;
    }
}

// transpose of the assembly: every table element sums the gradients of the pairs it was copied into (fixed order)
//   dprev_feat[b,t,:] += sum_d dX[(b,t,d)][prev cols] ; dfeat[b,d,:] += sum_t dX[(b,t,d)][cur cols] ; boxes likewise
__global__ void pair_concat_bwd_kernel(const float* __restrict__ dX, int B, int T, int D, int F, int nf, int kind, int ld,
                                       float* __restrict__ dprev_feat, float* __restrict__ dfeat,
                                       float* __restrict__ dprev_tab, float* __restrict__ ddet_tab) {
    const int which = blockIdx.y;  // 0: previous-side rows (b,t), 1: current-side rows (b,d)
    const int rows = which == 0 ? T : D;
    const int r = blockIdx.x % rows, b = blockIdx.x / rows;
    for (int c = threadIdx.x; c < ld; c += blockDim.x) {
        int src, off = 0;
        pair_col(kind, F, nf, c, src, off);
        if (src < 0 || (which == 0 && src >= 2) || (which == 1 && src < 2)) continue;
        float s = 0.0f;
        if (which == 0)
            for (int d = 0; d < D; ++d) s += dX[(((size_t)b * T + r) * D + d) * ld + c];
        else
            for (int t = 0; t < T; ++t) s += dX[(((size_t)b * T + t) * D + r) * ld + c];
        if (src == 0) dprev_feat[((size_t)b * T + r) * F + off] += s;
        else if (src == 1) dprev_tab[((size_t)b * T + r) * 8 + off] += s;
        else if (src == 2) dfeat[((size_t)b * D + r) * F + off] += s;
        else ddet_tab[((size_t)b * D + r) * 8 + off] += s;
    }
}

// ---- hand-designed residual (shasta.py:277-283), materialised, and its gradient w.r.t. the box tables -------------
__device__ __forceinline__ float hand_pair(const float* p, const float* q, int nf, float den, float& d2_out) {
    float d2 = 0.0f;
    for (int k = 0; k < nf; ++k) {
        const float df = p[k] - q[k];
        d2 += df * df;
    }
    d2_out = d2;
    const float dim = (fabsf(logf(p[3] + 1e-10f) - logf(q[3] + 1e-10f)) + fabsf(logf(p[4] + 1e-10f) - logf(q[4] + 1e-10f))) +
                      fabsf(logf(p[5] + 1e-10f) - logf(q[5] + 1e-10f));
    const float dc = cosf(p[6]) - cosf(q[6]), ds = sinf(p[6]) - sinf(q[6]);
    return (d2 / den + dim) + sqrtf(dc * dc + ds * ds);
}

// one block per (b, d): column norm, then dist[b, :, d]
__global__ __launch_bounds__(256) void hand_dist_fwd_kernel(const float* __restrict__ prev_tab, const float* __restrict__ det_tab,
                                                            int T, int D, int nf, float* __restrict__ dist, int ld,
                                                            float* __restrict__ denom) {
    __shared__ float red[256];
    const int d = blockIdx.x, b = blockIdx.y;
    const float* q = det_tab + ((size_t)b * D + d) * 8;
    float ssq = 0.0f;
    for (int t = threadIdx.x; t < T; t += 256) {
        const float* p = prev_tab + ((size_t)b * T + t) * 8;
        float d2 = 0.0f;
        for (int k = 0; k < nf; ++k) {
            const float df = p[k] - q[k];
            d2 += df * df;
        }
        ssq += d2 * d2;
    }
    red[threadIdx.x] = ssq;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    const float den = fmaxf(sqrtf(red[0]), 1e-12f);
    if (threadIdx.x == 0) denom[(size_t)b * D + d] = den;
    for (int t = threadIdx.x; t < T; t += 256) {
        float d2;
        dist[((size_t)b * T + t) * ld + d] = hand_pair(prev_tab + ((size_t)b * T + t) * 8, q, nf, den, d2);
    }
}

// gradient of dist w.r.t. box rows; one block per (b, side, row) where side 0 = previous table row t, 1 = detection row d.
// Only rows listed by the caller are evaluated (the anchor rows N, N+1: real boxes are inputs without gradient).
__global__ __launch_bounds__(256) void hand_dist_bwd_kernel(const float* __restrict__ gdist, int ldg,
                                                            const float* __restrict__ prev_tab, const float* __restrict__ det_tab,
                                                            const float* __restrict__ denom, int T, int D, int nf, int row0,
                                                            float* __restrict__ dprev_tab, float* __restrict__ ddet_tab) {
    __shared__ float red[8][256];
    const int side = blockIdx.y, b = blockIdx.z, r = row0 + blockIdx.x;
    float acc[7] = {0, 0, 0, 0, 0, 0, 0};
    const int n = side == 0 ? D : T;
    for (int o = threadIdx.x; o < n; o += 256) {
        const int t = side == 0 ? r : o, d = side == 0 ? o : r;
        const float* p = prev_tab + ((size_t)b * T + t) * 8;
        const float* q = det_tab + ((size_t)b * D + d) * 8;
        const float den = denom[(size_t)b * D + d];
        const float g = gdist[((size_t)b * T + t) * ldg + d];
        // d2 term incl. the normalisation: r = d2/den, den = ||d2[:,d]|| -> dr/dd2[t] = 1/den - d2[t]*S/den^3 with
        // S = sum_t' g[t',d]*d2[t',d] is handled by the caller-provided column sums folded into g2 below
        float d2 = 0.0f;
        for (int k = 0; k < nf; ++k) {
            const float df = p[k] - q[k];
            d2 += df * df;
        }
        // gs[b,d] = sum_t g[t,d]*d2[t,d] is stored by the caller in denom[B*D + b*D + d]
        const float gs = denom[(size_t)gridDim.z * D + (size_t)b * D + d];
        const float gd2 = den > 1e-12f ? g / den - gs * d2 / (den * den * den) : g / den;
        const float sgn = side == 0 ? 1.0f : -1.0f;
        for (int k = 0; k < nf; ++k) acc[k] += sgn * gd2 * 2.0f * (p[k] - q[k]);
        for (int k = 3; k < 6; ++k) {
            const float lp = logf(p[k] + 1e-10f), lq = logf(q[k] + 1e-10f);
            const float s = lp > lq ? 1.0f : (lp < lq ? -1.0f : 0.0f);
            acc[k] += side == 0 ? g * s / (p[k] + 1e-10f) : -g * s / (q[k] + 1e-10f);
        }
        const float cp = cosf(p[6]), sp = sinf(p[6]), cq = cosf(q[6]), sq = sinf(q[6]);
        const float dc = cp - cq, ds = sp - sq, rot = sqrtf(dc * dc + ds * ds);
        if (side == 0) acc[6] += g * (dc * (-sp) + ds * cp) / rot;
        else acc[6] += g * (dc * sq - ds * cq) / rot;
    }
    for (int k = 0; k < 7; ++k) red[k][threadIdx.x] = acc[k];
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s)
            for (int k = 0; k < 7; ++k) red[k][threadIdx.x] += red[k][threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x < 7) {
        float* o = side == 0 ? dprev_tab + ((size_t)b * T + r) * 8 : ddet_tab + ((size_t)b * D + r) * 8;
        o[threadIdx.x] += red[threadIdx.x][0];
    }
}

// gs[b,d] = sum_t g[b,t,d] * d2[b,t,d]  (column sums needed by the normalisation's gradient), stored behind denom
__global__ __launch_bounds__(256) void hand_gs_kernel(const float* __restrict__ gdist, int ldg, const float* __restrict__ prev_tab,
                                                      const float* __restrict__ det_tab, int T, int D, int nf, int B,
                                                      float* __restrict__ denom) {
    __shared__ float red[256];
    const int d = blockIdx.x, b = blockIdx.y;
    const float* q = det_tab + ((size_t)b * D + d) * 8;
    float s = 0.0f;
    for (int t = threadIdx.x; t < T; t += 256) {
        const float* p = prev_tab + ((size_t)b * T + t) * 8;
        float d2 = 0.0f;
        for (int k = 0; k < nf; ++k) {
            const float df = p[k] - q[k];
            d2 += df * df;
        }
        s += gdist[((size_t)b * T + t) * ldg + d] * d2;
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) {
        if ((int)threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k];
        __syncthreads();
    }
    if (threadIdx.x == 0) denom[(size_t)B * D + (size_t)b * D + d] = red[0];
}

// ---- combine (shasta.py:319) ------------------------------------------------------------------------------------------
// residual = alpha*fused + beta*dist + omega*shape ; coeff (P, ldc>=3), fused (P), shape (P), dist/residual (B,T,ld)
__global__ void combine_fwd_kernel(const float* __restrict__ coeff, int ldc, const float* __restrict__ fused, int ldf,
                                   const float* __restrict__ shape, int lds_, const float* __restrict__ dist, int T, int D, int ld,
                                   long P, float* __restrict__ residual) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    const int d = i % D;
    const long bt = i / D;
    const float a = coeff[i * ldc], be = coeff[i * ldc + 1], om = coeff[i * ldc + 2];
    residual[bt * ld + d] = (a * fused[i * ldf] + be * dist[bt * ld + d]) + om * shape[i * lds_];
}

__global__ void combine_bwd_kernel(const float* __restrict__ gres, const float* __restrict__ coeff, int ldc,
                                   const float* __restrict__ fused, int ldf, const float* __restrict__ shape, int lds_,
                                   const float* __restrict__ dist, int D, int ld, long P, float* __restrict__ gcoeff,
                                   float* __restrict__ gfused, float* __restrict__ gshape, float* __restrict__ gdist) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    const int d = i % D;
    const long bt = i / D;
    const float g = gres[bt * ld + d];
    const float a = coeff[i * ldc], be = coeff[i * ldc + 1], om = coeff[i * ldc + 2];
    gcoeff[i * ldc] = g * fused[i * ldf];
    gcoeff[i * ldc + 1] = g * dist[bt * ld + d];
    gcoeff[i * ldc + 2] = g * shape[i * lds_];
    for (int c = 3; c < ldc; ++c) gcoeff[i * ldc + c] = 0.0f;
    gfused[i * ldf] = g * a;
    for (int c = 1; c < ldf; ++c) gfused[i * ldf + c] = 0.0f;
    gshape[i * lds_] = g * om;
    for (int c = 1; c < lds_; ++c) gshape[i * lds_ + c] = 0.0f;
    gdist[bt * ld + d] = g * be;
}

// ---- softmax backward (shasta.py:324-325): gmatched = rows-part + cols-part ------------------------------------------
// rows t < N: gz = m1 * (g1 - sum_d g1*m1) over the D entries ; cols d < N: gz = m2 * (g2 - sum_t g2*m2) over the T entries
__global__ __launch_bounds__(256) void softmax_bwd_rows_kernel(const float* __restrict__ m1, const float* __restrict__ g1, int B,
                                                               int N, int T, int D, int ld, float* __restrict__ gm) {
    const int lane = threadIdx.x & 63;
    const int item = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (item >= B * T) return;
    const int b = item / T, t = item % T;
    float* o = gm + ((size_t)b * T + t) * ld;
    if (t >= N) {
        for (int d = lane; d < D; d += 64) o[d] = 0.0f;
        return;
    }
    const float* m = m1 + ((size_t)b * N + t) * D;
    const float* g = g1 + ((size_t)b * N + t) * D;
    float s = 0.0f;
    for (int d = lane; d < D; d += 64) s += g[d] * m[d];
    s = wave_sum(s);
    for (int d = lane; d < D; d += 64) o[d] = m[d] * (g[d] - s);
}

__global__ __launch_bounds__(256) void softmax_bwd_cols_kernel(const float* __restrict__ m2, const float* __restrict__ g2, int N,
                                                               int T, int ld, float* __restrict__ gm) {
    __shared__ float red[4][64];
    const int b = blockIdx.y, dl = threadIdx.x & 63, tq = threadIdx.x >> 6;
    const int d = blockIdx.x * 64 + dl;
    const int dc = min(d, N - 1);
    const float* m = m2 + (size_t)b * T * N + dc;
    const float* g = g2 + (size_t)b * T * N + dc;
    float s = 0.0f;
    for (int t = tq; t < T; t += 4) s += g[(size_t)t * N] * m[(size_t)t * N];
    red[tq][dl] = s;
    __syncthreads();
    s = (red[0][dl] + red[1][dl]) + (red[2][dl] + red[3][dl]);
    if (d >= N) return;
    for (int t = tq; t < T; t += 4) gm[((size_t)b * T + t) * ld + d] += m[(size_t)t * N] * (g[(size_t)t * N] - s);
}

// ---- small helpers ------------------------------------------------------------------------------------------------------
// out[n] = sum_m Y[m][n] (bias gradient), one block per 64 columns, fixed order
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ Y, int ldy, int M, int N, float* __restrict__ out) {
    __shared__ float red[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), q = threadIdx.x >> 6;
    float s = 0.0f;
    if (c < N)
        for (int m = q; m < M; m += 4) s += Y[(size_t)m * ldy + c];
    red[q][threadIdx.x & 63] = s;
    __syncthreads();
    if (q == 0 && c < N) out[c] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// y = |x| (mode 0) ; g_out = g * sign(x) (mode 1) over `cols` columns of each row, columns [c0, c1) only (others copied / passed)
__global__ void abs_kernel(const float* __restrict__ x, const float* __restrict__ g, float* __restrict__ out, long n, int cols,
                           int c0, int c1, int mode) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int c = i % cols;
    const bool in = c >= c0 && c < c1;
    if (mode == 0) out[i] = in ? fabsf(x[i]) : x[i];
    else out[i] = in ? (x[i] > 0.0f ? g[i] : (x[i] < 0.0f ? -g[i] : 0.0f)) : g[i];
}

// gather backward: dBEV[b, y, x, :] += w * dfeat[b, n, pt*C : (pt+1)*C] for the four corners of every point (atomics)
__global__ __launch_bounds__(256) void bev_gather_bwd_kernel(const float* __restrict__ dfeat, int H, int W, int C,
                                                             const float* __restrict__ boxes, int N, int box_stride,
                                                             int box_batch_stride, int num_point, float pc_x0, float pc_y0,
                                                             float vs_x, float vs_y, float out_stride_px, int row_stride,
                                                             int batch_stride, int total_points, float* __restrict__ dbev) {
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    if (wave >= total_points) return;
    const int pt = wave % num_point, n = (wave / num_point) % N, b = wave / (num_point * N);
    const float* box = boxes + (size_t)b * box_batch_stride + (size_t)n * box_stride;
    const float cx = box[0], cy = box[1];
    float px = cx, py = cy;
    const int edge = (num_point == 5) ? pt - 1 : (num_point == 4 ? pt : -1);
    if (edge >= 0) {
        const float w = box[3], l = box[4], yaw = box[6];
        const float s = sinf(yaw), c = cosf(yaw);
        const int ia = (edge == 0) ? 0 : (edge == 1) ? 2 : (edge == 2) ? 0 : 1;
        const int ib = (edge == 0) ? 1 : (edge == 1) ? 3 : (edge == 2) ? 3 : 2;
        float qx[2], qy[2];
        for (int k = 0; k < 2; ++k) {
            const int ci = k ? ib : ia;
            const float ux = (ci < 2) ? -0.5f : 0.5f, uy = (ci == 1 || ci == 2) ? 0.5f : -0.5f;
            const float dx = __fmul_rn(w, ux), dy = __fmul_rn(l, uy);
            qx[k] = __fadd_rn(__fadd_rn(__fmul_rn(dx, c), __fmul_rn(dy, s)), cx);
            qy[k] = __fadd_rn(__fadd_rn(__fmul_rn(-dx, s), __fmul_rn(dy, c)), cy);
        }
        px = __fdiv_rn(__fadd_rn(qx[0], qx[1]), 2.0f);
        py = __fdiv_rn(__fadd_rn(qy[0], qy[1]), 2.0f);
    }
    const float x = __fdiv_rn(__fdiv_rn(__fsub_rn(px, pc_x0), vs_x), out_stride_px);
    const float y = __fdiv_rn(__fdiv_rn(__fsub_rn(py, pc_y0), vs_y), out_stride_px);
    auto clampi = [](float f, int hi) -> int {
        if (!(f > -2.0f)) return -1;
        if (f > (float)(hi + 1)) return hi + 1;
        return (int)f;
    };
    int x0 = clampi(floorf(x), W), y0 = clampi(floorf(y), H);
    int x1 = x0 + 1, y1 = y0 + 1;
    x0 = min(max(x0, 0), W - 1); x1 = min(max(x1, 0), W - 1);
    y0 = min(max(y0, 0), H - 1); y1 = min(max(y1, 0), H - 1);
    const float wa = (x1 - x) * (y1 - y), wb = (x1 - x) * (y - y0), wc = (x - x0) * (y1 - y), wd = (x - x0) * (y - y0);
    float* im = dbev + (size_t)b * H * W * C;
    const float* g = dfeat + (size_t)b * batch_stride + (size_t)n * row_stride + (size_t)pt * C;
    for (int ch = lane; ch < C; ch += 64) {
        const float v = g[ch];
        atomicAdd(im + ((size_t)y0 * W + x0) * C + ch, v * wa);
        atomicAdd(im + ((size_t)y1 * W + x0) * C + ch, v * wb);
        atomicAdd(im + ((size_t)y0 * W + x1) * C + ch, v * wc);
        atomicAdd(im + ((size_t)y1 * W + x1) * C + ch, v * wd);
    }
}

}  // namespace shasta

using namespace shasta;

extern "C" int shasta_pair_concat_f32(const float* prev_feat, const float* feat, const float* prev_tab, const float* det_tab, int B,
                                      int T, int D, int F, int nf, int kind, int ld, float* X, shasta_stream_t stream) {
    SHASTA_REQUIRE(prev_feat && feat && prev_tab && det_tab && X, "pair_concat: null pointer");
    SHASTA_REQUIRE(kind >= 0 && kind <= 2 && nf >= 1 && nf <= 7, "pair_concat: bad kind / nf");
    const long rows = (long)B * T * D;
    SHASTA_REQUIRE(rows < (1L << 31), "pair_concat: too many pairs");
    if (rows == 0) return SHASTA_OK;
    hipLaunchKernelGGL(pair_concat_kernel, dim3((unsigned)rows), dim3(128), 0, as_stream(stream), prev_feat, feat, prev_tab, det_tab, B,
                       T, D, F, nf, kind, ld, X);
    return check_launch("pair_concat");
}

extern "C" int shasta_pair_concat_bwd_f32(const float* dX, int B, int T, int D, int F, int nf, int kind, int ld, float* dprev_feat,
                                          float* dfeat, float* dprev_tab, float* ddet_tab, shasta_stream_t stream) {
    SHASTA_REQUIRE(dX && dprev_feat && dfeat && dprev_tab && ddet_tab, "pair_concat_bwd: null pointer");
    SHASTA_REQUIRE(T == D, "pair_concat_bwd: T must equal D");
    if (B == 0) return SHASTA_OK;
    hipLaunchKernelGGL(pair_concat_bwd_kernel, dim3(B * T, 2), dim3(128), 0, as_stream(stream), dX, B, T, D, F, nf, kind, ld,
                       dprev_feat, dfeat, dprev_tab, ddet_tab);
    return check_launch("pair_concat_bwd");
}

extern "C" int shasta_hand_dist_f32(const float* prev_tab, const float* det_tab, int B, int T, int D, int nf, float* dist, int ld,
                                    float* denom, shasta_stream_t stream) {
    SHASTA_REQUIRE(prev_tab && det_tab && dist && denom, "hand_dist: null pointer");
    if (B == 0) return SHASTA_OK;
    hipLaunchKernelGGL(hand_dist_fwd_kernel, dim3(D, B), dim3(256), 0, as_stream(stream), prev_tab, det_tab, T, D, nf, dist, ld, denom);
    return check_launch("hand_dist_fwd");
}

// denom must have room for 2*B*D floats (norms, then the column sums written here)
extern "C" int shasta_hand_dist_bwd_f32(const float* gdist, int ldg, const float* prev_tab, const float* det_tab, float* denom, int B,
                                        int T, int D, int nf, int row0, int nrows, float* dprev_tab, float* ddet_tab,
                                        shasta_stream_t stream) {
    SHASTA_REQUIRE(gdist && prev_tab && det_tab && denom && dprev_tab && ddet_tab, "hand_dist_bwd: null pointer");
    if (B == 0 || nrows == 0) return SHASTA_OK;
    hipLaunchKernelGGL(hand_gs_kernel, dim3(D, B), dim3(256), 0, as_stream(stream), gdist, ldg, prev_tab, det_tab, T, D, nf, B, denom);
    int rc = check_launch("hand_gs");
    if (rc) return rc;
    hipLaunchKernelGGL(hand_dist_bwd_kernel, dim3(nrows, 2, B), dim3(256), 0, as_stream(stream), gdist, ldg, prev_tab, det_tab, denom,
                       T, D, nf, row0, dprev_tab, ddet_tab);
    return check_launch("hand_dist_bwd");
}

extern "C" int shasta_combine_f32(const float* coeff, int ldc, const float* fused, int ldf, const float* shape, int lds_, const float* dist,
                                  int B, int T, int D, int ld, float* residual, shasta_stream_t stream) {
    SHASTA_REQUIRE(coeff && fused && shape && dist && residual && ldc >= 3, "combine: bad argument");
    const long P = (long)B * T * D;
    if (P == 0) return SHASTA_OK;
    hipLaunchKernelGGL(combine_fwd_kernel, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, as_stream(stream), coeff, ldc, fused, ldf,
                       shape, lds_, dist, T, D, ld, P, residual);
    return check_launch("combine_fwd");
}

extern "C" int shasta_combine_bwd_f32(const float* gres, const float* coeff, int ldc, const float* fused, int ldf, const float* shape,
                                      int lds_, const float* dist, int B, int T, int D, int ld, float* gcoeff, float* gfused,
                                      float* gshape, float* gdist, shasta_stream_t stream) {
    SHASTA_REQUIRE(gres && coeff && fused && shape && dist && gcoeff && gfused && gshape && gdist && ldc >= 3, "combine_bwd: bad argument");
    const long P = (long)B * T * D;
    if (P == 0) return SHASTA_OK;
    hipLaunchKernelGGL(combine_bwd_kernel, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, as_stream(stream), gres, coeff, ldc, fused,
                       ldf, shape, lds_, dist, D, ld, P, gcoeff, gfused, gshape, gdist);
    return check_launch("combine_bwd");
}

extern "C" int shasta_softmax_bwd_f32(const float* m1, const float* g1, const float* m2, const float* g2, int B, int N, float* gmatched,
                                      int ld, shasta_stream_t stream) {
    SHASTA_REQUIRE(m1 && g1 && m2 && g2 && gmatched && ld >= N + 2, "softmax_bwd: bad argument");
    if (B == 0) return SHASTA_OK;
    const int T = N + 2, D = N + 2;
    hipLaunchKernelGGL(softmax_bwd_rows_kernel, dim3(cdiv(B * T, 4)), dim3(256), 0, as_stream(stream), m1, g1, B, N, T, D, ld, gmatched);
    int rc = check_launch("softmax_bwd_rows");
    if (rc) return rc;
    hipLaunchKernelGGL(softmax_bwd_cols_kernel, dim3(cdiv(N, 64), B), dim3(256), 0, as_stream(stream), m2, g2, N, T, ld, gmatched);
    return check_launch("softmax_bwd_cols");
}

extern "C" int shasta_colsum_f32(const float* Y, int ldy, int M, int N, float* out, shasta_stream_t stream) {
    SHASTA_REQUIRE(Y && out && M >= 0 && N >= 0, "colsum: bad argument");
    if (N == 0) return SHASTA_OK;
    hipLaunchKernelGGL(colsum_kernel, dim3(cdiv(N, 64)), dim3(256), 0, as_stream(stream), Y, ldy, M, N, out);
    return check_launch("colsum");
}

extern "C" int shasta_abs_f32(const float* x, const float* g, float* out, long n, int cols, int c0, int c1, int backward,
                              shasta_stream_t stream) {
    SHASTA_REQUIRE(x && out && cols > 0 && (!backward || g), "abs: bad argument");
    if (n == 0) return SHASTA_OK;
    hipLaunchKernelGGL(abs_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, as_stream(stream), x, g, out, n, cols, c0, c1,
                       backward ? 1 : 0);
    return check_launch("abs");
}

extern "C" int shasta_bev_gather_bwd_f32(const float* dfeat, int B, int H, int W, int C, const float* boxes, int N, int box_stride,
                                         int box_batch_stride, int num_point, float pc_x0, float pc_y0, float vs_x, float vs_y,
                                         float out_stride, int row_stride, int batch_stride, float* dbev, shasta_stream_t stream) {
    SHASTA_REQUIRE(dfeat && boxes && dbev, "bev_gather_bwd: null pointer");
    SHASTA_REQUIRE(num_point == 1 || num_point == 4 || num_point == 5, "bev_gather_bwd: num_point must be 1, 4 or 5");
    const long total = (long)B * N * num_point;
    if (total == 0) return SHASTA_OK;
    hipLaunchKernelGGL(bev_gather_bwd_kernel, dim3(cdiv((int)total, 4)), dim3(256), 0, as_stream(stream), dfeat, H, W, C, boxes, N,
                       box_stride, box_batch_stride, num_point, pc_x0, pc_y0, vs_x, vs_y, out_stride, row_stride, batch_stride,
                       (int)total, dbev);
    return check_launch("bev_gather_bwd");
}
