// Training path (SURVEY.md 8(a) row 19 / BASELINE config 5): auxiliary kernels of the backward pass.
// The dense layers run on the strided matrix-core GEMM (gemm_f32.hip: Y = X W^T, dX = dY W, dW = dY^T X); this file holds
// what is not a GEMM: the factorised first layer of the pair MLPs and its transpose (the cat/expand of
// det3d/models/tracker/shasta.py:286-313 and their autograd), the hand-designed residual and its gradient (:277-283), the
// combine's gradient (:319), the two softmax backward passes (:324-325), bias-gradient column sums, |x| backward and the
// gather's scatter-add.  The (B, T*D, 2F) pair tensor of the reference is not materialised; the narrow hidden activations
// of each pair are.  All reductions have a fixed order except the scatter-add into the BEV gradient, which uses float
// atomics (gradient of an input only).
#include <algorithm>
#include <math.h>

#include "common.hpp"

namespace shasta {

// ---- factorised first layer of the pair MLPs ---------------------------------------------------------------------------
// W0 . [prev_t ; cur_d] + b0 = UP[t] + UC[d] (pair_layout.hpp): the hidden activations of every pair are
//   H[(b,t,d)][j] = relu(UP[(b,t)][j] + UC[(b,d)][j]),
// and the gradient of the pre-activation Z folds back onto the table rows by two fixed-order sums,
//   gUP[(b,t)][j] = sum_d gZ[(b,t,d)][j],   gUC[(b,d)][j] = sum_t gZ[(b,t,d)][j],
// after which dW0, db0 and the table gradients are small GEMMs over B*T rows instead of B*T*D pairs.
__global__ __launch_bounds__(256) void pair_hidden_kernel(const float* __restrict__ UP, int ldp, const float* __restrict__ UC, int ldc,
                                                          int T, int D, int E, long total, float* __restrict__ H) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;  // element (b,t,d,j)
    if (i >= total) return;
    const int j = i % E;
    const long pr = i / E;
    const int d = pr % D;
    const long bt = pr / D;
    const long b = bt / T;
    H[i] = fmaxf(UP[bt * ldp + j] + UC[(b * D + d) * ldc + j], 0.0f);
}

// block = (row r of side `which`, batch b); threads = E columns x G row groups
__global__ __launch_bounds__(256) void pair_reduce_kernel(const float* __restrict__ gZ, int T, int D, int E, int EP,
                                                          float* __restrict__ gUP, float* __restrict__ gUC) {
    __shared__ float red[256];
    const int which = blockIdx.y, b = blockIdx.z, r = blockIdx.x;
    const int j = threadIdx.x % EP, q = threadIdx.x / EP, G = 256 / EP;
    const int n = which == 0 ? D : T;
    float s = 0.0f;
    if (j < E) {
        if (which == 0)
            for (int d = q; d < n; d += G) s += gZ[(((size_t)b * T + r) * D + d) * E + j];
        else
            for (int t = q; t < n; t += G) s += gZ[(((size_t)b * T + t) * D + r) * E + j];
    }
    red[threadIdx.x] = s;
    __syncthreads();
    if (q == 0 && j < E) {
        float t = 0.0f;
        for (int k = 0; k < G; ++k) t += red[k * EP + j];
        (which == 0 ? gUP : gUC)[((size_t)b * (which == 0 ? T : D) + r) * E + j] = t;
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

// ---- the training loss (tools/nusc_shasta/train.py:200-211) and its gradient ------------------------------------------------
//   loss = (sum(gt1 . -log(m1 + 1e-10)) / sum(gt1) + sum(gt2 . -log(m2 + 1e-10)) / sum(gt2)) / 2,  gt1 = gt[:, :N, :], gt2 = gt[:, :, :N]
//   (each quotient only where its sum(gt) > 0: train.py:208-209)
// one block per row (b, t) of gt: the four sums of the row in a fixed order; loss_finish_kernel adds the rows in order.
__global__ __launch_bounds__(256) void loss_rows_kernel(const float* __restrict__ m1, const float* __restrict__ m2, const float* __restrict__ gt,
                                                        int N, float* __restrict__ part) {
    __shared__ float red[4][256];
    const int T = N + 2, t = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const float* g = gt + ((size_t)b * T + t) * T;
    float s1 = 0.0f, c1 = 0.0f, s2 = 0.0f, c2 = 0.0f;
    for (int d = tid; d < T; d += 256) {
        const float w = g[d];
        if (t < N) {
            s1 = fmaf(w, -logf(m1[((size_t)b * N + t) * T + d] + 1e-10f), s1);
            c1 += w;
        }
        if (d < N) {
            s2 = fmaf(w, -logf(m2[((size_t)b * T + t) * N + d] + 1e-10f), s2);
            c2 += w;
        }
    }
    red[0][tid] = s1; red[1][tid] = c1; red[2][tid] = s2; red[3][tid] = c2;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if (tid < off)
#pragma unroll
            for (int k = 0; k < 4; ++k) red[k][tid] += red[k][tid + off];
        __syncthreads();
    }
    if (tid < 4) part[((size_t)b * T + t) * 4 + tid] = red[tid][0];
}

// sums[0..3] = (s1, c1, s2, c2) over all rows, sums[4] = the loss
__global__ __launch_bounds__(256) void loss_finish_kernel(const float* __restrict__ part, int rows, float* __restrict__ sums) {
    __shared__ float red[4][64];
    const int k = threadIdx.x & 3, q = threadIdx.x >> 2;
    float s = 0.0f;
#pragma unroll 8
    for (int r = q; r < rows; r += 64) s += part[(size_t)r * 4 + k];
    red[k][q] = s;
    __syncthreads();
    if (threadIdx.x < 4) {
        float t = 0.0f;
        for (int j = 0; j < 64; ++j) t += red[threadIdx.x][j];
        sums[threadIdx.x] = t;
    }
    __syncthreads();
    // (train.py:208-209: a direction without a single ground-truth entry contributes its plain sum - zero - not 0 / 0)
    if (threadIdx.x == 0) sums[4] = ((sums[1] > 0.0f ? sums[0] / sums[1] : sums[0]) + (sums[3] > 0.0f ? sums[2] / sums[3] : sums[2])) * 0.5f;
}

// g1 = dloss/dm1 = -gt1 / (m1 + 1e-10) * gloss / (2 sum(gt1)), g2 likewise; gloss read from device memory
__global__ __launch_bounds__(256) void loss_bwd_kernel(const float* __restrict__ m1, const float* __restrict__ m2, const float* __restrict__ gt,
                                                       const float* __restrict__ sums, const float* __restrict__ gloss, int N, long n1,
                                                       float* __restrict__ g1, float* __restrict__ g2) {
    const int T = N + 2;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n1) return;
    const float gl = gloss[0] * 0.5f;
    {  // element i of m1 (B, N, T)
        const int d = i % T;
        const long bt = i / T;
        const long b = bt / N;
        const int t = bt - b * N;
        g1[i] = -gt[((size_t)b * T + t) * T + d] / (m1[i] + 1e-10f) * (sums[1] > 0.0f ? gl / sums[1] : gl);
    }
    {  // element i of m2 (B, T, N): the same count
        const int d = i % N;
        const long bt = i / N;
        g2[i] = -gt[(size_t)bt * T + d] / (m2[i] + 1e-10f) * (sums[3] > 0.0f ? gl / sums[3] : gl);
    }
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
    // eight rows' loads in flight, the products added in the rows' order (the same sum as a row at a time: a loop of one dependent
    // load pair per step was latency bound - 77 us at N = 500 x 8 frame pairs on 64 workgroups)
    for (int t0 = tq; t0 < T; t0 += 32) {
        float tg[8], tm[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const size_t t = (size_t)min(t0 + 4 * u, T - 1);
            tg[u] = g[t * N];
            tm[u] = m[t * N];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (t0 + 4 * u < T) s += tg[u] * tm[u];
    }
    red[tq][dl] = s;
    __syncthreads();
    s = (red[0][dl] + red[1][dl]) + (red[2][dl] + red[3][dl]);
    if (d >= N) return;
    for (int t0 = tq; t0 < T; t0 += 16) {
        float tg[4], tm[4], to[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const size_t t = (size_t)min(t0 + 4 * u, T - 1);
            tg[u] = g[t * N];
            tm[u] = m[t * N];
            to[u] = gm[((size_t)b * T + t) * ld + d];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (t0 + 4 * u < T) gm[((size_t)b * T + t0 + 4 * u) * ld + d] = to[u] + tm[u] * (tg[u] - s);
    }
}

// ---- small helpers ------------------------------------------------------------------------------------------------------
// out[n] = sum_m Y[m][n] (bias gradient): block (x: 64 columns, y: row chunk) sums its rows in a fixed order into
// part[chunk][n]; a second launch adds the chunks in order (single chunk: written directly).  Deterministic.
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ Y, int ldy, int M, int N, int rows_per_chunk,
                                                     float* __restrict__ out) {
    __shared__ float red[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), q = threadIdx.x >> 6;
    const int m0 = blockIdx.y * rows_per_chunk, m1 = min(M, m0 + rows_per_chunk);
    float s0 = 0.0f, s1 = 0.0f;
    if (c < N) {
        int m = m0 + q;
        for (; m + 4 < m1; m += 8) {
            s0 += Y[(size_t)m * ldy + c];
            s1 += Y[(size_t)(m + 4) * ldy + c];
        }
        if (m < m1) s0 += Y[(size_t)m * ldy + c];
    }
    red[q][threadIdx.x & 63] = s0 + s1;
    __syncthreads();
    if (q == 0 && c < N)
        out[(size_t)blockIdx.y * N + c] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// narrow matrices (N < 64): one wave per row chunk would idle most lanes, so lanes split the rows instead: lane l of a
// 256-thread block owns column l % NP and every (256/NP)-th row of the chunk
__global__ __launch_bounds__(256) void colsum_narrow_kernel(const float* __restrict__ Y, int ldy, int M, int N, int NP,
                                                            int rows_per_chunk, float* __restrict__ out) {
    __shared__ float red[256];
    const int c = threadIdx.x % NP, q = threadIdx.x / NP, nq = 256 / NP;
    const int m0 = blockIdx.y * rows_per_chunk, m1 = min(M, m0 + rows_per_chunk);
    float s = 0.0f;
    if (c < N)
        for (int m = m0 + q; m < m1; m += nq) s += Y[(size_t)m * ldy + c];
    red[threadIdx.x] = s;
    __syncthreads();
    if (q == 0 && c < N) {
        float t = 0.0f;
        for (int k = 0; k < nq; ++k) t += red[k * NP + c];
        out[(size_t)blockIdx.y * N + c] = t;
    }
}

// one wave per column: lanes stride over the chunks, then a fixed-order wave reduction
__global__ __launch_bounds__(256) void colsum_finish_kernel(const float* __restrict__ part, int chunks, int N, float* __restrict__ out) {
    const int c = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (c >= N) return;
    float s = 0.0f;
    for (int k = lane; k < chunks; k += 64) s += part[(size_t)k * N + c];
    s = wave_sum(s);
    if (lane == 0) out[c] = s;
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

// Adam step (torch.optim.Adam semantics as used by tools/nusc_shasta/train.py:147: L2 weight decay added to the gradient,
// no amsgrad), one pass over p, g, m, v instead of the seven multi-tensor passes of the unfused optimizer:
//   g' = g + wd*p ; m += (g' - m)*(1-b1) ; v = b2*v + (1-b2)*g'*g' ; p -= (lr/bc1) * m / (sqrt(v)/sqrt(bc2) + eps)
struct AdamArgs {
    float lr_over_bc1, beta1, beta2, eps, weight_decay, rsqrt_bc2;
    // non-null: everything that changes from step to step comes from DEVICE memory (dyn = {lr / bc1, 1 / sqrt(bc2), beta1, beta2},
    // written by adam_prepare_kernel from a device-side step counter and the scheduler's lr / betas) - a training step replayed from a
    // captured hipGraph cannot take them as launch arguments, which are frozen at capture time
    const float* dyn;
};
__device__ __forceinline__ AdamArgs adam_resolve(AdamArgs a) {
    if (a.dyn) {
        a.lr_over_bc1 = a.dyn[0];
        a.rsqrt_bc2 = a.dyn[1];
        a.beta1 = a.dyn[2];
        a.beta2 = a.dyn[3];
    }
    return a;
}
// one step of the device-side schedule: ++*step; hyper = {lr, beta1, beta2} -> dyn = {lr / (1 - beta1^step), 1 / sqrt(1 - beta2^step), beta1, beta2}
__global__ void adam_prepare_kernel(int* __restrict__ step, const float* __restrict__ hyper, float* __restrict__ dyn) {
    if (threadIdx.x || blockIdx.x) return;
    const int s = *step + 1;
    *step = s;
    const float beta1 = hyper[1], beta2 = hyper[2];
    const double bc1 = 1.0 - pow((double)beta1, (double)s), bc2 = 1.0 - pow((double)beta2, (double)s);
    dyn[0] = (float)((double)hyper[0] / bc1);
    dyn[1] = (float)(1.0 / sqrt(bc2));
    dyn[2] = beta1;
    dyn[3] = beta2;
}

__device__ __forceinline__ void adam_one(float& p, float g, float& m, float& v, const AdamArgs& a) {
    g = fmaf(a.weight_decay, p, g);
    m = fmaf(g - m, 1.0f - a.beta1, m);
    v = fmaf(1.0f - a.beta2, g * g, a.beta2 * v);
    const float denom = fmaf(sqrtf(v), a.rsqrt_bc2, a.eps);
    p = p - a.lr_over_bc1 * (m / denom);
}

__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, long n, AdamArgs a_in) {
    const AdamArgs a = adam_resolve(a_in);
    const long n4 = n >> 2;
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        f32x4 pp = reinterpret_cast<f32x4*>(p)[i];
        const f32x4 gg = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(g) + i);
        f32x4 mm = reinterpret_cast<f32x4*>(m)[i], vv = reinterpret_cast<f32x4*>(v)[i];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float pk = pp[k], mk = mm[k], vk = vv[k];
            adam_one(pk, gg[k], mk, vk, a);
            pp[k] = pk;
            mm[k] = mk;
            vv[k] = vk;
        }
        reinterpret_cast<f32x4*>(p)[i] = pp;
        reinterpret_cast<f32x4*>(m)[i] = mm;
        reinterpret_cast<f32x4*>(v)[i] = vv;
    }
    const long i = (n4 << 2) + (long)blockIdx.x * blockDim.x + threadIdx.x;  // tail
    if (i < n) adam_one(p[i], g[i], m[i], v[i], a);
}

// The same update for up to ADAM_MULTI tensors in one launch (the ~60 small weight / bias tensors of rows 6-16: one launch instead of
// sixty of 5 us each): the tensor table travels in the kernel arguments; blockIdx.y = tensor, a grid-stride loop over its elements.
constexpr int ADAM_MULTI = 48;
struct AdamMulti {
    float* p[ADAM_MULTI];
    const float* g[ADAM_MULTI];
    float* m[ADAM_MULTI];
    float* v[ADAM_MULTI];
    long n[ADAM_MULTI];
};

__global__ __launch_bounds__(256) void adam_multi_kernel(AdamMulti t, AdamArgs a_in) {
    const AdamArgs a = adam_resolve(a_in);
    const int k = blockIdx.y;
    float* __restrict__ p = t.p[k];
    const float* __restrict__ g = t.g[k];
    float* __restrict__ m = t.m[k];
    float* __restrict__ v = t.v[k];
    const long n = t.n[k];
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) adam_one(p[i], g[i], m[i], v[i], a);
}

// Adam for a matrix whose gradient is a rank-R product, g[h][k] = sum_r G[r][h] X[r][k] (the first aug_shape layers: G = gradient of the
// hidden activations, X = the layer's inputs, R = frame-pairs of the step over all ranks): the gradient is formed in registers inside the
// Adam pass - it is never written to memory and never read back.  24 bytes per parameter (p, m, v in and out) instead of the 36 of
// "write the gradient, read it in the Adam kernel".  A thread owns COLS consecutive columns and keeps its R x COLS slice of X in
// registers; G[r][h] is uniform over the workgroup (scalar loads).
// RDX > 0: the same pass also forms Y = Gdx . W with the weights as they are BEFORE the update (Gdx: (Rdx, H) at ldgdx; the backward's
// dx = ghid . W1, which otherwise reads the matrix a second time: smallm_nn_kernel) - partial sums per row block into `part`
// ([block][r][k], summed by smallm_finish_kernel).
// (196 - 229 registers, two waves per SIMD; forced to three or four waves the pass is 5 - 10 % slower: 1.10 -> 1.18 / 1.22 ms per 1 GB matrix)
template <int RMAX, int COLS, int RDX = 0>
__global__ __launch_bounds__(256) void adam_lowrank_kernel(float* __restrict__ p, float* __restrict__ m, float* __restrict__ v,
                                                           const float* __restrict__ G, int ldg, const float* __restrict__ X, int ldx, int R,
                                                           int H, int K, int rows_per_block, AdamArgs a_in, const float* __restrict__ Gdx = nullptr,
                                                           int ldgdx = 0, int Rdx = 0, float* __restrict__ part = nullptr) {
    const AdamArgs a = adam_resolve(a_in);
    typedef float fv __attribute__((ext_vector_type(COLS)));
    const long k0 = ((long)blockIdx.x * 256 + threadIdx.x) * COLS;
    fv dacc[RDX > 0 ? RDX : 1];
#pragma unroll
    for (int r = 0; r < (RDX > 0 ? RDX : 1); ++r) dacc[r] = fv(0.0f);
    // RMAX > 16 (the factors of a step over several ranks): G[r][h] for one h and all r lies in R different cache lines - as scalar loads
    // they bound the pass (2.38 ms per 1 GB matrix at R = 64, 2.6 TB/s).  The block's (R, rows) slice of G goes through LDS instead,
    // transposed to [row][r], and is read back as broadcasts.
    constexpr bool G_LDS = RMAX > 16;
    __shared__ __attribute__((aligned(16))) float s_g[G_LDS ? 64 * RMAX : 4];
    fv xv[RMAX];
#pragma unroll
    for (int r = 0; r < RMAX; ++r) {
        xv[r] = fv(0.0f);
        if (r < R && k0 < K) xv[r] = *reinterpret_cast<const fv*>(X + (size_t)r * ldx + k0);
    }
    const int h0 = blockIdx.y * rows_per_block, h1 = min(H, h0 + rows_per_block);
    if (G_LDS) {  // (rows_per_block <= 64; every thread of the block gets here: threads past the last column leave after the barrier)
        const int rows = h1 - h0;
        for (int e = threadIdx.x; e < RMAX * rows; e += 256) {
            const int r = e / rows, hl = e - r * rows;
            s_g[hl * RMAX + r] = r < R ? G[(size_t)r * ldg + h0 + hl] : 0.0f;
        }
        __syncthreads();
    }
    if (k0 >= K) return;
    for (int h = h0; h < h1; ++h) {
        const size_t at = (size_t)h * K + k0;
        // (3 GB stream through once per step: non-temporal loads and stores, 1.06 -> 1.00 ms per 1 GB matrix = 6.1 TB/s over p, m, v in
        // and out; unrolling the row loop on top of that: nothing)
        fv pp = __builtin_nontemporal_load(reinterpret_cast<const fv*>(p + at)), mm = __builtin_nontemporal_load(reinterpret_cast<const fv*>(m + at)),
           vv = __builtin_nontemporal_load(reinterpret_cast<const fv*>(v + at));
        if (RDX > 0) {
#pragma unroll
            for (int r = 0; r < RDX; ++r) {
                const float gd = r < Rdx ? Gdx[(size_t)r * ldgdx + h] : 0.0f;
#pragma unroll
                for (int c = 0; c < COLS; ++c) dacc[r][c] = fmaf(gd, pp[c], dacc[r][c]);
            }
        }
        fv g = fv(0.0f);
#pragma unroll
        for (int r = 0; r < RMAX; ++r) {
            const float gr = G_LDS ? s_g[(h - h0) * RMAX + r] : (r < R ? G[(size_t)r * ldg + h] : 0.0f);  // wave-uniform: LDS broadcast / scalar load
#pragma unroll
            for (int c = 0; c < COLS; ++c) g[c] = fmaf(gr, xv[r][c], g[c]);
        }
#pragma unroll
        for (int c = 0; c < COLS; ++c) {
            float pk = pp[c], mk = mm[c], vk = vv[c];
            adam_one(pk, g[c], mk, vk, a);
            pp[c] = pk;
            mm[c] = mk;
            vv[c] = vk;
        }
        __builtin_nontemporal_store(pp, reinterpret_cast<fv*>(p + at));
        __builtin_nontemporal_store(mm, reinterpret_cast<fv*>(m + at));
        __builtin_nontemporal_store(vv, reinterpret_cast<fv*>(v + at));
    }
    if (RDX > 0) {
        for (int r = 0; r < Rdx; ++r) *reinterpret_cast<fv*>(part + ((size_t)blockIdx.y * Rdx + r) * K + k0) = dacc[r];
    }
}

__global__ void scale_kernel(float* __restrict__ x, long n, float alpha) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) x[i] *= alpha;
}

// ---- the two big streams of the anchor backward at small batch ----------------------------------------------------------
// dW1 = ghid^T x is a rank-R update of an (H, K) matrix (1 GB at N = 500) and dx = ghid W1 reads that matrix once; with R =
// (world x) batch <= 16 both are pure HBM streams with R FMAs per element, which a 64 x 64-tile GEMM with a K = R
// reduction serves at only ~2.7 TB/s.  One thread owns 4 consecutive columns (one 16-byte access per row).
template <int RMAX>
__global__ __launch_bounds__(256) void lowrank_outer_kernel(const float* __restrict__ G, int ldg, const float* __restrict__ X, int ldx,
                                                            int R, int H, int K, int rows_per_block, float* __restrict__ dW) {
    const long k4 = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (k4 >= K) return;
    f32x4 xv[RMAX];
#pragma unroll
    for (int r = 0; r < RMAX; ++r) xv[r] = r < R ? *reinterpret_cast<const f32x4*>(X + (size_t)r * ldx + k4) : f32x4{0, 0, 0, 0};
    const int h0 = blockIdx.y * rows_per_block, h1 = min(H, h0 + rows_per_block);
    for (int h = h0; h < h1; ++h) {
        f32x4 o = {0, 0, 0, 0};
#pragma unroll
        for (int r = 0; r < RMAX; ++r) {
            const float g = r < R ? G[(size_t)r * ldg + h] : 0.0f;  // wave-uniform: scalar load
#pragma unroll
            for (int c = 0; c < 4; ++c) o[c] = fmaf(g, xv[r][c], o[c]);
        }
        __builtin_nontemporal_store(o, reinterpret_cast<f32x4*>(dW + (size_t)h * K + k4));
    }
}

// part[chunk][r][k] = sum_{h in chunk} G[r][h] * W[h][k]
template <int RMAX>
__global__ __launch_bounds__(256) void smallm_nn_kernel(const float* __restrict__ G, int ldg, const float* __restrict__ W, int R, int H,
                                                        int K, int rows_per_chunk, float* __restrict__ part) {
    const long k4 = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (k4 >= K) return;
    f32x4 acc[RMAX];
#pragma unroll
    for (int r = 0; r < RMAX; ++r) acc[r] = f32x4{0, 0, 0, 0};
    const int h0 = blockIdx.y * rows_per_chunk, h1 = min(H, h0 + rows_per_chunk);
#pragma unroll 2
    for (int h = h0; h < h1; ++h) {
        const f32x4 wv = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(W + (size_t)h * K + k4));
#pragma unroll
        for (int r = 0; r < RMAX; ++r) {
            const float g = r < R ? G[(size_t)r * ldg + h] : 0.0f;
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[r][c] = fmaf(g, wv[c], acc[r][c]);
        }
    }
    for (int r = 0; r < R; ++r) *reinterpret_cast<f32x4*>(part + ((size_t)blockIdx.y * R + r) * K + k4) = acc[r];
}

__global__ __launch_bounds__(256) void smallm_finish_kernel(const float* __restrict__ part, int chunks, int R, int K, float* __restrict__ Y,
                                                            long ldy, int accumulate) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)R * K) return;
    const int r = (int)(i / K);
    const long k = i - (long)r * K;
    float s = 0.0f;
    for (int c = 0; c < chunks; ++c) s += part[((size_t)c * R + r) * K + k];
    float* y = Y + (size_t)r * ldy + k;
    *y = accumulate ? *y + s : s;
}

// ---- gather backward: dBEV[b, y, x, :] += w * dfeat[b, n, pt*C : (pt+1)*C] for the four corners of every point ---------------------
struct GatherBwdArgs {
    const float* dfeat;
    const float* boxes;
    float* dbev;
    int H, W, C, N, box_stride, box_batch_stride, num_point, row_stride, batch_stride;
    float pc_x0, pc_y0, vs_x, vs_y, out_stride_px;
};
// the four corner pixels (y0 x0, y1 x0, y0 x1, y1 x1 - the order of the forward's terms) and weights of point `pt` of box n of item b:
// the forward's arithmetic (bev_gather_kernel, center_utils.py:92-121), operation for operation
__device__ __forceinline__ void gather_bwd_corners(const GatherBwdArgs& a, int b, int n, int pt, int (&pix)[4], float (&wgt)[4]) {
    const float* box = a.boxes + (size_t)b * a.box_batch_stride + (size_t)n * a.box_stride;
    const float cx = box[0], cy = box[1];
    float px = cx, py = cy;
    const int edge = (a.num_point == 5) ? pt - 1 : (a.num_point == 4 ? pt : -1);
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
    const float x = __fdiv_rn(__fdiv_rn(__fsub_rn(px, a.pc_x0), a.vs_x), a.out_stride_px);
    const float y = __fdiv_rn(__fdiv_rn(__fsub_rn(py, a.pc_y0), a.vs_y), a.out_stride_px);
    auto clampi = [](float f, int hi) -> int {
        if (!(f > -2.0f)) return -1;
        if (f > (float)(hi + 1)) return hi + 1;
        return (int)f;
    };
    int x0 = clampi(floorf(x), a.W), y0 = clampi(floorf(y), a.H);
    int x1 = x0 + 1, y1 = y0 + 1;
    x0 = min(max(x0, 0), a.W - 1); x1 = min(max(x1, 0), a.W - 1);
    y0 = min(max(y0, 0), a.H - 1); y1 = min(max(y1, 0), a.H - 1);
    wgt[0] = (x1 - x) * (y1 - y); wgt[1] = (x1 - x) * (y - y0); wgt[2] = (x - x0) * (y1 - y); wgt[3] = (x - x0) * (y - y0);
    pix[0] = y0 * a.W + x0; pix[1] = y1 * a.W + x0; pix[2] = y0 * a.W + x1; pix[3] = y1 * a.W + x1;
}

// Scatter-add with float atomics: the order of the additions into a pixel several points touch is whatever the hardware makes it
// (results agree to fp32 rounding, not bit for bit).  Only for maps of 2^17 pixels or more (the sorted form's key has 17 pixel bits).
__global__ __launch_bounds__(256) void bev_gather_bwd_atomic_kernel(GatherBwdArgs a, int total_points) {
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    if (wave >= total_points) return;
    const int pt = wave % a.num_point, n = (wave / a.num_point) % a.N, b = wave / (a.num_point * a.N);
    int pix[4];
    float wgt[4];
    gather_bwd_corners(a, b, n, pt, pix, wgt);
    float* im = a.dbev + (size_t)b * a.H * a.W * a.C;
    const float* g = a.dfeat + (size_t)b * a.batch_stride + (size_t)n * a.row_stride + (size_t)pt * a.C;
    for (int ch = lane; ch < a.C; ch += 64) {
        const float v = g[ch];
#pragma unroll
        for (int k = 0; k < 4; ++k) atomicAdd(im + (size_t)pix[k] * a.C + ch, __fmul_rn(v, wgt[k]));
    }
}

// The same sums in a FIXED order (bit-reproducible).  G workgroups per batch item (G a power of two); workgroup q takes the pixels with
// pixel mod G == q: it walks the item's N * num_point * 4 contributions (contribution i = corner i & 3 of point i >> 2), keeps the
// ones that fall on its pixels - key (pixel << 15 | i) and bilinear weight in LDS -, a bitonic sort puts the contributions of a pixel
// next to each other in ascending i, and the first of every run - a quarter wavefront per run, lanes over the channels - adds the run's
// terms w * dfeat in that order and adds the sum to the pixel (nobody else writes that pixel).  Items with more than 16384
// contributions are taken in chunks of 16384, one after the other; a pixel keeps its workgroup over the chunks.
// (One workgroup per item sorting all 8000 keys of N = 500 and walking the runs a wavefront each: 1.1 ms; the runs by quarter
// wavefronts, 16 workgroups per item each sorting everything: 0.122 ms = 8 us keys + 51 us sort + 63 us runs; this form: see DESIGN.md.)
constexpr int GB_CHUNK = 16384, GB_THREADS = 1024;
// the compare-exchange steps with partner distance j <= 64 of the merge widths k_lo .. k_hi, on blocks of 128 keys held two per lane
// (positions base + lane and base + 64 + lane): j = 64 compares the lane's own two keys, smaller j exchange across lanes
__device__ __forceinline__ void gb_wave_steps(uint32_t* keys, int npad, int k_lo, int k_hi, int wave, int lane) {
    for (int base = wave * 128; base < npad; base += (GB_THREADS / 64) * 128) {
        const int p0 = base + lane, p1 = p0 + 64;
        uint32_t x0 = keys[p0], x1 = keys[p1];
        for (int k = k_lo; k <= k_hi; k <<= 1) {
            const bool up0 = (p0 & k) == 0, up1 = (p1 & k) == 0;
            for (int j = min(k >> 1, 64); j > 0; j >>= 1) {
                if (j == 64) {
                    if ((x0 > x1) == up0) {  // (p0 and p1 differ in bit 6 only: same direction for k > 64)
                        const uint32_t t = x0;
                        x0 = x1;
                        x1 = t;
                    }
                } else {
                    const uint32_t y0 = (uint32_t)__shfl_xor((int)x0, j, 64), y1 = (uint32_t)__shfl_xor((int)x1, j, 64);
                    const bool lower = (lane & j) == 0;
                    x0 = (lower == up0) ? min(x0, y0) : max(x0, y0);
                    x1 = (lower == up1) ? min(x1, y1) : max(x1, y1);
                }
            }
        }
        keys[p0] = x0;
        keys[p1] = x1;
    }
}
__global__ __launch_bounds__(GB_THREADS) void bev_gather_bwd_sorted_kernel(GatherBwdArgs a, int G) {
    extern __shared__ uint32_t gb_keys[];  // [npad] keys, then [chunk] weights
    const int b = blockIdx.x / G, q = blockIdx.x % G, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int per = a.N * a.num_point * 4;
    float* im = a.dbev + (size_t)b * a.H * a.W * a.C;
    int npad_max = 128;
    while (npad_max < per && npad_max < GB_CHUNK) npad_max <<= 1;
    float* gb_w = reinterpret_cast<float*>(gb_keys + npad_max);
    __shared__ int gb_cnt;
    const uint32_t gmask = (uint32_t)G - 1u;
    const bool vec4 = a.C % 4 == 0 && (a.row_stride | a.batch_stride) % 4 == 0 && (((uintptr_t)a.dfeat | (uintptr_t)a.dbev) & 15) == 0;
    for (int c0 = 0; c0 < per; c0 += GB_CHUNK) {
        const int n = min(GB_CHUNK, per - c0);
        if (tid == 0) gb_cnt = 0;
        __syncthreads();
        for (int p = tid; p < (n + 3) / 4; p += GB_THREADS) {  // a point: its (up to) four contributions
            const int pg = (c0 >> 2) + p;
            int pix[4];
            float wgt[4];
            gather_bwd_corners(a, b, pg / a.num_point, pg % a.num_point, pix, wgt);
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (4 * p + k < n && ((uint32_t)pix[k] & gmask) == (uint32_t)q) {
                    gb_keys[atomicAdd(&gb_cnt, 1)] = ((uint32_t)pix[k] << 15) | (uint32_t)(4 * p + k);  // (any order: sorted below)
                    gb_w[4 * p + k] = wgt[k];
                }
        }
        __syncthreads();
        const int m = gb_cnt;  // this workgroup's contributions of the chunk
        int npad = 128;
        while (npad < m) npad <<= 1;
        for (int i = m + tid; i < npad; i += GB_THREADS) gb_keys[i] = 0xffffffffu;
        __syncthreads();
#ifndef SHASTA_GB_SKIP_SORT  // timing diagnostic only (tools/time_gather_bwd.py): wrong results without the sort
        gb_wave_steps(gb_keys, npad, 2, 128, wave, lane);
        __syncthreads();
        for (int k = 256; k <= npad; k <<= 1) {
            for (int j = k >> 1; j >= 128; j >>= 1) {
                for (int t = tid; t < (npad >> 1); t += GB_THREADS) {
                    const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1)), l = i | j;
                    const uint32_t x = gb_keys[i], y = gb_keys[l];
                    if ((x > y) == ((i & k) == 0)) {
                        gb_keys[i] = y;
                        gb_keys[l] = x;
                    }
                }
                __syncthreads();
            }
            gb_wave_steps(gb_keys, npad, k, k, wave, lane);
            __syncthreads();
        }
#endif
        // runs -> pixels.  A quarter wavefront (16 lanes x 4 channels) per run, so that a wavefront has four runs' loads in flight;
        // quarter qw walks the sorted positions [m qw / 64, m (qw + 1) / 64) and takes the runs that START there.
#ifndef SHASTA_GB_SKIP_RUNS  // timing diagnostic only
        {
            const int qw = tid >> 4, ql = tid & 15;
            const int r1 = (int)((long)m * (qw + 1) / 64);
            int s = (int)((long)m * qw / 64);
            for (;;) {
                while (s < r1 && s > 0 && (gb_keys[s - 1] >> 15) == (gb_keys[s] >> 15)) ++s;  // the next start of a run
                if (!__any(s < r1)) break;  // the whole wavefront is through
                if (s < r1) {
                    const uint32_t pixel = gb_keys[s] >> 15;
                    float* o = im + (size_t)pixel * a.C;
                    if (vec4) {
                        for (int ch = 4 * ql; ch < a.C; ch += 64) {
                            const f32x4 cur = *reinterpret_cast<const f32x4*>(o + ch);
                            f32x4 sum = {0.0f, 0.0f, 0.0f, 0.0f};
                            for (int t = s; t < m && (gb_keys[t] >> 15) == pixel; ++t) {
                                const int li = (int)(gb_keys[t] & 0x7fffu), pg = (c0 + li) >> 2;
                                const float* g = a.dfeat + (size_t)b * a.batch_stride + (size_t)(pg / a.num_point) * a.row_stride + (size_t)(pg % a.num_point) * a.C;
                                const f32x4 v = *reinterpret_cast<const f32x4*>(g + ch);
                                const float w = gb_w[li];
#pragma unroll
                                for (int e = 0; e < 4; ++e) sum[e] = __fadd_rn(sum[e], __fmul_rn(v[e], w));
                            }
                            f32x4 r;
#pragma unroll
                            for (int e = 0; e < 4; ++e) r[e] = __fadd_rn(cur[e], sum[e]);
                            *reinterpret_cast<f32x4*>(o + ch) = r;
                        }
                    } else {
                        for (int ch = ql; ch < a.C; ch += 16) {
                            float sum = 0.0f;
                            for (int t = s; t < m && (gb_keys[t] >> 15) == pixel; ++t) {
                                const int li = (int)(gb_keys[t] & 0x7fffu), pg = (c0 + li) >> 2;
                                const float* g = a.dfeat + (size_t)b * a.batch_stride + (size_t)(pg / a.num_point) * a.row_stride + (size_t)(pg % a.num_point) * a.C;
                                sum = __fadd_rn(sum, __fmul_rn(g[ch], gb_w[li]));
                            }
                            o[ch] = __fadd_rn(o[ch], sum);
                        }
                    }
                    ++s;
                }
            }
        }
#endif
        __syncthreads();  // the next chunk reuses the LDS arrays - and may add to the same pixels: its reads come behind these stores
                          // (__syncthreads = release fence, barrier, acquire fence at workgroup scope; the wavefronts share the CU's L1)
    }
}

}  // namespace shasta

using namespace shasta;

extern "C" int shasta_pair_hidden_f32(const float* UP, int ldp, const float* UC, int ldc, int B, int T, int D, int E, float* H,
                                      shasta_stream_t stream) {
    SHASTA_REQUIRE(UP && UC && H && E >= 1 && ldp >= E && ldc >= E, "pair_hidden: bad argument");
    const long total = (long)B * T * D * E;
    if (total == 0) return SHASTA_OK;
    SHASTA_REQUIRE((total + 255) / 256 < (1L << 31), "pair_hidden: too many pairs");
    hipLaunchKernelGGL(pair_hidden_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, as_stream(stream), UP, ldp, UC, ldc, T, D, E,
                       total, H);
    return check_launch("pair_hidden");
}

extern "C" int shasta_pair_reduce_f32(const float* gZ, int B, int T, int D, int E, float* gUP, float* gUC, shasta_stream_t stream) {
    SHASTA_REQUIRE(gZ && gUP && gUC && E >= 1 && E <= 256 && T == D, "pair_reduce: bad argument (E <= 256, T == D)");
    if (B == 0 || T == 0) return SHASTA_OK;
    int EP = 1;
    while (EP < E) EP *= 2;
    hipLaunchKernelGGL(pair_reduce_kernel, dim3(T, 2, B), dim3(256), 0, as_stream(stream), gZ, T, D, E, EP, gUP, gUC);
    return check_launch("pair_reduce");
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

extern "C" int shasta_affinity_loss_f32(const float* m1, const float* m2, const float* gt, int B, int N, float* ws, float* sums,
                                        shasta_stream_t stream) {
    SHASTA_REQUIRE(m1 && m2 && gt && ws && sums && B > 0 && N > 0, "affinity_loss: bad argument");
    hipLaunchKernelGGL(loss_rows_kernel, dim3(N + 2, B), dim3(256), 0, as_stream(stream), m1, m2, gt, N, ws);
    int rc = check_launch("affinity_loss rows");
    if (rc) return rc;
    hipLaunchKernelGGL(loss_finish_kernel, dim3(1), dim3(256), 0, as_stream(stream), ws, B * (N + 2), sums);
    return check_launch("affinity_loss");
}

extern "C" int shasta_affinity_loss_bwd_f32(const float* m1, const float* m2, const float* gt, const float* sums, const float* gloss, int B,
                                            int N, float* g1, float* g2, shasta_stream_t stream) {
    SHASTA_REQUIRE(m1 && m2 && gt && sums && gloss && g1 && g2 && B > 0 && N > 0, "affinity_loss_bwd: bad argument");
    const long n1 = (long)B * N * (N + 2);
    hipLaunchKernelGGL(loss_bwd_kernel, dim3((unsigned)((n1 + 255) / 256)), dim3(256), 0, as_stream(stream), m1, m2, gt, sums, gloss, N, n1, g1,
                       g2);
    return check_launch("affinity_loss_bwd");
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

extern "C" int shasta_colsum_f32(const float* Y, int ldy, int M, int N, float* out, float* ws, size_t ws_bytes,
                                 shasta_stream_t stream) {
    SHASTA_REQUIRE(Y && out && M >= 0 && N >= 0, "colsum: bad argument");
    if (N == 0) return SHASTA_OK;
    // row chunks: enough blocks to fill the chip, at least 64 rows each, bounded by the scratch the caller gave
    const int colblocks = N >= 64 ? cdiv(N, 64) : 1;
    int chunks = std::min(std::max(1, 1024 / colblocks), std::max(1, M / 64));
    if (!ws) chunks = 1;
    else chunks = (int)std::min<size_t>((size_t)chunks, ws_bytes / (sizeof(float) * (size_t)N));
    chunks = std::max(chunks, 1);
    const int rpc = cdiv(std::max(M, 1), chunks);
    chunks = cdiv(std::max(M, 1), rpc);
    float* dst = chunks == 1 ? out : ws;
    if (N >= 64) {
        hipLaunchKernelGGL(colsum_kernel, dim3(colblocks, chunks), dim3(256), 0, as_stream(stream), Y, ldy, M, N, rpc, dst);
    } else {
        int NP = 1;
        while (NP < N) NP *= 2;
        hipLaunchKernelGGL(colsum_narrow_kernel, dim3(1, chunks), dim3(256), 0, as_stream(stream), Y, ldy, M, N, NP, rpc, dst);
    }
    int rc = check_launch("colsum");
    if (rc || chunks == 1) return rc;
    hipLaunchKernelGGL(colsum_finish_kernel, dim3(cdiv(N, 4)), dim3(256), 0, as_stream(stream), ws, chunks, N, out);
    return check_launch("colsum_finish");
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
    SHASTA_REQUIRE(total < (1L << 29) && H > 0 && W > 0 && C > 0, "bev_gather_bwd: bad size");
    GatherBwdArgs a;
    a.dfeat = dfeat; a.boxes = boxes; a.dbev = dbev;
    a.H = H; a.W = W; a.C = C; a.N = N; a.box_stride = box_stride; a.box_batch_stride = box_batch_stride; a.num_point = num_point;
    a.row_stride = row_stride; a.batch_stride = batch_stride;
    a.pc_x0 = pc_x0; a.pc_y0 = pc_y0; a.vs_x = vs_x; a.vs_y = vs_y; a.out_stride_px = out_stride;
    if ((long)H * W < (1L << 17)) {  // strictly: the largest key must stay below the padding word
        const int per = N * num_point * 4;
        int npad = 128;
        while (npad < per && npad < GB_CHUNK) npad <<= 1;
        const size_t lds = (size_t)(npad + (per < GB_CHUNK ? per : GB_CHUNK)) * sizeof(uint32_t);
        if (lds > 64 * 1024 && hipFuncSetAttribute((const void*)bev_gather_bwd_sorted_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
            set_error_msg("bev_gather_bwd: the device refuses 128 KB of LDS per workgroup");
            return SHASTA_E_UNSUPPORTED;
        }
        int G = 1;  // workgroups per batch item: two per CU as long as the items are few
        while (G < 64 && (long)B * G < 512) G <<= 1;
        hipLaunchKernelGGL(bev_gather_bwd_sorted_kernel, dim3((unsigned)(B * G)), dim3(GB_THREADS), lds, as_stream(stream), a, G);
    } else {
        hipLaunchKernelGGL(bev_gather_bwd_atomic_kernel, dim3(cdiv((int)total, 4)), dim3(256), 0, as_stream(stream), a, (int)total);
    }
    return check_launch("bev_gather_bwd");
}

// d_dyn (device, 4 floats, or NULL): when given, lr, betas and step are ignored - the kernels read lr / bc1, 1 / sqrt(bc2) and the betas
// from it (shasta_adam_prepare_f32 writes them from a device-side step counter and {lr, beta1, beta2} once per optimizer step)
static AdamArgs make_adam_args(float lr, float beta1, float beta2, float eps, float weight_decay, int step, const float* d_dyn) {
    const double s = step >= 1 ? (double)step : 1.0;
    const double bc1 = 1.0 - pow((double)beta1, s), bc2 = 1.0 - pow((double)beta2, s);
    AdamArgs a;
    a.lr_over_bc1 = (float)((double)lr / bc1);
    a.beta1 = beta1;
    a.beta2 = beta2;
    a.eps = eps;
    a.weight_decay = weight_decay;
    a.rsqrt_bc2 = (float)(1.0 / sqrt(bc2));
    a.dyn = d_dyn;
    return a;
}

extern "C" int shasta_adam_prepare_f32(int* d_step, const float* d_hyper, float* d_dyn, shasta_stream_t stream) {
    SHASTA_REQUIRE(d_step && d_hyper && d_dyn, "adam_prepare: null pointer");
    hipLaunchKernelGGL(adam_prepare_kernel, dim3(1), dim3(1), 0, as_stream(stream), d_step, d_hyper, d_dyn);
    return check_launch("adam_prepare");
}

extern "C" int shasta_adam_step_f32(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, long n, float lr, float beta1,
                                    float beta2, float eps, float weight_decay, int step, const float* d_dyn, shasta_stream_t stream) {
    SHASTA_REQUIRE(param && grad && exp_avg && exp_avg_sq && n >= 0 && (step >= 1 || d_dyn), "adam_step: bad argument");
    SHASTA_REQUIRE((((uintptr_t)param | (uintptr_t)grad | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15) == 0,
                   "adam_step: tensors must be 16-byte aligned");
    if (n == 0) return SHASTA_OK;
    const AdamArgs a = make_adam_args(lr, beta1, beta2, eps, weight_decay, step, d_dyn);
    const long blocks = std::min<long>((n / 4 + 255) / 256 + 1, 256L * 16);
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), param, grad, exp_avg, exp_avg_sq, n, a);
    return check_launch("adam_step");
}

extern "C" int shasta_adam_multi_f32(int count, float* const* param, const float* const* grad, float* const* exp_avg,
                                     float* const* exp_avg_sq, const long* n, float lr, float beta1, float beta2, float eps,
                                     float weight_decay, int step, const float* d_dyn, shasta_stream_t stream) {
    SHASTA_REQUIRE(count >= 0 && (step >= 1 || d_dyn) && (count == 0 || (param && grad && exp_avg && exp_avg_sq && n)), "adam_multi: bad argument");
    const AdamArgs a = make_adam_args(lr, beta1, beta2, eps, weight_decay, step, d_dyn);
    for (int k0 = 0; k0 < count; k0 += ADAM_MULTI) {
        AdamMulti t;
        const int c = std::min(ADAM_MULTI, count - k0);
        long nmax = 0;
        for (int k = 0; k < ADAM_MULTI; ++k) {
            const int j = k0 + std::min(k, c - 1);  // (unused slots repeat the last tensor with n = 0)
            SHASTA_REQUIRE(param[j] && grad[j] && exp_avg[j] && exp_avg_sq[j] && n[j] >= 0, "adam_multi: null tensor");
            t.p[k] = param[j]; t.g[k] = grad[j]; t.m[k] = exp_avg[j]; t.v[k] = exp_avg_sq[j];
            t.n[k] = k < c ? n[j] : 0;
            nmax = std::max(nmax, t.n[k]);
        }
        if (nmax == 0) continue;
        const unsigned bx = (unsigned)std::min<long>((nmax + 255) / 256, 64);
        hipLaunchKernelGGL(adam_multi_kernel, dim3(bx, c), dim3(256), 0, as_stream(stream), t, a);
        int rc = check_launch("adam_multi");
        if (rc) return rc;
    }
    return SHASTA_OK;
}

namespace {
int adam_lowrank_rows_per_block(int H, int K, int cols) {
    const int kblocks = shasta::cdiv(K / cols, 256);
    return std::max(1, std::min(64, shasta::cdiv(H, std::max(1, 4096 / kblocks))));  // >= ~4096 workgroups, <= 64 rows each
}
}  // namespace

extern "C" size_t shasta_adam_lowrank_dx_workspace_bytes(int H, int K, int Rdx) {
    if (H <= 0 || K <= 0 || Rdx <= 0) return 0;
    const int chunks = std::max(shasta::cdiv(H, adam_lowrank_rows_per_block(H, K, 4)), shasta::cdiv(H, adam_lowrank_rows_per_block(H, K, 2)));
    return (size_t)chunks * Rdx * K * sizeof(float);  // (4 columns per thread up to rank 16, 2 above)
}

extern "C" int shasta_adam_lowrank_dx_f32(float* param, float* exp_avg, float* exp_avg_sq, int H, int K, const float* G, int ldg, const float* X,
                                          int ldx, int R, const float* Gdx, int ldgdx, int Rdx, float* Y, long ldy, int accumulate, void* workspace,
                                          size_t workspace_bytes, float lr, float beta1, float beta2, float eps, float weight_decay, int step,
                                          const float* d_dyn, shasta_stream_t stream) {
    SHASTA_REQUIRE(param && exp_avg && exp_avg_sq && G && X && Gdx && Y && workspace && H >= 1 && K >= 4 && (step >= 1 || d_dyn), "adam_lowrank_dx: bad argument");
    SHASTA_REQUIRE(R >= 1 && R <= 64 && Rdx >= 1 && Rdx <= 16, "adam_lowrank_dx: 1 <= R <= 64, 1 <= Rdx <= 16");
    SHASTA_REQUIRE(K % 4 == 0 && ldx % 4 == 0 && ldg >= H && ldx >= K && ldgdx >= H, "adam_lowrank_dx: K and ldx multiples of 4, ldg, ldgdx >= H, ldx >= K");
    SHASTA_REQUIRE((((uintptr_t)param | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq | (uintptr_t)X | (uintptr_t)workspace) & 15) == 0,
                   "adam_lowrank_dx: 16-byte alignment");
    if (workspace_bytes < shasta_adam_lowrank_dx_workspace_bytes(H, K, Rdx)) {
        set_error_msg("adam_lowrank_dx: workspace too small");
        return SHASTA_E_WORKSPACE;
    }
    const AdamArgs a = make_adam_args(lr, beta1, beta2, eps, weight_decay, step, d_dyn);
    const int cols = R <= 16 ? 4 : 2;
    const int kblocks = cdiv(K / cols, 256), rpb = adam_lowrank_rows_per_block(H, K, cols), chunks = cdiv(H, rpb);
    float* part = static_cast<float*>(workspace);
    auto launch = [&](auto kern) {
        hipLaunchKernelGGL(kern, dim3(kblocks, chunks), dim3(256), 0, as_stream(stream), param, exp_avg, exp_avg_sq, G, ldg, X, ldx, R, H, K, rpb, a,
                           Gdx, ldgdx, Rdx, part);
    };
    if (R <= 8) Rdx <= 8 ? launch(adam_lowrank_kernel<8, 4, 8>) : launch(adam_lowrank_kernel<8, 4, 16>);
    else if (R <= 16) Rdx <= 8 ? launch(adam_lowrank_kernel<16, 4, 8>) : launch(adam_lowrank_kernel<16, 4, 16>);
    else if (R <= 32) Rdx <= 8 ? launch(adam_lowrank_kernel<32, 2, 8>) : launch(adam_lowrank_kernel<32, 2, 16>);
    else Rdx <= 8 ? launch(adam_lowrank_kernel<64, 2, 8>) : launch(adam_lowrank_kernel<64, 2, 16>);
    int rc = check_launch("adam_lowrank_dx");
    if (rc) return rc;
    hipLaunchKernelGGL(smallm_finish_kernel, dim3((unsigned)(((long)Rdx * K + 255) / 256)), dim3(256), 0, as_stream(stream), part, chunks, Rdx, K, Y,
                       ldy, accumulate);
    return check_launch("adam_lowrank_dx finish");
}

extern "C" int shasta_adam_lowrank_f32(float* param, float* exp_avg, float* exp_avg_sq, int H, int K, const float* G, int ldg, const float* X,
                                       int ldx, int R, float lr, float beta1, float beta2, float eps, float weight_decay, int step,
                                       const float* d_dyn, shasta_stream_t stream) {
    SHASTA_REQUIRE(param && exp_avg && exp_avg_sq && G && X && H >= 0 && K >= 0 && (step >= 1 || d_dyn), "adam_lowrank: bad argument");
    SHASTA_REQUIRE(R >= 1 && R <= 64, "adam_lowrank: 1 <= R <= 64 (frame-pairs of a step over all ranks)");
    SHASTA_REQUIRE(K % 4 == 0 && ldx % 4 == 0 && ldg >= H && ldx >= K, "adam_lowrank: K and ldx multiples of 4, ldg >= H, ldx >= K");
    SHASTA_REQUIRE((((uintptr_t)param | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq | (uintptr_t)X) & 15) == 0, "adam_lowrank: 16-byte alignment");
    if (H == 0 || K == 0) return SHASTA_OK;
    const AdamArgs a = make_adam_args(lr, beta1, beta2, eps, weight_decay, step, d_dyn);
    auto launch = [&](auto kern, int cols) {
        const int kblocks = cdiv(K / cols, 256);
        const int rpb = adam_lowrank_rows_per_block(H, K, cols);
        hipLaunchKernelGGL(kern, dim3(kblocks, cdiv(H, rpb)), dim3(256), 0, as_stream(stream), param, exp_avg, exp_avg_sq, G, ldg, X, ldx, R, H, K,
                           rpb, a, (const float*)nullptr, 0, 0, (float*)nullptr);
    };
    // Columns per thread.  Large matrices (the 2000 x 128000 first layers at N = 500): 4 up to rank 16, 2 above (1.10 ms per matrix at
    // R = 8; one column at R = 64: 1.31 -> 1.37 ms).  Small ones (the car configuration's 450 x 28800): fewer columns = more workgroups
    // and, above rank 32, half the registers for the X rows - 0.072 -> 0.063 ms at R = 8, 0.069 -> 0.063 at R = 32, 0.113 -> 0.092 at
    // R = 64 (two columns at R = 16: 0.056 -> 0.21, the scalar loads of G then bound it: kept at 4).  Same arithmetic per element.
    const bool small = (long)H * K <= (1L << 26);
    if (R <= 8) small ? launch(adam_lowrank_kernel<8, 2>, 2) : launch(adam_lowrank_kernel<8, 4>, 4);
    else if (R <= 16) launch(adam_lowrank_kernel<16, 4>, 4);
    else if (R <= 32) small ? launch(adam_lowrank_kernel<32, 1>, 1) : launch(adam_lowrank_kernel<32, 2>, 2);
    else small ? launch(adam_lowrank_kernel<64, 1>, 1) : launch(adam_lowrank_kernel<64, 2>, 2);
    return check_launch("adam_lowrank");
}

extern "C" int shasta_scale_f32(float* x, long n, float alpha, shasta_stream_t stream) {
    SHASTA_REQUIRE(x && n >= 0, "scale: bad argument");
    if (n == 0) return SHASTA_OK;
    hipLaunchKernelGGL(scale_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, as_stream(stream), x, n, alpha);
    return check_launch("scale");
}

extern "C" int shasta_lowrank_outer_f32(const float* G, int ldg, const float* X, int ldx, int R, int H, int K, float* dW,
                                        shasta_stream_t stream) {
    SHASTA_REQUIRE(G && X && dW && R >= 1 && R <= 16 && H >= 0 && K >= 0, "lowrank_outer: bad argument (1 <= R <= 16)");
    SHASTA_REQUIRE(K % 4 == 0 && ldx % 4 == 0 && (((uintptr_t)X | (uintptr_t)dW) & 15) == 0, "lowrank_outer: K, ldx multiples of 4, 16-byte aligned");
    if (H == 0 || K == 0) return SHASTA_OK;
    const int kblocks = cdiv(K / 4, 256);
    const int rpb = std::max(1, std::min(64, cdiv(H, std::max(1, 2048 / kblocks))));  // >= ~2048 workgroups, <= 64 rows each
    dim3 grid(kblocks, cdiv(H, rpb));
    if (R <= 4) hipLaunchKernelGGL(lowrank_outer_kernel<4>, grid, dim3(256), 0, as_stream(stream), G, ldg, X, ldx, R, H, K, rpb, dW);
    else if (R <= 8) hipLaunchKernelGGL(lowrank_outer_kernel<8>, grid, dim3(256), 0, as_stream(stream), G, ldg, X, ldx, R, H, K, rpb, dW);
    else hipLaunchKernelGGL(lowrank_outer_kernel<16>, grid, dim3(256), 0, as_stream(stream), G, ldg, X, ldx, R, H, K, rpb, dW);
    return check_launch("lowrank_outer");
}

extern "C" size_t shasta_smallm_nn_workspace_bytes(int R, int H, int K) {
    const int kblocks = cdiv(std::max(K, 4) / 4, 256);
    const int chunks = std::max(1, std::min(cdiv(std::max(H, 1), 64), cdiv(2048, kblocks)));
    return (size_t)chunks * std::max(R, 1) * std::max(K, 4) * sizeof(float);
}

extern "C" int shasta_smallm_nn_f32(const float* G, int ldg, const float* W, int R, int H, int K, float* Y, long ldy, int accumulate,
                                    void* workspace, size_t workspace_bytes, shasta_stream_t stream) {
    SHASTA_REQUIRE(G && W && Y && workspace && R >= 1 && R <= 16 && H >= 1 && K >= 4, "smallm_nn: bad argument (1 <= R <= 16)");
    SHASTA_REQUIRE(K % 4 == 0 && (((uintptr_t)W | (uintptr_t)workspace) & 15) == 0, "smallm_nn: K multiple of 4, 16-byte aligned");
    if (workspace_bytes < shasta_smallm_nn_workspace_bytes(R, H, K)) {
        set_error_msg("smallm_nn: workspace too small");
        return SHASTA_E_WORKSPACE;
    }
    const int kblocks = cdiv(K / 4, 256);
    int chunks = std::max(1, std::min(cdiv(H, 64), cdiv(2048, kblocks)));
    const int rpc = cdiv(H, chunks);
    chunks = cdiv(H, rpc);
    float* part = static_cast<float*>(workspace);
    dim3 grid(kblocks, chunks);
    if (R <= 4) hipLaunchKernelGGL(smallm_nn_kernel<4>, grid, dim3(256), 0, as_stream(stream), G, ldg, W, R, H, K, rpc, part);
    else if (R <= 8) hipLaunchKernelGGL(smallm_nn_kernel<8>, grid, dim3(256), 0, as_stream(stream), G, ldg, W, R, H, K, rpc, part);
    else hipLaunchKernelGGL(smallm_nn_kernel<16>, grid, dim3(256), 0, as_stream(stream), G, ldg, W, R, H, K, rpc, part);
    int rc = check_launch("smallm_nn");
    if (rc) return rc;
    hipLaunchKernelGGL(smallm_finish_kernel, dim3((unsigned)(((long)R * K + 255) / 256)), dim3(256), 0, as_stream(stream), part, chunks, R, K, Y,
                       ldy, accumulate);
    return check_launch("smallm_finish");
}
