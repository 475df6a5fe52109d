// K6: aff row-MLP + the two softmaxes.  Restates det3d/models/tracker/shasta.py:94-109 and :323-325:
//   matched  = aff(residual)                     six nn.Linear over the D axis, applied to each of the T rows
//   matched1 = softmax(matched[:, :-2, :], dim=2)   (B, N, N+2): each previous detection over {dets, dead, FN}
//   matched2 = softmax(matched[:, :, :-2], dim=1)   (B, N+2, N): each detection over {prev dets, newborn, FP}
// The six layers are six launches of the generic matrix-core GEMM over the B*T rows (bias + ReLU fused).
#include <algorithm>

#include "common.hpp"
#include "pair_layout.hpp"

namespace shasta {

// one wave per (b, t < N): softmax over the D entries of the row
// block = 64 detections x 16 track groups (1024 threads): softmax over the T rows of each column d < N.  A wave reads
// 64 consecutive columns of one row (256 B, coalesced) and keeps its <= MAXR rows of the column in registers (one pass
// over memory); max and sum are combined across the 16 groups in a fixed order.  MAXR = ceil(T / 16) <= 128.
template <int MAXR>
__global__ __launch_bounds__(1024) void softmax_cols_kernel(const float* __restrict__ matched, float* __restrict__ m2,
                                                            int N, int T, int ld) {
    __shared__ float red[64][17];
    const int b = blockIdx.y, dl = threadIdx.x & 63, tg = threadIdx.x >> 6;
    const int d = blockIdx.x * 64 + dl;
    const int dcl = min(d, N - 1);
    const float* x = matched + (size_t)b * T * ld + dcl;
    float v[MAXR];
    float mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < MAXR; ++i) {
        const int t = tg + 16 * i;
        v[i] = t < T ? x[(size_t)t * ld] : -INFINITY;
        mx = fmaxf(mx, v[i]);
    }
    red[dl][tg] = mx;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 16; ++i) mx = fmaxf(mx, red[dl][i]);
    __syncthreads();
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < MAXR; ++i) {
        v[i] = (tg + 16 * i < T) ? expf(v[i] - mx) : 0.0f;
        s += v[i];
    }
    red[dl][tg] = s;
    __syncthreads();
    s = 0.0f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += red[dl][i];
    if (d >= N) return;
    float* o = m2 + (size_t)b * T * N + d;
#pragma unroll
    for (int i = 0; i < MAXR; ++i) {
        const int t = tg + 16 * i;
        if (t < T) o[(size_t)t * N] = v[i] / s;
    }
}

// The same with TWO columns per lane (N even, ld even): a wave reads 128 consecutive columns of a row = 512 contiguous bytes per
// instruction instead of 256 (8-byte loads and stores).  Identical arithmetic per column - max, exp, fixed-order sums over the same
// 16 row groups - so the results are bit-identical to softmax_cols_kernel; 282 -> ... us at 512 frame-pairs (3.96 TB/s before).
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int MAXR>
__global__ __launch_bounds__(1024) void softmax_cols2_kernel(const float* __restrict__ matched, float* __restrict__ m2, int N, int T, int ld) {
    __shared__ f32x2 red[64][17];
    const int b = blockIdx.y, dl = threadIdx.x & 63, tg = threadIdx.x >> 6;
    const int d = blockIdx.x * 128 + 2 * dl;
    const int dcl = min(d, N - 2);  // N is even: the pair (dcl, dcl + 1) is always inside the row
    const float* x = matched + (size_t)b * T * ld + dcl;
    f32x2 v[MAXR];
    f32x2 mx = {-INFINITY, -INFINITY};
#pragma unroll
    for (int i = 0; i < MAXR; ++i) {
        const int t = tg + 16 * i;
        v[i] = t < T ? *reinterpret_cast<const f32x2*>(x + (size_t)t * ld) : f32x2{-INFINITY, -INFINITY};
        mx[0] = fmaxf(mx[0], v[i][0]);
        mx[1] = fmaxf(mx[1], v[i][1]);
    }
    red[dl][tg] = mx;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const f32x2 r = red[dl][i];
        mx[0] = fmaxf(mx[0], r[0]);
        mx[1] = fmaxf(mx[1], r[1]);
    }
    __syncthreads();
    f32x2 s = {0.0f, 0.0f};
#pragma unroll
    for (int i = 0; i < MAXR; ++i) {
        const bool in = tg + 16 * i < T;
        v[i][0] = in ? expf(v[i][0] - mx[0]) : 0.0f;
        v[i][1] = in ? expf(v[i][1] - mx[1]) : 0.0f;
        s[0] += v[i][0];
        s[1] += v[i][1];
    }
    red[dl][tg] = s;
    __syncthreads();
    s = f32x2{0.0f, 0.0f};
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const f32x2 r = red[dl][i];
        s[0] += r[0];
        s[1] += r[1];
    }
    if (d >= N) return;
    float* o = m2 + (size_t)b * T * N + d;
#pragma unroll
    for (int i = 0; i < MAXR; ++i) {
        const int t = tg + 16 * i;
        if (t < T) *reinterpret_cast<f32x2*>(o + (size_t)t * N) = f32x2{v[i][0] / s[0], v[i][1] / s[1]};
    }
}

// ------------------------------------------------------------------------------------------------------------------
// aff_fused: the six aff layers (shasta.py:94-106) and the row softmax (:324) for 16 residual rows per workgroup.
// Rows are independent, so a workgroup keeps its 16 rows on chip from the residual to matched1: activations live in LDS
// as [row][feature] (row stride = width + 4 floats: the 16 lanes of a ds_read_b128 group then start 4 banks apart),
// weights stream from L2 as MFMA A fragments, out^T[feature][row] = W[feature][k] . h^T[k][row] with
// v_mfma_f32_16x16x4_f32 (lane l: A = W[16*blk + (l&15)][k + (l>>4)*4 + q], B = h[row l&15][same k]); the D registers of
// lane (row, kq) are output features 16*blk + 4*kq + 0..3 = one float4 of the next layer's LDS image.
// ------------------------------------------------------------------------------------------------------------------
struct AffArgs {
    const float* W[6];
    const float* bias[6];
    const float* residual;
    float* matched;  // (M, ldm) pre-softmax, for the column softmax
    float* m1;       // (B, N, D)
    int M, T, N, D, Dp, ld, ldm;
};

constexpr int AFF_WAVES = 8;  // waves per workgroup

// One work item = (16-feature output block, chunk of KG 16-wide k-groups).  The weight fragments of item i+1 are
// requested before the MFMAs of item i, so the L2 / HBM latency of the (cold) weight rows overlaps the matrix work.
// KFIX > 0: compile-time reduction length, one chunk per block; KFIX == 0: runtime Kp in chunks of 128.
// RG = 16-row groups per workgroup: every weight fragment feeds RG MFMAs (the L2 -> register weight traffic per row, the
// limiter at large batch, drops by RG).
template <bool RELU, int KFIX, int RG>
__device__ __forceinline__ void aff_layer(const float* __restrict__ W, int ldw, const float* __restrict__ bias, int Mout,
                                          int Kp, const float* hin, int ldin, float* hout, int ldout, int out_limit,
                                          int lane, int wid) {
    constexpr int KG = KFIX > 0 ? KFIX / 16 : 8;
    const int p = lane & 15, kq = lane >> 4;
    const int nb = (Mout + 15) >> 4;
    const int nchunk = KFIX > 0 ? 1 : (Kp + 127) / 128;
    const int Kv = KFIX > 0 ? KFIX : Kp;
    const float* hrow = hin + p * ldin + 4 * kq;  // row group rg adds 16 * rg * ldin
    f32x4 cur[KG], nxt[KG];
    auto load = [&](f32x4(&dst)[KG], int blk, int ch) {
        const float* wrow = W + (size_t)min(16 * blk + p, Mout - 1) * ldw + 4 * kq + ch * 128;
#pragma unroll
        for (int g = 0; g < KG; ++g) {
            dst[g] = f32x4{0, 0, 0, 0};
            if (ch * 128 + 16 * g + 4 * kq < Kv) dst[g] = *reinterpret_cast<const f32x4*>(wrow + 16 * g);
        }
    };
    int blk = wid, ch = 0;
    if (blk < nb) load(cur, blk, 0);
    f32x4 acc[RG];
    while (blk < nb) {
        int nblk = blk, nch = ch + 1;
        if (nch == nchunk) {
            nch = 0;
            nblk += AFF_WAVES;
        }
        if (nblk < nb) load(nxt, nblk, nch);
        const int f0 = 16 * blk + 4 * kq;
        if (ch == 0) {
#pragma unroll
            for (int rg = 0; rg < RG; ++rg)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[rg][r] = (f0 + r < Mout) ? bias[f0 + r] : 0.0f;
        }
#pragma unroll
        for (int g = 0; g < KG; ++g) {
            f32x4 b4[RG];
#pragma unroll
            for (int rg = 0; rg < RG; ++rg) {
                b4[rg] = f32x4{0, 0, 0, 0};
                if (ch * 128 + 16 * g + 4 * kq < Kv)
                    b4[rg] = *reinterpret_cast<const f32x4*>(hrow + 16 * rg * ldin + ch * 128 + 16 * g);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int rg = 0; rg < RG; ++rg) acc[rg] = __builtin_amdgcn_mfma_f32_16x16x4f32(cur[g][q], b4[rg][q], acc[rg], 0, 0, 0);
        }
        if (ch == nchunk - 1) {
#pragma unroll
            for (int rg = 0; rg < RG; ++rg) {
                if (RELU) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[rg][r] = relu_nan(acc[rg][r]);
                }
                if (f0 < out_limit) *reinterpret_cast<f32x4*>(hout + (p + 16 * rg) * ldout + f0) = acc[rg];
            }
        }
#pragma unroll
        for (int g = 0; g < KG; ++g) cur[g] = nxt[g];
        blk = nblk;
        ch = nch;
    }
}

template <int RG>
__global__ __launch_bounds__(64 * AFF_WAVES) void aff_fused_kernel(AffArgs a) {
    constexpr int ROWS = 16 * RG;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int XS = a.Dp + 4, HS = 132;
    float* xb = sm;                   // [ROWS][XS]  residual rows, later the matched rows
    float* ha = sm + ROWS * max(XS, HS);  // [ROWS][HS]; the xb region must also hold hb ([ROWS][HS]) when D < 128
    // the second hidden buffer lives in xb: xb is dead between layer 0 (its last reader) and layer 5 (its next writer), which
    // is exactly the lifetime of hb (written by layers 1 and 3, read by layers 2 and 4)
    float* hb = xb;                   // [ROWS][HS]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g0 = blockIdx.x * ROWS;
#pragma unroll 4
    for (int e = tid; e < ROWS * XS; e += 64 * AFF_WAVES) {
        const int p = e / XS, c = e - p * XS;
        xb[e] = (g0 + p < a.M && c < a.D) ? a.residual[(size_t)(g0 + p) * a.ld + c] : 0.0f;
    }
    __syncthreads();
    aff_layer<true, 0, RG>(a.W[0], a.Dp, a.bias[0], 128, a.Dp, xb, XS, ha, HS, 128, lane, wid);
    __syncthreads();
    aff_layer<true, 128, RG>(a.W[1], 128, a.bias[1], 64, 128, ha, HS, hb, HS, 64, lane, wid);
    __syncthreads();
    aff_layer<true, 64, RG>(a.W[2], 64, a.bias[2], 32, 64, hb, HS, ha, HS, 32, lane, wid);
    __syncthreads();
    aff_layer<true, 32, RG>(a.W[3], 32, a.bias[3], 64, 32, ha, HS, hb, HS, 64, lane, wid);
    __syncthreads();
    aff_layer<true, 64, RG>(a.W[4], 64, a.bias[4], 128, 64, hb, HS, ha, HS, 128, lane, wid);
    __syncthreads();
    aff_layer<false, 128, RG>(a.W[5], 128, a.bias[5], a.D, 128, ha, HS, xb, XS, a.Dp, lane, wid);
    __syncthreads();
    // each wave: ROWS / AFF_WAVES rows: copy to `matched` (column softmax input) and row softmax for t < N.  One exp per
    // element (kept in the LDS row), one reciprocal per row: the f32 MFMAs of the other waves share the pipe with this.
    for (int pr = 0; pr < ROWS / AFF_WAVES; ++pr) {
        const int p = (ROWS / AFF_WAVES) * wid + pr, g = g0 + p;
        if (g >= a.M) break;
        float* x = xb + p * XS;
        float* mo = a.matched + (size_t)g * a.ldm;
        float mx = -INFINITY;
        for (int d = lane; d < a.D; d += 64) {
            const float v = x[d];
            mo[d] = v;
            mx = fmaxf(mx, v);
        }
        const int b = g / a.T, t = g - b * a.T;
        if (t >= a.N) continue;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
        float s = 0.0f;
        for (int d = lane; d < a.D; d += 64) {
            const float e = expf(x[d] - mx);
            x[d] = e;  // lane d % 64 owns element d in every pass: no synchronisation needed
            s += e;
        }
        s = wave_sum(s);
        const float inv = 1.0f / s;
        float* o = a.m1 + ((size_t)b * a.N + t) * a.D;
        for (int d = lane; d < a.D; d += 64) o[d] = x[d] * inv;
    }
}

size_t aff_frame_workspace_bytes(int B, int N);
static size_t aff_matched_bytes(int B, int N) {
    const int T = N + 2, Dp = (T + 3) / 4 * 4;
    return align_up((size_t)B * T * Dp * sizeof(float), 256);
}
// matched (B, T, Dp) between the row MLP and the column softmax (two-kernel forms), then the column partials and arrival counters of
// the one-pass form
size_t aff_workspace_bytes(int B, int N) { return aff_matched_bytes(B, N) + aff_frame_workspace_bytes(B, N); }

bool aff_pieces_serves(int D);
int launch_aff_frame16(const shasta_weights* w, const float* packed16, const float* residual, int ld, float* matched, int ldm, float* m1,
                       float* m2, int B, void* ws, hipStream_t st);
int launch_aff_frame(const shasta_weights* w, const float* packed_pieces, const float* residual, int ld, float* matched, int ldm, float* m1,
                     float* m2, int B, void* ws, hipStream_t st);
int launch_aff_pieces(const shasta_weights* w, const float* packed_pieces, const float* residual, int ld, float* matched, int ldm,
                      float* m1, int M, hipStream_t st);

// the piece forms of the six layers serve this call (and, unless SHASTA_OPT_TWO_PASS_AFF, the one-pass kernel with its sibling wait)
static bool aff_piece_form(const shasta_weights* w, int B, int ld, const void* residual, const void* ws) {
    const int M = B * (w->max_obj + 2);
    return (M >= 8192 || (w->options & SHASTA_OPT_ONE_PASS_AFF)) && !(w->options & SHASTA_OPT_F32_AFF) && aff_pieces_serves(w->max_obj + 2) && ld % 4 == 0 &&
           (uintptr_t)residual % 16 == 0 && (uintptr_t)ws % 16 == 0;
}

// Status word of the most recent aff launch on workspace `ws` (aff_workspace_bytes): 0, or bit 0 = a row group's wait for its siblings
// timed out (those rows of matched2 are NaN).  Only the one-pass form has something to report; the other forms give 0.  Synchronises
// on the stream.  `ld`: the residual's leading dimension of that launch.
int aff_status(const shasta_weights* w, int B, int ld, const void* ws, int* status, hipStream_t st) {
    *status = 0;
    if (B == 0 || !aff_piece_form(w, B, ld, nullptr, ws) || (w->options & SHASTA_OPT_TWO_PASS_AFF)) return SHASTA_OK;
    unsigned word = 0;
    const char* src = static_cast<const char*>(ws) + aff_matched_bytes(B, w->max_obj);
    hipError_t e = hipMemcpyAsync(&word, src, sizeof(word), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) {
        set_error("aff_status", e);
        return SHASTA_E_LAUNCH;
    }
    *status = (int)word;
    return SHASTA_OK;
}

int aff_softmax(const shasta_weights* w, const float* packed, int B, const float* residual, int ld, float* m1,
                float* m2, float* matched_out, void* ws, size_t ws_bytes, hipStream_t st) {
    const int N = w->max_obj, T = N + 2, D = N + 2, Dp = (T + 3) / 4 * 4;
    const PackedLayout P(N, w->num_feats, w->feat_dim);
    if (ws_bytes < aff_workspace_bytes(B, N)) {
        set_error_msg("aff_softmax: workspace too small");
        return SHASTA_E_WORKSPACE;
    }
    if (B == 0) return SHASTA_OK;
    char* base = static_cast<char*>(ws);
    float* matched = reinterpret_cast<float*>(base);
    const int M = B * T;
    int rc;
    // 32 rows per workgroup (RG = 2) halve the L2 -> register weight traffic per row, the limiter of this kernel (29 % matrix-
    // pipe utilisation at 16 rows).  With the second hidden buffer aliased into xb two such workgroups fit one CU (80 KB each
    // at N = 500); measured at B = 64: 187 us against 230 us for 16 rows.  16 rows stay the choice when there are too few
    // rows to give every CU two workgroups (small batches: parallelism matters more than traffic) or when 32 rows do not
    // fit twice.
    const size_t lds2 = (size_t)(32 * std::max(Dp + 4, 132) + 32 * 132) * sizeof(float);
    const int rg = (lds2 <= 80 * 1024 && M >= 32 * 512) ? 2 : 1;
    const size_t lds = (size_t)(16 * rg * std::max(Dp + 4, 132) + 16 * rg * 132) * sizeof(float);
    // From 8192 residual rows up the six layers run as exact bf16 piece products (aff_pieces.hip: 2.7 x fewer matrix cycles per
    // fp32 product) for tables up to 512 columns whose rows are 16-byte aligned; SHASTA_OPT_F32_AFF keeps the f32
    // kernel.  Small batches stay on the f32 kernel (16-row workgroups: more parallelism, less latency).
    const bool pieces = aff_piece_form(w, B, ld, residual, ws);
    // ... and, unless SHASTA_OPT_TWO_PASS_AFF asks for the two-kernel form, with both softmaxes in the same pass (aff_frame_kernel):
    // `matched` is then written only when the caller wants it
    if (pieces && !(w->options & SHASTA_OPT_TWO_PASS_AFF)) {
        // SHASTA_OPT_F16X2_AFF: the layers on fp16 pieces (aff_f16.hip: three products per fp32 product instead of six)
        if (w->options & SHASTA_OPT_F16X2_AFF)
            rc = launch_aff_frame16(w, packed + P.aff16, residual, ld, matched_out ? matched : nullptr, Dp, m1, m2, B, base + aff_matched_bytes(B, N), st);
        else
            rc = launch_aff_frame(w, packed + P.affp, residual, ld, matched_out ? matched : nullptr, Dp, m1, m2, B, base + aff_matched_bytes(B, N), st);
        if (rc) return rc;
    } else {
    if (pieces) {
        if ((rc = launch_aff_pieces(w, packed + P.affp, residual, ld, matched, Dp, m1, M, st))) return rc;
    } else if (lds > 160 * 1024) {  // max_obj <= 2046 (check_weights) keeps 16 rows within 140 KB
        set_error_msg("aff_softmax: max_obj too large for the on-chip row tile");
        return SHASTA_E_ARG;
    } else {
        AffArgs fa;
        fa.W[0] = packed + P.aff0;  // zero padded (128, Dp)
        fa.bias[0] = w->aff[0].bias;
        for (int i = 1; i < 6; ++i) {
            fa.W[i] = w->aff[i].weight;
            fa.bias[i] = w->aff[i].bias;
        }
        fa.residual = residual;
        fa.matched = matched;
        fa.m1 = m1;
        fa.M = M;
        fa.T = T;
        fa.N = N;
        fa.D = D;
        fa.Dp = Dp;
        fa.ld = ld;
        fa.ldm = Dp;
        if (rg == 2) {
            (void)hipFuncSetAttribute((const void*)aff_fused_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            hipLaunchKernelGGL(aff_fused_kernel<2>, dim3(cdiv(M, 32)), dim3(64 * AFF_WAVES), lds, st, fa);
        } else {
            if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)aff_fused_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            hipLaunchKernelGGL(aff_fused_kernel<1>, dim3(cdiv(M, 16)), dim3(64 * AFF_WAVES), lds, st, fa);
        }
        if ((rc = check_launch("aff_fused"))) return rc;
    }
    const bool two = N % 2 == 0 && Dp % 2 == 0 && N >= 128 && ((uintptr_t)matched | (uintptr_t)m2) % 8 == 0;
    if (two && T <= 512) hipLaunchKernelGGL(softmax_cols2_kernel<32>, dim3(cdiv(N, 128), B), dim3(1024), 0, st, matched, m2, N, T, Dp);
    else if (two && T <= 1024) hipLaunchKernelGGL(softmax_cols2_kernel<64>, dim3(cdiv(N, 128), B), dim3(1024), 0, st, matched, m2, N, T, Dp);
    else if (T <= 512) hipLaunchKernelGGL(softmax_cols_kernel<32>, dim3(cdiv(N, 64), B), dim3(1024), 0, st, matched, m2, N, T, Dp);
    else if (T <= 1024) hipLaunchKernelGGL(softmax_cols_kernel<64>, dim3(cdiv(N, 64), B), dim3(1024), 0, st, matched, m2, N, T, Dp);
    else hipLaunchKernelGGL(softmax_cols_kernel<128>, dim3(cdiv(N, 64), B), dim3(1024), 0, st, matched, m2, N, T, Dp);
    if ((rc = check_launch("softmax_cols"))) return rc;
    }
    if (matched_out) {
        hipError_t e = hipMemcpy2DAsync(matched_out, (size_t)D * sizeof(float), matched, (size_t)Dp * sizeof(float),
                                        (size_t)D * sizeof(float), (size_t)M, hipMemcpyDeviceToDevice, st);
        if (e != hipSuccess) {
            set_error("aff_softmax: copy matched", e);
            return SHASTA_E_LAUNCH;
        }
    }
    return SHASTA_OK;
}

}  // namespace shasta
