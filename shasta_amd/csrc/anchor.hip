// K3: anchor MLPs.
//  * aug_shape[0..3] (det3d/models/tracker/shasta.py:49-57, applied :241-244):
//      abs(Linear(N*F -> N*F/64) -> ReLU -> Linear(-> F)) on the flattened feature table.
//      [0]=newborn_geom, [1]=fp_geom from the CURRENT features, [2]=dead_trk_geom, [3]=fn_geom from the PREVIOUS.
//      Results are written straight into rows N, N+1 of the (B, N+2, F) tables (the cat of :246-247):
//      prev_feat gets newborn, fp ; feat gets dead_trk, fn.
//  * aug_dets[0..3] (shasta.py:69-76, applied :260-267): Linear(7N -> 7N//32) -> ReLU -> Linear(-> 7), dims abs'd,
//      evaluated on the boxes BEFORE back-projection; then back-projection (:270, in place) and the cat (:273-274).
//
// The first aug_shape layer is the HBM term of the whole path: 4 x (N*F/64) x (N*F) fp32 = 4.1 GB at N=500,F=256.
// It is a weight-streaming skinny GEMM: every weight is read exactly once per call (non-temporal loads, so the
// activation vectors stay in L2), R rows x BT batch items accumulate per lane in registers, lanes walk K with
// 16-byte loads (1 KiB per wave-instruction, fully coalesced), K is split across waves so that >= 4k waves stream
// concurrently, and the split-K partials are summed in a fixed order afterwards (no float atomics -> bitwise
// reproducible).
#include "common.hpp"

namespace shasta {

struct AnchorL1Args {
    const float* W[4];  // aug_shape.i.0.weight  (H, K)
    const float* x[2];  // [0] feat table (mlps 0,1)  [1] prev_feat table (mlps 2,3)
    float* part;        // (KS, B, 4H)
    int H, K, B, KS, Kc, x_batch_stride, groups_per_mlp;
};

template <int BT, int R>
__global__ __launch_bounds__(256) void anchor_l1_kernel(AnchorL1Args a) {
    const int lane = threadIdx.x & 63;
    const int item = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int G = 4 * a.groups_per_mlp;
    if (item >= G * a.KS) return;
    const int ks = item / G, g = item % G;
    const int mlp = g / a.groups_per_mlp, r0 = (g % a.groups_per_mlp) * R;
    const int b0 = blockIdx.y * BT;
    const int kbeg = ks * a.Kc, kend = min(a.K, kbeg + a.Kc);

    const float* wrow[R];
#pragma unroll
    for (int r = 0; r < R; ++r) wrow[r] = a.W[mlp] + (size_t)min(r0 + r, a.H - 1) * a.K;
    const float* xb[BT];
#pragma unroll
    for (int b = 0; b < BT; ++b) xb[b] = a.x[mlp >> 1] + (size_t)min(b0 + b, a.B - 1) * a.x_batch_stride;

    float acc[R][BT];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int b = 0; b < BT; ++b) acc[r][b] = 0.0f;

    for (int k = kbeg + 4 * lane; k < kend; k += 256) {
        f32x4 xv[BT], wv[R];
#pragma unroll
        for (int r = 0; r < R; ++r) wv[r] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(wrow[r] + k));
#pragma unroll
        for (int b = 0; b < BT; ++b) xv[b] = *reinterpret_cast<const f32x4*>(xb[b] + k);
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int b = 0; b < BT; ++b) {
                float s = acc[r][b];
                s = fmaf(wv[r][0], xv[b][0], s);
                s = fmaf(wv[r][1], xv[b][1], s);
                s = fmaf(wv[r][2], xv[b][2], s);
                s = fmaf(wv[r][3], xv[b][3], s);
                acc[r][b] = s;
            }
    }
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int b = 0; b < BT; ++b) {
            const float s = wave_sum(acc[r][b]);
            if (lane == 0 && r0 + r < a.H && b0 + b < a.B)
                a.part[((size_t)ks * a.B + (b0 + b)) * (4 * a.H) + mlp * a.H + r0 + r] = s;
        }
}

// hidden[b][r] = relu(bias1[r] + sum_ks part[ks][b][r]),  r over the 4H concatenated hidden units
__global__ void anchor_hidden_kernel(const float* __restrict__ part, const float* b0, const float* b1,
                                     const float* b2, const float* b3, float* __restrict__ hidden, int H, int B,
                                     int KS, const unsigned* xmax, const unsigned* wmax) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int total = B * 4 * H;
    if (i >= total) return;
    const int r = i % (4 * H);
    const int mlp = r / H, j = r % H;
    const float* bias = mlp == 0 ? b0 : mlp == 1 ? b1 : mlp == 2 ? b2 : b3;
    float s = part[i];
    for (int ks = 1; ks < KS; ++ks) s += part[(size_t)ks * total + i];
    // two-piece fp16 form of the weight stream: undo the row scalings 2^e_b (activations of batch row b) and 2^e_r (weight row), exactly
    if (xmax) s = __builtin_ldexpf(s, -(range_exponent_bits(xmax[(mlp >> 1) * B + i / (4 * H)]) + range_exponent_bits(wmax[r])));
    hidden[i] = relu_nan(s + bias[j]);
}

struct AnchorL2Args {
    const float* W[4];  // aug_shape.i.2.weight (F, H)
    const float* bias[4];
    const float* hidden;  // (B, 4H)
    float* feat;          // (B, N+2, F)
    float* prev_feat;
    int H, F, N, B;
};

// one wave per (batch chunk of 8, mlp, j): out[b] = abs(b2[j] + W2[j,:] . hidden[b, mlp, :]); the weight row is read once
// per 8 batch items
template <int BT>
__global__ __launch_bounds__(256) void anchor_l2_kernel(AnchorL2Args a) {
    const int lane = threadIdx.x & 63;
    const int item = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int nchunk = (a.B + BT - 1) / BT;
    if (item >= nchunk * 4 * a.F) return;
    const int j = item % a.F, mlp = (item / a.F) & 3, b0 = (item / (4 * a.F)) * BT;
    const float* w = a.W[mlp] + (size_t)j * a.H;
    const float* h[BT];
#pragma unroll
    for (int b = 0; b < BT; ++b) h[b] = a.hidden + (size_t)min(b0 + b, a.B - 1) * 4 * a.H + mlp * a.H;
    float s[BT];
#pragma unroll
    for (int b = 0; b < BT; ++b) s[b] = 0.0f;
#pragma unroll 4
    for (int i = lane; i < a.H; i += 64) {
        const float wv = w[i];
#pragma unroll
        for (int b = 0; b < BT; ++b) s[b] = fmaf(wv, h[b][i], s[b]);
    }
    // mlp 0,1 (newborn, fp) -> prev_feat rows N, N+1 ; mlp 2,3 (dead_trk, fn) -> feat rows N, N+1
    float* tab = (mlp < 2) ? a.prev_feat : a.feat;
    const float bias = a.bias[mlp][j];
#pragma unroll
    for (int b = 0; b < BT; ++b) {
        const float v = wave_sum(s[b]);
        if (lane == 0 && b0 + b < a.B) tab[((size_t)(b0 + b) * (a.N + 2) + a.N + (mlp & 1)) * a.F + j] = fabsf(v + bias);
    }
}

struct BoxL1Args {
    const float* W[4];  // aug_dets.i.0.weight (HD, 7N)
    const float* bias[4];
    const float* det;   // (B, N, box_stride), pre back-projection
    const float* prev;
    float* hid;  // (B, 4, HD)
    int HD, N, B, box_stride;
};

// one wave per (chunk of 4 batch items, mlp, 4 hidden units): a 4x4 register tile of hid = relu(b1[u] + W1[u,:] . boxes7_flat),
// lanes across k: 4 weight loads + 4 box loads feed 16 FMAs
__global__ __launch_bounds__(256) void box_l1_kernel(BoxL1Args a) {
    constexpr int BT = 4, UT = 4;
    const int lane = threadIdx.x & 63;
    const int item = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int ugroups = (a.HD + UT - 1) / UT, bchunks = (a.B + BT - 1) / BT;
    if (item >= bchunks * 4 * ugroups) return;
    const int ug = item % ugroups, mlp = (item / ugroups) & 3, b0 = (item / (4 * ugroups)) * BT;
    const int K = 7 * a.N;
    const float* w[UT];
#pragma unroll
    for (int u = 0; u < UT; ++u) w[u] = a.W[mlp] + (size_t)min(ug * UT + u, a.HD - 1) * K;
    const float* xs = (mlp < 2) ? a.det : a.prev;
    const float* x[BT];
#pragma unroll
    for (int b = 0; b < BT; ++b) x[b] = xs + (size_t)min(b0 + b, a.B - 1) * a.N * a.box_stride;
    float s[UT][BT];
#pragma unroll
    for (int u = 0; u < UT; ++u)
#pragma unroll
        for (int b = 0; b < BT; ++b) s[u][b] = 0.0f;
#pragma unroll 2
    for (int k = lane; k < K; k += 64) {
        const int n = k / 7, c = k - 7 * n;
        const size_t off = (size_t)n * a.box_stride + c;
        float wv[UT], xv[BT];
#pragma unroll
        for (int u = 0; u < UT; ++u) wv[u] = w[u][k];
#pragma unroll
        for (int b = 0; b < BT; ++b) xv[b] = x[b][off];
#pragma unroll
        for (int u = 0; u < UT; ++u)
#pragma unroll
            for (int b = 0; b < BT; ++b) s[u][b] = fmaf(wv[u], xv[b], s[u][b]);
    }
#pragma unroll
    for (int u = 0; u < UT; ++u)
#pragma unroll
        for (int b = 0; b < BT; ++b) {
            const float v = wave_sum(s[u][b]);
            const int uu = ug * UT + u;
            if (lane == 0 && uu < a.HD && b0 + b < a.B)
                a.hid[((size_t)(b0 + b) * 4 + mlp) * a.HD + uu] = relu_nan(v + a.bias[mlp][uu]);
        }
}

// Small batches (B <= 4): the register-tile kernel above only launches 28 workgroups per 4 batch items and each lane walks a
// 55-step dependent load chain (27 us at B = 1).  Here one workgroup owns one (batch item, mlp, hidden unit): its four waves
// take a quarter of K each with every load in flight at once, and the quarters are added in a fixed order through LDS.
__global__ __launch_bounds__(256) void box_l1_small_kernel(BoxL1Args a) {
    __shared__ float part[4];
    const int lane = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int u = blockIdx.x % a.HD, mlp = (blockIdx.x / a.HD) & 3, b = blockIdx.x / (4 * a.HD);
    const int K = 7 * a.N;
    const float* w = a.W[mlp] + (size_t)u * K;
    const float* x = ((mlp < 2) ? a.det : a.prev) + (size_t)b * a.N * a.box_stride;
    const int kq = (K + 3) / 4, k0 = q * kq, k1 = min(K, k0 + kq);
    float s0 = 0.0f, s1 = 0.0f;
#pragma unroll 4
    for (int k = k0 + lane; k < k1; k += 128) {
        const int k2 = k + 64;
        const int n = k / 7, c = k - 7 * n;
        s0 = fmaf(w[k], x[(size_t)n * a.box_stride + c], s0);
        if (k2 < k1) {
            const int n2 = k2 / 7, c2 = k2 - 7 * n2;
            s1 = fmaf(w[k2], x[(size_t)n2 * a.box_stride + c2], s1);
        }
    }
    const float v = wave_sum(s0 + s1);
    if (lane == 0) part[q] = v;
    __syncthreads();
    if (threadIdx.x == 0)
        a.hid[((size_t)b * 4 + mlp) * a.HD + u] = relu_nan(((part[0] + part[1]) + (part[2] + part[3])) + a.bias[mlp][u]);
}

// Batches >= 256: the four first layers as GEMMs.  Their input - the 7-vectors of the boxes BEFORE back-projection, flattened -
// is strided (7 of box_stride floats per row) in the caller's tensors: pack it once into (2, B, ldx) rows, ldx = ceil4(7N),
// zero tail (x7[0] = current boxes: newborn / fp, x7[1] = previous boxes: dead_trk / fn).
__global__ __launch_bounds__(256) void box_pack7_kernel(const float* __restrict__ det, const float* __restrict__ prev, float* __restrict__ x7,
                                                        int B, int N, int box_stride, int ldx) {
    const int b = blockIdx.y, which = blockIdx.z;
    const float* src = (which ? prev : det) + (size_t)b * N * box_stride;
    float* dst = x7 + ((size_t)which * B + b) * ldx;
    for (int k = blockIdx.x * 256 + threadIdx.x; k < ldx; k += gridDim.x * 256) {
        const int n = k / 7, c = k - 7 * n;
        dst[k] = k < 7 * N ? src[(size_t)n * box_stride + c] : 0.0f;
    }
}

struct BoxL2Args {
    const float* W[4];  // aug_dets.i.2.weight (7, HD)
    const float* bias[4];
    const float* hid;
    float* det;  // back-projected in place
    const float* prev;
    float* det_tab;  // (B, N+2, 8)
    float* prev_tab;
    float* anchors_out;  // (B, 4, 7) or null: newborn, fp, dead_trk, fn once more, for the caller's own (fresh) tensor
    int HD, N, B, box_stride;
};

// the 4x7 anchor outputs, back-projection (shasta.py:270) and the (N+2, 8) box tables (shasta.py:273-274; column 7 is
// padding and written as 0).
// grid (B, 1 + ceil(N / 256)): block y == 0 computes the anchors, blocks y >= 1 copy / back-project 256 table rows each.
__device__ __forceinline__ void box_l2_tables_role(const BoxL2Args& a, int b, int y, int tid) {
    if (y == 0) {
        // wave `mlp` computes the 7 outputs of aug_dets[mlp].2, lanes across the hidden units
        const int mlp = tid >> 6, lane = tid & 63;
        const float* h = a.hid + ((size_t)b * 4 + mlp) * a.HD;
        // mlp 0,1 (newborn, fp) extend the PREVIOUS boxes; 2,3 (dead_trk, fn) extend the CURRENT ones
        float* tab = ((mlp < 2) ? a.prev_tab : a.det_tab) + ((size_t)b * (a.N + 2) + a.N + (mlp & 1)) * 8;
        for (int c = 0; c < 7; ++c) {
            const float* w = a.W[mlp] + (size_t)c * a.HD;
            float s = 0.0f;
            for (int i = lane; i < a.HD; i += 64) s = fmaf(w[i], h[i], s);
            s = wave_sum(s);
            float v = s + a.bias[mlp][c];
            if (c >= 3 && c < 6) v = fabsf(v);
            if (lane == 0) {
                tab[c] = v;
                if (a.anchors_out) a.anchors_out[((size_t)b * 4 + mlp) * 7 + c] = v;
            }
        }
        if (lane == 0) tab[7] = 0.0f;
        return;
    }
    const int n = (y - 1) * 256 + tid;
    if (n < a.N) {
        float* d = a.det + ((size_t)b * a.N + n) * a.box_stride;
        const float* p = a.prev + ((size_t)b * a.N + n) * a.box_stride;
        const float dt = d[9];
        const float x = d[0] - d[7] * dt, y = d[1] - d[8] * dt;  // separately rounded mul, sub
        d[0] = x;
        d[1] = y;
        float* dr = a.det_tab + ((size_t)b * (a.N + 2) + n) * 8;
        float* pr = a.prev_tab + ((size_t)b * (a.N + 2) + n) * 8;
        dr[0] = x;
        dr[1] = y;
#pragma unroll
        for (int c = 2; c < 7; ++c) dr[c] = d[c];
        dr[7] = 0.0f;
#pragma unroll
        for (int c = 0; c < 7; ++c) pr[c] = p[c];
        pr[7] = 0.0f;
    }
}

__global__ __launch_bounds__(256) void box_l2_tables_kernel(BoxL2Args a) { box_l2_tables_role(a, blockIdx.x, blockIdx.y, threadIdx.x); }

// Small batches (one or a few frame-pairs: every launch costs ~4.5 us of a 115 us step at max_obj 90): everything behind the weight
// stream in ONE launch instead of three (anchor_hidden, anchor_l2, box_l2_tables).  Workgroups [0, nbox): the roles of
// box_l2_tables_kernel (b = blk / ybox, y = blk % ybox).  The others: one (batch item, MLP, 16 output features) of the second
// aug_shape layer each - the workgroup first forms that MLP's hidden activations relu(b1 + sum_ks part) in LDS (fixed order, as
// anchor_hidden_kernel; re-formed by every workgroup of the MLP: KS * H floats from L2), then every wave takes four output features
// with the lane / wave_sum order of anchor_l2_kernel, so the results are bit-identical to the three-launch form.
struct AnchorTailArgs {
    AnchorL2Args l2;
    BoxL2Args box;
    const float* part;   // (KS, B, 4H) split-K partials of the first layer
    const float* b1[4];  // aug_shape.i.0.bias
    int KS, nbox, ybox, jblocks;
};
__global__ __launch_bounds__(256) void anchor_tail_small_kernel(AnchorTailArgs a) {
    extern __shared__ float s_hid[];
    const int tid = threadIdx.x;
    int blk = blockIdx.x;
    if (blk < a.nbox) {
        box_l2_tables_role(a.box, blk / a.ybox, blk % a.ybox, tid);
        return;
    }
    blk -= a.nbox;
    const int H = a.l2.H, F = a.l2.F, B = a.l2.B;
    const int jb = blk % a.jblocks, mlp = (blk / a.jblocks) & 3, b = blk / (4 * a.jblocks);
    const size_t total = (size_t)B * 4 * H;
    const float* bias1 = a.b1[mlp];
    for (int i = tid; i < H; i += 256) {
        const float* p = a.part + (size_t)b * 4 * H + (size_t)mlp * H + i;
        float s = p[0];
        int ks = 1;
        for (; ks + 8 <= a.KS; ks += 8) {  // eight partials in flight, added in order
            float v[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = p[(size_t)(ks + q) * total];
#pragma unroll
            for (int q = 0; q < 8; ++q) s += v[q];
        }
        for (; ks < a.KS; ++ks) s += p[(size_t)ks * total];
        s_hid[i] = relu_nan(s + bias1[i]);
    }
    __syncthreads();
    const int lane = tid & 63, j0 = jb * 16 + (tid >> 6) * 4;
    const float* w[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) w[q] = a.l2.W[mlp] + (size_t)min(j0 + q, F - 1) * H;
    float s[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll 2
    for (int i = lane; i < H; i += 64) {
        const float hv = s_hid[i];
#pragma unroll
        for (int q = 0; q < 4; ++q) s[q] = fmaf(w[q][i], hv, s[q]);
    }
    float* tab = ((mlp < 2) ? a.l2.prev_feat : a.l2.feat) + ((size_t)b * (a.l2.N + 2) + a.l2.N + (mlp & 1)) * F;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float v = wave_sum(s[q]);
        if (lane == 0 && j0 + q < F) tab[j0 + q] = fabsf(v + a.l2.bias[mlp][j0 + q]);
    }
}
constexpr int ANCHOR_TAIL_FUSED_MAX_B = 8;

size_t anchor_split_workspace_bytes(int B, int K);
bool anchor_split_serves(int B, int K, int x_batch_stride);
void launch_split_x(const float* feat, const float* prev_feat, void* xs, int K, int B, int x_batch_stride, int np, const unsigned* xmax, bool precut,
                    hipStream_t st);
void launch_anchor_l1_split(const float* const W[4], const void* xs, float* part, int H, int K, int B, int* ks_out, int np,
                            const unsigned* wmax, const void* wimg, hipStream_t st);
int launch_x_maxima(const float* feat, const float* prev_feat, int K, int B, int x_batch_stride, unsigned* xmax, hipStream_t st);
int launch_w_maxima(const float* const W[4], int H, int K, unsigned* wmax, float* sumabs, hipStream_t st);


// [split-K partials, worst-case KS = 64][hidden (B, 4H)][bf16 activation image of anchor_split.hip, batches > 32 only]
size_t bev_absmax_slot_bytes(int items);

size_t anchor_shape_workspace_bytes(int B, int N, int F) {
    const int H = N * F / 64;
    return align_up((size_t)64 * B * 4 * H * sizeof(float), 256) + align_up((size_t)B * 4 * H * sizeof(float), 256) +
           anchor_split_workspace_bytes(B, N * F) + align_up((size_t)2 * B * sizeof(int), 256) + align_up((size_t)4 * H * sizeof(int), 256) +
           bev_absmax_slot_bytes(2 * B);
}

int launch_anchor_l1_mfma(const float* const W[4], const float* feat, const float* prev_feat, float* part, int H, int K,
                          int B, int x_batch_stride, int* ks_out, hipStream_t st);
int launch_gemm_nt_quad(const float* const A[4], const float* const W[4], const float* const bias[4], float* const C[4], int lda,
                        int ldw, int ldc, int M, int N, int K, int act, hipStream_t st);
bool gemm_nt_quad_direct_ok(const float* const A[4], const float* const W[4], int lda, int ldw, int K);
constexpr int GEMM_DIRECT_MIN_B = 256;  // below, the VALU kernels of the small batches are as fast as the direct skinny GEMM (gemm_f32.hip)

template <int BT, int R>
static void launch_l1(const AnchorL1Args& a, hipStream_t st) {
    const int G = 4 * a.groups_per_mlp;
    dim3 grid(cdiv(G * a.KS, 4), cdiv(a.B, BT));
    hipLaunchKernelGGL((anchor_l1_kernel<BT, R>), grid, dim3(256), 0, st, a);
}

// where anchor_shape leaves relu(W1 x + b1): (B, 4H), MLP-major inside a row; valid until the workspace is re-used
const float* anchor_shape_hidden(const void* ws, int B, int N, int F) {
    const size_t H = (size_t)N * F / 64;
    return reinterpret_cast<const float*>(static_cast<const char*>(ws) + align_up((size_t)64 * B * 4 * H * sizeof(float), 256));
}

// the pre-cut piece image of the first-layer weights (SHASTA_OPT_PRECUT_WEIGHT_STREAM + a companion buffer built with it), or null
static const void* precut_image(const shasta_weights* w) {
    const int K = w->max_obj * w->feat_dim, H = K / 64;
    const bool on = (w->options & SHASTA_OPT_F16X2_WEIGHT_STREAM) && (w->options & SHASTA_OPT_PRECUT_WEIGHT_STREAM) &&
                    !(w->options & SHASTA_OPT_F32_WEIGHT_STREAM) && w->aug_shape_aux && K % 32 == 0 && H > 0;
    return on ? static_cast<const char*>(w->aug_shape_aux) + aux_image_offset((size_t)H) : nullptr;
}
// from how many frame-pairs per call the pre-cut fp16 stream replaces the f32 MFMA / bf16-piece kernels of the smaller batches.
// Measured per step at N=500 (with / without): 2: 0.828 / 0.769 ms, 8: 0.922 / 0.878, 16: 1.019 / 1.006 (the 16x16x4 f32 kernel
// streams 16 rows at 0.60 - 0.65 ms and needs no row maxima / activation image), 32: 1.250 / 1.335, 48: 1.534 / 1.707,
// 64: 1.715 / 1.893, 128: 2.844 / 2.872.
constexpr int PRECUT_MIN_BATCH = 17;
// does anchor_shape take the two-piece fp16 weight stream for this call (and therefore need the activation row maxima)?
bool anchor_shape_uses_xmax(const shasta_weights* w, int B) {
    const int N = w->max_obj, F = w->feat_dim;
    if ((w->options & SHASTA_OPT_F32_WEIGHT_STREAM) || !(w->options & SHASTA_OPT_F16X2_WEIGHT_STREAM)) return false;
    if (precut_image(w) && B >= PRECUT_MIN_BATCH) return true;
    return anchor_split_serves(B, N * F, (N + 2) * F) && B > 64;
}
// where anchor_shape keeps those maxima ([2 frames: feat, prev_feat][B] float bit patterns): a producer of the tables that already
// knows them (the gather of shasta_affinity_from_bev_f32) writes them here and passes xmax_ready
unsigned* anchor_shape_xmax(void* ws, int B, int N, int F) {
    const size_t H = (size_t)N * F / 64;
    char* hidden = static_cast<char*>(ws) + align_up((size_t)64 * B * 4 * H * sizeof(float), 256);
    char* xs = hidden + align_up((size_t)B * 4 * H * sizeof(float), 256);
    return reinterpret_cast<unsigned*>(xs + anchor_split_workspace_bytes(B, N * F));
}
// ... and the scratch lines the gather posts into before they are reduced to those maxima (bev_gather.hip): [2 B][slots][128 B]
unsigned* anchor_shape_xmax_slots(void* ws, int B, int N, int F) {
    const size_t H = (size_t)N * F / 64;
    return reinterpret_cast<unsigned*>(reinterpret_cast<char*>(anchor_shape_xmax(ws, B, N, F)) + align_up((size_t)2 * B * sizeof(int), 256) +
                                       align_up((size_t)4 * H * sizeof(int), 256));
}

// tail: the deferred box roles of anchor_boxes_impl; when given (small batches) everything behind the first layer goes into ONE launch
static int anchor_shape_impl(const shasta_weights* w, int B, float* feat, float* prev_feat, void* ws, size_t ws_bytes, hipStream_t st,
                             hipEvent_t ev0, hipEvent_t ev1, const unsigned* wmax, bool xmax_ready, const BoxL2Args* tail) {
    const int N = w->max_obj, F = w->feat_dim;
    const int K = N * F, H = K / 64;
    if (ws_bytes < anchor_shape_workspace_bytes(B, N, F)) {
        set_error_msg("anchor_shape: workspace too small");
        return SHASTA_E_WORKSPACE;
    }
    if (B == 0) return SHASTA_OK;
    constexpr int R = 8;
    AnchorL1Args a;
    for (int i = 0; i < 4; ++i) a.W[i] = w->aug_shape[i][0].weight;
    a.x[0] = feat;
    a.x[1] = prev_feat;
    a.H = H;
    a.K = K;
    a.B = B;
    a.x_batch_stride = (N + 2) * F;
    a.groups_per_mlp = cdiv(H, R);
    const int G = 4 * a.groups_per_mlp;
    // enough waves to fill 256 CUs x 16 streaming waves, K chunks of >= 1024 floats, multiples of 256
    int ks = max(1, min(min(64, cdiv(4096, G)), cdiv(K, 1024)));
    a.Kc = cdiv(cdiv(K, ks), 256) * 256;
    a.KS = cdiv(K, a.Kc);
    float* part = static_cast<float*>(ws);
    float* hidden = reinterpret_cast<float*>(static_cast<char*>(ws) + align_up((size_t)64 * B * 4 * H * sizeof(float), 256));
    a.part = part;
    // B == 1: VALU weight-streaming GEMV (nothing to amortise; measured 6.8 TB/s).  2 <= B <= 32: the f32 matrix-core kernel
    // streams every weight once per 16 / 32 batch items at HBM speed (anchor_mfma.hip).  B > 32: the same fp32 arithmetic as
    // exact bf16 piece products, 64 / 128 items per weight pass (anchor_split.hip); with SHASTA_OPT_F32_WEIGHT_STREAM in
    // w->options the f32 MFMA kernel serves every B >= 2 (64 items per pass, matrix-pipe bound).  K = N*F is a multiple of 64.
    const bool force_f32 = (w->options & SHASTA_OPT_F32_WEIGHT_STREAM) != 0;
    void* xs = reinterpret_cast<char*>(hidden) + align_up((size_t)B * 4 * H * sizeof(float), 256);
    // the two-piece fp16 form when asked for: above 64 frame-pairs (up to 64 the three-piece bf16 kernel's 64-row pass is the faster one
    // while the weights are cut inside the kernel), and from PRECUT_MIN_BATCH frame-pairs when the pre-cut image is there: with nothing
    // but DMA, LDS reads and 6 / 12 MFMAs per 4 KB weight tile the small batches run at the speed of the stream as well
    const void* wimg = B >= PRECUT_MIN_BATCH ? precut_image(w) : nullptr;
    const bool f16x2 = anchor_shape_uses_xmax(w, B);
    const bool split = !force_f32 && (anchor_split_serves(B, K, a.x_batch_stride) || f16x2);
    const int np = f16x2 ? 2 : 3;
    unsigned* xmax = reinterpret_cast<unsigned*>(static_cast<char*>(xs) + anchor_split_workspace_bytes(B, K));
    if (f16x2) {
        int rc0;
        if (!wmax) {  // stage entry point without a packed buffer: one extra pass over the weights into the workspace
            unsigned* wm = reinterpret_cast<unsigned*>(reinterpret_cast<char*>(xmax) + align_up((size_t)2 * B * sizeof(int), 256));
            if ((rc0 = launch_w_maxima(a.W, H, K, wm, nullptr, st))) return rc0;
            wmax = wm;
        }
        if (!xmax_ready && (rc0 = launch_x_maxima(feat, prev_feat, K, B, a.x_batch_stride, xmax, st))) return rc0;
    }
    if (split) launch_split_x(feat, prev_feat, xs, K, B, a.x_batch_stride, np, xmax, f16x2 && wimg, st);
    if (ev0) (void)hipEventRecord(ev0, st);
    if (B == 1) launch_l1<1, R>(a, st);
    else if (split) launch_anchor_l1_split(a.W, xs, part, H, K, B, &a.KS, np, wmax, f16x2 ? wimg : nullptr, st);
    else launch_anchor_l1_mfma(a.W, feat, prev_feat, part, H, K, B, a.x_batch_stride, &a.KS, st);
    if (ev1) (void)hipEventRecord(ev1, st);
    int rc = check_launch("anchor_l1");
    if (rc) return rc;
    const int total = B * 4 * H;
    if (tail) {
        if (f16x2) {
            set_error_msg("anchor_shape: the fused tail serves the f32 weight-stream kernels only");
            return SHASTA_E_ARG;
        }
        AnchorTailArgs t;
        for (int i = 0; i < 4; ++i) {
            t.l2.W[i] = w->aug_shape[i][1].weight;
            t.l2.bias[i] = w->aug_shape[i][1].bias;
            t.b1[i] = w->aug_shape[i][0].bias;
        }
        t.l2.hidden = nullptr;
        t.l2.feat = feat;
        t.l2.prev_feat = prev_feat;
        t.l2.H = H;
        t.l2.F = F;
        t.l2.N = N;
        t.l2.B = B;
        t.box = *tail;
        t.part = part;
        t.KS = a.KS;
        t.ybox = 1 + cdiv(N, 256);
        t.nbox = B * t.ybox;
        t.jblocks = cdiv(F, 16);
        hipLaunchKernelGGL(anchor_tail_small_kernel, dim3(t.nbox + B * 4 * t.jblocks), dim3(256), (size_t)H * sizeof(float), st, t);
        return check_launch("anchor_tail_small");
    }
    hipLaunchKernelGGL(anchor_hidden_kernel, dim3(cdiv(total, 256)), dim3(256), 0, st, part,
                       w->aug_shape[0][0].bias, w->aug_shape[1][0].bias, w->aug_shape[2][0].bias,
                       w->aug_shape[3][0].bias, hidden, H, B, a.KS, f16x2 ? xmax : nullptr, wmax);
    rc = check_launch("anchor_hidden");
    if (rc) return rc;
    AnchorL2Args l2;
    for (int i = 0; i < 4; ++i) {
        l2.W[i] = w->aug_shape[i][1].weight;
        l2.bias[i] = w->aug_shape[i][1].bias;
    }
    l2.hidden = hidden;
    l2.feat = feat;
    l2.prev_feat = prev_feat;
    l2.H = H;
    l2.F = F;
    l2.N = N;
    l2.B = B;
    {
        // a plain GEMM per MLP: (B, H) x (F, H)^T on the matrix cores, |.| and the table row as the epilogue's target (row N + (i & 1) of
        // prev_feat for newborn / fp, of feat for dead_trk / fn; leading dimension = one batch item).  From 256 frame-pairs always (the
        // 64 x 64 tiles take 80 - 100 us whatever the batch, the direct form 50 us, the VALU kernel below 39 us at 64 frame-pairs, 238 at 512).
        const float* A[4];
        const float* W2[4];
        const float* b2[4];
        float* C[4];
        for (int i = 0; i < 4; ++i) {
            A[i] = hidden + (size_t)i * H;
            W2[i] = l2.W[i];
            b2[i] = l2.bias[i];
            C[i] = ((i < 2) ? prev_feat : feat) + (size_t)(N + (i & 1)) * F;
        }
        if (B >= 256 || (B >= GEMM_DIRECT_MIN_B && gemm_nt_quad_direct_ok(A, W2, 4 * H, H, H)))
            return launch_gemm_nt_quad(A, W2, b2, C, 4 * H, H, (N + 2) * F, B, F, H, 2, st);
    }
    // small batches: no duplicated activation loads (BT = batch items that share one weight row read)
    if (B == 1) hipLaunchKernelGGL(anchor_l2_kernel<1>, dim3(cdiv(B * 4 * F, 4)), dim3(256), 0, st, l2);
    else if (B == 2) hipLaunchKernelGGL(anchor_l2_kernel<2>, dim3(cdiv(cdiv(B, 2) * 4 * F, 4)), dim3(256), 0, st, l2);
    else if (B <= 4) hipLaunchKernelGGL(anchor_l2_kernel<4>, dim3(cdiv(cdiv(B, 4) * 4 * F, 4)), dim3(256), 0, st, l2);
    else hipLaunchKernelGGL(anchor_l2_kernel<8>, dim3(cdiv(cdiv(B, 8) * 4 * F, 4)), dim3(256), 0, st, l2);
    return check_launch("anchor_l2");
}
int anchor_shape(const shasta_weights* w, int B, float* feat, float* prev_feat, void* ws, size_t ws_bytes,
                 hipStream_t st, hipEvent_t ev0, hipEvent_t ev1, const unsigned* wmax, bool xmax_ready) {
    return anchor_shape_impl(w, B, feat, prev_feat, ws, ws_bytes, st, ev0, ev1, wmax, xmax_ready, nullptr);
}

// [hidden (B, 4, HD)][packed box rows (2, B, ceil4(7N)) for the GEMM form]
size_t anchor_boxes_workspace_bytes(int B, int N) {
    return align_up((size_t)B * 4 * max(1, 7 * N / 32) * sizeof(float), 256) + align_up((size_t)2 * B * ((7 * N + 3) / 4 * 4) * sizeof(float), 256);
}

// defer: when given, the second layers / back-projection / tables are not launched but described there (anchor_shape_impl's tail)
static int anchor_boxes_impl(const shasta_weights* w, int B, float* det_boxes, const float* prev_det_boxes, int box_stride,
                             float* det_tab, float* prev_tab, float* hid_ws, hipStream_t st, float* anchors_out, BoxL2Args* defer) {
    const int N = w->max_obj, HD = 7 * N / 32;
    if (B == 0) return SHASTA_OK;
    if (HD > 0) {
        BoxL1Args a;
        for (int i = 0; i < 4; ++i) {
            a.W[i] = w->aug_dets[i][0].weight;
            a.bias[i] = w->aug_dets[i][0].bias;
        }
        a.det = det_boxes;
        a.prev = prev_det_boxes;
        a.hid = hid_ws;
        a.HD = HD;
        a.N = N;
        a.B = B;
        a.box_stride = box_stride;
        const int ldx = (7 * N + 3) / 4 * 4;
        const float* W1d[4] = {a.W[0], a.W[1], a.W[2], a.W[3]};
        // (the packed box rows x7 are 16-byte aligned rows of ldx floats; the weights' rows are when 7 N % 4 == 0)
        const bool direct = B >= GEMM_DIRECT_MIN_B && gemm_nt_quad_direct_ok(W1d, W1d, ldx, 7 * N, 7 * N);
        if (B >= 256 || direct) {
            float* x7 = reinterpret_cast<float*>(reinterpret_cast<char*>(hid_ws) + align_up((size_t)B * 4 * HD * sizeof(float), 256));
            hipLaunchKernelGGL(box_pack7_kernel, dim3(cdiv(ldx, 256), B, 2), dim3(256), 0, st, det_boxes, prev_det_boxes, x7, B, N, box_stride, ldx);
            int rc = check_launch("box_pack7");
            if (rc) return rc;
            const float* A[4] = {x7, x7, x7 + (size_t)B * ldx, x7 + (size_t)B * ldx};
            const float* W1[4];
            const float* b1[4];
            float* C[4];
            for (int i = 0; i < 4; ++i) {
                W1[i] = a.W[i];
                b1[i] = a.bias[i];
                C[i] = hid_ws + (size_t)i * HD;  // hid (B, 4, HD): MLP i at column block i, leading dimension 4 HD
            }
            if ((rc = launch_gemm_nt_quad(A, W1, b1, C, ldx, 7 * N, 4 * HD, B, HD, 7 * N, 1, st))) return rc;
        } else if (B <= 4) hipLaunchKernelGGL(box_l1_small_kernel, dim3(B * 4 * HD), dim3(256), 0, st, a);
        else hipLaunchKernelGGL(box_l1_kernel, dim3(cdiv(cdiv(B, 4) * 4 * cdiv(HD, 4), 4)), dim3(256), 0, st, a);
        int rc = check_launch("box_l1");
        if (rc) return rc;
    }
    BoxL2Args b2;
    for (int i = 0; i < 4; ++i) {
        b2.W[i] = w->aug_dets[i][1].weight;
        b2.bias[i] = w->aug_dets[i][1].bias;
    }
    b2.hid = hid_ws;
    b2.det = det_boxes;
    b2.prev = prev_det_boxes;
    b2.det_tab = det_tab;
    b2.prev_tab = prev_tab;
    b2.anchors_out = anchors_out;
    b2.HD = HD;
    b2.N = N;
    b2.B = B;
    b2.box_stride = box_stride;
    if (defer) {
        *defer = b2;
        return SHASTA_OK;
    }
    hipLaunchKernelGGL(box_l2_tables_kernel, dim3(B, 1 + cdiv(N, 256)), dim3(256), 0, st, b2);
    return check_launch("box_l2_tables");
}
int anchor_boxes(const shasta_weights* w, int B, float* det_boxes, const float* prev_det_boxes, int box_stride,
                 float* det_tab, float* prev_tab, float* hid_ws, hipStream_t st, float* anchors_out) {
    return anchor_boxes_impl(w, B, det_boxes, prev_det_boxes, box_stride, det_tab, prev_tab, hid_ws, st, anchors_out, nullptr);
}

// Both anchor stages of a small batch (forward_impl): the first box layers (they read the boxes before the back-projection), the
// first aug_shape layer, then anchor_tail_small_kernel.  ws: anchor_shape's workspace followed by anchor_boxes'.
bool anchor_stage_fused_serves(const shasta_weights* w, int B) {
    return B >= 1 && B <= ANCHOR_TAIL_FUSED_MAX_B && (size_t)w->max_obj * w->feat_dim / 64 * sizeof(float) <= 48 * 1024;
}
int anchor_stage_fused(const shasta_weights* w, int B, float* feat, float* prev_feat, float* det_boxes, const float* prev_det_boxes,
                       int box_stride, float* det_tab, float* prev_tab, void* ws, size_t ws_bytes, hipStream_t st, hipEvent_t ev0,
                       hipEvent_t ev1, const unsigned* wmax, bool xmax_ready, float* anchors_out) {
    const size_t shape_bytes = anchor_shape_workspace_bytes(B, w->max_obj, w->feat_dim);
    if (ws_bytes < shape_bytes + anchor_boxes_workspace_bytes(B, w->max_obj)) {
        set_error_msg("anchor stages: workspace too small");
        return SHASTA_E_WORKSPACE;
    }
    BoxL2Args tail;
    int rc = anchor_boxes_impl(w, B, det_boxes, prev_det_boxes, box_stride, det_tab, prev_tab,
                               reinterpret_cast<float*>(static_cast<char*>(ws) + shape_bytes), st, anchors_out, &tail);
    if (rc) return rc;
    return anchor_shape_impl(w, B, feat, prev_feat, ws, shape_bytes, st, ev0, ev1, wmax, xmax_ready, &tail);
}

}  // namespace shasta
