// K3a for more than 32 frame-pairs per step: first aug_shape layer (det3d/models/tracker/shasta.py:54, applied :241-244):
//   part[ks][b][n] = sum_{k in chunk ks} W[n][k] * x[b][k]       W: 4 x (N*F/64, N*F) fp32, 4.1 GB at N=500,F=256
// The f32 MFMA kernel of anchor_mfma.hip is matrix-pipe bound from 64 batch rows per pass (2048 SIMD cycles per 4 KB weight
// tile against ~1300 that HBM needs).  This kernel keeps fp32 ARITHMETIC but runs it on the bf16 matrix path, which is 16 x
// faster per product: every fp32 operand is cut - exactly, by truncation - into three bf16 pieces
//   a = a_hi + a_mid + a_lo      (8 + 8 + 8 significand bits; a_lo is exact because the remainder has at most 8 bits left)
// and  w * x  is accumulated (in the fp32 accumulator of v_mfma_f32_32x32x16_bf16) as the six piece products of weight
// 2^0 .. 2^-16:  w_lo x_hi + w_hi x_lo + w_mid x_mid + w_mid x_hi + w_hi x_mid + w_hi x_hi.
// Each bf16 x bf16 product is exact in fp32; the three dropped products are below 2^-24 |w x|, i.e. below the rounding error
// of the fp32 FMA they replace (tests/test_hip_parity.py compares both kernels with the float64 oracle: same error level).
// Cost per 4 KB weight tile and 64 batch rows: 24 MFMA x 32 = 768 cycles instead of 2048 -> the kernel is HBM-bound again.
//
// Data path.  The weights stay fp32 in HBM (they are the reference's checkpoint tensors) and are streamed exactly once per
// pass of 64 / 128 batch rows: LDS-DMA into a [32 rows][8 x 16 B] image per wave (source-side swizzle, see anchor_mfma.hip),
// ds_read_b128 -> registers, cut into pieces on the VALU between the MFMAs (the bf16 MFMA leaves 24 of its 32 cycles to
// other vector instructions; 16 weights per lane and tile = 88 instructions).  The activations are cut ONCE by
// split_x_kernel into a bf16 image that is already in MFMA-fragment order (1 KB = one operand fragment of one wave), so the
// x tile of a workgroup is a linear 12 / 24 KB copy and every fragment read is a conflict-free linear ds_read_b128.
// The four waves of a workgroup (4 x 32 weight rows) share one x tile: x traffic from L2 is 0.75 x the weight bytes instead
// of 2 x, and the ring is 5 (64 rows) or 4 (128 rows) slots deep = 64 / 48 KB of weights in flight per CU.
// Sync: one s_barrier per tile.  Wave w waits (counted vmcnt) for its own share of tile t+1, then the barrier certifies
// (a) every share of x tile t+1 has landed and (b) every wave has finished reading tile t out of its slot, which is then
// refilled with tile t+NS.  Workgroups that share an x stream (same K chunk, same frame) sit on one XCD (blockIdx.x % 8).
#include "common.hpp"

// the LDS-DMA asm below names m0 in its clobber list on purpose (it writes it)
#pragma clang diagnostic ignored "-Winline-asm"
#include <stdlib.h>

#include <type_traits>

namespace shasta {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// a = h + m + l exactly, each with at most 8 significand bits (bf16-representable by truncation)
__device__ __forceinline__ void split3(float a, float& h, float& m, float& l) {
    h = __uint_as_float(__float_as_uint(a) & 0xffff0000u);
    const float r = a - h;
    m = __uint_as_float(__float_as_uint(r) & 0xffff0000u);
    l = r - m;
}
// {bf16(even), bf16(odd)} of two floats whose low 16 bits are not needed
__device__ __forceinline__ uint32_t pack_top(float even, float odd) {
    return __builtin_amdgcn_perm(__float_as_uint(odd), __float_as_uint(even), 0x07060302u);
}

// Largest magnitude of every row: grid (chunks, rows, sources), one atomicMax per workgroup on the bit pattern of a non-negative
// float (order-preserving as unsigned).  `out` ([source][rows] uint32) must be zero on entry.  Rows are 16-byte aligned,
// len % 4 == 0.  The consumers turn the maximum into the row's range exponent themselves (range_exponent_bits).
struct RowMaxArgs {
    const float* base[4];
    long stride;     // floats between consecutive rows of one source
    int rows, len;   // rows per source, floats per row
    unsigned* out;   // [source][rows]
    float* sumabs;   // [source][rows] sum of |x| of every row (zero on entry), or null
};
__global__ __launch_bounds__(256) void row_max_kernel(RowMaxArgs a) {
    __shared__ float red[4];
    const int row = blockIdx.y, src = blockIdx.z;
    const int n4 = a.len / 4, per = (n4 + gridDim.x - 1) / gridDim.x;
    const int beg = blockIdx.x * per, end = min(n4, beg + per);
    const f32x4* p = reinterpret_cast<const f32x4*>(a.base[src] + (size_t)row * a.stride);
    float m = 0.0f, sa = 0.0f;
    for (int i = beg + threadIdx.x; i < end; i += 256) {
        const f32x4 v = __builtin_nontemporal_load(p + i);
        m = fmaxf(fmaxf(m, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
        sa += (fabsf(v[0]) + fabsf(v[1])) + (fabsf(v[2]) + fabsf(v[3]));
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) atomicMax(a.out + src * a.rows + row, __float_as_uint(fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]))));
    if (a.sumabs) {  // a statistic for the range guard (aux_ratio_kernel), not an operand: float atomics in any order are fine
        sa = wave_sum(sa);
        if ((threadIdx.x & 63) == 0) atomicAdd(a.sumabs + src * a.rows + row, sa);
    }
}

// largest (row maximum) / (mean |w| of the row's other entries) over all 4 H rows -> summary[0], its row -> summary[1]; one workgroup
__global__ __launch_bounds__(256) void aux_ratio_kernel(const unsigned* __restrict__ wmax, const float* __restrict__ sumabs, int rows, int K,
                                                        float* __restrict__ summary) {
    __shared__ float rbest[4];
    __shared__ int ibest[4];
    float best = 0.0f;
    int where = 0;
    for (int r = threadIdx.x; r < rows; r += 256) {
        // the row's largest entry against the mean magnitude of the OTHER K - 1 (with the maximum inside the mean a single huge entry
        // would hide behind itself); an all-zero row has nothing to lose; a row whose other entries vanish against the maximum in the
        // fp32 sum (or are exactly zero) is refused
        const float mx = __uint_as_float(wmax[r]), sa = sumabs[r], rest = sa - mx;
        const float ratio = !(mx > 0.0f) || K < 2 ? (mx == mx ? 0.0f : mx) : rest > sa * 1e-6f ? mx * (float)(K - 1) / rest : INFINITY;
        if (!(ratio <= best)) {  // a NaN ratio (non-finite weights) wins
            best = ratio;
            where = r;
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const float ob = __shfl_xor(best, off, 64);
        const int ow = __shfl_xor(where, off, 64);
        if (!(ob <= best)) {
            best = ob;
            where = ow;
        }
    }
    if ((threadIdx.x & 63) == 0) {
        rbest[threadIdx.x >> 6] = best;
        ibest[threadIdx.x >> 6] = where;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w)
            if (!(rbest[w] <= best)) {
                best = rbest[w];
                where = ibest[w];
            }
        summary[0] = best;
        summary[1] = (float)where;
    }
}

static int row_max_chunks(int rows, int sources, int len) {  // enough workgroups to fill the chip, at least 16 KB per workgroup
    int c = 1;
    while (c < 64 && rows * sources * c < 2048 && len / (c * 2) >= 4096) c *= 2;
    return c;
}
// maxima of the batch rows of both feature tables (frame 0: feat, frame 1: prev_feat): xmax[2][B]
int launch_x_maxima(const float* feat, const float* prev_feat, int K, int B, int x_batch_stride, unsigned* xmax, hipStream_t st) {
    if (hipMemsetAsync(xmax, 0, (size_t)2 * B * sizeof(unsigned), st) != hipSuccess) return SHASTA_E_LAUNCH;
    RowMaxArgs r;
    r.base[0] = feat;
    r.base[1] = prev_feat;
    r.base[2] = r.base[3] = nullptr;
    r.stride = x_batch_stride;
    r.rows = B;
    r.len = K;
    r.out = xmax;
    r.sumabs = nullptr;
    hipLaunchKernelGGL(row_max_kernel, dim3(row_max_chunks(B, 2, K), B, 2), dim3(256), 0, st, r);
    return check_launch("row_max (activations)");
}
// maxima of the 4 x H weight rows of the aug_shape first layers (pack time): wmax[4][H]
// sumabs (optional): [4][H] floats + 16 floats of summary behind them (the stats section of the companion buffer, common.hpp)
int launch_w_maxima(const float* const W[4], int H, int K, unsigned* wmax, float* sumabs, hipStream_t st) {
    if (hipMemsetAsync(wmax, 0, (size_t)4 * H * sizeof(unsigned), st) != hipSuccess) return SHASTA_E_LAUNCH;
    if (sumabs && hipMemsetAsync(sumabs, 0, ((size_t)4 * H + 16) * sizeof(float), st) != hipSuccess) return SHASTA_E_LAUNCH;
    RowMaxArgs r;
    for (int i = 0; i < 4; ++i) r.base[i] = W[i];
    r.stride = K;
    r.rows = H;
    r.len = K;
    r.out = wmax;
    r.sumabs = sumabs;
    hipLaunchKernelGGL(row_max_kernel, dim3(row_max_chunks(H, 4, K), H, 4), dim3(256), 0, st, r);
    int rc = check_launch("row_max (weights)");
    if (rc || !sumabs) return rc;
    hipLaunchKernelGGL(aux_ratio_kernel, dim3(1), dim3(256), 0, st, wmax, sumabs, 4 * H, K, sumabs + (size_t)4 * H);
    return check_launch("aux_ratio");
}

// Two-piece fp16 form (NP = 2): a * 2^e = h + l + err with h = fp16(a 2^e) and l = fp16(a 2^e - h), both rounded to nearest:
// |err| <= 2^-24 |a 2^e|, half an ulp of the fp32 value itself.  2^e is an exact power of two - one per batch row of the
// activations, one per weight row - that puts the row's largest magnitude into (2^13, 2^14] (range_exponent); the result row /
// column is scaled back by 2^-(e_b + e_m) in anchor_hidden_kernel, exactly.  w * x is then the THREE products w_l x_h + w_h x_l + w_h x_h (each exact in the fp32 accumulator of
// v_mfma_f32_32x32x16_f16; the dropped w_l x_l is below 2^-24 |w x|), half the matrix work of the six bf16 piece products.
__device__ __forceinline__ void split2h(float a, _Float16& h, _Float16& l) {
    h = (_Float16)a;
    l = (_Float16)(a - (float)h);
}
__device__ __forceinline__ uint32_t pack_h2(_Float16 even, _Float16 odd) {
    const f16x2 v = {even, odd};
    return __builtin_bit_cast(uint32_t, v);
}

// Experiment switch (-DSHASTA_L1_SHAPE16=1, tools/build_variant.py): the pre-cut fp16 form on v_mfma_f32_16x16x32_f16 instead of
// v_mfma_f32_32x32x16_f16 - the same matrix cycles per tile (96 x 16 instead of 48 x 32 at 256 items per pass) and the same operand
// bytes; tried because the chip can hold a higher clock on the 16 x 16 shape (MI355X_MICROARCH.md, DVFS give-back item 7).  Measured
// (DESIGN.md section 4, K3a round 3): bit-for-bit the same sums are not expected (k = 32 per product instead of 2 x 16), results equal
// within the pins; 1024 frame-pairs per step: weight stream 5.33 ms against 5.02 ms, and the pair kernel behind it 8.9 against 8.3 ms,
// at 1170 W instead of 1280 W - slower at less power, so the 32 x 32 form stays.
// Fragments of that shape: lane (i = lane & 15, kb = lane >> 4) holds row / item i, k = 8 kb .. 8 kb + 7 of a 32-wide k tile.
#ifndef SHASTA_L1_SHAPE16
#define SHASTA_L1_SHAPE16 0
#endif
constexpr bool kPrecutShape16 = SHASTA_L1_SHAPE16 != 0;

struct SplitXArgs {
    const float* x[2];
    uint32_t* xs;  // [2 frames][NBLK][KT][XT][2 k-steps (shape16: 16-item halves)][NP pieces][64 lanes][8 bf16 / fp16]
    int B, KT, NBLK, XT, x_batch_stride, NP;
    int shape16;   // NP = 2 only: fragments for v_mfma_f32_16x16x32_f16 (the pre-cut weight stream)
    const unsigned* xmax;  // NP = 2: [2 frames][B] row maxima (row_max_kernel): batch row b of frame f is cut as x * 2^e, e = its range exponent
};

// grid (cdiv(KT, 8), NBLK * XT, 2): 32 batch rows x 256 k per block, transposed through LDS so that both the fp32 reads
// (1 KB per row) and the fragment-order writes (1 KB per fragment) are whole lines.
__global__ __launch_bounds__(256) void split_x_kernel(SplitXArgs a) {
    __shared__ __attribute__((aligned(16))) float tile[32][264];
    const int src = blockIdx.z, bblk = blockIdx.y / a.XT, u = blockIdx.y % a.XT;
    const int kt0 = blockIdx.x * 8, nkt = min(8, a.KT - kt0);
    const float* x = a.x[src];
    const int brow0 = (bblk * a.XT + u) * 32;
    for (int i = threadIdx.x; i < 32 * 64; i += 256) {
        const int r = i >> 6, c4 = i & 63, b = brow0 + r;
        f32x4 v = {0.0f, 0.0f, 0.0f, 0.0f};
        if (b < a.B && (c4 >> 3) < nkt)
            v = *reinterpret_cast<const f32x4*>(x + (size_t)b * a.x_batch_stride + (size_t)kt0 * 32 + c4 * 4);
        *reinterpret_cast<f32x4*>(&tile[r][c4 * 4]) = v;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 1024; i += 256) {
        const int lane = i & 63, s = (i >> 6) & 1, ktl = i >> 7;
        if (ktl >= nkt) continue;
        // 32x32x16 fragments: (item r, k half h) of k step s; 16x16x32: (item 16 s + i, k block kb) of the whole tile
        const int r = a.shape16 ? 16 * s + (lane & 15) : lane & 31, h = lane >> 5;
        const float* p = a.shape16 ? &tile[r][ktl * 32 + 8 * (lane >> 4)] : &tile[r][ktl * 32 + 16 * s + 8 * h];
        const f32x4 v0 = *reinterpret_cast<const f32x4*>(p), v1 = *reinterpret_cast<const f32x4*>(p + 4);
        const float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
        if (a.NP == 2) {
            const int e = range_exponent_bits(a.xmax[src * a.B + min(brow0 + r, a.B - 1)]);
            u32x4 hi, lo;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                _Float16 h0, l0, h1, l1;
                split2h(__builtin_ldexpf(v[2 * j], e), h0, l0);
                split2h(__builtin_ldexpf(v[2 * j + 1], e), h1, l1);
                hi[j] = pack_h2(h0, h1);
                lo[j] = pack_h2(l0, l1);
            }
            const size_t frag0 = ((((size_t)src * a.NBLK + bblk) * a.KT + kt0 + ktl) * (4 * a.XT)) + (size_t)(u * 2 + s) * 2;
            u32x4* o = reinterpret_cast<u32x4*>(a.xs) + frag0 * 64 + lane;
            o[0] = hi;
            o[64] = lo;
            continue;
        }
        u32x4 hi, mid, lo;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float h0, m0, l0, h1, m1, l1;
            split3(v[2 * j], h0, m0, l0);
            split3(v[2 * j + 1], h1, m1, l1);
            hi[j] = pack_top(h0, h1);
            mid[j] = pack_top(m0, m1);
            lo[j] = pack_top(l0, l1);
        }
        const size_t frag0 = ((((size_t)src * a.NBLK + bblk) * a.KT + kt0 + ktl) * (6 * a.XT)) + (size_t)(u * 2 + s) * 3;
        u32x4* o = reinterpret_cast<u32x4*>(a.xs) + frag0 * 64 + lane;
        o[0] = hi;
        o[64] = mid;
        o[128] = lo;
    }
}

// ---- pre-cut weight image (SHASTA_OPT_PRECUT_WEIGHT_STREAM) ---------------------------------------------------------------------
// The fp16 form cuts every fp32 weight into its two pieces on the VALU, between the MFMAs, once per BATCH BLOCK (twice per step at
// 512 frame-pairs).  The image holds the pieces instead: per (MLP, 32-row group, 32-wide k tile) one 4 KB block =
// [k step 2 (shape16: 16-row half)][piece 2][lane 64][8 fp16] - exactly the four A-operand fragments of a wave, in lane order - i.e. the same 4 bytes per
// weight on the stream, +4.1 GB resident next to the fp32 checkpoint tensors at N=500, and nothing but LDS-DMA, ds_read and MFMA
// in the loop.  Built by shasta_aug_shape_aux_f32 behind the row maxima; rebuilt when the weights change, like them.
struct PrecutArgs {
    const float* W[4];
    const unsigned* wmax;  // [4][H]
    uint32_t* img;
    int H, K, KT, G;       // G = 32-row groups per MLP
};
// grid (cdiv(KT, 8), G, 4), 256 threads: 32 rows x 256 k through LDS (whole 1 KB row segments in, whole 4 KB blocks out)
__global__ __launch_bounds__(256) void precut_weights_kernel(PrecutArgs a) {
    __shared__ __attribute__((aligned(16))) float tile[32][260];
    const int mlp = blockIdx.z, g = blockIdx.y, kt0 = blockIdx.x * 8, nkt = min(8, a.KT - kt0);
    const float* W = a.W[mlp];
    for (int i = threadIdx.x; i < 32 * 64; i += 256) {
        const int r = i >> 6, c4 = i & 63, row = g * 32 + r;
        f32x4 v = {0.0f, 0.0f, 0.0f, 0.0f};
        if (row < a.H && (c4 >> 3) < nkt) v = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(W + (size_t)row * a.K + (size_t)kt0 * 32) + c4);
        *reinterpret_cast<f32x4*>(&tile[r][c4 * 4]) = v;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    for (int f = threadIdx.x >> 6; f < 2 * nkt; f += 4) {  // (k tile, k step / row half) pairs, one per wave
        const int ktl = f >> 1, sstep = f & 1;
        const int frow = kPrecutShape16 ? 16 * sstep + (lane & 15) : lane & 31;
        const int wex = range_exponent_bits(a.wmax[mlp * a.H + min(g * 32 + frow, a.H - 1)]);
        const float* p = kPrecutShape16 ? &tile[frow][ktl * 32 + 8 * (lane >> 4)] : &tile[frow][ktl * 32 + 16 * sstep + 8 * (lane >> 5)];
        u32x4 hi, lo;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            _Float16 h0, l0, h1, l1;
            split2h(__builtin_ldexpf(p[2 * j], wex), h0, l0);
            split2h(__builtin_ldexpf(p[2 * j + 1], wex), h1, l1);
            hi[j] = pack_h2(h0, h1);
            lo[j] = pack_h2(l0, l1);
        }
        u32x4* o = reinterpret_cast<u32x4*>(a.img) + ((((size_t)mlp * a.G + g) * a.KT + kt0 + ktl) * 4 + sstep * 2) * 64 + lane;
        o[0] = hi;
        o[64] = lo;
    }
}
size_t precut_image_bytes(int H, int K) { return (K % 32 != 0 || H == 0) ? 0 : (size_t)4 * cdiv(H, 32) * (K / 32) * 4096; }
int launch_precut_weights(const float* const W[4], const unsigned* wmax, void* img, int H, int K, hipStream_t st) {
    PrecutArgs a;
    for (int i = 0; i < 4; ++i) a.W[i] = W[i];
    a.wmax = wmax;
    a.img = static_cast<uint32_t*>(img);
    a.H = H;
    a.K = K;
    a.KT = K / 32;
    a.G = cdiv(H, 32);
    hipLaunchKernelGGL(precut_weights_kernel, dim3(cdiv(a.KT, 8), a.G, 4), dim3(256), 0, st, a);
    return check_launch("precut_weights");
}

struct AnchorSplitArgs {
    const uint32_t* wimg;  // PRECUT: the piece image of the weights (above), else unused
    const float* W[4];
    const uint32_t* xs;
    float* part;
    int H, K, B, KS, Kc, KT, NBLK, groups_per_mlp;
    const unsigned* wmax;  // NP = 2: [4][H] maxima of the weight rows (launch_w_maxima, pack time)
};

#ifdef SHASTA_L1_STAMP  // diagnostic build only (tools/probes/l1_split_probe.hip): in-kernel clock and cycles per tile
__device__ unsigned long long g_split_stamp[4096][3];
#endif
#ifdef SHASTA_L1_TIMELINE  // diagnostic build only (tools/l1_timeline.py): when and where every workgroup ran
__device__ unsigned long long g_split_timeline[4096][4];  // {shader cycles, s_memrealtime at start, at end, XCC_ID << 32 | HW_ID}
#endif

template <int N>
__device__ __forceinline__ void wait_vm_split() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// XT = batch rows per pass / 32 (2, 4 or 8); NS = ring slots; NP = pieces per operand: 3 (bf16, six products) or 2 (fp16, three)
template <int XT, int NS, int NP, bool PRECUT = false>
__global__ __launch_bounds__(256) void anchor_l1_split_kernel(AnchorSplitArgs a) {
    static_assert(!PRECUT || NP == 2, "the pre-cut image holds fp16 pieces");
    constexpr int NPROD = NP == 3 ? 6 : 3;
    constexpr bool S16 = PRECUT && kPrecutShape16;  // experiment: 16x16x32 fragments and accumulators (the comment at SHASTA_L1_SHAPE16)
    constexpr int XCH = 2 * NP * XT;             // 1 KB fragments of one x tile
    constexpr int XPW = XCH / 4;                 // of which every wave fetches this many
    static_assert(XCH % 4 == 0, "x fragments are dealt to four waves");
    constexpr int SLOT = 4 * 1024 + XCH * 256;   // dwords per ring slot: 4 private W tiles + the shared x tile
    constexpr int PER_TILE = 4 + XPW;            // vmcnt units a wave spends per tile
    static_assert(PER_TILE * (NS - 1) <= 63, "vmcnt is 6 bits");
    constexpr int NM = (S16 ? 4 : 2) * NPROD * XT;  // MFMAs per tile
    constexpr int ND = PER_TILE;                 // LDS-DMA instructions per tile and wave
    constexpr int NR = 4 + XCH;                  // ds_read_b128 per tile and wave
    constexpr int SG = PRECUT ? 1 : (NM - 8) / 16;  // MFMA gaps between two weight elements being cut
    constexpr int RPG = PRECUT ? (NR + NM - 2) / (NM - 1) : 1;  // ds_read_b128 per MFMA gap (the pre-cut form also serves 32 / 64 items per pass)
    static_assert(PRECUT ? (1 + (NR + RPG - 1) / RPG <= NM && 1 + ND <= NM) : (1 + NR < NM && 6 + 15 * SG < NM),
                  "side work must fit the MFMA gaps of one tile");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);

    const int ncombo = 2 * a.KS;                 // (K chunk, frame) pairs: consecutive blocks -> consecutive XCDs
    // 1-D grid: the NBLK batch blocks of one (K chunk, frame, row quad) are ncombo apart in the dispatch order - neighbours in
    // time and, with ncombo a multiple of 8, on the same XCD: the later readers of a weight tile find it in that XCD's L2.
    // (With the batch block as grid.y the second pass over the weights started when the first had finished: 8.9 GB of fabric
    // reads per launch at 512 frame-pairs instead of 4.6; the step is 2.6 % shorter with the blocks side by side, 13.4 -> 13.1 J.)
    const int combo = blockIdx.x % ncombo, rest = blockIdx.x / ncombo;
    const int quad = rest / a.NBLK;
    const int ks = combo >> 1, src = combo & 1;
    const int G2 = 2 * a.groups_per_mlp;         // 32-row groups over the two MLPs that read this frame
    const bool active = quad * 4 + wid < G2;     // a spare wave repeats the last group (it still fetches x and joins barriers)
    const int gg = min(quad * 4 + wid, G2 - 1);
    const int mlp = 2 * src + gg / a.groups_per_mlp, r0 = (gg % a.groups_per_mlp) * 32;
    const int bblk = rest % a.NBLK;
    const int tpc = a.Kc >> 5;
    const int kt_beg = ks * tpc, NT = min(a.KT, kt_beg + tpc) - kt_beg;

    const float* wbase = mlp == 0 ? a.W[0] : mlp == 1 ? a.W[1] : mlp == 2 ? a.W[2] : a.W[3];
    const char* wub = PRECUT ? reinterpret_cast<const char*>(a.wimg) + (((size_t)mlp * a.groups_per_mlp + (r0 >> 5)) * a.KT + kt_beg) * 4096
                             : reinterpret_cast<const char*>(wbase + (size_t)r0 * a.K + (size_t)kt_beg * 32);
    const char* xub = reinterpret_cast<const char*>(a.xs) + ((((size_t)src * a.NBLK + bblk) * a.KT + kt_beg) * XCH) * 1024;
    uint32_t woff[4];
    {
        const int cpos = lane & 7, rl = lane >> 3;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int r = 8 * j + rl;
            const int c = cpos ^ ((r >> 1) & 7);
            woff[j] = (uint32_t)(((min(r0 + r, a.H - 1) - r0) * a.K + 4 * c) * 4);
        }
        if (PRECUT) woff[0] = (uint32_t)(lane * 16);  // fragments are lane-linear in the image
    }
    const uint32_t xoff = (uint32_t)(lane * 16 + wid * 1024);
    const uint32_t lds0 = (uint32_t)(size_t)((__attribute__((address_space(3))) float*)lds);
    // instruction idx of this wave for tile t (relative to kt_beg) into ring slot `slot`; no VALU: scalar base + 32-bit lane offset
    auto dma = [&](int t, int slot, int idx) {
        if (idx < 4) {
            const char* base = PRECUT ? wub + (size_t)t * 4096 + (size_t)idx * 1024 : wub + (size_t)t * 128;
            const uint32_t dst = lds0 + (uint32_t)((slot * SLOT + wid * 1024 + idx * 256) * 4);
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 nt" ::"v"(PRECUT ? woff[0] : woff[idx]), "s"(base), "s"(dst)
                         : "memory", "m0");
        } else {
            const int i = idx - 4;
            const char* base = xub + (size_t)t * (XCH * 1024) + (size_t)i * 4096;
            const uint32_t dst = lds0 + (uint32_t)((slot * SLOT + 4096 + (wid + 4 * i) * 256) * 4);
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(xoff), "s"(base), "s"(dst)
                         : "memory", "m0");
        }
    };
    auto issue = [&](int t, int slot) {
#pragma unroll
        for (int j = 0; j < ND; ++j) dma(t, slot, j);
    };

    struct Frag {
        u32x4 A[2][NP];      // weight pieces [k-step (S16: 16-row half)][piece]
        u32x4 X[XT][2][NP];  // activation pieces [32-item block][k-step (S16: 16-item half)][piece]
    };
    f32x4 raw[4];           // fp32 weights of the tile being cut: [2 * k-step + half]
    const int frow = lane & 31, fh = lane >> 5, fsw = (frow >> 1) & 7;
    auto read_one = [&](int slot, Frag& f, int idx) {
        const float* sl = lds + slot * SLOT;
        if (idx < 4) {
            const int s = idx >> 1, qq = idx & 1;
            if constexpr (PRECUT) f.A[s][qq] = *reinterpret_cast<const u32x4*>(sl + wid * 1024 + idx * 256 + lane * 4);  // [k step][piece]
            else raw[idx] = *reinterpret_cast<const f32x4*>(sl + wid * 1024 + frow * 32 + (((4 * s + 2 * fh + qq) ^ fsw) * 4));
        } else {
            const int j = idx - 4;
            f.X[j / (2 * NP)][(j / NP) % 2][j % NP] = *reinterpret_cast<const u32x4*>(sl + 4096 + j * 256 + lane * 4);
        }
    };
    // cut weight element e (0..15) of the tile in `raw`; the pair (e-1, e) is packed when e is odd.  (Packed-f32 subtractions,
    // 9 instead of 11 VALU instructions per pair, measured the same: the kernel is not bound by VALU issue.)
    float ph = 0.0f, pm = 0.0f, pl = 0.0f;
    _Float16 qh = 0, ql = 0;
    int wex = 0;  // exponent of this lane's weight row
    if constexpr (NP == 2 && !PRECUT) wex = range_exponent_bits(a.wmax[mlp * a.H + min(r0 + frow, a.H - 1)]);
    auto cut_one = [&](Frag& f, int e) {
        if constexpr (PRECUT) return;
        const int s = e >> 3, d = (e & 7) >> 1;
        if constexpr (NP == 2) {
            _Float16 h, l;
            split2h(__builtin_ldexpf(raw[e >> 2][e & 3], wex), h, l);
            if ((e & 1) == 0) {
                qh = h;
                ql = l;
            } else {
                f.A[s][0][d] = pack_h2(qh, h);
                f.A[s][1][d] = pack_h2(ql, l);
            }
        } else {
            float h, m, l;
            split3(raw[e >> 2][e & 3], h, m, l);
            if ((e & 1) == 0) {
                ph = h;
                pm = m;
                pl = l;
            } else {
                f.A[s][0][d] = pack_top(ph, h);
                f.A[s][1][d] = pack_top(pm, m);
                f.A[s][NP - 1][d] = pack_top(pl, l);
            }
        }
    };

    f32x16 acc[S16 ? 1 : XT];
    f32x4 acc4[S16 ? XT : 1][2][2];  // S16: the four 16 x 16 blocks [item half c][row half rb] of every 32 x 32 block
#pragma unroll
    for (int u = 0; u < (S16 ? 1 : XT); ++u) acc[u] = f32x16{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int u = 0; u < (S16 ? XT : 1); ++u)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc4[u][c >> 1][c & 1] = f32x4{0, 0, 0, 0};
    // piece products, small to large
    constexpr int PW[6] = {2, 0, 1, 1, 0, 0}, PW2[3] = {1, 0, 0};
    constexpr int PX[6] = {0, 2, 1, 0, 1, 0}, PX2[3] = {0, 1, 0};
    auto mma_one = [&](const Frag& f, int i) {
        if constexpr (S16) {
            // per 32-item block (its fragments die as the next tile's arrive, as in the 32 x 32 form): product-major over the block's
            // four accumulators, so the three products of an accumulator (small to large, as above) are four instructions apart
            const int u = i / (4 * NPROD), pr = (i % (4 * NPROD)) >> 2, c = (i >> 1) & 1, rb = i & 1;
            acc4[u][c][rb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, f.A[rb][PW2[pr]]),
                                                                    __builtin_bit_cast(f16x8, f.X[u][c][PX2[pr]]), acc4[u][c][rb], 0, 0, 0);
            return;
        } else {
        const int s = i / (NPROD * XT), u = (i / NPROD) % XT, pr = i % NPROD;
#ifdef SPLIT_EXP_NOMFMA  // probe: data movement only
        if (i % NPROD != 0) return;
#endif
        if constexpr (NP == 3)
            acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, f.A[s][PW[pr]]),
                                                             __builtin_bit_cast(bf16x8, f.X[u][s][PX[pr]]), acc[u], 0, 0, 0);
        else
            acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, f.A[s][PW2[pr]]),
                                                            __builtin_bit_cast(f16x8, f.X[u][s][PX2[pr]]), acc[u], 0, 0, 0);
        }
    };

#if defined(SHASTA_L1_STAMP) || defined(SHASTA_L1_TIMELINE)
    const unsigned long long st0 = __builtin_amdgcn_s_memtime(), sr0 = __builtin_amdgcn_s_memrealtime();
#endif
    // Prologue: fill the ring, take tile 0 into registers.
    {
        int s = 0;
#pragma unroll 1
        for (int t = 0; t < NS && t < NT; ++t, ++s) issue(t, s);
    }
    Frag fa, fb;
    if (NT >= NS) wait_vm_split<PER_TILE*(NS - 1)>();
    else wait_vm_split<0>();
    __builtin_amdgcn_s_barrier();
    if (NT > 0) {
#pragma unroll
        for (int i = 0; i < NR; ++i) read_one(0, fa, i);
#pragma unroll
        for (int e = 0; e < 16; ++e) cut_one(fa, e);
    }
    // One tile.  `cur` holds tile t in registers; slot sc (tile t's) is refilled with tile t+NS after the barrier; tile t+1
    // is read out of slot sn and cut into `nxt`.  Everything that is not an MFMA sits between two MFMAs.
    auto step = [&](const Frag& cur, Frag& nxt, int t, int sc, auto steady) {
        constexpr bool STEADY = decltype(steady)::value;
        const int sn = sc + 1 == NS ? 0 : sc + 1;
#pragma unroll
        for (int i = 0; i < NM; ++i) {
            mma_one(cur, i);
            __builtin_amdgcn_sched_barrier(0);
            if (i == 0) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // my reads of tile t's slot have retired
                if constexpr (STEADY) wait_vm_split<PER_TILE*(NS - 2)>();
                else wait_vm_split<0>();
#ifndef SPLIT_EXP_NOBAR  // probe: no workgroup coupling (results are wrong)
                __builtin_amdgcn_s_barrier();
#endif
                if constexpr (!STEADY) {
                    if (t + NS < NT) issue(t + NS, sc);
                }
            }
            if (i >= 1 && i - 1 < ND) {
                if constexpr (STEADY) dma(t + NS, sc, i - 1);
            }
            if (i >= 1) {
#pragma unroll
                for (int r = 0; r < RPG; ++r)
                    if ((i - 1) * RPG + r < NR) read_one(sn, nxt, (i - 1) * RPG + r);
            }
            if constexpr (!PRECUT) {
                if (i >= 6 && (i - 6) % SG == 0 && (i - 6) / SG < 16) cut_one(nxt, (i - 6) / SG);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    int t = 0, sc = 0;
    auto next = [&](int s) { return s + 1 == NS ? 0 : s + 1; };
#pragma unroll 1
    for (; t + NS + 1 < NT; t += 2) {
        step(fa, fb, t, sc, std::true_type{});
        sc = next(sc);
        step(fb, fa, t + 1, sc, std::true_type{});
        sc = next(sc);
    }
#pragma unroll 1
    for (; t + 1 < NT; t += 2) {
        step(fa, fb, t, sc, std::false_type{});
        sc = next(sc);
        step(fb, fa, t + 1, sc, std::false_type{});
        sc = next(sc);
    }
    if (t < NT) step(fa, fb, t, sc, std::false_type{});

#ifdef SHASTA_L1_STAMP
    if (lane == 0 && wid == 0 && bblk == 0 && blockIdx.x < 4096) {
        g_split_stamp[blockIdx.x][0] = __builtin_amdgcn_s_memtime() - st0;
        g_split_stamp[blockIdx.x][1] = __builtin_amdgcn_s_memrealtime() - sr0;
        g_split_stamp[blockIdx.x][2] = (unsigned long long)NT;
    }
#endif
#ifdef SHASTA_L1_TIMELINE
    if (lane == 0 && wid == 0 && blockIdx.x < 4096) {
        g_split_timeline[blockIdx.x][0] = __builtin_amdgcn_s_memtime() - st0;
        g_split_timeline[blockIdx.x][1] = sr0;
        g_split_timeline[blockIdx.x][2] = __builtin_amdgcn_s_memrealtime();
        g_split_timeline[blockIdx.x][3] = ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32) | __builtin_amdgcn_s_getreg((31 << 11) | 4);
    }
#endif
    // D[i = weight row][j = batch row]
    if constexpr (S16) {  // 16 x 16 blocks: lane = (item lane & 15, rows 4 (lane >> 4) .. + 3)
#pragma unroll
        for (int u = 0; u < XT; ++u) {
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const int b = (bblk * XT + u) * 32 + 16 * c + (lane & 15);
                if (b >= a.B || !active) continue;
                float* o = a.part + ((size_t)ks * a.B + b) * (4 * a.H) + mlp * a.H;
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    const int row = r0 + 16 * (r >> 2) + 4 * (lane >> 4) + (r & 3);
                    if (row < a.H) o[row] = acc4[u][c][r >> 2][r & 3];
                }
            }
        }
    } else if (active) {
#pragma unroll
        for (int u = 0; u < (S16 ? 1 : XT); ++u) {
            const int b = (bblk * XT + u) * 32 + frow;
            if (b < a.B) {
                float* o = a.part + ((size_t)ks * a.B + b) * (4 * a.H) + mlp * a.H;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = r0 + (r & 3) + 8 * (r >> 2) + 4 * fh;
                    if (row < a.H) o[row] = acc[u][r];
                }
            }
        }
    }
}

// ---- 512 items per weight pass (pre-cut fp16 form) ---------------------------------------------------------------------------------
// The kernel above at 256 items per pass runs with the matrix pipe 76 % busy on a chip that sits on its power cap (~1.54 GHz), reads
// every weight tile once per 256 items (four passes at 1024 frame-pairs: 9 GB of HBM-side traffic for 5.2 GB of operands) and has every
// one of its four waves read the whole 32 KB activation tile out of LDS (144 KB of ds_read per 48 x 4 MFMAs).  This form:
//  * 512 items per pass: the weights cross the fabric half as often;
//  * a workgroup of EIGHT waves = 2 row-group pairs x 4 item groups: a wave owns 64 weight rows x 128 items (8 accumulators as
//    before), reads 4 weight + 8 activation fragments per 24 MFMAs - 96 KB of ds_read per k step and workgroup for twice the items,
//    a third less per product - and two waves share a SIMD, so one's waits (barrier, LDS, DMA) are the other's issue slots;
//  * a ring slot holds ONE k step (16 wide): 8 KB of weight fragments (4 row groups x 2 pieces) + 32 KB of activation fragments
//    (16 item blocks x 2 pieces) = 40 KB, four slots = the whole LDS.  The pre-cut image and the activation image are unchanged (their
//    4 KB / 64 KB blocks are k-step-major resp. hold whole fragments: a slot takes 1 KB fragments wherever they lie).
// Per accumulator the products arrive in the order of the 256-item kernel (k steps ascending; w_l x_h, w_h x_l, w_h x_h).
// Counters (profiles/r05_pmc_*): the matrix pipe is 85 % busy in CYCLES (76 % in the 256-item form); what still moves the kernel is the
// CLOCK the chip holds at its power cap: ablation builds (WIDE_EXP_*) run at 1.82 GHz without the ds_reads, 1.62 GHz without the
// LDS-DMA, 1.45 GHz with everything (under the profiler).  A four-wave form with 128 x 128 register tiles whose activation fragments
// go from L2 straight into registers (32 KB of ds_read per k step instead of 96) was built and measured: correct, but one wave per SIMD
// leaves the pipe 76 % busy and the launch 3 % longer (LABNOTES section 12; git history: anchor_l1_direct_kernel).
template <int NS>
__global__ __launch_bounds__(512) void anchor_l1_wide_kernel(AnchorSplitArgs a) {
    constexpr int XT = 16;                         // 32-item blocks per pass
    constexpr int SLOT = (8 + 2 * XT) * 256;       // dwords per ring slot
    constexpr int ND = 5, NR = 12, NM = 24;        // per k step and wave: LDS-DMA instructions, ds_read_b128, MFMAs
    static_assert(ND * (NS - 1) <= 63, "vmcnt is 6 bits");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int rg2 = wid & 1, ig = wid >> 1;        // row-group pair, item group

    const int ncombo = 2 * a.KS;                   // (K chunk, frame) pairs: consecutive blocks -> consecutive XCDs (as above)
    const int combo = blockIdx.x % ncombo, rest = blockIdx.x / ncombo;
    const int quad = rest / a.NBLK, bblk = rest % a.NBLK;
    const int ks = combo >> 1, src = combo & 1;
    const int G2 = 2 * a.groups_per_mlp;           // 32-row groups over the two MLPs that read this frame
    const int tpc = a.Kc >> 5;
    const int kt_beg = ks * tpc, NT = 2 * (min(a.KT, kt_beg + tpc) - kt_beg);  // k steps of this chunk

    auto group_base = [&](int gq, int& mlp, int& r0) {  // group gq of the quad (a spare one repeats the last)
        const int gg = min(quad * 4 + gq, G2 - 1);
        mlp = 2 * src + gg / a.groups_per_mlp;
        r0 = (gg % a.groups_per_mlp) * 32;
    };
    // this wave's share of a slot: weight fragment wid = (group wid >> 1, piece wid & 1), activation fragments 4 wid .. 4 wid + 3
    int mlpw, r0w;
    group_base(wid >> 1, mlpw, r0w);
    const char* wub = reinterpret_cast<const char*>(a.wimg) + (((size_t)mlpw * a.groups_per_mlp + (r0w >> 5)) * a.KT + kt_beg) * 4096 + (wid & 1) * 1024;
    const char* xub = reinterpret_cast<const char*>(a.xs) + ((((size_t)src * a.NBLK + bblk) * a.KT + kt_beg) * (4 * XT)) * 1024;
    const uint32_t loff = (uint32_t)(lane * 16);
    const uint32_t lds0 = (uint32_t)(size_t)((__attribute__((address_space(3))) float*)lds);
    auto dma = [&](int t, int slot, int idx) {  // k step t of the chunk = (k tile t >> 1, step t & 1)
        if (idx == 0) {
            const char* base = wub + (size_t)(t >> 1) * 4096 + (size_t)(t & 1) * 2048;
            const uint32_t dst = lds0 + (uint32_t)((slot * SLOT + wid * 256) * 4);
            // (no `nt`: the workgroup of the other 512-item block reads the same fragments at about the same time on this XCD - without the
            // hint more of its reads hit the L2: 6.62 -> 6.23 GB fetched per 1024 frame-pairs, the launch time unchanged)
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(loff), "s"(base), "s"(dst) : "memory", "m0");
        } else {
            const int f = 4 * wid + idx - 1, u = f >> 1, pc = f & 1;  // fragment (item block u, piece) of the slot
            const char* base = xub + (size_t)(t >> 1) * (4 * XT * 1024) + (size_t)(((u * 2 + (t & 1)) * 2 + pc) * 1024);
            const uint32_t dst = lds0 + (uint32_t)((slot * SLOT + 2048 + f * 256) * 4);
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(loff), "s"(base), "s"(dst) : "memory", "m0");
        }
    };
    struct Frag {
        u32x4 A[2][2];  // [row group of the pair][piece]
        u32x4 X[4][2];  // [item block of the group][piece]
    };
    auto read_one = [&](int slot, Frag& f, int idx) {
        const float* sl = lds + slot * SLOT + lane * 4;
        if (idx < 4) f.A[idx >> 1][idx & 1] = *reinterpret_cast<const u32x4*>(sl + ((2 * rg2 + (idx >> 1)) * 2 + (idx & 1)) * 256);
        else f.X[(idx - 4) >> 1][idx & 1] = *reinterpret_cast<const u32x4*>(sl + 2048 + ((4 * ig + ((idx - 4) >> 1)) * 2 + (idx & 1)) * 256);
    };
    f32x16 acc[2][4];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int u = 0; u < 4; ++u) acc[j][u] = f32x16{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    constexpr int PW2[3] = {1, 0, 0}, PX2[3] = {0, 1, 0};  // piece products, small to large
    auto mma_one = [&](const Frag& f, int i) {             // product-major: eight independent accumulators between two products of one
        const int pr = i >> 3, j = (i >> 2) & 1, u = i & 3;
#ifdef WIDE_EXP_NOMFMA  // probe: data movement only
        if (pr != 0) return;
#endif
        acc[j][u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, f.A[j][PW2[pr]]), __builtin_bit_cast(f16x8, f.X[u][PX2[pr]]),
                                                           acc[j][u], 0, 0, 0);
    };
    {
        int sl = 0;
#pragma unroll 1
        for (int t = 0; t < NS && t < NT; ++t, ++sl)
#pragma unroll
            for (int j = 0; j < ND; ++j) dma(t, sl, j);
    }
    Frag fa, fb;
    if (NT >= NS) wait_vm_split<ND*(NS - 1)>();
    else wait_vm_split<0>();
    __builtin_amdgcn_s_barrier();
    if (NT > 0) {
#pragma unroll
        for (int i = 0; i < NR; ++i) read_one(0, fa, i);
    }
    // One k step: `cur` holds step t in registers; slot sc (step t's) is refilled with step t + NS after the barrier; step t + 1 is read
    // out of slot sn into `nxt`.  Everything that is not an MFMA sits between two MFMAs.
    auto step = [&](const Frag& cur, Frag& nxt, int t, int sc, auto steady) {
        constexpr bool STEADY = decltype(steady)::value;
        const int sn = sc + 1 == NS ? 0 : sc + 1;
#pragma unroll
        for (int i = 0; i < NM; ++i) {
            mma_one(cur, i);
            __builtin_amdgcn_sched_barrier(0);
            if (i == 0) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // my reads of step t's slot have retired
                if constexpr (STEADY) wait_vm_split<ND*(NS - 2)>();
                else wait_vm_split<0>();
#ifndef WIDE_EXP_NOBAR  // probe: no workgroup coupling (results are wrong)
                __builtin_amdgcn_s_barrier();
#endif
                if constexpr (!STEADY) {
                    if (t + NS < NT) {
#pragma unroll
                        for (int j = 0; j < ND; ++j) dma(t + NS, sc, j);
                    }
                }
            }
#ifndef WIDE_EXP_NODMA  // probe: the ring is never refilled (results are wrong)
            if (i >= 1 && i - 1 < ND) {
                if constexpr (STEADY) dma(t + NS, sc, i - 1);
            }
#endif
#ifndef WIDE_EXP_NOREAD  // probe: the fragments of the first k step for ever (results are wrong)
            if (i >= 1 && i - 1 < NR) read_one(sn, nxt, i - 1);
#endif
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    int t = 0, sc = 0;
    auto next = [&](int s) { return s + 1 == NS ? 0 : s + 1; };
#pragma unroll 1
    for (; t + NS + 1 < NT; t += 2) {
        step(fa, fb, t, sc, std::true_type{});
        sc = next(sc);
        step(fb, fa, t + 1, sc, std::true_type{});
        sc = next(sc);
    }
#pragma unroll 1
    for (; t + 1 < NT; t += 2) {
        step(fa, fb, t, sc, std::false_type{});
        sc = next(sc);
        step(fb, fa, t + 1, sc, std::false_type{});
        sc = next(sc);
    }
    if (t < NT) step(fa, fb, t, sc, std::false_type{});
    // D[i = weight row][j = item]
    const int frow = lane & 31, fh = lane >> 5;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        if (quad * 4 + 2 * rg2 + j >= G2) continue;  // a spare group
        int mlp, r0;
        group_base(2 * rg2 + j, mlp, r0);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int b = (bblk * XT + 4 * ig + u) * 32 + frow;
            if (b >= a.B) continue;
            float* o = a.part + ((size_t)ks * a.B + b) * (4 * a.H) + mlp * a.H;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = r0 + (r & 3) + 8 * (r >> 2) + 4 * fh;
                if (row < a.H) o[row] = acc[j][u][r];
            }
        }
    }
}

// the wide form serves the pre-cut fp16 stream whenever 512-item passes pad the batch no more than 256-item passes do
static inline bool split_wide(int B, int np, bool precut) { return np == 2 && precut && !kPrecutShape16 && B > 256 && cdiv(B, 512) * 512 <= cdiv(B, 256) * 256; }

// np = pieces per operand: 3 = bf16 (six products), 2 = fp16 (three products, SHASTA_OPT_F16X2_WEIGHT_STREAM)
// np == 2 without the pre-cut image is used above 64 rows only; with it, 32 / 64 items per pass serve the smaller batches too
static inline int split_xt(int B, int np) { return np == 2 ? (B > 128 ? 8 : B > 64 ? 4 : B > 32 ? 2 : 1) : (B <= 64 ? 2 : 4); }
static inline int split_nblk(int B, int np) { return cdiv(B, 32 * split_xt(B, np)); }

// bytes of the piece image of the activations (0 for batches the f32 kernels serve): sized for the larger of the two forms
size_t anchor_split_workspace_bytes(int B, int K) {
    if (B < 2 || K % 32 != 0) return 0;
    const size_t b3 = (size_t)2 * split_nblk(B, 3) * 32 * split_xt(B, 3) * (size_t)K * 6;
    const size_t b2 = (size_t)2 * split_nblk(B, 2) * 32 * split_xt(B, 2) * (size_t)K * 4;
    return align_up(b3 > b2 ? b3 : b2, 256);
}

// true when anchor_l1_split_kernel serves this shape (otherwise the f32 kernels of anchor_mfma.hip / anchor.hip do)
bool anchor_split_serves(int B, int K, int x_batch_stride) { return B > 32 && K % 32 == 0 && (x_batch_stride & 3) == 0; }

// cut the activations of both frames into the bf16 fragment image `xs`
void launch_split_x(const float* feat, const float* prev_feat, void* xs, int K, int B, int x_batch_stride, int np, const unsigned* xmax,
                    bool precut, hipStream_t st) {
    const bool wide = split_wide(B, np, precut);
    const int XT = wide ? 16 : split_xt(B, np), NBLK = wide ? cdiv(B, 512) : split_nblk(B, np), KT = K / 32;
    SplitXArgs sx;
    sx.x[0] = feat;
    sx.x[1] = prev_feat;
    sx.xs = static_cast<uint32_t*>(xs);
    sx.B = B;
    sx.KT = KT;
    sx.NBLK = NBLK;
    sx.XT = XT;
    sx.x_batch_stride = x_batch_stride;
    sx.NP = np;
    sx.shape16 = np == 2 && precut && kPrecutShape16;
    sx.xmax = xmax;
    hipLaunchKernelGGL(split_x_kernel, dim3(cdiv(KT, 8), NBLK * XT, 2), dim3(256), 0, st, sx);
}

void launch_anchor_l1_split(const float* const W[4], const void* xs, float* part, int H, int K, int B, int* ks_out, int np,
                            const unsigned* wmax, const void* wimg, hipStream_t st) {
    const bool wide = split_wide(B, np, wimg != nullptr);
    const int XT = wide ? 16 : split_xt(B, np), NBLK = wide ? cdiv(B, 512) : split_nblk(B, np), KT = K / 32;
    AnchorSplitArgs a;
    for (int i = 0; i < 4; ++i) a.W[i] = W[i];
    a.xs = static_cast<const uint32_t*>(xs);
    a.part = part;
    a.H = H;
    a.K = K;
    a.B = B;
    a.KT = KT;
    a.NBLK = NBLK;
    a.groups_per_mlp = cdiv(H, 32);
    a.wmax = wmax;
    a.wimg = static_cast<const uint32_t*>(wimg);
    const int quads = cdiv(2 * a.groups_per_mlp, 4);  // workgroups per (K chunk, frame)
    int ncu = 256;
    {
        int dev = 0;
        if (hipGetDevice(&dev) == hipSuccess) {
            int v = 0;
            if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ncu = v;
        }
    }
    // one workgroup per CU (the ring takes the whole LDS): pick the K split that fills whole rounds of the CU array best
    // ... among splits of at least 4 chunks (when K allows): an fp32 accumulator then sums at most K/4 products, which keeps the
    // accumulation rounding at the level of the 64-row f32 kernel's (measured against float64 at 512 frame-pairs per launch:
    // 2.9e-5 with one 128 000-term chain per accumulator, 7e-6 ... 1.2e-5 with chains of 32 000)
    const int cmin = KT >= 64 ? 4 : 1;
    int ks = cmin;
    double best = -1.0;
    // (the split is chosen for the 256-item form's grid whichever form runs: an accumulator's chunk of K - and with it every sum, bit for
    // bit - is then the same with and without the pre-cut image, 256 or 512 items per pass)
    const int nblk_split = split_nblk(B, np);
    for (int c = cmin; c <= 64 && c * 16 <= KT; ++c) {
        const int wgs = 2 * cdiv(KT, cdiv(KT, c)) * quads * nblk_split;
        const int rounds = cdiv(wgs, ncu);
        const double eff = (double)wgs / ((double)rounds * ncu);
        if (eff > best + 1e-9) {
            best = eff;
            ks = c;
        }
    }
    const int tiles_per = cdiv(KT, ks);
    a.Kc = tiles_per * 32;
    a.KS = cdiv(KT, tiles_per);
    *ks_out = a.KS;
    auto launch = [&](auto kern, int ns, int xt) {
        const size_t ldsb = (size_t)ns * (4 * 1024 + 2 * np * xt * 256) * sizeof(float);
        (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
        hipLaunchKernelGGL(kern, dim3(2 * a.KS * quads * NBLK), dim3(256), ldsb, st, a);
    };
    if (wide) {
        const size_t ldsb = (size_t)4 * (8 + 2 * 16) * 1024;  // four slots of 40 KB: all of the CU's LDS
#ifdef WIDE_EXP_NS3
        if (false) {
#else
        if (hipFuncSetAttribute((const void*)anchor_l1_wide_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb) == hipSuccess) {
#endif
            hipLaunchKernelGGL(anchor_l1_wide_kernel<4>, dim3(2 * a.KS * quads * NBLK), dim3(512), ldsb, st, a);
        } else {  // a device that grants less: three slots
            (void)hipGetLastError();
            const size_t lds3 = (size_t)3 * (8 + 2 * 16) * 1024;
            (void)hipFuncSetAttribute((const void*)anchor_l1_wide_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds3);
            hipLaunchKernelGGL(anchor_l1_wide_kernel<3>, dim3(2 * a.KS * quads * NBLK), dim3(512), lds3, st, a);
        }
    } else if (np == 2 && wimg) {
        if (XT == 1) launch(anchor_l1_split_kernel<1, 6, 2, true>, 6, 1);
        else if (XT == 2) launch(anchor_l1_split_kernel<2, 5, 2, true>, 5, 2);
        else if (XT == 4) launch(anchor_l1_split_kernel<4, 4, 2, true>, 4, 4);
        else launch(anchor_l1_split_kernel<8, 3, 2, true>, 3, 8);
    } else if (np == 2) {
        if (XT == 4) launch(anchor_l1_split_kernel<4, 4, 2>, 4, 4);
        else launch(anchor_l1_split_kernel<8, 3, 2>, 3, 8);
    } else if (XT == 2) launch(anchor_l1_split_kernel<2, 5, 3>, 5, 2);  // 4 slots measure the same: the ring depth is not the limit
    else launch(anchor_l1_split_kernel<4, 4, 3>, 4, 4);
}

}  // namespace shasta

#ifdef SHASTA_L1_TIMELINE
extern "C" __attribute__((visibility("default"))) int shasta_debug_l1_timeline(unsigned long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(shasta::g_split_timeline), sizeof(shasta::g_split_timeline));
}
#endif
