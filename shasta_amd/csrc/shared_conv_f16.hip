// K0 on the fp16 matrix path: shared_conv of the affinity network (det3d/models/tracker/shasta.py:42-47, applied :223-228)
//   Conv2d(Cin -> 64, 3x3, padding 1, bias) -> BatchNorm2d(64) in eval mode -> ReLU -> permute to NHWC
// with fp32 operands in HBM, fp32 accumulation and every fp32 product formed from THREE fp16 products (the two-piece form of the
// weight stream, anchor_split.hip):  a 2^e = a_h + a_l,  a_h = fp16(a 2^e),  a_l = fp16(a 2^e - a_h), both rounded to nearest;
// w x = w_l x_h + w_h x_l + w_h x_h (the dropped w_l x_l is below 2^-22 |w x|).  2^e is an exact power of two - one per OUTPUT CHANNEL for
// the weights (pack time), one per IMAGE for the activations (a max-reduction pass over the map, conv16_absmax_kernel) - that puts
// the largest magnitude into (2^13, 2^14]; the epilogue scales the sums back exactly.  v_mfma_f32_32x32x16_f16 forms 16x the
// products per cycle of v_mfma_f32_32x32x2_f32, so the 38.2 GFLOP of a frame pair have a 0.046 ms matrix floor instead of 0.24 ms.
//
// Implicit GEMM: M = pixels, N = 64 output channels (two 32-wide blocks), K = 9 taps x Cin, walked in chunks of 16 input channels
// (one k-step of 16 = one tap of one chunk; lane half h of an operand fragment holds channels 8h .. 8h+7).
// A workgroup (8 waves) owns 256 CONSECUTIVE pixels of the flattened image (254 workgroups per 180 x 180 map, no padding waste),
// wave w the 32 pixels 32w .. 32w+31 and both channel blocks (2 accumulators).  Per chunk:
//  * input tile: exactly the flat pixel range the 256 pixels touch (one image row + one pixel either side) is loaded along the
//    pixel axis (a lane = 1 pixel x 8 channels, three such items per lane), cut ONCE into its fp16 pieces on the VALU and written
//    to LDS as [piece][channel octet][padded pixel slot][8 fp16]: an operand fragment of a tap is one ds_read_b128 per lane at
//    (a per-lane base) + (a compile-time offset), lane-linear, i.e. conflict-free.  Slots are numbered as in an image whose rows
//    have one padding column either side: the padding slots are zeroed once and never written, so the x-boundary of the
//    convolution costs nothing in the loop (the y-boundary: rows outside the image are never written either).
//  * weights: pre-cut at pack time into the fragment order [chunk][tap][channel block][piece][lane][8 fp16] (36 KB per chunk) and
//    copied straight into LDS by LDS-DMA (global_load_lds_dwordx4), shared by the eight waves.
//  Both tiles are double buffered; one barrier per chunk.  The two waves of a SIMD (w, w+4) cut their share of the next input
//  tile at different points of the chunk, so that one of them always feeds the matrix pipe.
// Several class heads (the seven per-class models of tools/nusc_shasta, official_val.sh) run in ONE launch: blockIdx -> (tile, head)
// with the heads of a tile next to each other on one XCD, so the map is read from HBM once and 6 of 7 tile reads hit that L2.
#include "common.hpp"

#include <string.h>

#pragma clang diagnostic ignored "-Winline-asm"
#include <type_traits>

namespace shasta {

typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));
typedef uint32_t w32x4 __attribute__((ext_vector_type(4)));

constexpr int C16_TILE = 256;                  // pixels per workgroup
constexpr int C16_NSLOT = 640;                 // padded pixel slots of one staged tile (W = 180 needs 626) incl. the 8 trash slots at the end
constexpr int C16_PLANE = C16_NSLOT * 16;      // bytes of one [slot][8 fp16] plane
constexpr int C16_INBUF = 4 * C16_PLANE;       // [piece 2][octet 2] planes
constexpr int C16_WBUF = 9 * 2 * 2 * 1024;     // [tap 9][channel block 2][piece 2] fragments of 1 KB
constexpr int C16_LDS = 2 * C16_WBUF + 2 * C16_INBUF;  // 155 648 bytes: one workgroup per CU
constexpr int C16_MAXH = 8;                    // class heads per launch
constexpr int C16_PARAMS = 320;                // floats behind the fragments: alpha[64], beta'[64], bias[64], 2^-e[64], then [256] = 1.0 for a RAW head
                                               // (train mode: conv + bias as is, no BatchNorm, no ReLU - shared_conv_train.hip takes it from there)

__device__ __forceinline__ void cut2(float a, _Float16& h, _Float16& l) {
    h = (_Float16)a;
    l = (_Float16)(a - (float)h);
}
// a pointer the compiler should keep in scalar registers (operands of the LDS-DMA asm)
__device__ __forceinline__ const char* uniform_ptr(const char* p) {
    const uint64_t v = reinterpret_cast<uint64_t>(p);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return reinterpret_cast<const char*>(((uint64_t)hi << 32) | lo);
}
// x - h, exact in fp32, with h = the low / high half of a packed fp16 pair read as an fp16 operand (v_fma_mix_f32); the results go
// through a compiler-generated conversion before anything else reads them (hazard rule of pair_f16.hip)
typedef float c16f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float c16_res_lo(float x, uint32_t hpk) {
    float r;
    asm("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel_hi:[0,0,1]" : "=v"(r) : "v"(x), "v"(hpk));
    return r;
}
__device__ __forceinline__ float c16_res_hi(float x, uint32_t hpk) {
    float r;
    asm("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(r) : "v"(x), "v"(hpk));
    return r;
}
__device__ __forceinline__ uint32_t pack2h(_Float16 even, _Float16 odd) {
    const h16x2 v = {even, odd};
    return __builtin_bit_cast(uint32_t, v);
}

// ---- pack: one workgroup per output channel --------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void conv16_pack_kernel(const float* __restrict__ w, const float* __restrict__ bias,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          const float* __restrict__ mean, const float* __restrict__ var, float eps, int Cin,
                                                          char* __restrict__ out, int raw) {
    __shared__ float red[4];
    const int n = blockIdx.x, tid = threadIdx.x, K = Cin * 9;
    const float* wn = w + (size_t)n * K;
    float m = 0.0f;
    for (int i = tid; i < K; i += 256) m = absmax_keep_nan(m, fabsf(wn[i]));
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = absmax_keep_nan(m, __shfl_xor(m, off, 64));
    if ((tid & 63) == 0) red[tid >> 6] = m;
    __syncthreads();
    m = absmax_keep_nan(absmax_keep_nan(red[0], red[1]), absmax_keep_nan(red[2], red[3]));
    const int e = range_exponent_bits(__float_as_uint(m));
    const int nchunk = Cin / 16, nb = n >> 5, nl = n & 31;
    for (int it = tid; it < nchunk * 18; it += 256) {
        const int c = it / 18, r = it - c * 18, tap = r >> 1, h = r & 1;
        w32x4 hi, lo;
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            _Float16 h0, l0, h1, l1;
            cut2(__builtin_ldexpf(wn[(16 * c + 8 * h + 2 * jj) * 9 + tap], e), h0, l0);
            cut2(__builtin_ldexpf(wn[(16 * c + 8 * h + 2 * jj + 1) * 9 + tap], e), h1, l1);
            hi[jj] = pack2h(h0, h1);
            lo[jj] = pack2h(l0, l1);
        }
        char* f = out + (size_t)c * C16_WBUF + ((tap * 2 + nb) * 2) * 1024 + (h * 32 + nl) * 16;
        *reinterpret_cast<w32x4*>(f) = hi;
        *reinterpret_cast<w32x4*>(f + 1024) = lo;
    }
    if (tid == 0) {
        float* par = reinterpret_cast<float*>(out + (size_t)nchunk * C16_WBUF);
        const float alpha = raw ? 1.0f : (1.0f / sqrtf(var[n] + eps)) * gamma[n];
        par[n] = alpha;
        par[64 + n] = raw ? 0.0f : beta[n] - mean[n] * alpha;
        par[128 + n] = bias[n];
        par[192 + n] = __builtin_ldexpf(1.0f, -e);
        if (n == 0) par[256] = raw ? 1.0f : 0.0f;
    }
}

// ---- largest magnitude of every image: grid (slices, images); out[image] must be zero on entry ----------------------------------------
__global__ __launch_bounds__(256) void conv16_absmax_kernel(const float* __restrict__ xa, const float* __restrict__ xb, long per_image, int B,
                                                            unsigned* __restrict__ out) {
    __shared__ float red[4];
    const int z = blockIdx.y;
    const float* x = (z >= B ? xb + (size_t)(z - B) * per_image : xa + (size_t)z * per_image);
    const long n4 = per_image / 4, per = (n4 + gridDim.x - 1) / gridDim.x;
    const long beg = blockIdx.x * per, end = min(n4, beg + per);
    const f32x4* p = reinterpret_cast<const f32x4*>(x);
    float m = 0.0f;
    for (long i = beg + threadIdx.x; i < end; i += 256) {
        const f32x4 v = p[i];
        m = absmax_keep_nan(absmax_keep_nan(m, absmax_keep_nan(fabsf(v[0]), fabsf(v[1]))), absmax_keep_nan(fabsf(v[2]), fabsf(v[3])));
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = absmax_keep_nan(m, __shfl_xor(m, off, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0)
        atomicMax(out + z, __float_as_uint(absmax_keep_nan(absmax_keep_nan(red[0], red[1]), absmax_keep_nan(red[2], red[3]))));
}

struct Conv16Args {
    const float* x[2];              // current / previous neck outputs (B, Cin, H, W)
    float* out[2][C16_MAXH];        // per head: (B, H, W, 64)
    const char* packed;             // heads x head_stride bytes
    size_t head_stride;
    const unsigned* xmax;           // [nmaps] bit patterns of the image maxima
    int B, Cin, H, W, heads, tiles_per_map, ntiles, tiles_per_xcd;
};

// PB = 32-pixel blocks per wave.  PB = 1: 8 waves (two per SIMD), each 32 pixels x 64 channels.  PB = 2: 4 waves (one per SIMD, up to 512
// registers), each 64 pixels x 64 channels: every weight fragment read from LDS feeds two pixel blocks (0.67 instead of 1 ds_read_b128
// per MFMA) and no other wave's vector instructions compete with a wave's matrix instructions for issue slots.
template <int PB>
__global__ __launch_bounds__(512 / PB, PB == 1 ? 2 : 1) void shared_conv_f16_kernel(Conv16Args a) {
    constexpr int NW = 8 / PB;            // waves per workgroup
    constexpr int NIT = 24 / NW;          // staged (64-pixel block, octet) items per lane and chunk: up to 24 blocks over the waves
    constexpr int NDMA_LO = (36 + NW - 1) / NW, NDMA_HI = 36 / NW;  // weight fragments per wave: the first waves take one more when 36 % NW != 0
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    // block -> (tile, head): blocks 8 apart share an XCD; an XCD takes a contiguous range of tiles (neighbours share halo rows in
    // its L2) and runs the heads of a tile back to back
    const int xl = blockIdx.x & 7, bslot = blockIdx.x >> 3;
    const int head = bslot % a.heads, tl = bslot / a.heads;
    const int t = xl * a.tiles_per_xcd + tl;
    if (t >= a.ntiles) return;
    const int z = t / a.tiles_per_map, tile = t - z * a.tiles_per_map;
    const bool second = z >= a.B;
    const int b = second ? z - a.B : z;
    const int W = a.W, WT = W + 2, npix = a.H * W, Cin = a.Cin;
    const float* xin = (second ? a.x[1] : a.x[0]) + (size_t)b * Cin * npix;
    float* out = (second ? a.out[1][head] : a.out[0][head]) + (size_t)b * npix * 64;
    const char* wsrc = a.packed + (size_t)head * a.head_stride;
    const int eimg = range_exponent_bits(a.xmax[z]);
    const int p0 = tile * C16_TILE;
    const int y0 = p0 / W, x0 = p0 - y0 * W;
    const int first = (y0 - 1) * WT + x0;  // padded index (y * WT + x + 1) of pixel (y0 - 1, x0 - 1) = slot 0
    char* const in_lds = lds + 2 * C16_WBUF;

    // zero both input buffers once: padding slots and rows outside the image are never written afterwards
    {
        const w32x4 zz = {0u, 0u, 0u, 0u};
        for (int i = tid; i < 2 * C16_INBUF / 16; i += 64 * NW) reinterpret_cast<w32x4*>(in_lds)[i] = zz;
    }

    // staging roles of this lane: three (pixel, channel octet) items of every chunk.  The flat pixel range the tile touches is dealt in
    // blocks of 64 consecutive pixels, first all blocks of octet 0, then those of octet 1; block 8 i + w goes to wave w as its item i:
    // consecutive lanes = consecutive pixels, so the 4-byte loads of a channel row are whole 256-byte lines and the 16-byte LDS stores
    // of a piece are lane-linear (a lane holding FOUR consecutive pixels would store 64 bytes apart: a 4-way bank conflict on every store)
    const int p_last = min(p0 + C16_TILE, npix) - 1;
    const int qs = p0 - W - 1, qe = p_last + W + 1;   // first / last flat pixel a tap of this tile reads (may lie outside the image)
    const int nblk = (qe - qs + 64) >> 6;             // 64-pixel blocks per octet (the host guarantees 2 nblk <= 24)
    int st_addr[NIT];      // LDS byte offset inside an input buffer (octet plane + slot); lanes without a pixel to stage point at one
                               // of the eight trash slots C16_NSLOT - 8 .. - 1 of their octet plane, which no tap ever reads
    unsigned ld_off[NIT];  // byte offset of channel 0 of the octet inside a chunk; always a valid address
    bool item_live[NIT];   // wave-uniform: this wave's item holds pixels at all (its cut and stores are skipped otherwise)
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int blk = it * NW + wv;
        const int oct = blk >= nblk ? 1 : 0;
        const int q = qs + 64 * (blk - oct * nblk) + lane;
        item_live[it] = blk < 2 * nblk;
        const bool ok = item_live[it] && q >= 0 && q < npix && q <= qe;
        const int yy = ok ? q / W : 0, xx = q - yy * W;
        const int sl = yy * WT + xx + 1 - first;
        st_addr[it] = (min(oct, 1) * C16_PLANE) + ((ok && sl >= 0 && sl < C16_NSLOT - 8) ? sl : C16_NSLOT - 8 + (lane & 7)) * 16;  // (sl < 0: the pixel left of the halo when the tile starts a row)
        ld_off[it] = 4u * (unsigned)(oct * 8 * npix + (ok ? q : 0));
    }
    const float scale = __builtin_ldexpf(1.0f, eimg);
    const c16f2 scale2 = {scale, scale};

    // The raw tile travels in registers for a whole trip.  Its loads are inline asm: hipcc builds a 64-bit vector address per load
    // otherwise (24 v_lshl_add_u64 per trip - vector instructions of a wave take issue slots from its SIMD partner's matrix
    // instructions), and the waits below can then be exact: the compiler knows nothing of these loads, every s_waitcnt vmcnt is ours.
    float r[NIT][8];
    auto load_chunk = [&](int ch) __attribute__((always_inline)) {
        const char* xc = reinterpret_cast<const char*>(xin + (size_t)ch * 16 * npix);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const char* xj = uniform_ptr(xc + (size_t)j * npix * 4);
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
#ifdef C16_DBG_CLOADS
                r[it][j] = *reinterpret_cast<const float*>(xj + ld_off[it]);
#else
                asm volatile("global_load_dword %0, %1, %2" : "=v"(r[it][j]) : "v"(ld_off[it]), "s"(xj) : "memory");
#endif
            }
        }
    };
    // wait until at most `left` of this wave's youngest vector-memory operations are in flight, and tie r[] to the wait so that
    // nothing reads a register before it has landed
    auto wait_tile = [&](auto left) __attribute__((always_inline)) {
#if defined(C16_DBG_WAIT0) || defined(C16_DBG_CLOADS)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
        asm volatile("s_waitcnt vmcnt(%8)"
                     : "+v"(r[0][0]), "+v"(r[0][1]), "+v"(r[0][2]), "+v"(r[0][3]), "+v"(r[0][4]), "+v"(r[0][5]), "+v"(r[0][6]), "+v"(r[0][7])
                     : "n"(decltype(left)::value));
#pragma unroll
        for (int it = 1; it < NIT; ++it)
            asm volatile("" : "+v"(r[it][0]), "+v"(r[it][1]), "+v"(r[it][2]), "+v"(r[it][3]), "+v"(r[it][4]), "+v"(r[it][5]), "+v"(r[it][6]), "+v"(r[it][7]));
    };
    const uint32_t lds0 = (uint32_t)(size_t)((__attribute__((address_space(3))) char*)lds);
    const uint32_t dma_off = (uint32_t)(lane * 16);
    // the 36 fragments of a chunk are dealt to the NW waves: wave w copies fragments w, w + NW, ...
    auto dma_weights = [&](int ch, int buf, auto ndma) __attribute__((always_inline)) {
        const char* src = wsrc + (size_t)ch * C16_WBUF + wv * 1024;
        const uint32_t dst0 = lds0 + (uint32_t)(buf * C16_WBUF + wv * 1024);
        const uint32_t off = dma_off;  // (a generic lambda does not capture a variable named only in an asm operand)
#pragma unroll
        for (int j = 0; j < decltype(ndma)::value; ++j) {
            const char* base = uniform_ptr(src + j * (NW * 1024));
            const uint32_t dst = __builtin_amdgcn_readfirstlane(dst0 + (uint32_t)(j * (NW * 1024)));
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(off), "s"(base), "s"(dst) : "memory", "m0");
        }
    };
    auto cut_store = [&](auto bufc) __attribute__((always_inline)) {
        constexpr int IB = 2 * C16_WBUF + decltype(bufc)::value * C16_INBUF;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
#ifndef C16_DBG_NOSKIP
            if (!item_live[it]) continue;  // wave-uniform
#endif
            w32x4 hi, lo;
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {  // 2.5 vector instructions per value: packed scale, packed convert, two exact residuals, packed convert
                const c16f2 sv = c16f2{r[it][2 * jj], r[it][2 * jj + 1]} * scale2;
                const uint32_t hp = pack2h((_Float16)sv[0], (_Float16)sv[1]);
                hi[jj] = hp;
                lo[jj] = pack2h((_Float16)c16_res_lo(sv[0], hp), (_Float16)c16_res_hi(sv[1], hp));
            }
            *reinterpret_cast<w32x4*>(lds + IB + st_addr[it]) = hi;
            *reinterpret_cast<w32x4*>(lds + IB + 2 * C16_PLANE + st_addr[it]) = lo;
        }
    };

    // operand addresses of this lane: pixel block pb of this wave = pixels (32 PB) w + 32 pb + (lane & 31)
    const int li = lane & 31, h = lane >> 5;
    int a_row[PB][3];
#pragma unroll
    for (int pb = 0; pb < PB; ++pb) {
        const int p = min(p0 + 32 * (PB * wv + pb) + li, npix - 1);
        const int py = p / W, px = p - py * W;
        const int sc = py * WT + px + 1 - first;  // slot of the pixel itself (>= WT + 1)
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) a_row[pb][dy] = h * C16_PLANE + (sc + (dy - 1) * WT - 1) * 16;  // tap (dy, dx = 0); dx adds 16 bytes each
    }
    const int b_lane = lane * 16;

    const f32x16 zero16 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    f32x16 acc[PB][2];
#pragma unroll
    for (int pb = 0; pb < PB; ++pb) acc[pb][0] = acc[pb][1] = zero16;
    struct Frag {
        h16x8 ah[PB], al[PB], b0h, b0l, b1h, b1l;
    };
    auto read_tap = [&](auto bufc, int tap, Frag& f) __attribute__((always_inline)) {
        const char* ib = in_lds + decltype(bufc)::value * C16_INBUF;
        const char* wb = lds + decltype(bufc)::value * C16_WBUF + b_lane;
        const int dy = tap / 3, dx = tap % 3;
#pragma unroll
        for (int pb = 0; pb < PB; ++pb) {
            f.ah[pb] = *reinterpret_cast<const h16x8*>(ib + a_row[pb][dy] + dx * 16);
            f.al[pb] = *reinterpret_cast<const h16x8*>(ib + a_row[pb][dy] + dx * 16 + 2 * C16_PLANE);
        }
        f.b0h = *reinterpret_cast<const h16x8*>(wb + (tap * 4 + 0) * 1024);
        f.b0l = *reinterpret_cast<const h16x8*>(wb + (tap * 4 + 1) * 1024);
        f.b1h = *reinterpret_cast<const h16x8*>(wb + (tap * 4 + 2) * 1024);
        f.b1l = *reinterpret_cast<const h16x8*>(wb + (tap * 4 + 3) * 1024);
    };
    auto mma_tap = [&](const Frag& f) __attribute__((always_inline)) {  // piece products, small to large
#ifdef C16_ABL_NO_MFMA  // ablation build (tools/build_variant.py): everything but the matrix instructions; results are wrong
        asm volatile("" ::"v"(f.ah[0]), "v"(f.al[0]), "v"(f.b0h), "v"(f.b0l), "v"(f.b1h), "v"(f.b1l));
        return;
#endif
#pragma unroll
        for (int pb = 0; pb < PB; ++pb) {
            acc[pb][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.al[pb], f.b0h, acc[pb][0], 0, 0, 0);
            acc[pb][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.al[pb], f.b1h, acc[pb][1], 0, 0, 0);
        }
#pragma unroll
        for (int pb = 0; pb < PB; ++pb) {
            acc[pb][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.ah[pb], f.b0l, acc[pb][0], 0, 0, 0);
            acc[pb][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.ah[pb], f.b1l, acc[pb][1], 0, 0, 0);
        }
#pragma unroll
        for (int pb = 0; pb < PB; ++pb) {
            acc[pb][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.ah[pb], f.b0h, acc[pb][0], 0, 0, 0);
            acc[pb][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.ah[pb], f.b1h, acc[pb][1], 0, 0, 0);
        }
    };

    const int nchunk = Cin / 16;
    constexpr int NLD = 8 * NIT;  // global loads of one staged chunk per lane
    static_assert(NLD + NDMA_LO <= 63, "vmcnt is 6 bits");
    using B0 = std::integral_constant<int, 0>;
    using B1 = std::integral_constant<int, 1>;
    constexpr int NSPLIT = 36 % NW == 0 ? NW / 2 : 36 % NW;  // waves below it: NDMA_LO fragments and the early cut; the others NDMA_HI, late cut
    if (wv < NSPLIT) dma_weights(0, 0, std::integral_constant<int, NDMA_LO>{});
    else dma_weights(0, 0, std::integral_constant<int, NDMA_HI>{});
    load_chunk(0);
    __syncthreads();  // the zero fill is complete
    wait_tile(std::integral_constant<int, 0>{});
    cut_store(B0{});
    load_chunk(min(1, nchunk - 1));
    // Everything has landed before the loop is entered, the second tile included: the compiler thinks an asm load's result is there
    // at once and may COPY it (it does, into the loop's registers, right behind this barrier) - a copy of a register whose load is
    // still in flight reads stale data.  Inside the loop the loads write the loop-carried registers themselves (checked in the ISA;
    // tests/test_hip_parity.py::test_shared_conv_vs_oracle fails loudly if a compiler ever changes that).
    wait_tile(std::integral_constant<int, 0>{});
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    // One chunk.  On entry LDS holds chunk ch (tile + weights) and the registers r[] the raw tile of chunk ch + 1, on its way since the
    // middle of the previous trip: it is cut into the other buffer after CUT taps, and the loads of chunk ch + 2 follow at once, so a
    // load has a whole trip to land.  The last trips stage the last chunk again, into the buffer nobody reads any more: no branch in
    // the loop.  The fragments of tap t + 1 are read while tap t is multiplied.  The two waves of a SIMD (w, w + 4) cut at different
    // points of the trip.  Waits: before the cut everything but this trip's LDS-DMA (the loads are older), at the end everything but
    // the NLD loads just issued (the LDS-DMA is older).  Buffer parity is a compile-time constant (two trips per loop iteration).
    auto chunk = [&](int ch, auto bufc, auto cut_after, auto ndma) __attribute__((always_inline)) {
        constexpr int CUT = decltype(cut_after)::value, CUR = decltype(bufc)::value;
        using NXT = std::integral_constant<int, CUR ^ 1>;
        const int nxt = min(ch + 1, nchunk - 1), nx2 = min(ch + 2, nchunk - 1);
#ifndef C16_ABL_NO_DMA
        dma_weights(nxt, CUR ^ 1, ndma);
#endif
        Frag fa, fb;
        read_tap(bufc, 0, fa);
#pragma unroll
        for (int tap = 0; tap < 9; tap += 2) {
            if (tap + 1 < 9) read_tap(bufc, tap + 1, fb);
            mma_tap(fa);
            if (tap + 1 == CUT) {
                __builtin_amdgcn_sched_barrier(0);  // (the scheduler otherwise lifts the cut - and its wait for the loads - to the top of the trip)
#ifndef C16_ABL_NO_STAGE
                wait_tile(ndma);
                cut_store(NXT{});
                load_chunk(nx2);
#endif
                __builtin_amdgcn_sched_barrier(0);
            }
            if (tap + 1 < 9) {
                if (tap + 2 < 9) read_tap(bufc, tap + 2, fa);
                mma_tap(fb);
                if (tap + 2 == CUT) {
                    __builtin_amdgcn_sched_barrier(0);
#ifndef C16_ABL_NO_STAGE
                    wait_tile(ndma);
                    cut_store(NXT{});
                    load_chunk(nx2);
#endif
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
#ifdef C16_ABL_NO_STAGE
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#else
#if defined(C16_DBG_WAIT0) || defined(C16_DBG_CLOADS)
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#else
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(NLD) : "memory");
#endif
#endif
#ifndef C16_ABL_NO_BARRIER
        __builtin_amdgcn_s_barrier();
#endif
        asm volatile("" ::: "memory");
    };
#ifndef C16_CUT_A
#define C16_CUT_A 2
#endif
#ifndef C16_CUT_B
#define C16_CUT_B 6
#endif
    auto all_chunks = [&](auto cut_after, auto ndma) __attribute__((always_inline)) {
        int ch = 0;
#pragma unroll 1
        for (; ch + 1 < nchunk; ch += 2) {
            chunk(ch, B0{}, cut_after, ndma);
            chunk(ch + 1, B1{}, cut_after, ndma);
        }
        if (ch < nchunk) chunk(ch, B0{}, cut_after, ndma);
    };
    if (wv < NSPLIT) all_chunks(std::integral_constant<int, C16_CUT_A>{}, std::integral_constant<int, NDMA_LO>{});
    else all_chunks(std::integral_constant<int, C16_CUT_B>{}, std::integral_constant<int, NDMA_HI>{});
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the staged-again last chunk: nothing may be in flight when the wave ends

    // epilogue: D[pixel][channel]: lane = channel (32 nb + li), pixel = (r & 3) + 8 (r >> 2) + 4 h of the block's 32
    const float* par = reinterpret_cast<const float*>(wsrc + (size_t)nchunk * C16_WBUF);
    const float back = __builtin_ldexpf(1.0f, -eimg);
    const bool raw = par[256] != 0.0f;  // uniform
#pragma unroll
    for (int pb = 0; pb < PB; ++pb) {
        const int pblk = p0 + 32 * (PB * wv + pb);
        if (pblk >= npix) continue;  // wave-uniform
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
            const int chn = 32 * nb + li;
            const float alpha = par[chn], beta2 = par[64 + chn], bias = par[128 + chn], un = par[192 + chn] * back;
#pragma unroll
            for (int rr = 0; rr < 16; ++rr) {
                const int pp = pblk + (rr & 3) + 8 * (rr >> 2) + 4 * h;
                if (pp < npix) {
                    const float sv = acc[pb][nb][rr] * un;
                    const float v = (sv + bias) * alpha + beta2;
                    out[(size_t)pp * 64 + chn] = raw ? v : relu_nan(v);
                }
            }
        }
    }
}

// ---- 512-pixel tiles (maps in bulk) -----------------------------------------------------------------------------------------------
// Per chunk the kernel above reads 432 KB of operand fragments out of LDS (3 375 cycles of its 128 B / clk) for 3 456 cycles of matrix
// work per SIMD: the two limits coincide and do not overlap perfectly - a chunk takes ~6 300 cycles.  Three quarters of those reads are
// the weight fragments, which every wave fetches for its single 32-pixel block.  Here a wave owns TWO pixel blocks (64 pixels x 64
// channels, four accumulators) and the workgroup of eight waves 512 consecutive pixels: a weight fragment feeds two MFMAs, 8 instead
// of 12 ds_read_b128 per 12 MFMAs, and the weights cross from L2 into LDS once per 512 pixels.  Two waves per SIMD as before.  LDS: the
// input tile grows to 2 x 57 KB (912 padded slots), so the weights of a chunk are no longer double-buffered whole: their 36 fragments
// sit in ONE 36 KB region in two halves - taps 0-4 and taps 5-8 - each refilled by LDS-DMA as soon as every wave has passed it (a
// barrier in the middle of the chunk, one at its end): half B of chunk c is requested at the top of chunk c and has the time of taps
// 0-4 to land, half A of chunk c + 1 is requested behind the middle barrier and has taps 5-8.
// Waits (every vector-memory operation of the loop is ours, see above): the waves that cut early (w < 4, the SIMD partners of the late
// ones) issue their 32 loads of chunk c + 2 between the two DMA batches of a trip - middle: everything but those loads, end:
// everything; the late waves issue them behind both - middle: everything, end: everything but the loads.
constexpr int W2_TILE = 512;
constexpr int W2_NSLOT = 912;                  // padded pixel slots incl. the 8 trash slots (W = 187: 900 + 8)
constexpr int W2_PLANE = W2_NSLOT * 16;
constexpr int W2_INBUF = 4 * W2_PLANE;         // [piece 2][octet 2] planes
constexpr int W2_LDS = C16_WBUF + 2 * W2_INBUF;  // 153 600 bytes
constexpr int W2_NIT = 4;                      // staged (64-pixel block, octet) items per lane and chunk: up to 32 blocks over 8 waves
constexpr int W2_HALF = 20;                    // fragments of taps 0-4

__global__ __launch_bounds__(512, 2) void shared_conv_f16w_kernel(Conv16Args a) {
    constexpr int NW = 8, PB = 2, NIT = W2_NIT;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int xl = blockIdx.x & 7, bslot = blockIdx.x >> 3;
    const int head = bslot % a.heads, tl = bslot / a.heads;
    const int t = xl * a.tiles_per_xcd + tl;
    if (t >= a.ntiles) return;
    const int z = t / a.tiles_per_map, tile = t - z * a.tiles_per_map;
    const bool second = z >= a.B;
    const int b = second ? z - a.B : z;
    const int W = a.W, WT = W + 2, npix = a.H * W, Cin = a.Cin;
    const float* xin = (second ? a.x[1] : a.x[0]) + (size_t)b * Cin * npix;
    float* out = (second ? a.out[1][head] : a.out[0][head]) + (size_t)b * npix * 64;
    const char* wsrc = a.packed + (size_t)head * a.head_stride;
    const int eimg = range_exponent_bits(a.xmax[z]);
    const int p0 = tile * W2_TILE;
    const int y0 = p0 / W, x0 = p0 - y0 * W;
    const int first = (y0 - 1) * WT + x0;  // padded index (y * WT + x + 1) of pixel (y0 - 1, x0 - 1) = slot 0
    char* const in_lds = lds + C16_WBUF;
    {   // zero both input buffers once: padding slots and rows outside the image are never written afterwards
        const w32x4 zz = {0u, 0u, 0u, 0u};
        for (int i = tid; i < 2 * W2_INBUF / 16; i += 64 * NW) reinterpret_cast<w32x4*>(in_lds)[i] = zz;
    }
    // staging roles of this lane (as above): block 8 i + w of the 64-pixel blocks (first all of octet 0, then those of octet 1) is item i of wave w
    const int p_last = min(p0 + W2_TILE, npix) - 1;
    const int qs = p0 - W - 1, qe = p_last + W + 1;
    const int nblk = (qe - qs + 64) >> 6;  // (the host guarantees 2 nblk <= 32)
    int st_addr[NIT];
    unsigned ld_off[NIT];
    bool item_live[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int blk = it * NW + wv;
        const int oct = blk >= nblk ? 1 : 0;
        const int q = qs + 64 * (blk - oct * nblk) + lane;
        item_live[it] = blk < 2 * nblk;
        const bool ok = item_live[it] && q >= 0 && q < npix && q <= qe;
        const int yy = ok ? q / W : 0, xx = q - yy * W;
        const int sl = yy * WT + xx + 1 - first;
        st_addr[it] = (min(oct, 1) * W2_PLANE) + ((ok && sl >= 0 && sl < W2_NSLOT - 8) ? sl : W2_NSLOT - 8 + (lane & 7)) * 16;
        ld_off[it] = 4u * (unsigned)(oct * 8 * npix + (ok ? q : 0));
    }
    const float scale = __builtin_ldexpf(1.0f, eimg);
    const c16f2 scale2 = {scale, scale};
    float r[NIT][8];
    auto load_chunk = [&](int ch) __attribute__((always_inline)) {
        const char* xc = reinterpret_cast<const char*>(xin + (size_t)ch * 16 * npix);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const char* xj = uniform_ptr(xc + (size_t)j * npix * 4);
#pragma unroll
            for (int it = 0; it < NIT; ++it) asm volatile("global_load_dword %0, %1, %2" : "=v"(r[it][j]) : "v"(ld_off[it]), "s"(xj) : "memory");
        }
    };
    auto wait_tile = [&](auto left) __attribute__((always_inline)) {
        asm volatile("s_waitcnt vmcnt(%8)"
                     : "+v"(r[0][0]), "+v"(r[0][1]), "+v"(r[0][2]), "+v"(r[0][3]), "+v"(r[0][4]), "+v"(r[0][5]), "+v"(r[0][6]), "+v"(r[0][7])
                     : "n"(decltype(left)::value));
#pragma unroll
        for (int it = 1; it < NIT; ++it)
            asm volatile("" : "+v"(r[it][0]), "+v"(r[it][1]), "+v"(r[it][2]), "+v"(r[it][3]), "+v"(r[it][4]), "+v"(r[it][5]), "+v"(r[it][6]), "+v"(r[it][7]));
    };
    const uint32_t lds0 = (uint32_t)(size_t)((__attribute__((address_space(3))) char*)lds);
    const uint32_t dma_off = (uint32_t)(lane * 16);
    // fragments first .. first + count - 1 of chunk ch (its 36 fragments lie in the packed buffer as in LDS): wave w copies first + w, + 8, ...
    auto dma_frags = [&](int ch, auto firstc, auto countc) __attribute__((always_inline)) {
        constexpr int F0 = decltype(firstc)::value, CNT = decltype(countc)::value;
        const char* src = wsrc + (size_t)ch * C16_WBUF + (F0 + wv) * 1024;
        const uint32_t dst0 = lds0 + (uint32_t)((F0 + wv) * 1024);
        const uint32_t off = dma_off;
#pragma unroll
        for (int j = 0; j < CNT; ++j) {
            const char* base = uniform_ptr(src + j * (NW * 1024));
            const uint32_t dst = __builtin_amdgcn_readfirstlane(dst0 + (uint32_t)(j * (NW * 1024)));
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(off), "s"(base), "s"(dst) : "memory", "m0");
        }
    };
    auto cut_store = [&](auto bufc) __attribute__((always_inline)) {
        constexpr int IB = C16_WBUF + decltype(bufc)::value * W2_INBUF;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            if (!item_live[it]) continue;  // wave-uniform
            w32x4 hi, lo;
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const c16f2 sv = c16f2{r[it][2 * jj], r[it][2 * jj + 1]} * scale2;
                const uint32_t hp = pack2h((_Float16)sv[0], (_Float16)sv[1]);
                hi[jj] = hp;
                lo[jj] = pack2h((_Float16)c16_res_lo(sv[0], hp), (_Float16)c16_res_hi(sv[1], hp));
            }
            *reinterpret_cast<w32x4*>(lds + IB + st_addr[it]) = hi;
            *reinterpret_cast<w32x4*>(lds + IB + 2 * W2_PLANE + st_addr[it]) = lo;
        }
    };
    // operand addresses of this lane: pixel block pb of this wave = pixels 64 w + 32 pb + (lane & 31)
    const int li = lane & 31, h = lane >> 5;
    int a_row[PB][3];
#pragma unroll
    for (int pb = 0; pb < PB; ++pb) {
        const int p = min(p0 + 32 * (PB * wv + pb) + li, npix - 1);
        const int py = p / W, px = p - py * W;
        const int sc = py * WT + px + 1 - first;
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) a_row[pb][dy] = h * W2_PLANE + (sc + (dy - 1) * WT - 1) * 16;
    }
    const int b_lane = lane * 16;
    const f32x16 zero16 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    f32x16 acc[PB][2];
#pragma unroll
    for (int pb = 0; pb < PB; ++pb) acc[pb][0] = acc[pb][1] = zero16;
    struct Frag {
        h16x8 ah[PB], al[PB], b0h, b0l, b1h, b1l;
    };
    auto read_tap = [&](auto bufc, int tap, Frag& f) __attribute__((always_inline)) {
        const char* ib = in_lds + decltype(bufc)::value * W2_INBUF;
        const char* wb = lds + b_lane;
        const int dy = tap / 3, dx = tap % 3;
#pragma unroll
        for (int pb = 0; pb < PB; ++pb) {
            f.ah[pb] = *reinterpret_cast<const h16x8*>(ib + a_row[pb][dy] + dx * 16);
            f.al[pb] = *reinterpret_cast<const h16x8*>(ib + a_row[pb][dy] + dx * 16 + 2 * W2_PLANE);
        }
        f.b0h = *reinterpret_cast<const h16x8*>(wb + (tap * 4 + 0) * 1024);
        f.b0l = *reinterpret_cast<const h16x8*>(wb + (tap * 4 + 1) * 1024);
        f.b1h = *reinterpret_cast<const h16x8*>(wb + (tap * 4 + 2) * 1024);
        f.b1l = *reinterpret_cast<const h16x8*>(wb + (tap * 4 + 3) * 1024);
    };
    auto mma_tap = [&](const Frag& f) __attribute__((always_inline)) {  // piece products, small to large
#ifdef W2_ABL_NO_MFMA
        asm volatile("" ::"v"(f.ah[0]), "v"(f.al[0]), "v"(f.ah[1]), "v"(f.al[1]), "v"(f.b0h), "v"(f.b0l), "v"(f.b1h), "v"(f.b1l));
        return;
#endif
#pragma unroll
        for (int pb = 0; pb < PB; ++pb) {
            acc[pb][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.al[pb], f.b0h, acc[pb][0], 0, 0, 0);
            acc[pb][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.al[pb], f.b1h, acc[pb][1], 0, 0, 0);
        }
#pragma unroll
        for (int pb = 0; pb < PB; ++pb) {
            acc[pb][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.ah[pb], f.b0l, acc[pb][0], 0, 0, 0);
            acc[pb][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.ah[pb], f.b1l, acc[pb][1], 0, 0, 0);
        }
#pragma unroll
        for (int pb = 0; pb < PB; ++pb) {
            acc[pb][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.ah[pb], f.b0h, acc[pb][0], 0, 0, 0);
            acc[pb][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.ah[pb], f.b1h, acc[pb][1], 0, 0, 0);
        }
    };
    const int nchunk = Cin / 16;
    constexpr int NLD = 8 * NIT;
    using B0 = std::integral_constant<int, 0>;
    using B1 = std::integral_constant<int, 1>;
    using I0 = std::integral_constant<int, 0>;
    const bool early = wv < 4;  // three fragments of half A and the early cut; the SIMD partner w + 4: two and the late cut
    // prologue: all 36 fragments of chunk 0, tile 0 cut into buffer 0, the raw tile of chunk 1 in registers; everything has landed (see
    // the note at this point of the kernel above)
    if (early) dma_frags(0, I0{}, std::integral_constant<int, 3>{});
    else dma_frags(0, I0{}, std::integral_constant<int, 2>{});
    dma_frags(0, std::integral_constant<int, W2_HALF>{}, std::integral_constant<int, 2>{});
    load_chunk(0);
    __syncthreads();  // the zero fill is complete
    wait_tile(I0{});
    cut_store(B0{});
    load_chunk(min(1, nchunk - 1));
    wait_tile(I0{});
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    // One chunk (EARLY / buffer parity compile-time).  On entry: LDS holds tile ch, ALL 36 weight fragments of chunk ch when ch == 0,
    // otherwise half A of chunk ch (half B still chunk ch - 1's, consumed by everybody: requested now); r[] = the raw tile of ch + 1.
    auto chunk = [&](int ch, auto bufc, auto earlyc) __attribute__((always_inline)) {
        constexpr int CUR = decltype(bufc)::value;
        constexpr bool EARLY = decltype(earlyc)::value;
        constexpr int CUT = EARLY ? 2 : 7;  // taps multiplied before this wave cuts the next tile
        using NXT = std::integral_constant<int, CUR ^ 1>;
        const int nxt = min(ch + 1, nchunk - 1), nx2 = min(ch + 2, nchunk - 1);
        if (ch > 0) dma_frags(ch, std::integral_constant<int, W2_HALF>{}, std::integral_constant<int, 2>{});  // half B of this chunk
        auto stage = [&]() __attribute__((always_inline)) {
            __builtin_amdgcn_sched_barrier(0);
            wait_tile(std::integral_constant<int, 2>{});  // everything but the two fragments this wave requested last
#ifndef W2_ABL_NO_CUT   // ablation builds (tools/build_variant.py; results are wrong)
            cut_store(NXT{});
#endif
#ifndef W2_ABL_NO_LOAD
            load_chunk(nx2);
#endif
            __builtin_amdgcn_sched_barrier(0);
        };
        Frag fa, fb;
        read_tap(bufc, 0, fa);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            Frag& cur = (tap & 1) ? fb : fa;
            Frag& nx = (tap & 1) ? fa : fb;
            if (tap + 1 < 9 && tap != 4) read_tap(bufc, tap + 1, nx);  // (tap 5's weights: behind the middle barrier)
            mma_tap(cur);
            if (tap + 1 == CUT) stage();
            if (tap == 4) {
                // middle: half B of this chunk has landed (this wave's share; the barrier: everybody's) and everybody is past half A,
                // which takes chunk ch + 1's fragments
                if (EARLY) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(NLD) : "memory");
                else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                if (EARLY) dma_frags(nxt, I0{}, std::integral_constant<int, 3>{});
                else dma_frags(nxt, I0{}, std::integral_constant<int, 2>{});
                read_tap(bufc, 5, nx);
            }
        }
        if (EARLY) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(NLD) : "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
    auto all_chunks = [&](auto earlyc) __attribute__((always_inline)) {
        int ch = 0;
#pragma unroll 1
        for (; ch + 1 < nchunk; ch += 2) {
            chunk(ch, B0{}, earlyc);
            chunk(ch + 1, B1{}, earlyc);
        }
        if (ch < nchunk) chunk(ch, B0{}, earlyc);
    };
    if (early) all_chunks(std::true_type{});
    else all_chunks(std::false_type{});
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the staged-again last chunk and the last refill of half A: nothing in flight at the end

    // epilogue: D[pixel][channel]: lane = channel (32 nb + li), pixel = (r & 3) + 8 (r >> 2) + 4 h of the block's 32
    const float* par = reinterpret_cast<const float*>(wsrc + (size_t)nchunk * C16_WBUF);
    const float back = __builtin_ldexpf(1.0f, -eimg);
    const bool raw = par[256] != 0.0f;  // uniform
#pragma unroll
    for (int pb = 0; pb < PB; ++pb) {
        const int pblk = p0 + 32 * (PB * wv + pb);
        if (pblk >= npix) continue;  // wave-uniform
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
            const int chn = 32 * nb + li;
            const float alpha = par[chn], beta2 = par[64 + chn], bias = par[128 + chn], un = par[192 + chn] * back;
#pragma unroll
            for (int rr = 0; rr < 16; ++rr) {
                const int pp = pblk + (rr & 3) + 8 * (rr >> 2) + 4 * h;
                if (pp < npix) {
                    const float sv = acc[pb][nb][rr] * un;
                    const float v = (sv + bias) * alpha + beta2;
                    out[(size_t)pp * 64 + chn] = raw ? v : relu_nan(v);
                }
            }
        }
    }
}

// ---- the input cut once for all heads (maps in bulk, several class heads) ----------------------------------------------------------
// Ablation builds of the kernel above (W2_ABL_*, 8 frame pairs, one head): 0.89 ms with everything, 0.67 ms without its staging (loads,
// cut, LDS stores: the vector work that competes with the SIMD partner's matrix instructions), and the seven class heads of
// tools/nusc_shasta/eval.py:86-101 each cut the SAME input tile again.  Here a bandwidth-bound pre-pass (conv16_precut_kernel) cuts every
// map once into the fp16 piece image of ALL its tiles - [chunk][piece][octet][padded pixel slot][8 fp16], slots numbered as in the image
// with one zero column either side and one zero row above and below, i.e. exactly the numbering of the staged tiles - and a tile of any
// workgroup is a CONTIGUOUS window of that image: the matrix kernel (shared_conv_f16p_kernel) fetches it by LDS-DMA and holds nothing
// but DMA, ds_read and MFMA in its loop (no staging registers: 218 -> ~150 VGPRs).  Costs one write and one read of the map's worth of
// bytes extra (0.4 ms at 8 frame pairs): taken from three heads per launch on; 68 MB of workspace per map.
constexpr int P3_NSLOT = 896;                  // slots of a staged tile: 14 LDS-DMA instructions of 1 KB per plane (W <= 185)
constexpr int P3_PLANE = P3_NSLOT * 16;
constexpr int P3_INBUF = 4 * P3_PLANE;         // [piece 2][octet 2] planes
constexpr int P3_LDS = C16_WBUF + 2 * P3_INBUF;  // 151 552 bytes
// slots of one plane of a map's piece image: the padded image and a zero tail as long as a tile window
static inline long conv16p_plane_slots(int H, int W) { return ((long)(H + 2) * (W + 2) + P3_NSLOT + 63) / 64 * 64; }

// grid (ceil(plane slots / 256), 2 chunks-octets ..., maps): thread = (padded slot, octet) of one chunk
__global__ __launch_bounds__(256) void conv16_precut_kernel(const float* __restrict__ xa, const float* __restrict__ xb, int B, int Cin, int H, int W,
                                                            const unsigned* __restrict__ xmax, w32x4* __restrict__ img, long plane_slots) {
    const int z = blockIdx.z, co = blockIdx.y, ch = co >> 1, oct = co & 1;
    const long s = (long)blockIdx.x * 256 + threadIdx.x;
    if (s >= plane_slots) return;
    const int WT = W + 2, npix = H * W;
    const long row = s / WT;
    const int xp = (int)(s - row * WT);
    const bool real = row >= 1 && row <= H && xp >= 1 && xp <= W;
    w32x4 hi = {0u, 0u, 0u, 0u}, lo = {0u, 0u, 0u, 0u};
    if (real) {
        const float* x = (z >= B ? xb + (size_t)(z - B) * Cin * npix : xa + (size_t)z * Cin * npix) + (size_t)(16 * ch + 8 * oct) * npix + (row - 1) * W + (xp - 1);
        const float scale = __builtin_ldexpf(1.0f, range_exponent_bits(xmax[z]));
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const float v0 = x[(size_t)(2 * jj) * npix] * scale, v1 = x[(size_t)(2 * jj + 1) * npix] * scale;
            const uint32_t hp = pack2h((_Float16)v0, (_Float16)v1);
            hi[jj] = hp;
            lo[jj] = pack2h((_Float16)c16_res_lo(v0, hp), (_Float16)c16_res_hi(v1, hp));
        }
    }
    w32x4* o = img + (((size_t)z * (Cin / 16) + ch) * 4 + oct) * plane_slots + s;
    o[0] = hi;
    o[2 * plane_slots] = lo;
}

struct Conv16pArgs {
    Conv16Args c;
    const char* img;   // [map][chunk][piece 2][octet 2][plane slots][16 B]
    long plane_slots;
};

__global__ __launch_bounds__(512, 2) void shared_conv_f16p_kernel(Conv16pArgs pa) {
    constexpr int NW = 8, PB = 2;
    const Conv16Args& a = pa.c;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int xl = blockIdx.x & 7, bslot = blockIdx.x >> 3;
    const int head = bslot % a.heads, tl = bslot / a.heads;
    const int t = xl * a.tiles_per_xcd + tl;
    if (t >= a.ntiles) return;
    const int z = t / a.tiles_per_map, tile = t - z * a.tiles_per_map;
    const bool second = z >= a.B;
    const int b = second ? z - a.B : z;
    const int W = a.W, WT = W + 2, npix = a.H * W, Cin = a.Cin;
    float* out = (second ? a.out[1][head] : a.out[0][head]) + (size_t)b * npix * 64;
    const char* wsrc = a.packed + (size_t)head * a.head_stride;
    const int eimg = range_exponent_bits(a.xmax[z]);
    const int p0 = tile * W2_TILE;
    const int y0 = p0 / W, x0 = p0 - y0 * W;
    const long gidx0 = (long)y0 * WT + x0;  // image slot of tile slot 0 = pixel (y0 - 1, x0 - 1): (row y0 - 1 + 1) * WT + (x0 - 1 + 1)
    char* const in_lds = lds + C16_WBUF;
    const int nchunk = Cin / 16;
    const char* tsrc = pa.img + ((size_t)z * nchunk * 4 * pa.plane_slots + (size_t)gidx0) * 16;  // plane (chunk 0, piece 0, octet 0), slot gidx0
    const size_t plane_bytes = (size_t)pa.plane_slots * 16;

    const uint32_t lds0 = (uint32_t)(size_t)((__attribute__((address_space(3))) char*)lds);
    const uint32_t dma_off = (uint32_t)(lane * 16);
    auto dma_frags = [&](int ch, auto firstc, auto countc) __attribute__((always_inline)) {
        constexpr int F0 = decltype(firstc)::value, CNT = decltype(countc)::value;
        const char* src = wsrc + (size_t)ch * C16_WBUF + (F0 + wv) * 1024;
        const uint32_t dst0 = lds0 + (uint32_t)((F0 + wv) * 1024);
        const uint32_t off = dma_off;
#pragma unroll
        for (int j = 0; j < CNT; ++j) {
            const char* base = uniform_ptr(src + j * (NW * 1024));
            const uint32_t dst = __builtin_amdgcn_readfirstlane(dst0 + (uint32_t)(j * (NW * 1024)));
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(off), "s"(base), "s"(dst) : "memory", "m0");
        }
    };
    // the tile of chunk ch into input buffer `buf`: 4 planes x 14 KB = 56 instructions of 1 KB, seven per wave (instruction 8 j + w)
    auto dma_tile = [&](int ch, int buf) __attribute__((always_inline)) {
        const uint32_t off = dma_off;
#pragma unroll
        for (int j = 0; j < 7; ++j) {
            const int i = 8 * j + wv, pl = i / 14, k = i - pl * 14;  // plane (piece * 2 + octet), KB within it
            const char* base = uniform_ptr(tsrc + ((size_t)ch * 4 + pl) * plane_bytes + (size_t)k * 1024);
            const uint32_t dst = __builtin_amdgcn_readfirstlane(lds0 + (uint32_t)(C16_WBUF + buf * P3_INBUF + pl * P3_PLANE + k * 1024));
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(off), "s"(base), "s"(dst) : "memory", "m0");
        }
    };
    const int li = lane & 31, h = lane >> 5;
    int a_row[PB][3];
#pragma unroll
    for (int pb = 0; pb < PB; ++pb) {
        const int p = min(p0 + 32 * (PB * wv + pb) + li, npix - 1);
        const int py = p / W, px = p - py * W;
        const int sc = (int)((long)(py + 1) * WT + px + 1 - gidx0);  // tile slot of the pixel itself
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) a_row[pb][dy] = h * P3_PLANE + (sc + (dy - 1) * WT - 1) * 16;
    }
    const int b_lane = lane * 16;
    const f32x16 zero16 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    f32x16 acc[PB][2];
#pragma unroll
    for (int pb = 0; pb < PB; ++pb) acc[pb][0] = acc[pb][1] = zero16;
    struct Frag {
        h16x8 ah[PB], al[PB], b0h, b0l, b1h, b1l;
    };
    auto read_tap = [&](auto bufc, int tap, Frag& f) __attribute__((always_inline)) {
        const char* ib = in_lds + decltype(bufc)::value * P3_INBUF;
        const char* wb = lds + b_lane;
        const int dy = tap / 3, dx = tap % 3;
#pragma unroll
        for (int pb = 0; pb < PB; ++pb) {
            f.ah[pb] = *reinterpret_cast<const h16x8*>(ib + a_row[pb][dy] + dx * 16);
            f.al[pb] = *reinterpret_cast<const h16x8*>(ib + a_row[pb][dy] + dx * 16 + 2 * P3_PLANE);
        }
        f.b0h = *reinterpret_cast<const h16x8*>(wb + (tap * 4 + 0) * 1024);
        f.b0l = *reinterpret_cast<const h16x8*>(wb + (tap * 4 + 1) * 1024);
        f.b1h = *reinterpret_cast<const h16x8*>(wb + (tap * 4 + 2) * 1024);
        f.b1l = *reinterpret_cast<const h16x8*>(wb + (tap * 4 + 3) * 1024);
    };
    auto mma_tap = [&](const Frag& f) __attribute__((always_inline)) {  // piece products, small to large
#pragma unroll
        for (int pb = 0; pb < PB; ++pb) {
            acc[pb][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.al[pb], f.b0h, acc[pb][0], 0, 0, 0);
            acc[pb][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.al[pb], f.b1h, acc[pb][1], 0, 0, 0);
        }
#pragma unroll
        for (int pb = 0; pb < PB; ++pb) {
            acc[pb][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.ah[pb], f.b0l, acc[pb][0], 0, 0, 0);
            acc[pb][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.ah[pb], f.b1l, acc[pb][1], 0, 0, 0);
        }
#pragma unroll
        for (int pb = 0; pb < PB; ++pb) {
            acc[pb][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.ah[pb], f.b0h, acc[pb][0], 0, 0, 0);
            acc[pb][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.ah[pb], f.b1h, acc[pb][1], 0, 0, 0);
        }
    };
    using B0 = std::integral_constant<int, 0>;
    using B1 = std::integral_constant<int, 1>;
    using I0 = std::integral_constant<int, 0>;
    const bool three = wv < 4;  // three fragments of half A (the others two)
    // prologue: all 36 fragments and the tile of chunk 0
    if (three) dma_frags(0, I0{}, std::integral_constant<int, 3>{});
    else dma_frags(0, I0{}, std::integral_constant<int, 2>{});
    dma_frags(0, std::integral_constant<int, W2_HALF>{}, std::integral_constant<int, 2>{});
    dma_tile(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    // One chunk.  Top: half B of this chunk (2) and the next chunk's tile (7) are requested; middle: half B has landed (the 7 tile
    // requests are younger), barrier, half A of the next chunk is requested; end: everything has landed, barrier.
    auto chunk = [&](int ch, auto bufc, auto threec) __attribute__((always_inline)) {
        constexpr int CUR = decltype(bufc)::value;
        const int nxt = min(ch + 1, nchunk - 1);
        if (ch > 0) dma_frags(ch, std::integral_constant<int, W2_HALF>{}, std::integral_constant<int, 2>{});
        dma_tile(nxt, CUR ^ 1);
        Frag fa, fb;
        read_tap(bufc, 0, fa);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            Frag& cur = (tap & 1) ? fb : fa;
            Frag& nx = (tap & 1) ? fa : fb;
            if (tap + 1 < 9 && tap != 4) read_tap(bufc, tap + 1, nx);
            mma_tap(cur);
            if (tap == 4) {
                asm volatile("s_waitcnt vmcnt(7) lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                dma_frags(nxt, I0{}, threec);
                read_tap(bufc, 5, nx);
            }
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
    auto all_chunks = [&](auto threec) __attribute__((always_inline)) {
        int ch = 0;
#pragma unroll 1
        for (; ch + 1 < nchunk; ch += 2) {
            chunk(ch, B0{}, threec);
            chunk(ch + 1, B1{}, threec);
        }
        if (ch < nchunk) chunk(ch, B0{}, threec);
    };
    if (three) all_chunks(std::integral_constant<int, 3>{});
    else all_chunks(std::integral_constant<int, 2>{});

    // epilogue: D[pixel][channel]: lane = channel (32 nb + li), pixel = (r & 3) + 8 (r >> 2) + 4 h of the block's 32
    const float* par = reinterpret_cast<const float*>(wsrc + (size_t)nchunk * C16_WBUF);
    const float back = __builtin_ldexpf(1.0f, -eimg);
    const bool raw = par[256] != 0.0f;  // uniform
#pragma unroll
    for (int pb = 0; pb < PB; ++pb) {
        const int pblk = p0 + 32 * (PB * wv + pb);
        if (pblk >= npix) continue;  // wave-uniform
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
            const int chn = 32 * nb + li;
            const float alpha = par[chn], beta2 = par[64 + chn], bias = par[128 + chn], un = par[192 + chn] * back;
#pragma unroll
            for (int rr = 0; rr < 16; ++rr) {
                const int pp = pblk + (rr & 3) + 8 * (rr >> 2) + 4 * h;
                if (pp < npix) {
                    const float sv = acc[pb][nb][rr] * un;
                    const float v = (sv + bias) * alpha + beta2;
                    out[(size_t)pp * 64 + chn] = raw ? v : relu_nan(v);
                }
            }
        }
    }
}

// the pre-cut form serves a launch of at least three heads whose maps fit its tile window
static bool conv16p_serves(int H, int W, int nmaps, int heads) {
    const int np = min(W2_TILE, H * W);
    const int wraps = (W - 1 + np - 1) / W;
    if (np + 2 * (W + 2) + 2 + 2 * wraps > P3_NSLOT) return false;
    return heads >= 3 && (long)cdiv(H * W, W2_TILE) * nmaps * heads >= 512;
}
static size_t conv16p_image_bytes(int in_channels, int H, int W, int nmaps) {
    return (size_t)nmaps * (in_channels / 16) * 4 * (size_t)conv16p_plane_slots(H, W) * 16;
}

static int conv16w_slots(int H, int W) {
    const int np = min(W2_TILE, H * W);
    const int wraps = (W - 1 + np - 1) / W;
    return np + 2 * (W + 2) + 2 + 2 * wraps;
}
// the 512-pixel form serves a launch when the map fits its staging and there is enough work to fill the chip with its (half as many) tiles
static bool conv16w_serves(int H, int W, int nmaps, int heads) {
    if (conv16w_slots(H, W) > W2_NSLOT - 8) return false;
    if (2 * ((min(W2_TILE, H * W) + 2 * W + 2 + 63) / 64) > 8 * W2_NIT) return false;
    return (long)cdiv(H * W, W2_TILE) * nmaps * heads >= 512;
}

// slots one staged tile needs at this map width (see the kernel: np + 2 WT + 2 + 2 x row wraps)
static int conv16_slots(int H, int W) {
    const int np = min(C16_TILE, H * W);
    const int wraps = (W - 1 + np - 1) / W;
    return np + 2 * (W + 2) + 2 + 2 * wraps;
}

}  // namespace shasta

using namespace shasta;

extern "C" int shasta_shared_conv_f16x2_supported(int in_channels, int H, int W) {
    if (in_channels <= 0 || in_channels % 16 || H <= 0 || W <= 0) return 0;
    if ((long)in_channels * H * W >= (1L << 31)) return 0;
    if (conv16_slots(H, W) > C16_NSLOT - 8) return 0;
    return 2 * ((min(C16_TILE, H * W) + 2 * W + 2 + 63) / 64) <= 24;
}

extern "C" size_t shasta_shared_conv_f16x2_packed_bytes(int in_channels) {
    if (in_channels <= 0 || in_channels % 16) return 0;
    return (size_t)(in_channels / 16) * C16_WBUF + C16_PARAMS * sizeof(float);
}

extern "C" int shasta_shared_conv_pack_f16x2(const float* weight, const float* bias, const float* bn_weight, const float* bn_bias,
                                             const float* bn_mean, const float* bn_var, float bn_eps, int in_channels, void* packed,
                                             size_t packed_bytes, shasta_stream_t stream) {
    SHASTA_REQUIRE(weight && bias && bn_weight && bn_bias && bn_mean && bn_var && packed, "shared_conv_pack_f16x2: null pointer");
    SHASTA_REQUIRE(in_channels > 0 && in_channels % 16 == 0, "shared_conv_pack_f16x2: in_channels must be a multiple of 16");
    SHASTA_REQUIRE((uintptr_t)packed % 16 == 0, "shared_conv_pack_f16x2: packed buffer must be 16-byte aligned");
    if (packed_bytes < shasta_shared_conv_f16x2_packed_bytes(in_channels)) {
        set_error_msg("shared_conv_pack_f16x2: packed buffer too small");
        return SHASTA_E_WORKSPACE;
    }
    hipLaunchKernelGGL(conv16_pack_kernel, dim3(64), dim3(256), 0, as_stream(stream), weight, bias, bn_weight, bn_bias, bn_mean, bn_var,
                       bn_eps, in_channels, static_cast<char*>(packed), 0);
    return check_launch("shared_conv_pack_f16x2");
}

extern "C" int shasta_shared_conv_pack_raw_f16x2(const float* weight, const float* bias, int in_channels, void* packed, size_t packed_bytes,
                                                 shasta_stream_t stream) {
    SHASTA_REQUIRE(weight && bias && packed, "shared_conv_pack_raw_f16x2: null pointer");
    SHASTA_REQUIRE(in_channels > 0 && in_channels % 16 == 0, "shared_conv_pack_raw_f16x2: in_channels must be a multiple of 16");
    SHASTA_REQUIRE((uintptr_t)packed % 16 == 0, "shared_conv_pack_raw_f16x2: packed buffer must be 16-byte aligned");
    if (packed_bytes < shasta_shared_conv_f16x2_packed_bytes(in_channels)) {
        set_error_msg("shared_conv_pack_raw_f16x2: packed buffer too small");
        return SHASTA_E_WORKSPACE;
    }
    hipLaunchKernelGGL(conv16_pack_kernel, dim3(64), dim3(256), 0, as_stream(stream), weight, bias, nullptr, nullptr, nullptr, nullptr, 0.0f,
                       in_channels, static_cast<char*>(packed), 1);
    return check_launch("shared_conv_pack_raw_f16x2");
}

extern "C" size_t shasta_shared_conv_multi_workspace_bytes(int B) { return B <= 0 ? 0 : align_up((size_t)2 * B * sizeof(unsigned), 256); }

extern "C" size_t shasta_shared_conv_multi_workspace_bytes_for(int B, int in_channels, int H, int W, int heads, int two_maps) {
    if (B <= 0 || in_channels <= 0 || in_channels % 16 || H <= 0 || W <= 0) return 0;
    const int nmaps = two_maps ? 2 * B : B;
    size_t n = shasta_shared_conv_multi_workspace_bytes(B);
    if (conv16p_serves(H, W, nmaps, heads)) n += conv16p_image_bytes(in_channels, H, W, nmaps);
    return n;
}

static int conv_multi(const float* x, const float* x_prev, int B, int in_channels, int H, int W, const void* packed, size_t head_stride_bytes,
                      int heads, float* const* h_out, float* const* h_out_prev, void* workspace, size_t workspace_bytes, float x_bound,
                      shasta_stream_t stream);

extern "C" int shasta_shared_conv_multi_f32(const float* x, const float* x_prev, int B, int in_channels, int H, int W, const void* packed,
                                            size_t head_stride_bytes, int heads, float* const* h_out, float* const* h_out_prev,
                                            void* workspace, size_t workspace_bytes, shasta_stream_t stream) {
    return conv_multi(x, x_prev, B, in_channels, H, W, packed, head_stride_bytes, heads, h_out, h_out_prev, workspace, workspace_bytes, 0.0f, stream);
}

extern "C" int shasta_shared_conv_multi_bounded_f32(const float* x, const float* x_prev, int B, int in_channels, int H, int W, const void* packed,
                                                    size_t head_stride_bytes, int heads, float* const* h_out, float* const* h_out_prev,
                                                    void* workspace, size_t workspace_bytes, float x_absmax_bound, shasta_stream_t stream) {
    SHASTA_REQUIRE(x_absmax_bound > 0.0f && x_absmax_bound < INFINITY, "shared_conv_multi_bounded: the bound must be positive and finite");
    return conv_multi(x, x_prev, B, in_channels, H, W, packed, head_stride_bytes, heads, h_out, h_out_prev, workspace, workspace_bytes,
                      x_absmax_bound, stream);
}

static int conv_multi(const float* x, const float* x_prev, int B, int in_channels, int H, int W, const void* packed, size_t head_stride_bytes,
                      int heads, float* const* h_out, float* const* h_out_prev, void* workspace, size_t workspace_bytes, float x_bound,
                      shasta_stream_t stream) {
    SHASTA_REQUIRE(x && packed && h_out, "shared_conv_multi: null pointer");
    SHASTA_REQUIRE((x_prev == nullptr) == (h_out_prev == nullptr), "shared_conv_multi: x_prev and h_out_prev go together");
    SHASTA_REQUIRE(heads >= 1 && heads <= C16_MAXH, "shared_conv_multi: 1 to 8 heads per call");
    SHASTA_REQUIRE(B >= 0 && H > 0 && W > 0, "shared_conv_multi: bad size");
    SHASTA_REQUIRE(shasta_shared_conv_f16x2_supported(in_channels, H, W),
                   "shared_conv_multi: shape not served by the fp16 kernel (in_channels % 16, map width; see shasta_shared_conv_f16x2_supported)");
    SHASTA_REQUIRE((uintptr_t)packed % 16 == 0 && head_stride_bytes % 16 == 0, "shared_conv_multi: packed buffer / head stride must be 16-byte aligned");
    SHASTA_REQUIRE(head_stride_bytes >= shasta_shared_conv_f16x2_packed_bytes(in_channels) || heads == 1, "shared_conv_multi: head stride too small");
    if (B == 0) return SHASTA_OK;
    SHASTA_REQUIRE(workspace, "shared_conv_multi: null workspace");
    if (workspace_bytes < shasta_shared_conv_multi_workspace_bytes(B)) {
        set_error_msg("shared_conv_multi: workspace too small");
        return SHASTA_E_WORKSPACE;
    }
    hipStream_t st = as_stream(stream);
    const int nmaps = x_prev ? 2 * B : B;
    const long per_image = (long)in_channels * H * W;
    unsigned* xmax = static_cast<unsigned*>(workspace);
    int rc = SHASTA_OK;
    if (x_bound > 0.0f) {
        // the caller vouches for max |x| <= x_bound (the producer of the maps knows it): every image is cut under the bound's scale and
        // the maps are not read a second time.  The bound need not be tight (a piece pair keeps 22 bits of every element within 2^-17
        // of it); an element beyond 4 x the bound overflows fp16 and comes out as a NaN / Inf, never as a wrong finite number
        unsigned bits;
        memcpy(&bits, &x_bound, sizeof(bits));
        if (hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(xmax), (int)bits, (size_t)nmaps, st) != hipSuccess) return SHASTA_E_LAUNCH;
    } else {
        if (hipMemsetAsync(xmax, 0, (size_t)nmaps * sizeof(unsigned), st) != hipSuccess) return SHASTA_E_LAUNCH;
        int slices = 1;
        while (slices < 1024 && slices * nmaps < 2048 && per_image / (slices * 2) >= 16384) slices *= 2;
        hipLaunchKernelGGL(conv16_absmax_kernel, dim3(slices, nmaps), dim3(256), 0, st, x, x_prev, per_image, B, xmax);
        if ((rc = check_launch("shared_conv_multi (image maxima)")) != SHASTA_OK) return rc;
    }

    Conv16Args a;
    a.x[0] = x;
    a.x[1] = x_prev;
    for (int i = 0; i < C16_MAXH; ++i) {
        a.out[0][i] = i < heads ? h_out[i] : nullptr;
        a.out[1][i] = (i < heads && h_out_prev) ? h_out_prev[i] : nullptr;
        if (i < heads) SHASTA_REQUIRE(a.out[0][i] && (!h_out_prev || a.out[1][i]), "shared_conv_multi: null output pointer");
    }
    a.packed = static_cast<const char*>(packed);
    a.head_stride = head_stride_bytes;
    a.xmax = xmax;
    a.B = B;
    a.Cin = in_channels;
    a.H = H;
    a.W = W;
    a.heads = heads;
#ifndef C16_NO_PRECUT
    // several heads over many maps and a workspace that holds the piece image: the input is cut once for all of them
    if (conv16p_serves(H, W, nmaps, heads) &&
        workspace_bytes >= shasta_shared_conv_multi_workspace_bytes(B) + conv16p_image_bytes(in_channels, H, W, nmaps) &&
        hipFuncSetAttribute((const void*)shared_conv_f16p_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, P3_LDS) == hipSuccess) {
        Conv16pArgs pa;
        pa.plane_slots = conv16p_plane_slots(H, W);
        char* img = static_cast<char*>(workspace) + shasta_shared_conv_multi_workspace_bytes(B);
        pa.img = img;
        hipLaunchKernelGGL(conv16_precut_kernel, dim3((unsigned)cdiv((int)pa.plane_slots, 256), 2 * (in_channels / 16), nmaps), dim3(256), 0, st, x, x_prev, B,
                           in_channels, H, W, xmax, reinterpret_cast<w32x4*>(img), pa.plane_slots);
        if ((rc = check_launch("shared_conv_multi (piece image)")) != SHASTA_OK) return rc;
        a.tiles_per_map = cdiv(H * W, W2_TILE);
        a.ntiles = a.tiles_per_map * nmaps;
        a.tiles_per_xcd = cdiv(a.ntiles, 8);
        pa.c = a;
        hipLaunchKernelGGL(shared_conv_f16p_kernel, dim3(8 * a.tiles_per_xcd * heads), dim3(512), P3_LDS, st, pa);
        return check_launch("shared_conv_f16p");
    }
    (void)hipGetLastError();
#endif
#ifndef C16_NO_WIDE
    if (conv16w_serves(H, W, nmaps, heads)) {
        a.tiles_per_map = cdiv(H * W, W2_TILE);
        a.ntiles = a.tiles_per_map * nmaps;
        a.tiles_per_xcd = cdiv(a.ntiles, 8);
        if (hipFuncSetAttribute((const void*)shared_conv_f16w_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, W2_LDS) == hipSuccess) {
            hipLaunchKernelGGL(shared_conv_f16w_kernel, dim3(8 * a.tiles_per_xcd * heads), dim3(512), W2_LDS, st, a);
            return check_launch("shared_conv_f16w");
        }
        (void)hipGetLastError();  // a device that grants less LDS: the 256-pixel form below
    }
#endif
    a.tiles_per_map = cdiv(H * W, C16_TILE);
    a.ntiles = a.tiles_per_map * nmaps;
    a.tiles_per_xcd = cdiv(a.ntiles, 8);
#ifndef C16_PB
#define C16_PB 1
#endif
    (void)hipFuncSetAttribute((const void*)shared_conv_f16_kernel<C16_PB>, hipFuncAttributeMaxDynamicSharedMemorySize, C16_LDS);
    hipLaunchKernelGGL(shared_conv_f16_kernel<C16_PB>, dim3(8 * a.tiles_per_xcd * heads), dim3(512 / C16_PB), C16_LDS, st, a);
    return check_launch("shared_conv_f16");
}
