// K0: shared_conv of the affinity network (det3d/models/tracker/shasta.py:42-47, applied :223-228):
//   Conv2d(Cin -> 64, 3x3, padding 1, bias) -> BatchNorm2d(64) in eval mode -> ReLU -> permute to NHWC.
// Input: the neck output (B, Cin, H, W) fp32 NCHW (Cin = 512, H = W = 180 in the shipped configs); output
// (B, H, W, 64) fp32 NHWC = example['bev_feature'].  19.1 GFLOP per map: the largest dense op of a real nuScenes forward.
//
// Implicit GEMM on the matrix cores (v_mfma_f32_32x32x2_f32, exact fp32): M = pixels, N = 64 output channels,
// K = Cin*9.  A workgroup owns 4 image rows x 32 columns x all 64 channels; wave w owns row w: two 32x32 accumulators
// (channels 0-31, 32-63) that share every A fragment.  K is walked in chunks of 8 input channels: the (6 x 34 x 8) input
// halo tile and the (72 x 64) weight slice are staged through LDS (double buffered, next chunk's global loads are issued
// before the current chunk's 72 MFMAs and written to LDS after them).  The two k values of one MFMA are the same filter
// tap of channels c and c+4, so that every LDS address is a per-lane constant plus an immediate offset.
// Weights are pre-packed once as [chunk][k = c_local*9 + tap][n] (a straight 18 KB copy per chunk); BatchNorm is applied
// in the epilogue exactly as PyTorch's eval kernel does: y = (acc + bias) * alpha + beta', alpha = gamma / sqrt(var + eps),
// beta' = beta - mean * alpha.
#include "common.hpp"

namespace shasta {

constexpr int CV_CK = 8;                       // input channels per chunk
constexpr int CV_TR = 4, CV_TC = 32;           // tile rows / columns
constexpr int CV_IN = CV_CK * (CV_TR + 2) * (CV_TC + 2);  // 1632 floats
constexpr int CV_WT = CV_CK * 9 * 64;          // 4608 floats
constexpr int CV_IN_PT = (CV_IN + 255) / 256;  // 7 staged input floats per thread
constexpr int CV_WT_PT = (CV_WT / 4 + 255) / 256;  // 5 staged weight float4 per thread

// packed layout: [Cin/8][72][64] weights, then alpha[64], beta'[64], bias[64], then [192] = 1.0 for a RAW pack (conv + bias, no BN, no ReLU)
__global__ void shared_conv_pack_kernel(const float* __restrict__ w, const float* __restrict__ bias,
                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                        const float* __restrict__ mean, const float* __restrict__ var, float eps, int Cin,
                                        float* __restrict__ out, int raw) {
    const int tid = blockIdx.x * blockDim.x + threadIdx.x, nth = gridDim.x * blockDim.x;
    const int total = Cin * 9 * 64;
    for (int e = tid; e < total; e += nth) {
        const int n = e & 63, k = (e >> 6) % 72, chunk = (e >> 6) / 72;
        const int c = chunk * CV_CK + k / 9, tap = k % 9;
        out[e] = w[((size_t)n * Cin + c) * 9 + tap];
    }
    for (int n = tid; n < 64; n += nth) {
        const float alpha = raw ? 1.0f : (1.0f / sqrtf(var[n] + eps)) * gamma[n];
        out[total + n] = alpha;
        out[total + 64 + n] = raw ? 0.0f : beta[n] - mean[n] * alpha;
        out[total + 128 + n] = bias[n];
        if (n == 0) out[total + 192] = raw ? 1.0f : 0.0f;
    }
}

// grid.z runs over the images of BOTH maps of a frame pair (current: z < B, previous: z >= B), so one launch fills the
// chip with twice as many workgroups.
__global__ __launch_bounds__(256) void shared_conv_kernel(const float* __restrict__ xa, const float* __restrict__ xb,
                                                          const float* __restrict__ packed, float* __restrict__ outa,
                                                          float* __restrict__ outb, int B, int Cin, int H, int W) {
    __shared__ __attribute__((aligned(16))) float s_in[2][CV_IN];
    __shared__ __attribute__((aligned(16))) float s_w[2][CV_WT];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int x0 = blockIdx.x * CV_TC, y0 = blockIdx.y * CV_TR;
    const bool second = (int)blockIdx.z >= B;
    const int b = second ? blockIdx.z - B : blockIdx.z;
    const float* x = second ? xb : xa;
    float* out = second ? outb : outa;
    const int nchunk = Cin / CV_CK;
    const float* xin = x + (size_t)b * Cin * H * W;

    // staging roles (fixed per thread): input element e -> (channel, tile row, tile col)
    int in_off[CV_IN_PT];   // offset inside one channel-chunk of the input, or -1 when outside the image / tile
    int in_dst[CV_IN_PT];
#pragma unroll
    for (int j = 0; j < CV_IN_PT; ++j) {
        const int e = tid + 256 * j;
        in_dst[j] = e < CV_IN ? e : -1;
        const int c = e / ((CV_TR + 2) * (CV_TC + 2)), rem = e % ((CV_TR + 2) * (CV_TC + 2));
        const int r = rem / (CV_TC + 2), col = rem % (CV_TC + 2);
        const int gy = y0 - 1 + r, gx = x0 - 1 + col;
        in_off[j] = (e < CV_IN && gy >= 0 && gy < H && gx >= 0 && gx < W) ? (c * H + gy) * W + gx : -1;
    }
    float rin[CV_IN_PT];
    f32x4 rwt[CV_WT_PT];
    auto load_chunk = [&](int ch) {
        const float* xc = xin + (size_t)ch * CV_CK * H * W;
#pragma unroll
        for (int j = 0; j < CV_IN_PT; ++j) rin[j] = in_off[j] >= 0 ? xc[in_off[j]] : 0.0f;
        const f32x4* wc = reinterpret_cast<const f32x4*>(packed + (size_t)ch * CV_WT);
#pragma unroll
        for (int j = 0; j < CV_WT_PT; ++j) {
            const int e = tid + 256 * j;
            rwt[j] = e < CV_WT / 4 ? wc[e] : f32x4{0, 0, 0, 0};
        }
    };
    auto store_chunk = [&](int buf) {
#pragma unroll
        for (int j = 0; j < CV_IN_PT; ++j)
            if (in_dst[j] >= 0) s_in[buf][in_dst[j]] = rin[j];
        f32x4* wd = reinterpret_cast<f32x4*>(s_w[buf]);
#pragma unroll
        for (int j = 0; j < CV_WT_PT; ++j) {
            const int e = tid + 256 * j;
            if (e < CV_WT / 4) wd[e] = rwt[j];
        }
    };

    f32x16 acc0 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    f32x16 acc1 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const int i = lane & 31, h = lane >> 5;
    // A: s_in[c][row][col] with c = c_lo + 4h, row = wv + ky, col = i + kx ; B: s_w[k + 36h][32*nb + i]
    const int a_base = (h * 4) * ((CV_TR + 2) * (CV_TC + 2)) + wv * (CV_TC + 2) + i;
    const int b_base = (h * 36) * 64 + i;

    load_chunk(0);
    store_chunk(0);
    __syncthreads();
    for (int ch = 0; ch < nchunk; ++ch) {
        const int cur = ch & 1;
        if (ch + 1 < nchunk) load_chunk(ch + 1);
        const float* ain = s_in[cur] + a_base;
        const float* bw = s_w[cur] + b_base;
#pragma unroll
        for (int kp = 0; kp < 36; ++kp) {
            const int c_lo = kp / 9, tap = kp % 9, ky = tap / 3, kx = tap % 3;
            const float av = ain[c_lo * ((CV_TR + 2) * (CV_TC + 2)) + ky * (CV_TC + 2) + kx];
            const float b0 = bw[kp * 64], b1 = bw[kp * 64 + 32];
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b0, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b1, acc1, 0, 0, 0);
        }
        if (ch + 1 < nchunk) {
            store_chunk(cur ^ 1);  // the other buffer was last read in iteration ch-1, before the barrier below
            __syncthreads();
        }
    }
    // epilogue: D[pixel i][channel j]; lane j = lane&31, pixel = (r&3) + 8*(r>>2) + 4*h
    const int gy = y0 + wv;
    if (gy >= H) return;
    const float* par = packed + (size_t)Cin * 9 * 64;
    const bool raw = par[192] != 0.0f;  // uniform
    float* orow = out + (((size_t)b * H + gy) * W) * 64;
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
        const int chn = 32 * nb + (lane & 31);
        const float alpha = par[chn], beta2 = par[64 + chn], bias = par[128 + chn];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int px = x0 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (px < W) {
                const float a = nb ? acc1[r] : acc0[r];
                const float v = (a + bias) * alpha + beta2;
                orow[(size_t)px * 64 + chn] = raw ? v : relu_nan(v);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Flattened-pixel variant (maps up to 256 columns wide, i.e. every shipped config): a workgroup owns 128 CONSECUTIVE
// pixels of the flattened (y*W + x) image instead of a 4 x 32 rectangle, each wave 32 of them.  No column padding
// (180 = 5.6 x 32 wasted 6 % in the rectangular tiling) and 254 workgroups per 180 x 180 map instead of 270, so one map
// is exactly one wave per SIMD and a frame pair (two maps) two.  The staged input tile covers the <= 128/W + 2 image
// rows the pixels touch plus a one-pixel halo, full width; K chunks are 4 channels (k pair = same tap of channels c and
// c + 2) so that the double-buffered tiles (2 x (11.6 + 9.2) KB at W = 180) leave room for 3 workgroups per CU.
// The packed weights are the same buffer: [Cin/8][72][64] read as [Cin/4][36][64].
// ------------------------------------------------------------------------------------------------------------------
constexpr int CF_CK = 4;
constexpr int CF_WT = CF_CK * 9 * 64;  // 2304 floats per chunk
constexpr int CF_IN_PT = 18;           // staged input floats per thread, upper bound (W <= 256)

// IN_PT = staged input floats per thread for this map width (ceil(4 * rows * (W + 2) / 256)).
// The f32 MFMA does not overlap VALU work, so the chunk loop is kept free of it: the halo / padding decisions are taken once
// (clamped 32-bit offsets + a select per element instead of predicated loads), the loads use the scalar chunk base + a
// per-lane offset, and the steady loop has no conditional around the accumulators (the last chunk is peeled; with the
// condition inside, the register allocator moved all 32 accumulators through VGPRs every chunk).  Before: 233 VALU
// instructions per 36 MFMAs.
template <int IN_PT>
__global__ __launch_bounds__(256) void shared_conv_flat_kernel(const float* __restrict__ xa, const float* __restrict__ xb,
                                                               const float* __restrict__ packed, float* __restrict__ outa,
                                                               float* __restrict__ outb, int B, int Cin, int H, int W,
                                                               int rt_max) {
    extern __shared__ __attribute__((aligned(16))) float s_cf[];
    const int WT = W + 2;
    const int in_cap = 256 * IN_PT;                   // floats per input buffer (>= CF_CK * rt_max * WT)
    float* s_in0 = s_cf;
    float* s_in1 = s_cf + in_cap;
    float* s_w0 = s_cf + 2 * in_cap;
    float* s_w1 = s_w0 + CF_WT;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool second = (int)blockIdx.z >= B;
    const int b = second ? blockIdx.z - B : blockIdx.z;
    const float* x = second ? xb : xa;
    float* out = second ? outb : outa;
    const int npix = H * W, p0 = blockIdx.x * 128;
    const int y_first = p0 / W, y_last = min(H - 1, (min(p0 + 127, npix - 1)) / W);
    const int RT = y_last - y_first + 3;             // rows of the tile incl. halo
    const int plane = RT * WT;
    const int nchunk = Cin / CF_CK;
    const float* xin = x + (size_t)b * Cin * npix;

    unsigned in_off[IN_PT];  // element offset inside one 4-channel chunk (clamped to 0 where the tile is outside the image)
    bool in_ok[IN_PT];
#pragma unroll
    for (int j = 0; j < IN_PT; ++j) {
        const int e = tid + 256 * j;
        int off = -1;
        if (e < CF_CK * plane) {
            const int c = e / plane, rem = e - c * plane;
            const int r = rem / WT, col = rem - r * WT;
            const int gy = y_first - 1 + r, gx = col - 1;
            if (gy >= 0 && gy < H && gx >= 0 && gx < W) off = c * npix + gy * W + gx;
        }
        in_ok[j] = off >= 0;
        in_off[j] = (unsigned)max(off, 0);
    }
    float rin[IN_PT];
    f32x4 rwt[3];
    auto load_chunk = [&](int ch) {
        const float* xc = xin + (size_t)ch * CF_CK * npix;  // wave-uniform
#pragma unroll
        for (int j = 0; j < IN_PT; ++j) {
            const float v = xc[in_off[j]];  // always a valid address
            rin[j] = in_ok[j] ? v : 0.0f;
        }
        const f32x4* wc = reinterpret_cast<const f32x4*>(packed + (size_t)ch * CF_WT);
        rwt[0] = wc[tid];
        rwt[1] = wc[tid + 256];
        rwt[2] = wc[min(tid, 63) + 512];
    };
    auto store_chunk = [&](float* si, float* sw) {
#pragma unroll
        for (int j = 0; j < IN_PT; ++j) si[tid + 256 * j] = rin[j];  // the buffer holds all 256 * IN_PT slots
        reinterpret_cast<f32x4*>(sw)[tid] = rwt[0];
        reinterpret_cast<f32x4*>(sw)[tid + 256] = rwt[1];
        if (tid < 64) reinterpret_cast<f32x4*>(sw)[tid + 512] = rwt[2];
    };

    f32x16 acc0 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    f32x16 acc1 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const int i = lane & 31, h = lane >> 5;
    const int p = min(p0 + 32 * wv + i, npix - 1);
    const int py = p / W, px = p - py * W;
    // A: tile[c = c_lo + 2h][py - y_first + 1 + dy][px + 1 + dx] ; B: s_w[kp + 18h][32*nb + i]
    const int a_base = (2 * h) * plane + (py - y_first + 1) * WT + px + 1;
    const int b_base = (18 * h) * 64 + i;
    auto compute = [&](const float* si, const float* sw) {
        const float* ain = si + a_base;
        const float* bw = sw + b_base;
#pragma unroll
        for (int kp = 0; kp < 18; ++kp) {
            const int c_lo = kp / 9, tap = kp % 9, dy = tap / 3 - 1, dx = tap % 3 - 1;
            const float av = ain[c_lo * plane + dy * WT + dx];
            const float b0 = bw[kp * 64], b1 = bw[kp * 64 + 32];
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b0, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b1, acc1, 0, 0, 0);
        }
    };

    load_chunk(0);
    store_chunk(s_in0, s_w0);
    __syncthreads();
    // steady state: two chunks per trip so that the buffer roles are compile-time; every trip stages two more chunks
    int ch = 0;
#pragma unroll 1
    for (; ch + 2 < nchunk; ch += 2) {
        load_chunk(ch + 1);
        compute(s_in0, s_w0);
        store_chunk(s_in1, s_w1);
        __syncthreads();
        load_chunk(ch + 2);
        compute(s_in1, s_w1);
        store_chunk(s_in0, s_w0);
        __syncthreads();
    }
    // tail: one or two chunks left, the first of them is in buffer 0
    if (ch + 1 < nchunk) {
        load_chunk(ch + 1);
        compute(s_in0, s_w0);
        store_chunk(s_in1, s_w1);
        __syncthreads();
        compute(s_in1, s_w1);
    } else {
        compute(s_in0, s_w0);
    }
    const float* par = packed + (size_t)Cin * 9 * 64;
    const bool raw = par[192] != 0.0f;  // uniform
    float* obase = out + (size_t)b * npix * 64;
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
        const int chn = 32 * nb + (lane & 31);
        const float alpha = par[chn], beta2 = par[64 + chn], bias = par[128 + chn];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int pp = p0 + 32 * wv + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (pp < npix) {
                const float a = nb ? acc1[r] : acc0[r];
                const float v = (a + bias) * alpha + beta2;
                obase[(size_t)pp * 64 + chn] = raw ? v : relu_nan(v);
            }
        }
    }
}

}  // namespace shasta

using namespace shasta;

extern "C" size_t shasta_shared_conv_packed_bytes(int in_channels) {
    if (in_channels <= 0 || in_channels % CV_CK) return 0;
    return ((size_t)in_channels * 9 * 64 + 256) * sizeof(float);
}

extern "C" int shasta_shared_conv_pack_f32(const float* weight, const float* bias, const float* bn_weight,
                                           const float* bn_bias, const float* bn_mean, const float* bn_var, float bn_eps,
                                           int in_channels, void* packed, size_t packed_bytes, shasta_stream_t stream) {
    SHASTA_REQUIRE(weight && bias && bn_weight && bn_bias && bn_mean && bn_var && packed, "shared_conv_pack: null pointer");
    SHASTA_REQUIRE(in_channels > 0 && in_channels % CV_CK == 0, "shared_conv: in_channels must be a multiple of 8");
    SHASTA_REQUIRE((uintptr_t)packed % 16 == 0, "shared_conv_pack: packed buffer must be 16-byte aligned");
    if (packed_bytes < shasta_shared_conv_packed_bytes(in_channels)) {
        set_error_msg("shared_conv_pack: packed buffer too small");
        return SHASTA_E_WORKSPACE;
    }
    hipLaunchKernelGGL(shared_conv_pack_kernel, dim3(256), dim3(256), 0, as_stream(stream), weight, bias, bn_weight, bn_bias,
                       bn_mean, bn_var, bn_eps, in_channels, static_cast<float*>(packed), 0);
    return check_launch("shared_conv_pack");
}

extern "C" int shasta_shared_conv_pack_raw_f32(const float* weight, const float* bias, int in_channels, void* packed, size_t packed_bytes,
                                               shasta_stream_t stream) {
    SHASTA_REQUIRE(weight && bias && packed, "shared_conv_pack_raw: null pointer");
    SHASTA_REQUIRE(in_channels > 0 && in_channels % CV_CK == 0, "shared_conv_pack_raw: in_channels must be a multiple of 8");
    SHASTA_REQUIRE((uintptr_t)packed % 16 == 0, "shared_conv_pack_raw: packed buffer must be 16-byte aligned");
    if (packed_bytes < shasta_shared_conv_packed_bytes(in_channels)) {
        set_error_msg("shared_conv_pack_raw: packed buffer too small");
        return SHASTA_E_WORKSPACE;
    }
    hipLaunchKernelGGL(shared_conv_pack_kernel, dim3(256), dim3(256), 0, as_stream(stream), weight, bias, nullptr, nullptr, nullptr, nullptr,
                       0.0f, in_channels, static_cast<float*>(packed), 1);
    return check_launch("shared_conv_pack_raw");
}

extern "C" int shasta_shared_conv_f32(const float* x, const float* x_prev, int B, int in_channels, int H, int W,
                                      const void* packed, float* out, float* out_prev, shasta_stream_t stream) {
    SHASTA_REQUIRE(x && packed && out, "shared_conv: null pointer");
    SHASTA_REQUIRE((x_prev == nullptr) == (out_prev == nullptr), "shared_conv: x_prev and out_prev go together");
    SHASTA_REQUIRE(B >= 0 && H > 0 && W > 0, "shared_conv: bad size");
    SHASTA_REQUIRE(in_channels > 0 && in_channels % CV_CK == 0, "shared_conv: in_channels must be a multiple of 8");
    SHASTA_REQUIRE((uintptr_t)packed % 16 == 0, "shared_conv: packed buffer must be 16-byte aligned");
    SHASTA_REQUIRE((long)in_channels * H * W < (1L << 31), "shared_conv: one image exceeds 2^31 elements");
    if (B == 0) return SHASTA_OK;
    const int rt_max = min(H, 128 / W + 2) + 2;
    const int in_need = cdiv(CF_CK * rt_max * (W + 2), 256);  // staged input floats per thread
    const int in_pt = in_need <= 8 ? 8 : in_need <= 12 ? 12 : in_need <= 14 ? 14 : in_need <= 16 ? 16 : in_need <= 18 ? 18 : 19;  // template value (19: does not fit)
    const size_t lds = ((size_t)2 * 256 * in_pt + 2 * CF_WT) * sizeof(float);
    if (W <= 256 && in_pt <= CF_IN_PT && lds <= 64 * 1024) {
        dim3 grid(cdiv(H * W, 128), 1, x_prev ? 2 * B : B);
        const float* pk = static_cast<const float*>(packed);
#define SHASTA_CONV_FLAT(PT) \
    hipLaunchKernelGGL(shared_conv_flat_kernel<PT>, grid, dim3(256), lds, as_stream(stream), x, x_prev, pk, out, out_prev, B, in_channels, H, W, rt_max)
        if (in_pt == 8) SHASTA_CONV_FLAT(8);
        else if (in_pt == 12) SHASTA_CONV_FLAT(12);
        else if (in_pt == 14) SHASTA_CONV_FLAT(14);
        else if (in_pt == 16) SHASTA_CONV_FLAT(16);
        else SHASTA_CONV_FLAT(18);
#undef SHASTA_CONV_FLAT
        return check_launch("shared_conv_flat");
    }
    dim3 grid(cdiv(W, CV_TC), cdiv(H, CV_TR), x_prev ? 2 * B : B);
    hipLaunchKernelGGL(shared_conv_kernel, grid, dim3(256), 0, as_stream(stream), x, x_prev, static_cast<const float*>(packed),
                       out, out_prev, B, in_channels, H, W);
    return check_launch("shared_conv");
}
