// K3a: first aug_shape layer (det3d/models/tracker/shasta.py:54, applied :241-244) for batches of 2..32 frame-pairs:
//   part[ks][b][n] = sum_{k in chunk ks} W[n][k] * x[b][k]       W: 4 x (N*F/64, N*F) fp32, 4.1 GB at N=500,F=256
// The weights are streamed from HBM exactly once per 32 batch items; the kernel is a weight-streaming skinny GEMM on the
// matrix cores with the 32 weight rows of a wave as the A operand and up to 32 (or 16) batch items as the B operand:
//   v_mfma_f32_32x32x2_f32 (B in 17..32): 16 MFMA = 1024 SIMD cycles per 4 KB weight tile  -> 16 B/clk/CU consumable
//   v_mfma_f32_16x16x4_f32 (B <= 16)    : 16 MFMA =  512 SIMD cycles per 4 KB weight tile  -> 32 B/clk/CU consumable
// against ~10 B/clk/CU that HBM delivers, so the kernel stays HBM-bound for every batch size it serves.
// Data path: global_load_lds_dwordx4 (LDS-DMA, no VGPR round trip): each wave-instruction moves 8 rows x 128 B, i.e.
// whole 128-byte lines, into a lane-linear [32 rows][8 x 16 B] LDS image.  The 16-byte chunk c of row r is fetched into
// chunk position c ^ ((r>>1)&7) (swizzle applied on the SOURCE address, the LDS destination of an LDS-DMA cannot be
// scattered) so that the ds_read_b128 fragment reads (16 different rows per lane group) hit distinct banks.
// Every wave owns a private ring of NS tiles and runs ahead of its own MFMAs by NS-1 tiles with counted
// s_waitcnt vmcnt(N): no workgroup barrier anywhere.  Split-K partials are reduced afterwards in a fixed order.
#include "common.hpp"
#include <stdlib.h>

namespace shasta {

struct AnchorMfmaArgs {
    const float* W[4];
    const float* x[2];
    float* part;
    int H, K, B, KS, Kc, x_batch_stride, groups_per_mlp;
};

#define GLDS16(gsrc, ldst)                                                                         \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gsrc),        \
                                     (__attribute__((address_space(3))) void*)(ldst), 16, 0, 0)
// weights are read exactly once per launch: non-temporal (aux = 2) keeps the activation vectors and the small
// weights of the following kernels resident in L2 / Infinity Cache (MI355X_MICROARCH.md, row nt-weights)
#define GLDS16_NT(gsrc, ldst)                                                                      \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gsrc),        \
                                     (__attribute__((address_space(3))) void*)(ldst), 16, 0, 2)

template <int N>
__device__ __forceinline__ void wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// XT == 0: 16 batch rows per pass with 16x16x4 MFMA; XT == 1 / 2: 32 / 64 batch rows per pass with 32x32x2 MFMA
// (XT accumulators share every weight fragment; at 64 rows the kernel is MFMA-bound, 2048 SIMD cycles per 4 KB tile).
template <int XT, int NS>
__global__ __launch_bounds__(256) void anchor_l1_mfma_kernel(AnchorMfmaArgs a) {
    constexpr bool WIDE = XT > 0;
    constexpr int XR = WIDE ? 32 * XT : 16;     // batch rows staged per tile
    constexpr int XI = XR / 8;                  // LDS-DMA instructions per x tile
    constexpr int SLOT = 1024 + XR * 32;        // floats per ring slot: W tile (32 x 32) + x tile (XR x 32)
    constexpr int PER_TILE = 4 + XI;            // vmcnt units per tile
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // provably wave-uniform -> scalar control flow
    float* ring = lds + (size_t)wid * NS * SLOT;

    const int item = blockIdx.x * 4 + wid;
    const int G = 4 * a.groups_per_mlp;  // 32-row blocks over the four MLPs
    if (item >= G * a.KS) return;
    const int ks = item / G, g = item % G;
    const int mlp = g / a.groups_per_mlp, r0 = (g % a.groups_per_mlp) * 32;
    const int b0 = blockIdx.y * XR;
    const int kbeg = ks * a.Kc, kend = min(a.K, kbeg + a.Kc);
    const int NT = (kend - kbeg) >> 5;  // 32-float tiles in this chunk

    // staging roles: instruction j covers rows 8j..8j+7, lane -> (row 8j + lane/8, chunk position lane%8)
    const float* wsrc[4];
    const float* xsrc[XI];
    const float* wbase = mlp == 0 ? a.W[0] : mlp == 1 ? a.W[1] : mlp == 2 ? a.W[2] : a.W[3];
    const float* xbase = mlp < 2 ? a.x[0] : a.x[1];
    {
        const int cpos = lane & 7, rl = lane >> 3;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int r = 8 * j + rl;
            const int c = cpos ^ ((r >> 1) & 7);
            wsrc[j] = wbase + (size_t)min(r0 + r, a.H - 1) * a.K + kbeg + 4 * c;
        }
#pragma unroll
        for (int j = 0; j < XI; ++j) {
            const int r = 8 * j + rl;  // row of the staged x tile (0..XR-1); sub-tile u = r / 32 keeps the same swizzle
            const int c = cpos ^ (((r & 31) >> 1) & 7);
            xsrc[j] = xbase + (size_t)min(b0 + r, a.B - 1) * a.x_batch_stride + kbeg + 4 * c;
        }
    }
    auto issue = [&](int t) {
        float* s = ring + (t % NS) * SLOT;
        const int ko = t * 32;
#pragma unroll
        for (int j = 0; j < 4; ++j) GLDS16_NT(wsrc[j] + ko, s + j * 256);
#pragma unroll
        for (int j = 0; j < XI; ++j) GLDS16(xsrc[j] + ko, s + 1024 + j * 256);
    };

    int issued = 0;
#pragma unroll 1
    for (; issued < NS - 1 && issued < NT; ++issued) issue(issued);

    f32x16 acc32[WIDE ? XT : 1];
#pragma unroll
    for (int u = 0; u < (WIDE ? XT : 1); ++u) acc32[u] = f32x16{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    f32x4 acc16[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};

#pragma unroll 1
    for (int t = 0; t < NT; ++t) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // fragment reads of tile t-1 retired before its slot is refilled
        if (issued < NT) {
            issue(issued);
            ++issued;
        }
        // tiles issued after tile t may stay in flight
        const int ahead = issued - (t + 1);
        if (ahead >= NS - 1) wait_vm<PER_TILE*(NS - 1)>();
        else if (ahead == 2 && NS > 3) wait_vm<PER_TILE * 2>();
        else if (ahead == 1) wait_vm<PER_TILE>();
        else if (ahead == 0) wait_vm<0>();
        else if (ahead == 3 && NS > 4) wait_vm<PER_TILE * 3>();
        else if (ahead == 4 && NS > 5) wait_vm<PER_TILE * 4>();
        else wait_vm<0>();
        const float* s = ring + (t % NS) * SLOT;
        if constexpr (WIDE) {
            const int i = lane & 31, h = lane >> 5, f = (i >> 1) & 7;
            f32x4 wv[4], xv[XT > 0 ? XT : 1][4];
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const int pos = (2 * m + h) ^ f;
                wv[m] = *reinterpret_cast<const f32x4*>(s + i * 32 + pos * 4);
#pragma unroll
                for (int u = 0; u < XT; ++u) xv[u][m] = *reinterpret_cast<const f32x4*>(s + 1024 + u * 1024 + i * 32 + pos * 4);
            }
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int u = 0; u < XT; ++u)
                        acc32[u] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[m][q], xv[u][m][q], acc32[u], 0, 0, 0);
        } else {
            const int i = lane & 15, kq = lane >> 4;
            f32x4 wv[2][2], xv[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int c = 4 * u + kq;
                xv[u] = *reinterpret_cast<const f32x4*>(s + 1024 + i * 32 + (c ^ ((i >> 1) & 7)) * 4);
#pragma unroll
                for (int rb = 0; rb < 2; ++rb) {
                    const int row = 16 * rb + i;
                    wv[rb][u] = *reinterpret_cast<const f32x4*>(s + row * 32 + (c ^ ((row >> 1) & 7)) * 4);
                }
            }
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int rb = 0; rb < 2; ++rb)
                        acc16[rb] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[rb][u][q], xv[u][q], acc16[rb], 0, 0, 0);
        }
    }
    // D[i = weight row][j = batch item]
    if constexpr (WIDE) {
        const int h = lane >> 5;
#pragma unroll
        for (int u = 0; u < XT; ++u) {
            const int b = b0 + 32 * u + (lane & 31);
            if (b < a.B) {
                float* o = a.part + ((size_t)ks * a.B + b) * (4 * a.H) + mlp * a.H;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = r0 + (r & 3) + 8 * (r >> 2) + 4 * h;
                    if (row < a.H) o[row] = acc32[u][r];
                }
            }
        }
    } else {
        const int b = b0 + (lane & 15), kq = lane >> 4;
        if (b < a.B) {
            float* o = a.part + ((size_t)ks * a.B + b) * (4 * a.H) + mlp * a.H;
#pragma unroll
            for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = r0 + 16 * rb + 4 * kq + r;
                    if (row < a.H) o[row] = acc16[rb][r];
                }
        }
    }
}

// returns 0 when launched, 1 when the shape is not served by this kernel
int launch_anchor_l1_mfma(const float* const W[4], const float* feat, const float* prev_feat, float* part, int H, int K,
                          int B, int x_batch_stride, int* ks_out, hipStream_t st) {
    AnchorMfmaArgs a;
    for (int i = 0; i < 4; ++i) a.W[i] = W[i];
    a.x[0] = feat;
    a.x[1] = prev_feat;
    a.part = part;
    a.H = H;
    a.K = K;
    a.B = B;
    a.x_batch_stride = x_batch_stride;
    a.groups_per_mlp = cdiv(H, 32);
    const int G = 4 * a.groups_per_mlp;
    const int tiles = K / 32;
    // One workgroup (4 waves, one per SIMD) per CU because of the LDS ring: pick the K split whose workgroup count fills
    // whole rounds of the CU array best (e.g. 252 workgroups on 256 CUs, not 315 = one full round + a 23 % one).
    int ncu = 256;
    {
        int dev = 0;
        if (hipGetDevice(&dev) == hipSuccess) {
            int v = 0;
            if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ncu = v;
        }
    }
    int ks = 1;
    double best = -1.0;
    for (int c = 1; c <= 64 && c * 8 <= tiles; ++c) {
        const int wgs = cdiv(G * cdiv(tiles, cdiv(tiles, c)), 4);
        const int rounds = cdiv(wgs, ncu);
        const double eff = (double)wgs / ((double)rounds * ncu);
        if (eff > best + 1e-9) {
            best = eff;
            ks = c;
        }
    }
    const int tiles_per = cdiv(tiles, ks);
    a.Kc = tiles_per * 32;
    a.KS = cdiv(tiles, tiles_per);
    *ks_out = a.KS;
    auto launch = [&](auto kern, int ns, int xr) {
        const size_t lds = (size_t)4 * ns * (1024 + xr * 32) * sizeof(float);
        (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(kern, dim3(cdiv(G * a.KS, 4), cdiv(B, xr)), dim3(256), lds, st, a);
    };
    static const bool no64 = getenv("SHASTA_L1_NO64") != nullptr;
    if (B <= 16) launch(anchor_l1_mfma_kernel<0, 6>, 6, 16);
    else if (B <= 32 || no64) launch(anchor_l1_mfma_kernel<1, 4>, 4, 32);
    else launch(anchor_l1_mfma_kernel<2, 3>, 3, 64);
    return 0;
}

}  // namespace shasta
