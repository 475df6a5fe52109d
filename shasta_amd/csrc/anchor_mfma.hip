// K3a: first aug_shape layer (det3d/models/tracker/shasta.py:54, applied :241-244) for batches of 2 and more frame-pairs:
//   part[ks][b][n] = sum_{k in chunk ks} W[n][k] * x[b][k]       W: 4 x (N*F/64, N*F) fp32, 4.1 GB at N=500,F=256
// The weights are streamed from HBM exactly once per pass of 16 / 32 / 64 batch items; the kernel is a weight-streaming skinny
// GEMM on the matrix cores with the 32 weight rows of a wave as the A operand and the batch items as the B operand:
//   v_mfma_f32_32x32x2_f32 (B in 17..32): 16 MFMA = 1024 SIMD cycles per 4 KB weight tile  -> 16 B/clk/CU consumable
//   the same with two accumulators (B > 32, 64 rows per pass): 32 MFMA = 2048 cycles per tile -> 8 B/clk/CU: matrix-pipe bound
//   v_mfma_f32_16x16x4_f32 (B <= 16)    : 16 MFMA =  512 SIMD cycles per 4 KB weight tile  -> 32 B/clk/CU consumable
// against ~10 B/clk/CU that HBM delivers: HBM-bound up to 32 batch items per pass, matrix-pipe bound at 64.
// Data path: global_load_lds_dwordx4 (LDS-DMA, no VGPR round trip): each wave-instruction moves 8 rows x 128 B, i.e.
// whole 128-byte lines, into a lane-linear [32 rows][8 x 16 B] LDS image.  The 16-byte chunk c of row r is fetched into
// chunk position c ^ ((r>>1)&7) (swizzle applied on the SOURCE address, the LDS destination of an LDS-DMA cannot be
// scattered) so that the ds_read_b128 fragment reads (16 different rows per lane group) hit distinct banks.
// Every wave owns a private ring of NS tiles and runs ahead of its own MFMAs by NS-1 tiles with counted
// s_waitcnt vmcnt(N): no workgroup barrier anywhere.  (A variant that shared the x tile between the four waves of a
// workgroup to afford a 6-8 deep ring - 2.5x the weight bytes in flight - was not faster, 2450 vs 2200 cycles per tile at
// 64 rows with the per-tile barrier: the prefetch depth is not what limits this kernel.)
// What limited it at 32/64 rows was VALU work between the MFMAs: the f32 MFMA runs at the vector rate and does not overlap
// VALU instructions, so the 64-bit address arithmetic of every LDS-DMA instruction stretched a 2048-cycle tile to 2560
// cycles.  The loop below is VALU-free (scalar base + 32-bit lane offset addressing); in-kernel stamps
// (tools/probes/l1_clock_probe.hip) read 2200 cycles per tile at 64 rows = 93 % of the matrix pipe, at an in-kernel clock
// of 1.98 GHz (32 rows: 1140 of 1024 cycles at 1.72 GHz - the chip lowers its clock under the combined HBM + MFMA load).  Split-K partials are reduced afterwards in a fixed order.
#include "common.hpp"

// the LDS-DMA asm below names m0 in its clobber list on purpose (it writes it)
#pragma clang diagnostic ignored "-Winline-asm"

#include <type_traits>

namespace shasta {

struct AnchorMfmaArgs {
    const float* W[4];
    const float* x[2];
    float* part;
    int H, K, B, KS, Kc, x_batch_stride, groups_per_mlp;
};

// weights are read exactly once per launch: the weight stream is issued non-temporal (nt) so that the activation vectors
// and the small weights of the following kernels stay resident in L2 / Infinity Cache (MI355X_MICROARCH.md, row nt-weights)
#ifdef SHASTA_L1_STAMP
__device__ unsigned long long g_l1_stamp[4096][3];
#endif

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
    constexpr int XSLOT = XR * 32;
    constexpr int SLOT = 1024 + XSLOT;          // floats per private ring slot: W tile (32 x 32) + x tile (XR x 32)
    constexpr int PER_TILE = 4 + XI;            // vmcnt units per tile
    static_assert(PER_TILE * (NS - 1) <= 63, "vmcnt is 6 bits");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // provably wave-uniform -> scalar control flow
    float* const wring = lds + (size_t)wid * NS * SLOT;  // slot = [W tile | x tile]
    float* const xring = wring + 1024;
    constexpr int WSTRIDE = SLOT, XSTRIDE = SLOT;  // floats between consecutive slots

    const int item = blockIdx.x * 4 + wid;
    const int G = 4 * a.groups_per_mlp;  // 32-row blocks over the four MLPs
    if (item >= G * a.KS) return;
    const int ks = item / G, g = item % G;
    const int mlp = g / a.groups_per_mlp, r0 = (g % a.groups_per_mlp) * 32;
    const int b0 = blockIdx.y * XR;
    const int kbeg = ks * a.Kc, kend = min(a.K, kbeg + a.Kc);
    const int NT = (kend - kbeg) >> 5;  // 32-float tiles in this chunk

    // staging roles: instruction j covers rows 8j..8j+7, lane -> (row 8j + lane/8, chunk position lane%8).
    // Source address = wave-uniform base (advanced per tile on the scalar unit) + a 32-bit per-lane byte offset.
    const float* wbase = mlp == 0 ? a.W[0] : mlp == 1 ? a.W[1] : mlp == 2 ? a.W[2] : a.W[3];
    const float* xbase = mlp < 2 ? a.x[0] : a.x[1];
    const char* wub = reinterpret_cast<const char*>(wbase + (size_t)r0 * a.K + kbeg);
    const char* xub = reinterpret_cast<const char*>(xbase + (size_t)b0 * a.x_batch_stride + kbeg);
    uint32_t woff[4], xoff[XI];
    {
        const int cpos = lane & 7, rl = lane >> 3;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int r = 8 * j + rl;
            const int c = cpos ^ ((r >> 1) & 7);
            woff[j] = (uint32_t)(((min(r0 + r, a.H - 1) - r0) * a.K + 4 * c) * 4);
        }
#pragma unroll
        for (int j = 0; j < XI; ++j) {
            const int r = 8 * j + rl;  // row of the staged x tile (0..XR-1); sub-tile u = r / 32 keeps the same swizzle
            const int c = cpos ^ (((r & 31) >> 1) & 7);
            xoff[j] = (uint32_t)(((min(b0 + r, a.B - 1) - b0) * a.x_batch_stride + 4 * c) * 4);
        }
    }
    constexpr int ND = PER_TILE;                       // LDS-DMA instructions per tile
    constexpr int NR = WIDE ? 4 + 4 * XT : 6;          // ds_read_b128 per tile
    constexpr int NM = WIDE ? 16 * XT : 16;            // MFMAs per tile
    constexpr int DPS = XT == 2 ? 1 : 2;               // DMA instructions issued behind one MFMA
    constexpr int DSLOTS = (ND + DPS - 1) / DPS;
    static_assert(DSLOTS + NR < NM, "the fragment reads must end a few MFMAs before the tile does");
    // The f32 MFMA does not overlap VALU work, so the loop must not contain any: the tile advance is added to the uniform
    // base on the scalar unit and the per-lane 32-bit offset goes into the instruction's VGPR-offset field
    // (global_load_lds_dwordx4 voff, s[base:base+1]).  hipcc only emits the 64-bit-VGPR-address form for the LDS-DMA
    // builtin (two v_lshl_add_u64 per instruction: measured 2560 instead of 2048 cycles per tile at 64 rows), hence the asm.
    const uint32_t wlds = (uint32_t)(size_t)((__attribute__((address_space(3))) float*)wring);
    const uint32_t xlds = (uint32_t)(size_t)((__attribute__((address_space(3))) float*)xring);
    auto dma = [&](int t, int idx) {  // instruction idx of tile t
        const size_t ko = (size_t)t * 128;
        if (idx < 4) {
            const char* base = wub + ko;
            const uint32_t dst = wlds + (uint32_t)(((t % NS) * WSTRIDE + idx * 256) * 4);
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 nt" ::"v"(woff[idx]), "s"(base), "s"(dst)
                         : "memory", "m0");
        } else {
            const char* base = xub + ko;
            const uint32_t dst = xlds + (uint32_t)(((t % NS) * XSTRIDE + (idx - 4) * 256) * 4);
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(xoff[idx - 4]), "s"(base), "s"(dst)
                         : "memory", "m0");
        }
    };
    auto issue = [&](int t) {
#pragma unroll
        for (int j = 0; j < ND; ++j) dma(t, j);
    };

    // Fragment registers of one tile.  Two sets ping-pong: the MFMAs of tile t run from one set while the other receives
    // tile t+1.  One wave per SIMD issues in order, so anything that is not an MFMA only hides if it sits BETWEEN two MFMAs
    // (it issues while the matrix pipe is busy with the previous one): the LDS-DMA instructions of tile t+NS follow the
    // first MFMAs one by one, then the counted wait, then one ds_read_b128 of tile t+1 behind each following MFMA, and the
    // last MFMAs cover the latency of the last read.  Before this interleave ~600 of every 2650 cycles per tile (64 batch
    // rows) had the matrix pipe idle behind DMA issue, the wait and the fragment reads.
    struct Frag {
        f32x4 w[4];                     // WIDE: w[m]; else w[2*rb + u]
        f32x4 x[WIDE ? XT : 1][4];      // WIDE: x[u][m]; else x[0][u] (u < 2)
    };
    const int frag_row = WIDE ? (lane & 31) : (lane & 15);
    auto read_one = [&](int slot, Frag& f, int idx) {
        const float* sw_ = wring + slot * WSTRIDE;
        const float* sx_ = xring + slot * XSTRIDE;
        if constexpr (WIDE) {
            const int h = lane >> 5, sw = (frag_row >> 1) & 7;
            const int m = idx / (1 + XT), k = idx % (1 + XT);
            const int pos = (2 * m + h) ^ sw;
            if (k == 0) f.w[m] = *reinterpret_cast<const f32x4*>(sw_ + frag_row * 32 + pos * 4);
            else f.x[k - 1][m] = *reinterpret_cast<const f32x4*>(sx_ + (k - 1) * 1024 + frag_row * 32 + pos * 4);
        } else {
            const int kq = lane >> 4;
            const int u = idx / 3, k = idx % 3;
            const int c = 4 * u + kq;
            if (k == 0) f.x[0][u] = *reinterpret_cast<const f32x4*>(sx_ + frag_row * 32 + (c ^ ((frag_row >> 1) & 7)) * 4);
            else {
                const int row = 16 * (k - 1) + frag_row;
                f.w[2 * (k - 1) + u] = *reinterpret_cast<const f32x4*>(sw_ + row * 32 + (c ^ ((row >> 1) & 7)) * 4);
            }
        }
    };
    auto read_frags = [&](int slot, Frag& f) {
#pragma unroll
        for (int i = 0; i < NR; ++i) read_one(slot, f, i);
    };

    f32x16 acc32[WIDE ? XT : 1];
#pragma unroll
    for (int u = 0; u < (WIDE ? XT : 1); ++u) acc32[u] = f32x16{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    f32x4 acc16[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};

    auto mma_one = [&](const Frag& f, int i) {
        if constexpr (WIDE) {
            const int m = i / (4 * XT), q = (i / XT) % 4, u = i % XT;
            acc32[u] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.w[m][q], f.x[u][m][q], acc32[u], 0, 0, 0);
        } else {
            const int u = i / 8, q = (i / 2) % 4, rb = i % 2;
            acc16[rb] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.w[2 * rb + u][q], f.x[0][u][q], acc16[rb], 0, 0, 0);
        }
    };

#ifdef SHASTA_L1_STAMP  // diagnostic build only (tools/probes/l1_clock_probe.hip): in-kernel clock and cycles per tile
    const unsigned long long st0 = __builtin_amdgcn_s_memtime(), sr0 = __builtin_amdgcn_s_memrealtime();
#endif
    // Prologue: fill the ring, fetch the fragments of tile 0.
    int issued = 0;
#pragma unroll 1
    for (; issued < NS && issued < NT; ++issued) issue(issued);
    Frag fa, fb;
    if (NT >= NS) wait_vm<PER_TILE*(NS - 1)>();
    else wait_vm<0>();
    if (NT > 0) read_frags(0, fa);
    // One tile: `cur` holds tile t (so its ring slot is free again), `nxt` receives tile t+1.
    // STEADY: tile t+NS exists, so exactly NS tiles are in flight at the wait and the vmcnt immediate is a compile-time
    // constant; the last NS+1 tiles (STEADY == false) issue what is left up front and drain the queue.  Straight-line
    // bodies on purpose: a data-dependent choice of the immediate inside the loop made the register allocator shuttle all
    // 32 accumulators AGPR -> VGPR -> AGPR every iteration.
    auto step = [&](const Frag& cur, Frag& nxt, int t, auto steady) {
        constexpr bool STEADY = decltype(steady)::value;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the fragment reads of tile t have retired
        if constexpr (!STEADY) {
            if (t + NS < NT) issue(t + NS);  // at most one tile is still unissued when the steady loop ends
        }
        const int sn = (t + 1) % NS;  // past the last tile: a harmless read of a stale slot
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < NM; ++i) {
            mma_one(cur, i);
            __builtin_amdgcn_sched_barrier(0);
            if (i < DSLOTS) {
                if constexpr (STEADY) {
#pragma unroll
                    for (int d = 0; d < DPS; ++d)
                        if (i * DPS + d < ND) dma(t + NS, i * DPS + d);
                }
                if (i == DSLOTS - 1) {
                    if constexpr (STEADY) wait_vm<PER_TILE*(NS - 1)>();
                    else wait_vm<0>();
                }
                __builtin_amdgcn_sched_barrier(0);
            } else if (i - DSLOTS < NR) {
                read_one(sn, nxt, i - DSLOTS);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
    int t = 0;
#pragma unroll 1
    for (; t + NS + 1 < NT; t += 2) {
        step(fa, fb, t, std::true_type{});
        step(fb, fa, t + 1, std::true_type{});
    }
#pragma unroll 1
    for (; t + 1 < NT; t += 2) {
        step(fa, fb, t, std::false_type{});
        step(fb, fa, t + 1, std::false_type{});
    }
    if (t < NT) step(fa, fb, t, std::false_type{});
#ifdef SHASTA_L1_STAMP
    if (lane == 0 && wid == 0 && blockIdx.y == 0 && blockIdx.x < 4096) {
        g_l1_stamp[blockIdx.x][0] = __builtin_amdgcn_s_memtime() - st0;
        g_l1_stamp[blockIdx.x][1] = __builtin_amdgcn_s_memrealtime() - sr0;
        g_l1_stamp[blockIdx.x][2] = (unsigned long long)NT;
    }
#endif
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
    if (B <= 16) launch(anchor_l1_mfma_kernel<0, 6>, 6, 16);
    else if (B <= 32) launch(anchor_l1_mfma_kernel<1, 4>, 4, 32);
    else launch(anchor_l1_mfma_kernel<2, 3>, 3, 64);
    return 0;
}

}  // namespace shasta
