// Shared by the two one-pass aff kernels (aff_pieces.hip: bf16 pieces, aff_f16.hip: fp16 pieces): workgroup shapes, kernel arguments and
// the tail that turns the logits held in the accumulator registers of layer 6 into matched1 and matched2 (det3d/models/tracker/
// shasta.py:323-325) - the row softmax inside the workgroup, the column softmax through per-column partials exchanged between the
// workgroups of a frame.
#pragma once
#include "common.hpp"
#include "pair_layout.hpp"

namespace shasta {

typedef uint32_t qu32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t qu32x2 __attribute__((ext_vector_type(2)));

constexpr int AP_AROW = 272;  // bytes per row of image A (128 bf16 + 16 B pad: 68 dwords, conflict-free b128 reads)
constexpr int AP_BROW = 144;  // bytes per row of image B (64 bf16 + 16 B pad: 36 dwords, conflict-free)
constexpr int AP_SCOLS = 256, AP_SROW = AP_SCOLS + 4;  // output staging: ROWS x 256 features per pass
// Shape of a workgroup: ROWS residual rows (RB = ROWS / 32 row blocks) on WAVES = 2 RB wavefronts.
//  <128, 8>: one workgroup per CU (160 KB of LDS); every weight fragment feeds 4 x 6 MFMAs.
//  < 64, 4>: two workgroups per CU (80 KB each); fragments feed 2 x 6 MFMAs (twice the L2 -> CU weight traffic per row), but the
//            phases of the two co-resident workgroups overlap: one streams its output rows while the other is in its MFMAs.
template <int ROWS, int WAVES>
struct ApShape {
    static_assert(WAVES * 16 == ROWS, "WAVES = 2 * row blocks");
    static constexpr int RB = ROWS / 32;
    static constexpr int NFW = 16 / WAVES;                              // feature blocks of layer 6 per wave (tables up to 512 columns)
    static constexpr int AIMG = ROWS * AP_AROW, BIMG = ROWS * AP_BROW;  // one piece of image A / B
    static constexpr int ABYTES = 3 * AIMG, BBYTES = 3 * BIMG;          // 128 rows: 104448 + 55296 = 159744 B; 64 rows: 79872 B
    static constexpr int STAT = ROWS * AP_SROW * 4;                     // byte offset of the softmax scratch behind the staging
    static constexpr int L1X = ROWS * 128, L1SLOT = L1X + 24 * 1024;    // layer-1 ring slot: x chunk + 24 weight fragments
    static constexpr int NS = (ABYTES + BBYTES) / L1SLOT >= 3 ? 3 : 2;  // ring slots
    static constexpr int WPW = 24 / WAVES, PER = 2 + WPW;               // LDS-DMA instructions per chunk and wave: weights, total
    static_assert(NS * L1SLOT <= ABYTES + BBYTES, "the layer-1 ring lies over the two images");
    static_assert(STAT + (2 * WAVES + 2) * ROWS * 4 <= ABYTES + BBYTES, "staging + softmax scratch must fit the two images");
    static_assert(PER * (NS - 1) <= 63, "vmcnt is 6 bits");
};

#ifdef SHASTA_AFF_STAMP  // diagnostic build only (tools/probes/aff_probe.hip): s_memtime at the phase boundaries of a workgroup
__device__ unsigned long long g_aff_stamp[4096][8];
#define AP_STAMP(i) \
    if (threadIdx.x == 0 && blockIdx.x < 4096) g_aff_stamp[blockIdx.x][i] = __builtin_amdgcn_s_memtime()
#else
#define AP_STAMP(i)
#endif

struct AffPiecesArgs {
    const uint32_t* wp;  // piece fragments of the six layers
    const float* bias[6];
    const float* residual;
    float* matched;  // (M, ldm) pre-softmax, for the column softmax
    float* m1;       // (B, N, D)
    int M, T, N, D, Dp, ld, ldm;
};

// ---------------------------------------------------------------------------------------------------------------------------------
// aff_frame_kernel: the six layers AND BOTH softmaxes (shasta.py:323-325) in one pass over the residual - `matched` is never written
// (unless the caller asks for it).  The rows of a frame are dealt to G = ceil(T / ROWS) workgroups with consecutive block ids, each
// keeps its ROWS x D logits in the accumulator registers of layer 6, and
//   * matched1 = softmax over the columns of a row: complete inside the workgroup, as in aff_pieces_kernel;
//   * matched2 = softmax over the T rows of a column: every workgroup reduces its rows to (max, sum of exp(x - max)) per column,
//     publishes the 2 x D floats, counts itself on the frame's arrival counter and waits for its G - 1 siblings; all of them then
//     combine the G partials in the same fixed order and write their rows of matched2 straight from the registers.
// Waiting is safe by construction: a workgroup does not take its row group from blockIdx but from a TICKET (one atomic counter per
// launch, ap_take_ticket): tickets are handed out in the order the workgroups actually start, whatever order the hardware dispatches
// block ids in, so the siblings of a running workgroup (the tickets of the same frame) are running, finished, or the very next ones
// to start - never queued behind a workgroup that waits (the look-back scans of rocPRIM / CUB number their tiles the same way).
// A wait that nevertheless outlasts ~2 s poisons this workgroup's rows of matched2 with NaN instead of hanging the device AND sets
// bit 0 of the launch's status word (shasta_aff_status / shasta_forward_status read it): the C call has long returned SHASTA_OK.
// Ordering of the exchange: every thread's partial stores are agent-scope atomics and complete (s_waitcnt vmcnt(0)) before the
// workgroup barrier; thread 0 then counts the workgroup in with a RELEASE fence at agent scope in front of the read-modify-write
// and puts an ACQUIRE fence behind the poll that saw all G arrivals; the barrier that follows hands that edge to the other threads.
// Partials cross the XCDs' L2s as agent-scope (sc1) stores / loads - no cache-wide write-back or invalidate.
// exp(x - m) = v_exp_f32(fma(x, log2 e, c)), c = fl(-m log2 e): the rounding of c is common to a whole row (column), i.e. it cancels
// between the sum and the terms; the G column partials are rescaled by exp2(c_frame - c_group) with the very same constants.
struct AffFrameArgs {
    AffPiecesArgs p;   // p.matched == nullptr: do not write the logits
    float* m2;         // (B, T, N)
    float* part;       // [B][G][2][512]: per row group the column maxima, then the column sums
    unsigned* arrive;  // [B], zero on entry
    unsigned* ticket;  // [1], zero on entry: the next logical tile (b, q) = (ticket / G, ticket % G)
    unsigned* status;  // [1], zero on entry: bit 0 = a sibling wait timed out (rows of matched2 are NaN)
    int G;
};

// control words of a one-pass launch at the head of its workspace: [status, ticket, arrive[B]] (one memset per launch)
__host__ __device__ inline size_t aff_frame_ctrl_bytes(int B) { return ((size_t)(B + 2) * sizeof(unsigned) + 255) / 256 * 256; }

// the logical tile of this workgroup: tickets in start order when row groups wait for each other (G > 1), else the block id.
// `slot`: four bytes of LDS nobody else touches until the second barrier.
__device__ __forceinline__ unsigned ap_take_ticket(const AffFrameArgs& fa, unsigned* slot) {
    if (fa.G <= 1) return blockIdx.x;
#ifdef SHASTA_AFF_NO_TICKET  // A/B diagnostic only (tools/gpu_aff_ab.sh): block ids as tiles, i.e. trust in-order dispatch
    return blockIdx.x;
#endif
    if (threadIdx.x == 0) *slot = __hip_atomic_fetch_add(fa.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const unsigned t = *slot;
    __syncthreads();
    return __builtin_amdgcn_readfirstlane(t);
}

constexpr float AP_LOG2E = 1.44269504088896340736f;

// Maximum / sum over the 32 lanes of a half wave for four values at once, valid in lanes 16-31 (48-63): quad swaps, half-row mirror,
// row mirror, then lane 15 of rows 0 and 2 into rows 1 and 3 (row_bcast:15; rows 0 and 2 are not written).  Hand-written: through
// fmaxf the compiler puts a canonicalising v_max between every DPP move and its use and chains the steps with s_nops (813 v_max +
// 320 v_mov_dpp + 279 s_nop per wave for the 64 reductions of the tail); here the four chains interleave, which covers the two wait
// states a DPP read of a freshly written register needs (the leading s_nop covers the compiler's instructions before the block).
#define AP_DPP4(op, ctrl)                              \
    op " %0, %0, %0 " ctrl " bank_mask:0xf\n\t"       \
    op " %1, %1, %1 " ctrl " bank_mask:0xf\n\t"       \
    op " %2, %2, %2 " ctrl " bank_mask:0xf\n\t"       \
    op " %3, %3, %3 " ctrl " bank_mask:0xf\n\t"
#define AP_HALF4(op)                                                                                                          \
    asm("s_nop 1\n\t" AP_DPP4(op, "quad_perm:[1,0,3,2] row_mask:0xf") AP_DPP4(op, "quad_perm:[2,3,0,1] row_mask:0xf")          \
            AP_DPP4(op, "row_half_mirror row_mask:0xf") AP_DPP4(op, "row_mirror row_mask:0xf") AP_DPP4(op, "row_bcast:15 row_mask:0xa") \
        : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]))
__device__ __forceinline__ void ap_half_max4(float (&v)[4]) { AP_HALF4("v_max_f32_dpp"); }
__device__ __forceinline__ void ap_half_sum4(float (&v)[4]) { AP_HALF4("v_add_f32_dpp"); }
// the additive constant of exp2 for a maximum m: -m log2(e); an empty (-inf) maximum gives 0 so that its terms are exp2(-inf) = 0
__device__ __forceinline__ float ap_expc(float m) { return m == -INFINITY ? 0.0f : -m * AP_LOG2E; }
__device__ __forceinline__ float ap_exp(float x, float c) { return __builtin_amdgcn_exp2f(__builtin_fmaf(x, AP_LOG2E, c)); }

// four consecutive columns of an output row: streaming (nt) stores; `o` has the same alignment in every lane of the wave (rows are
// written by whole waves).  A 16-byte store needs no more than the 4-byte alignment of its dwords in hardware (unaligned access mode of
// the HSA runtime), which the odd rows of matched1 (D = 2 mod 4: 8 bytes off) rely on: two 8-byte stores per lane instead cost a third
// of the write-out of a workgroup (33.7 k -> 23 k cycles on an otherwise idle chip).
__device__ __forceinline__ void ap_store4(float* o, const f32x4& e, int nvalid) {
    if (nvalid >= 4) {
        asm volatile("global_store_dwordx4 %0, %1, off nt" ::"v"(o), "v"(e) : "memory");
    } else {
#pragma unroll
        for (int q = 0; q < 3; ++q)
            if (q < nvalid) __builtin_nontemporal_store(e[q], o + q);
    }
}

// The tail of a one-pass aff kernel.  On entry acc[i][rb] holds the layer-6 sums of features (wid + WAVES i) 32 + 8 (r >> 2) +
// 4 (lane >> 5) + (r & 3) for row 32 rb + (lane & 31) of the row group (b, q) - plain (SCALED = false: logit = acc + bias) or in the
// block-scaled form of the fp16 kernels (SCALED: logit = acc dsc[feature] rs[rb] + bias, dsc and rs exact powers of two).  The tail
// lays its staging and scratch over the first ApShape::ABYTES + BBYTES bytes of the workgroup's LDS: nothing the caller still needs may
// live there (the other waves may still READ the layer-5 image behind `smem` - the tail's first barrier waits for them).
template <int ROWS, int WAVES, bool SCALED>
__device__ __forceinline__ void ap_frame_tail(const AffFrameArgs& fa, char* smem, f32x16 (&acc)[ApShape<ROWS, WAVES>::NFW][ApShape<ROWS, WAVES>::RB],
                                              int b, int q, int nrows, int g0, int tid, int lane, int wid, const float* __restrict__ dsc,
                                              const float (&rs)[ApShape<ROWS, WAVES>::RB]) {
    using S = ApShape<ROWS, WAVES>;
    constexpr int RB = S::RB, NFW = S::NFW, NT = 64 * WAVES;
    static_assert(S::STAT + (2 * WAVES + 3) * ROWS * 4 + 4 * 512 * 4 <= S::ABYTES + S::BBYTES, "staging + softmax scratch must fit the two images");
    const AffPiecesArgs& a = fa.p;
    const int G = fa.G;
    const int D = a.D, N = a.N, nfb = ap_fblocks(5, D);

    float* xs = reinterpret_cast<float*>(smem);              // [ROWS][AP_SROW] staging
    float* pmax = reinterpret_cast<float*>(smem + S::STAT);  // [WAVES][ROWS]
    float* psum = pmax + WAVES * ROWS;                       // [WAVES][ROWS]
    float* rnm = psum + WAVES * ROWS;                        // [ROWS] -max log2(e) of the row
    float* rinv = rnm + ROWS;                                // [ROWS] 1 / sum
    float* cmaxl = rinv + 2 * ROWS;                          // [512] column maxima of this row group (16-byte aligned)
    float* csuml = cmaxl + 512;                              // [512] column sums
    float* ccst = csuml + 512;                               // [512] frame-wide: -max log2(e)
    float* cinv = ccst + 512;                                // [512] frame-wide: 1 / sum
    const int n = lane & 31, hh = lane >> 5;
    // Rows beyond the frame's last (last row group) and features beyond D (last feature block) take no part in anything: their
    // logits become -inf (exp2(-inf) = 0, a maximum ignores it), so that the loops below need no per-element masks.
    float mx[RB];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) mx[rb] = -INFINITY;
    __syncthreads();  // every wave is done reading image A: the scratch may overwrite it
    const bool ragged_rows = nrows < ROWS;  // workgroup-uniform
#pragma unroll
    for (int i = 0; i < NFW; ++i) {
        const int fb = wid + WAVES * i;
        if (fb >= nfb) continue;  // wave-uniform
        const bool ragged_f = (fb + 1) * 32 > D;  // wave-uniform
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float cm[4];
            const int f0 = fb * 32 + 8 * g + 4 * hh;
            f32x4 dv4 = {0, 0, 0, 0};
            if (SCALED) dv4 = *reinterpret_cast<const f32x4*>(dsc + f0);  // padded to whole feature blocks
            float bv4[4];
            if (!ragged_f) {
#pragma unroll
                for (int j = 0; j < 4; ++j) bv4[j] = a.bias[5][f0 + j];
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) bv4[j] = f0 + j < D ? a.bias[5][f0 + j] : 0.0f;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int f = f0 + j;
                const float bv = bv4[j], dv = dv4[j];
#pragma unroll
                for (int rb = 0; rb < RB; ++rb)
                    acc[i][rb][4 * g + j] = SCALED ? __builtin_fmaf(acc[i][rb][4 * g + j], dv * rs[rb], bv) : acc[i][rb][4 * g + j] + bv;
                if (ragged_f) {
#pragma unroll
                    for (int rb = 0; rb < RB; ++rb) acc[i][rb][4 * g + j] = f < D ? acc[i][rb][4 * g + j] : -INFINITY;
                }
                if (ragged_rows) {
#pragma unroll
                    for (int rb = 0; rb < RB; ++rb) acc[i][rb][4 * g + j] = rb * 32 + n < nrows ? acc[i][rb][4 * g + j] : -INFINITY;
                }
                float c = -INFINITY;
#pragma unroll
                for (int rb = 0; rb < RB; ++rb) {
                    mx[rb] = fmaxf(mx[rb], acc[i][rb][4 * g + j]);
                    c = fmaxf(c, acc[i][rb][4 * g + j]);
                }
                cm[j] = c;
            }
            ap_half_max4(cm);
            if (n == 16) *reinterpret_cast<f32x4*>(cmaxl + fb * 32 + 8 * g + 4 * hh) = f32x4{cm[0], cm[1], cm[2], cm[3]};
        }
    }
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
        const float o = fmaxf(mx[rb], __shfl_xor(mx[rb], 32, 64));
        if (hh == 0) pmax[wid * ROWS + rb * 32 + n] = o;
    }
    __syncthreads();
    float nm[RB], se[RB];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
        float m = pmax[rb * 32 + n];
#pragma unroll
        for (int w = 1; w < WAVES; ++w) m = fmaxf(m, pmax[w * ROWS + rb * 32 + n]);
        nm[rb] = ap_expc(m);
        se[rb] = 0.0f;
    }
    // sums: of a row over the valid columns, of a column over the valid rows of this group (relative to the group's own maximum)
#pragma unroll
    for (int i = 0; i < NFW; ++i) {
        const int fb = wid + WAVES * i;
        if (fb >= nfb) continue;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 cmx = *reinterpret_cast<const f32x4*>(cmaxl + fb * 32 + 8 * g + 4 * hh);  // written by this wave
            float cs[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float cc = ap_expc(cmx[j]);
                float s = 0.0f;
#pragma unroll
                for (int rb = 0; rb < RB; ++rb) {
                    const float v = acc[i][rb][4 * g + j];
                    se[rb] += ap_exp(v, nm[rb]);
                    s += ap_exp(v, cc);
                }
                cs[j] = s;
            }
            ap_half_sum4(cs);
            if (n == 16) *reinterpret_cast<f32x4*>(csuml + fb * 32 + 8 * g + 4 * hh) = f32x4{cs[0], cs[1], cs[2], cs[3]};
        }
    }
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
        const float o = se[rb] + __shfl_xor(se[rb], 32, 64);
        if (hh == 0) psum[wid * ROWS + rb * 32 + n] = o;
    }
    __syncthreads();
    if (tid < ROWS) {
        float s = psum[tid];
#pragma unroll
        for (int w = 1; w < WAVES; ++w) s += psum[w * ROWS + tid];  // fixed order
        float m = pmax[tid];
#pragma unroll
        for (int w = 1; w < WAVES; ++w) m = fmaxf(m, pmax[w * ROWS + tid]);
        rnm[tid] = ap_expc(m);
        rinv[tid] = 1.0f / s;
    }
    // publish this group's column partials (agent scope: the siblings sit behind other L2s)
    float* mypart = fa.part + ((size_t)b * G + q) * 1024;
    if (G > 1) {
        for (int c = tid; c < N; c += NT) {
            __hip_atomic_store(mypart + c, cmaxl[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(mypart + 512 + c, csuml[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    AP_STAMP(4);
    bool poisoned = false;
    // Write-out in two passes of HR = ROWS / 2 rows through the staging [HR][512 + 4]: a wave then owns WHOLE rows - its stores of one
    // row are 2 KB contiguous and the eight waves write eight consecutive rows, i.e. the workgroup walks linearly through its (dense)
    // blocks of matched1 and matched2 (staging by column halves - 1 KB pieces 2 KB apart - left the DRAM side of the stores at
    // ~1.2 TB/s).  matched1 = exp(x - row max) / row sum for the rows t < N, matched2 = exp(x - column max) / column sum for the
    // columns d < N, the logits only on request.
    constexpr int HR = ROWS / 2, XROW = 2 * AP_SCOLS + 4;
    static_assert(HR * XROW * 4 <= S::STAT, "row-half staging must fit below the softmax scratch");
    f32x4 kc[2], ki[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        if (i == 1) __syncthreads();  // the read-out of pass 0 is finished
#pragma unroll
        for (int k = 0; k < NFW; ++k) {
            const int fb = wid + WAVES * k;
            if (fb < nfb) {
#pragma unroll
                for (int rb = i * RB / 2; rb < (i + 1) * RB / 2; ++rb)
#pragma unroll
                    for (int g = 0; g < 4; ++g)
                        *reinterpret_cast<f32x4*>(xs + ((rb - i * RB / 2) * 32 + n) * XROW + fb * 32 + 8 * g + 4 * hh) =
                            f32x4{acc[k][rb][4 * g], acc[k][rb][4 * g + 1], acc[k][rb][4 * g + 2], acc[k][rb][4 * g + 3]};
            }
        }
        if (i == 0) {
            // arrival: all partial stores of this workgroup have completed (vmcnt) before thread 0 counts it in; then the frame-wide
            // column statistics, identically in every sibling
            if (G > 1) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                if (tid == 0) {
#ifdef SHASTA_AFF_FORCE_TIMEOUT  // test build (tests/test_aff_stage.py): row group 0 never counts itself in, everybody's wait runs out quickly
                    constexpr unsigned SPIN_LIMIT = 1u << 8;
                    const unsigned mine = q == 0 ? 0u : 1u;
#else
                    constexpr unsigned SPIN_LIMIT = 1u << 21;
                    const unsigned mine = 1u;
#endif
                    // release: the partials of this workgroup (complete, see above) are ordered before the count the siblings poll
#ifndef SHASTA_AFF_NO_FENCE  // (A/B diagnostic only, tools/gpu_aff_ab.sh)
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
#endif
                    // polled with the same read-modify-write path that counts the arrivals
                    unsigned seen = __hip_atomic_fetch_add(fa.arrive + b, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + mine, spins = 0;
                    while (seen < (unsigned)G) {
                        __builtin_amdgcn_s_sleep(16);
                        seen = __hip_atomic_fetch_add(fa.arrive + b, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (++spins > SPIN_LIMIT) {
                            poisoned = true;
                            break;
                        }
                    }
                    // acquire: the siblings' partials are read after the count that announced them
#ifndef SHASTA_AFF_NO_FENCE
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#endif
                    if (poisoned) __hip_atomic_fetch_or(fa.status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    cinv[511] = poisoned ? 1.0f : 0.0f;  // column 511 is never a column of matched2
                }
                __syncthreads();
                poisoned = cinv[511] != 0.0f;
                for (int c = tid; c < N; c += NT) {
                    float mg[512 / ROWS], sg[512 / ROWS];
                    const float* fp = fa.part + (size_t)b * G * 1024 + c;
#pragma unroll
                    for (int g = 0; g < 512 / ROWS; ++g)
                        if (g < G) {
                            mg[g] = __hip_atomic_load(fp + g * 1024, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            sg[g] = __hip_atomic_load(fp + g * 1024 + 512, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        }
                    float m = mg[0];
#pragma unroll
                    for (int g = 1; g < 512 / ROWS; ++g)
                        if (g < G) m = fmaxf(m, mg[g]);
                    const float cf = ap_expc(m);
                    float s = 0.0f;
#pragma unroll
                    for (int g = 0; g < 512 / ROWS; ++g)
                        if (g < G) s += mg[g] == -INFINITY ? 0.0f : sg[g] * __builtin_amdgcn_exp2f(cf - ap_expc(mg[g]));
                    ccst[c] = cf;
                    cinv[c] = poisoned ? __builtin_nanf("") : 1.0f / s;
                }
            } else {
                __syncthreads();
                for (int c = tid; c < N; c += NT) {
                    ccst[c] = ap_expc(cmaxl[c]);
                    cinv[c] = 1.0f / csuml[c];
                }
            }
        }
        __syncthreads();
        if (i == 0) {
            AP_STAMP(5);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int col = h * AP_SCOLS + 4 * lane;
                kc[h] = f32x4{0, 0, 0, 0}, ki[h] = f32x4{0, 0, 0, 0};
                if (col < N) {  // N <= 510: the float4s of ccst / cinv reach at most column 511
                    kc[h] = *reinterpret_cast<const f32x4*>(ccst + col);
                    ki[h] = *reinterpret_cast<const f32x4*>(cinv + col);
                }
            }
        }
#pragma unroll 2
        for (int rr = 0; rr < HR / WAVES; ++rr) {
            const int rl = wid + rr * WAVES, r = i * HR + rl;
            if (r >= nrows) break;  // wave-uniform
            const float cr = rnm[r], inv = rinv[r];
            const int t = q * ROWS + r;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int col = h * AP_SCOLS + 4 * lane;
                if (col >= a.ldm) continue;
                const f32x4 v = *reinterpret_cast<const f32x4*>(xs + rl * XROW + col);
                if (a.matched) {
                    f32x4 z = v;
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (col + e >= D) z[e] = 0.0f;  // the padding columns of `matched`
                    *reinterpret_cast<f32x4*>(a.matched + (size_t)(g0 + r) * a.ldm + col) = z;
                }
#ifndef AP_ABL_NO_M1  // ablation builds of tools/probes/aff_frame_probe.hip (results are wrong)
                if (t < N && col < D) {
                    f32x4 e;
#pragma unroll
                    for (int k = 0; k < 4; ++k) e[k] = ap_exp(v[k], cr) * inv;
                    ap_store4(a.m1 + ((size_t)b * N + t) * D + col, e, D - col);
                }
#endif
#ifndef AP_ABL_NO_M2
                if (col < N) {
                    f32x4 e;
#pragma unroll
                    for (int k = 0; k < 4; ++k) e[k] = ap_exp(v[k], kc[h][k]) * ki[h][k];
                    ap_store4(fa.m2 + ((size_t)b * a.T + t) * N + col, e, N - col);
                }
#endif
            }
        }
    }
    AP_STAMP(6);
}

}  // namespace shasta
