// f-4 (second half): the N x M centre-distance matrix and greedy assignment of the public tracker
// (tools/nusc_shasta/pub_tracker.py:94-108, track_utils.py:3-14), batched over independent scenes.
//   dist[i][j] = sqrt((tx_j - dx_i)^2 + (ty_j - dy_i)^2)                 float32, numpy's operation order
//   invalid    = dist > max_diff[i]  or  det_cat[i] != trk_cat[j]
//   dist64     = double(dist) + (invalid ? 1e18 : 0)                     (numpy promotes bool * 1e18 to float64)
//   greedy     : for i = 0..N-1: j = argmin_j dist64[i][j] (first minimum; columns already taken count as 1e18);
//                if that minimum < 1e16: match (i, j), take column j.
// The greedy loop is sequential in i by definition; one wavefront owns one scene (no barrier: the taken-column flags live
// in LDS words private to the wave, the argmin is a butterfly over (value, index) pairs), scenes run side by side.
#include "common.hpp"

namespace shasta {

struct GreedyArgs {
    const float* det_xy;   // (S, Nmax, 2)
    const float* trk_xy;   // (S, Mmax, 2)
    const int* det_cat;    // (S, Nmax)
    const int* trk_cat;    // (S, Mmax)
    const float* max_diff; // (S, Nmax)
    const int* n;          // (S,)
    const int* m;          // (S,)
    double* dist;          // (S, Nmax, Mmax) or nullptr
    int* match;            // (S, Nmax): matched track index or -1
    int* row_any;          // (S, Nmax) or nullptr: 1 when some track is inside the detection's gate (class + distance)
    int* col_any;          // (S, Mmax) or nullptr: 1 when some detection is inside the gate of the track
    int Nmax, Mmax;
};

__device__ __forceinline__ double pair_dist64(float dx, float dy, float tx, float ty, float md, int dc, int tc) {
    const float ex = __fsub_rn(tx, dx), ey = __fsub_rn(ty, dy);
    // correctly rounded float32 sqrt like numpy's: through the double sqrt (53 bits >= 2*24+2, so the double rounding is exact)
    const float d = (float)sqrt((double)__fadd_rn(__fmul_rn(ex, ex), __fmul_rn(ey, ey)));
    const bool invalid = (d > md) || (dc != tc);
    return (double)d + (invalid ? 1e18 : 0.0);
}

constexpr int GREEDY_MAX_M = 4096;

__global__ __launch_bounds__(64) void center_greedy_kernel(GreedyArgs a) {
    __shared__ unsigned taken[GREEDY_MAX_M / 32];
    const int s = blockIdx.x, lane = threadIdx.x;
    const int N = a.n[s], M = a.m[s];
    const float* dxy = a.det_xy + (size_t)s * a.Nmax * 2;
    const float* txy = a.trk_xy + (size_t)s * a.Mmax * 2;
    const int* dcat = a.det_cat + (size_t)s * a.Nmax;
    const int* tcat = a.trk_cat + (size_t)s * a.Mmax;
    const float* md = a.max_diff + (size_t)s * a.Nmax;
    int* match = a.match + (size_t)s * a.Nmax;
    for (int w = lane; w < GREEDY_MAX_M / 32; w += 64) taken[w] = 0u;
    for (int i = lane; i < a.Nmax; i += 64) match[i] = -1;
    if (a.dist || a.row_any || a.col_any) {
        // the flags are what the tracker's newborn / dead rules need from the matrix ((dist <= gate).sum() > 0 along a row
        // or a column, pub_tracker.py:156,178): with them the N x M float64 matrix does not have to travel to the host
        double* D = a.dist ? a.dist + (size_t)s * a.Nmax * a.Mmax : nullptr;
        unsigned long long colbits = 0ull;  // bit k: column lane + 64 k has a valid pair
        for (int i = 0; i < N; ++i) {
            const float dx = dxy[2 * i], dy = dxy[2 * i + 1], mdi = md[i];
            const int dc = dcat[i];
            bool any = false;
            for (int j = lane, k = 0; j < M; j += 64, ++k) {
                const double v = pair_dist64(dx, dy, txy[2 * j], txy[2 * j + 1], mdi, dc, tcat[j]);
                if (D) D[(size_t)i * a.Mmax + j] = v;
                if (v < 1e16) {
                    any = true;
                    colbits |= 1ull << k;
                }
            }
            const bool row_hit = __any(any);
            if (a.row_any && lane == 0) a.row_any[(size_t)s * a.Nmax + i] = row_hit ? 1 : 0;
        }
        if (a.col_any)
            for (int j = lane, k = 0; j < M; j += 64, ++k) a.col_any[(size_t)s * a.Mmax + j] = (int)((colbits >> k) & 1ull);
    }
    __builtin_amdgcn_wave_barrier();
    if (M == 0) return;
    for (int i = 0; i < N; ++i) {
        const float dx = dxy[2 * i], dy = dxy[2 * i + 1], mdi = md[i];
        const int dc = dcat[i];
        double best = 1.0e300;
        int bj = 0x7fffffff;
        for (int j = lane; j < M; j += 64) {
            const bool tk = (taken[j >> 5] >> (j & 31)) & 1u;
            const double v = tk ? 1e18 : pair_dist64(dx, dy, txy[2 * j], txy[2 * j + 1], mdi, dc, tcat[j]);
            if (v < best) {  // ascending j per lane: the first minimum of this lane's columns
                best = v;
                bj = j;
            }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const double ov = __shfl_xor(best, off, 64);
            const int oj = __shfl_xor(bj, off, 64);
            if (ov < best || (ov == best && oj < bj)) {
                best = ov;
                bj = oj;
            }
        }
        if (best < 1e16) {
            if (lane == 0) {
                match[i] = bj;
                taken[bj >> 5] |= 1u << (bj & 31);
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

}  // namespace shasta

using namespace shasta;

extern "C" int shasta_center_greedy_f32(const float* det_xy, const float* trk_xy, const int32_t* det_cat, const int32_t* trk_cat,
                                        const float* max_diff, const int32_t* n, const int32_t* m, int scenes, int Nmax, int Mmax,
                                        double* dist, int32_t* match, int32_t* row_any, int32_t* col_any, shasta_stream_t stream) {
    SHASTA_REQUIRE(det_xy && trk_xy && det_cat && trk_cat && max_diff && n && m && match, "center_greedy: null pointer");
    SHASTA_REQUIRE(scenes >= 0 && Nmax >= 1 && Mmax >= 1 && Mmax <= GREEDY_MAX_M, "center_greedy: bad size (Mmax <= 4096)");
    if (scenes == 0) return SHASTA_OK;
    GreedyArgs a{det_xy, trk_xy, det_cat, trk_cat, max_diff, n, m, dist, match, row_any, col_any, Nmax, Mmax};
    hipLaunchKernelGGL(center_greedy_kernel, dim3(scenes), dim3(64), 0, as_stream(stream), a);
    return check_launch("center_greedy");
}
