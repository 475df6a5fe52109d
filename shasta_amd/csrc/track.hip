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


// ---- the merged tracker of a whole scene in one launch --------------------------------------------------------------------------
// tools/nusc_shasta/pub_tracker_merged.py:57-225 as pub_test.py:88-162 drives it (one tracker per scene, reset at the scene's first
// frame): every frame of the scene, in order - per-class association by the matrix / gate / greedy rule above, confidence refinement
// (TRK_REF), the `newborn` / `dead` suppression rules, coasting of unmatched tracks up to max_age with ct += tracking * -1.  The host
// keeps only the conversions: detection dicts -> arrays, results -> rows.  One wavefront per scene; the track list and the frame's
// detections live in LDS; bookkeeping is serial on lane 0 (it IS serial: ids, ages, list order), distances and the argmin run over the
// lanes.  Greedy over all detections of a frame in file order equals the per-class problems of step_batch_merged: a pair of different
// classes is invalid (1e18), so a taken column only ever matters to detections of its own class, and "first minimum" over the whole
// track list picks the same track as over the list restricted to the class (relative order is kept).
constexpr int TM_TCAP = 768;   // tracks of a scene alive at one time (matched + new + coasting)
constexpr int TM_DCAP = 512;   // detections of one frame
constexpr int TM_NCLS = 8;
// dynamic LDS (one workgroup = one wavefront per CU): 2 track buffers x 60 B x TM_TCAP + 64 B x TM_DCAP + flags = ~130 KB
constexpr size_t TM_LDS_BYTES = (size_t)2 * TM_TCAP * (5 * 8 + 5 * 4) + (size_t)TM_DCAP * (6 * 8 + 2 * 4 + 4 * 4) + (size_t)TM_TCAP * 4 + TM_TCAP / 32 * 4;

struct TrackMergedArgs {
    const double* det_xy;     // (D, 2) translation[:2]
    const double* det_vel;    // (D, 2) velocity[:2]
    const int* det_cls;       // (D,) tracking-class label 0 .. ncls-1, or -1 (ignored)
    const double* det_score;  // (D,) detection_score
    const double* det_ref;    // (D,) ref_detection_score of the decode
    const int* det_flags;     // (D,) bit 0: 'newborn' in det, bit 1: 'dead' in det
    const int* frame_off;     // (S, Fmax + 1) offsets into the detection arrays
    const double* frame_lag;  // (S, Fmax) time_lag of step_centertrack
    const int* n_frames;      // (S,)
    int* out_status;          // (D,) 0: not in the frame's result, 1: matched to a track, 2: new track
    int* out_id;              // (D,) tracking_id
    double* out_ref;          // (D,) refined ref_detection_score
    int* out_err;             // (S,) 0 ok, 1: a frame holds more than TM_DCAP detections, 2: more than TM_TCAP tracks
    int Fmax, max_age, ncls;
    int plain;                // 0: PubTrackerMerged (class by class, TRK_REF tables); 1: PubTracker (one list, refine / alpha[0] / beta[0])
    float gate[TM_NCLS];
    int ref_on[TM_NCLS];
    double alpha[TM_NCLS], beta[TM_NCLS];
};

__global__ __launch_bounds__(64) void track_merged_kernel(TrackMergedArgs a) {
    extern __shared__ __attribute__((aligned(16))) double tm_lds[];
    // carve-up: doubles first, then the 4-byte arrays
    double (*t_cx)[TM_TCAP] = reinterpret_cast<double (*)[TM_TCAP]>(tm_lds);
    double (*t_cy)[TM_TCAP] = t_cx + 2;
    double (*t_tx)[TM_TCAP] = t_cy + 2;
    double (*t_ty)[TM_TCAP] = t_tx + 2;
    double (*t_ref)[TM_TCAP] = t_ty + 2;
    double* d_cx = reinterpret_cast<double*>(t_ref + 2);
    double* d_cy = d_cx + TM_DCAP;
    double* d_tx = d_cy + TM_DCAP;
    double* d_ty = d_tx + TM_DCAP;
    double* d_sc = d_ty + TM_DCAP;
    double* d_rf = d_sc + TM_DCAP;
    int (*t_id)[TM_TCAP] = reinterpret_cast<int (*)[TM_TCAP]>(d_rf + TM_DCAP);
    int (*t_age)[TM_TCAP] = t_id + 2;
    int (*t_act)[TM_TCAP] = t_age + 2;
    int (*t_cls)[TM_TCAP] = t_act + 2;
    int (*t_flg)[TM_TCAP] = t_cls + 2;
    float* d_fx = reinterpret_cast<float*>(t_flg + 2);
    float* d_fy = d_fx + TM_DCAP;
    int* d_cl = reinterpret_cast<int*>(d_fy + TM_DCAP);
    int* d_fl = d_cl + TM_DCAP;
    int* d_match = d_fl + TM_DCAP;
    int* d_near = d_match + TM_DCAP;
    int* t_near = d_near + TM_DCAP;
    unsigned* taken = reinterpret_cast<unsigned*>(t_near + TM_TCAP);
    const int s = blockIdx.x, lane = threadIdx.x;
    const int* off = a.frame_off + (size_t)s * (a.Fmax + 1);
    const int F = a.n_frames[s];
    int cur = 0, nt = 0, idc = 0, err = 0;  // uniform: current track buffer, tracks in it, id counter
    for (int f = 0; f < F && !err; ++f) {
        const int g0 = off[f], n = off[f + 1] - g0;
        const double lag = a.frame_lag[(size_t)s * a.Fmax + f];
        if (n == 0) {  // pub_tracker_merged.py:85-87: an empty frame drops every track
            nt = 0;
            continue;
        }
        if (n > TM_DCAP) {
            err = 1;
            break;
        }
        for (int i = lane; i < n; i += 64) {
            const double cx = a.det_xy[2 * (size_t)(g0 + i)], cy = a.det_xy[2 * (size_t)(g0 + i) + 1];
            const double tx = (a.det_vel[2 * (size_t)(g0 + i)] * -1.0) * lag, ty = (a.det_vel[2 * (size_t)(g0 + i) + 1] * -1.0) * lag;
            d_cx[i] = cx; d_cy[i] = cy; d_tx[i] = tx; d_ty[i] = ty;
            d_fx[i] = (float)(cx + (double)(float)tx);  // (ct + tracking.astype(float32)).astype(float32)
            d_fy[i] = (float)(cy + (double)(float)ty);
            d_sc[i] = a.det_score[g0 + i]; d_rf[i] = a.det_ref[g0 + i];
            d_cl[i] = a.det_cls[g0 + i]; d_fl[i] = a.det_flags[g0 + i];
            d_match[i] = -1; d_near[i] = 0;
        }
        for (int j = lane; j < nt; j += 64) t_near[j] = 0;
        for (int w = lane; w < TM_TCAP / 32; w += 64) taken[w] = 0u;
        __syncthreads();
        // which detections / tracks have a partner inside the gate, then the greedy assignment in file order
        if (nt > 0) {
            for (int i = 0; i < n; ++i) {
                const int dc = d_cl[i];
                if (dc < 0) continue;
                const float md = a.gate[dc], dx = d_fx[i], dy = d_fy[i];
                bool any = false;
                for (int j = lane; j < nt; j += 64) {
                    const double v = pair_dist64(dx, dy, (float)t_cx[cur][j], (float)t_cy[cur][j], md, dc, t_cls[cur][j]);
                    if (v < 1e16) {
                        any = true;
                        t_near[j] = 1;
                    }
                }
                if (__any(any) && lane == 0) d_near[i] = 1;
            }
            __syncthreads();
            for (int i = 0; i < n; ++i) {
                const int dc = d_cl[i];
                if (dc < 0) continue;
                const float md = a.gate[dc], dx = d_fx[i], dy = d_fy[i];
                double best = 1.0e300;
                int bj = 0x7fffffff;
                for (int j = lane; j < nt; j += 64) {
                    const bool tk = (taken[j >> 5] >> (j & 31)) & 1u;
                    const double v = tk ? 1e18 : pair_dist64(dx, dy, (float)t_cx[cur][j], (float)t_cy[cur][j], md, dc, t_cls[cur][j]);
                    if (v < best) {
                        best = v;
                        bj = j;
                    }
                }
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) {
                    const double ov = __shfl_xor(best, o, 64);
                    const int oj = __shfl_xor(bj, o, 64);
                    if (ov < best || (ov == best && oj < bj)) {
                        best = ov;
                        bj = oj;
                    }
                }
                if (best < 1e16 && lane == 0) {
                    d_match[i] = bj;
                    taken[bj >> 5] |= 1u << (bj & 31);
                }
                __syncthreads();
            }
        }
        __syncthreads();
        // the new track list: class by class (pub_tracker_merged.py:100-223) or, for the plain tracker (pub_tracker.py:110-208), all
        // tracking classes as one group; serial by nature
        if (lane == 0) {
            const int nxt = cur ^ 1;
            int nr = 0;
            int cnt_t[TM_NCLS];
            for (int c = 0; c < a.ncls; ++c) cnt_t[c] = 0;
            for (int j = 0; j < nt; ++j) cnt_t[t_cls[cur][j]] += 1;
            const int ngroups = a.plain ? 1 : a.ncls;
            for (int c = 0; c < ngroups && !err; ++c) {
#define TM_DIN(i) (a.plain ? d_cl[i] >= 0 : d_cl[i] == c)
#define TM_TIN(j) (a.plain ? true : t_cls[cur][j] == c)
                bool has = false;
                for (int i = 0; i < n; ++i) has = has || TM_DIN(i);
                if (!has) continue;  // merged: nothing of this class in the frame: its tracks are dropped
                const double al = a.alpha[c], be = a.beta[c];
                const bool rf = a.ref_on[c] != 0;
                const int ntrk = a.plain ? nt : cnt_t[c];
                for (int i = 0; i < n && !err; ++i) {  // matched detections take over their track
                    if (!TM_DIN(i) || d_match[i] < 0) continue;
                    if (nr >= TM_TCAP) { err = 2; break; }
                    const int j = d_match[i];
                    const double refined = (d_rf[i] > al ? 1.0 : 0.0) * be * d_sc[i] + (1.0 - be) * t_ref[cur][j];
                    const double r = rf ? refined : (a.plain ? d_rf[i] : d_sc[i]);
                    t_cx[nxt][nr] = d_cx[i]; t_cy[nxt][nr] = d_cy[i]; t_tx[nxt][nr] = d_tx[i]; t_ty[nxt][nr] = d_ty[i];
                    t_ref[nxt][nr] = r; t_id[nxt][nr] = t_id[cur][j]; t_age[nxt][nr] = 1; t_act[nxt][nr] = t_act[cur][j] + 1;
                    t_cls[nxt][nr] = d_cl[i]; t_flg[nxt][nr] = d_fl[i];
                    a.out_status[g0 + i] = 1; a.out_id[g0 + i] = t_id[cur][j]; a.out_ref[g0 + i] = r;
                    ++nr;
                }
                for (int i = 0; i < n && !err; ++i) {  // unmatched detections start tracks unless suppressed
                    if (!TM_DIN(i) || d_match[i] >= 0) continue;
                    if (ntrk > 0 && !(d_fl[i] & 1) && d_near[i]) continue;
                    if (nr >= TM_TCAP) { err = 2; break; }
                    const double r = (rf && !a.plain) ? be * d_sc[i] : d_sc[i];
                    ++idc;
                    t_cx[nxt][nr] = d_cx[i]; t_cy[nxt][nr] = d_cy[i]; t_tx[nxt][nr] = d_tx[i]; t_ty[nxt][nr] = d_ty[i];
                    t_ref[nxt][nr] = r; t_id[nxt][nr] = idc; t_age[nxt][nr] = 1; t_act[nxt][nr] = 1;
                    t_cls[nxt][nr] = d_cl[i]; t_flg[nxt][nr] = d_fl[i];
                    a.out_status[g0 + i] = 2; a.out_id[g0 + i] = idc; a.out_ref[g0 + i] = r;
                    ++nr;
                }
                for (int j = 0; j < nt && !err; ++j) {  // unmatched tracks coast
                    if (!TM_TIN(j)) continue;
                    if ((taken[j >> 5] >> (j & 31)) & 1u) continue;
                    if ((t_flg[cur][j] & 2) && t_near[j]) continue;
                    if (t_age[cur][j] < a.max_age) {
                        if (nr >= TM_TCAP) { err = 2; break; }
                        t_cx[nxt][nr] = t_cx[cur][j] + t_tx[cur][j] * -1.0; t_cy[nxt][nr] = t_cy[cur][j] + t_ty[cur][j] * -1.0;
                        t_tx[nxt][nr] = t_tx[cur][j]; t_ty[nxt][nr] = t_ty[cur][j];
                        t_ref[nxt][nr] = (rf && !a.plain) ? (1.0 - be) * t_ref[cur][j] : t_ref[cur][j];
                        t_id[nxt][nr] = t_id[cur][j]; t_age[nxt][nr] = t_age[cur][j] + 1; t_act[nxt][nr] = 0;
                        t_cls[nxt][nr] = t_cls[cur][j]; t_flg[nxt][nr] = t_flg[cur][j];
                        ++nr;
                    }
                }
#undef TM_DIN
#undef TM_TIN
            }
            nt = nr;
        }
        nt = __shfl(nt, 0, 64);
        idc = __shfl(idc, 0, 64);
        err = __shfl(err, 0, 64);
        cur ^= 1;
        __syncthreads();
    }
    if (lane == 0) a.out_err[s] = err;
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

extern "C" int shasta_track_merged_f64(const double* det_xy, const double* det_vel, const int32_t* det_cls, const double* det_score,
                                       const double* det_ref, const int32_t* det_flags, const int32_t* frame_off, const double* frame_lag,
                                       const int32_t* n_frames, int scenes, int Fmax, int n_cls, const float* cls_gate,
                                       const int32_t* cls_ref, const double* cls_alpha, const double* cls_beta, int max_age, int plain,
                                       int32_t* out_status, int32_t* out_id, double* out_ref, int32_t* out_err, shasta_stream_t stream) {
    SHASTA_REQUIRE(det_xy && det_vel && det_cls && det_score && det_ref && det_flags && frame_off && frame_lag && n_frames, "track_merged: null input");
    SHASTA_REQUIRE(cls_gate && cls_ref && cls_alpha && cls_beta && out_status && out_id && out_ref && out_err, "track_merged: null pointer");
    SHASTA_REQUIRE(scenes >= 0 && Fmax >= 1 && n_cls >= 1 && n_cls <= TM_NCLS && max_age >= 0, "track_merged: bad size (at most 8 classes)");
    if (scenes == 0) return SHASTA_OK;
    TrackMergedArgs a;
    a.det_xy = det_xy; a.det_vel = det_vel; a.det_cls = det_cls; a.det_score = det_score; a.det_ref = det_ref; a.det_flags = det_flags;
    a.frame_off = frame_off; a.frame_lag = frame_lag; a.n_frames = n_frames;
    a.out_status = out_status; a.out_id = out_id; a.out_ref = out_ref; a.out_err = out_err;
    a.Fmax = Fmax; a.max_age = max_age; a.ncls = n_cls; a.plain = plain ? 1 : 0;
    for (int c = 0; c < TM_NCLS; ++c) {  // (host arrays: class constants)
        a.gate[c] = c < n_cls ? cls_gate[c] : 0.0f;
        a.ref_on[c] = c < n_cls ? cls_ref[c] : 0;
        a.alpha[c] = c < n_cls ? cls_alpha[c] : 0.0;
        a.beta[c] = c < n_cls ? cls_beta[c] : 0.0;
    }
    if (hipFuncSetAttribute((const void*)track_merged_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)TM_LDS_BYTES) != hipSuccess) {
        (void)hipGetLastError();
        set_error_msg("track_merged: the device does not grant the kernel's LDS (one workgroup holds a scene's tracks): use the per-frame tracker");
        return SHASTA_E_UNSUPPORTED;
    }
    hipLaunchKernelGGL(track_merged_kernel, dim3(scenes), dim3(64), TM_LDS_BYTES, as_stream(stream), a);
    return check_launch("track_merged");
}
