// aff_frame_kernel (aff_pieces.hip: six aff layers + both softmaxes in one pass, sibling workgroups exchange column partials) against
// the two-kernel form's logits: the logits of aff_pieces_kernel are downloaded and both softmaxes are recomputed on the host in double.
// Two residual buffers alternate between launches, so a sibling that read the previous launch's partials would be seen.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form=1 [-DSHASTA_AFF_STAMP] [-DAP_SHAPE_64] \
//         -Ishasta_amd/csrc -Iinclude tools/probes/aff_frame_probe.hip -o /tmp/affframe && /tmp/affframe [frame-pairs] [N] [checked frames]
#include "aff_pieces.hip"
#include "aff_f16.hip"

#include <algorithm>
#include <cmath>
#include <vector>

namespace shasta {
void set_error_msg(const char* m) { fprintf(stderr, "error: %s\n", m); }
void set_error(const char* what, hipError_t e) { fprintf(stderr, "%s: %s\n", what, hipGetErrorString(e)); }
}  // namespace shasta

__global__ void fill(float* p, size_t n, unsigned seed, float scale) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        unsigned h = (unsigned)(i * 2654435761u) ^ seed;
        h ^= h >> 13; h *= 0x5bd1e995u; h ^= h >> 15;
        p[i] = ((h & 0xffffff) / 8388608.0f - 1.0f) * scale;
    }
}

// the padding columns of the residual rows (D .. Dp - 1) hold whatever the caller left there
__global__ void poison_pad(float* p, int M, int D, int Dp) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < M)
        for (int d = D; d < Dp; ++d) p[(size_t)i * Dp + d] = __builtin_nanf("");
}

int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 512;
    const int N = argc > 2 ? atoi(argv[2]) : 500;
    const int NCHK = std::min(B, argc > 3 ? atoi(argv[3]) : 3);
    const float wscale = argc > 4 ? (float)atof(argv[4]) : 0.08f;
    const int T = N + 2, D = N + 2, Dp = (T + 3) / 4 * 4, M = B * T;
    const int kin[6] = {D, 128, 64, 32, 64, 128}, nout[6] = {128, 64, 32, 64, 128, D};
    shasta_weights w = {};
    w.max_obj = N;
    for (int i = 0; i < 6; ++i) {
        float *W, *b;
        hipMalloc(&W, (size_t)kin[i] * nout[i] * 4);
        hipMalloc(&b, (size_t)nout[i] * 4);
        hipLaunchKernelGGL(fill, dim3(256), dim3(256), 0, 0, W, (size_t)kin[i] * nout[i], 31u * i + 5, i == 5 ? wscale * 8 : wscale);
        hipLaunchKernelGGL(fill, dim3(4), dim3(256), 0, 0, b, (size_t)nout[i], 7u * i + 3, 0.05f);
        w.aff[i].weight = W;
        w.aff[i].bias = b;
    }
    float *packed, *packed16, *res[2], *matched, *m1, *m2, *m1o;
    void* ws;
    const size_t wsb = shasta::aff_frame_workspace_bytes(B, N);
    hipMalloc(&packed, shasta::ap_layer_offset(6, D) * 256 * 4);
    hipMalloc(&packed16, shasta::ap16_total(D) * 4);
    hipMalloc(&res[0], (size_t)M * Dp * 4 + 4096);
    hipMalloc(&res[1], (size_t)M * Dp * 4 + 4096);
    hipMalloc(&matched, (size_t)M * Dp * 4);
    hipMalloc(&m1, (size_t)B * N * D * 4);
    hipMalloc(&m1o, (size_t)B * N * D * 4);
    hipMalloc(&m2, (size_t)B * T * N * 4);
    hipMalloc(&ws, wsb);
    hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, 0, res[0], (size_t)M * Dp, 99u, 1.0f);
    hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, 0, res[1], (size_t)M * Dp, 12345u, 3.0f);
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(poison_pad, dim3((M + 255) / 256), dim3(256), 0, 0, res[i], M, D, Dp);
    if (shasta::aff_pieces_pack(&w, packed, nullptr)) return 1;
    if (shasta::aff_f16_pack(&w, packed16, nullptr)) return 1;
    const bool f16 = getenv("AFF_F16") != nullptr;
    std::vector<float> hm((size_t)NCHK * T * Dp), h1((size_t)NCHK * N * D), h2((size_t)NCHK * T * N), hmm((size_t)NCHK * T * Dp);
    int bad = 0;
    for (int round = 0; round < 6; ++round) {
        const float* r = res[round & 1];
        hipMemset(m1, 0xff, (size_t)B * N * D * 4);
        hipMemset(m2, 0xff, (size_t)B * T * N * 4);
        if (shasta::launch_aff_pieces(&w, packed, r, Dp, matched, Dp, m1o, M, nullptr)) return 1;
        hipMemcpy(hm.data(), matched, hm.size() * 4, hipMemcpyDeviceToHost);
        hipMemset(matched, 0, (size_t)M * Dp * 4);
        if (f16 ? shasta::launch_aff_frame16(&w, packed16, r, Dp, matched, Dp, m1, m2, B, ws, nullptr)
                : shasta::launch_aff_frame(&w, packed, r, Dp, (round & 2) ? matched : nullptr, Dp, m1, m2, B, ws, nullptr))
            return 1;
        if (hipDeviceSynchronize() != hipSuccess) return 2;
        hipMemcpy(h1.data(), m1, h1.size() * 4, hipMemcpyDeviceToHost);
        hipMemcpy(h2.data(), m2, h2.size() * 4, hipMemcpyDeviceToHost);
        hipMemcpy(hmm.data(), matched, hmm.size() * 4, hipMemcpyDeviceToHost);
        double e1 = 0, e2 = 0, em = 0, lmax = 0;
        if (f16) {  // the fp16 kernel's own logits are the softmax reference; their distance to the bf16-piece kernel's is reported
            for (size_t i = 0; i < hm.size(); ++i) em = std::max(em, (double)std::fabs(hm[i] - hmm[i]));
            hm = hmm;
        }
        long arg1 = 0, arg2 = 0;
        for (int b = 0; b < NCHK; ++b) {
            const float* x = hm.data() + (size_t)b * T * Dp;
            for (int t = 0; t < T; ++t)
                for (int d = 0; d < D; ++d) lmax = std::max(lmax, (double)std::fabs(x[t * Dp + d]));
            if ((round & 2) && !f16)
                for (int t = 0; t < T; ++t)
                    for (int d = 0; d < Dp; ++d) em = std::max(em, (double)std::fabs(x[t * Dp + d] - hmm[((size_t)b * T + t) * Dp + d]));
            for (int t = 0; t < N; ++t) {
                double mx = -1e300, s = 0;
                int am = 0, am2 = 0;
                for (int d = 0; d < D; ++d)
                    if (x[t * Dp + d] > mx) mx = x[t * Dp + d], am = d;
                for (int d = 0; d < D; ++d) s += std::exp((double)x[t * Dp + d] - mx);
                const float* o = h1.data() + ((size_t)b * N + t) * D;
                float best = -1;
                for (int d = 0; d < D; ++d) {
                    const double ref = std::exp((double)x[t * Dp + d] - mx) / s;
                    e1 = std::max(e1, std::fabs(ref - o[d]));
                    if (!(o[d] <= best)) best = o[d], am2 = d;
                }
                arg1 += am != am2;
            }
            for (int d = 0; d < N; ++d) {
                double mx = -1e300, s = 0;
                int am = 0, am2 = 0;
                for (int t = 0; t < T; ++t)
                    if (x[t * Dp + d] > mx) mx = x[t * Dp + d], am = t;
                for (int t = 0; t < T; ++t) s += std::exp((double)x[t * Dp + d] - mx);
                float best = -1;
                for (int t = 0; t < T; ++t) {
                    const double ref = std::exp((double)x[t * Dp + d] - mx) / s;
                    const float o = h2[((size_t)b * T + t) * N + d];
                    e2 = std::max(e2, std::fabs(ref - o));
                    if (!(o <= best)) best = o, am2 = t;
                }
                arg2 += am != am2;
            }
        }
        printf("round %d: max|logit| %.3g  max|m1 - ref| %.3g  max|m2 - ref| %.3g  argmax mismatches %ld / %ld  matched diff %.3g\n", round, lmax, e1, e2,
               arg1, arg2, em);
        if (!(e1 < 1e-6) || !(e2 < 1e-6) || arg1 || arg2 || (f16 ? !(em <= 3e-6 * lmax) : em != 0)) bad = 1;
    }
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int reps = 50;
    float ms = 0;
    for (int r = 0; r < 5; ++r) shasta::launch_aff_pieces(&w, packed, res[r & 1], Dp, matched, Dp, m1o, M, nullptr);
    hipEventRecord(e0, nullptr);
    for (int r = 0; r < reps; ++r) shasta::launch_aff_pieces(&w, packed, res[r & 1], Dp, matched, Dp, m1o, M, nullptr);
    hipEventRecord(e1, nullptr);
    if (hipDeviceSynchronize() != hipSuccess) return 2;
    hipEventElapsedTime(&ms, e0, e1);
    printf("B=%d N=%d  aff_pieces (without the column softmax) %.3f ms\n", B, N, ms / reps);
    for (int r = 0; r < 5 + reps; ++r) {
        if (r == 5) hipEventRecord(e0, nullptr);
        if (f16) shasta::launch_aff_frame16(&w, packed16, res[r & 1], Dp, nullptr, Dp, m1, m2, B, ws, nullptr);
        else shasta::launch_aff_frame(&w, packed, res[r & 1], Dp, nullptr, Dp, m1, m2, B, ws, nullptr);
    }
    hipEventRecord(e1, nullptr);
    if (hipDeviceSynchronize() != hipSuccess) return 2;
    hipEventElapsedTime(&ms, e0, e1);
    printf("B=%d N=%d  aff_frame%s (everything) %.3f ms  [%s]\n", B, N, f16 ? "16" : "", ms / reps, bad ? "MISMATCH" : "ok");
#ifdef SHASTA_AFF_STAMP
    static unsigned long long h[4096][8];
    hipMemcpyFromSymbol(h, HIP_SYMBOL(shasta::g_aff_stamp), sizeof(h));
    const char* names[7] = {"layer 1 (load + cut + mfma)", "layers 2-5", "layer 6 mfma", "bias + statistics + publish", "staging + wait + combine", "write-out", "-"};
    const int nwg = std::min(4096, B * ((T + 63) / 64));
    for (int ph = 0; ph < 6; ++ph) {
        std::vector<double> v;
        for (int i = 0; i < nwg; ++i)
            if (h[i][ph + 1] > h[i][ph]) v.push_back((double)(h[i][ph + 1] - h[i][ph]));
        std::sort(v.begin(), v.end());
        if (!v.empty()) printf("%-30s s_memtime ticks (100 MHz): p10 %.0f median %.0f p90 %.0f\n", names[ph], v[v.size() / 10], v[v.size() / 2], v[v.size() * 9 / 10]);
    }
#endif
    return bad;
}
