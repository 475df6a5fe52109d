// Diagnostic build of aff_pieces_kernel (aff_pieces.hip) with s_memtime stamps at its phase boundaries: where do the ~110 us
// of a workgroup (128 residual rows) go?  Back-to-back launches on random data, median over the stamped workgroups.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DSHASTA_AFF_STAMP -Ishasta_amd/csrc -Iinclude \
//         tools/probes/aff_probe.hip -o /tmp/affprobe && /tmp/affprobe [frame-pairs]
#include "aff_pieces.hip"

#include <algorithm>
#include <vector>

namespace shasta {
void set_error_msg(const char*) {}
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

int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 512;
    const int N = 500, T = N + 2, D = N + 2, Dp = 504, M = B * T;
    const int kin[6] = {D, 128, 64, 32, 64, 128}, nout[6] = {128, 64, 32, 64, 128, D};
    shasta_weights w = {};
    w.max_obj = N;
    for (int i = 0; i < 6; ++i) {
        float *W, *b;
        hipMalloc(&W, (size_t)kin[i] * nout[i] * 4);
        hipMalloc(&b, (size_t)nout[i] * 4);
        hipLaunchKernelGGL(fill, dim3(256), dim3(256), 0, 0, W, (size_t)kin[i] * nout[i], 31u * i + 5, 0.08f);
        hipLaunchKernelGGL(fill, dim3(4), dim3(256), 0, 0, b, (size_t)nout[i], 7u * i + 3, 0.05f);
        w.aff[i].weight = W;
        w.aff[i].bias = b;
    }
    float *packed, *res, *matched, *m1;
    hipMalloc(&packed, shasta::ap_layer_offset(6, D) * 256 * 4);
    hipMalloc(&res, (size_t)M * Dp * 4 + 4096);
    hipMalloc(&matched, (size_t)M * Dp * 4);
    hipMalloc(&m1, (size_t)B * N * D * 4);
    hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, 0, res, (size_t)M * Dp, 99u, 1.0f);
    if (shasta::aff_pieces_pack(&w, packed, nullptr)) return 1;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int reps = 100;
    for (int r = 0; r < 10; ++r) shasta::launch_aff_pieces(&w, packed, res, Dp, matched, Dp, m1, M, nullptr);
    hipEventRecord(e0, nullptr);
    for (int r = 0; r < reps; ++r) shasta::launch_aff_pieces(&w, packed, res, Dp, matched, Dp, m1, M, nullptr);
    hipEventRecord(e1, nullptr);
    if (hipDeviceSynchronize() != hipSuccess) return 2;
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    printf("B=%d rows=%d workgroups=%d  launch %.3f ms\n", B, M, (M + 127) / 128, ms / reps);
#ifdef SHASTA_AFF_STAMP
    static unsigned long long h[4096][8];
    hipMemcpyFromSymbol(h, HIP_SYMBOL(shasta::g_aff_stamp), sizeof(h));
    const char* names[7] = {"layer 1 (load + cut + mfma)", "layers 2-5", "layer 6 mfma", "bias + softmax stats", "write-out (staging, stores)", "-", "-"};
    const int nwg = std::min(4096, (M + 63) / 64);
    for (int ph = 0; ph < 5; ++ph) {
        std::vector<double> v;
        for (int i = 0; i < nwg; ++i)
            if (h[i][ph + 1] > h[i][ph]) v.push_back((double)(h[i][ph + 1] - h[i][ph]));
        std::sort(v.begin(), v.end());
        if (!v.empty()) printf("%-30s s_memtime ticks (100 MHz): p10 %.0f median %.0f p90 %.0f\n", names[ph], v[v.size() / 10], v[v.size() / 2], v[v.size() * 9 / 10]);
    }
#endif
    return 0;
}
