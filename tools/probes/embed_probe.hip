// Diagnostic build of embed_rows_kernel (embed_rows.hip) with s_memtime stamps at its phase boundaries.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form -DSHASTA_EMBED_STAMP -Ishasta_amd/csrc -Iinclude \
//         tools/probes/embed_probe.hip -o /tmp/embedprobe && /tmp/embedprobe [frame-pairs]
#include "embed_rows.hip"

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
    const int N = 500, T = N + 2, F = 256, nf = 7, M = B * T;
    shasta_weights w = {};
    w.max_obj = N;
    w.num_feats = nf;
    w.feat_dim = F;
    const shasta::PackedLayout P(N, nf, F);
    const shasta::PairDims d(F);
    float *packed, *x0, *x1, *t0, *t1, *up, *uc, *h0, *h1;
    hipMalloc(&packed, P.total * 4);
    hipMalloc(&x0, (size_t)M * F * 4); hipMalloc(&x1, (size_t)M * F * 4);
    hipMalloc(&t0, (size_t)M * 8 * 4); hipMalloc(&t1, (size_t)M * 8 * 4);
    hipMalloc(&up, (size_t)M * d.ET * 4); hipMalloc(&uc, (size_t)M * d.ET * 4);
    hipMalloc(&h0, (size_t)M * 16 * 4); hipMalloc(&h1, (size_t)M * 16 * 4);
    hipLaunchKernelGGL(fill, dim3(1024), dim3(256), 0, 0, packed, P.total, 3u, 0.05f);
    hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, 0, x0, (size_t)M * F, 5u, 1.0f);
    hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, 0, x1, (size_t)M * F, 7u, 1.0f);
    hipLaunchKernelGGL(fill, dim3(256), dim3(256), 0, 0, t0, (size_t)M * 8, 9u, 1.0f);
    hipLaunchKernelGGL(fill, dim3(256), dim3(256), 0, 0, t1, (size_t)M * 8, 11u, 1.0f);
    if (shasta::embed_pack(&w, packed, nullptr)) return 1;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int reps = 100;
    for (int r = 0; r < 10; ++r) shasta::launch_embed_rows(&w, packed, x0, x1, t0, t1, up, uc, h0, h1, M, nullptr);
    hipEventRecord(e0, nullptr);
    for (int r = 0; r < reps; ++r) shasta::launch_embed_rows(&w, packed, x0, x1, t0, t1, up, uc, h0, h1, M, nullptr);
    hipEventRecord(e1, nullptr);
    if (hipDeviceSynchronize() != hipSuccess) return 2;
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    printf("B=%d rows=%d workgroups=2 x %d  launch %.3f ms\n", B, M, (M + 255) / 256, ms / reps);
#ifdef SHASTA_EMBED_STAMP
    static unsigned long long h[4096][8];
    hipMemcpyFromSymbol(h, HIP_SYMBOL(shasta::g_embed_stamp), sizeof(h));
    const char* names[4] = {"prologue (box weights, box row)", "layer (DMA ring, cut, mfma)", "staging write", "row pass (box part, max, store)"};
    const int nwg = std::min(4096, (M + 255) / 256);
    for (int ph = 0; ph < 4; ++ph) {
        std::vector<double> v;
        for (int i = 0; i < nwg; ++i)
            if (h[i][ph + 1] > h[i][ph]) v.push_back((double)(h[i][ph + 1] - h[i][ph]));
        std::sort(v.begin(), v.end());
        if (!v.empty()) printf("%-34s cycles: p10 %.0f median %.0f p90 %.0f\n", names[ph], v[v.size() / 10], v[v.size() / 2], v[v.size() * 9 / 10]);
    }
#endif
    return 0;
}
