// Diagnostic build of the bf16-piece anchor kernel (anchor_split.hip) with in-kernel stamps: clock = d(s_memtime) /
// d(s_memrealtime) * 100 MHz, cycles per tile = d(s_memtime) / tiles, after back-to-back launches on random data.
// Build variants with -DSPLIT_EXP_NOCUT / -DSPLIT_EXP_NOBAR / -DSPLIT_EXP_NOMFMA (wrong results, timing only) to see what a
// tile's time is made of.  Run on the GPU box from the repo root:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DSHASTA_L1_STAMP -Ishasta_amd/csrc \
//         tools/probes/l1_split_probe.hip -o /tmp/splitprobe && /tmp/splitprobe [batch]
#include "anchor_split.hip"

#include <algorithm>
#include <vector>

namespace shasta {
void set_error_msg(const char*) {}
}  // namespace shasta

__global__ void fill(float* p, size_t n, unsigned seed) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        unsigned h = (unsigned)(i * 2654435761u) ^ seed;
        h ^= h >> 13; h *= 0x5bd1e995u; h ^= h >> 15;
        p[i] = ((h & 0xffffff) / 8388608.0f - 1.0f) * 0.05f;
    }
}

int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 64;
    const int N = 500, F = 256, K = N * F, H = K / 64, T = N + 2;
    float* W[4];
    for (int i = 0; i < 4; ++i) {
        if (hipMalloc(&W[i], (size_t)H * K * 4) != hipSuccess) return 1;
        hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, 0, W[i], (size_t)H * K, 17u * i + 1);
    }
    float *x0, *x1, *part;
    void* xs;
    hipMalloc(&x0, (size_t)B * T * F * 4);
    hipMalloc(&x1, (size_t)B * T * F * 4);
    hipMalloc(&part, (size_t)64 * B * 4 * H * 4);
    hipMalloc(&xs, shasta::anchor_split_workspace_bytes(B, K));
    hipLaunchKernelGGL(fill, dim3(1024), dim3(256), 0, 0, x0, (size_t)B * T * F, 99u);
    hipLaunchKernelGGL(fill, dim3(1024), dim3(256), 0, 0, x1, (size_t)B * T * F, 77u);
    shasta::launch_split_x(x0, x1, xs, K, B, T * F, nullptr);
    int ks = 0;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int reps = 200;  // > 0.1 s of back-to-back launches
    for (int r = 0; r < 20; ++r) shasta::launch_anchor_l1_split(W, xs, part, H, K, B, &ks, nullptr);
    hipEventRecord(e0, nullptr);
    for (int r = 0; r < reps; ++r) shasta::launch_anchor_l1_split(W, xs, part, H, K, B, &ks, nullptr);
    hipEventRecord(e1, nullptr);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    static unsigned long long h[4096][3];
    hipMemcpyFromSymbol(h, HIP_SYMBOL(shasta::g_split_stamp), sizeof(h));
    std::vector<double> clk, cpt;
    for (int i = 0; i < 4096; ++i)
        if (h[i][2] > 0 && h[i][1] > 0) {
            clk.push_back((double)h[i][0] / (double)h[i][1] * 0.1);  // GHz
            cpt.push_back((double)h[i][0] / (double)h[i][2]);
        }
    std::sort(clk.begin(), clk.end());
    std::sort(cpt.begin(), cpt.end());
    printf("B=%d KS=%d workgroups stamped=%zu  launch %.3f ms\n", B, ks, clk.size(), ms / reps);
    if (!clk.empty())
        printf("in-kernel clock GHz: min %.3f median %.3f max %.3f ; shader cycles per 32-float tile: min %.0f median %.0f max %.0f\n",
               clk.front(), clk[clk.size() / 2], clk.back(), cpt.front(), cpt[cpt.size() / 2], cpt.back());
    return 0;
}
