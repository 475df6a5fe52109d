// Probe of the v_mfma_f32_4x4x1_16B_f32 operand layout on gfx950 (run on the GPU box):
//   hipcc --offload-arch=gfx950 -O2 mfma4x4_probe.hip -o /tmp/probe && /tmp/probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(float* out, long long* cyc) {
    const int l = threadIdx.x;
    // A[block][i] = 100*block + 10*i + 1 ; B[block][j] = 1000 + j  (exact in fp32)
    const float a = 100.0f * (l / 4) + 10.0f * (l % 4) + 1.0f;
    const float b = 1000.0f + (l % 4) + 0.0f * (l / 4);
    f32x4 c = {0, 0, 0, 0};
    f32x4 d = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) out[l * 4 + r] = d[r];
    // broadcast A of block 0 to all blocks: cbsz = 4, abid = 0
    f32x4 e = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 4, 0, 0);
    for (int r = 0; r < 4; ++r) out[256 + l * 4 + r] = e[r];
    // timing: N MFMAs on 4 independent accumulators, results consumed before the second time stamp
    f32x4 acc[4] = {c, c, c, c};
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < 256; ++it)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[q], 0, 0, 0);
    float sink = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
    asm volatile("" : "+v"(sink));
    long long t1 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < 256; ++it)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[q], 0, 0, 0);
    float sink2 = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
    asm volatile("" : "+v"(sink2));
    long long t2 = __builtin_amdgcn_s_memtime();
    out[512 + l] = sink + sink2;
    if (l == 0) { cyc[0] = t1 - t0; cyc[1] = t2 - t1; }
}
int main() {
    float* d; long long* c;
    hipMalloc(&d, 1024 * 4); hipMalloc(&c, 16);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, c);
    float h[1024]; long long hc[2];
    hipMemcpy(h, d, 1024 * 4, hipMemcpyDeviceToHost); hipMemcpy(hc, c, 16, hipMemcpyDeviceToHost);
    for (int l = 0; l < 12; ++l) printf("lane %2d: d = %.0f %.0f %.0f %.0f   | bcast: %.0f %.0f %.0f %.0f\n", l, h[l*4], h[l*4+1], h[l*4+2], h[l*4+3], h[256+l*4], h[256+l*4+1], h[256+l*4+2], h[256+l*4+3]);
    for (int l = 60; l < 64; ++l) printf("lane %2d: d = %.0f %.0f %.0f %.0f   | bcast: %.0f %.0f %.0f %.0f\n", l, h[l*4], h[l*4+1], h[l*4+2], h[l*4+3], h[256+l*4], h[256+l*4+1], h[256+l*4+2], h[256+l*4+3]);
    printf("1024 MFMA 4x4x1: %lld ticks -> %.2f each ; 1024 MFMA 16x16x4: %lld ticks -> %.2f each (ratio %.2f)\n", hc[0], hc[0] / 1024.0, hc[1], hc[1] / 1024.0, (double)hc[1] / hc[0]);
    return 0;
}
