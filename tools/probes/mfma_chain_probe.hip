// Issue cost of MFMA accumulator chains, one wave: v_mfma_f32_16x16x32_f16 16 cycles whatever the chaining; v_mfma_f32_4x4x1_f32 12.2 cycles in a
// single dependent chain, 8.1 with two or more alternating chains.  hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_chain_probe.hip -o /tmp/mfmachain
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
#define LOOPS 200
template <int OP, int CHAINS>
__global__ void k(float* out, unsigned long long* cyc, int slot) {
    f32x4 acc[8]; for (int i = 0; i < 8; ++i) acc[i] = f32x4{0, 0, 0, 0};
    h8 a, b; for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.01f + i); b[i] = (_Float16)(i * 0.5f); }
    float fa = threadIdx.x * 0.1f, fb = 0.25f;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int l = 0; l < LOOPS; ++l) {
#pragma unroll
        for (int r = 0; r < 64 / CHAINS; ++r)
#pragma unroll
            for (int i = 0; i < CHAINS; ++i) {
                if (OP == 0) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[i], 0, 0, 0);
                if (OP == 1) acc[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(fa, fb, acc[i], 0, 0, 0);
            }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0; for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][3];
    out[threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[slot] = t1 - t0;
}
int main() {
    float* out; unsigned long long* cyc; hipMalloc(&out, 1024); hipMalloc(&cyc, 128);
    hipLaunchKernelGGL((k<0, 1>), dim3(1), dim3(64), 0, 0, out, cyc, 0); hipLaunchKernelGGL((k<0, 3>), dim3(1), dim3(64), 0, 0, out, cyc, 1);
    hipLaunchKernelGGL((k<0, 8>), dim3(1), dim3(64), 0, 0, out, cyc, 2); hipLaunchKernelGGL((k<1, 1>), dim3(1), dim3(64), 0, 0, out, cyc, 3);
    hipLaunchKernelGGL((k<1, 2>), dim3(1), dim3(64), 0, 0, out, cyc, 4); hipLaunchKernelGGL((k<1, 4>), dim3(1), dim3(64), 0, 0, out, cyc, 5);
    hipLaunchKernelGGL((k<1, 8>), dim3(1), dim3(64), 0, 0, out, cyc, 6);
    unsigned long long h[8]; hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
    const char* n[7] = {"mfma 16x16x32 f16, 1 chain", "mfma 16x16x32 f16, 3 chains", "mfma 16x16x32 f16, 8 chains", "mfma 4x4x1 f32, 1 chain", "mfma 4x4x1 f32, 2 chains", "mfma 4x4x1 f32, 4 chains", "mfma 4x4x1 f32, 8 chains"};
    for (int i = 0; i < 7; ++i) printf("%-30s %.2f cycles per instruction (one wave)\n", n[i], (double)h[i] / (64 * LOOPS));
    return 0;
}
