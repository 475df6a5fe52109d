// Issue cost of the VALU instructions of the per-pair cut (pair_f16.hip), one wave, eight independent chains, s_memtime around 12 800 instructions.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/valu_cost_probe.hip -o /tmp/valucost && /tmp/valucost
// MI355X: v_fma_f32 6.3, v_pk_fma_f32 5.3, v_pk_mul_f32 5.3, v_fma_mix_f32 6.3, v_fma_mixlo/hi_f16 8.9, v_cvt_pk_f16_f32 8.1, v_max_f32 4.8 cycles.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float pf2 __attribute__((ext_vector_type(2)));
#define REP 64
#define LOOPS 200
// one wave per SIMD would be ideal; we launch 1 wave per workgroup, 1 workgroup: cycles per instruction for 8 independent chains
template <int OP>
__global__ void k(float* out, unsigned long long* cyc) {
    float a[8], b = 1.0001f, c = 0.5f; pf2 p[8]; unsigned h[8];
    for (int i = 0; i < 8; ++i) { a[i] = threadIdx.x + i; p[i] = pf2{a[i], a[i] + 1}; h[i] = i; }
    const pf2 b2 = {b, b};
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int l = 0; l < LOOPS; ++l) {
#pragma unroll
        for (int r = 0; r < REP / 8; ++r)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (OP == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
                if (OP == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(p[i]) : "v"(b2));
                if (OP == 2) asm volatile("v_fma_mix_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
                if (OP == 3) asm volatile("v_fma_mixlo_f16 %0, %1, %2, 0" : "+v"(h[i]) : "v"(a[i]), "v"(b));
                if (OP == 4) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(h[i]) : "v"(a[i]), "v"(b));
                if (OP == 5) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
                if (OP == 6) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(b2));
                if (OP == 7) asm volatile("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(h[i]) : "v"(a[i]), "v"(b));
            }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0; for (int i = 0; i < 8; ++i) s += a[i] + p[i][0] + p[i][1] + h[i];
    out[threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[OP] = t1 - t0;
}
int main() {
    float* out; unsigned long long* cyc; hipMalloc(&out, 1024); hipMalloc(&cyc, 64);
    hipLaunchKernelGGL(k<0>, dim3(1), dim3(64), 0, 0, out, cyc); hipLaunchKernelGGL(k<1>, dim3(1), dim3(64), 0, 0, out, cyc);
    hipLaunchKernelGGL(k<2>, dim3(1), dim3(64), 0, 0, out, cyc); hipLaunchKernelGGL(k<3>, dim3(1), dim3(64), 0, 0, out, cyc);
    hipLaunchKernelGGL(k<4>, dim3(1), dim3(64), 0, 0, out, cyc); hipLaunchKernelGGL(k<5>, dim3(1), dim3(64), 0, 0, out, cyc);
    hipLaunchKernelGGL(k<6>, dim3(1), dim3(64), 0, 0, out, cyc); hipLaunchKernelGGL(k<7>, dim3(1), dim3(64), 0, 0, out, cyc);
    unsigned long long h[8]; hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
    const char* n[8] = {"v_fma_f32", "v_pk_fma_f32", "v_fma_mix_f32", "v_fma_mixlo_f16", "v_cvt_pk_f16_f32", "v_max_f32", "v_pk_mul_f32", "v_fma_mixhi_f16"};
    for (int i = 0; i < 8; ++i) printf("%-18s %.2f cycles per instruction (one wave)\n", n[i], (double)h[i] / (REP * LOOPS));
    return 0;
}
