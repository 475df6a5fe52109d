// Issue cost of candidate instructions for the per-pair work of pair_f16.hip with ONE and with TWO waves per SIMD (one workgroup of
// 4 / 8 waves on one CU): cycles per instruction and wave, eight independent chains.  Answers (a) what the packed fp16 add / max cost
// that a fixed-grid piece representation would use, (b) whether two waves of a SIMD overlap their VALU instructions (time per
// instruction and wave unchanged from 4 to 8 waves) or share one issue stream (doubled).
//   hipcc --offload-arch=gfx950 -O3 tools/probes/valu_pair_probe.hip -o /tmp/valupair && /tmp/valupair
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float pf2 __attribute__((ext_vector_type(2)));
#define REP 64
#define LOOPS 200
#define NOPS 7
template <int OP>
__global__ void k(float* out, unsigned long long* cyc, int slot) {
    float a[8], b = 1.0001f, c = 0.5f; pf2 p[8]; unsigned h[8], g[8];
    for (int i = 0; i < 8; ++i) { a[i] = threadIdx.x + i; p[i] = pf2{a[i], a[i] + 1}; h[i] = 0x3c003c00u + i; g[i] = 0x38003800u; }
    const pf2 b2 = {b, b};
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int l = 0; l < LOOPS; ++l) {
#pragma unroll
        for (int r = 0; r < REP / 8; ++r)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (OP == 0) asm volatile("v_pk_add_f16 %0, %0, %1" : "+v"(h[i]) : "v"(g[i]));
                if (OP == 1) asm volatile("v_pk_max_f16 %0, %0, %1 neg_lo:[0,1] neg_hi:[0,1]" : "+v"(h[i]) : "v"(g[i]));
                if (OP == 2) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(h[i]) : "v"(a[i]), "v"(b));
                if (OP == 3) asm volatile("v_fma_mix_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
                if (OP == 4) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(p[i]) : "v"(b2));
                if (OP == 5) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
                if (OP == 6) asm volatile("v_pk_add_f16 %0, %0, %1 clamp" : "+v"(h[i]) : "v"(g[i]));
            }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0; for (int i = 0; i < 8; ++i) s += a[i] + p[i][0] + p[i][1] + h[i];
    out[threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[slot * NOPS + OP] = t1 - t0;
}
int main() {
    float* out; unsigned long long* cyc; (void)hipMalloc(&out, 8192); (void)hipMalloc(&cyc, 8 * 4 * NOPS);
    for (int w = 0; w < 4; ++w) {
        const int threads = 256 * (w + 1);
        hipLaunchKernelGGL(k<0>, dim3(1), dim3(threads), 0, 0, out, cyc, w); hipLaunchKernelGGL(k<1>, dim3(1), dim3(threads), 0, 0, out, cyc, w);
        hipLaunchKernelGGL(k<2>, dim3(1), dim3(threads), 0, 0, out, cyc, w); hipLaunchKernelGGL(k<3>, dim3(1), dim3(threads), 0, 0, out, cyc, w);
        hipLaunchKernelGGL(k<4>, dim3(1), dim3(threads), 0, 0, out, cyc, w); hipLaunchKernelGGL(k<5>, dim3(1), dim3(threads), 0, 0, out, cyc, w);
        hipLaunchKernelGGL(k<6>, dim3(1), dim3(threads), 0, 0, out, cyc, w);
    }
    unsigned long long h[4 * NOPS]; (void)hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    const char* n[NOPS] = {"v_pk_add_f16", "v_pk_max_f16 (neg)", "v_cvt_pk_f16_f32", "v_fma_mix_f32", "v_pk_fma_f32", "v_fma_f32", "v_pk_add_f16 clamp"};
    for (int i = 0; i < NOPS; ++i)
        printf("%-20s cycles per instruction and wave at 1 / 2 / 3 / 4 waves per SIMD: %.2f  %.2f  %.2f  %.2f\n", n[i], (double)h[i] / (REP * LOOPS),
               (double)h[NOPS + i] / (REP * LOOPS), (double)h[2 * NOPS + i] / (REP * LOOPS), (double)h[3 * NOPS + i] / (REP * LOOPS));
    return 0;
}
