// Does v_mfma_f32_16x16x32_f16 / v_mfma_f32_32x32x16_f16 take fp16 SUBNORMAL inputs at their value (or flush them to zero)?
// And does v_cvt_pk_f16_f32 produce them?  hipcc --offload-arch=gfx950 -O2 mfma_denorm_probe.hip -o _bin/mfma_denorm_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
__global__ void probe(const float* x, float* out) {
    const int lane = threadIdx.x;
    h8 a, b;
    for (int j = 0; j < 8; ++j) {
        a[j] = (_Float16)x[j];        // converted on the device: v_cvt_f16_f32 (subnormal results for |x| < 2^-14)
        b[j] = (_Float16)1.0f;
    }
    f4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    f16v d = {0};
    d = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, d, 0, 0, 0);
    if (lane == 0) {
        out[0] = c[0];
        out[1] = d[0];
        for (int j = 0; j < 8; ++j) out[2 + j] = (float)a[j];
    }
}
int main() {
    float hx[8] = {0x1p-15f, 0x1p-16f, 0x1p-18f, 0x1p-20f, 0x1p-22f, 0x1p-24f, 0x1.8p-17f, 0x1p-14f};
    float *dx, *dout, ho[16];
    hipMalloc(&dx, sizeof hx);
    hipMalloc(&dout, sizeof ho);
    hipMemcpy(dx, hx, sizeof hx, hipMemcpyHostToDevice);
    probe<<<1, 64>>>(dx, dout);
    hipMemcpy(ho, dout, sizeof ho, hipMemcpyDeviceToHost);
    double want = 0;
    for (int j = 0; j < 8; ++j) want += hx[j];
    // 16x16x32: lane 0 holds k = 0..7 of row 0; the other k blocks (lanes 16, 32, 48 of A) hold the same values -> 4 x the sum
    printf("sum of the 8 inputs %.9g (x4 = %.9g)\n16x16x32 result %.9g\n32x32x16 result %.9g (2 k blocks: x2 = %.9g)\n", want, 4 * want, ho[0], ho[1], 2 * want);
    for (int j = 0; j < 8; ++j) printf("x %.9g -> f16 -> %.9g\n", hx[j], ho[2 + j]);
    return 0;
}
