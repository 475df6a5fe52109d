// LDS access patterns of pair_f16_kernel, one kernel each, for rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE:
//   hipcc --offload-arch=gfx950 -O3 tools/probes/lds_pattern_probe.hip -o /tmp/ldsprobe
//   rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS --kernel-trace --output-format csv -d out -o p -- /tmp/ldsprobe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define LOOP 2000
// each kernel: 64-lane waves, 8 waves, LDS array, accumulate to defeat DCE
template <int STRIDE>
__global__ __launch_bounds__(512) void uc_read(float* out) {  // lane (p = lane & 15, kb = lane >> 4): row p, 8 kb + 32 s
    __shared__ __attribute__((aligned(16))) float s[64 * STRIDE];
    for (int i = threadIdx.x; i < 64 * STRIDE; i += 512) s[i] = i;
    __syncthreads();
    const int lane = threadIdx.x & 63, p = lane & 15, kb = lane >> 4;
    f32x4 acc = {0, 0, 0, 0};
    for (int it = 0; it < LOOP; ++it) {
        asm volatile("" ::: "memory");  // re-read every iteration
        const int sub = it & 3;
#pragma unroll
        for (int st = 0; st < 4; ++st) {
            acc += *reinterpret_cast<const f32x4*>(s + (16 * sub + p) * STRIDE + 8 * kb + 32 * st);
            acc += *reinterpret_cast<const f32x4*>(s + (16 * sub + p) * STRIDE + 8 * kb + 32 * st + 4);
        }
    }
    out[blockIdx.x * 512 + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
}
template <int TS>
__global__ __launch_bounds__(512) void tile_write(float* out) {  // lane (p, kb) writes row 16 sub + p, cols 4 kb (+16, +32)
    __shared__ __attribute__((aligned(16))) float s[8 * 64 * TS];
    const int lane = threadIdx.x & 63, p = lane & 15, kb = lane >> 4, wid = threadIdx.x >> 6;
    float* my = s + wid * 64 * TS;
    f32x4 v = {1, 2, 3, 4};
    for (int it = 0; it < LOOP; ++it) {
        const int sub = it & 3;
        float* row = my + (16 * sub + p) * TS + 4 * kb;
        *reinterpret_cast<f32x4*>(row) = v;
        *reinterpret_cast<f32x4*>(row + 16) = v;
        if (kb < 2) *reinterpret_cast<f32x4*>(row + 32) = v;
        v[0] += 1.0f;
    }
    __syncthreads();
    out[blockIdx.x * 512 + threadIdx.x] = s[threadIdx.x];
}
template <int TS>
__global__ __launch_bounds__(512) void tile_read(float* out) {  // lane = pair: row lane, 10 float4
    __shared__ __attribute__((aligned(16))) float s[8 * 64 * TS];
    for (int i = threadIdx.x; i < 8 * 64 * TS; i += 512) s[i] = i;
    __syncthreads();
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const float* mine = s + wid * 64 * TS + lane * TS;
    f32x4 acc = {0, 0, 0, 0};
    for (int it = 0; it < LOOP; ++it) {
        asm volatile("" ::: "memory");
#pragma unroll
        for (int g = 0; g < 10; ++g) acc += *reinterpret_cast<const f32x4*>(mine + 4 * g) * (float)it;
    }
    out[blockIdx.x * 512 + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
}
__global__ __launch_bounds__(512) void up_broadcast(float* out) {  // the 16 lanes of a k block read one address: 8 kb + 32 s
    __shared__ __attribute__((aligned(16))) float s[8 * 768];
    for (int i = threadIdx.x; i < 8 * 768; i += 512) s[i] = i;
    __syncthreads();
    const int lane = threadIdx.x & 63, kb = lane >> 4, wid = threadIdx.x >> 6;
    const float* up = s + wid * 768;
    f32x4 acc = {0, 0, 0, 0};
    for (int it = 0; it < LOOP; ++it) {
        const float* u = up + (it % 3) * 256;
#pragma unroll
        for (int st = 0; st < 4; ++st) {
            acc += *reinterpret_cast<const f32x4*>(u + 32 * st + 8 * kb) * (float)it;
            acc += *reinterpret_cast<const f32x4*>(u + 32 * st + 8 * kb + 4);
        }
    }
    out[blockIdx.x * 512 + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
}
__global__ __launch_bounds__(512) void a4_read(float* out) {  // 4x4x1 operand rows: lane & 3 picks one of four consecutive float4
    __shared__ __attribute__((aligned(16))) float s[2200];
    for (int i = threadIdx.x; i < 2200; i += 512) s[i] = i;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const float* arow = s + (lane & 3) * 4;
    f32x4 acc = {0, 0, 0, 0};
    for (int it = 0; it < LOOP; ++it)
#pragma unroll
        for (int g = 0; g < 16; ++g) acc += *reinterpret_cast<const f32x4*>(arow + ((it & 7) * 16 + g) * 16) * (float)it;
    out[blockIdx.x * 512 + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
}
int main() {
    float* out;
    hipMalloc(&out, 256 * 512 * 4);
    hipLaunchKernelGGL(uc_read<132>, dim3(256), dim3(512), 0, 0, out);
    hipLaunchKernelGGL(uc_read<136>, dim3(256), dim3(512), 0, 0, out);
    hipLaunchKernelGGL(uc_read<140>, dim3(256), dim3(512), 0, 0, out);
    hipLaunchKernelGGL(tile_write<44>, dim3(256), dim3(512), 0, 0, out);
    hipLaunchKernelGGL(tile_write<40>, dim3(256), dim3(512), 0, 0, out);
    hipLaunchKernelGGL(tile_write<36>, dim3(256), dim3(512), 0, 0, out);
    hipLaunchKernelGGL(tile_write<52>, dim3(256), dim3(512), 0, 0, out);
    hipLaunchKernelGGL(tile_read<44>, dim3(256), dim3(512), 0, 0, out);
    hipLaunchKernelGGL(tile_read<40>, dim3(256), dim3(512), 0, 0, out);
    hipLaunchKernelGGL(tile_read<36>, dim3(256), dim3(512), 0, 0, out);
    hipLaunchKernelGGL(tile_read<52>, dim3(256), dim3(512), 0, 0, out);
    hipLaunchKernelGGL(up_broadcast, dim3(256), dim3(512), 0, 0, out);
    hipLaunchKernelGGL(a4_read, dim3(256), dim3(512), 0, 0, out);
    hipDeviceSynchronize();
    printf("done\n");
    return 0;
}
