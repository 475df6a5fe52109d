// Which lanes of a wavefront does a ds_read_b128 serve in the same LDS cycle?  Every lane reads one broadcast address except two lanes
// that read different addresses of one bank: they cost a conflict cycle exactly when they are served together.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/lds_lane_groups_probe.hip -o /tmp/lanegroups
//   rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS --kernel-trace --output-format csv -d out -o p -- /tmp/lanegroups
// Result on MI355X (dispatch order = li in {0, 5, 16, 40} x lj): lanes {0-3, 12-15, 20-23, 24-27}, {4-11, 16-19, 28-31}, {32-35, 44-47,
// 52-55, 56-59}, {36-43, 48-51, 60-63} - not 16 consecutive lanes.  pair_f16_kernel's detection tile is laid out for these groups.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define LOOP 2000
// every lane reads its own 16-byte chunk (lane * 4 floats: conflict-free linear) except lane j, which reads lane i's bank at another address
__global__ __launch_bounds__(64) void pairprobe(float* out, int li, int lj) {
    __shared__ __attribute__((aligned(16))) float s[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) s[i] = i;
    __syncthreads();
    const int lane = threadIdx.x;
    int off = 32;                           // everybody: one broadcast address (banks 32..35)
    if (lane == li) off = 0;                // banks 0..3
    if (lane == lj) off = 1024;             // banks 0..3 again, another address: a conflict iff li and lj are served in the same cycle
    f32x4 acc = {0, 0, 0, 0};
    for (int it = 0; it < LOOP; ++it) {
        asm volatile("" ::: "memory");
        acc += *reinterpret_cast<const f32x4*>(s + off + (it & 1) * 2048);
    }
    out[blockIdx.x * 64 + lane] = acc[0] + acc[1] + acc[2] + acc[3];
}
int main() {
    float* out; hipMalloc(&out, 64 * 64 * 4);
    hipLaunchKernelGGL(pairprobe, dim3(8), dim3(64), 0, 0, out, 0, 0);  // baseline: linear
    for (int li : {0, 5, 16, 40}) for (int lj = 0; lj < 64; ++lj) if (lj != li) hipLaunchKernelGGL(pairprobe, dim3(8), dim3(64), 0, 0, out, li, lj);
    hipDeviceSynchronize(); printf("done\n"); return 0;
}
