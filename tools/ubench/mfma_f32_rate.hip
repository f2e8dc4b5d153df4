// Micro-benchmark: issue rate of v_mfma_f32_16x16x4_f32 with the operand pattern of the propagate kernel
// (A from registers, B from LDS via ds_read_b32 fetched one k-step ahead), vs waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>  // 0: registers only, 1: B from LDS (prefetched), 2: + barrier every 4 k-steps
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    __shared__ float lds[32 * 116];
    for (int i = threadIdx.x; i < 32 * 116; i += 256) lds[i] = 0.001f * i;
    __syncthreads();
    const int lane = threadIdx.x & 63, frow = lane & 15, g = lane >> 4;
    f32x4 acc[7];
    for (int c = 0; c < 7; ++c) acc[c] = (f32x4){0, 0, 0, 0};
    float a = 1.0f + lane, b[2][7];
    for (int c = 0; c < 7; ++c) b[0][c] = b[1][c] = 0.5f + c;
    const float* hb = &lds[frow];
    for (int it = 0; it < iters; ++it) {
        if (MODE >= 1) { for (int c = 0; c < 7; ++c) b[0][c] = hb[(4 * g) * 116 + 16 * c]; }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (MODE >= 1 && j + 1 < 4) { for (int c = 0; c < 7; ++c) b[(j + 1) & 1][c] = hb[(4 * g + j + 1) * 116 + 16 * c]; }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int c = 0; c < 7; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b[j & 1][c], acc[c], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (MODE == 2) __syncthreads();
    }
    float s = 0;
    for (int c = 0; c < 7; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE>
void run(const char* name, int wg_per_cu) {
    int iters = 4000;
    int grid = 256 * wg_per_cu;
    float* out;
    hipMalloc(&out, grid * 256 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, out, 10);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double mfma = (double)grid * 4 * iters * 28;
    double tf = mfma * 2048 / (ms * 1e-3) / 1e12;
    double cyc = ms * 1e-3 * 2.4e9 / ((double)iters * 28 * wg_per_cu);
    printf("%-28s waves/SIMD %d : %7.1f TFLOP/s  (%.1f cycles@2.4GHz per MFMA per SIMD)\n", name, wg_per_cu, tf, cyc);
    hipFree(out);
}

int main() {
    for (int w : {1, 2, 4, 6}) run<0>("regs only", w);
    for (int w : {1, 2, 4, 6}) run<1>("B from LDS, prefetched", w);
    for (int w : {1, 2, 4, 6}) run<2>("B from LDS + barrier/16k", w);
    return 0;
}
