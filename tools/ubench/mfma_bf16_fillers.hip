// Micro-benchmark: how many independent VALU instructions ride in the shadow of one bf16 MFMA on gfx950?
// Per iteration: NM MFMAs (4 accumulators round-robin), each followed by F fillers (v_and / v_sub / v_perm on
// registers the MFMAs do not touch), pinned with scheduling barriers.  Uses s_memtime for the wave's own
// cycle count (clock-independent) plus wall time.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int SHAPE, int F>  // SHAPE 0: 32x32x16, 1: 16x16x32
__global__ __launch_bounds__(256) void k(float* out, long long* cyc, int iters) {
    const int lane = threadIdx.x & 63;
    f32x16 acc[4];
    f32x4 acc4[4];
    for (int c = 0; c < 4; ++c) {
        for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
        acc4[c] = (f32x4){0, 0, 0, 0};
    }
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(1.0f + lane + e); b[e] = (__bf16)(0.5f + e); }
    float x0 = 1.0f + lane, x1 = 2.0f + lane;
    unsigned m;
    asm volatile("s_mov_b32 %0, 0xffff0000" : "=s"(m));
    unsigned pk = 0;
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            if (SHAPE == 0) acc[j & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[j & 3], 0, 0, 0);
            else acc4[j & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc4[j & 3], 0, 0, 0);
#pragma unroll
            for (int f = 0; f < F; ++f) {
                // the cutting chain: and, sub, (perm)
                if ((f % 5) == 0) x0 = x0 - __builtin_bit_cast(float, __builtin_bit_cast(unsigned, x0) & m);
                else if ((f % 5) == 1) x1 = x1 - __builtin_bit_cast(float, __builtin_bit_cast(unsigned, x1) & m);
                else if ((f % 5) == 2) pk += __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, x1), __builtin_bit_cast(unsigned, x0), 0x07060302u);
                else if ((f % 5) == 3) x0 = x0 * 1.0001f;
                else x1 = x1 + 0.5f;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    long long t1 = __builtin_readcyclecounter();
    float s = x0 + x1 + pk;
    for (int c = 0; c < 4; ++c) s += acc[c][0] + acc4[c][0];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int SHAPE, int F>
void run(int wg_per_cu) {
    int iters = 2000;
    int grid = 256 * wg_per_cu;
    float* out; long long* cyc;
    hipMalloc(&out, grid * 256 * 4); hipMalloc(&cyc, 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<SHAPE, F>), dim3(grid), dim3(256), 0, 0, out, cyc, 10);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<SHAPE, F>), dim3(grid), dim3(256), 0, 0, out, cyc, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    double per = (double)c / (iters * 16.0);
    double ns = ms * 1e6 / (iters * 16.0 * wg_per_cu);
    printf("%s fillers/MFMA %d waves/SIMD %d : %6.1f wave-cycles per MFMA slot, %6.2f ns wall per MFMA per SIMD\n",
           SHAPE == 0 ? "32x32x16" : "16x16x32", F, wg_per_cu, per, ns);
    hipFree(out); hipFree(cyc);
}

int main() {
    run<0, 0>(1); run<0, 2>(1); run<0, 4>(1); run<0, 5>(1); run<0, 6>(1); run<0, 8>(1); run<0, 12>(1);
    run<0, 0>(2); run<0, 4>(2); run<0, 6>(2); run<0, 8>(2); run<0, 12>(2);
    run<1, 0>(1); run<1, 2>(1); run<1, 4>(1); run<1, 6>(1);
    run<1, 0>(2); run<1, 2>(2); run<1, 4>(2); run<1, 6>(2);
    return 0;
}
