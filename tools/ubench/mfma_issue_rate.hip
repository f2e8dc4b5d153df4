// Micro-benchmark: what sets the issue rate of v_mfma_f32_32x32x16_bf16 from ONE wave per SIMD?
// (The producer / consumer K6 kernel measured 44 cycles per MFMA for a lone MFMA-only wave and 32 for two waves per SIMD,
// profiles/r04_k6_producer_consumer.md.)  Variants of a 48-MFMA "chunk" (4 accumulators round-robin, like the kernels):
//   0  constant A / B registers                       (the round-1 filler benchmark's case: 32.8 cycles)
//   1  A reused by 4 MFMAs, a DIFFERENT B register set per MFMA (12 B fragments preloaded, no LDS traffic)
//   2  as 1 + the 30 ds_read_b128 of a chunk interleaved (fragments really come from LDS)
//   3  as 2 with TWO accumulators (the 16-wave kernel's consumer: 32 rows x 64 columns)
// run with 1 and 2 waves per SIMD.   hipcc --offload-arch=gfx950 -O3 mfma_issue_rate.hip -o mfma_issue_rate && ./mfma_issue_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ f32x16 mm(u32x4 a, u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

template <int V>
__global__ __launch_bounds__(256) void k(float* out, long long* cyc, int iters) {
    __shared__ __attribute__((aligned(16))) unsigned lds[16 * 1024];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 16 * 1024; i += 256) lds[i] = 0x3f803f80u + i;
    __syncthreads();
    constexpr int NA = (V == 3) ? 2 : 4;
    f32x16 acc[4];
    for (int c = 0; c < 4; ++c)
        for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
    u32x4 fa[3], fb[3][4];
    for (int x = 0; x < 3; ++x) {
        fa[x] = (u32x4){0x3f803f80u + lane, 0x3f803f80u, 0x3f803f80u + x, 0x3f803f80u};
        for (int c = 0; c < 4; ++c) fb[x][c] = (u32x4){0x3f003f00u + lane + c, 0x3f003f00u + x, 0x3f003f00u, 0x3f003f00u};
    }
    const unsigned* base = lds + lane * 4;
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int kh = 0; kh < 2; ++kh) {
            if (V >= 2) {
#pragma unroll
                for (int x = 0; x < 3; ++x) {
                    fa[x] = *reinterpret_cast<const u32x4*>(base + 256 * (x + 3 * kh));
#pragma unroll
                    for (int c = 0; c < NA; ++c) fb[x][c] = *reinterpret_cast<const u32x4*>(base + 256 * (6 + 4 * x + c + 12 * kh));
                }
            }
            // a3b1 a2b1 a1b1 | a2b2 a1b2 | a1b3
            const int xa[6] = {2, 1, 0, 1, 0, 0}, xb[6] = {0, 0, 0, 1, 1, 2};
#pragma unroll
            for (int p = 0; p < 6; ++p)
#pragma unroll
                for (int c = 0; c < NA; ++c) {
                    if (V == 0) acc[c] = mm(fa[0], fb[0][0], acc[c]);
                    else acc[c] = mm(fa[xa[p]], fb[xb[p]][c], acc[c]);
                }
        }
    }
    long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int c = 0; c < 4; ++c) s += acc[c][0] + acc[c][7];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int V>
void run(int wg_per_cu) {
    int iters = 2000;
    int grid = 256 * wg_per_cu;
    float* out; long long* cyc;
    hipMalloc(&out, grid * 256 * 4); hipMalloc(&cyc, 8);
    hipLaunchKernelGGL((k<V>), dim3(grid), dim3(256), 0, 0, out, cyc, 10);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<V>), dim3(grid), dim3(256), 0, 0, out, cyc, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    const int per_chunk = (V == 3) ? 24 : 48;
    printf("variant %d, %d wave(s) per SIMD: %6.1f wave-cycles per MFMA, %6.2f ns wall per MFMA per SIMD (%d MFMAs per chunk)\n", V,
           wg_per_cu, (double)c / (iters * (double)per_chunk), ms * 1e6 / (iters * (double)per_chunk * wg_per_cu), per_chunk);
    hipFree(out); hipFree(cyc);
}

int main() {
    run<0>(1); run<1>(1); run<2>(1); run<3>(1);
    run<0>(2); run<1>(2); run<2>(2); run<3>(2);
    return 0;
}
