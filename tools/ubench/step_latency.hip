// Micro-benchmark: cost of one "recurrence step" skeleton on gfx950 -- N FMAs per lane (6 independent chains),
// LDS broadcast reads, an optional shuffle, transcendental gate math and one workgroup barrier.
#include <hip/hip_runtime.h>
#include <cstdio>

template <int NFMA, int MODE>
__global__ __launch_bounds__(256) void k(float* out, int steps, long long* clk) {
    __shared__ __attribute__((aligned(16))) float hs[2][104];
    float w[NFMA];
    for (int i = 0; i < NFMA; ++i) w[i] = 1e-3f * (threadIdx.x + i);
    if (threadIdx.x < 208) (&hs[0][0])[threadIdx.x] = 0.01f * threadIdx.x;
    __syncthreads();
    long long c0 = clock64(), w0 = wall_clock64();
    float hp = 0.f;
    for (int s = 0; s < steps; ++s) {
        const int cur = s & 1;
        float a[6] = {0, 0, 0, 0, 0, 0};
        const float4* hv = reinterpret_cast<const float4*>(&hs[cur][(threadIdx.x & 1) * 52]);
#pragma unroll
        for (int k4 = 0; k4 < NFMA / 12; ++k4) {
            const float4 h4 = hv[k4 % 13];
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                a[j] = fmaf(w[12 * k4 + j], h4.x, a[j]);
                a[3 + j] = fmaf(w[12 * k4 + 3 + j], h4.y, a[3 + j]);
                a[j] = fmaf(w[12 * k4 + 6 + j], h4.z, a[j]);
                a[3 + j] = fmaf(w[12 * k4 + 9 + j], h4.w, a[3 + j]);
            }
        }
        float ar = a[0] + a[3], az = a[1] + a[4], an = a[2] + a[5];
        if (MODE >= 1) { ar += __shfl_xor(ar, 1, 64); az += __shfl_xor(az, 1, 64); an += __shfl_xor(an, 1, 64); }
        if (MODE >= 2 && (threadIdx.x & 1) == 0 && threadIdx.x < 200) {
            const float rr = 1.0f / (1.0f + expf(-ar)), zz = 1.0f / (1.0f + expf(-az));
            const float nn = tanhf(an * rr);
            hp = (1.0f - zz) * nn + zz * hp;
            hs[cur ^ 1][threadIdx.x >> 1] = hp;
        } else if (MODE < 2 && threadIdx.x < 100) {
            hp = ar + az + an;
            hs[cur ^ 1][threadIdx.x] = hp * 1e-3f;
        }
        __syncthreads();
    }
    long long c1 = clock64(), w1 = wall_clock64();
    out[blockIdx.x * 256 + threadIdx.x] = hp;
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = c1 - c0; clk[1] = w1 - w0; }
}

typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float pair_swap(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true));
}
// MODE2: 0 = pk_fma + dpp + fast gates; 1 = + 3 LDS operand reads and 5 LDS result writes per gate lane
template <int MODE2>
__global__ __launch_bounds__(256) void k2(float* out, int steps, long long* clk) {
    __shared__ __attribute__((aligned(16))) float hs[2][104];
    __shared__ float in_s[8][300];
    __shared__ float out_s[8][500];
    f32x2 wr[26], wz[26], wn[26];
    for (int i = 0; i < 26; ++i) { wr[i] = {1e-3f * (threadIdx.x + i), 2e-3f}; wz[i] = {3e-3f * i, 1e-3f}; wn[i] = {1e-3f, 2e-3f * i}; }
    if (threadIdx.x < 208) (&hs[0][0])[threadIdx.x] = 0.01f * threadIdx.x;
    for (int i = threadIdx.x; i < 2400; i += 256) (&in_s[0][0])[i] = 0.001f * i;
    __syncthreads();
    long long c0 = clock64();
    float hp = 0.f;
    const int u = threadIdx.x >> 1;
    for (int s = 0; s < steps; ++s) {
        const int cur = s & 1;
        f32x2 ar0 = {0, 0}, az0 = {0, 0}, an0 = {0, 0}, ar1 = {0, 0}, az1 = {0, 0}, an1 = {0, 0};
        const float4* hv = reinterpret_cast<const float4*>(&hs[cur][(threadIdx.x & 1) * 52]);
#pragma unroll
        for (int k4 = 0; k4 < 13; ++k4) {
            const float4 h4 = hv[k4];
            const f32x2 ha = {h4.x, h4.y}, hb = {h4.z, h4.w};
            ar0 = __builtin_elementwise_fma(wr[2 * k4], ha, ar0); az0 = __builtin_elementwise_fma(wz[2 * k4], ha, az0); an0 = __builtin_elementwise_fma(wn[2 * k4], ha, an0);
            ar1 = __builtin_elementwise_fma(wr[2 * k4 + 1], hb, ar1); az1 = __builtin_elementwise_fma(wz[2 * k4 + 1], hb, az1); an1 = __builtin_elementwise_fma(wn[2 * k4 + 1], hb, an1);
        }
        float ar = (ar0.x + ar0.y) + (ar1.x + ar1.y), az = (az0.x + az0.y) + (az1.x + az1.y), an = (an0.x + an0.y) + (an1.x + an1.y);
        ar += pair_swap(ar); az += pair_swap(az); an += pair_swap(an);
        if ((threadIdx.x & 1) == 0 && u < 100) {
            float g0 = 0.f, g1 = 0.f, g2 = 0.f;
            if (MODE2 >= 1) { const float* gp = &in_s[s & 7][u]; g0 = gp[0]; g1 = gp[100]; g2 = gp[200]; }
            const float rr = __builtin_amdgcn_rcpf(1.0f + __expf(-(ar + g0))), zz = __builtin_amdgcn_rcpf(1.0f + __expf(-(az + g1)));
            const float nn = 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(2.0f * (g2 + an * rr)));
            hp = (1.0f - zz) * nn + zz * hp;
            hs[cur ^ 1][u] = hp;
            if (MODE2 >= 1) { float* op = &out_s[s & 7][u]; op[0] = hp; op[100] = rr; op[200] = zz; op[300] = nn; op[400] = an; }
        }
        __syncthreads();
    }
    long long c1 = clock64();
    out[blockIdx.x * 256 + threadIdx.x] = hp + out_s[1][threadIdx.x];
    if (blockIdx.x == 0 && threadIdx.x == 0) clk[0] = c1 - c0;
}
// 4 lanes per hidden unit (K split in four 25-wide quarters), 448 threads: half the FMAs / LDS reads per lane,
// two DPP reduction steps, a 7-wave barrier
__device__ __forceinline__ float quad_swap2(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, true));  // quad_perm [2,3,0,1]
}
__global__ __launch_bounds__(448) void k4(float* out, int steps, long long* clk) {
    __shared__ __attribute__((aligned(16))) float hs[2][112];
    __shared__ float in_s[8][300];
    __shared__ float out_s[8][500];
    f32x2 wr[14], wz[14], wn[14];
    for (int i = 0; i < 14; ++i) { wr[i] = {1e-3f * (threadIdx.x + i), 2e-3f}; wz[i] = {3e-3f * i, 1e-3f}; wn[i] = {1e-3f, 2e-3f * i}; }
    if (threadIdx.x < 224) (&hs[0][0])[threadIdx.x] = 0.01f * threadIdx.x;
    for (int i = threadIdx.x; i < 2400; i += 448) (&in_s[0][0])[i] = 0.001f * i;
    __syncthreads();
    long long c0 = clock64();
    float hp = 0.f;
    const int u = threadIdx.x >> 2;
    const int q = threadIdx.x & 3;
    for (int s = 0; s < steps; ++s) {
        const int cur = s & 1;
        f32x2 ar0 = {0, 0}, az0 = {0, 0}, an0 = {0, 0}, ar1 = {0, 0}, az1 = {0, 0}, an1 = {0, 0};
        const float4* hv = reinterpret_cast<const float4*>(&hs[cur][q * 28]);
#pragma unroll
        for (int k4 = 0; k4 < 7; ++k4) {
            const float4 h4 = hv[k4];
            const f32x2 ha = {h4.x, h4.y}, hb = {h4.z, h4.w};
            ar0 = __builtin_elementwise_fma(wr[2 * k4], ha, ar0); az0 = __builtin_elementwise_fma(wz[2 * k4], ha, az0); an0 = __builtin_elementwise_fma(wn[2 * k4], ha, an0);
            ar1 = __builtin_elementwise_fma(wr[2 * k4 + 1], hb, ar1); az1 = __builtin_elementwise_fma(wz[2 * k4 + 1], hb, az1); an1 = __builtin_elementwise_fma(wn[2 * k4 + 1], hb, an1);
        }
        float ar = (ar0.x + ar0.y) + (ar1.x + ar1.y), az = (az0.x + az0.y) + (az1.x + az1.y), an = (an0.x + an0.y) + (an1.x + an1.y);
        ar += pair_swap(ar); az += pair_swap(az); an += pair_swap(an);
        ar += quad_swap2(ar); az += quad_swap2(az); an += quad_swap2(an);
        if (q == 0 && u < 100) {
            const float* gp = &in_s[s & 7][u];
            const float g0 = gp[0], g1 = gp[100], g2 = gp[200];
            const float rr = __builtin_amdgcn_rcpf(1.0f + __expf(-(ar + g0))), zz = __builtin_amdgcn_rcpf(1.0f + __expf(-(az + g1)));
            const float nn = 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(2.0f * (g2 + an * rr)));
            hp = (1.0f - zz) * nn + zz * hp;
            hs[cur ^ 1][(u / 25) * 28 + (u % 25)] = hp;
            float* op = &out_s[s & 7][u]; op[0] = hp; op[100] = rr; op[200] = zz; op[300] = nn; op[400] = an;
        }
        __syncthreads();
    }
    long long c1 = clock64();
    out[blockIdx.x * 448 + threadIdx.x] = hp + out_s[1][threadIdx.x];
    if (blockIdx.x == 0 && threadIdx.x == 0) clk[0] = c1 - c0;
}
void run4(const char* name, int grid) {
    float* out; long long* clk; long long h[2];
    hipMalloc(&out, grid * 448 * 4); hipMalloc(&clk, 16);
    const int steps = 2000;
    hipLaunchKernelGGL(k4, dim3(grid), dim3(448), 0, 0, out, 10, clk);
    hipLaunchKernelGGL(k4, dim3(grid), dim3(448), 0, 0, out, steps, clk);
    hipDeviceSynchronize();
    hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    printf("%-44s grid %4d: %lld shader cycles/step\n", name, grid, h[0] / steps);
    hipFree(out); hipFree(clk);
}

template <int MODE2>
void run2(const char* name, int grid) {
    float* out; long long* clk; long long h[2];
    hipMalloc(&out, grid * 256 * 4); hipMalloc(&clk, 16);
    const int steps = 2000;
    hipLaunchKernelGGL((k2<MODE2>), dim3(grid), dim3(256), 0, 0, out, 10, clk);
    hipLaunchKernelGGL((k2<MODE2>), dim3(grid), dim3(256), 0, 0, out, steps, clk);
    hipDeviceSynchronize();
    hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    printf("%-44s grid %4d: %lld shader cycles/step\n", name, grid, h[0] / steps);
    hipFree(out); hipFree(clk);
}

template <int NFMA, int MODE>
void run(const char* name, int grid) {
    float* out; long long* clk; long long h[2];
    hipMalloc(&out, grid * 256 * 4); hipMalloc(&clk, 16);
    const int steps = 2000;
    hipLaunchKernelGGL((k<NFMA, MODE>), dim3(grid), dim3(256), 0, 0, out, 10, clk);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NFMA, MODE>), dim3(grid), dim3(256), 0, 0, out, steps, clk);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    printf("%-34s grid %4d: %.3f us/step, %lld shader cycles/step, wall ticks/step %.1f (100 MHz => %.2f GHz)\n", name, grid,
           ms * 1e3 / steps, h[0] / steps, (double)h[1] / steps, (double)h[0] / ((double)h[1] * 10.0));
    hipFree(out); hipFree(clk);
}

int main() {
    run2<0>("pk_fma + dpp + fast gates + barrier", 224);
    run2<1>("  + LDS operand reads / result writes", 224);
    run4("4 lanes per unit, 448 threads (all of it)", 224);
    run<156, 0>("156 fma + barrier", 224);
    run<156, 1>("156 fma + shfl + barrier", 224);
    run<156, 2>("156 fma + shfl + gates + barrier", 224);
    run<60, 2>("60 fma + shfl + gates + barrier", 224);
    run<156, 2>("156 fma + shfl + gates + barrier", 1);
    run<12, 0>("12 fma + barrier", 224);
    return 0;
}
