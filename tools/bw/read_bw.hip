// Streaming-read microbenchmark: what HBM read rate can a plain kernel reach on this box?  (tools/bw/run.sh)
// Each thread keeps U 16-byte loads in flight; grid-stride over the buffer; the sum is written once per workgroup so the loads
// cannot be dropped.  Variants: U = 4 / 8 / 16, plain / nontemporal loads, workgroups per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
template <int U, bool NT>
__global__ __launch_bounds__(256) void read_kernel(const float4* __restrict__ p, size_t n4, float* out) {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + (U - 1) * stride < n4; i += U * stride) {
        float4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const float4* q = p + i + u * stride;
            if (NT) {
                v[u].x = __builtin_nontemporal_load(&q->x); v[u].y = __builtin_nontemporal_load(&q->y);
                v[u].z = __builtin_nontemporal_load(&q->z); v[u].w = __builtin_nontemporal_load(&q->w);
            } else {
                v[u] = *q;
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
    }
    for (; i < n4; i += stride) { const float4 v = p[i]; acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
    if (acc.x + acc.y + acc.z + acc.w == 1.2345f) out[blockIdx.x] = 1.f;
}
template <int U, bool NT>
void run(const char* name, std::vector<float4*>& bufs, size_t n4, float* out, int wg_per_cu) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int grid = 256 * wg_per_cu;
    for (int k = 0; k < 3; ++k) hipLaunchKernelGGL((read_kernel<U, NT>), dim3(grid), dim3(256), 0, 0, bufs[k % bufs.size()], n4, out);
    hipDeviceSynchronize();
    const int iters = 12;
    hipEventRecord(e0);
    for (int k = 0; k < iters; ++k) hipLaunchKernelGGL((read_kernel<U, NT>), dim3(grid), dim3(256), 0, 0, bufs[k % bufs.size()], n4, out);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-28s %2d WG/CU: %.2f TB/s\n", name, wg_per_cu, (double)n4 * 16 * iters / (ms * 1e-3) / 1e12);
}
int main(int argc, char** argv) {
    const size_t mb = argc > 1 ? atoi(argv[1]) : 800;
    const size_t n4 = mb * 1024 * 1024 / 16;
    std::vector<float4*> bufs(3);
    for (auto& b : bufs) { hipMalloc(&b, n4 * 16); hipMemset(b, 0, n4 * 16); }
    float* out; hipMalloc(&out, 1 << 20);
    printf("%zu MB per launch, 3 rotating buffers\n", mb);
    for (int wg : {2, 4, 8, 16}) {
        run<4, false>("4 x 16 B in flight", bufs, n4, out, wg);
        run<8, false>("8 x 16 B in flight", bufs, n4, out, wg);
        run<16, false>("16 x 16 B in flight", bufs, n4, out, wg);
        run<8, true>("8 x 16 B, nontemporal", bufs, n4, out, wg);
    }
    return 0;
}
