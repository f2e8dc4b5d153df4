// Fused Adam step over the flat parameter / gradient buffers.
// Semantics of torch.optim.Adam(lr, betas, eps, weight_decay=l2) as the reference uses it
// (run_train_erc.py:512): L2 folded into the gradient (NOT AdamW), bias-corrected moments,
//   g' = g + wd * p;  m = b1 m + (1-b1) g';  v = b2 v + (1-b2) g'^2;
//   p -= lr / (1 - b1^t) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)
// One launch for all live parameters (the ~50 per-tensor launches of the unfused optimizer collapse to 1).
#include "mmdfn_internal.h"
#include "../../include/mmdfn_hip.h"

namespace {

__global__ void adam_step_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                 float* __restrict__ v, int64_t n, float lr, float beta1, float beta2, float eps,
                                 float wd, float bc1, float bc2_sqrt) {
    const int64_t n4 = n / 4;
    const float step_size = lr / bc1;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        float4 pv = reinterpret_cast<float4*>(p)[i];
        const float4 gv = reinterpret_cast<const float4*>(g)[i];
        float4 mv = reinterpret_cast<float4*>(m)[i];
        float4 vv = reinterpret_cast<float4*>(v)[i];
#define ADAM1(F)                                                       \
    {                                                                  \
        const float gg = gv.F + wd * pv.F;                             \
        mv.F = beta1 * mv.F + (1.0f - beta1) * gg;                     \
        vv.F = beta2 * vv.F + (1.0f - beta2) * gg * gg;                \
        pv.F -= step_size * mv.F / (sqrtf(vv.F) / bc2_sqrt + eps);     \
    }
        ADAM1(x) ADAM1(y) ADAM1(z) ADAM1(w)
#undef ADAM1
        reinterpret_cast<float4*>(p)[i] = pv;
        reinterpret_cast<float4*>(m)[i] = mv;
        reinterpret_cast<float4*>(v)[i] = vv;
    }
    // tail (n not a multiple of 4)
    const int64_t t = n4 * 4 + blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (t < n) {
        const float gg = g[t] + wd * p[t];
        const float mm = beta1 * m[t] + (1.0f - beta1) * gg;
        const float vv = beta2 * v[t] + (1.0f - beta2) * gg * gg;
        m[t] = mm;
        v[t] = vv;
        p[t] -= step_size * mm / (sqrtf(vv) / bc2_sqrt + eps);
    }
}

}  // namespace

extern "C" int mmdfn_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1,
                               float beta2, float eps, float weight_decay, int step, void* stream) {
    if (n <= 0 || step < 1) return -1;
    const float bc1 = 1.0f - powf(beta1, (float)step);
    const float bc2_sqrt = sqrtf(1.0f - powf(beta2, (float)step));
    int64_t blocks = (n / 4 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(adam_step_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, lr,
                       beta1, beta2, eps, weight_decay, bc1, bc2_sqrt);
    MMDFN_CHECK_LAUNCH();
    return 0;
}
