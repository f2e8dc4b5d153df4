// K10: FocalLoss forward / backward as one launch each.
//
// Replaces the gather / exp / pow / mul / mean chain of FocalLoss.forward (reference loss.py:14-34) and its autograd:
//   logpt_i = log_prob[i, t_i];  pt_i = exp(logpt_i)  (treated as a constant: the reference detaches it, loss.py:26)
//   loss = reduce_i( -(1 - pt_i)^gamma * alpha[t_i] * logpt_i ),  reduce = mean (size_average) or sum
//   d loss / d log_prob[i, c] = (c == t_i) * ( -(1 - pt_i)^gamma * alpha[t_i] ) * (1/N or 1)
// Forward: ONE workgroup walks the rows with a fixed row -> thread assignment and reduces in a fixed tree, so the sum
// is bit-reproducible (no float atomics); it also stores the per-row coefficient the backward multiplies with.
#include "mmdfn_internal.h"
#include "../../include/mmdfn_hip.h"

namespace {

constexpr int FL_NT = 1024;

__global__ __launch_bounds__(FL_NT) void focal_loss_fwd_kernel(const float* __restrict__ logp, const int64_t* __restrict__ target,
                                                               const float* __restrict__ alpha, float* __restrict__ loss,
                                                               float* __restrict__ coef, int64_t N, int C, float gamma,
                                                               float scale, float* __restrict__ dunit) {
    __shared__ float part[FL_NT / 64];
    float acc = 0.f;
    for (int64_t i = threadIdx.x; i < N; i += FL_NT) {
        int64_t t = target[i];
        // an out-of-range label raises in the reference (gather, loss.py:23).  A kernel cannot raise without a host
        // sync, so it poisons the loss instead: NaN loss and NaN gradients stop a run as surely, nothing is read out of bounds
        const bool bad = t < 0 || t >= C;
        t = bad ? 0 : t;
        const float lp = bad ? __builtin_nanf("") : logp[i * C + t];
        const float pt = expf(lp);
        float wgt = (gamma == 0.f) ? 1.f : powf(fmaxf(1.f - pt, 0.f), gamma);
        if (alpha != nullptr) wgt *= alpha[t];
        if (bad) wgt = __builtin_nanf("");               // (fmaxf / powf above swallow the NaN of lp)
        coef[i] = -wgt * scale;
        acc += -wgt * lp;
        // d loss / d log_prob for an upstream gradient of exactly 1 (the usual case: loss.backward()): written here, so the
        // backward pass of the step needs no launch of its own for the loss (a bad label poisons its whole row, as the
        // backward kernel does)
        if (dunit != nullptr)
            for (int c = 0; c < C; ++c) dunit[i * C + c] = (c == t || bad) ? -wgt * scale : 0.f;
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < FL_NT / 64; ++w) s += part[w];
        loss[0] = s * scale;
    }
}

__global__ __launch_bounds__(256) void focal_loss_bwd_kernel(const float* __restrict__ coef, const int64_t* __restrict__ target,
                                                             const float* __restrict__ dloss, float* __restrict__ dlogp,
                                                             int64_t N, int C) {
    const int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (idx >= N * C) return;
    const int64_t i = idx / C;
    const int c = (int)(idx - i * C);
    const int64_t t = target[i];
    dlogp[idx] = (c == t || t < 0 || t >= C) ? coef[i] * dloss[0] : 0.f;   // bad label: coef is NaN, the whole row is poisoned
}

// The same with rows to leave out (target == ignore_index): they add nothing to the loss, get a zero gradient row, and
// the mean divides by the number of rows that count -- computed here, on the device, so that a captured step serves
// batches with different numbers of real utterances (train.StepGraphCache pads a batch to its bucket with a dummy
// dialogue whose labels are ignore_index).  scale_out[0] = 1 / count (or 1 without size_average); coef is NOT pre-scaled.
__global__ __launch_bounds__(FL_NT) void focal_loss_fwd_ignore_kernel(const float* __restrict__ logp, const int64_t* __restrict__ target,
                                                                      const float* __restrict__ alpha, float* __restrict__ loss,
                                                                      float* __restrict__ coef, float* __restrict__ scale_out,
                                                                      int64_t N, int C, float gamma, int size_average,
                                                                      int64_t ignore_index) {
    __shared__ float part[FL_NT / 64];
    __shared__ int cpart[FL_NT / 64];
    float acc = 0.f;
    int cnt = 0;
    for (int64_t i = threadIdx.x; i < N; i += FL_NT) {
        int64_t t = target[i];
        if (t == ignore_index) {
            coef[i] = 0.f;
            continue;
        }
        const bool bad = t < 0 || t >= C;
        t = bad ? 0 : t;
        const float lp = bad ? __builtin_nanf("") : logp[i * C + t];
        const float pt = expf(lp);
        float wgt = (gamma == 0.f) ? 1.f : powf(fmaxf(1.f - pt, 0.f), gamma);
        if (alpha != nullptr) wgt *= alpha[t];
        if (bad) wgt = __builtin_nanf("");
        coef[i] = -wgt;
        acc += -wgt * lp;
        ++cnt;
    }
    acc = wave_sum(acc);
    for (int off = 32; off > 0; off >>= 1) cnt += __shfl_xor(cnt, off);
    if ((threadIdx.x & 63) == 0) {
        part[threadIdx.x >> 6] = acc;
        cpart[threadIdx.x >> 6] = cnt;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float s = 0.f;
        int n = 0;
#pragma unroll
        for (int w = 0; w < FL_NT / 64; ++w) {
            s += part[w];
            n += cpart[w];
        }
        const float scale = size_average ? 1.0f / (float)(n > 0 ? n : 1) : 1.0f;
        loss[0] = s * scale;
        scale_out[0] = scale;
    }
}

__global__ __launch_bounds__(256) void focal_loss_bwd_ignore_kernel(const float* __restrict__ coef, const int64_t* __restrict__ target,
                                                                    const float* __restrict__ dloss, const float* __restrict__ scale,
                                                                    float* __restrict__ dlogp, int64_t N, int C,
                                                                    int64_t ignore_index) {
    const int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (idx >= N * C) return;
    const int64_t i = idx / C;
    const int c = (int)(idx - i * C);
    const int64_t t = target[i];
    float v = 0.f;
    if (t != ignore_index && (c == t || t < 0 || t >= C)) v = coef[i] * scale[0] * dloss[0];
    dlogp[idx] = v;
}

}  // namespace

extern "C" int mmdfn_focal_loss_fwd_ignore(const float* logp, const int64_t* target, const float* alpha, float* loss, float* coef,
                                           float* scale_out, int64_t N, int C, float gamma, int size_average,
                                           int64_t ignore_index, void* stream) {
    if (N <= 0 || C <= 0) return -1;
    hipLaunchKernelGGL(focal_loss_fwd_ignore_kernel, dim3(1), dim3(FL_NT), 0, (hipStream_t)stream, logp, target, alpha, loss, coef,
                       scale_out, N, C, gamma, size_average, ignore_index);
    MMDFN_CHECK_LAUNCH();
    return 0;
}

extern "C" int mmdfn_focal_loss_bwd_ignore(const float* coef, const int64_t* target, const float* dloss, const float* scale,
                                           float* dlogp, int64_t N, int C, int64_t ignore_index, void* stream) {
    if (N <= 0 || C <= 0) return -1;
    const int64_t total = N * C;
    hipLaunchKernelGGL(focal_loss_bwd_ignore_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       coef, target, dloss, scale, dlogp, N, C, ignore_index);
    MMDFN_CHECK_LAUNCH();
    return 0;
}

extern "C" int mmdfn_focal_loss_fwd_grad(const float* logp, const int64_t* target, const float* alpha, float* loss, float* coef,
                                         float* dlogp_unit, int64_t N, int C, float gamma, int size_average, void* stream) {
    if (N <= 0 || C <= 0) return -1;
    const float scale = size_average ? 1.0f / (float)N : 1.0f;
    hipLaunchKernelGGL(focal_loss_fwd_kernel, dim3(1), dim3(FL_NT), 0, (hipStream_t)stream, logp, target, alpha, loss, coef, N, C,
                       gamma, scale, dlogp_unit);
    MMDFN_CHECK_LAUNCH();
    return 0;
}

extern "C" int mmdfn_focal_loss_fwd(const float* logp, const int64_t* target, const float* alpha, float* loss, float* coef,
                                    int64_t N, int C, float gamma, int size_average, void* stream) {
    return mmdfn_focal_loss_fwd_grad(logp, target, alpha, loss, coef, nullptr, N, C, gamma, size_average, stream);
}

extern "C" int mmdfn_focal_loss_bwd(const float* coef, const int64_t* target, const float* dloss, float* dlogp, int64_t N,
                                    int C, void* stream) {
    if (N <= 0 || C <= 0) return -1;
    const int64_t total = N * C;
    hipLaunchKernelGGL(focal_loss_bwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, coef,
                       target, dloss, dlogp, N, C);
    MMDFN_CHECK_LAUNCH();
    return 0;
}
