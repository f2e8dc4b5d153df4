// Pointwise / row-reduction kernels of the two secondary fusion modules; their dense projections run
// on the few-row grouped MFMA kernel (linear_small.hip), so neither module touches a library GEMM.
//
//   MFN (reference model_fusion.py:62-120), per timestep:
//     attended = softmax(att1(cStar), dim=1) * cStar                       (:96-97)   softmax_scale_{fwd,bwd}
//     cHat = tanh(u);  g1 = sigmoid(v1);  g2 = sigmoid(v2);  mem' = g1 mem + g2 cHat   (:98-102)   mfn_mem_{fwd,bwd}
//   MMGatedAttention 'general' (reference model.py:761-781), per modality pair (m, n):
//     z = sigmoid(w . [x_m | x_n | x_m * x_n] + b);  h = z tanh(p_m) + (1 - z) tanh(p_n)   gated_pair_{fwd,bwd}
//     and the row contraction of the gate weight gradient                                  rowscale_colsum
#include "mmdfn_internal.h"
#include "../../include/mmdfn_hip.h"

namespace {

__device__ __forceinline__ float sigm(float x) { return 1.0f / (1.0f + expf(-x)); }

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// one wave per row: att = softmax(z[r, :W]); out = att * c
__global__ __launch_bounds__(256) void softmax_scale_fwd_kernel(const float* __restrict__ z, const float* __restrict__ c,
                                                                float* __restrict__ att, float* __restrict__ out, int R, int W) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= R) return;
    const float* zr = z + (int64_t)row * W;
    float mx = -INFINITY;
    for (int j = lane; j < W; j += 64) mx = fmaxf(mx, zr[j]);
    mx = wave_max(mx);
    float sum = 0.f;
    for (int j = lane; j < W; j += 64) sum += expf(zr[j] - mx);
    sum = wave_sum(sum);
    const float inv = 1.0f / sum;
    for (int j = lane; j < W; j += 64) {
        const float a = expf(zr[j] - mx) * inv;
        att[(int64_t)row * W + j] = a;
        out[(int64_t)row * W + j] = a * c[(int64_t)row * W + j];
    }
}

// dc = dout * att;  dz = att * (dout * c - sum_j dout_j c_j att_j)
__global__ __launch_bounds__(256) void softmax_scale_bwd_kernel(const float* __restrict__ att, const float* __restrict__ c,
                                                                const float* __restrict__ dout, float* __restrict__ dz,
                                                                float* __restrict__ dc, int R, int W) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= R) return;
    const int64_t o = (int64_t)row * W;
    float dot = 0.f;
    for (int j = lane; j < W; j += 64) dot += dout[o + j] * c[o + j] * att[o + j];
    dot = wave_sum(dot);
    for (int j = lane; j < W; j += 64) {
        const float a = att[o + j], d = dout[o + j];
        dc[o + j] = d * a;
        dz[o + j] = a * (d * c[o + j] - dot);
    }
}

// cHat = tanh(u), g1 = sigmoid(v1), g2 = sigmoid(v2), mem' = g1 mem + g2 cHat; saved: (cHat, g1, g2) for the backward pass
__global__ void mfn_mem_fwd_kernel(const float* __restrict__ u, const float* __restrict__ v1, const float* __restrict__ v2,
                                   const float* __restrict__ mem, float* __restrict__ out, float* __restrict__ saved, int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float ch = tanhf(u[i]), g1 = sigm(v1[i]), g2 = sigm(v2[i]);
        out[i] = g1 * mem[i] + g2 * ch;
        saved[i] = ch;
        saved[n + i] = g1;
        saved[2 * n + i] = g2;
    }
}

__global__ void mfn_mem_bwd_kernel(const float* __restrict__ saved, const float* __restrict__ mem, const float* __restrict__ dout,
                                   float* __restrict__ du, float* __restrict__ dv1, float* __restrict__ dv2,
                                   float* __restrict__ dmem, int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float ch = saved[i], g1 = saved[n + i], g2 = saved[2 * n + i], d = dout[i];
        du[i] = d * g2 * (1.0f - ch * ch);
        dv1[i] = d * mem[i] * g1 * (1.0f - g1);
        dv2[i] = d * ch * g2 * (1.0f - g2);
        dmem[i] = d * g1;
    }
}

// one wave per row.  z = sigmoid(w[0:D].xm + w[D:2D].xn + w[2D:3D].(xm*xn) + b);  out = z tanh(pm) + (1-z) tanh(pn)
__global__ __launch_bounds__(256) void gated_pair_fwd_kernel(const float* __restrict__ xm, const float* __restrict__ xn,
                                                             const float* __restrict__ w, const float* __restrict__ b,
                                                             const float* __restrict__ pm, const float* __restrict__ pn,
                                                             float* __restrict__ out, float* __restrict__ zs, int R, int D, int C) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= R) return;
    const float* a = xm + (int64_t)row * D;
    const float* v = xn + (int64_t)row * D;
    float s = 0.f;
    for (int j = lane; j < D; j += 64) s += w[j] * a[j] + w[D + j] * v[j] + w[2 * D + j] * (a[j] * v[j]);
    s = wave_sum(s) + b[0];
    const float z = sigm(s);
    if (lane == 0) zs[row] = z;
    for (int j = lane; j < C; j += 64)
        out[(int64_t)row * C + j] = z * tanhf(pm[(int64_t)row * C + j]) + (1.0f - z) * tanhf(pn[(int64_t)row * C + j]);
}

// dpm = z dout (1 - hm^2), dpn = (1-z) dout (1 - hn^2), dpre = z (1-z) sum_c dout (hm - hn);
// dxm = dpre (w1 + w3 xn), dxn = dpre (w2 + w3 xm);  dpre is written out (operand of the gate weight gradient)
__global__ __launch_bounds__(256) void gated_pair_bwd_kernel(const float* __restrict__ xm, const float* __restrict__ xn,
                                                             const float* __restrict__ w, const float* __restrict__ pm,
                                                             const float* __restrict__ pn, const float* __restrict__ zs,
                                                             const float* __restrict__ dout, float* __restrict__ dxm,
                                                             float* __restrict__ dxn, float* __restrict__ dpm,
                                                             float* __restrict__ dpn, float* __restrict__ dpre, int R, int D,
                                                             int C) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= R) return;
    const float z = zs[row];
    float acc = 0.f;
    for (int j = lane; j < C; j += 64) {
        const int64_t o = (int64_t)row * C + j;
        const float hm = tanhf(pm[o]), hn = tanhf(pn[o]), d = dout[o];
        acc += d * (hm - hn);
        dpm[o] = z * d * (1.0f - hm * hm);
        dpn[o] = (1.0f - z) * d * (1.0f - hn * hn);
    }
    const float dp = wave_sum(acc) * z * (1.0f - z);
    if (lane == 0) dpre[row] = dp;
    for (int j = lane; j < D; j += 64) {
        const int64_t o = (int64_t)row * D + j;
        dxm[o] = dp * (w[j] + w[2 * D + j] * xn[o]);
        dxn[o] = dp * (w[D + j] + w[2 * D + j] * xm[o]);
    }
}

// out[0:D] = sum_r s[r] xm[r], out[D:2D] = sum_r s[r] xn[r], out[2D:3D] = sum_r s[r] xm[r] xn[r]; out[3D] = sum_r s[r]
// one workgroup per 64 columns, rows strided over the four waves, partials summed in a fixed order
__global__ __launch_bounds__(256) void rowscale_colsum_kernel(const float* __restrict__ s, const float* __restrict__ xm,
                                                              const float* __restrict__ xn, float* __restrict__ out, int R,
                                                              int D) {
    __shared__ float part[4][4][64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int j = blockIdx.x * 64 + lane;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    if (j < D) {
        for (int r = wv; r < R; r += 4) {
            const float sr = s[r], a = xm[(int64_t)r * D + j], v = xn[(int64_t)r * D + j];
            a0 += sr * a;
            a1 += sr * v;
            a2 += sr * a * v;
            a3 += sr;
        }
    }
    part[wv][0][lane] = a0; part[wv][1][lane] = a1; part[wv][2][lane] = a2; part[wv][3][lane] = a3;
    __syncthreads();
    if (wv == 0 && j < D) {
        float t0 = 0.f, t1 = 0.f, t2 = 0.f, t3 = 0.f;
        for (int q = 0; q < 4; ++q) { t0 += part[q][0][lane]; t1 += part[q][1][lane]; t2 += part[q][2][lane]; t3 += part[q][3][lane]; }
        out[j] = t0;
        out[D + j] = t1;
        out[2 * D + j] = t2;
        if (j == 0) out[3 * D] = t3;
    }
}

inline unsigned grid_for(int64_t n) {
    int64_t b = (n + 255) / 256;
    return (unsigned)(b < 1 ? 1 : (b > 4096 ? 4096 : b));
}

}  // namespace

extern "C" int mmdfn_softmax_scale_fwd(const float* z, const float* c, float* att, float* out, int R, int W, void* stream) {
    if (R <= 0 || W <= 0) return -1;
    hipLaunchKernelGGL(softmax_scale_fwd_kernel, dim3((R + 3) / 4), dim3(256), 0, (hipStream_t)stream, z, c, att, out, R, W);
    MMDFN_CHECK_LAUNCH();
    return 0;
}

extern "C" int mmdfn_softmax_scale_bwd(const float* att, const float* c, const float* dout, float* dz, float* dc, int R, int W,
                                       void* stream) {
    if (R <= 0 || W <= 0) return -1;
    hipLaunchKernelGGL(softmax_scale_bwd_kernel, dim3((R + 3) / 4), dim3(256), 0, (hipStream_t)stream, att, c, dout, dz, dc, R, W);
    MMDFN_CHECK_LAUNCH();
    return 0;
}

extern "C" int mmdfn_mfn_mem_fwd(const float* u, const float* v1, const float* v2, const float* mem, float* out, float* saved,
                                 int64_t n, void* stream) {
    if (n <= 0) return -1;
    hipLaunchKernelGGL(mfn_mem_fwd_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, u, v1, v2, mem, out, saved, n);
    MMDFN_CHECK_LAUNCH();
    return 0;
}

extern "C" int mmdfn_mfn_mem_bwd(const float* saved, const float* mem, const float* dout, float* du, float* dv1, float* dv2,
                                 float* dmem, int64_t n, void* stream) {
    if (n <= 0) return -1;
    hipLaunchKernelGGL(mfn_mem_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, saved, mem, dout, du, dv1, dv2,
                       dmem, n);
    MMDFN_CHECK_LAUNCH();
    return 0;
}

extern "C" int mmdfn_gated_pair_fwd(const float* xm, const float* xn, const float* w, const float* b, const float* pm,
                                    const float* pn, float* out, float* zs, int R, int D, int C, void* stream) {
    if (R <= 0 || D <= 0 || C <= 0) return -1;
    hipLaunchKernelGGL(gated_pair_fwd_kernel, dim3((R + 3) / 4), dim3(256), 0, (hipStream_t)stream, xm, xn, w, b, pm, pn, out, zs,
                       R, D, C);
    MMDFN_CHECK_LAUNCH();
    return 0;
}

extern "C" int mmdfn_gated_pair_bwd(const float* xm, const float* xn, const float* w, const float* pm, const float* pn,
                                    const float* zs, const float* dout, float* dxm, float* dxn, float* dpm, float* dpn,
                                    float* dpre, int R, int D, int C, void* stream) {
    if (R <= 0 || D <= 0 || C <= 0) return -1;
    hipLaunchKernelGGL(gated_pair_bwd_kernel, dim3((R + 3) / 4), dim3(256), 0, (hipStream_t)stream, xm, xn, w, pm, pn, zs, dout,
                       dxm, dxn, dpm, dpn, dpre, R, D, C);
    MMDFN_CHECK_LAUNCH();
    return 0;
}

extern "C" int mmdfn_rowscale_colsum(const float* s, const float* xm, const float* xn, float* out, int R, int D, void* stream) {
    if (R <= 0 || D <= 0) return -1;
    hipLaunchKernelGGL(rowscale_colsum_kernel, dim3((D + 63) / 64), dim3(256), 0, (hipStream_t)stream, s, xm, xn, out, R, D);
    MMDFN_CHECK_LAUNCH();
    return 0;
}
