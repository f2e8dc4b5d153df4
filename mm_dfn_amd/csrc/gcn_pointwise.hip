// K7 / K8 pointwise stages of one GCNII "dynamic fusion" layer (reference model_GCN.py:461-472, 176-189):
//   * LSTM-cell gate math of the reasoning module  (nn.LSTM with seq_len 1, model_GCN.py:466)
//   * GCNII update  relu(theta * [A.x || h0] W + (1-theta) * ((1-alpha) A.x + alpha h0)), dropout mask, + q
// The dense contractions (gate pre-activations, support . W) are GEMMs issued by the host side; these
// kernels fuse everything around them so a layer is ~8 launches instead of ~26 elementwise ones.
// All kernels are grid-stride over float4 elements (H and d are multiples of 4).
#include "mmdfn_internal.h"
#include "../../include/mmdfn_hip.h"

namespace {

__device__ __forceinline__ float sigm(float x) { return 1.0f / (1.0f + expf(-x)); }

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }

// G: (R, 4H) pre-activations, gate order i, f, g, o.  c_prev may be null (zero state).
__global__ void lstm_pointwise_fwd_kernel(const float* __restrict__ G, const float* __restrict__ c_prev,
                                          float* __restrict__ h_out, float* __restrict__ c_out, int64_t R, int H) {
    const int H4 = H / 4;
    const int64_t total = R * H4;
    for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = idx / H4;
        const int u = (int)(idx - r * H4) * 4;
        const float* g = G + r * 4 * H + u;
        const float4 gi = ld4(g), gf = ld4(g + H), gg = ld4(g + 2 * H), go = ld4(g + 3 * H);
        float4 cp = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c_prev) cp = ld4(c_prev + r * H + u);
        float4 c, h;
#define LSTM1(F)                                          \
    {                                                     \
        const float cc = sigm(gf.F) * cp.F + sigm(gi.F) * tanhf(gg.F); \
        c.F = cc;                                         \
        h.F = sigm(go.F) * tanhf(cc);                     \
    }
        LSTM1(x) LSTM1(y) LSTM1(z) LSTM1(w)
#undef LSTM1
        st4(c_out + r * H + u, c);
        st4(h_out + r * H + u, h);
    }
}

// dc_next may be null (no gradient flows into the cell state from later layers).
__global__ void lstm_pointwise_bwd_kernel(const float* __restrict__ G, const float* __restrict__ c_prev,
                                          const float* __restrict__ c_new, const float* __restrict__ dh,
                                          const float* __restrict__ dc_next, float* __restrict__ dG,
                                          float* __restrict__ dc_prev, int64_t R, int H) {
    const int H4 = H / 4;
    const int64_t total = R * H4;
    for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = idx / H4;
        const int u = (int)(idx - r * H4) * 4;
        const float* g = G + r * 4 * H + u;
        const float4 gi = ld4(g), gf = ld4(g + H), gg = ld4(g + 2 * H), go = ld4(g + 3 * H);
        float4 cp = make_float4(0.f, 0.f, 0.f, 0.f), dcn = cp, dhv = cp;
        if (c_prev) cp = ld4(c_prev + r * H + u);
        if (dc_next) dcn = ld4(dc_next + r * H + u);
        if (dh) dhv = ld4(dh + r * H + u);
        const float4 cn = ld4(c_new + r * H + u);
        float4 di, df, dg, dO, dcp;
#define LSTM1(F)                                                     \
    {                                                                \
        const float i = sigm(gi.F), f = sigm(gf.F), gt = tanhf(gg.F), o = sigm(go.F); \
        const float tc = tanhf(cn.F);                                \
        const float dc = dcn.F + dhv.F * o * (1.0f - tc * tc);       \
        dO.F = dhv.F * tc * o * (1.0f - o);                          \
        di.F = dc * gt * i * (1.0f - i);                             \
        df.F = dc * cp.F * f * (1.0f - f);                           \
        dg.F = dc * i * (1.0f - gt * gt);                            \
        dcp.F = dc * f;                                              \
    }
        LSTM1(x) LSTM1(y) LSTM1(z) LSTM1(w)
#undef LSTM1
        float* o = dG + r * 4 * H + u;
        st4(o, di);
        st4(o + H, df);
        st4(o + 2 * H, dg);
        st4(o + 3 * H, dO);
        st4(dc_prev + r * H + u, dcp);
    }
}

// out = relu(theta*P + (1-theta)*((1-alpha)*hi + alpha*h0)) * mask + q ;  S2 = [hi | h0] (R, 2d)
__global__ void gcnii_combine_fwd_kernel(const float* __restrict__ P, const float* __restrict__ S2,
                                         const float* __restrict__ q, const float* __restrict__ mask,
                                         float* __restrict__ out, float theta, float alpha, int64_t R, int d) {
    const int d4 = d / 4;
    const int64_t total = R * d4;
    for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = idx / d4;
        const int c = (int)(idx - r * d4) * 4;
        const float4 p = ld4(P + r * d + c), hi = ld4(S2 + r * 2 * d + c), h0 = ld4(S2 + r * 2 * d + d + c);
        float4 m = make_float4(1.f, 1.f, 1.f, 1.f), qv = make_float4(0.f, 0.f, 0.f, 0.f);
        if (mask) m = ld4(mask + r * d + c);
        if (q) qv = ld4(q + r * d + c);
        float4 o;
#define CMB(F) o.F = fmaxf(theta * p.F + (1.0f - theta) * ((1.0f - alpha) * hi.F + alpha * h0.F), 0.f) * m.F + qv.F;
        CMB(x) CMB(y) CMB(z) CMB(w)
#undef CMB
        st4(out + r * d + c, o);
    }
}

// g = dout * mask * [pre > 0];  dP = theta g;  dS2 = [(1-theta)(1-alpha) g | (1-theta) alpha g]
__global__ void gcnii_combine_bwd_kernel(const float* __restrict__ P, const float* __restrict__ S2,
                                         const float* __restrict__ mask, const float* __restrict__ dout,
                                         float* __restrict__ dP, float* __restrict__ dS2, float theta, float alpha,
                                         int64_t R, int d) {
    const int d4 = d / 4;
    const int64_t total = R * d4;
    const float a1 = (1.0f - theta) * (1.0f - alpha), a2 = (1.0f - theta) * alpha;
    for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = idx / d4;
        const int c = (int)(idx - r * d4) * 4;
        const float4 p = ld4(P + r * d + c), hi = ld4(S2 + r * 2 * d + c), h0 = ld4(S2 + r * 2 * d + d + c);
        const float4 go = ld4(dout + r * d + c);
        float4 m = make_float4(1.f, 1.f, 1.f, 1.f);
        if (mask) m = ld4(mask + r * d + c);
        float4 g;
#define CMB(F)                                                                                   \
    {                                                                                            \
        const float pre = theta * p.F + (1.0f - theta) * ((1.0f - alpha) * hi.F + alpha * h0.F); \
        g.F = (pre > 0.f) ? go.F * m.F : 0.f;                                                    \
    }
        CMB(x) CMB(y) CMB(z) CMB(w)
#undef CMB
        st4(dP + r * d + c, make_float4(theta * g.x, theta * g.y, theta * g.z, theta * g.w));
        st4(dS2 + r * 2 * d + c, make_float4(a1 * g.x, a1 * g.y, a1 * g.z, a1 * g.w));
        st4(dS2 + r * 2 * d + d + c, make_float4(a2 * g.x, a2 * g.y, a2 * g.z, a2 * g.w));
    }
}

inline int grid_for(int64_t total4) {
    int64_t b = (total4 + 255) / 256;
    if (b > 2048) b = 2048;
    if (b < 1) b = 1;
    return (int)b;
}

}  // namespace

extern "C" int mmdfn_lstm_pointwise_fwd(const float* G, const float* c_prev, float* h_out, float* c_out, int64_t R,
                                        int H, void* stream) {
    if (R <= 0 || H <= 0 || (H & 3)) return -1;
    hipLaunchKernelGGL(lstm_pointwise_fwd_kernel, dim3(grid_for(R * (H / 4))), dim3(256), 0, (hipStream_t)stream, G,
                       c_prev, h_out, c_out, R, H);
    MMDFN_CHECK_LAUNCH();
    return 0;
}

extern "C" int mmdfn_lstm_pointwise_bwd(const float* G, const float* c_prev, const float* c_new, const float* dh,
                                        const float* dc_next, float* dG, float* dc_prev, int64_t R, int H,
                                        void* stream) {
    if (R <= 0 || H <= 0 || (H & 3)) return -1;
    hipLaunchKernelGGL(lstm_pointwise_bwd_kernel, dim3(grid_for(R * (H / 4))), dim3(256), 0, (hipStream_t)stream, G,
                       c_prev, c_new, dh, dc_next, dG, dc_prev, R, H);
    MMDFN_CHECK_LAUNCH();
    return 0;
}

extern "C" int mmdfn_gcnii_combine_fwd(const float* P, const float* S2, const float* q, const float* mask,
                                       float* out, float theta, float alpha, int64_t R, int d, void* stream) {
    if (R <= 0 || d <= 0 || (d & 3)) return -1;
    hipLaunchKernelGGL(gcnii_combine_fwd_kernel, dim3(grid_for(R * (d / 4))), dim3(256), 0, (hipStream_t)stream, P, S2,
                       q, mask, out, theta, alpha, R, d);
    MMDFN_CHECK_LAUNCH();
    return 0;
}

extern "C" int mmdfn_gcnii_combine_bwd(const float* P, const float* S2, const float* mask, const float* dout,
                                       float* dP, float* dS2, float theta, float alpha, int64_t R, int d,
                                       void* stream) {
    if (R <= 0 || d <= 0 || (d & 3)) return -1;
    hipLaunchKernelGGL(gcnii_combine_bwd_kernel, dim3(grid_for(R * (d / 4))), dim3(256), 0, (hipStream_t)stream, P, S2,
                       mask, dout, dP, dS2, theta, alpha, R, d);
    MMDFN_CHECK_LAUNCH();
    return 0;
}
