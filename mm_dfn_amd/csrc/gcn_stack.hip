// Fused kernels of the GCNII "dynamic fusion" stack (reference model_GCN.py:444-488 GCNII_lyc.forward and
// :176-189 GraphConvolution.forward), one launch per stage instead of one per tensor op:
//
//   input stage   x_d = x (.) m_x ;  h0 = relu(x_d W0^T + b0) ;  cur0 = h0 (.) m_0              (model_GCN.py:453-456)
//   gate  (K8)    G = [q | h] [W_ih | W_hh]^T + (b_ih + b_hh) ;  LSTM-cell gate math -> (h', c')    (:463-467, nn.LSTM seq_len 1)
//   layer (K7)    out = relu(theta [hi | h0] W + (1 - theta)((1 - alpha) hi + alpha h0)) (.) m_i + q  (:178-186, :469-472)
// and their backward counterparts.  (hi = A_hat . h' is K6, propagate.hip.)  The dense contractions are true
// contractions and run on the matrix cores as exact-f32 MFMA (v_mfma_f32_16x16x4_f32, k-ordered fp32 fma chain); the
// pointwise work around them lives in the same kernel's prologue (operands formed while they are loaded) and
// epilogue (straight from the accumulators), so no intermediate (P, G, S2, dP ...) makes a round trip through HBM that
// the algorithm does not need, and nothing is concatenated.
//
// Work decomposition: one workgroup = 16 rows x ALL output columns, 4 waves splitting the 16-column MFMA tiles round
// robin.  Both MFMA operands come straight from global memory / L2 in MFMA layout (lane (i, g) holds the 4 values
// k = k0 + 4g .. +3 of row / column i; MFMA step j consumes component j of both -- the same k permutation on A and B):
// activations are k-contiguous (one 16-byte load), weights are read in whichever layout the parameter has
// (k-contiguous rows: one 16-byte load; k-major: four 4-byte loads, 64 contiguous bytes per 16 lanes) -- never
// transposed or copied.  The next 16-wide k chunk is prefetched into a second register set.  These launches are
// latency-bound at dialogue-graph sizes (5 280 rows at BASELINE cfg2): many small workgroups, no LDS staging.
// Long-dialogue batches (cfg5: ~10^5 rows) keep the unfused path whose contractions run on the bf16-piece pipeline.
#include "mmdfn_internal.h"
#include "../../include/mmdfn_hip.h"

namespace {

constexpr int RB = 16;   // rows per workgroup

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float4 zero4() { return make_float4(0.f, 0.f, 0.f, 0.f); }
__device__ __forceinline__ float sigm(float x) { return 1.0f / (1.0f + expf(-x)); }

// acc[t] += A (16 x K) . B_t (K x 16) for this wave's tiles t < nt, K walked in chunks of 16 (nk chunks).
//   load_a(kc)    -> this lane's A values  A[row = lane & 15][16 kc + 4 g + j],       j = 0..3   (zeros outside)
//   load_b(t, kc) -> this lane's B values  B_t[k = 16 kc + 4 g + j][col = lane & 15],  j = 0..3   (zeros outside)
template <int MAXT, class LA, class LB>
__device__ __forceinline__ void rowgemm(f32x4 (&acc)[MAXT], int nt, int nk, LA load_a, LB load_b) {
    float4 a0, a1, b0[MAXT], b1[MAXT];
    a0 = load_a(0);
#pragma unroll
    for (int t = 0; t < MAXT; ++t) b0[t] = (t < nt) ? load_b(t, 0) : zero4();
#define RG_STEP(AC, BC, AN, BN, KC)                                                                     \
    do {                                                                                                \
        if ((KC) + 1 < nk) {                                                                            \
            AN = load_a((KC) + 1);                                                                      \
            _Pragma("unroll") for (int t = 0; t < MAXT; ++t) BN[t] = (t < nt) ? load_b(t, (KC) + 1) : zero4(); \
        }                                                                                               \
        __builtin_amdgcn_sched_barrier(0);                                                              \
        const float av_[4] = {AC.x, AC.y, AC.z, AC.w};                                                  \
        _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                   \
            _Pragma("unroll") for (int t = 0; t < MAXT; ++t) {                                          \
                if (t < nt) {                                                                           \
                    const float bv_ = (j == 0) ? BC[t].x : (j == 1) ? BC[t].y : (j == 2) ? BC[t].z : BC[t].w; \
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av_[j], bv_, acc[t], 0, 0, 0);        \
                }                                                                                       \
            }                                                                                           \
        __builtin_amdgcn_sched_barrier(0);                                                              \
    } while (0)
    for (int kc = 0; kc < nk; kc += 2) {
        RG_STEP(a0, b0, a1, b1, kc);
        if (kc + 1 < nk) RG_STEP(a1, b1, a0, b0, kc + 1);
    }
#undef RG_STEP
}

// weight fragment loads.  kcontig: W (N, K) rows k-contiguous (nn.Linear layout), tile column n, ld = row stride.
__device__ __forceinline__ float4 wfrag_kcontig(const float* __restrict__ W, int ld, int n, bool n_ok, int k, int K) {
    if (!n_ok || k >= K) return zero4();
    return ld4(W + (int64_t)n * ld + k);
}
// kmajor: W (K, N) rows n-contiguous, element [k][n]
__device__ __forceinline__ float4 wfrag_kmajor(const float* __restrict__ W, int ld, int n, bool n_ok, int k, int K) {
    if (!n_ok || k >= K) return zero4();
    const float* p = W + (int64_t)k * ld + n;
    return make_float4(p[0], p[ld], p[2 * ld], p[3 * ld]);
}

// ------------------------------------------------------------------------------------------------------------------
// input stage:  x_d = x (.) m_x (written to xd, row stride ldxd);  h0 = relu(x_d W0^T + b0);  cur0 = h0 (.) m_0
//   x (R, F) contiguous, W0 (H, F) nn.Linear layout, masks may be null (= ones).
// ------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gcn_input_fwd_kernel(const float* __restrict__ x, const float* __restrict__ mx,
                                                            const float* __restrict__ W0, const float* __restrict__ b0,
                                                            const float* __restrict__ m0, float* __restrict__ xd,
                                                            float* __restrict__ h0, float* __restrict__ cur0, int R,
                                                            int F, int H, int ldxd) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, fi = lane & 15, g = lane >> 4;
    const int row0 = blockIdx.x * RB;
    const int row = row0 + fi;
    const bool rok = row < R;
    const int ntiles = (H + 15) / 16;
    int nt = 0;
    for (int t = w; t < ntiles; t += 4) ++nt;
    f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    auto load_a = [&](int kc) -> float4 {
        const int k = 16 * kc + 4 * g;
        if (!rok || k >= F) return zero4();
        float4 v = ld4(x + (int64_t)row * F + k);
        if (mx) {
            const float4 m = ld4(mx + (int64_t)row * F + k);
            v.x *= m.x; v.y *= m.y; v.z *= m.z; v.w *= m.w;
        }
        if (w == 0) *reinterpret_cast<float4*>(xd + (int64_t)row * ldxd + k) = v;   // each element passes here once
        return v;
    };
    auto load_b = [&](int t, int kc) -> float4 {
        const int n = 16 * (w + 4 * t) + fi;
        return wfrag_kcontig(W0, F, n, n < H, 16 * kc + 4 * g, F);
    };
    rowgemm<2>(acc, nt, (F + 15) / 16, load_a, load_b);
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int n = 16 * (w + 4 * t) + fi;
        if (t >= nt || n >= H) continue;
        const float bb = b0 ? b0[n] : 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int rr = row0 + 4 * g + r;
            if (rr >= R) continue;
            const float v = fmaxf(acc[t][r] + bb, 0.f);
            h0[(int64_t)rr * H + n] = v;
            cur0[(int64_t)rr * H + n] = m0 ? v * m0[(int64_t)rr * H + n] : v;
        }
    }
}

// backward of the input stage:
//   dpre = (dcur0 (.) m_0 + dh0) (.) [h0 > 0]          (written out: operand of the weight gradients)
//   dx   = (dpre W0 + dxd) (.) m_x                       dxd = gradient reaching x_d directly (row stride lddxd), may be null
__global__ __launch_bounds__(256) void gcn_input_bwd_kernel(const float* __restrict__ dcur0, const float* __restrict__ m0,
                                                            const float* __restrict__ dh0, const float* __restrict__ h0,
                                                            const float* __restrict__ W0, const float* __restrict__ dxd,
                                                            const float* __restrict__ mx, float* __restrict__ dpre,
                                                            float* __restrict__ dx, int R, int F, int H, int lddxd) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, fi = lane & 15, g = lane >> 4;
    const int row0 = blockIdx.x * RB;
    const int row = row0 + fi;
    const bool rok = row < R;
    const int ntiles = (F + 15) / 16;
    int nt = 0;
    for (int t = w; t < ntiles; t += 4) ++nt;
    f32x4 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    auto load_a = [&](int kc) -> float4 {
        const int k = 16 * kc + 4 * g;
        if (!rok || k >= H) return zero4();
        const int64_t o = (int64_t)row * H + k;
        float4 v = dcur0 ? ld4(dcur0 + o) : zero4();
        if (m0) {
            const float4 m = ld4(m0 + o);
            v.x *= m.x; v.y *= m.y; v.z *= m.z; v.w *= m.w;
        }
        if (dh0) {
            const float4 d = ld4(dh0 + o);
            v.x += d.x; v.y += d.y; v.z += d.z; v.w += d.w;
        }
        const float4 h = ld4(h0 + o);
        v.x = h.x > 0.f ? v.x : 0.f; v.y = h.y > 0.f ? v.y : 0.f; v.z = h.z > 0.f ? v.z : 0.f; v.w = h.w > 0.f ? v.w : 0.f;
        if (w == 0) *reinterpret_cast<float4*>(dpre + o) = v;
        return v;
    };
    auto load_b = [&](int t, int kc) -> float4 {        // dx = dpre . W0 : B[k][n] = W0[k][n], W0 is (H, F) = k-major
        const int n = 16 * (w + 4 * t) + fi;
        return wfrag_kmajor(W0, F, n, n < F, 16 * kc + 4 * g, H);
    };
    rowgemm<4>(acc, nt, (H + 15) / 16, load_a, load_b);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int n = 16 * (w + 4 * t) + fi;
        if (t >= nt || n >= F) continue;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int rr = row0 + 4 * g + r;
            if (rr >= R) continue;
            float v = acc[t][r];
            if (dxd) v += dxd[(int64_t)rr * lddxd + n];
            if (mx) v *= mx[(int64_t)rr * F + n];
            dx[(int64_t)rr * F + n] = v;
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// K8 forward: LSTM-cell step of the reasoning module.  G = q W_ih^T + h W_hh^T + bsum (gate order i, f, g, o);
//   c' = sig(f) c + sig(i) tanh(g);  h' = sig(o) tanh(c').   h / c may be null (zero state, first layer).
//   Saves the gate ACTIVATIONS (R, 4H) for the backward pass.  W_ih, W_hh: (4H, H) nn.LSTM layout.
//   A wave owns unit tiles {w, w + 4} and, for each, the four gate tiles of those units, so the cell math of a unit
//   happens in the registers of the lane that accumulated it.
// ------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void lstm_gate_fwd_kernel(const float* __restrict__ q, const float* __restrict__ h,
                                                            const float* __restrict__ c, const float* __restrict__ Wih,
                                                            const float* __restrict__ Whh, const float* __restrict__ bsum,
                                                            float* __restrict__ gates, float* __restrict__ h_out,
                                                            float* __restrict__ c_out, int R, int H) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, fi = lane & 15, g = lane >> 4;
    const int row0 = blockIdx.x * RB;
    const int row = row0 + fi;
    const bool rok = row < R;
    const int utiles = (H + 15) / 16;
    int nslot = 0;
    for (int ut = w; ut < utiles; ut += 4) ++nslot;
    const int K = h ? 2 * H : H;
    f32x4 acc[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    auto load_a = [&](int kc) -> float4 {
        const int k = 16 * kc + 4 * g;
        if (!rok || k >= K) return zero4();
        return (k < H) ? ld4(q + (int64_t)row * H + k) : ld4(h + (int64_t)row * H + (k - H));
    };
    auto load_b = [&](int t, int kc) -> float4 {        // tile t = 4 * slot + gate
        const int u = 16 * (w + 4 * (t >> 2)) + fi;
        const int k = 16 * kc + 4 * g;
        if (u >= H || k >= K) return zero4();
        const int64_t wrow = (int64_t)((t & 3) * H + u) * H;
        return (k < H) ? ld4(Wih + wrow + k) : ld4(Whh + wrow + (k - H));
    };
    rowgemm<8>(acc, 4 * nslot, (K + 15) / 16, load_a, load_b);
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const int u = 16 * (w + 4 * s) + fi;
        if (s >= nslot || u >= H) continue;
        const float bi = bsum[u], bf = bsum[H + u], bg = bsum[2 * H + u], bo = bsum[3 * H + u];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int rr = row0 + 4 * g + r;
            if (rr >= R) continue;
            const float gi = sigm(acc[4 * s + 0][r] + bi), gf = sigm(acc[4 * s + 1][r] + bf);
            const float gg = tanhf(acc[4 * s + 2][r] + bg), go = sigm(acc[4 * s + 3][r] + bo);
            const float cp = c ? c[(int64_t)rr * H + u] : 0.f;
            const float cn = gf * cp + gi * gg;
            float* gr = gates + (int64_t)rr * 4 * H + u;
            gr[0] = gi; gr[H] = gf; gr[2 * H] = gg; gr[3 * H] = go;
            c_out[(int64_t)rr * H + u] = cn;
            h_out[(int64_t)rr * H + u] = go * tanhf(cn);
        }
    }
}

// K8 backward: dh' = dh_a + dh_b (either may be null), dc' (may be null) ->
//   dG (R, 4H) pre-activation gradients (written out: operand of the weight gradients), dc (R, H),
//   [dq | dh] = dG [W_ih | W_hh]  with  dq += dres  (the residual path of the layer input, may be null).
//   has_h = 0 (first layer: zero incoming state): dh / dc are not produced.
//   Phase 1 (all threads): the pointwise gate backward for the 16 rows, dG to global memory and to LDS;
//   phase 2: the contraction over the 4H gate columns with A read from LDS.
__global__ __launch_bounds__(256) void lstm_gate_bwd_kernel(const float* __restrict__ gates, const float* __restrict__ c_prev,
                                                            const float* __restrict__ c_new, const float* __restrict__ dh_a,
                                                            const float* __restrict__ dh_b, const float* __restrict__ dc_next,
                                                            const float* __restrict__ Wih, const float* __restrict__ Whh,
                                                            const float* __restrict__ dres, float* __restrict__ dG,
                                                            float* __restrict__ dc_prev, float* __restrict__ dq,
                                                            float* __restrict__ dh_prev, int R, int H, int has_h,
                                                            int lddres) {
    extern __shared__ __attribute__((aligned(16))) float sG[];      // [16][4H + 4]
    const int ldg = 4 * H + 4;
    const int row0 = blockIdx.x * RB;
    for (int idx = threadIdx.x; idx < RB * H; idx += 256) {
        const int rl = idx / H, u = idx - rl * H;
        const int rr = row0 + rl;
        float di = 0.f, df = 0.f, dg = 0.f, dO = 0.f;
        if (rr < R) {
            const float* gr = gates + (int64_t)rr * 4 * H + u;
            const float gi = gr[0], gf = gr[H], gg = gr[2 * H], go = gr[3 * H];
            const int64_t o = (int64_t)rr * H + u;
            const float dh = (dh_a ? dh_a[o] : 0.f) + (dh_b ? dh_b[o] : 0.f);
            const float tc = tanhf(c_new[o]);
            const float dc = (dc_next ? dc_next[o] : 0.f) + dh * go * (1.0f - tc * tc);
            const float cp = c_prev ? c_prev[o] : 0.f;
            dO = dh * tc * go * (1.0f - go);
            di = dc * gg * gi * (1.0f - gi);
            df = dc * cp * gf * (1.0f - gf);
            dg = dc * gi * (1.0f - gg * gg);
            float* d = dG + (int64_t)rr * 4 * H + u;
            d[0] = di; d[H] = df; d[2 * H] = dg; d[3 * H] = dO;
            if (has_h) dc_prev[o] = dc * gf;
        }
        float* s = sG + rl * ldg + u;
        s[0] = di; s[H] = df; s[2 * H] = dg; s[3 * H] = dO;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, fi = lane & 15, g = lane >> 4;
    const int Nout = has_h ? 2 * H : H;
    const int ntiles = (Nout + 15) / 16;
    int nt = 0;
    for (int t = w; t < ntiles; t += 4) ++nt;
    const int K = 4 * H;
    f32x4 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    auto load_a = [&](int kc) -> float4 {
        const int k = 16 * kc + 4 * g;
        if (k >= K) return zero4();
        return *reinterpret_cast<const float4*>(sG + fi * ldg + k);    // rows past R hold zeros
    };
    auto load_b = [&](int t, int kc) -> float4 {        // B[k][n] = n < H ? W_ih[k][n] : W_hh[k][n - H]  (k-major)
        const int n = 16 * (w + 4 * t) + fi;
        const int k = 16 * kc + 4 * g;
        if (n >= Nout) return zero4();
        return (n < H) ? wfrag_kmajor(Wih, H, n, true, k, K) : wfrag_kmajor(Whh, H, n - H, true, k, K);
    };
    rowgemm<4>(acc, nt, (K + 15) / 16, load_a, load_b);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int n = 16 * (w + 4 * t) + fi;
        if (t >= nt || n >= Nout) continue;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int rr = row0 + 4 * g + r;
            if (rr >= R) continue;
            if (n < H) {
                const int64_t o = (int64_t)rr * H + n;
                dq[o] = acc[t][r] + (dres ? dres[(int64_t)rr * lddres + n] : 0.f);
            } else {
                dh_prev[(int64_t)rr * H + (n - H)] = acc[t][r];
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// K7 forward: GCNII update.  pre = theta [hi | h0] W + (1 - theta)((1 - alpha) hi + alpha h0);
//   out = relu(pre) (.) m + q   (m, q may be null);  gmask = m (.) [pre > 0]  (saved for the backward pass).
//   W (2H, H) as stored by GraphConvolution (k-major for this product); out row stride ldo.
// ------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gcnii_layer_fwd_kernel(const float* __restrict__ hi, const float* __restrict__ h0,
                                                              const float* __restrict__ W, const float* __restrict__ q,
                                                              const float* __restrict__ m, float* __restrict__ out,
                                                              float* __restrict__ gmask, float theta, float alpha, int R,
                                                              int H, int ldo) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, fi = lane & 15, g = lane >> 4;
    const int row0 = blockIdx.x * RB;
    const int row = row0 + fi;
    const bool rok = row < R;
    const int ntiles = (H + 15) / 16;
    int nt = 0;
    for (int t = w; t < ntiles; t += 4) ++nt;
    const int K = 2 * H;
    f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    auto load_a = [&](int kc) -> float4 {
        const int k = 16 * kc + 4 * g;
        if (!rok || k >= K) return zero4();
        return (k < H) ? ld4(hi + (int64_t)row * H + k) : ld4(h0 + (int64_t)row * H + (k - H));
    };
    auto load_b = [&](int t, int kc) -> float4 {
        const int n = 16 * (w + 4 * t) + fi;
        return wfrag_kmajor(W, H, n, n < H, 16 * kc + 4 * g, K);
    };
    rowgemm<2>(acc, nt, (K + 15) / 16, load_a, load_b);
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int n = 16 * (w + 4 * t) + fi;
        if (t >= nt || n >= H) continue;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int rr = row0 + 4 * g + r;
            if (rr >= R) continue;
            const int64_t o = (int64_t)rr * H + n;
            const float pre = theta * acc[t][r] + (1.0f - theta) * ((1.0f - alpha) * hi[o] + alpha * h0[o]);
            const float mm = m ? m[o] : 1.0f;
            out[(int64_t)rr * ldo + n] = fmaxf(pre, 0.f) * mm + (q ? q[o] : 0.f);
            gmask[o] = pre > 0.f ? mm : 0.f;
        }
    }
}

// K7 backward:  gg = dout (.) gmask  (dout row stride lddo);   dP = theta gg  (written out: operand of dW = [hi | h0]^T dP)
//   [dhi | dh0] = dP W^T + [(1 - theta)(1 - alpha) gg | (1 - theta) alpha gg];   dh0 accumulates when acc_h0 != 0.
//   W (2H, H): row n of W is the k-contiguous weight row of output column n of this product.
__global__ __launch_bounds__(256) void gcnii_layer_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ gmask,
                                                              const float* __restrict__ W, float* __restrict__ dP,
                                                              float* __restrict__ dhi, float* __restrict__ dh0,
                                                              float theta, float alpha, int R, int H, int lddo, int acc_h0) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, fi = lane & 15, g = lane >> 4;
    const int row0 = blockIdx.x * RB;
    const int row = row0 + fi;
    const bool rok = row < R;
    const int ntiles = (2 * H + 15) / 16;
    int nt = 0;
    for (int t = w; t < ntiles; t += 4) ++nt;
    f32x4 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    auto load_a = [&](int kc) -> float4 {
        const int k = 16 * kc + 4 * g;
        if (!rok || k >= H) return zero4();
        float4 v = ld4(dout + (int64_t)row * lddo + k);
        const float4 mk = ld4(gmask + (int64_t)row * H + k);
        v.x *= theta * mk.x; v.y *= theta * mk.y; v.z *= theta * mk.z; v.w *= theta * mk.w;
        if (w == 0) *reinterpret_cast<float4*>(dP + (int64_t)row * H + k) = v;
        return v;
    };
    auto load_b = [&](int t, int kc) -> float4 {
        const int n = 16 * (w + 4 * t) + fi;
        return wfrag_kcontig(W, H, n, n < 2 * H, 16 * kc + 4 * g, H);
    };
    rowgemm<4>(acc, nt, (H + 15) / 16, load_a, load_b);
    const float a1 = (1.0f - theta) * (1.0f - alpha), a2 = (1.0f - theta) * alpha;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int n = 16 * (w + 4 * t) + fi;
        if (t >= nt || n >= 2 * H) continue;
        const int nn = n < H ? n : n - H;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int rr = row0 + 4 * g + r;
            if (rr >= R) continue;
            const int64_t o = (int64_t)rr * H + nn;
            const float gg = dout[(int64_t)rr * lddo + nn] * gmask[o];
            if (n < H) {
                dhi[o] = acc[t][r] + a1 * gg;
            } else {
                const float v = acc[t][r] + a2 * gg;
                dh0[o] = acc_h0 ? dh0[o] + v : v;
            }
        }
    }
}

inline bool bad_dims(int64_t R, int H) { return R <= 0 || H < 4 || (H & 3) || H > 128 || R > (int64_t)1 << 30; }

}  // namespace

extern "C" int mmdfn_gcn_input_fwd(const float* x, const float* mx, const float* W0, const float* b0, const float* m0,
                                   float* xd, float* h0, float* cur0, int R, int F, int H, int ldxd, void* stream) {
    if (bad_dims(R, H) || F < 4 || (F & 3) || F > 256 || ldxd < F || (ldxd & 3)) return -1;
    hipLaunchKernelGGL(gcn_input_fwd_kernel, dim3((R + RB - 1) / RB), dim3(256), 0, (hipStream_t)stream, x, mx, W0, b0, m0, xd,
                       h0, cur0, R, F, H, ldxd);
    MMDFN_CHECK_LAUNCH();
    return 0;
}

extern "C" int mmdfn_gcn_input_bwd(const float* dcur0, const float* m0, const float* dh0, const float* h0, const float* W0,
                                   const float* dxd, const float* mx, float* dpre, float* dx, int R, int F, int H, int lddxd,
                                   void* stream) {
    if (bad_dims(R, H) || F < 4 || (F & 3) || F > 256 || (dxd && (lddxd < F))) return -1;
    hipLaunchKernelGGL(gcn_input_bwd_kernel, dim3((R + RB - 1) / RB), dim3(256), 0, (hipStream_t)stream, dcur0, m0, dh0, h0, W0,
                       dxd, mx, dpre, dx, R, F, H, lddxd);
    MMDFN_CHECK_LAUNCH();
    return 0;
}

extern "C" int mmdfn_lstm_gate_fwd(const float* q, const float* h, const float* c, const float* Wih, const float* Whh,
                                   const float* bsum, float* gates, float* h_out, float* c_out, int R, int H, void* stream) {
    if (bad_dims(R, H) || (h == nullptr) != (c == nullptr)) return -1;
    hipLaunchKernelGGL(lstm_gate_fwd_kernel, dim3((R + RB - 1) / RB), dim3(256), 0, (hipStream_t)stream, q, h, c, Wih, Whh, bsum,
                       gates, h_out, c_out, R, H);
    MMDFN_CHECK_LAUNCH();
    return 0;
}

extern "C" int mmdfn_lstm_gate_bwd(const float* gates, const float* c_prev, const float* c_new, const float* dh_a,
                                   const float* dh_b, const float* dc_next, const float* Wih, const float* Whh,
                                   const float* dres, float* dG, float* dc_prev, float* dq, float* dh_prev, int R, int H,
                                   int has_h, int lddres, void* stream) {
    if (bad_dims(R, H) || (has_h && (dc_prev == nullptr || dh_prev == nullptr || c_prev == nullptr))) return -1;
    if (dres != nullptr && lddres < H) return -1;
    const int lds = RB * (4 * H + 4) * (int)sizeof(float);
    hipLaunchKernelGGL(lstm_gate_bwd_kernel, dim3((R + RB - 1) / RB), dim3(256), lds, (hipStream_t)stream, gates, c_prev, c_new,
                       dh_a, dh_b, dc_next, Wih, Whh, dres, dG, dc_prev, dq, dh_prev, R, H, has_h, lddres);
    MMDFN_CHECK_LAUNCH();
    return 0;
}

extern "C" int mmdfn_gcnii_layer_fwd(const float* hi, const float* h0, const float* W, const float* q, const float* m,
                                     float* out, float* gmask, float theta, float alpha, int R, int H, int ldo, void* stream) {
    if (bad_dims(R, H) || ldo < H) return -1;
    hipLaunchKernelGGL(gcnii_layer_fwd_kernel, dim3((R + RB - 1) / RB), dim3(256), 0, (hipStream_t)stream, hi, h0, W, q, m, out,
                       gmask, theta, alpha, R, H, ldo);
    MMDFN_CHECK_LAUNCH();
    return 0;
}

extern "C" int mmdfn_gcnii_layer_bwd(const float* dout, const float* gmask, const float* W, float* dP, float* dhi, float* dh0,
                                     float theta, float alpha, int R, int H, int lddo, int acc_h0, void* stream) {
    if (bad_dims(R, H) || lddo < H || (lddo & 3)) return -1;
    hipLaunchKernelGGL(gcnii_layer_bwd_kernel, dim3((R + RB - 1) / RB), dim3(256), 0, (hipStream_t)stream, dout, gmask, W, dP,
                       dhi, dh0, theta, alpha, R, H, lddo, acc_h0);
    MMDFN_CHECK_LAUNCH();
    return 0;
}
