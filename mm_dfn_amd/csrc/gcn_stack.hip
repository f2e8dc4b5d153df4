// Fused kernels of the GCNII "dynamic fusion" stack (reference model_GCN.py:444-488 GCNII_lyc.forward and
// :176-189 GraphConvolution.forward), one launch per stage instead of one per tensor op:
//
//   input stage   x_d = x (.) m_x ;  h0 = relu(x_d W0^T + b0) ;  cur0 = h0 (.) m_0              (model_GCN.py:453-456)
//   gate  (K8)    G = [q | h] [W_ih | W_hh]^T + (b_ih + b_hh) ;  LSTM-cell gate math -> (h', c')    (:463-467, nn.LSTM seq_len 1)
//   layer (K7)    out = relu(theta [hi | h0] W + (1 - theta)((1 - alpha) hi + alpha h0)) (.) m_i + q  (:178-186, :469-472)
// and their backward counterparts.  (hi = A_hat . h' is K6, propagate.hip.)  The dense contractions are true
// contractions and run on the matrix cores as exact-f32 MFMA (v_mfma_f32_16x16x4_f32, k-ordered fp32 fma chain); the
// pointwise work around them lives in the same kernel -- operands are formed while they are staged, results leave
// straight from the accumulators -- so no intermediate (P, G, S2, dP ...) makes a round trip through HBM that the
// algorithm does not need, and nothing is concatenated or transposed in memory.
//
// One structure for all six kernels ("weight-stationary"): the contraction is rows x weights with tiny weights
// (<= 400 x 200) and many rows, so a workgroup parks its slice of the WEIGHTS in LDS once (as k-contiguous rows
// whatever layout the parameter has; row stride = 4 * odd dwords: conflict-free 16-byte fragment reads) and then
// streams 16-row blocks of the activations past it:
//     stage A(block)   : all 256 threads load the block's operands with 16-byte loads, form the A matrix element-wise
//                        (mask, ReLU gate, LSTM gate backward ...), write side outputs (x_d, dpre, dP, dG) and park A
//                        in LDS; the raw loads of block i+1 are issued before the MFMAs of block i,
//     contract         : 4 waves x their 16-column tiles, both MFMA fragments are 16-byte LDS reads
//                        (lane (i, g) holds k = k0 + 4g .. +3 of row / column i; MFMA step j consumes component j),
//     epilogue         : from the accumulators (the LSTM cell needs the four gates of a unit: the four waves own one
//                        gate each and meet in LDS).
// grid = (row groups, column blocks): <= ~256 workgroups, each looping over its row blocks.  These launches are
// latency-bound at dialogue-graph sizes (5 280 rows at BASELINE cfg2); long-dialogue batches (cfg5: ~10^5 rows) keep
// the unfused path whose contractions run on the bf16-piece pipeline.
#include "mmdfn_internal.h"
#include "../../include/mmdfn_hip.h"

namespace {

constexpr int RB = 16;   // rows per block

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ float4 zero4() { return make_float4(0.f, 0.f, 0.f, 0.f); }
__device__ __forceinline__ float4 mul4(float4 a, float4 b) { return make_float4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w); }
__device__ __forceinline__ float4 scl4(float4 a, float s) { return make_float4(a.x * s, a.y * s, a.z * s, a.w * s); }
__device__ __forceinline__ float4 add4(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float sigm(float x) { return 1.0f / (1.0f + expf(-x)); }

// LDS row stride (floats) for rows of K floats: 16-byte aligned and (stride / 4) odd, so that the 16 rows a quarter
// wave reads at one k land on 16 different 16-byte bank groups.
__host__ __device__ inline int lds_stride(int K) {
    int ld = K + 4;
    if (((ld >> 2) & 1) == 0) ld += 4;
    return ld;
}

// Park `ncols` weight rows (output columns n0 .. n0+ncols-1, all K) in LDS as k-contiguous rows sW[n][k].
//   kmajor = false: the parameter is (N, K) with k-contiguous rows (nn.Linear / nn.LSTM layout), row stride ldsrc
//   kmajor = true : the parameter is (K, N) (GraphConvolution.weight; any W used as "x . W"), element [k][n] at k * ldsrc + n
__device__ __forceinline__ void stage_weights(float* sW, int ldw, const float* __restrict__ W, int ldsrc, int n0, int ncols,
                                              int K, bool kmajor) {
    if (!kmajor) {
        const int K4 = K >> 2;
        for (int i = threadIdx.x; i < ncols * K4; i += 256) {
            const int n = i / K4, k = (i - n * K4) * 4;
            st4(sW + n * ldw + k, ld4(W + (int64_t)(n0 + n) * ldsrc + k));
        }
    } else {
        for (int i = threadIdx.x; i < ncols * K; i += 256) {      // consecutive threads: consecutive n (coalesced reads)
            const int k = i / ncols, n = i - k * ncols;
            sW[n * ldw + k] = W[(int64_t)k * ldsrc + n0 + n];
        }
    }
}

// acc[t] += A (16 x K, LDS rows sA[i][k]) . W_t^T (tile t = 16 LDS weight rows starting at sW + wrow0[t] * ldw).
template <int MAXT>
__device__ __forceinline__ void contract(f32x4 (&acc)[MAXT], int nt, const int (&wrow0)[MAXT], const float* sA, int lda,
                                         const float* sW, int ldw, int K) {
    const int lane = threadIdx.x & 63, fi = lane & 15, g = lane >> 4;
    const float* ap = sA + fi * lda + 4 * g;
    const float* bp[MAXT];
#pragma unroll
    for (int t = 0; t < MAXT; ++t) bp[t] = sW + (wrow0[t < nt ? t : 0] + fi) * ldw + 4 * g;
    const int nk = (K + 15) >> 4;
    // fragments of chunk kc + 1 are fetched before the MFMAs of chunk kc (two static register sets)
    float4 a0, a1, b0[MAXT], b1[MAXT];
#define CT_LOAD(A_, B_, KC)                                                                                \
    do {                                                                                                   \
        const bool ok_ = 16 * (KC) + 4 * g < K;      /* K % 4 == 0: a float4 is inside or outside as a whole */ \
        A_ = ok_ ? ld4(ap + 16 * (KC)) : zero4();                                                          \
        _Pragma("unroll") for (int t = 0; t < MAXT; ++t) B_[t] = (ok_ && t < nt) ? ld4(bp[t] + 16 * (KC)) : zero4(); \
    } while (0)
#define CT_MMA(A_, B_)                                                                                     \
    do {                                                                                                   \
        const float av_[4] = {A_.x, A_.y, A_.z, A_.w};                                                     \
        _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                      \
            _Pragma("unroll") for (int t = 0; t < MAXT; ++t) {                                             \
                if (t < nt) {                                                                              \
                    const float bv_ = (j == 0) ? B_[t].x : (j == 1) ? B_[t].y : (j == 2) ? B_[t].z : B_[t].w; \
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av_[j], bv_, acc[t], 0, 0, 0);           \
                }                                                                                          \
            }                                                                                              \
    } while (0)
    CT_LOAD(a0, b0, 0);
    for (int kc = 0; kc < nk; kc += 2) {
        if (kc + 1 < nk) CT_LOAD(a1, b1, kc + 1);
        CT_MMA(a0, b0);
        if (kc + 1 < nk) {
            if (kc + 2 < nk) CT_LOAD(a0, b0, kc + 2);
            CT_MMA(a1, b1);
        }
    }
#undef CT_LOAD
#undef CT_MMA
}

// rows blocks of this workgroup: rb = blockIdx.x, blockIdx.x + gridDim.x, ...
#define FOR_ROW_BLOCKS(rb, nrb) for (int rb = blockIdx.x; rb < (nrb); rb += gridDim.x)

// ------------------------------------------------------------------------------------------------------------------
// input stage:  x_d = x (.) m_x * ms (written to xd, row stride ldxd);  h0 = relu(x_d W0^T + b0);  cur0 = h0 (.) m_0 * ms
//   x (R, F) contiguous, W0 (H, F) nn.Linear layout; masks are keep flags (any value, scaled by ms) or null (= ones).
// ------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gcn_input_fwd_kernel(const float* __restrict__ x, const float* __restrict__ mx,
                                                            const float* __restrict__ W0, const float* __restrict__ b0,
                                                            const float* __restrict__ m0, float* __restrict__ xd,
                                                            float* __restrict__ h0, float* __restrict__ cur0, int R,
                                                            int F, int H, int ldxd, float ms) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int ldw = lds_stride(F);
    float* sW = smem;                         // [H][ldw]
    float* sA = smem + H * ldw;               // [2][16][ldw]
    stage_weights(sW, ldw, W0, F, 0, H, F, false);
    const int nrb = (R + RB - 1) / RB;
    const int F4 = F >> 2;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, fi = lane & 15, g = lane >> 4;
    const int ntiles = (H + 15) >> 4;
    int nt = 0, wrow0[2] = {0, 0};
    for (int t = w; t < ntiles && nt < 2; t += 4) wrow0[nt++] = 16 * t;
    constexpr int NS = 4;                     // float4 slots per thread per block (16 * F / 4 / 256 <= 4 for F <= 256)
    float4 rx[NS], rm[NS];
    auto issue = [&](int rb) {
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int i = threadIdx.x + 256 * s;
            const int r = i / F4, k = (i - r * F4) * 4, row = rb * RB + r;
            const bool ok = i < RB * F4 && row < R;
            rx[s] = ok ? ld4(x + (int64_t)row * F + k) : zero4();
            rm[s] = (ok && mx) ? scl4(ld4(mx + (int64_t)row * F + k), ms) : make_float4(1.f, 1.f, 1.f, 1.f);
        }
    };
    auto park = [&](int rb, float* dst) {
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int i = threadIdx.x + 256 * s;
            if (i >= RB * F4) continue;
            const int r = i / F4, k = (i - r * F4) * 4, row = rb * RB + r;
            const float4 v = mul4(rx[s], rm[s]);
            st4(dst + r * ldw + k, v);
            if (row < R) st4(xd + (int64_t)row * ldxd + k, v);
        }
    };
    int rb0 = blockIdx.x;
    if (rb0 < nrb) { issue(rb0); park(rb0, sA); }
    __syncthreads();
    int buf = 0;
    FOR_ROW_BLOCKS(rb, nrb) {
        const int nxt = rb + gridDim.x;
        if (nxt < nrb) issue(nxt);
        f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
        contract<2>(acc, nt, wrow0, sA + buf * RB * ldw, ldw, sW, ldw, F);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int n = wrow0[t] + fi;
            if (t >= nt || n >= H) continue;
            const float bb = b0 ? b0[n] : 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int rr = rb * RB + 4 * g + r;
                if (rr >= R) continue;
                const float v = fmaxf(acc[t][r] + bb, 0.f);
                h0[(int64_t)rr * H + n] = v;
                cur0[(int64_t)rr * H + n] = m0 ? v * m0[(int64_t)rr * H + n] * ms : v;
            }
        }
        if (nxt < nrb) park(nxt, sA + (buf ^ 1) * RB * ldw);
        __syncthreads();
        buf ^= 1;
    }
}

// backward of the input stage:
//   dpre = (dcur0 (.) m_0 ms + dh0) (.) [h0 > 0]       (written out: operand of the weight gradients)
//   dx   = (dpre W0 + dxd) (.) m_x ms                    dxd = gradient reaching x_d directly (row stride lddxd), may be null
__global__ __launch_bounds__(256) void gcn_input_bwd_kernel(const float* __restrict__ dcur0, const float* __restrict__ m0,
                                                            const float* __restrict__ dh0, const float* __restrict__ h0,
                                                            const float* __restrict__ W0, const float* __restrict__ dxd,
                                                            const float* __restrict__ mx, float* __restrict__ dpre,
                                                            float* __restrict__ dx, int R, int F, int H, int lddxd, float ms) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int ldw = lds_stride(H);
    float* sW = smem;                         // [F][ldw]  (dx = dpre . W0: output column n, contraction index k < H)
    float* sA = smem + F * ldw;               // [2][16][ldw]
    stage_weights(sW, ldw, W0, F, 0, F, H, true);
    const int nrb = (R + RB - 1) / RB;
    const int H4 = H >> 2;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, fi = lane & 15, g = lane >> 4;
    const int ntiles = (F + 15) >> 4;
    int nt = 0, wrow0[4] = {0, 0, 0, 0};
    for (int t = w; t < ntiles && nt < 4; t += 4) wrow0[nt++] = 16 * t;
    constexpr int NS = 2;                     // 16 * H / 4 / 256 <= 2 for H <= 128
    float4 rd[NS], rmk[NS], rh0[NS], rdh[NS];
    auto issue = [&](int rb) {
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int i = threadIdx.x + 256 * s;
            const int r = i / H4, k = (i - r * H4) * 4, row = rb * RB + r;
            const bool ok = i < RB * H4 && row < R;
            const int64_t o = (int64_t)row * H + k;
            rd[s] = (ok && dcur0) ? ld4(dcur0 + o) : zero4();
            rmk[s] = (ok && m0) ? scl4(ld4(m0 + o), ms) : make_float4(1.f, 1.f, 1.f, 1.f);
            rdh[s] = (ok && dh0) ? ld4(dh0 + o) : zero4();
            rh0[s] = ok ? ld4(h0 + o) : zero4();
        }
    };
    auto park = [&](int rb, float* dst) {
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int i = threadIdx.x + 256 * s;
            if (i >= RB * H4) continue;
            const int r = i / H4, k = (i - r * H4) * 4, row = rb * RB + r;
            float4 v = add4(mul4(rd[s], rmk[s]), rdh[s]);
            v.x = rh0[s].x > 0.f ? v.x : 0.f; v.y = rh0[s].y > 0.f ? v.y : 0.f;
            v.z = rh0[s].z > 0.f ? v.z : 0.f; v.w = rh0[s].w > 0.f ? v.w : 0.f;
            st4(dst + r * ldw + k, v);
            if (row < R) st4(dpre + (int64_t)row * H + k, v);
        }
    };
    if ((int)blockIdx.x < nrb) { issue(blockIdx.x); park(blockIdx.x, sA); }
    __syncthreads();
    int buf = 0;
    FOR_ROW_BLOCKS(rb, nrb) {
        const int nxt = rb + gridDim.x;
        if (nxt < nrb) issue(nxt);
        f32x4 acc[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
        contract<4>(acc, nt, wrow0, sA + buf * RB * ldw, ldw, sW, ldw, H);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int n = wrow0[t] + fi;
            if (t >= nt || n >= F) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int rr = rb * RB + 4 * g + r;
                if (rr >= R) continue;
                float v = acc[t][r];
                if (dxd) v += dxd[(int64_t)rr * lddxd + n];
                if (mx) v *= mx[(int64_t)rr * F + n] * ms;
                dx[(int64_t)rr * F + n] = v;
            }
        }
        if (nxt < nrb) park(nxt, sA + (buf ^ 1) * RB * ldw);
        __syncthreads();
        buf ^= 1;
    }
}

// ------------------------------------------------------------------------------------------------------------------
// K7 forward: GCNII update.  pre = theta [hi | h0] W + (1 - theta)((1 - alpha) hi + alpha h0);
//   out = relu(pre) (.) m ms + q   (m, q may be null);  gmask = m ms (.) [pre > 0]  (saved for the backward pass).
//   W (2H, H) as stored by GraphConvolution (k-major for this product); out row stride ldo.
// ------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gcnii_layer_fwd_kernel(const float* __restrict__ hi, const float* __restrict__ h0,
                                                              const float* __restrict__ W, const float* __restrict__ q,
                                                              const float* __restrict__ m, float* __restrict__ out,
                                                              float* __restrict__ gmask, float theta, float alpha, int R,
                                                              int H, int ldo, float ms) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int K = 2 * H;
    const int ldw = lds_stride(K);
    float* sW = smem;                         // [H][ldw]
    float* sA = smem + H * ldw;               // [2][16][ldw]   rows [hi | h0]
    stage_weights(sW, ldw, W, H, 0, H, K, true);
    const int nrb = (R + RB - 1) / RB;
    const int H4 = H >> 2;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, fi = lane & 15, g = lane >> 4;
    const int ntiles = (H + 15) >> 4;
    int nt = 0, wrow0[2] = {0, 0};
    for (int t = w; t < ntiles && nt < 2; t += 4) wrow0[nt++] = 16 * t;
    constexpr int NS = 2;                     // per source: 16 * H / 4 / 256 <= 2
    float4 ra[NS], rb_[NS];
    auto issue = [&](int rb) {
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int i = threadIdx.x + 256 * s;
            const int r = i / H4, k = (i - r * H4) * 4, row = rb * RB + r;
            const bool ok = i < RB * H4 && row < R;
            ra[s] = ok ? ld4(hi + (int64_t)row * H + k) : zero4();
            rb_[s] = ok ? ld4(h0 + (int64_t)row * H + k) : zero4();
        }
    };
    auto park = [&](float* dst) {
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int i = threadIdx.x + 256 * s;
            if (i >= RB * H4) continue;
            const int r = i / H4, k = (i - r * H4) * 4;
            st4(dst + r * ldw + k, ra[s]);
            st4(dst + r * ldw + H + k, rb_[s]);
        }
    };
    if ((int)blockIdx.x < nrb) { issue(blockIdx.x); park(sA); }
    __syncthreads();
    int buf = 0;
    FOR_ROW_BLOCKS(rb, nrb) {
        const int nxt = rb + gridDim.x;
        if (nxt < nrb) issue(nxt);
        const float* A = sA + buf * RB * ldw;
        f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
        contract<2>(acc, nt, wrow0, A, ldw, sW, ldw, K);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int n = wrow0[t] + fi;
            if (t >= nt || n >= H) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int rl = 4 * g + r, rr = rb * RB + rl;
                if (rr >= R) continue;
                const int64_t o = (int64_t)rr * H + n;
                const float pre = theta * acc[t][r] + (1.0f - theta) * ((1.0f - alpha) * A[rl * ldw + n] + alpha * A[rl * ldw + H + n]);
                const float mm = m ? m[o] * ms : 1.0f;
                out[(int64_t)rr * ldo + n] = fmaxf(pre, 0.f) * mm + (q ? q[o] : 0.f);
                gmask[o] = pre > 0.f ? mm : 0.f;
            }
        }
        if (nxt < nrb) park(sA + (buf ^ 1) * RB * ldw);
        __syncthreads();
        buf ^= 1;
    }
}

// K7 backward:  gg = dout (.) gmask  (dout row stride lddo);   dP = theta gg  (written out: operand of dW = [hi | h0]^T dP)
//   [dhi | dh0] = dP W^T + [(1 - theta)(1 - alpha) gg | (1 - theta) alpha gg];   dh0 accumulates when acc_h0 != 0.
//   W (2H, H): row n of W is the k-contiguous weight row of output column n of this product.
__global__ __launch_bounds__(256) void gcnii_layer_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ gmask,
                                                              const float* __restrict__ W, float* __restrict__ dP,
                                                              float* __restrict__ dhi, float* __restrict__ dh0,
                                                              float theta, float alpha, int R, int H, int lddo, int acc_h0) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int N = 2 * H;
    const int ldw = lds_stride(H);
    float* sW = smem;                         // [2H][ldw]
    float* sA = smem + N * ldw;               // [2][16][ldw]   rows dP
    stage_weights(sW, ldw, W, H, 0, N, H, false);
    const int nrb = (R + RB - 1) / RB;
    const int H4 = H >> 2;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, fi = lane & 15, g = lane >> 4;
    const int ntiles = (N + 15) >> 4;
    int nt = 0, wrow0[4] = {0, 0, 0, 0};
    for (int t = w; t < ntiles && nt < 4; t += 4) wrow0[nt++] = 16 * t;
    constexpr int NS = 2;
    float4 rd[NS], rg[NS];
    auto issue = [&](int rb) {
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int i = threadIdx.x + 256 * s;
            const int r = i / H4, k = (i - r * H4) * 4, row = rb * RB + r;
            const bool ok = i < RB * H4 && row < R;
            rd[s] = ok ? ld4(dout + (int64_t)row * lddo + k) : zero4();
            rg[s] = ok ? ld4(gmask + (int64_t)row * H + k) : zero4();
        }
    };
    auto park = [&](int rb, float* dst) {
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int i = threadIdx.x + 256 * s;
            if (i >= RB * H4) continue;
            const int r = i / H4, k = (i - r * H4) * 4, row = rb * RB + r;
            const float4 v = scl4(mul4(rd[s], rg[s]), theta);
            st4(dst + r * ldw + k, v);
            if (row < R) st4(dP + (int64_t)row * H + k, v);
        }
    };
    if ((int)blockIdx.x < nrb) { issue(blockIdx.x); park(blockIdx.x, sA); }
    __syncthreads();
    // (1 - theta)(1 - alpha) gg = c1 dP, (1 - theta) alpha gg = c2 dP   (theta = ln(lamda / l + 1) > 0)
    const float c1 = (1.0f - theta) * (1.0f - alpha) / theta, c2 = (1.0f - theta) * alpha / theta;
    int buf = 0;
    FOR_ROW_BLOCKS(rb, nrb) {
        const int nxt = rb + gridDim.x;
        if (nxt < nrb) issue(nxt);
        const float* A = sA + buf * RB * ldw;
        f32x4 acc[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
        contract<4>(acc, nt, wrow0, A, ldw, sW, ldw, H);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int n = wrow0[t] + fi;
            if (t >= nt || n >= N) continue;
            const int nn = n < H ? n : n - H;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int rl = 4 * g + r, rr = rb * RB + rl;
                if (rr >= R) continue;
                const int64_t o = (int64_t)rr * H + nn;
                const float dp = A[rl * ldw + nn];
                if (n < H) {
                    dhi[o] = acc[t][r] + c1 * dp;
                } else {
                    const float v = acc[t][r] + c2 * dp;
                    dh0[o] = acc_h0 ? dh0[o] + v : v;
                }
            }
        }
        if (nxt < nrb) park(nxt, sA + (buf ^ 1) * RB * ldw);
        __syncthreads();
        buf ^= 1;
    }
}

// ------------------------------------------------------------------------------------------------------------------
// K8 forward: LSTM-cell step of the reasoning module.  G = q W_ih^T + h W_hh^T + bsum (gate order i, f, g, o);
//   c' = sig(f) c + sig(i) tanh(g);  h' = sig(o) tanh(c').   h / c may be null (zero state, first layer).
//   Saves the gate ACTIVATIONS (R, 4H) for the backward pass.  W_ih, W_hh: (4H, H) nn.LSTM layout.
//   Column block (blockIdx.y) = 32 hidden units x their 4 gates (128 weight rows of K = 2H in LDS); wave w contracts
//   gate w of those units; the four gate tiles of a (row, unit) meet in LDS for the cell math.
// ------------------------------------------------------------------------------------------------------------------
constexpr int UB = 32;   // hidden units per column block

__global__ __launch_bounds__(256) void lstm_gate_fwd_kernel(const float* __restrict__ q, const float* __restrict__ h,
                                                            const float* __restrict__ c, const float* __restrict__ Wih,
                                                            const float* __restrict__ Whh, const float* __restrict__ bsum,
                                                            float* __restrict__ gates, float* __restrict__ h_out,
                                                            float* __restrict__ c_out, int R, int H) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int K = h ? 2 * H : H;
    const int ldw = lds_stride(K);
    float* sW = smem;                                  // [4 gates][UB][ldw]
    float* sA = sW + 4 * UB * ldw;                     // [2][16][ldw]   rows [q | h]
    float* sE = sA + 2 * RB * ldw;                     // [4 gates][16][UB + 1] pre-activations
    const int u0 = blockIdx.y * UB;
    const int nu = min(UB, H - u0);
    // weights: LDS row (gate, ul) <- [W_ih | W_hh] row gate * H + u0 + ul; rows of missing units stay zero
    for (int i = threadIdx.x; i < 4 * UB * (ldw >> 2); i += 256) st4(sW + 4 * i, zero4());
    __syncthreads();
    const int H4 = H >> 2;
    for (int i = threadIdx.x; i < 4 * nu * H4; i += 256) {
        const int rowl = i / H4, k = (i - rowl * H4) * 4;
        const int gate = rowl / nu, ul = rowl - gate * nu;
        const int64_t src = (int64_t)(gate * H + u0 + ul) * H + k;
        st4(sW + (gate * UB + ul) * ldw + k, ld4(Wih + src));
        if (h) st4(sW + (gate * UB + ul) * ldw + H + k, ld4(Whh + src));
    }
    const int nrb = (R + RB - 1) / RB;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, fi = lane & 15, g = lane >> 4;
    const int nt = (nu + 15) >> 4;                     // unit tiles of this block (1 or 2), gate = wave
    const int wrow0[2] = {w * UB, w * UB + 16};
    constexpr int NS = 2;
    float4 rq[NS], rh[NS];
    auto issue = [&](int rb) {
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int i = threadIdx.x + 256 * s;
            const int r = i / H4, k = (i - r * H4) * 4, row = rb * RB + r;
            const bool ok = i < RB * H4 && row < R;
            rq[s] = ok ? ld4(q + (int64_t)row * H + k) : zero4();
            rh[s] = (ok && h) ? ld4(h + (int64_t)row * H + k) : zero4();
        }
    };
    auto park = [&](float* dst) {
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int i = threadIdx.x + 256 * s;
            if (i >= RB * H4) continue;
            const int r = i / H4, k = (i - r * H4) * 4;
            st4(dst + r * ldw + k, rq[s]);
            if (h) st4(dst + r * ldw + H + k, rh[s]);
        }
    };
    if ((int)blockIdx.x < nrb) { issue(blockIdx.x); park(sA); }
    __syncthreads();
    int buf = 0;
    FOR_ROW_BLOCKS(rb, nrb) {
        const int nxt = rb + gridDim.x;
        if (nxt < nrb) issue(nxt);
        f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
        contract<2>(acc, nt, wrow0, sA + buf * RB * ldw, ldw, sW, ldw, K);
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) sE[(w * RB + 4 * g + r) * (UB + 1) + 16 * t + fi] = acc[t][r];
        __syncthreads();
        for (int i = threadIdx.x; i < RB * nu; i += 256) {
            const int rl = i / nu, ul = i - rl * nu, rr = rb * RB + rl, u = u0 + ul;
            if (rr >= R) continue;
            const float gi = sigm(sE[(0 * RB + rl) * (UB + 1) + ul] + bsum[u]);
            const float gf = sigm(sE[(1 * RB + rl) * (UB + 1) + ul] + bsum[H + u]);
            const float gg = tanhf(sE[(2 * RB + rl) * (UB + 1) + ul] + bsum[2 * H + u]);
            const float go = sigm(sE[(3 * RB + rl) * (UB + 1) + ul] + bsum[3 * H + u]);
            const int64_t o = (int64_t)rr * H + u;
            const float cn = gf * (c ? c[o] : 0.f) + gi * gg;
            float* gr = gates + (int64_t)rr * 4 * H + u;
            gr[0] = gi; gr[H] = gf; gr[2 * H] = gg; gr[3 * H] = go;
            c_out[o] = cn;
            h_out[o] = go * tanhf(cn);
        }
        if (nxt < nrb) park(sA + (buf ^ 1) * RB * ldw);
        __syncthreads();
        buf ^= 1;
    }
}

// K8 backward: dh' = dh_a + dh_b (either may be null), dc' (may be null) ->
//   dG (R, 4H) pre-activation gradients (written out: operand of the weight gradients), dc (R, H),
//   dq = dG W_ih + dres  (dres: the residual path of the layer input, row stride lddres, may be null),  dh = dG W_hh.
//   has_h = 0 (first layer: zero incoming state): dh / dc are not produced.
//   Column block (blockIdx.y) = 64 output columns of dq (blocks 0 .. nb-1) or of dh (blocks nb .. 2 nb-1), one 16-column
//   tile per wave, K = 4H.  Staging a row block IS the pointwise gate backward (every column block recomputes it, block
//   0 writes dG / dc to memory).
constexpr int CBW = 64;

__global__ __launch_bounds__(256) void lstm_gate_bwd_kernel(const float* __restrict__ gates, const float* __restrict__ c_prev,
                                                            const float* __restrict__ c_new, const float* __restrict__ dh_a,
                                                            const float* __restrict__ dh_b, const float* __restrict__ dc_next,
                                                            const float* __restrict__ Wih, const float* __restrict__ Whh,
                                                            const float* __restrict__ dres, float* __restrict__ dG,
                                                            float* __restrict__ dc_prev, float* __restrict__ dq,
                                                            float* __restrict__ dh_prev, int R, int H, int has_h,
                                                            int lddres) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int K = 4 * H;
    const int ldw = lds_stride(K);
    float* sW = smem;                                  // [CBW][ldw]
    float* sA = sW + CBW * ldw;                        // [16][ldw]   rows dG
    const int nb = (H + CBW - 1) / CBW;
    const bool is_dh = (int)blockIdx.y >= nb;
    const int n0 = (is_dh ? blockIdx.y - nb : blockIdx.y) * CBW;
    const int ncols = min(CBW, H - n0);
    stage_weights(sW, ldw, is_dh ? Whh : Wih, H, n0, ncols, K, true);
    const int nrb = (R + RB - 1) / RB;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, fi = lane & 15, g = lane >> 4;
    const int nt = (16 * w < ncols) ? 1 : 0;
    const int wrow0[1] = {16 * w};
    const bool writer = blockIdx.y == 0;
    FOR_ROW_BLOCKS(rb, nrb) {
        __syncthreads();                               // the previous block's fragment reads (and the weights) are done
        for (int idx = threadIdx.x; idx < RB * H; idx += 256) {
            const int rl = idx / H, u = idx - rl * H;
            const int rr = rb * RB + rl;
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
                if (writer) {
                    float* d = dG + (int64_t)rr * 4 * H + u;
                    d[0] = di; d[H] = df; d[2 * H] = dg; d[3 * H] = dO;
                    if (has_h) dc_prev[o] = dc * gf;
                }
            }
            float* s = sA + rl * ldw + u;
            s[0] = di; s[H] = df; s[2 * H] = dg; s[3 * H] = dO;
        }
        __syncthreads();
        f32x4 acc[1] = {{0.f, 0.f, 0.f, 0.f}};
        contract<1>(acc, nt, wrow0, sA, ldw, sW, ldw, K);
        const int n = n0 + 16 * w + fi;
        if (nt && n < H) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int rr = rb * RB + 4 * g + r;
                if (rr >= R) continue;
                const int64_t o = (int64_t)rr * H + n;
                if (is_dh) dh_prev[o] = acc[0][r];
                else dq[o] = acc[0][r] + (dres ? dres[(int64_t)rr * lddres + n] : 0.f);
            }
        }
    }
}

inline bool bad_dims(int64_t R, int H) { return R <= 0 || H < 4 || (H & 3) || H > 112 || R > (int64_t)1 << 30; }

inline int row_groups(int R, int column_blocks) {
    const int nrb = (R + RB - 1) / RB;
    int gq = (256 + column_blocks - 1) / column_blocks;     // ~ one workgroup per CU (LDS holds one weight slice per CU)
    if (gq > nrb) gq = nrb;
    return gq < 1 ? 1 : gq;
}

}  // namespace

extern "C" int mmdfn_gcn_input_fwd(const float* x, const float* mx, const float* W0, const float* b0, const float* m0,
                                   float* xd, float* h0, float* cur0, int R, int F, int H, int ldxd, float mscale,
                                   void* stream) {
    if (bad_dims(R, H) || F < 4 || (F & 3) || F > 256 || ldxd < F || (ldxd & 3)) return -1;
    const size_t lds = (size_t)(H + 2 * RB) * lds_stride(F) * sizeof(float);
    if (lds > 160 * 1024) return -1;
    if (int e = mmdfn_allow_big_lds(gcn_input_fwd_kernel)) return e;
    hipLaunchKernelGGL(gcn_input_fwd_kernel, dim3(row_groups(R, 1)), dim3(256), lds, (hipStream_t)stream, x, mx, W0, b0, m0, xd,
                       h0, cur0, R, F, H, ldxd, mscale);
    MMDFN_CHECK_LAUNCH();
    return 0;
}

extern "C" int mmdfn_gcn_input_bwd(const float* dcur0, const float* m0, const float* dh0, const float* h0, const float* W0,
                                   const float* dxd, const float* mx, float* dpre, float* dx, int R, int F, int H, int lddxd,
                                   float mscale, void* stream) {
    if (bad_dims(R, H) || F < 4 || (F & 3) || F > 256 || (dxd && (lddxd < F))) return -1;
    const size_t lds = (size_t)(F + 2 * RB) * lds_stride(H) * sizeof(float);
    if (lds > 160 * 1024) return -1;
    if (int e = mmdfn_allow_big_lds(gcn_input_bwd_kernel)) return e;
    hipLaunchKernelGGL(gcn_input_bwd_kernel, dim3(row_groups(R, 1)), dim3(256), lds, (hipStream_t)stream, dcur0, m0, dh0, h0, W0,
                       dxd, mx, dpre, dx, R, F, H, lddxd, mscale);
    MMDFN_CHECK_LAUNCH();
    return 0;
}

extern "C" int mmdfn_lstm_gate_fwd(const float* q, const float* h, const float* c, const float* Wih, const float* Whh,
                                   const float* bsum, float* gates, float* h_out, float* c_out, int R, int H, void* stream) {
    if (bad_dims(R, H) || (h == nullptr) != (c == nullptr)) return -1;
    const int ncb = (H + UB - 1) / UB;
    const size_t lds = ((size_t)(4 * UB + 2 * RB) * lds_stride(h ? 2 * H : H) + 4 * RB * (UB + 1)) * sizeof(float);
    if (lds > 160 * 1024) return -1;
    if (int e = mmdfn_allow_big_lds(lstm_gate_fwd_kernel)) return e;
    hipLaunchKernelGGL(lstm_gate_fwd_kernel, dim3(row_groups(R, ncb), ncb), dim3(256), lds, (hipStream_t)stream, q, h, c, Wih,
                       Whh, bsum, gates, h_out, c_out, R, H);
    MMDFN_CHECK_LAUNCH();
    return 0;
}

extern "C" int mmdfn_lstm_gate_bwd(const float* gates, const float* c_prev, const float* c_new, const float* dh_a,
                                   const float* dh_b, const float* dc_next, const float* Wih, const float* Whh,
                                   const float* dres, float* dG, float* dc_prev, float* dq, float* dh_prev, int R, int H,
                                   int has_h, int lddres, void* stream) {
    if (bad_dims(R, H) || (has_h && (dc_prev == nullptr || dh_prev == nullptr || c_prev == nullptr))) return -1;
    if (dres != nullptr && lddres < H) return -1;
    const int nb = (H + CBW - 1) / CBW;
    const int ncb = has_h ? 2 * nb : nb;
    const size_t lds = (size_t)(CBW + RB) * lds_stride(4 * H) * sizeof(float);
    if (lds > 160 * 1024) return -1;
    if (int e = mmdfn_allow_big_lds(lstm_gate_bwd_kernel)) return e;
    hipLaunchKernelGGL(lstm_gate_bwd_kernel, dim3(row_groups(R, ncb), ncb), dim3(256), lds, (hipStream_t)stream, gates, c_prev,
                       c_new, dh_a, dh_b, dc_next, Wih, Whh, dres, dG, dc_prev, dq, dh_prev, R, H, has_h, lddres);
    MMDFN_CHECK_LAUNCH();
    return 0;
}

extern "C" int mmdfn_gcnii_layer_fwd(const float* hi, const float* h0, const float* W, const float* q, const float* m,
                                     float* out, float* gmask, float theta, float alpha, int R, int H, int ldo, float mscale,
                                     void* stream) {
    if (bad_dims(R, H) || ldo < H) return -1;
    const size_t lds = (size_t)(H + 2 * RB) * lds_stride(2 * H) * sizeof(float);
    if (lds > 160 * 1024) return -1;
    if (int e = mmdfn_allow_big_lds(gcnii_layer_fwd_kernel)) return e;
    hipLaunchKernelGGL(gcnii_layer_fwd_kernel, dim3(row_groups(R, 1)), dim3(256), lds, (hipStream_t)stream, hi, h0, W, q, m, out,
                       gmask, theta, alpha, R, H, ldo, mscale);
    MMDFN_CHECK_LAUNCH();
    return 0;
}

extern "C" int mmdfn_gcnii_layer_bwd(const float* dout, const float* gmask, const float* W, float* dP, float* dhi, float* dh0,
                                     float theta, float alpha, int R, int H, int lddo, int acc_h0, void* stream) {
    if (bad_dims(R, H) || lddo < H || (lddo & 3) || !(theta > 0.f)) return -1;
    const size_t lds = (size_t)(2 * H + 2 * RB) * lds_stride(H) * sizeof(float);
    if (lds > 160 * 1024) return -1;
    if (int e = mmdfn_allow_big_lds(gcnii_layer_bwd_kernel)) return e;
    hipLaunchKernelGGL(gcnii_layer_bwd_kernel, dim3(row_groups(R, 1)), dim3(256), lds, (hipStream_t)stream, dout, gmask, W, dP,
                       dhi, dh0, theta, alpha, R, H, lddo, acc_h0);
    MMDFN_CHECK_LAUNCH();
    return 0;
}
