// K8 forward for many-row launches (BASELINE cfg5: 24 576 .. 98 304 graph nodes): the LSTM cell of the GCN stack
//     G = [q | h] . [W_ih | W_hh]^T + b_ih + b_hh,  i, f, g, o = sigma / tanh(G),  c' = f c + i g,  h' = o tanh(c')
// (reference model_GCN.py:463-467: nn.LSTM with seq_len 1) with the contraction on the bf16 matrix path.
//
// gcn_stack.hip's lstm_gate_fwd_ws_kernel runs the same stage on exact-f32 MFMAs (16 x 16 x 4: 1/16 of the bf16 rate) and
// measures 61 us = 64 TFLOP/s at 24 576 rows, its matrix pipe ~35 % busy.  Here every fp32 operand is cut exactly into three
// bf16 pieces and the six piece products of weight >= 2^-16 are issued as v_mfma_f32_32x32x16_bf16 -- the arithmetic and the
// software pipeline of propagate_split.hip / linear_split.hip (split_mfma_pipeline.h: fp32-level error, 2.7x less
// matrix-pipe time).  What is specific to this kernel:
//   * workgroup = 128 rows x (4 gates x 32 units): accumulator column tile ct IS gate ct, so a lane ends up with the four
//     pre-activations of its (row, unit) pairs in its own registers and the cell math runs straight from the
//     accumulators -- no staging of G through LDS or memory;
//   * both operands are two-block along k ([q | h] rows, [W_ih | W_hh] weight rows): a 16-byte group lies in one block
//     (H % 4 == 0), so each load picks its source with one select;
//   * gate activations (read again only by the backward pass) leave through nontemporal stores.
// Unit blocks are the fast grid index: the four blocks of a row tile run back to back and share its q / h rows through L2.
#include "mmdfn_internal.h"
#include <stdlib.h>
#include "../../include/mmdfn_hip.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4v __attribute__((ext_vector_type(4)));

constexpr int SBK = 32;
constexpr int SROW = 20;
#ifndef GATE_BWD_WAVES
#define GATE_BWD_WAVES 1      // waves per SIMD the backward kernel is compiled for (2: 256 registers, 17 of them spilled)
#endif

__device__ __forceinline__ float as_f(uint32_t u) { return __builtin_bit_cast(float, u); }
__device__ __forceinline__ uint32_t as_u(float f) { return __builtin_bit_cast(uint32_t, f); }

__device__ __forceinline__ f32x16 mfma_bf16(u32x4 a, u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// the gate non-linearities of gcn_stack.hip (hardware exp / rcp forms, |err| < 3e-7)
__device__ __forceinline__ float sigm(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float tanhf_(float x) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(2.0f * x)); }

__global__ __launch_bounds__(256, 2) void lstm_gate_fwd_split_kernel(
    const float* __restrict__ q, const float* __restrict__ h, const float* __restrict__ c, const float* __restrict__ Wih,
    const float* __restrict__ Whh, const float* __restrict__ bsum, const float* __restrict__ bsum2, float* __restrict__ gates,
    float* __restrict__ h_out, float* __restrict__ c_out, int R, int H, int ldh) {
    constexpr int NCT = 4;
    constexpr int ABLC = 0;
    constexpr int WROWS = 32;
    constexpr int split_stride = 128 * SROW;
    constexpr int stage_stride = 3 * split_stride;
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];

    const int nub = (H + 31) >> 5;
    const int bm = blockIdx.x / nub;
    const int ub = blockIdx.x - bm * nub;
    const int r0 = bm * 128, u0 = ub * 32;
    const int nu = (H - u0 < 32) ? H - u0 : 32;
    const int K = h ? 2 * H : H;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = tid >> 6;
    const int l32 = lane & 31;
    const int kg = lane >> 5;
    const int wrow0 = WROWS * w;

    f32x16 acc[NCT];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[ct][r] = 0.f;

    // B staging tasks: thread -> accumulator column (tid & 127) = (gate, unit); slots (kh = 0 and 1, kg = tid >> 7)
    const int bcol = tid & 127;
    const int bgate = bcol >> 5, bul = bcol & 31;
    const bool bok = bul < nu;
    const int bkg = __builtin_amdgcn_readfirstlane(tid >> 7);
    const int blds = bcol * SROW + 4 * bkg;
    const int64_t wrow = (int64_t)(bgate * H + u0 + (bok ? bul : nu - 1)) * H;
    const float* wih_lane = Wih + wrow;
    const float* whh_lane = h ? Whh + wrow : Wih + wrow;

    const int arow = r0 + wrow0 + l32;
    const int64_t aoff = (int64_t)(arow < R ? arow : R - 1) * H;
    const float* q_lane = q + aoff;
    const float* h_lane = h ? h + (int64_t)(arow < R ? arow : R - 1) * ldh : q + aoff;
    int boff[NCT];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) boff[ct] = (32 * ct + l32) * SROW + 4 * kg;

    const int nchunks = (K + SBK - 1) / SBK;
    const int klast = (nchunks - 1) * SBK;
    const int nfull = K / SBK;
    const int limA = K - 4 * kg;
    const int limB = bok ? K - 4 * bkg : -(1 << 30);

    // a 16-byte group starting at k (a multiple of 4) lies in the first block (k < H) or in the second (H % 4 == 0)
#define SPLIT_ISSUE(SET, K0, SAFE)                                                                         \
    do {                                                                                                   \
        _Pragma("unroll") for (int e = 0; e < 2; ++e)                                                      \
            _Pragma("unroll") for (int h2 = 0; h2 < 2; ++h2) {                                             \
                const int kb_ = (K0) + 16 * e + 8 * h2 + 4 * bkg;                                          \
                const int kc_ = (!(SAFE) || kb_ < K) ? kb_ : K - 4;                                        \
                const float4 v_ = *reinterpret_cast<const float4*>(kc_ < H ? wih_lane + kc_ : whh_lane + (kc_ - H)); \
                braw[SET][e][4 * h2 + 0] = v_.x; braw[SET][e][4 * h2 + 1] = v_.y;                          \
                braw[SET][e][4 * h2 + 2] = v_.z; braw[SET][e][4 * h2 + 3] = v_.w;                          \
            }                                                                                              \
        _Pragma("unroll") for (int f = 0; f < 4; ++f) {                                                    \
            const int ka_ = (K0) + 8 * f + 4 * kg;                                                         \
            const int kc_ = (!(SAFE) || ka_ < K) ? ka_ : K - 4;                                            \
            araw[SET][f] = *reinterpret_cast<const float4*>(kc_ < H ? q_lane + kc_ : h_lane + (kc_ - H));   \
        }                                                                                                  \
    } while (0)

#include "split_mfma_pipeline.h"

    // ---- cell math straight from the accumulators.  C/D layout of a tile: column = lane & 31 (the unit), row = (r & 3) +
    // 8 (r >> 2) + 4 (lane >> 5); tile ct = gate ct (PyTorch order i, f, g, o).
    if (l32 >= nu) return;
    const int unit = u0 + l32;
    float bi = bsum[unit], bf = bsum[H + unit], bg = bsum[2 * H + unit], bo = bsum[3 * H + unit];
    if (bsum2) { bi += bsum2[unit]; bf += bsum2[H + unit]; bg += bsum2[2 * H + unit]; bo += bsum2[3 * H + unit]; }
    float cp[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = r0 + wrow0 + (r & 3) + 8 * (r >> 2) + 4 * kg;
        cp[r] = (c && row < R) ? c[(int64_t)row * H + unit] : 0.f;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = r0 + wrow0 + (r & 3) + 8 * (r >> 2) + 4 * kg;
        if (row >= R) continue;
        const float gi = sigm(acc[0][r] + bi), gf = sigm(acc[1][r] + bf), gg = tanhf_(acc[2][r] + bg), go = sigm(acc[3][r] + bo);
        const float cn = gf * cp[r] + gi * gg;
        const float hn = go * tanhf_(cn);
        float* gr = gates + (int64_t)row * 4 * H + unit;
        __builtin_nontemporal_store(gi, gr);
        __builtin_nontemporal_store(gf, gr + H);
        __builtin_nontemporal_store(gg, gr + 2 * H);
        __builtin_nontemporal_store(go, gr + 3 * H);
        c_out[(int64_t)row * H + unit] = cn;
        h_out[(int64_t)row * ldh + unit] = hn;
    }
}

// ----------------------------------------------------------------------------------------------------------------------
// K8 backward for many-row launches (round 5): [dq | dh_prev] = dG . [W_ih | W_hh] (+ the residual addend of dq) with
//     dG = the LSTM cell's gate gradients, computed HERE from the saved gate values and the incoming gradients
// on the same bf16-piece pipeline.  gcn_stack.hip's lstm_gate_bwd_ws_kernel measures 329 us at 98 304 rows (0.30 of the
// exact-f32 matrix rate, 21 % of the cfg5 step): 16-row blocks against 64-column weight slices, so every row's operand loads
// and gate math are redone by four column blocks.  Here:
//   * workgroup = 128 rows x one OUTPUT (blockIdx.y = 0: dq, 1: dh_prev; H <= 128 columns = four 32-column tiles), so the gate
//     math is done twice per row instead of four times, and the contraction runs at the bf16 matrix rate;
//   * the contraction index k = (gate, unit) is walked in the order  chunk c = units 8c .. 8c+7 x the four gates,
//     k_local = 8 gate + unit offset  (any order is valid as long as both operands use it): the lane (row, kg) of the
//     pipeline's A layout then needs exactly the four gates of units 8c + 4kg .. +3 per chunk -- nine 16-byte loads (i, f, g, o,
//     c', c, dh (two addends), dc') -- and its four float4 of A ARE the four gate gradients of those units (SPLIT_PREP hook
//     of split_mfma_pipeline.h); they also leave for dG / dc_prev from the same registers (gate halves split over the two
//     column-block workgroups of a row block);
//   * B[k][n] = W[gate H + unit][n]: rows of the parameter as stored, read with lane-adjacent columns (coalesced).
// Units past H (the last chunk of H = 100 holds four) contribute exact zeros on both sides.
// ----------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, GATE_BWD_WAVES) void lstm_gate_bwd_split_kernel(
    const float* __restrict__ gates, const float* __restrict__ c_prev, const float* __restrict__ c_new,
    const float* __restrict__ dh_a, const float* __restrict__ dh_b, const float* __restrict__ dc_next,
    const float* __restrict__ Wih, const float* __restrict__ Whh, const float* __restrict__ dres, float* __restrict__ dG,
    float* __restrict__ dc_prev, float* __restrict__ dq, float* __restrict__ dh_prev, int R, int H, int has_h, int lddres, int abl) {
    constexpr int NCT = 4;
    constexpr int ABLC = 0;
    constexpr int WROWS = 32;
    constexpr int split_stride = 128 * SROW;
    constexpr int stage_stride = 3 * split_stride;
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];

    const int r0 = blockIdx.x * 128;
    const bool is_dh = blockIdx.y != 0;
    const float* __restrict__ W = is_dh ? Whh : Wih;
    const int nyb = gridDim.y;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = tid >> 6;
    const int l32 = lane & 31;
    const int kg = lane >> 5;
    const int wrow0 = WROWS * w;

    f32x16 acc[NCT];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[ct][r] = 0.f;

    // B staging tasks: thread -> output column (tid & 127); k slots (kh = 0 and 1, bkg = tid >> 7)
    const int bcol = tid & 127;
    const bool bok = bcol < H;
    const int bkg = __builtin_amdgcn_readfirstlane(tid >> 7);
    const int blds = bcol * SROW + 4 * bkg;
    const float* w_lane = W + (bok ? bcol : H - 1);

    const int arow = r0 + wrow0 + l32;
    const bool arow_ok = arow < R;
    const int64_t arow_c = arow_ok ? arow : R - 1;
    const float* g_lane = gates + arow_c * 4 * H;
    const int64_t aoff = arow_c * H;
    int boff[NCT];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) boff[ct] = (32 * ct + l32) * SROW + 4 * kg;

    // the pipeline sees K = 32 * nchunks with every chunk "full": units past H are clamped at load and zeroed at use here
    const int nchunks = (H + 7) >> 3;
    const int klast = (nchunks - 1) * SBK;
    const int nfull = nchunks;
    const int limA = 1 << 30, limB = 1 << 30;      // (the pipeline's own masks never fire: SPLIT_PREP zeroes what lies past H)

    float4 graw[9];                        // i f g o | c' c dh_a dh_b dc'  of units u0 .. u0 + 3 of the chunk in flight
#define SPLIT_ISSUE(SET, K0, SAFE)                                                                         \
    do {                                                                                                   \
        const int cu_ = ((K0) >> 5) * 8;                  /* first unit of the chunk */                    \
        _Pragma("unroll") for (int e = 0; e < 2; ++e)                                                      \
            _Pragma("unroll") for (int j = 0; j < 8; ++j) {                                                \
                const int kl_ = 16 * e + 4 * bkg + (j & 3) + 8 * (j >> 2);     /* k_local = 8 gate + unit offset */ \
                const int un_ = cu_ + (kl_ & 7);                                                           \
                const int src_ = (kl_ >> 3) * H + (un_ < H ? un_ : H - 1);                                 \
                braw[SET][e][j] = (abl & 8) ? 0.25f : w_lane[(int64_t)src_ * H];                          \
            }                                                                                              \
        if (!(abl & 4)) {                                                                                  \
            const int u0_ = cu_ + 4 * kg;                                                                  \
            const int uc_ = u0_ + 4 <= H ? u0_ : H - 4;                                                    \
            graw[0] = *reinterpret_cast<const float4*>(g_lane + uc_);                                      \
            graw[1] = *reinterpret_cast<const float4*>(g_lane + H + uc_);                                  \
            graw[2] = *reinterpret_cast<const float4*>(g_lane + 2 * H + uc_);                              \
            graw[3] = *reinterpret_cast<const float4*>(g_lane + 3 * H + uc_);                              \
            graw[4] = *reinterpret_cast<const float4*>(c_new + aoff + uc_);                                \
            graw[5] = c_prev ? *reinterpret_cast<const float4*>(c_prev + aoff + uc_) : make_float4(0.f, 0.f, 0.f, 0.f); \
            graw[6] = dh_a ? *reinterpret_cast<const float4*>(dh_a + aoff + uc_) : make_float4(0.f, 0.f, 0.f, 0.f); \
            graw[7] = dh_b ? *reinterpret_cast<const float4*>(dh_b + aoff + uc_) : make_float4(0.f, 0.f, 0.f, 0.f); \
            graw[8] = dc_next ? *reinterpret_cast<const float4*>(dc_next + aoff + uc_) : make_float4(0.f, 0.f, 0.f, 0.f); \
        }                                                                                                  \
    } while (0)

    // gate gradients of the chunk whose raw values are in graw -> araw[SET][gate]; dG / dc_prev leave from here.  B values of
    // units past H are zeroed as well (their k slots must not contribute).
#define GB1_(F)                                                                                            \
    {                                                                                                      \
        const float gi = graw[0].F, gf = graw[1].F, gg = graw[2].F, go = graw[3].F;                        \
        const float tc = tanhf_(graw[4].F);                                                                \
        const float dhv = graw[6].F + graw[7].F;                                                           \
        const float dc = graw[8].F + dhv * go * (1.0f - tc * tc);                                          \
        dO_.F = dhv * tc * go * (1.0f - go);                                                               \
        di_.F = dc * gg * gi * (1.0f - gi);                                                                \
        df_.F = dc * graw[5].F * gf * (1.0f - gf);                                                         \
        dg_.F = dc * gi * (1.0f - gg * gg);                                                                \
        dcp_.F = dc * gf;                                                                                  \
    }
#define SPLIT_PREP(SET, K0)                                                                                \
    do {                                                                                                   \
        const int cu_ = ((K0) >> 5) * 8;                                                                   \
        const int u0_ = cu_ + 4 * kg;                                                                      \
        const bool uok_ = u0_ + 4 <= H;                   /* H % 4 == 0: a 4-unit group is inside or outside */ \
        float4 di_, df_, dg_, dO_, dcp_;                                                                   \
        if (abl & 2) { di_ = graw[0]; df_ = graw[1]; dg_ = graw[2]; dO_ = graw[3]; dcp_ = graw[4]; }       \
        else { GB1_(x) GB1_(y) GB1_(z) GB1_(w) }                                                           \
        if (!uok_) { di_ = df_ = dg_ = dO_ = make_float4(0.f, 0.f, 0.f, 0.f); }                            \
        araw[SET][0] = di_; araw[SET][1] = df_; araw[SET][2] = dg_; araw[SET][3] = dO_;                    \
        if (uok_ && arow_ok && (K0) == pk0_ && !(abl & 1)) {            /* (the clamped repeats of the last chunk do not store again) */ \
            float* d_ = dG + (int64_t)arow * 4 * H + u0_;                                                  \
            if (!is_dh || nyb == 1) {                                                                      \
                *reinterpret_cast<float4*>(d_) = di_;                                                      \
                *reinterpret_cast<float4*>(d_ + H) = df_;                                                  \
            }                                                                                              \
            if (is_dh || nyb == 1) {                                                                       \
                *reinterpret_cast<float4*>(d_ + 2 * H) = dg_;                                              \
                *reinterpret_cast<float4*>(d_ + 3 * H) = dO_;                                              \
                if (has_h) *reinterpret_cast<float4*>(dc_prev + (int64_t)arow * H + u0_) = dcp_;           \
            }                                                                                              \
        }                                                                                                  \
        pk0_ = (K0) + SBK;                                                                                 \
        _Pragma("unroll") for (int e = 0; e < 2; ++e)                                                      \
            _Pragma("unroll") for (int j = 0; j < 8; ++j) {                                                \
                const int kl_ = 16 * e + 4 * bkg + (j & 3) + 8 * (j >> 2);                                 \
                if (cu_ + (kl_ & 7) >= H || !bok) braw[SET][e][j] = 0.f;                                   \
            }                                                                                              \
    } while (0)
    int pk0_ = 0;                          // the k of the next chunk whose gradients have not been stored yet

#include "split_mfma_pipeline.h"
#undef GB1_

    // ---- results straight from the accumulators.  C/D layout of a tile: column = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 kg
    float* const outp = is_dh ? dh_prev : dq;
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) {
        const int col = 32 * ct + l32;
        if (col >= H) continue;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = r0 + wrow0 + (r & 3) + 8 * (r >> 2) + 4 * kg;
            if (row >= R) continue;
            float v = acc[ct][r];
            if (!is_dh && dres) v += dres[(int64_t)row * lddres + col];
            outp[(int64_t)row * H + col] = v;
        }
    }
}

}  // namespace

// -2: shape not covered (the caller keeps the exact-f32 kernels of gcn_stack.hip)
int mmdfn_launch_lstm_gate_bwd_split(const float* gates, const float* c_prev, const float* c_new, const float* dh_a,
                                     const float* dh_b, const float* dc_next, const float* Wih, const float* Whh,
                                     const float* dres, float* dG, float* dc_prev, float* dq, float* dh_prev, int R, int H,
                                     int has_h, int lddres, hipStream_t s) {
    if (H < 8 || (H & 3) || H > 128 || R <= 0) return -2;
    int abl = 0;
#ifdef MMDFN_TUNING
    if (const char* e = getenv("MMDFN_GATE_BWD_ABL")) abl = atoi(e);
#endif
    const int lds_bytes = 2 * 3 * 128 * SROW * 4;
    dim3 grid((R + 127) / 128, has_h ? 2 : 1);
    hipLaunchKernelGGL(lstm_gate_bwd_split_kernel, grid, dim3(256), lds_bytes, s, gates, c_prev, c_new, dh_a, dh_b, dc_next, Wih,
                       Whh, dres, dG, dc_prev, dq, dh_prev, R, H, has_h, lddres, abl);
    MMDFN_CHECK_LAUNCH();
    return 0;
}

// -2: shape not covered (the caller keeps the exact-f32 kernels of gcn_stack.hip)
int mmdfn_launch_lstm_gate_fwd_split(const float* q, const float* h, const float* c, const float* Wih, const float* Whh,
                                     const float* bsum, const float* bsum2, float* gates, float* h_out, float* c_out, int R,
                                     int H, int ldh, hipStream_t s) {
    if (H < 8 || (H & 3) || R <= 0) return -2;
    const int nub = (H + 31) / 32;
    const int lds_bytes = 2 * 3 * 128 * SROW * 4;
    dim3 grid(((R + 127) / 128) * nub);
    hipLaunchKernelGGL(lstm_gate_fwd_split_kernel, grid, dim3(256), lds_bytes, s, q, h, c, Wih, Whh, bsum, bsum2, gates, h_out,
                       c_out, R, H, ldh);
    MMDFN_CHECK_LAUNCH();
    return 0;
}
