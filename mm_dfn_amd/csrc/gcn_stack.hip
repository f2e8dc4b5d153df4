// Fused kernels of the GCNII "dynamic fusion" stack (reference model_GCN.py:444-488 GCNII_lyc.forward and
// :176-189 GraphConvolution.forward), one launch per stage instead of one per tensor op:
//
//   input stage   x_d = x (.) m_x ;  h0 = relu(x_d W0^T + b0) ;  cur0 = h0 (.) m_0              (model_GCN.py:453-456)
//   gate  (K8)    G = [q | h] [W_ih | W_hh]^T + (b_ih + b_hh) ;  LSTM-cell gate math -> (h', c')    (:463-467, nn.LSTM seq_len 1)
//   layer (K7)    out = relu(theta [hi | h0] W + (1 - theta)((1 - alpha) hi + alpha h0)) (.) m_i + q  (:178-186, :469-472)
// and their backward counterparts.  (hi = A_hat . h' is K6, propagate.hip.)  The dense contractions are true
// contractions and run on the matrix cores as exact-f32 MFMA (v_mfma_f32_16x16x4_f32, k-ordered fp32 fma chain); the
// pointwise work around them lives in the same kernel -- operands are formed while they are staged, results are
// finished row-wise from LDS -- so no intermediate (P, G, S2, dP ...) makes a round trip through HBM that the
// algorithm does not need, and nothing is concatenated or transposed in memory.
//
// One structure for all six kernels ("weight-stationary"): the contraction is rows x weights with tiny weights
// (<= 400 x 200) and many rows, so a workgroup parks its slice of the WEIGHTS in LDS once (as k-contiguous rows
// whatever layout the parameter has; row stride = 4 * odd dwords: conflict-free 16-byte fragment reads; rows
// zero-padded to a multiple of 16 so the MFMA loop has no tail predicate) and then streams 16-row blocks of the
// activations past it:
//     stage A(block)   : all 256 threads load the block's operands with 16-byte loads, form the A matrix element-wise
//                        (mask, ReLU gate, LSTM gate backward ...), write side outputs (x_d, dpre, dP, dG) and park A
//                        in LDS; the raw loads of block i+1 (and the epilogue operands of block i) are issued before
//                        the MFMAs of block i,
//     contract         : 4 waves x their 16-column tiles, straight-line code: both MFMA fragments are unconditional
//                        16-byte LDS reads (lane (i, g) holds k = k0 + 4g .. +3 of row / column i; MFMA step j
//                        consumes component j), every wave runs the same number of tiles (a surplus tile recomputes
//                        tile 0 and is dropped) -- a divergent-looking tile predicate makes hipcc wrap every MFMA in
//                        an exec-mask branch and copy the accumulator out after each one (4x slower, measured),
//     epilogue         : accumulators -> LDS -> row-wise 16-byte loads / stores by all threads.
// grid = (row groups, column blocks): <= ~256 workgroups (LDS holds one weight slice per CU), each looping over its
// row blocks.  These launches are latency-bound at dialogue-graph sizes (5 280 rows at BASELINE cfg2); long-dialogue
// batches (cfg5: ~10^5 rows) keep the unfused path whose contractions run on the bf16-piece pipeline.
#include "mmdfn_internal.h"
#include "../../include/mmdfn_hip.h"
#include <stdlib.h>

namespace {

constexpr int RB = 16;   // rows per block

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ float4 zero4() { return make_float4(0.f, 0.f, 0.f, 0.f); }
__device__ __forceinline__ float4 one4() { return make_float4(1.f, 1.f, 1.f, 1.f); }
__device__ __forceinline__ float4 mul4(float4 a, float4 b) { return make_float4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w); }
__device__ __forceinline__ float4 scl4(float4 a, float s) { return make_float4(a.x * s, a.y * s, a.z * s, a.w * s); }
__device__ __forceinline__ float4 add4(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 fma4(float4 a, float s, float4 b) { return make_float4(a.x * s + b.x, a.y * s + b.y, a.z * s + b.z, a.w * s + b.w); }
// Gate non-linearities on the hardware transcendental units (v_exp_f32 / v_rcp_f32, ~1 ulp each), as in gru.hip: the
// accurate libm expf / tanhf are ~60-100 instructions each and the LSTM cell evaluates six per element -- more issue
// slots than the contraction next to it.  Absolute error < 3e-7 (parity budget of the stack: 1e-5).
__device__ __forceinline__ float sigm(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float tanhf_(float x) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(2.0f * x)); }

// Conflict-free 16-byte fragment reads.  A wave's ds_read_b128 is serviced in the lane groups {0-3, 12-15, 20-27}, {4-11, 16-19,
// 28-31} (+32): half of a group comes from k-group g, the other half from g + 1, so with lane (i, g) reading row i, 16-byte unit
// 4 kc + g of a row-major image two lanes of a group always meet on a bank whatever the (odd) row stride (PMC: SQ_LDS_BANK_CONFLICT
// = 50-64 % of SQ_LDS_IDX_ACTIVE in the round-2 stack kernels).  The MFMA does not care which data row its row i carries nor in which
// order k is walked, as long as the result is filed accordingly and both operands agree: lane (i, g) reads data row frag_row(i) --
// even rows for the lanes {0-3, 12-15}, odd rows for {4-11} -- and unit 4 kc + frag_unit(g) with frag_unit = 0, 2, 1, 3.  Within a
// lane group the even rows then sit on even units and the odd rows on odd units (the stride is an odd number of units): 16 distinct
// bank slots.  Accumulator element (r of lane (i, g)) belongs to data row frag_row(4 g + r), output column frag_row(i) of the tile.
typedef float f32x2_t __attribute__((ext_vector_type(2)));
// 16-byte store around the L2 for results the NEXT kernels do not read (operands of the end-of-backward weight-gradient batch)
__device__ __forceinline__ void st4_nt(float* p, float4 v) { __builtin_nontemporal_store((f32x4){v.x, v.y, v.z, v.w}, reinterpret_cast<f32x4*>(p)); }
__device__ __forceinline__ void nt_store2(float* p, float2 v) { __builtin_nontemporal_store((f32x2_t){v.x, v.y}, reinterpret_cast<f32x2_t*>(p)); }
__device__ __forceinline__ int frag_row(int i) { return i < 4 ? 2 * i : (i < 12 ? 2 * (i - 4) + 1 : 2 * (i - 12) + 8); }
__device__ __forceinline__ int frag_unit(int g) { return ((g & 1) << 1) | (g >> 1); }

// LDS row stride (floats) for rows of K floats: room for K rounded up to 16 (the MFMA loop reads whole 16-wide chunks;
// the padding holds zeros), 16-byte aligned and (stride / 4) odd, so that the 16 rows a quarter wave reads at one k
// land on 16 different 16-byte bank groups.
__host__ __device__ inline int lds_stride(int K) {
    int ld = ((K + 15) & ~15) + 4;
    if (((ld >> 2) & 1) == 0) ld += 4;
    return ld;
}

// i / d for the small non-negative indices of the staging loops (i < 2^21 / d) without the ~30-instruction integer division by a
// runtime value: (i + 0.5) / d is at least 0.5 / d away from an integer, far more than the float product's rounding error.
// inv = 1.0f / (float)d, computed once per kernel.
__device__ __forceinline__ int qdiv(int i, float inv) { return (int)(((float)i + 0.5f) * inv); }

// zero columns K .. ld-1 of `nrows` LDS rows (addresses nobody else writes: no barrier needed against the data stores)
__device__ __forceinline__ void zero_pads(float* base, int nrows, int ld, int K) {
    const int np4 = (ld - K) >> 2;
    const float rnp4 = 1.0f / (float)(np4 > 0 ? np4 : 1);
    for (int i = threadIdx.x; i < nrows * np4; i += 256) {
        const int r = qdiv(i, rnp4), j = i - r * np4;
        st4(base + r * ld + K + 4 * j, zero4());
    }
}

// Park `ncols` weight rows (output columns n0 .. n0+ncols-1, all K) in LDS as k-contiguous rows sW[n][k].
//   KMAJOR = false: the parameter is (N, K) with k-contiguous rows (nn.Linear / nn.LSTM layout), row stride ldsrc
//   KMAJOR = true : the parameter is (K, N) (GraphConvolution.weight; any W used as "x . W"), element [k][n] at k * ldsrc + n
// Two phases so that a kernel can put EVERY load of its prologue in flight at once (the weight slice, the first row
// block, the first epilogue operands) and pay the memory latency once: issue() only loads (all <= 26 16-byte slots of
// a thread), commit() writes them to LDS.
constexpr int NWS = 26;      // 26 slots x 256 threads x 16 B = 104 KB >= every weight slice used below
template <bool KMAJOR>
struct WeightStager {
    float4 v[NWS];
    // slot e of a thread holds element i = threadIdx.x + 256 e = (row r, 16-byte unit j) of the slice; (r, j) are stepped from
    // slot to slot (256 = qa inner + qb) instead of divided out per slot: a division by a runtime value is ~30 instructions,
    // and 2 x 26 of them were 1 500 instructions in front of every stack kernel's first MFMA
    __device__ __forceinline__ void issue(const float* __restrict__ W, int ldsrc, int n0, int ncols, int K) {
        const int inner = KMAJOR ? (ncols >> 2) : (K >> 2);       // 16-byte slots per source row
        const int total = KMAJOR ? K * inner : ncols * inner;
        const int qa = 256 / inner, qb = 256 - qa * inner;
        int r = threadIdx.x / inner, j = threadIdx.x - r * inner;
#pragma unroll
        for (int e = 0; e < NWS; ++e) {
            const int i = threadIdx.x + 256 * e;
            const float* src = KMAJOR ? W + (int64_t)r * ldsrc + n0 + 4 * j : W + (int64_t)(n0 + r) * ldsrc + 4 * j;
            v[e] = (i < total) ? ld4(src) : zero4();
            r += qa;
            j += qb;
            if (j >= inner) { j -= inner; ++r; }
        }
    }
    __device__ __forceinline__ void commit(float* sW, int ldw, int ncols, int K) const {
        const int inner = KMAJOR ? (ncols >> 2) : (K >> 2);
        const int total = KMAJOR ? K * inner : ncols * inner;
        const int qa = 256 / inner, qb = 256 - qa * inner;
        int r = threadIdx.x / inner, j = threadIdx.x - r * inner;
#pragma unroll
        for (int e = 0; e < NWS; ++e) {
            const int i = threadIdx.x + 256 * e;
            if (i < total) {
                if (KMAJOR) {                                         // r = k, columns 4j .. 4j+3
                    float* d = sW + (4 * j) * ldw + r;
                    d[0] = v[e].x; d[ldw] = v[e].y; d[2 * ldw] = v[e].z; d[3 * ldw] = v[e].w;
                } else {                                              // r = column, k = 4j
                    st4(sW + r * ldw + 4 * j, v[e]);
                }
            }
            r += qa;
            j += qb;
            if (j >= inner) { j -= inner; ++r; }
        }
        zero_pads(sW, ncols, ldw, K);
    }
};

// acc[t] += A (16 x K, LDS rows sA[i][k]) . W_t^T  for t < NT, tile t = the 16 LDS weight rows starting at row
// wrow0[t].  Straight-line: no predicate on the tile or on k (K is walked in whole chunks of 16: the LDS rows are
// zero-padded), fragments of chunk kc + 1 are fetched before the MFMAs of chunk kc.
template <int NT>
__device__ __forceinline__ void contract(f32x4 (&acc)[NT], const int (&wrow0)[NT], const float* sA, int lda, const float* sW,
                                         int ldw, int K) {
    const int lane = threadIdx.x & 63, fi = lane & 15, g = lane >> 4;
    const float* ap = sA + frag_row(fi) * lda + 4 * frag_unit(g);
    const float* bp[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) bp[t] = sW + (wrow0[t] + frag_row(fi)) * ldw + 4 * frag_unit(g);
    const int nk = (K + 15) >> 4;
    float4 a0, a1, b0[NT], b1[NT];
#define CT_LOAD(A_, B_, KC)                                                          \
    do {                                                                             \
        A_ = ld4(ap + 16 * (KC));                                                    \
        _Pragma("unroll") for (int t = 0; t < NT; ++t) B_[t] = ld4(bp[t] + 16 * (KC)); \
    } while (0)
#define CT_MMA(A_, B_)                                                                                         \
    do {                                                                                                       \
        const float av_[4] = {A_.x, A_.y, A_.z, A_.w};                                                         \
        _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                          \
            _Pragma("unroll") for (int t = 0; t < NT; ++t) {                                                   \
                const float bv_ = (j == 0) ? B_[t].x : (j == 1) ? B_[t].y : (j == 2) ? B_[t].z : B_[t].w;     \
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av_[j], bv_, acc[t], 0, 0, 0);                   \
            }                                                                                                  \
    } while (0)
    CT_LOAD(a0, b0, 0);
    int kc = 0;
    for (; kc + 2 <= nk; kc += 2) {
        CT_LOAD(a1, b1, kc + 1);
        CT_MMA(a0, b0);
        if (kc + 2 < nk) CT_LOAD(a0, b0, kc + 2);      // uniform scalar branch
        CT_MMA(a1, b1);
    }
    if (kc < nk) CT_MMA(a0, b0);
#undef CT_LOAD
#undef CT_MMA
}

// tiles of this wave: global tile indices w, w + 4, ... < ntiles; a surplus slot recomputes tile 0 (result dropped)
template <int NT>
__device__ __forceinline__ int wave_tiles(int (&wrow0)[NT], int ntiles) {
    const int w = threadIdx.x >> 6;
    int nt = 0;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int tile = w + 4 * t;
        wrow0[t] = tile < ntiles ? 16 * tile : 0;
        nt += tile < ntiles ? 1 : 0;
    }
    return nt;
}

// accumulators -> LDS sP[16][ldp] (C/D layout of the 16x16 tile: col = lane & 15, row = 4 (lane >> 4) + r, both through frag_row)
template <int NT>
__device__ __forceinline__ void spill_tiles(const f32x4 (&acc)[NT], const int (&wrow0)[NT], int nt, float* sP, int ldp) {
    const int lane = threadIdx.x & 63, fi = lane & 15, g = lane >> 4;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        if (t < nt) {
#pragma unroll
            for (int r = 0; r < 4; ++r) sP[frag_row(4 * g + r) * ldp + wrow0[t] + frag_row(fi)] = acc[t][r];
        }
    }
}

#define FOR_ROW_BLOCKS(rb, nrb) for (int rb = blockIdx.x; rb < (nrb); rb += gridDim.x)

// ------------------------------------------------------------------------------------------------------------------
// input stage:  x_d = x (.) m_x * ms (written to xd, row stride ldxd);  h0 = relu(x_d W0^T + b0);  cur0 = h0 (.) m_0 * ms
//   x (R, F) contiguous, W0 (H, F) nn.Linear layout; masks are 0 / 1 keep flags (scaled by ms) or null (= ones).
// ------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gcn_input_fwd_kernel(const float* __restrict__ x, const float* __restrict__ mx,
                                                            const float* __restrict__ W0, const float* __restrict__ b0,
                                                            const float* __restrict__ m0, float* __restrict__ xd,
                                                            float* __restrict__ h0, float* __restrict__ cur0, int R,
                                                            int F, int H, int ldxd, float ms) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int ldw = lds_stride(F), ldp = ((H + 15) & ~15) + 4;   // whole 16-column tiles fit a row of sP
    float* sW = smem;                         // [H][ldw]
    float* sA = sW + H * ldw;                 // [2][16][ldw]
    float* sP = sA + 2 * RB * ldw;            // [16][ldp]
    WeightStager<false> wst;
    wst.issue(W0, F, 0, H, F);
    const int nrb = (R + RB - 1) / RB;
    const int F4 = F >> 2, H4 = H >> 2;
    const float rF4 = 1.0f / (float)F4, rH4 = 1.0f / (float)H4;
    int wrow0[2];
    const int nt = wave_tiles<2>(wrow0, (H + 15) >> 4);
    constexpr int NS = 4;                     // float4 slots per thread per block (16 * F / 4 / 256 <= 4 for F <= 256)
    constexpr int NE = 2;                     // epilogue slots (16 * H / 4 / 256 <= 2)
    float4 rx[NS], rm[NS];
    auto issue = [&](int rb) {
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int i = threadIdx.x + 256 * s;
            const int r = qdiv(i, rF4), k = (i - r * F4) * 4, row = rb * RB + r;
            const bool ok = i < RB * F4 && row < R;
            rx[s] = ok ? ld4(x + (int64_t)row * F + k) : zero4();
            rm[s] = (ok && mx) ? ld4(mx + (int64_t)row * F + k) : one4();      // raw flags: arithmetic on a value just loaded would wait for it here
        }
    };
    auto park = [&](int rb, float* dst) {
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int i = threadIdx.x + 256 * s;
            if (i >= RB * F4) continue;
            const int r = qdiv(i, rF4), k = (i - r * F4) * 4, row = rb * RB + r;
            const float4 v = mul4(rx[s], scl4(rm[s], mx ? ms : 1.0f));
            st4(dst + r * ldw + k, v);
            if (row < R) st4_nt(xd + (int64_t)row * ldxd + k, v);       // (read again by the weight-gradient batch only)
        }
    };
    const int rb0 = blockIdx.x;
    if (rb0 < nrb) issue(rb0);
    wst.commit(sW, ldw, H, F);
    zero_pads(sA, 2 * RB, ldw, F);
    if (rb0 < nrb) park(rb0, sA);
    __syncthreads();
    int buf = 0;
    FOR_ROW_BLOCKS(rb, nrb) {
        const int nxt = rb + gridDim.x;
        if (nxt < nrb) issue(nxt);
        float4 em[NE], eb[NE];
#pragma unroll
        for (int s = 0; s < NE; ++s) {
            const int i = threadIdx.x + 256 * s;
            const int r = qdiv(i, rH4), n = (i - r * H4) * 4, row = rb * RB + r;
            const bool ok = i < RB * H4 && row < R;
            em[s] = (ok && m0) ? ld4(m0 + (int64_t)row * H + n) : one4();
            eb[s] = (ok && b0) ? ld4(b0 + n) : zero4();
        }
        f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
        contract<2>(acc, wrow0, sA + buf * RB * ldw, ldw, sW, ldw, F);
        spill_tiles<2>(acc, wrow0, nt, sP, ldp);
        __syncthreads();
#pragma unroll
        for (int s = 0; s < NE; ++s) {
            const int i = threadIdx.x + 256 * s;
            const int r = qdiv(i, rH4), n = (i - r * H4) * 4, row = rb * RB + r;
            if (i >= RB * H4 || row >= R) continue;
            float4 v = add4(ld4(sP + r * ldp + n), eb[s]);
            v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
            st4(h0 + (int64_t)row * H + n, v);
            st4(cur0 + (int64_t)row * H + n, mul4(v, scl4(em[s], m0 ? ms : 1.0f)));
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
    const int ldw = lds_stride(H), ldp = ((F + 15) & ~15) + 4;
    float* sW = smem;                         // [F][ldw]  (dx = dpre . W0: output column n, contraction index k < H)
    float* sA = sW + F * ldw;                 // [2][16][ldw]
    float* sP = sA + 2 * RB * ldw;            // [16][ldp]
    WeightStager<true> wst;
    wst.issue(W0, F, 0, F, H);
    const int nrb = (R + RB - 1) / RB;
    const int H4 = H >> 2, F4 = F >> 2;
    const float rF4 = 1.0f / (float)F4, rH4 = 1.0f / (float)H4;
    int wrow0[4];
    const int nt = wave_tiles<4>(wrow0, (F + 15) >> 4);
    constexpr int NS = 2;                     // 16 * H / 4 / 256 <= 2 for H <= 128
    constexpr int NE = 4;                     // 16 * F / 4 / 256 <= 4
    float4 rd[NS], rmk[NS], rh0[NS], rdh[NS];
    auto issue = [&](int rb) {
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int i = threadIdx.x + 256 * s;
            const int r = qdiv(i, rH4), k = (i - r * H4) * 4, row = rb * RB + r;
            const bool ok = i < RB * H4 && row < R;
            const int64_t o = (int64_t)row * H + k;
            rd[s] = (ok && dcur0) ? ld4(dcur0 + o) : zero4();
            rmk[s] = (ok && m0) ? ld4(m0 + o) : one4();
            rdh[s] = (ok && dh0) ? ld4(dh0 + o) : zero4();
            rh0[s] = ok ? ld4(h0 + o) : zero4();
        }
    };
    auto park = [&](int rb, float* dst) {
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int i = threadIdx.x + 256 * s;
            if (i >= RB * H4) continue;
            const int r = qdiv(i, rH4), k = (i - r * H4) * 4, row = rb * RB + r;
            float4 v = add4(mul4(rd[s], scl4(rmk[s], m0 ? ms : 1.0f)), rdh[s]);
            v.x = rh0[s].x > 0.f ? v.x : 0.f; v.y = rh0[s].y > 0.f ? v.y : 0.f;
            v.z = rh0[s].z > 0.f ? v.z : 0.f; v.w = rh0[s].w > 0.f ? v.w : 0.f;
            st4(dst + r * ldw + k, v);
            if (row < R) st4_nt(dpre + (int64_t)row * H + k, v);
        }
    };
    if ((int)blockIdx.x < nrb) issue(blockIdx.x);
    wst.commit(sW, ldw, F, H);
    zero_pads(sA, 2 * RB, ldw, H);
    if ((int)blockIdx.x < nrb) park(blockIdx.x, sA);
    __syncthreads();
    int buf = 0;
    FOR_ROW_BLOCKS(rb, nrb) {
        const int nxt = rb + gridDim.x;
        if (nxt < nrb) issue(nxt);
        float4 ed[NE], em[NE];
#pragma unroll
        for (int s = 0; s < NE; ++s) {
            const int i = threadIdx.x + 256 * s;
            const int r = qdiv(i, rF4), n = (i - r * F4) * 4, row = rb * RB + r;
            const bool ok = i < RB * F4 && row < R;
            ed[s] = (ok && dxd) ? ld4(dxd + (int64_t)row * lddxd + n) : zero4();
            em[s] = (ok && mx) ? ld4(mx + (int64_t)row * F + n) : one4();
        }
        f32x4 acc[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
        contract<4>(acc, wrow0, sA + buf * RB * ldw, ldw, sW, ldw, H);
        spill_tiles<4>(acc, wrow0, nt, sP, ldp);
        __syncthreads();
#pragma unroll
        for (int s = 0; s < NE; ++s) {
            const int i = threadIdx.x + 256 * s;
            const int r = qdiv(i, rF4), n = (i - r * F4) * 4, row = rb * RB + r;
            if (i >= RB * F4 || row >= R) continue;
            st4(dx + (int64_t)row * F + n, mul4(add4(ld4(sP + r * ldp + n), ed[s]), scl4(em[s], mx ? ms : 1.0f)));
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
    const int ldw = lds_stride(K), ldp = ((H + 15) & ~15) + 4;
    float* sW = smem;                         // [H][ldw]
    float* sA = sW + H * ldw;                 // [2][16][ldw]   rows [hi | h0]
    float* sP = sA + 2 * RB * ldw;            // [16][ldp]
    WeightStager<true> wst;
    wst.issue(W, H, 0, H, K);
    const int nrb = (R + RB - 1) / RB;
    const int H4 = H >> 2;
    const float rH4 = 1.0f / (float)H4;
    int wrow0[2];
    const int nt = wave_tiles<2>(wrow0, (H + 15) >> 4);
    constexpr int NS = 2;                     // per source: 16 * H / 4 / 256 <= 2
    float4 ra[NS], rb_[NS];
    auto issue = [&](int rb) {
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int i = threadIdx.x + 256 * s;
            const int r = qdiv(i, rH4), k = (i - r * H4) * 4, row = rb * RB + r;
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
            const int r = qdiv(i, rH4), k = (i - r * H4) * 4;
            st4(dst + r * ldw + k, ra[s]);
            st4(dst + r * ldw + H + k, rb_[s]);
        }
    };
    if ((int)blockIdx.x < nrb) issue(blockIdx.x);
    wst.commit(sW, ldw, H, K);
    zero_pads(sA, 2 * RB, ldw, K);
    if ((int)blockIdx.x < nrb) park(sA);
    __syncthreads();
    int buf = 0;
    FOR_ROW_BLOCKS(rb, nrb) {
        const int nxt = rb + gridDim.x;
        if (nxt < nrb) issue(nxt);
        const float* A = sA + buf * RB * ldw;
        float4 eq[NS], em[NS];                // epilogue operands of this block: in flight while the matrix cores work
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int i = threadIdx.x + 256 * s;
            const int r = qdiv(i, rH4), n = (i - r * H4) * 4, row = rb * RB + r;
            const bool ok = i < RB * H4 && row < R;
            eq[s] = (ok && q) ? ld4(q + (int64_t)row * H + n) : zero4();
            em[s] = (ok && m) ? ld4(m + (int64_t)row * H + n) : one4();
        }
        f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
        contract<2>(acc, wrow0, A, ldw, sW, ldw, K);
        spill_tiles<2>(acc, wrow0, nt, sP, ldp);
        __syncthreads();
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int i = threadIdx.x + 256 * s;
            const int r = qdiv(i, rH4), n = (i - r * H4) * 4, row = rb * RB + r;
            if (i >= RB * H4 || row >= R) continue;
            const float4 p = ld4(sP + r * ldp + n), vh = ld4(A + r * ldw + n), v0 = ld4(A + r * ldw + H + n);
            const float4 ems = scl4(em[s], m ? ms : 1.0f);
            float4 o, gm;
#define K7F(F_)                                                                                            \
    {                                                                                                      \
        const float pre = theta * p.F_ + (1.0f - theta) * ((1.0f - alpha) * vh.F_ + alpha * v0.F_);       \
        o.F_ = fmaxf(pre, 0.f) * ems.F_ + eq[s].F_;                                                        \
        gm.F_ = pre > 0.f ? ems.F_ : 0.f;                                                                  \
    }
            K7F(x) K7F(y) K7F(z) K7F(w)
#undef K7F
            st4(out + (int64_t)row * ldo + n, o);
            st4_nt(gmask + (int64_t)row * H + n, gm);            // (saved for the backward pass)
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
                                                              float theta, float alpha, int R, int H, int lddo, int acc_h0, int lddhi) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int N = 2 * H;
    const int ldw = lds_stride(H), ldp = ((N + 15) & ~15) + 4;
    float* sW = smem;                         // [2H][ldw]
    float* sA = sW + N * ldw;                 // [2][16][ldw]   rows dP
    float* sP = sA + 2 * RB * ldw;            // [16][ldp]
    WeightStager<false> wst;
    wst.issue(W, H, 0, N, H);
    const int nrb = (R + RB - 1) / RB;
    const int H4 = H >> 2;
    const float rH4 = 1.0f / (float)H4;
    int wrow0[4];
    const int nt = wave_tiles<4>(wrow0, (N + 15) >> 4);
    constexpr int NS = 2;
    float4 rd[NS], rg[NS];
    auto issue = [&](int rb) {
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int i = threadIdx.x + 256 * s;
            const int r = qdiv(i, rH4), k = (i - r * H4) * 4, row = rb * RB + r;
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
            const int r = qdiv(i, rH4), k = (i - r * H4) * 4, row = rb * RB + r;
            const float4 v = scl4(mul4(rd[s], rg[s]), theta);
            st4(dst + r * ldw + k, v);
            if (row < R) st4_nt(dP + (int64_t)row * H + k, v);
        }
    };
    if ((int)blockIdx.x < nrb) issue(blockIdx.x);
    wst.commit(sW, ldw, N, H);
    zero_pads(sA, 2 * RB, ldw, H);
    if ((int)blockIdx.x < nrb) park(blockIdx.x, sA);
    __syncthreads();
    // (1 - theta)(1 - alpha) gg = c1 dP, (1 - theta) alpha gg = c2 dP   (theta = ln(lamda / l + 1) > 0)
    const float c1 = (1.0f - theta) * (1.0f - alpha) / theta, c2 = (1.0f - theta) * alpha / theta;
    int buf = 0;
    FOR_ROW_BLOCKS(rb, nrb) {
        const int nxt = rb + gridDim.x;
        if (nxt < nrb) issue(nxt);
        const float* A = sA + buf * RB * ldw;
        float4 eo[NS];                        // running dh0 of this block (the h0 half is the accumulating one)
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int i = threadIdx.x + 256 * s;
            const int r = qdiv(i, rH4), n = (i - r * H4) * 4, row = rb * RB + r;
            eo[s] = (acc_h0 && i < RB * H4 && row < R) ? ld4(dh0 + (int64_t)row * H + n) : zero4();
        }
        f32x4 acc[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
        contract<4>(acc, wrow0, A, ldw, sW, ldw, H);
        spill_tiles<4>(acc, wrow0, nt, sP, ldp);
        __syncthreads();
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int i = threadIdx.x + 256 * s;
            const int r = qdiv(i, rH4), n = (i - r * H4) * 4, row = rb * RB + r;
            if (i >= RB * H4 || row >= R) continue;
            const float4 dp = ld4(A + r * ldw + n);
            st4(dhi + (int64_t)row * lddhi + n, fma4(dp, c1, ld4(sP + r * ldp + n)));
            st4(dh0 + (int64_t)row * H + n, add4(fma4(dp, c2, ld4(sP + r * ldp + H + n)), eo[s]));
        }
        if (nxt < nrb) park(nxt, sA + (buf ^ 1) * RB * ldw);
        __syncthreads();
        buf ^= 1;
    }
}

// ------------------------------------------------------------------------------------------------------------------
// K7 producer / consumer forms (round 5; the structure of the K8 kernels below).  The 4-wave kernels above run a row block
// as load -> park -> barrier -> 100 MFMAs per wave -> LDS round trip -> epilogue with everybody in every phase: at 98 304
// rows the launch reaches 0.25 of the exact-f32 matrix rate and 0.3 of HBM (profiles/r04_cfg5_stream_b32_kernel_stats.csv).
// Here waves 0-3 only contract (A(i) against the parked weights, results to sP[i & 1]) while waves 4-7 run the epilogue of
// block i - 1 (from sP[(i - 1) & 1] and the A rows of block i - 1, which they read back from the LDS buffer right before they
// overwrite it with block i + 1 -- same thread, same elements), keep block i + 2's rows and block i's epilogue operands in
// flight, and park block i + 1.  One barrier per block; every SIMD holds one wave of each kind.
// ------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512) void gcnii_layer_fwd_ws_kernel(const float* __restrict__ hi, const float* __restrict__ h0,
                                                                 const float* __restrict__ W, const float* __restrict__ q,
                                                                 const float* __restrict__ m, float* __restrict__ out,
                                                                 float* __restrict__ gmask, float theta, float alpha, int R,
                                                                 int H, int ldo, float ms) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int K = 2 * H;
    const int ldw = lds_stride(K), ldp = ((H + 15) & ~15) + 4;
    float* sW = smem;                         // [H][ldw]
    float* sA = sW + H * ldw;                 // [2][16][ldw]   rows [hi | h0]
    float* sP = sA + 2 * RB * ldw;            // [2][16][ldp]
    const int nrb = (R + RB - 1) / RB;
    const int H4 = H >> 2;
    const float rH4 = 1.0f / (float)H4;
    const int tid = threadIdx.x & 255;
    const bool producer = threadIdx.x >= 256;
    const int first = blockIdx.x, stride = gridDim.x;
    constexpr int NS = 2;                     // per source: 16 * H / 4 / 256 <= 2
    float4 ra[NS], rb_[NS], eq[NS], em[NS];
    auto issue = [&](int rb) {
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int i = tid + 256 * s;
            const int r = qdiv(i, rH4), k = (i - r * H4) * 4, row = rb * RB + r;
            const bool ok = i < RB * H4 && row < R;
            ra[s] = ok ? ld4(hi + (int64_t)row * H + k) : zero4();
            rb_[s] = ok ? ld4(h0 + (int64_t)row * H + k) : zero4();
        }
    };
    auto park = [&](float* dst) {
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int i = tid + 256 * s;
            if (i >= RB * H4) continue;
            const int r = qdiv(i, rH4), k = (i - r * H4) * 4;
            st4(dst + r * ldw + k, ra[s]);
            st4(dst + r * ldw + H + k, rb_[s]);
        }
    };
    auto issue_epi = [&](int rb) {            // q / keep flags of block rb: requested one iteration before its epilogue
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int i = tid + 256 * s;
            const int r = qdiv(i, rH4), n = (i - r * H4) * 4, row = rb * RB + r;
            const bool ok = i < RB * H4 && row < R;
            eq[s] = (ok && q) ? ld4(q + (int64_t)row * H + n) : zero4();
            em[s] = (ok && m) ? ld4(m + (int64_t)row * H + n) : one4();
        }
    };
    auto epilogue = [&](int rb, const float* A, const float* P) {
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int i = tid + 256 * s;
            const int r = qdiv(i, rH4), n = (i - r * H4) * 4, row = rb * RB + r;
            if (i >= RB * H4 || row >= R) continue;
            const float4 p = ld4(P + r * ldp + n), vh = ld4(A + r * ldw + n), v0 = ld4(A + r * ldw + H + n);
            const float4 ems = scl4(em[s], m ? ms : 1.0f);
            float4 o, gm;
#define K7F(F_)                                                                                            \
    {                                                                                                      \
        const float pre = theta * p.F_ + (1.0f - theta) * ((1.0f - alpha) * vh.F_ + alpha * v0.F_);       \
        o.F_ = fmaxf(pre, 0.f) * ems.F_ + eq[s].F_;                                                        \
        gm.F_ = pre > 0.f ? ems.F_ : 0.f;                                                                  \
    }
            K7F(x) K7F(y) K7F(z) K7F(w)
#undef K7F
            st4(out + (int64_t)row * ldo + n, o);
            st4_nt(gmask + (int64_t)row * H + n, gm);            // (saved for the backward pass)
        }
    };
    if (producer) {
        if (first < nrb) { issue(first); issue_epi(first); }
    } else {
        WeightStager<true> wst;
        wst.issue(W, H, 0, H, K);
        wst.commit(sW, ldw, H, K);
        zero_pads(sA, 2 * RB, ldw, K);
    }
    if (producer) {
        if (first < nrb) park(sA);
        if (first + stride < nrb) issue(first + stride);
    }
    __syncthreads();
    int wrow0[2];
    const int nt = wave_tiles<2>(wrow0, (H + 15) >> 4);        // (consumer waves: threadIdx.x >> 6 = 0..3)
    int buf = 0, prev_rb = -1;
    for (int rb = first; rb < nrb; rb += stride) {
        const int nxt = rb + stride;
        if (producer) {
            // epilogue of the previous block (its A rows still sit in the buffer that block nxt is parked into right after),
            // then this block's epilogue operands and block nxt + stride's rows go out
            if (prev_rb >= 0) epilogue(prev_rb, sA + (buf ^ 1) * RB * ldw, sP + (buf ^ 1) * RB * ldp);
            if (prev_rb >= 0) issue_epi(rb);
            if (nxt < nrb) {
                park(sA + (buf ^ 1) * RB * ldw);
                if (nxt + stride < nrb) issue(nxt + stride);
            }
        } else {
            f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
            contract<2>(acc, wrow0, sA + buf * RB * ldw, ldw, sW, ldw, K);
            spill_tiles<2>(acc, wrow0, nt, sP + buf * RB * ldp, ldp);
        }
        __syncthreads();
        prev_rb = rb;
        buf ^= 1;
    }
    if (producer && prev_rb >= 0) epilogue(prev_rb, sA + (buf ^ 1) * RB * ldw, sP + (buf ^ 1) * RB * ldp);
}

__global__ __launch_bounds__(512) void gcnii_layer_bwd_ws_kernel(const float* __restrict__ dout, const float* __restrict__ gmask,
                                                                 const float* __restrict__ W, float* __restrict__ dP,
                                                                 float* __restrict__ dhi, float* __restrict__ dh0,
                                                                 float theta, float alpha, int R, int H, int lddo, int acc_h0,
                                                                 int lddhi) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int N = 2 * H;
    const int ldw = lds_stride(H), ldp = ((N + 15) & ~15) + 4;
    float* sW = smem;                         // [2H][ldw]
    float* sA = sW + N * ldw;                 // [2][16][ldw]   rows dP
    float* sP = sA + 2 * RB * ldw;            // [2][16][ldp]
    const int nrb = (R + RB - 1) / RB;
    const int H4 = H >> 2;
    const float rH4 = 1.0f / (float)H4;
    const int tid = threadIdx.x & 255;
    const bool producer = threadIdx.x >= 256;
    const int first = blockIdx.x, stride = gridDim.x;
    constexpr int NS = 2;
    float4 rd[NS], rg[NS], eo[NS];
    auto issue = [&](int rb) {
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int i = tid + 256 * s;
            const int r = qdiv(i, rH4), k = (i - r * H4) * 4, row = rb * RB + r;
            const bool ok = i < RB * H4 && row < R;
            rd[s] = ok ? ld4(dout + (int64_t)row * lddo + k) : zero4();
            rg[s] = ok ? ld4(gmask + (int64_t)row * H + k) : zero4();
        }
    };
    auto park = [&](int rb, float* dst) {
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int i = tid + 256 * s;
            if (i >= RB * H4) continue;
            const int r = qdiv(i, rH4), k = (i - r * H4) * 4, row = rb * RB + r;
            const float4 v = scl4(mul4(rd[s], rg[s]), theta);
            st4(dst + r * ldw + k, v);
            if (row < R) st4_nt(dP + (int64_t)row * H + k, v);
        }
    };
    auto issue_epi = [&](int rb) {            // running dh0 of block rb (the h0 half is the accumulating one)
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int i = tid + 256 * s;
            const int r = qdiv(i, rH4), n = (i - r * H4) * 4, row = rb * RB + r;
            eo[s] = (acc_h0 && i < RB * H4 && row < R) ? ld4(dh0 + (int64_t)row * H + n) : zero4();
        }
    };
    // (1 - theta)(1 - alpha) gg = c1 dP, (1 - theta) alpha gg = c2 dP   (theta = ln(lamda / l + 1) > 0)
    const float c1 = (1.0f - theta) * (1.0f - alpha) / theta, c2 = (1.0f - theta) * alpha / theta;
    auto epilogue = [&](int rb, const float* A, const float* P) {
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int i = tid + 256 * s;
            const int r = qdiv(i, rH4), n = (i - r * H4) * 4, row = rb * RB + r;
            if (i >= RB * H4 || row >= R) continue;
            const float4 dp = ld4(A + r * ldw + n);
            st4(dhi + (int64_t)row * lddhi + n, fma4(dp, c1, ld4(P + r * ldp + n)));
            st4(dh0 + (int64_t)row * H + n, add4(fma4(dp, c2, ld4(P + r * ldp + H + n)), eo[s]));
        }
    };
    if (producer) {
        if (first < nrb) { issue(first); issue_epi(first); }
    } else {
        WeightStager<false> wst;
        wst.issue(W, H, 0, N, H);
        wst.commit(sW, ldw, N, H);
        zero_pads(sA, 2 * RB, ldw, H);
    }
    if (producer) {
        if (first < nrb) park(first, sA);
        if (first + stride < nrb) issue(first + stride);
    }
    __syncthreads();
    int wrow0[4];
    const int nt = wave_tiles<4>(wrow0, (N + 15) >> 4);
    int buf = 0, prev_rb = -1;
    for (int rb = first; rb < nrb; rb += stride) {
        const int nxt = rb + stride;
        if (producer) {
            if (prev_rb >= 0) epilogue(prev_rb, sA + (buf ^ 1) * RB * ldw, sP + (buf ^ 1) * RB * ldp);
            if (prev_rb >= 0) issue_epi(rb);
            if (nxt < nrb) {
                park(nxt, sA + (buf ^ 1) * RB * ldw);
                if (nxt + stride < nrb) issue(nxt + stride);
            }
        } else {
            f32x4 acc[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
            contract<4>(acc, wrow0, sA + buf * RB * ldw, ldw, sW, ldw, H);
            spill_tiles<4>(acc, wrow0, nt, sP + buf * RB * ldp, ldp);
        }
        __syncthreads();
        prev_rb = rb;
        buf ^= 1;
    }
    if (producer && prev_rb >= 0) epilogue(prev_rb, sA + (buf ^ 1) * RB * ldw, sP + (buf ^ 1) * RB * ldp);
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
                                                            const float* __restrict__ Whh, const float* __restrict__ bsum, const float* __restrict__ bsum2,
                                                            float* __restrict__ gates, float* __restrict__ h_out,
                                                            float* __restrict__ c_out, int R, int H, int ldh) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int K = h ? 2 * H : H;
    const int ldw = lds_stride(K), lde = UB + 4;
    float* sW = smem;                                  // [4 gates][UB][ldw]
    float* sA = sW + 4 * UB * ldw;                     // [2][16][ldw]   rows [q | h]
    float* sE = sA + 2 * RB * ldw;                     // [4 gates][16][lde] pre-activations
    const int u0 = blockIdx.y * UB;
    const int nu = min(UB, H - u0);
    // weights: LDS row (gate, ul) <- [W_ih | W_hh] row gate * H + u0 + ul; rows of missing units are zero.  All loads
    // of the slice (<= 13 + 13 16-byte slots per thread) go out at once, the first row block's right behind them.
    const int H4 = H >> 2;
    const float rH4 = 1.0f / (float)H4;
    constexpr int NW1 = 13;                            // 4 gates x 32 units x (H / 4 <= 25) / 256 threads
    float4 vi[NW1], vh[NW1];
    {
        const int total = 4 * UB * H4;
#pragma unroll
        for (int e = 0; e < NW1; ++e) {
            const int i = threadIdx.x + 256 * e;
            const int rowl = qdiv(i, rH4), k = (i - rowl * H4) * 4;
            const int gate = rowl / UB, ul = rowl - gate * UB;
            const bool ok = i < total && ul < nu;
            const int64_t src = (int64_t)(gate * H + u0 + ul) * H + k;
            vi[e] = ok ? ld4(Wih + src) : zero4();
            vh[e] = (ok && h) ? ld4(Whh + src) : zero4();
        }
    }
    const int nrb = (R + RB - 1) / RB;
    const int w = threadIdx.x >> 6;
    const int wrow0[2] = {w * UB, w * UB + 16};        // gate = wave, both unit tiles (units past nu: zero weight rows)
    constexpr int NS = 2;
    float4 rq[NS], rh[NS];
    auto issue = [&](int rb) {
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int i = threadIdx.x + 256 * s;
            const int r = qdiv(i, rH4), k = (i - r * H4) * 4, row = rb * RB + r;
            const bool ok = i < RB * H4 && row < R;
            rq[s] = ok ? ld4(q + (int64_t)row * H + k) : zero4();
            rh[s] = (ok && h) ? ld4(h + (int64_t)row * ldh + k) : zero4();
        }
    };
    auto park = [&](float* dst) {
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int i = threadIdx.x + 256 * s;
            if (i >= RB * H4) continue;
            const int r = qdiv(i, rH4), k = (i - r * H4) * 4;
            st4(dst + r * ldw + k, rq[s]);
            if (h) st4(dst + r * ldw + H + k, rh[s]);
        }
    };
    if ((int)blockIdx.x < nrb) issue(blockIdx.x);
    {
        const int total = 4 * UB * H4;
#pragma unroll
        for (int e = 0; e < NW1; ++e) {
            const int i = threadIdx.x + 256 * e;
            if (i >= total) continue;
            const int rowl = qdiv(i, rH4), k = (i - rowl * H4) * 4;
            st4(sW + rowl * ldw + k, vi[e]);                       // rowl = gate * UB + ul
            if (h) st4(sW + rowl * ldw + H + k, vh[e]);
        }
        zero_pads(sW, 4 * UB, ldw, K);
    }
    zero_pads(sA, 2 * RB, ldw, K);
    if ((int)blockIdx.x < nrb) park(sA);
    // the cell math: every thread owns one (row, 2 units) group of the 16 x 32 block
    const int er = threadIdx.x >> 4, eu = (threadIdx.x & 15) * 2;
    const bool ethread = eu < nu;                      // nu is a multiple of 4
    float2 bi = make_float2(0.f, 0.f), bf = bi, bg = bi, bo = bi;
    if (ethread) {
        bi = *reinterpret_cast<const float2*>(bsum + u0 + eu); bf = *reinterpret_cast<const float2*>(bsum + H + u0 + eu);
        bg = *reinterpret_cast<const float2*>(bsum + 2 * H + u0 + eu); bo = *reinterpret_cast<const float2*>(bsum + 3 * H + u0 + eu);
        if (bsum2) {                                   // b_ih + b_hh formed here: no separate add launch per forward
            const float2 ci = *reinterpret_cast<const float2*>(bsum2 + u0 + eu), cf = *reinterpret_cast<const float2*>(bsum2 + H + u0 + eu);
            const float2 cg = *reinterpret_cast<const float2*>(bsum2 + 2 * H + u0 + eu), co = *reinterpret_cast<const float2*>(bsum2 + 3 * H + u0 + eu);
            bi.x += ci.x; bi.y += ci.y; bf.x += cf.x; bf.y += cf.y; bg.x += cg.x; bg.y += cg.y; bo.x += co.x; bo.y += co.y;
        }
    }
    __syncthreads();
    int buf = 0;
    FOR_ROW_BLOCKS(rb, nrb) {
        const int nxt = rb + gridDim.x;
        if (nxt < nrb) issue(nxt);
        const int erow = rb * RB + er;
        const float2 cp = (c && ethread && erow < R) ? *reinterpret_cast<const float2*>(c + (int64_t)erow * H + u0 + eu)
                                                     : make_float2(0.f, 0.f);
        f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
        contract<2>(acc, wrow0, sA + buf * RB * ldw, ldw, sW, ldw, K);
        {
            const int lane = threadIdx.x & 63, fi = lane & 15, g = lane >> 4;
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) sE[(w * RB + frag_row(4 * g + r)) * lde + 16 * t + frag_row(fi)] = acc[t][r];
        }
        __syncthreads();
        if (ethread && erow < R) {
            const float2 pi = *reinterpret_cast<const float2*>(sE + (0 * RB + er) * lde + eu);
            const float2 pf = *reinterpret_cast<const float2*>(sE + (1 * RB + er) * lde + eu);
            const float2 pg = *reinterpret_cast<const float2*>(sE + (2 * RB + er) * lde + eu);
            const float2 po = *reinterpret_cast<const float2*>(sE + (3 * RB + er) * lde + eu);
            float2 gi, gf, gg, go, cn, hn;
#define K8F(F_)                                                                 \
    {                                                                           \
        gi.F_ = sigm(pi.F_ + bi.F_); gf.F_ = sigm(pf.F_ + bf.F_); gg.F_ = tanhf_(pg.F_ + bg.F_); go.F_ = sigm(po.F_ + bo.F_); \
        cn.F_ = gf.F_ * cp.F_ + gi.F_ * gg.F_;                                  \
        hn.F_ = go.F_ * tanhf_(cn.F_);                                          \
    }
            K8F(x) K8F(y)
#undef K8F
            float* gr = gates + (int64_t)erow * 4 * H + u0 + eu;
            *reinterpret_cast<float2*>(gr) = gi; *reinterpret_cast<float2*>(gr + H) = gf;
            *reinterpret_cast<float2*>(gr + 2 * H) = gg; *reinterpret_cast<float2*>(gr + 3 * H) = go;
            *reinterpret_cast<float2*>(c_out + (int64_t)erow * H + u0 + eu) = cn;
            *reinterpret_cast<float2*>(h_out + (int64_t)erow * ldh + u0 + eu) = hn;
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
constexpr int GATE_BWD_WIDE_NKC = 26;     // K-chunks of 16 a consumer wave's register tile covers (lstm_gate_bwd_ws_kernel<., true>)

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
    const int ldw = lds_stride(K), ldp = CBW + 4;
    float* sW = smem;                                  // [CBW][ldw]
    float* sA = sW + CBW * ldw;                        // [2][16][ldw]   rows dG; the consumed buffer doubles as sP
    const int nb = (H + CBW - 1) / CBW;
    const bool is_dh = (int)blockIdx.y >= nb;
    const int n0 = (is_dh ? blockIdx.y - nb : blockIdx.y) * CBW;
    const int ncols = min(CBW, H - n0);
    WeightStager<true> wst;
    wst.issue(is_dh ? Whh : Wih, H, n0, ncols, K);
    const int nrb = (R + RB - 1) / RB;
    const int w = threadIdx.x >> 6;
    const int wrow0[1] = {16 * w < ncols ? 16 * w : 0};
    const int nt = 16 * w < ncols ? 1 : 0;
    const bool writer = blockIdx.y == 0;
    const int H4 = H >> 2;
    const float rH4 = 1.0f / (float)H4;
    constexpr int NS = 2;                              // 16 * H / 4 / 256 <= 2 for H <= 128
    float4 rgi[NS], rgf[NS], rgg[NS], rgo[NS], rcn[NS], rcp[NS], rdh[NS], rdh2[NS], rdc[NS];
    auto issue = [&](int rb) {
#pragma unroll
        for (int s_ = 0; s_ < NS; ++s_) {
            const int i = threadIdx.x + 256 * s_;
            const int r = qdiv(i, rH4), u = (i - r * H4) * 4, row = rb * RB + r;
            const bool ok = i < RB * H4 && row < R;
            const int64_t o = (int64_t)row * H + u;
            const float* gr = gates + (int64_t)row * 4 * H + u;
            rgi[s_] = ok ? ld4(gr) : zero4();
            rgf[s_] = ok ? ld4(gr + H) : zero4();
            rgg[s_] = ok ? ld4(gr + 2 * H) : zero4();
            rgo[s_] = ok ? ld4(gr + 3 * H) : zero4();
            rcn[s_] = ok ? ld4(c_new + o) : zero4();
            rcp[s_] = (ok && c_prev) ? ld4(c_prev + o) : zero4();
            rdh[s_] = (ok && dh_a) ? ld4(dh_a + o) : zero4();
            rdh2[s_] = (ok && dh_b) ? ld4(dh_b + o) : zero4();          // (added at use: an add here would wait for both loads)
            rdc[s_] = (ok && dc_next) ? ld4(dc_next + o) : zero4();
        }
    };
    auto park = [&](int rb, float* dst) {
#pragma unroll
        for (int s_ = 0; s_ < NS; ++s_) {
            const int i = threadIdx.x + 256 * s_;
            if (i >= RB * H4) continue;
            const int r = qdiv(i, rH4), u = (i - r * H4) * 4, row = rb * RB + r;
            float4 di, df, dg, dO, dcp;
#define GB1(F)                                                                      \
    {                                                                               \
        const float gi = rgi[s_].F, gf = rgf[s_].F, gg = rgg[s_].F, go = rgo[s_].F; \
        const float tc = tanhf_(rcn[s_].F);                                         \
        const float dhv = rdh[s_].F + rdh2[s_].F;                                   \
        const float dc = rdc[s_].F + dhv * go * (1.0f - tc * tc);                   \
        dO.F = dhv * tc * go * (1.0f - go);                                         \
        di.F = dc * gg * gi * (1.0f - gi);                                          \
        df.F = dc * rcp[s_].F * gf * (1.0f - gf);                                   \
        dg.F = dc * gi * (1.0f - gg * gg);                                          \
        dcp.F = dc * gf;                                                            \
    }
            GB1(x) GB1(y) GB1(z) GB1(w)
#undef GB1
            float* sp = dst + r * ldw + u;               // rows past R carry zeros (their loads were zeroed)
            st4(sp, di); st4(sp + H, df); st4(sp + 2 * H, dg); st4(sp + 3 * H, dO);
            if (writer && row < R) {
                float* d = dG + (int64_t)row * 4 * H + u;
                st4(d, di); st4(d + H, df); st4(d + 2 * H, dg); st4(d + 3 * H, dO);
                if (has_h) st4(dc_prev + (int64_t)row * H + u, dcp);
            }
        }
    };
    if ((int)blockIdx.x < nrb) issue(blockIdx.x);
    wst.commit(sW, ldw, ncols, K);
    if (ncols < CBW) {                                 // weight rows of missing columns: zeros (their results are dropped)
        for (int i = threadIdx.x; i < (CBW - ncols) * (ldw >> 2); i += 256) st4(sW + ncols * ldw + 4 * i, zero4());
    }
    zero_pads(sA, 2 * RB, ldw, K);
    if ((int)blockIdx.x < nrb) park(blockIdx.x, sA);
    __syncthreads();
    // epilogue mapping: thread -> (row, 4 columns) of the 16 x 64 block
    const int er = threadIdx.x >> 4, en = (threadIdx.x & 15) * 4;
    int buf = 0;
    FOR_ROW_BLOCKS(rb, nrb) {
        const int nxt = rb + gridDim.x;
        if (nxt < nrb) issue(nxt);
        const int erow = rb * RB + er;
        const bool eok = en < ncols && erow < R;
        const float4 rv = (!is_dh && dres && eok) ? ld4(dres + (int64_t)erow * lddres + n0 + en) : zero4();
        f32x4 acc[1] = {{0.f, 0.f, 0.f, 0.f}};
        float* Acur = sA + buf * RB * ldw;
        contract<1>(acc, wrow0, Acur, ldw, sW, ldw, K);
        __syncthreads();                               // every wave is done reading A(cur): its space becomes sP
        spill_tiles<1>(acc, wrow0, nt, Acur, ldp);
        __syncthreads();
        if (eok) {
            const float4 v = add4(ld4(Acur + er * ldp + en), rv);
            st4((is_dh ? dh_prev : dq) + (int64_t)erow * H + n0 + en, v);
        }
        if (nxt < nrb) park(nxt, sA + (buf ^ 1) * RB * ldw);
        __syncthreads();
        buf ^= 1;
    }
}

// ------------------------------------------------------------------------------------------------------------------
// K8 backward, producer / consumer form.  Timing ablations of the four-wave kernel above (tools/ablate_gate_bwd.py, 24 576
// rows: base 12 us + operand loads 32 + gate math 22 + MFMAs 30 + result stores 29 = the 105-110 us measured; a 32-row-block
// variant that reuses every weight fragment for two row tiles changed that by 5 %) show the phases of a row block simply add up: four waves that all stage, then all contract, then all store leave the matrix cores idle
// while the memory system works and vice versa.  Here the workgroup has EIGHT waves: waves 4-7 are producers (operand
// loads of block i + 2 in flight, gate math of block i + 1, dG / dc written out, A(i + 1) parked in the other LDS buffer),
// waves 0-3 are consumers (contraction of A(i) against the weight slice, results stored straight from the accumulators);
// one barrier per block hands the buffers over.  Every SIMD then holds one wave of each kind, so the consumer's MFMAs
// run under the producer's loads, VALU work and stores.
// WIDE (round 5, 64 < H <= 104): ONE column block per product (dq | dh) instead of two -- the producer side (operand loads, gate
// math, dG / dc stores: 2/3 of the launch at 98 304 rows) is done twice per row instead of four times.  The 128-column weight
// slice does not fit the LDS (206 KB), so consumer wave w contracts its SECOND tile (columns 64 + 16 w ..) against weight
// fragments held in REGISTERS (<= 26 float4 per lane: K = 4 H <= 416, loaded once per workgroup straight from the parameter); the first tile
// comes from the LDS slice as before, both share the A fragments.
// ------------------------------------------------------------------------------------------------------------------
template <int ABL, bool WIDE = false>
__global__ __launch_bounds__(512) void lstm_gate_bwd_ws_kernel(const float* __restrict__ gates, const float* __restrict__ c_prev,
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
    float* sA = sW + CBW * ldw;                        // [2][16][ldw]   rows dG
    const int nb = WIDE ? 1 : (H + CBW - 1) / CBW;
    const bool is_dh = (int)blockIdx.y >= nb;
    const int n0 = WIDE ? 0 : (is_dh ? blockIdx.y - nb : blockIdx.y) * CBW;
    const int ncols = min(CBW, H - n0);                // columns of the LDS slice (WIDE: + the register tiles behind them)
    const int nrb = (R + RB - 1) / RB;
    const int tid = threadIdx.x & 255;                 // index inside the half (consumer / producer) of the workgroup
    const bool producer = threadIdx.x >= 256;
    // dG / dc leave through ALL column blocks of a row group, a gate slice each (one writer block per row group put all
    // 49 MB through a quarter of the compute units: 37 us of stores at 24 576 rows): gate j by block j % gridDim.y, dc by the last
    const int nyb = gridDim.y, yb = blockIdx.y;
    const bool wr_i = (0 % nyb) == yb, wr_f = (1 % nyb) == yb, wr_g = (2 % nyb) == yb, wr_o = (3 % nyb) == yb, wr_c = yb == nyb - 1;
    const int H4 = H >> 2;
    const float rH4 = 1.0f / (float)H4;
    // ---- prologue: all 512 threads park the weight slice (k-contiguous rows sW[n][k] from the (K, H) parameter); all of a
    // thread's loads (<= 13 slots) go out before its first LDS store
    {
        const float* Wsrc = is_dh ? Whh : Wih;
        const int inner = ncols >> 2;                  // 16-byte slots per source row (k)
        const int total = K * inner;
        constexpr int NW2 = 13;                        // 400 x 16 / 512
        float4 v[NW2];
#pragma unroll
        for (int e = 0; e < NW2; ++e) {
            const int i0 = threadIdx.x + 512 * e, i = i0 < total ? i0 : total - 1;
            const int r = i / inner, j = i - r * inner;
            v[e] = ld4(Wsrc + (int64_t)r * H + n0 + 4 * j);
        }
#pragma unroll
        for (int e = 0; e < NW2; ++e) {
            const int i = threadIdx.x + 512 * e;
            if (i >= total) continue;
            const int r = i / inner, j = i - r * inner;
            float* d = sW + (4 * j) * ldw + r;
            d[0] = v[e].x; d[ldw] = v[e].y; d[2 * ldw] = v[e].z; d[3 * ldw] = v[e].w;
        }
        const int np4 = (ldw - K) >> 2;                // zero pads of the weight rows and of both A buffers; missing columns
        const float rnp4 = 1.0f / (float)(np4 > 0 ? np4 : 1);
        for (int i = threadIdx.x; i < ncols * np4; i += 512) { const int r = qdiv(i, rnp4), j = i - r * np4; st4(sW + r * ldw + K + 4 * j, zero4()); }
        for (int i = threadIdx.x; i < (CBW - ncols) * (ldw >> 2); i += 512) st4(sW + ncols * ldw + 4 * i, zero4());
        for (int i = threadIdx.x; i < 2 * RB * np4; i += 512) { const int r = qdiv(i, rnp4), j = i - r * np4; st4(sA + r * ldw + K + 4 * j, zero4()); }
    }
    constexpr int NS = 2;                              // 16 * H / 4 / 256 <= 2 for H <= 128
    float4 rgi[NS], rgf[NS], rgg[NS], rgo[NS], rcn[NS], rcp[NS], rdh[NS], rdh2[NS], rdc[NS];
    auto issue = [&](int rb) {
#pragma unroll
        for (int s_ = 0; s_ < NS; ++s_) {
            const int i = tid + 256 * s_;
            const int r = qdiv(i, rH4), u = (i - r * H4) * 4, row = rb * RB + r;
            const bool ok = !(ABL & 2) && i < RB * H4 && row < R;
            const int64_t o = (int64_t)row * H + u;
            const float* gr = gates + (int64_t)row * 4 * H + u;
            rgi[s_] = ok ? ld4(gr) : zero4();
            rgf[s_] = ok ? ld4(gr + H) : zero4();
            rgg[s_] = ok ? ld4(gr + 2 * H) : zero4();
            rgo[s_] = ok ? ld4(gr + 3 * H) : zero4();
            rcn[s_] = ok ? ld4(c_new + o) : zero4();
            rcp[s_] = (ok && c_prev) ? ld4(c_prev + o) : zero4();
            rdh[s_] = (ok && dh_a) ? ld4(dh_a + o) : zero4();
            rdh2[s_] = (ok && dh_b) ? ld4(dh_b + o) : zero4();          // (added at use: an add here would wait for both loads)
            rdc[s_] = (ok && dc_next) ? ld4(dc_next + o) : zero4();
        }
    };
    // gate math of the block whose operands sit in the registers: results to LDS (A rows) and kept for flush()
    float4 odi[NS], odf[NS], odg[NS], odo[NS], odc[NS];
    auto park = [&](float* dst) {
#pragma unroll
        for (int s_ = 0; s_ < NS; ++s_) {
            const int i = tid + 256 * s_;
            float4 di, df, dg, dO, dcp;
#define GB1(F)                                                                      \
    {                                                                               \
        const float gi = rgi[s_].F, gf = rgf[s_].F, gg = rgg[s_].F, go = rgo[s_].F; \
        const float tc = tanhf_(rcn[s_].F);                                         \
        const float dhv = rdh[s_].F + rdh2[s_].F;                                   \
        const float dc = rdc[s_].F + dhv * go * (1.0f - tc * tc);                   \
        dO.F = dhv * tc * go * (1.0f - go);                                         \
        di.F = dc * gg * gi * (1.0f - gi);                                          \
        df.F = dc * rcp[s_].F * gf * (1.0f - gf);                                   \
        dg.F = dc * gi * (1.0f - gg * gg);                                          \
        dcp.F = dc * gf;                                                            \
    }
            if (ABL & 4) { di = rgi[s_]; df = rgf[s_]; dg = rgg[s_]; dO = rgo[s_]; dcp = rcn[s_]; }
            else { GB1(x) GB1(y) GB1(z) GB1(w) }
#undef GB1
            odi[s_] = di; odf[s_] = df; odg[s_] = dg; odo[s_] = dO; odc[s_] = dcp;
            if (i >= RB * H4) continue;
            const int r = qdiv(i, rH4), u = (i - r * H4) * 4;
            float* sp = dst + r * ldw + u;               // rows past R carry zeros (their loads were zeroed)
            st4(sp, di); st4(sp + H, df); st4(sp + 2 * H, dg); st4(sp + 3 * H, dO);
        }
    };
    // ... and out to memory.  Called AFTER the next block's operand loads are issued: the memory counter is in order, so a
    // wait for loads that were issued behind these stores would also wait for the stores' write acknowledgements
    auto flush = [&](int rb) {
        if (ABL & 8) return;
#pragma unroll
        for (int s_ = 0; s_ < NS; ++s_) {
            const int i = tid + 256 * s_;
            if (i >= RB * H4) continue;
            const int r = qdiv(i, rH4), u = (i - r * H4) * 4, row = rb * RB + r;
            if (row >= R) continue;
            float* d = dG + (int64_t)row * 4 * H + u;
            if (wr_i) st4_nt(d, odi[s_]);
            if (wr_f) st4_nt(d + H, odf[s_]);
            if (wr_g) st4_nt(d + 2 * H, odg[s_]);
            if (wr_o) st4_nt(d + 3 * H, odo[s_]);
            if (has_h && wr_c) st4(dc_prev + (int64_t)row * H + u, odc[s_]);
        }
    };
    const int first = blockIdx.x, stride = gridDim.x;
    if (producer) {
        if (first < nrb) { issue(first); park(sA); }
        if (first + stride < nrb) issue(first + stride);
        if (first < nrb) flush(first);
    }
    __syncthreads();
    // consumer constants: wave w contracts the 16-column tile w of the slice; lane (fi, g) holds column fi, rows 4 g .. + 3
    const int w = tid >> 6, lane = tid & 63, fi = lane & 15, g = lane >> 4;
    const int wrow0[1] = {16 * w < ncols ? 16 * w : 0};
    const int ecol = n0 + 16 * w + frag_row(fi);        // (output column / rows of this lane's accumulator elements: frag_row)
    const bool ecol_ok = (16 * w + frag_row(fi)) < ncols;
    float* const outp = is_dh ? dh_prev : dq;
    // WIDE: the second tile's weight fragments, lane (fi, g): column ecol2, k = 16 kc + 4 frag_unit(g) .. + 3 (K = 4 H <= 512)
    constexpr int NKC = WIDE ? GATE_BWD_WIDE_NKC : 1;
    const int ecol2 = CBW + 16 * w + frag_row(fi);
    const bool ecol2_ok = WIDE && ecol2 < H;
    const bool tile2 = WIDE && (CBW + 16 * w) < H;      // (wave-uniform: the whole tile lies past H)
    float4 wreg[NKC];
    if (WIDE && !producer) {
        const float* Wsrc = is_dh ? Whh : Wih;
        const int cc = ecol2_ok ? ecol2 : 0;
#pragma unroll
        for (int kc = 0; kc < NKC; ++kc) {
            const int k0 = 16 * kc + 4 * frag_unit(g);
            float v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = (ecol2_ok && k0 + j < K) ? Wsrc[(int64_t)(k0 + j) * H + cc] : 0.f;
            wreg[kc] = make_float4(v[0], v[1], v[2], v[3]);
        }
    }
    int buf = 0;
    // (one loop per role: inside a common loop the register allocation is the SUM of the producers' operand sets and the
    // consumers' weight fragments; the barrier count is the same on both sides)
    if (producer) {
        for (int rb = first; rb < nrb; rb += stride) {
            const int nxt = rb + stride;
            // block nxt's operands were requested one iteration ago: gate math, park, then request block nxt + stride
            if (nxt < nrb) {
                park(sA + (buf ^ 1) * RB * ldw);
                if (nxt + stride < nrb) issue(nxt + stride);
                flush(nxt);
            }
            __syncthreads();                           // A(nxt) is parked; A(rb)'s buffer is free
            buf ^= 1;
        }
        return;
    }
    for (int rb = first; rb < nrb; rb += stride) {
        {
            float rv[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int erow = rb * RB + frag_row(4 * g + r);
                rv[r] = (!is_dh && dres && ecol_ok && erow < R) ? dres[(int64_t)erow * lddres + ecol] : 0.f;
            }
            f32x4 acc[1] = {{0.f, 0.f, 0.f, 0.f}};
            if constexpr (WIDE) {
                float rv2[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int erow = rb * RB + frag_row(4 * g + r);
                    rv2[r] = (!is_dh && dres && ecol2_ok && erow < R) ? dres[(int64_t)erow * lddres + ecol2] : 0.f;
                }
                f32x4 acc2 = {0.f, 0.f, 0.f, 0.f};
                // both tiles against the same A fragments; K walked in whole chunks of 16 (the LDS rows are zero-padded, the
                // register fragments past K hold zeros); fragments of chunk kc + 1 are fetched before the MFMAs of chunk kc
                const float* ap = sA + buf * RB * ldw + frag_row(fi) * ldw + 4 * frag_unit(g);
                const float* bp = sW + (wrow0[0] + frag_row(fi)) * ldw + 4 * frag_unit(g);
                const int nk = (K + 15) >> 4;
                float4 a_n = ld4(ap), b_n = ld4(bp);
#pragma unroll
                for (int kc = 0; kc < NKC; ++kc) {
                    if (kc < nk) {                                  // uniform
                        const float4 a_c = a_n, b_c = b_n;
                        if (kc + 1 < nk) { a_n = ld4(ap + 16 * (kc + 1)); b_n = ld4(bp + 16 * (kc + 1)); }
                        const float av[4] = {a_c.x, a_c.y, a_c.z, a_c.w}, bv[4] = {b_c.x, b_c.y, b_c.z, b_c.w};
                        const float cv[4] = {wreg[kc].x, wreg[kc].y, wreg[kc].z, wreg[kc].w};
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j], bv[j], acc[0], 0, 0, 0);
                            if (tile2) acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j], cv[j], acc2, 0, 0, 0);
                        }
                    }
                }
                if (ecol2_ok && !(ABL & 8)) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int erow = rb * RB + frag_row(4 * g + r);
                        if (erow < R) outp[(int64_t)erow * H + ecol2] = acc2[r] + rv2[r];
                    }
                }
            } else {
                if (!(ABL & 1)) contract<1>(acc, wrow0, sA + buf * RB * ldw, ldw, sW, ldw, K);
                else acc[0][0] = sA[buf * RB * ldw + fi * ldw + g];
            }
            if (ecol_ok && !(ABL & 8)) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int erow = rb * RB + frag_row(4 * g + r);
                    if (erow < R) outp[(int64_t)erow * H + ecol] = acc[0][r] + rv[r];
                }
            }
        }
        __syncthreads();                               // A(nxt) is parked; A(rb)'s buffer is free
        buf ^= 1;
    }
}

// K8 forward, producer / consumer form (see the backward above): waves 0-3 contract A(i) (wave = gate, both unit tiles) and
// leave the pre-activations in sE[i & 1]; waves 4-7 run the cell math of block i - 1 from the other sE buffer and store
// gates / c' / h', park A(i + 1) and keep block i + 2's rows in flight.  One barrier per block.
__global__ __launch_bounds__(512) void lstm_gate_fwd_ws_kernel(const float* __restrict__ q, const float* __restrict__ h,
                                                               const float* __restrict__ c, const float* __restrict__ Wih,
                                                               const float* __restrict__ Whh, const float* __restrict__ bsum, const float* __restrict__ bsum2,
                                                               float* __restrict__ gates, float* __restrict__ h_out,
                                                               float* __restrict__ c_out, int R, int H, int ldh) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int K = h ? 2 * H : H;
    const int ldw = lds_stride(K), lde = UB + 4;
    float* sW = smem;                                  // [4 gates][UB][ldw]
    float* sA = sW + 4 * UB * ldw;                     // [2][16][ldw]   rows [q | h]
    float* sE = sA + 2 * RB * ldw;                     // [2][4 gates][16][lde] pre-activations
    const int u0 = blockIdx.y * UB;
    const int nu = min(UB, H - u0);
    const int H4 = H >> 2;
    const float rH4 = 1.0f / (float)H4;
    const int tid = threadIdx.x & 255;
    const bool producer = threadIdx.x >= 256;
    // ---- prologue (all 512 threads): weight rows (gate, ul) <- [W_ih | W_hh] row gate * H + u0 + ul; missing units: zeros.
    // Every load of the slice goes out before the first LDS store (7 + 7 slots per thread, unconditional loads from clamped
    // addresses: one memory round trip for the slice, not one per slot)
    {
        const int total = 4 * UB * H4;
        constexpr int NW2 = 7;                             // 4 gates x 32 units x (H / 4 <= 25) / 512 threads
        float4 vi[NW2], vh[NW2];
#pragma unroll
        for (int e = 0; e < NW2; ++e) {
            const int i0 = threadIdx.x + 512 * e, i = i0 < total ? i0 : total - 1;
            const int rowl = qdiv(i, rH4), k = (i - rowl * H4) * 4;
            const int gate = rowl / UB, ul = rowl - gate * UB;
            const int64_t src = (int64_t)(gate * H + u0 + (ul < nu ? ul : nu - 1)) * H + k;
            vi[e] = ld4(Wih + src);
            vh[e] = h ? ld4(Whh + src) : zero4();
        }
#pragma unroll
        for (int e = 0; e < NW2; ++e) {
            const int i = threadIdx.x + 512 * e;
            if (i >= total) continue;
            const int rowl = qdiv(i, rH4), k = (i - rowl * H4) * 4;
            const int ul = rowl - (rowl / UB) * UB;
            st4(sW + rowl * ldw + k, ul < nu ? vi[e] : zero4());
            if (h) st4(sW + rowl * ldw + H + k, ul < nu ? vh[e] : zero4());
        }
        const int np4 = (ldw - K) >> 2;
        const float rnp4 = 1.0f / (float)(np4 > 0 ? np4 : 1);
        for (int i = threadIdx.x; i < 4 * UB * np4; i += 512) { const int r = qdiv(i, rnp4), j = i - r * np4; st4(sW + r * ldw + K + 4 * j, zero4()); }
        for (int i = threadIdx.x; i < 2 * RB * np4; i += 512) { const int r = qdiv(i, rnp4), j = i - r * np4; st4(sA + r * ldw + K + 4 * j, zero4()); }
    }
    const int nrb = (R + RB - 1) / RB;
    constexpr int NS = 2;
    float4 rq[NS], rh[NS];
    auto issue = [&](int rb) {
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int i = tid + 256 * s;
            const int r = qdiv(i, rH4), k = (i - r * H4) * 4, row = rb * RB + r;
            const bool ok = i < RB * H4 && row < R;
            rq[s] = ok ? ld4(q + (int64_t)row * H + k) : zero4();
            rh[s] = (ok && h) ? ld4(h + (int64_t)row * ldh + k) : zero4();
        }
    };
    auto park = [&](float* dst) {
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int i = tid + 256 * s;
            if (i >= RB * H4) continue;
            const int r = qdiv(i, rH4), k = (i - r * H4) * 4;
            st4(dst + r * ldw + k, rq[s]);
            if (h) st4(dst + r * ldw + H + k, rh[s]);
        }
    };
    // the cell math (producers): every thread owns one (row, 2 units) group of the 16 x 32 block
    const int er = tid >> 4, eu = (tid & 15) * 2;
    const bool ethread = eu < nu;                      // nu is a multiple of 4
    float2 bi = make_float2(0.f, 0.f), bf = bi, bg = bi, bo = bi;
    if (producer && ethread) {
        bi = *reinterpret_cast<const float2*>(bsum + u0 + eu); bf = *reinterpret_cast<const float2*>(bsum + H + u0 + eu);
        bg = *reinterpret_cast<const float2*>(bsum + 2 * H + u0 + eu); bo = *reinterpret_cast<const float2*>(bsum + 3 * H + u0 + eu);
        if (bsum2) {
            const float2 ci = *reinterpret_cast<const float2*>(bsum2 + u0 + eu), cf = *reinterpret_cast<const float2*>(bsum2 + H + u0 + eu);
            const float2 cg = *reinterpret_cast<const float2*>(bsum2 + 2 * H + u0 + eu), co = *reinterpret_cast<const float2*>(bsum2 + 3 * H + u0 + eu);
            bi.x += ci.x; bi.y += ci.y; bf.x += cf.x; bf.y += cf.y; bg.x += cg.x; bg.y += cg.y; bo.x += co.x; bo.y += co.y;
        }
    }
    auto cell = [&](int rb, const float* E, float2 cp) {
        const int erow = rb * RB + er;
        if (!(ethread && erow < R)) return;
        const float2 pi = *reinterpret_cast<const float2*>(E + (0 * RB + er) * lde + eu);
        const float2 pf = *reinterpret_cast<const float2*>(E + (1 * RB + er) * lde + eu);
        const float2 pg = *reinterpret_cast<const float2*>(E + (2 * RB + er) * lde + eu);
        const float2 po = *reinterpret_cast<const float2*>(E + (3 * RB + er) * lde + eu);
        float2 gi, gf, gg, go, cn, hn;
#define K8F(F_)                                                                 \
    {                                                                           \
        gi.F_ = sigm(pi.F_ + bi.F_); gf.F_ = sigm(pf.F_ + bf.F_); gg.F_ = tanhf_(pg.F_ + bg.F_); go.F_ = sigm(po.F_ + bo.F_); \
        cn.F_ = gf.F_ * cp.F_ + gi.F_ * gg.F_;                                  \
        hn.F_ = go.F_ * tanhf_(cn.F_);                                          \
    }
        K8F(x) K8F(y)
#undef K8F
        // the gate activations are read again only by the backward pass: around the L2 (nontemporal), h' and c' stay cached
        float* gr = gates + (int64_t)erow * 4 * H + u0 + eu;
        nt_store2(gr, gi); nt_store2(gr + H, gf); nt_store2(gr + 2 * H, gg); nt_store2(gr + 3 * H, go);
        *reinterpret_cast<float2*>(c_out + (int64_t)erow * H + u0 + eu) = cn;
        *reinterpret_cast<float2*>(h_out + (int64_t)erow * ldh + u0 + eu) = hn;
    };
    auto load_c = [&](int rb) {
        const int erow = rb * RB + er;
        return (c && ethread && erow < R) ? *reinterpret_cast<const float2*>(c + (int64_t)erow * H + u0 + eu) : make_float2(0.f, 0.f);
    };
    const int first = blockIdx.x, stride = gridDim.x;
    float2 cp_prev = make_float2(0.f, 0.f);            // c rows of the block whose cell math runs next
    if (producer) {
        if (first < nrb) { issue(first); park(sA); cp_prev = load_c(first); }
        if (first + stride < nrb) issue(first + stride);
    }
    __syncthreads();
    const int w = tid >> 6;
    const int wrow0[2] = {w * UB, w * UB + 16};        // consumer wave = gate, both unit tiles
    int buf = 0, prev_rb = -1;
    for (int rb = first; rb < nrb; rb += stride) {
        const int nxt = rb + stride;
        if (producer) {
            if (prev_rb >= 0) cell(prev_rb, sE + (buf ^ 1) * 4 * RB * lde, cp_prev);
            cp_prev = load_c(rb);
            if (nxt < nrb) {
                park(sA + (buf ^ 1) * RB * ldw);
                if (nxt + stride < nrb) issue(nxt + stride);
            }
        } else {
            f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
            contract<2>(acc, wrow0, sA + buf * RB * ldw, ldw, sW, ldw, K);
            const int lane = tid & 63, fi = lane & 15, g = lane >> 4;
            float* E = sE + buf * 4 * RB * lde;
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) E[(w * RB + frag_row(4 * g + r)) * lde + 16 * t + frag_row(fi)] = acc[t][r];
        }
        __syncthreads();
        prev_rb = rb;
        buf ^= 1;
    }
    if (producer && prev_rb >= 0) cell(prev_rb, sE + (buf ^ 1) * 4 * RB * lde, cp_prev);
}

inline bool bad_dims(int64_t R, int H) { return R <= 0 || H < 4 || (H & 3) || H > 100 || R > (int64_t)1 << 30; }

// producer / consumer forms of the K8 kernels (tuning build: MMDFN_GATE_WS=0|1 forces)
inline bool gate_ws(int R) {
#ifdef MMDFN_TUNING
    if (const char* e = getenv("MMDFN_GATE_WS")) return atoi(e) != 0;
#endif
    return R >= 64;
}

// producer / consumer forms of the K7 kernels (tuning build: MMDFN_LAYER_WS=0|1 forces)
constexpr int LAYER_WS_ROWS = 64;        // measured ahead of the 4-wave kernels from 5 280 rows (cfg2: 15.1 / 14.9 -> 13.4 / 12.3 us) to 98 304 (97 / 101 -> 84 / 94 us), tools/bench_layer.py
inline bool layer_ws(int R) {
#ifdef MMDFN_TUNING
    if (const char* e = getenv("MMDFN_LAYER_WS")) return atoi(e) != 0;
#endif
    return R >= LAYER_WS_ROWS;
}

// bf16-piece form of the K8 forward (lstm_gate_split.hip) from GATE_SPLIT_ROWS rows on (tuning build: MMDFN_GATE_SPLIT=0|1 forces)
constexpr int GATE_SPLIT_ROWS = 16384;
// two-tile consumers of the K8 backward (one column block per product) from this many rows on (tuning build: MMDFN_GATE_BWD_WIDE)
constexpr int GATE_BWD_WIDE_ROWS = 16384;
inline bool gate_split(int R, int H) {
#ifdef MMDFN_TUNING
    if (const char* e = getenv("MMDFN_GATE_SPLIT")) return atoi(e) != 0;
#endif
    return R >= GATE_SPLIT_ROWS && H >= 32;
}

inline int row_groups(int R, int column_blocks) {
    const int nrb = (R + RB - 1) / RB;
    int gq = (256 + column_blocks - 1) / column_blocks;     // ~ one workgroup per CU (LDS holds one weight slice per CU)
    if (gq > nrb) gq = nrb;
    // even out the row blocks per workgroup: with 330 blocks on 256 groups 74 workgroups would do two and set the pace
    const int per = (nrb + gq - 1) / gq;
    gq = (nrb + per - 1) / per;
    return gq < 1 ? 1 : gq;
}

#define LAUNCH_BIG_LDS(kern, grid, lds, stream, ...)                                  \
    do {                                                                              \
        if ((lds) > 156 * 1024) return -1;                                            \
        if (int e_ = mmdfn_allow_big_lds(kern)) return e_;                            \
        hipLaunchKernelGGL(kern, grid, dim3(256), lds, (hipStream_t)(stream), __VA_ARGS__); \
        MMDFN_CHECK_LAUNCH();                                                         \
    } while (0)

}  // namespace

extern "C" int mmdfn_gcn_input_fwd(const float* x, const float* mx, const float* W0, const float* b0, const float* m0,
                                   float* xd, float* h0, float* cur0, int R, int F, int H, int ldxd, float mscale,
                                   void* stream) {
    if (bad_dims(R, H) || F < 4 || (F & 3) || F > 256 || ldxd < F || (ldxd & 3)) return -1;
    const size_t lds = ((size_t)(H + 2 * RB) * lds_stride(F) + RB * (((H + 15) & ~15) + 4)) * sizeof(float);
    LAUNCH_BIG_LDS(gcn_input_fwd_kernel, dim3(row_groups(R, 1)), lds, stream, x, mx, W0, b0, m0, xd, h0, cur0, R, F, H, ldxd,
                   mscale);
    return 0;
}

extern "C" int mmdfn_gcn_input_bwd(const float* dcur0, const float* m0, const float* dh0, const float* h0, const float* W0,
                                   const float* dxd, const float* mx, float* dpre, float* dx, int R, int F, int H, int lddxd,
                                   float mscale, void* stream) {
    if (bad_dims(R, H) || F < 4 || (F & 3) || F > 256 || (dxd && (lddxd < F || (lddxd & 3)))) return -1;
    const size_t lds = ((size_t)(F + 2 * RB) * lds_stride(H) + RB * (((F + 15) & ~15) + 4)) * sizeof(float);
    LAUNCH_BIG_LDS(gcn_input_bwd_kernel, dim3(row_groups(R, 1)), lds, stream, dcur0, m0, dh0, h0, W0, dxd, mx, dpre, dx, R, F, H,
                   lddxd, mscale);
    return 0;
}

static int lstm_gate_fwd_impl(const float* q, const float* h, const float* c, const float* Wih, const float* Whh,
                              const float* bsum, const float* bsum2, float* gates, float* h_out, float* c_out, int R, int H,
                              int ldh, const void* planes, void* stream);

extern "C" int mmdfn_lstm_gate_fwd_ld(const float* q, const float* h, const float* c, const float* Wih, const float* Whh,
                                      const float* bsum, const float* bsum2, float* gates, float* h_out, float* c_out, int R, int H,
                                      int ldh, void* stream) {
    return lstm_gate_fwd_impl(q, h, c, Wih, Whh, bsum, bsum2, gates, h_out, c_out, R, H, ldh, nullptr, stream);
}

// The cell's weights as bf16 piece planes for the many-row form (ABI 14): the cell is shared by all layers of a stack and
// constant inside a step, so the caller cuts it ONCE (mmdfn_lstm_gate_cut_weights into mmdfn_lstm_gate_planes_workspace(H)
// floats) and hands the planes to every layer's launch; launches that do not take the many-row form ignore them.
extern "C" int64_t mmdfn_lstm_gate_planes_workspace(int H) { return (H < 8 || (H & 3)) ? -1 : mmdfn_lstm_gate_planes_floats(H); }
extern "C" int mmdfn_lstm_gate_cut_weights(const float* Wih, const float* Whh, float* planes, int H, void* stream) {
    return mmdfn_launch_lstm_gate_cut(Wih, Whh, planes, H, (hipStream_t)stream);
}
extern "C" int mmdfn_lstm_gate_fwd_pre(const float* q, const float* h, const float* c, const float* Wih, const float* Whh,
                                       const float* bsum, const float* bsum2, float* gates, float* h_out, float* c_out, int R,
                                       int H, int ldh, const float* planes, void* stream) {
    return lstm_gate_fwd_impl(q, h, c, Wih, Whh, bsum, bsum2, gates, h_out, c_out, R, H, ldh, planes, stream);
}
extern "C" int mmdfn_lstm_gate_takes_planes(int R, int H) { return (!bad_dims(R, H) && gate_split(R, H)) ? 1 : 0; }

static int lstm_gate_fwd_impl(const float* q, const float* h, const float* c, const float* Wih, const float* Whh,
                              const float* bsum, const float* bsum2, float* gates, float* h_out, float* c_out, int R, int H,
                              int ldh, const void* planes, void* stream) {
    if (bad_dims(R, H) || (h == nullptr) != (c == nullptr) || ldh < H || (ldh & 3)) return -1;
    if (gate_split(R, H)) {
        // many rows: the contraction on the bf16 matrix path, cell math from the accumulators (lstm_gate_split.hip)
        const int rc = mmdfn_launch_lstm_gate_fwd_split(q, h, c, Wih, Whh, bsum, bsum2, gates, h_out, c_out, R, H,
                                                        ldh, planes, (hipStream_t)stream);
        if (rc != -2) return rc;
    }
    const int ncb = (H + UB - 1) / UB;
    if (gate_ws(R)) {
        const size_t ldsw = ((size_t)(4 * UB + 2 * RB) * lds_stride(h ? 2 * H : H) + 2 * 4 * RB * (UB + 4)) * sizeof(float);
        if (ldsw > 156 * 1024) return -1;
        if (int e_ = mmdfn_allow_big_lds(lstm_gate_fwd_ws_kernel)) return e_;
        hipLaunchKernelGGL(lstm_gate_fwd_ws_kernel, dim3(row_groups(R, ncb), ncb), dim3(512), ldsw, (hipStream_t)stream, q, h, c, Wih,
                           Whh, bsum, bsum2, gates, h_out, c_out, R, H, ldh);
        MMDFN_CHECK_LAUNCH();
        return 0;
    }
    const size_t lds = ((size_t)(4 * UB + 2 * RB) * lds_stride(h ? 2 * H : H) + 4 * RB * (UB + 4)) * sizeof(float);
    LAUNCH_BIG_LDS(lstm_gate_fwd_kernel, dim3(row_groups(R, ncb), ncb), lds, stream, q, h, c, Wih, Whh, bsum, bsum2, gates,
                   h_out, c_out, R, H, ldh);
    return 0;
}

extern "C" int mmdfn_lstm_gate_fwd(const float* q, const float* h, const float* c, const float* Wih, const float* Whh,
                                   const float* bsum, const float* bsum2, float* gates, float* h_out, float* c_out, int R, int H,
                                   void* stream) {
    return mmdfn_lstm_gate_fwd_ld(q, h, c, Wih, Whh, bsum, bsum2, gates, h_out, c_out, R, H, H, stream);
}

extern "C" int mmdfn_lstm_gate_bwd(const float* gates, const float* c_prev, const float* c_new, const float* dh_a,
                                   const float* dh_b, const float* dc_next, const float* Wih, const float* Whh,
                                   const float* dres, float* dG, float* dc_prev, float* dq, float* dh_prev, int R, int H,
                                   int has_h, int lddres, void* stream) {
    if (bad_dims(R, H) || (has_h && (dc_prev == nullptr || dh_prev == nullptr || c_prev == nullptr))) return -1;
    if (dres != nullptr && (lddres < H || (lddres & 3))) return -1;
    const int nb = (H + CBW - 1) / CBW;
    const int ncb = has_h ? 2 * nb : nb;
    if (gate_ws(R)) {
        const size_t ldsw = (size_t)(CBW + 2 * RB) * lds_stride(4 * H) * sizeof(float);
        if (ldsw > 156 * 1024) return -1;
#ifdef MMDFN_TUNING
#define GBWS_ABL(A_)                                                                                                         \
    case A_:                                                                                                                 \
        if (int e_ = mmdfn_allow_big_lds(lstm_gate_bwd_ws_kernel<A_>)) return e_;                                            \
        hipLaunchKernelGGL(lstm_gate_bwd_ws_kernel<A_>, dim3(row_groups(R, ncb), ncb), dim3(512), ldsw, (hipStream_t)stream, \
                           gates, c_prev, c_new, dh_a, dh_b, dc_next, Wih, Whh, dres, dG, dc_prev, dq, dh_prev, R, H, has_h, \
                           lddres);                                                                                          \
        MMDFN_CHECK_LAUNCH();                                                                                                \
        return 0;
        if (const char* e = getenv("MMDFN_GATE_ABL")) {
            switch (atoi(e)) {
                GBWS_ABL(1) GBWS_ABL(2) GBWS_ABL(4) GBWS_ABL(8) GBWS_ABL(3) GBWS_ABL(6) GBWS_ABL(7) GBWS_ABL(15) GBWS_ABL(12) GBWS_ABL(14)
                default: break;
            }
        }
#undef GBWS_ABL
#endif
        // WIDE: one column block per product while the register tiles reach H (64 < H <= 128) and the launch is long enough for
        // the halved producer side to matter (the extra weight-fragment loads are a fixed cost per workgroup)
        bool wide = H > CBW && 4 * H <= 16 * GATE_BWD_WIDE_NKC && R >= GATE_BWD_WIDE_ROWS;
#ifdef MMDFN_TUNING
        if (const char* e = getenv("MMDFN_GATE_BWD_WIDE")) wide = H > CBW && 4 * H <= 16 * GATE_BWD_WIDE_NKC && atoi(e) != 0;
#endif
        if (wide) {
            const int ncw = has_h ? 2 : 1;
            if (int e_ = mmdfn_allow_big_lds(lstm_gate_bwd_ws_kernel<0, true>)) return e_;
            hipLaunchKernelGGL((lstm_gate_bwd_ws_kernel<0, true>), dim3(row_groups(R, ncw), ncw), dim3(512), ldsw, (hipStream_t)stream,
                               gates, c_prev, c_new, dh_a, dh_b, dc_next, Wih, Whh, dres, dG, dc_prev, dq, dh_prev, R, H, has_h, lddres);
            MMDFN_CHECK_LAUNCH();
            return 0;
        }
        if (int e_ = mmdfn_allow_big_lds(lstm_gate_bwd_ws_kernel<0>)) return e_;
        hipLaunchKernelGGL(lstm_gate_bwd_ws_kernel<0>, dim3(row_groups(R, ncb), ncb), dim3(512), ldsw, (hipStream_t)stream, gates,
                           c_prev, c_new, dh_a, dh_b, dc_next, Wih, Whh, dres, dG, dc_prev, dq, dh_prev, R, H, has_h, lddres);
        MMDFN_CHECK_LAUNCH();
        return 0;
    }
    const size_t lds = (size_t)(CBW + 2 * RB) * lds_stride(4 * H) * sizeof(float);
    LAUNCH_BIG_LDS(lstm_gate_bwd_kernel, dim3(row_groups(R, ncb), ncb), lds, stream, gates, c_prev, c_new, dh_a, dh_b, dc_next,
                   Wih, Whh, dres, dG, dc_prev, dq, dh_prev, R, H, has_h, lddres);
    return 0;
}

extern "C" int mmdfn_gcnii_layer_fwd(const float* hi, const float* h0, const float* W, const float* q, const float* m,
                                     float* out, float* gmask, float theta, float alpha, int R, int H, int ldo, float mscale,
                                     void* stream) {
    if (bad_dims(R, H) || ldo < H || (ldo & 3)) return -1;
    if (layer_ws(R)) {
        const size_t ldsw = ((size_t)(H + 2 * RB) * lds_stride(2 * H) + 2 * RB * (((H + 15) & ~15) + 4)) * sizeof(float);
        if (ldsw > 156 * 1024) return -1;
        if (int e_ = mmdfn_allow_big_lds(gcnii_layer_fwd_ws_kernel)) return e_;
        hipLaunchKernelGGL(gcnii_layer_fwd_ws_kernel, dim3(row_groups(R, 1)), dim3(512), ldsw, (hipStream_t)stream, hi, h0, W, q, m,
                           out, gmask, theta, alpha, R, H, ldo, mscale);
        MMDFN_CHECK_LAUNCH();
        return 0;
    }
    const size_t lds = ((size_t)(H + 2 * RB) * lds_stride(2 * H) + RB * (((H + 15) & ~15) + 4)) * sizeof(float);
    LAUNCH_BIG_LDS(gcnii_layer_fwd_kernel, dim3(row_groups(R, 1)), lds, stream, hi, h0, W, q, m, out, gmask, theta, alpha, R, H,
                   ldo, mscale);
    return 0;
}

extern "C" int mmdfn_gcnii_layer_bwd_ld(const float* dout, const float* gmask, const float* W, float* dP, float* dhi, float* dh0,
                                        float theta, float alpha, int R, int H, int lddo, int acc_h0, int lddhi, void* stream) {
    if (bad_dims(R, H) || lddo < H || (lddo & 3) || lddhi < H || (lddhi & 3) || !(theta > 0.f)) return -1;
    if (layer_ws(R)) {
        const size_t ldsw = ((size_t)(2 * H + 2 * RB) * lds_stride(H) + 2 * RB * (((2 * H + 15) & ~15) + 4)) * sizeof(float);
        if (ldsw > 156 * 1024) return -1;
        if (int e_ = mmdfn_allow_big_lds(gcnii_layer_bwd_ws_kernel)) return e_;
        hipLaunchKernelGGL(gcnii_layer_bwd_ws_kernel, dim3(row_groups(R, 1)), dim3(512), ldsw, (hipStream_t)stream, dout, gmask, W,
                           dP, dhi, dh0, theta, alpha, R, H, lddo, acc_h0, lddhi);
        MMDFN_CHECK_LAUNCH();
        return 0;
    }
    const size_t lds = ((size_t)(2 * H + 2 * RB) * lds_stride(H) + RB * (((2 * H + 15) & ~15) + 4)) * sizeof(float);
    LAUNCH_BIG_LDS(gcnii_layer_bwd_kernel, dim3(row_groups(R, 1)), lds, stream, dout, gmask, W, dP, dhi, dh0, theta, alpha, R, H,
                   lddo, acc_h0, lddhi);
    return 0;
}

extern "C" int mmdfn_gcnii_layer_bwd(const float* dout, const float* gmask, const float* W, float* dP, float* dhi, float* dh0,
                                     float theta, float alpha, int R, int H, int lddo, int acc_h0, void* stream) {
    return mmdfn_gcnii_layer_bwd_ld(dout, gmask, W, dP, dhi, dh0, theta, alpha, R, H, lddo, acc_h0, H, stream);
}
