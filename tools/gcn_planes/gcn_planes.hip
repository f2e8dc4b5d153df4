// K7p: the GCNII layer update and its backward (reference GraphConvolution.forward model_GCN.py:176-189 inside GCNII_lyc.forward
// :469-472) for MANY-ROW launches, with the layer's weight taken from bf16 piece planes that are cut once per optimizer step
// (planes_common.h / planes_pipeline.h; the plane registry lives in mm_dfn_amd/ops_linear.py).
//
// The exact-f32 kernels of gcn_stack.hip stage W in LDS per workgroup and multiply 16-row blocks on v_mfma_f32_16x16x4_f32: at
// BASELINE cfg5 (98 304 rows) they run at 0.27-0.29 of the exact-f32 matrix rate (84 / 92 us, VERDICT r05 "weak" 2).  Here the
// contraction is the shared piece-plane pipeline (six bf16 piece products per MAC, fp32-level error), 64 rows per workgroup:
//   forward   pre = theta [hi | h0] W + (1 - theta)((1 - alpha) hi + alpha h0);  out = relu(pre) (.) m ms + q;  gmask = m ms [pre > 0]
//             A = [hi | h0] read from its two sources (K = 2H), B = W^T planes (N = H: one column block of four tiles), the blend,
//             ReLU, dropout flags and the residual in the epilogue (hi / h0 / m / q re-read at the accumulator positions: whole
//             128-byte row segments)
//   backward  dP = theta dout (.) gmask (written out: operand of dW);  [dhi | dh0] = dP W^T + [c1 dP | c2 dP]
//             A = dP formed while staging (K = H), B = W planes as stored (N = 2H: two column blocks; block 0 also writes dP)
// Same operands, layouts and results (to fp32 rounding) as mmdfn_gcnii_layer_fwd / _bwd_ld.
// NOT PART OF THE LIBRARY: measured slower than the exact-f32 kernels in the step (README.md next to this file).
#include "planes_common.h"      // (mm_dfn_amd/csrc/)
#include "../../include/mmdfn_hip.h"

namespace {

// ---------------------------------------------------------------------------------------------------------------------------
template <int RH>
__global__ __launch_bounds__(256, 4) void gcnii_layer_fwd_planes_kernel(
    const float* __restrict__ hi, const float* __restrict__ h0, const u32x4* __restrict__ planes, const float* __restrict__ q,
    const float* __restrict__ m, float* __restrict__ out, float* __restrict__ gmask, float theta, float alpha, int R, int H,
    int ldo, float ms, int nrb, int ncb) {
    __shared__ u32x4 As[PL_LDS];
    constexpr int TPW = 4 / (3 - RH);
    constexpr int NACC = (RH == 2) ? 1 : 2;
    int rb, cb;
    if (!pl_decode(nrb, ncb, rb, cb, (int)blockIdx.x)) return;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = tid >> 6;
    const int K = 2 * H;
    const int KS = (K + 15) >> 4;
    const int NPH = (KS + PL_STG - 1) / PL_STG;
    const int NT = (H + 31) >> 5;
    const int ct = TPW * cb + (RH == 2 ? w : (w >> 1));
    const int myh = (RH == 2) ? 0 : (w & 1);
    const bool has_tile = ct < NT;
    const int r0 = rb * PL_BM;

    f32x16 acc[RH][NACC];
#pragma unroll
    for (int h = 0; h < RH; ++h)
#pragma unroll
        for (int c = 0; c < NACC; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[h][c][r] = 0.f;

    const u32x4* bsrc = planes + ((int64_t)(has_tile ? ct : 0) * KS) * 3 * 64 + lane;
    f32x4 raw[2][2];
    const int shf = w & 1, sksl = w >> 1;
    const int srow = r0 + 32 * shf + (lane & 31);
    const int64_t srow_c = srow < R ? srow : R - 1;
    const float* hirow = hi + srow_c * H;
    const float* h0row = h0 + srow_c * H;
    const int skofs = 16 * sksl + 8 * (lane >> 5);

    // a float4 of the concatenated row [hi | h0] lies in ONE source (H % 4 == 0)
#define K7_SRC(K4) ((K4) < H ? hirow + (K4) : ((K4) < K ? h0row + ((K4) - H) : hirow))
#define PL_ISSUE_X(PAR, PH)                                                                                 \
    do {                                                                                                    \
        const int k0_ = 16 * PL_STG * (PH) + skofs;                                                         \
        raw[PAR][0] = *reinterpret_cast<const f32x4*>(K7_SRC(k0_));                                         \
        raw[PAR][1] = *reinterpret_cast<const f32x4*>(K7_SRC(k0_ + 4));                                     \
    } while (0)
#define PL_PARK(PAR, PH, BUF)                                                                               \
    do {                                                                                                    \
        const int k0_ = 16 * PL_STG * (PH) + skofs;                                                         \
        const bool ok0_ = k0_ < K, ok1_ = k0_ + 4 < K;                                                      \
        float x_[8] = {ok0_ ? raw[PAR][0].x : 0.f, ok0_ ? raw[PAR][0].y : 0.f, ok0_ ? raw[PAR][0].z : 0.f, ok0_ ? raw[PAR][0].w : 0.f, \
                       ok1_ ? raw[PAR][1].x : 0.f, ok1_ ? raw[PAR][1].y : 0.f, ok1_ ? raw[PAR][1].z : 0.f, ok1_ ? raw[PAR][1].w : 0.f}; \
        u32x4 p1_, p2_, p3_;                                                                                \
        pl_cut8(x_, p1_, p2_, p3_);                                                                         \
        u32x4* dst_ = &As[(BUF) * (PL_LDS / 2) + ((sksl * 3) * 2 + shf) * 64 + lane];                       \
        dst_[0] = p1_;                                                                                      \
        dst_[2 * 64] = p2_;                                                                                 \
        dst_[4 * 64] = p3_;                                                                                 \
    } while (0)
#include "planes_pipeline.h"
#undef PL_PARK
#undef PL_ISSUE_X
#undef K7_SRC

    if (!has_tile) return;
    const int col = 32 * ct + (lane & 31);
    if (col >= H) return;
    const float omt = 1.0f - theta, ca = (1.0f - alpha), emul = m ? ms : 1.0f;
#pragma unroll
    for (int h = 0; h < RH; ++h)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = r0 + 32 * (RH == 2 ? h : myh) + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            if (row < R) {
                const int64_t o = (int64_t)row * H + col;
                const float p = NACC == 2 ? acc[h][0][r] + acc[h][NACC - 1][r] : acc[h][0][r];
                const float pre = theta * p + omt * (ca * hi[o] + alpha * h0[o]);
                const float em = (m ? m[o] : 1.0f) * emul;
                out[(int64_t)row * ldo + col] = fmaxf(pre, 0.f) * em + (q ? q[o] : 0.f);
                __builtin_nontemporal_store(pre > 0.f ? em : 0.f, &gmask[o]);       // (saved for the backward pass)
            }
        }
}

// ---------------------------------------------------------------------------------------------------------------------------
template <int RH>
__global__ __launch_bounds__(256, 4) void gcnii_layer_bwd_planes_kernel(
    const float* __restrict__ dout, const float* __restrict__ gmask, const u32x4* __restrict__ planes, float* __restrict__ dP,
    float* __restrict__ dhi, float* __restrict__ dh0, float theta, float alpha, int R, int H, int lddo, int acc_h0, int lddhi,
    int nrb, int ncb) {
    __shared__ u32x4 As[PL_LDS];
    constexpr int TPW = 4 / (3 - RH);
    constexpr int NACC = (RH == 2) ? 1 : 2;
    int rb, cb;
    if (!pl_decode(nrb, ncb, rb, cb, (int)blockIdx.x)) return;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = tid >> 6;
    const int K = H;
    const int N = 2 * H;
    const int KS = (K + 15) >> 4;
    const int NPH = (KS + PL_STG - 1) / PL_STG;
    const int NT = (N + 31) >> 5;
    const int ct = TPW * cb + (RH == 2 ? w : (w >> 1));
    const int myh = (RH == 2) ? 0 : (w & 1);
    const bool has_tile = ct < NT;
    const int r0 = rb * PL_BM;

    f32x16 acc[RH][NACC];
#pragma unroll
    for (int h = 0; h < RH; ++h)
#pragma unroll
        for (int c = 0; c < NACC; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[h][c][r] = 0.f;

    const u32x4* bsrc = planes + ((int64_t)(has_tile ? ct : 0) * KS) * 3 * 64 + lane;
    f32x4 raw[2][4];                                     // [phase parity][dout x 2, gmask x 2]
    const int shf = w & 1, sksl = w >> 1;
    const int srow = r0 + 32 * shf + (lane & 31);
    const bool srow_ok = srow < R;
    const int64_t srow_c = srow_ok ? srow : R - 1;
    const float* dorow = dout + srow_c * lddo;
    const float* gmrow = gmask + srow_c * H;
    float* dprow = dP + srow_c * H;
    const int skofs = 16 * sksl + 8 * (lane >> 5);
    const bool writes_dp = (cb == 0) && srow_ok;         // the first column block of a row block also writes dP

#define PL_ISSUE_X(PAR, PH)                                                                                 \
    do {                                                                                                    \
        const int k0_ = 16 * PL_STG * (PH) + skofs;                                                         \
        const int ka_ = k0_ < K ? k0_ : 0, kb_ = k0_ + 4 < K ? k0_ + 4 : 0;                                 \
        raw[PAR][0] = *reinterpret_cast<const f32x4*>(dorow + ka_);                                         \
        raw[PAR][1] = *reinterpret_cast<const f32x4*>(dorow + kb_);                                         \
        raw[PAR][2] = *reinterpret_cast<const f32x4*>(gmrow + ka_);                                         \
        raw[PAR][3] = *reinterpret_cast<const f32x4*>(gmrow + kb_);                                         \
    } while (0)
#define PL_PARK(PAR, PH, BUF)                                                                               \
    do {                                                                                                    \
        const int k0_ = 16 * PL_STG * (PH) + skofs;                                                         \
        const bool ok0_ = k0_ < K, ok1_ = k0_ + 4 < K;                                                      \
        const f32x4 pa_ = raw[PAR][0] * raw[PAR][2] * theta, pb_ = raw[PAR][1] * raw[PAR][3] * theta;       \
        float x_[8] = {ok0_ ? pa_.x : 0.f, ok0_ ? pa_.y : 0.f, ok0_ ? pa_.z : 0.f, ok0_ ? pa_.w : 0.f,      \
                       ok1_ ? pb_.x : 0.f, ok1_ ? pb_.y : 0.f, ok1_ ? pb_.z : 0.f, ok1_ ? pb_.w : 0.f};     \
        if (writes_dp) {                                                                                    \
            if (ok0_) *reinterpret_cast<f32x4*>(dprow + k0_) = pa_;                                         \
            if (ok1_) *reinterpret_cast<f32x4*>(dprow + k0_ + 4) = pb_;                                     \
        }                                                                                                   \
        u32x4 p1_, p2_, p3_;                                                                                \
        pl_cut8(x_, p1_, p2_, p3_);                                                                         \
        u32x4* dst_ = &As[(BUF) * (PL_LDS / 2) + ((sksl * 3) * 2 + shf) * 64 + lane];                       \
        dst_[0] = p1_;                                                                                      \
        dst_[2 * 64] = p2_;                                                                                 \
        dst_[4 * 64] = p3_;                                                                                 \
    } while (0)
#include "planes_pipeline.h"
#undef PL_PARK
#undef PL_ISSUE_X

    if (!has_tile) return;
    const int col = 32 * ct + (lane & 31);
    if (col >= N) return;
    // (1 - theta)(1 - alpha) gg = c1 dP, (1 - theta) alpha gg = c2 dP   (theta = ln(lamda / l + 1) > 0)
    const bool left = col < H;
    const int c = left ? col : col - H;
    const float cc = left ? (1.0f - theta) * (1.0f - alpha) : (1.0f - theta) * alpha;      // times gg = dout (.) gmask
#pragma unroll
    for (int h = 0; h < RH; ++h)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = r0 + 32 * (RH == 2 ? h : myh) + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            if (row < R) {
                const float gg = dout[(int64_t)row * lddo + c] * gmask[(int64_t)row * H + c];
                const float v = (NACC == 2 ? acc[h][0][r] + acc[h][NACC - 1][r] : acc[h][0][r]) + cc * gg;
                if (left) dhi[(int64_t)row * lddhi + c] = v;
                else {
                    float* p = dh0 + (int64_t)row * H + c;
                    *p = acc_h0 ? *p + v : v;
                }
            }
        }
}

// ---------------------------------------------------------------------------------------------------------------------------
// K8p backward: the reasoning module's LSTM cell (nn.LSTM seq_len 1, model_GCN.py:463-467) for many-row launches.
//   dh' = dh_a + dh_b, dc = dc_next + dh' o (1 - tanh^2 c_new) -> dG = (di, df, dg, do) pre-activation gradients (R, 4H: operand of
//   dW_ih / dW_hh / db), dc_prev = dc f;  [dq | dh_prev] = dG [W_ih | W_hh] (+ dres on the dq half).
// The contraction index is walked GATE-INTERLEAVED (k = 4 u + g: the four gates of a hidden unit are neighbours; the planes are
// cut in the same order, mmdfn_cut_weight_planes mode 3), so a staging item = (row, unit) forms its four gate gradients from ONE
// set of nine operand loads and drops them as one 8-byte write per piece into the A-fragment image; a phase (32 k) = 8 units of
// the workgroup's 64 rows = two items per thread.  Column block 0 of a row block also writes dG and dc_prev.
__device__ __forceinline__ float k8_tanh(float x) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(2.0f * x)); }

template <int RH>
__global__ __launch_bounds__(256, 3) void lstm_gate_bwd_planes_kernel(
    const float* __restrict__ gates, const float* __restrict__ c_prev, const float* __restrict__ c_new,
    const float* __restrict__ dh_a, const float* __restrict__ dh_b, const float* __restrict__ dc_next,
    const u32x4* __restrict__ planes, const float* __restrict__ dres, float* __restrict__ dG, float* __restrict__ dc_prev,
    float* __restrict__ dq, float* __restrict__ dh_prev, int R, int H, int has_h, int lddres, int nrb, int ncb) {
    __shared__ u32x4 As[PL_LDS];
    constexpr int TPW = 4 / (3 - RH);
    constexpr int NACC = (RH == 2) ? 1 : 2;
    int rb, cb;
    if (!pl_decode(nrb, ncb, rb, cb, (int)blockIdx.x)) return;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = tid >> 6;
    const int K = 4 * H;
    const int N = has_h ? 2 * H : H;
    const int KS = (K + 15) >> 4;
    const int NPH = (KS + PL_STG - 1) / PL_STG;
    const int NT = (N + 31) >> 5;
    const int ct = TPW * cb + (RH == 2 ? w : (w >> 1));
    const int myh = (RH == 2) ? 0 : (w & 1);
    const bool has_tile = ct < NT;
    const int r0 = rb * PL_BM;

    f32x16 acc[RH][NACC];
#pragma unroll
    for (int h = 0; h < RH; ++h)
#pragma unroll
        for (int c = 0; c < NACC; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[h][c][r] = 0.f;

    const u32x4* bsrc = planes + ((int64_t)(has_tile ? ct : 0) * KS) * 3 * 64 + lane;
    // staging items of this thread in every phase: (row r0 + (tid >> 3) + 32 i, unit 8 PH + (tid & 7)), i = 0, 1
    const int uloc = tid & 7;
    float raw[2][2][9];                                  // [phase parity][item][gi gf gg go c_new c_prev dh_a dh_b dc_next]
    int64_t irow[2];
    bool irow_ok[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = r0 + (tid >> 3) + 32 * i;
        irow_ok[i] = row < R;
        irow[i] = irow_ok[i] ? row : R - 1;
    }
    const bool writer = cb == 0;

#define PL_ISSUE_X(PAR, PH)                                                                                 \
    do {                                                                                                    \
        const int u_ = 8 * (PH) + uloc;                                                                     \
        const int uc_ = u_ < H ? u_ : H - 1;                                                                \
        _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_) {                                                  \
            const float* g_ = gates + irow[i_] * K + uc_;                                                   \
            const int64_t o_ = irow[i_] * H + uc_;                                                          \
            raw[PAR][i_][0] = g_[0];                                                                        \
            raw[PAR][i_][1] = g_[H];                                                                        \
            raw[PAR][i_][2] = g_[2 * H];                                                                    \
            raw[PAR][i_][3] = g_[3 * H];                                                                    \
            raw[PAR][i_][4] = c_new[o_];                                                                    \
            raw[PAR][i_][5] = c_prev ? c_prev[o_] : 0.f;                                                    \
            raw[PAR][i_][6] = dh_a ? dh_a[o_] : 0.f;                                                        \
            raw[PAR][i_][7] = dh_b ? dh_b[o_] : 0.f;                                                        \
            raw[PAR][i_][8] = dc_next ? dc_next[o_] : 0.f;                                                  \
        }                                                                                                   \
    } while (0)
#define PL_PARK(PAR, PH, BUF)                                                                               \
    do {                                                                                                    \
        const int u_ = 8 * (PH) + uloc;                                                                     \
        _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_) {                                                  \
            const bool ok_ = u_ < H && irow_ok[i_];                                                         \
            const float gi_ = raw[PAR][i_][0], gf_ = raw[PAR][i_][1], gg_ = raw[PAR][i_][2], go_ = raw[PAR][i_][3]; \
            const float tc_ = k8_tanh(raw[PAR][i_][4]);                                                     \
            const float dhv_ = raw[PAR][i_][6] + raw[PAR][i_][7];                                           \
            const float dc_ = raw[PAR][i_][8] + dhv_ * go_ * (1.0f - tc_ * tc_);                            \
            const float dO_ = ok_ ? dhv_ * tc_ * go_ * (1.0f - go_) : 0.f;                                  \
            const float di_ = ok_ ? dc_ * gg_ * gi_ * (1.0f - gi_) : 0.f;                                   \
            const float df_ = ok_ ? dc_ * raw[PAR][i_][5] * gf_ * (1.0f - gf_) : 0.f;                       \
            const float dg_ = ok_ ? dc_ * gi_ * (1.0f - gg_ * gg_) : 0.f;                                   \
            if (writer && ok_) {                                                                            \
                float* d_ = dG + irow[i_] * K + u_;                                                         \
                d_[0] = di_; d_[H] = df_; d_[2 * H] = dg_; d_[3 * H] = dO_;                                 \
                if (has_h) dc_prev[irow[i_] * H + u_] = dc_ * gf_;                                          \
            }                                                                                               \
            uint32_t a1_, a2_, a3_, b1_, b2_, b3_;                                                          \
            pl_cut2(di_, df_, a1_, a2_, a3_);                                                               \
            pl_cut2(dg_, dO_, b1_, b2_, b3_);                                                               \
            /* k = 4 u + g inside the phase: k-step uloc >> 2, k group (uloc & 3) >> 1, 8-byte half uloc & 1 */ \
            const int rl_ = (tid >> 3) + 32 * i_;                                                           \
            const int slot_ = (rl_ & 31) + 32 * ((uloc & 3) >> 1);                                          \
            u32x2* dst_ = reinterpret_cast<u32x2*>(&As[(BUF) * (PL_LDS / 2) + (((uloc >> 2) * 3) * 2 + (rl_ >> 5)) * 64 + slot_]) \
                          + (uloc & 1);                                                                     \
            dst_[0] = u32x2{a1_, b1_};                                                                      \
            dst_[2 * 2 * 64] = u32x2{a2_, b2_};           /* (next piece: 2 halves x 64 slots x 2 u32x2) */  \
            dst_[4 * 2 * 64] = u32x2{a3_, b3_};                                                             \
        }                                                                                                   \
    } while (0)
#include "planes_pipeline.h"
#undef PL_PARK
#undef PL_ISSUE_X

    if (!has_tile) return;
    const int col = 32 * ct + (lane & 31);
    if (col >= N) return;
    const bool left = col < H;
    const int c = left ? col : col - H;
#pragma unroll
    for (int h = 0; h < RH; ++h)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = r0 + 32 * (RH == 2 ? h : myh) + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            if (row < R) {
                float v = NACC == 2 ? acc[h][0][r] + acc[h][NACC - 1][r] : acc[h][0][r];
                if (left) {
                    if (dres) v += dres[(int64_t)row * lddres + c];
                    dq[(int64_t)row * H + c] = v;
                } else {
                    dh_prev[(int64_t)row * H + c] = v;
                }
            }
        }
}

inline bool pl_narrow(int nrb, int NT) { return (int64_t)nrb * ((NT + 3) / 4) < 400; }

}  // namespace

extern "C" {

int mmdfn_gcnii_layer_fwd_planes(const float* hi, const float* h0, const void* planes, const float* q, const float* m,
                                 float* out, float* gmask, float theta, float alpha, int R, int H, int ldo, float mscale,
                                 void* stream) {
    hipStream_t s = (hipStream_t)stream;
    if (R <= 0) return 0;
    if (!hi || !h0 || !planes || !out || !gmask || H < 4 || ldo < H) return -1;
    if ((H & 3) || (reinterpret_cast<uintptr_t>(planes) & 15) || (reinterpret_cast<uintptr_t>(hi) & 15) ||
        (reinterpret_cast<uintptr_t>(h0) & 15))
        return -2;
    const int nrb = (R + PL_BM - 1) / PL_BM, NT = (H + 31) / 32;
    const bool narrow = pl_narrow(nrb, NT);
    const int ncb = narrow ? (NT + 1) / 2 : (NT + 3) / 4;
    const int64_t grid = pl_grid(nrb, ncb);
    if (grid > (1ll << 30)) return -1;
    if (narrow)
        hipLaunchKernelGGL(gcnii_layer_fwd_planes_kernel<1>, dim3((unsigned)grid), dim3(256), 0, s, hi, h0,
                           reinterpret_cast<const u32x4*>(planes), q, m, out, gmask, theta, alpha, R, H, ldo, mscale, nrb, ncb);
    else
        hipLaunchKernelGGL(gcnii_layer_fwd_planes_kernel<2>, dim3((unsigned)grid), dim3(256), 0, s, hi, h0,
                           reinterpret_cast<const u32x4*>(planes), q, m, out, gmask, theta, alpha, R, H, ldo, mscale, nrb, ncb);
    MMDFN_CHECK_LAUNCH();
    return 0;
}

int mmdfn_gcnii_layer_bwd_planes(const float* dout, const float* gmask, const void* planes, float* dP, float* dhi,
                                 float* dh0, float theta, float alpha, int R, int H, int lddo, int acc_h0, int lddhi,
                                 void* stream) {
    hipStream_t s = (hipStream_t)stream;
    if (R <= 0) return 0;
    if (!dout || !gmask || !planes || !dP || !dhi || !dh0 || H < 4 || lddo < H || lddhi < H || !(theta > 0.f)) return -1;
    if ((H & 3) || (lddo & 3) || (reinterpret_cast<uintptr_t>(planes) & 15) || (reinterpret_cast<uintptr_t>(dout) & 15) ||
        (reinterpret_cast<uintptr_t>(gmask) & 15) || (reinterpret_cast<uintptr_t>(dP) & 15))
        return -2;
    const int nrb = (R + PL_BM - 1) / PL_BM, NT = (2 * H + 31) / 32;
    const bool narrow = pl_narrow(nrb, NT);
    const int ncb = narrow ? (NT + 1) / 2 : (NT + 3) / 4;
    const int64_t grid = pl_grid(nrb, ncb);
    if (grid > (1ll << 30)) return -1;
    if (narrow)
        hipLaunchKernelGGL(gcnii_layer_bwd_planes_kernel<1>, dim3((unsigned)grid), dim3(256), 0, s, dout, gmask,
                           reinterpret_cast<const u32x4*>(planes), dP, dhi, dh0, theta, alpha, R, H, lddo, acc_h0, lddhi, nrb, ncb);
    else
        hipLaunchKernelGGL(gcnii_layer_bwd_planes_kernel<2>, dim3((unsigned)grid), dim3(256), 0, s, dout, gmask,
                           reinterpret_cast<const u32x4*>(planes), dP, dhi, dh0, theta, alpha, R, H, lddo, acc_h0, lddhi, nrb, ncb);
    MMDFN_CHECK_LAUNCH();
    return 0;
}

int mmdfn_lstm_gate_bwd_planes(const float* gates, const float* c_prev, const float* c_new, const float* dh_a, const float* dh_b,
                               const float* dc_next, const void* planes, const float* dres, float* dG, float* dc_prev, float* dq,
                               float* dh_prev, int R, int H, int has_h, int lddres, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    if (R <= 0) return 0;
    if (!gates || !c_new || !planes || !dG || !dq || H < 4 || (H & 3) || (dres && lddres < H)) return -1;
    if (has_h && (!dc_prev || !dh_prev)) return -1;
    if (reinterpret_cast<uintptr_t>(planes) & 15) return -1;
    const int nrb = (R + PL_BM - 1) / PL_BM, NT = ((has_h ? 2 * H : H) + 31) / 32;
    const bool narrow = pl_narrow(nrb, NT);
    const int ncb = narrow ? (NT + 1) / 2 : (NT + 3) / 4;
    const int64_t grid = pl_grid(nrb, ncb);
    if (grid > (1ll << 30)) return -1;
    if (narrow)
        hipLaunchKernelGGL(lstm_gate_bwd_planes_kernel<1>, dim3((unsigned)grid), dim3(256), 0, s, gates, c_prev, c_new, dh_a, dh_b,
                           dc_next, reinterpret_cast<const u32x4*>(planes), dres, dG, dc_prev, dq, dh_prev, R, H, has_h, lddres, nrb,
                           ncb);
    else
        hipLaunchKernelGGL(lstm_gate_bwd_planes_kernel<2>, dim3((unsigned)grid), dim3(256), 0, s, gates, c_prev, c_new, dh_a, dh_b,
                           dc_next, reinterpret_cast<const u32x4*>(planes), dres, dG, dc_prev, dq, dh_prev, R, H, has_h, lddres, nrb,
                           ncb);
    MMDFN_CHECK_LAUNCH();
    return 0;
}

}  // extern "C"
