// K1: dense feature projections  Y = act(X W^T + b) (+ Y)  on exact-f32 MFMA.
//
// Replaces nn.Linear / F.linear on the hot path (modality projections model.py:1065,1094,1129; the hoisted
// GRU input contraction; the GCN input layer model_GCN.py:454; the LSTM gate pre-activations
// model_GCN.py:466; the GCNII support product model_GCN.py:186).  These are true contractions, so they run
// on the matrix cores: v_mfma_f32_16x16x4_f32 (k-ordered fp32 fma chain, no reduced precision).
//
// Both operands are k-contiguous (X: (R, K) rows, W: (N, K) rows = nn.Linear layout), so both MFMA
// fragments are loaded straight from global memory / L2 with 16-byte loads: lane (i, g) holds
// X[row0+i][k0+4g..+3] resp. W[col0+i][k0+4g..+3] and MFMA step j consumes component j of both (the same
// k-permutation on A and B).  Register blocking RT x CT 16x16 tiles per wave; the next 16-wide k-chunk is
// prefetched into a second register set while the current one feeds the MFMAs; no LDS.
#include "mmdfn_internal.h"
#include "../../include/mmdfn_hip.h"
#include <stdlib.h>

namespace {

template <int RT, int CT, int WR, int WC>
__global__ __launch_bounds__(64 * WR * WC) void linear_kernel(const float* __restrict__ X, const float* __restrict__ W,
                                                             const float* __restrict__ W2, int N1,
                                                             const float* __restrict__ bias, const float* __restrict__ bias2,
                                                             float* __restrict__ Y, int R, int K, int N, int ldx, int ldy,
                                                             int act, int accumulate) {
    // output columns [0, N1) use rows of W / bias, columns [N1, N) rows of W2 / bias2 (the two directions of a
    // bidirectional GRU layer keep their own weight_ih parameters: no concatenated copy per step); N1 = N: one block
    constexpr int BM = 16 * RT * WR;
    constexpr int BN = 16 * CT * WC;
    const int nbn = (N + BN - 1) / BN;
    const int bm = blockIdx.x / nbn;
    const int bn = blockIdx.x - bm * nbn;
    const int lane = threadIdx.x & 63;
    const int w = threadIdx.x >> 6;
    const int wr = w % WR;
    const int wc = w / WR;
    const int fi = lane & 15;
    const int g = lane >> 4;
    const int row0 = bm * BM + wr * 16 * RT;
    const int col0 = bn * BN + wc * 16 * CT;
    if (row0 >= R || col0 >= N) return;

    const float* xp[RT];
    bool xok[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        const int r = row0 + 16 * rt + fi;
        xok[rt] = r < R;
        xp[rt] = X + (int64_t)(r < R ? r : R - 1) * ldx;
    }
    const float* wp[CT];
    bool wok[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
        const int c = col0 + 16 * ct + fi;
        wok[ct] = c < N;
        const int cc = c < N ? c : N - 1;
        wp[ct] = (cc < N1) ? W + (int64_t)cc * K : W2 + (int64_t)(cc - N1) * K;
    }

    f32x4 acc[RT][CT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) acc[rt][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};

    float4 a[2][RT], b[2][CT];
    const int nk = (K + 15) / 16;
#define LIN_LOAD(SET, KC)                                                                        \
    do {                                                                                         \
        const int k_ = 16 * (KC) + 4 * g;                                                        \
        const int kc_ = k_ < K ? k_ : K - 4;                                                     \
        _Pragma("unroll") for (int rt = 0; rt < RT; ++rt) a[SET][rt] = *reinterpret_cast<const float4*>(xp[rt] + kc_); \
        _Pragma("unroll") for (int ct = 0; ct < CT; ++ct) b[SET][ct] = *reinterpret_cast<const float4*>(wp[ct] + kc_); \
    } while (0)
#define LIN_MMA(SET, KC)                                                                         \
    do {                                                                                         \
        const bool kok = (16 * (KC) + 4 * g) < K;                                                \
        float av[RT][4], bv[CT][4];                                                              \
        _Pragma("unroll") for (int rt = 0; rt < RT; ++rt) {                                      \
            const bool ok = kok && xok[rt];                                                      \
            av[rt][0] = ok ? a[SET][rt].x : 0.f; av[rt][1] = ok ? a[SET][rt].y : 0.f;            \
            av[rt][2] = ok ? a[SET][rt].z : 0.f; av[rt][3] = ok ? a[SET][rt].w : 0.f;            \
        }                                                                                        \
        _Pragma("unroll") for (int ct = 0; ct < CT; ++ct) {                                      \
            const bool ok = kok && wok[ct];                                                      \
            bv[ct][0] = ok ? b[SET][ct].x : 0.f; bv[ct][1] = ok ? b[SET][ct].y : 0.f;            \
            bv[ct][2] = ok ? b[SET][ct].z : 0.f; bv[ct][3] = ok ? b[SET][ct].w : 0.f;            \
        }                                                                                        \
        if ((KC) + 1 < nk) LIN_LOAD(1 - (SET), (KC) + 1);                                        \
        __builtin_amdgcn_sched_barrier(0);                                                       \
        _Pragma("unroll") for (int j = 0; j < 4; ++j)                                            \
            _Pragma("unroll") for (int rt = 0; rt < RT; ++rt)                                    \
                _Pragma("unroll") for (int ct = 0; ct < CT; ++ct)                                \
                    acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[rt][j], bv[ct][j], acc[rt][ct], 0, 0, 0); \
        __builtin_amdgcn_sched_barrier(0);                                                       \
    } while (0)

    LIN_LOAD(0, 0);
    for (int kc = 0; kc < nk; kc += 2) {
        LIN_MMA(0, kc);
        if (kc + 1 < nk) LIN_MMA(1, kc + 1);
    }
#undef LIN_LOAD
#undef LIN_MMA

    // epilogue: C/D layout col = lane&15, row = 4g + r
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
        const int c = col0 + 16 * ct + fi;
        if (c >= N) continue;
        const float bb = bias ? (c < N1 ? bias[c] : bias2[c - N1]) : 0.f;
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = row0 + 16 * rt + 4 * g + r;
                if (row >= R) continue;
                float v = acc[rt][ct][r] + bb;
                float* y = Y + (int64_t)row * ldy + c;
                if (accumulate) v += *y;
                if (act == 1) v = fmaxf(v, 0.f);
                *y = v;
            }
    }
}

template <int RT, int CT, int WR, int WC>
int launch(const float* X, const float* W, const float* W2, int N1, const float* bias, const float* bias2, float* Y, int R,
           int K, int N, int ldx, int ldy, int act, int accumulate, hipStream_t s) {
    const int BM = 16 * RT * WR, BN = 16 * CT * WC;
    dim3 grid(((R + BM - 1) / BM) * ((N + BN - 1) / BN));
    hipLaunchKernelGGL((linear_kernel<RT, CT, WR, WC>), grid, dim3(64 * WR * WC), 0, s, X, W, W2, N1, bias, bias2, Y, R, K,
                       N, ldx, ldy, act, accumulate);
    MMDFN_CHECK_LAUNCH();
    return 0;
}

}  // namespace

extern "C" int mmdfn_linear2(const float* X, const float* W, const float* W2, int N1, const float* bias,
                             const float* bias2, float* Y, int R, int K, int N, int ldx, int ldy, int act, int accumulate,
                             void* stream) {
    if (R <= 0 || K < 4 || N <= 0 || (K & 3) || (ldx & 3) || ldx < K || ldy < N) return -1;
    if (N1 <= 0 || N1 > N || (N1 < N && W2 == nullptr) || (bias != nullptr && N1 < N && bias2 == nullptr)) return -1;
    hipStream_t s = (hipStream_t)stream;
#ifdef MMDFN_TUNING
    const char* e = getenv("MMDFN_LIN_CFG");  // tools/bench_linear.py
    const int ov = e ? atoi(e) : -1;
    if (ov == 0) return launch<2, 4, 2, 2>(X, W, W2, N1, bias, bias2, Y, R, K, N, ldx, ldy, act, accumulate, s);
    if (ov == 1) return launch<1, 7, 4, 1>(X, W, W2, N1, bias, bias2, Y, R, K, N, ldx, ldy, act, accumulate, s);
    if (ov == 2) return launch<1, 13, 4, 1>(X, W, W2, N1, bias, bias2, Y, R, K, N, ldx, ldy, act, accumulate, s);
    if (ov == 3) return launch<2, 4, 1, 4>(X, W, W2, N1, bias, bias2, Y, R, K, N, ldx, ldy, act, accumulate, s);
    if (ov == 4) return launch<4, 4, 2, 2>(X, W, W2, N1, bias, bias2, Y, R, K, N, ldx, ldy, act, accumulate, s);
    if (ov == 5) return launch<2, 2, 2, 2>(X, W, W2, N1, bias, bias2, Y, R, K, N, ldx, ldy, act, accumulate, s);
    if (ov == 6) return launch<1, 4, 4, 1>(X, W, W2, N1, bias, bias2, Y, R, K, N, ldx, ldy, act, accumulate, s);
    if (ov == 7) {
        const int rc = mmdfn_launch_linear_split(X, W, W2, N1, bias, bias2, Y, R, K, N, ldx, ldy, act, accumulate, s);
        if (rc != -2) return rc;
    }
#else
    constexpr int ov = -1;
#endif
    if (ov != 8) {
        // many 128 x 128 output tiles: the bf16-piece kernel (fp32-level error) outruns the exact-f32 MFMA path
        // 1.5-2x (tools/bench_linear.py: 10560 x 200 -> 600: 54 -> 33 us; 98304 x 200 -> 100: 79 -> 51 us);
        // with fewer tiles its 4-wave 128 x 128 workgroups leave the chip idle
        const long tiles = (long)((R + 127) / 128) * ((N + 127) / 128);
        if (tiles >= 256 && K >= 32) {
            const int rc = mmdfn_launch_linear_split(X, W, W2, N1, bias, bias2, Y, R, K, N, ldx, ldy, act, accumulate, s);
            if (rc != -2) return rc;
        }
    }
    // measured (tools/bench_linear.py): the 64 x 64 workgroup tile (2 x 2 MFMA tiles per wave) wins on every
    // hot-path shape -- the kernel is latency-bound, so more, smaller workgroups beat bigger register tiles
    return launch<2, 2, 2, 2>(X, W, W2, N1, bias, bias2, Y, R, K, N, ldx, ldy, act, accumulate, s);
}

extern "C" int mmdfn_linear(const float* X, const float* W, const float* bias, float* Y, int R, int K, int N, int ldx,
                            int ldy, int act, int accumulate, void* stream) {
    return mmdfn_linear2(X, W, nullptr, N, bias, nullptr, Y, R, K, N, ldx, ldy, act, accumulate, stream);
}
