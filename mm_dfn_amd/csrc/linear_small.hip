// K1 (few-row variant): a GROUP of small dense projections  Y_p = act(X_p W_p^T + b_p) (+ Y_p)  in ONE launch, exact-f32 MFMA.
//
// Replaces nn.Linear / F.linear / the autograd input-gradient GEMMs on the encoder side of the hot path where the row
// count is a few thousand (BASELINE cfg2-cfg4: 1 056 .. 14 080 rows): the three modality projections
// (model.py:1065,1094,1129), the hoisted GRU input contractions (model.py:1082,1132) and their input gradients
// dX = dY . [W_ih; W_ih_reverse].  Rounds 1-2 left these to hipBLASLt: at 1 760 rows a 64 x 64-tile kernel has 112
// workgroups, each streaming 128 K-floats of operands through ONE compute unit's L1 (~30-60 GB/s per CU, the limiter
// of every small-operand kernel on this chip, profiles/r03_k6_memory_path.md) and issuing 16 k-steps x 16 exact-f32 MFMAs
// back to back on one wave per SIMD.  Here:
//   * workgroup = ONE 32 x 32 output tile, its four waves split the CONTRACTION (each wave a quarter of K, the whole
//     tile): 4x more workgroups than 64 x 64 tiles with the same operand traffic per output element of a 32 x 32 tile, the
//     four partial tiles meet in LDS in a fixed order (bit-reproducible) and the epilogue (bias, ReLU, accumulate) is
//     finished row-wise with 16-byte stores;
//   * every operand chunk of a wave is requested before the first MFMA (a ring of PRE chunks in registers: a wave has
//     at most ~10 chunks of 16 k), so the kernel is one memory round trip deep instead of one per chunk;
//   * up to 8 problems per launch (the three modality projections; the context and the party GRU of a layer), because at
//     these sizes a launch boundary (~2 us) is a third of a projection.
// The weight may be (N, K) with k-contiguous rows, given as two row blocks (the two directions of a bidirectional GRU
// layer: no concatenated copy), or (K, N) with n-contiguous rows (KMAJOR: dX = dY . Wcat reads Wcat as stored).
#include "mmdfn_internal.h"
#include "../../include/mmdfn_hip.h"
#include <stdlib.h>

namespace {

constexpr int SG_MAX = 8;
constexpr int PRE = 4;            // operand chunks (16 k) in flight per wave (8 measured slower: 102 -> 150 VGPRs)
constexpr int MAXC = 12;          // chunks per wave: K <= 4 * 16 * MAXC = 768

struct SmallGroup {
    int n;
    int act;
    const float* X[SG_MAX];
    const float* W[SG_MAX];
    const float* W2[SG_MAX];
    const float* b[SG_MAX];
    const float* b2[SG_MAX];
    float* Y[SG_MAX];
    int R[SG_MAX], K[SG_MAX], N[SG_MAX], N1[SG_MAX], ldx[SG_MAX], ldw[SG_MAX], ldy[SG_MAX], kmajor[SG_MAX], accumulate[SG_MAX];
    int tile0[SG_MAX + 1];
    int ntn[SG_MAX];
};

// ABL (tuning build, timing only): 1 no MFMA, 2 no operand loads, 4 no reduction / epilogue
template <int ABL>
__global__ __launch_bounds__(256) void linear_small_kernel(SmallGroup G) {
    __shared__ __attribute__((aligned(16))) float part[4][32][36];

    int p = 0;
    while (p + 1 < G.n && (int)blockIdx.x >= G.tile0[p + 1]) ++p;
    const int t = (int)blockIdx.x - G.tile0[p];
    const int ntn = G.ntn[p];
    const int tr = t / ntn, tc = t - tr * ntn;
    const float* __restrict__ X = G.X[p];
    const float* __restrict__ W = G.W[p];
    const float* __restrict__ W2 = G.W2[p];
    const int R = G.R[p], K = G.K[p], N = G.N[p], N1 = G.N1[p], ldx = G.ldx[p], ldw = G.ldw[p], ldy = G.ldy[p];
    const bool kmajor = G.kmajor[p] != 0;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fi = lane & 15, g = lane >> 4;
    const int row0 = 32 * tr, col0 = 32 * tc;

    // this wave's chunks of the contraction
    const int nk = (K + 15) >> 4;
    const int c0 = (w * nk) >> 2, c1 = ((w + 1) * nk) >> 2;
    const int nc = c1 - c0;

    const float* xp[2];
    bool xok[2];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
        const int r = row0 + 16 * rt + fi;
        xok[rt] = r < R;
        xp[rt] = X + (int64_t)(r < R ? r : R - 1) * ldx;
    }
    const float* wp[2];       // NK: the weight row of the lane's column;  KN: column offset into row k
    bool wok[2];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
        const int c = col0 + 16 * ct + fi;
        wok[ct] = c < N;
        const int cc = c < N ? c : N - 1;
        if (kmajor) wp[ct] = W + cc;
        else wp[ct] = (cc < N1) ? W + (int64_t)cc * ldw : W2 + (int64_t)(cc - N1) * ldw;
    }

    f32x4 acc[2][2];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) acc[rt][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};

    float4 a[PRE][2], b[PRE][2];
    if (ABL & 2) {
#pragma unroll
        for (int q = 0; q < PRE; ++q)
#pragma unroll
            for (int e = 0; e < 2; ++e) { a[q][e] = make_float4(1.f, 2.f, 3.f, 4.f); b[q][e] = a[q][e]; }
    }
    auto load_chunk = [&](int slot, int c) {
        if (ABL & 2) return;
        const int k = 16 * (c0 + c) + 4 * g;
        const int kc = k < K ? k : K - 4;
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) a[slot][rt] = *reinterpret_cast<const float4*>(xp[rt] + kc);
        if (kmajor) {
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {
                const float* q = wp[ct] + (int64_t)kc * ldw;
                b[slot][ct] = make_float4(q[0], q[ldw], q[2 * (int64_t)ldw], q[3 * (int64_t)ldw]);
            }
        } else {
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) b[slot][ct] = *reinterpret_cast<const float4*>(wp[ct] + kc);
        }
    };
#pragma unroll
    for (int c = 0; c < PRE; ++c)
        if (c < nc) load_chunk(c, c);
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
        if (c < nc) {             // (wave-uniform; a `break` would keep the loop rolled and the operand ring in scratch)
        const int slot = c % PRE;
        const bool kok = (16 * (c0 + c) + 4 * g) < K;
        float av[2][4], bv[2][4];
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
            const bool ok = kok && xok[rt];
            av[rt][0] = ok ? a[slot][rt].x : 0.f; av[rt][1] = ok ? a[slot][rt].y : 0.f;
            av[rt][2] = ok ? a[slot][rt].z : 0.f; av[rt][3] = ok ? a[slot][rt].w : 0.f;
        }
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
            const bool ok = kok && wok[ct];
            bv[ct][0] = ok ? b[slot][ct].x : 0.f; bv[ct][1] = ok ? b[slot][ct].y : 0.f;
            bv[ct][2] = ok ? b[slot][ct].z : 0.f; bv[ct][3] = ok ? b[slot][ct].w : 0.f;
        }
        if (c + PRE < nc) load_chunk(slot, c + PRE);     // the slot is free once its values sit in av / bv
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int ct = 0; ct < 2; ++ct)
                    if (!(ABL & 1)) acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[rt][j], bv[ct][j], acc[rt][ct], 0, 0, 0);
                    else acc[rt][ct][j] += av[rt][j] * bv[ct][j];
        }
    }

    if (ABL & 4) {
        if (acc[0][0][0] + acc[1][1][1] + acc[0][1][2] + acc[1][0][3] == 1.2345f) G.Y[p][tid] = 1.f;
        return;
    }
    // the four partial tiles meet in LDS (C/D layout of a 16 x 16 tile: column fi, rows 4 g + r)
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int r = 0; r < 4; ++r) part[w][16 * rt + 4 * g + r][16 * ct + fi] = acc[rt][ct][r];
    __syncthreads();
    // epilogue: thread -> (row = tid / 8, four columns 4 (tid % 8) ..); partials summed in wave order
    const int er = tid >> 3, ec = 4 * (tid & 7);
    const int row = row0 + er, col = col0 + ec;
    if (row >= R || col >= N) return;
    float4 s = *reinterpret_cast<const float4*>(&part[0][er][ec]);
#pragma unroll
    for (int q = 1; q < 4; ++q) {
        const float4 v = *reinterpret_cast<const float4*>(&part[q][er][ec]);
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    float o[4] = {s.x, s.y, s.z, s.w};
    const float* bias = G.b[p];
    const float* bias2 = G.b2[p];
    float* y = G.Y[p] + (int64_t)row * ldy + col;
    const bool vec = (col + 3 < N) && ((ldy & 3) == 0) && ((reinterpret_cast<uintptr_t>(G.Y[p]) & 15) == 0);
#pragma unroll
    for (int e = 0; e < 4; ++e)
        if (bias && col + e < N) o[e] += (col + e < N1) ? bias[col + e] : bias2[col + e - N1];
    if (G.accumulate[p]) {
        if (vec) {
            const float4 old = *reinterpret_cast<const float4*>(y);
            o[0] += old.x; o[1] += old.y; o[2] += old.z; o[3] += old.w;
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (col + e < N) o[e] += y[e];
        }
    }
    if (G.act == 1) {
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = fmaxf(o[e], 0.f);
    }
    if (vec) {
        *reinterpret_cast<float4*>(y) = make_float4(o[0], o[1], o[2], o[3]);
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (col + e < N) y[e] = o[e];
    }
}

}  // namespace

// 1: the shape is one the few-row kernel covers
extern "C" int mmdfn_linear_group_supported(int R, int K, int N) {
    return (R > 0 && N > 0 && K >= 4 && (K & 3) == 0 && K <= 4 * 16 * MAXC) ? 1 : 0;
}

extern "C" int mmdfn_linear_group(int n, const float* const* X, const float* const* W, const float* const* W2, const int* N1,
                                  const float* const* bias, const float* const* bias2, float* const* Y, const int* R,
                                  const int* K, const int* N, const int* ldx, const int* ldw, const int* ldy,
                                  const int* kmajor, const int* accumulate, int act, void* stream) {
    if (n <= 0 || n > SG_MAX) return -1;
    SmallGroup G;
    G.n = n;
    G.act = act;
    int t0 = 0;
    for (int p = 0; p < n; ++p) {
        if (!mmdfn_linear_group_supported(R[p], K[p], N[p])) return -1;
        if ((ldx[p] & 3) || ldx[p] < K[p] || ldy[p] < N[p]) return -1;
        if (kmajor[p]) {
            if (ldw[p] < N[p] || N1[p] != N[p]) return -1;
        } else {
            if ((ldw[p] & 3) || ldw[p] < K[p] || N1[p] <= 0 || N1[p] > N[p] || (N1[p] < N[p] && W2[p] == nullptr)) return -1;
            if (bias[p] != nullptr && N1[p] < N[p] && bias2[p] == nullptr) return -1;
        }
        G.X[p] = X[p]; G.W[p] = W[p]; G.W2[p] = W2[p]; G.b[p] = bias[p]; G.b2[p] = bias2[p]; G.Y[p] = Y[p];
        G.R[p] = R[p]; G.K[p] = K[p]; G.N[p] = N[p]; G.N1[p] = N1[p]; G.ldx[p] = ldx[p]; G.ldw[p] = ldw[p]; G.ldy[p] = ldy[p];
        G.kmajor[p] = kmajor[p]; G.accumulate[p] = accumulate[p];
        G.ntn[p] = (N[p] + 31) / 32;
        G.tile0[p] = t0;
        t0 += ((R[p] + 31) / 32) * G.ntn[p];
    }
    for (int p = n; p <= SG_MAX; ++p) G.tile0[p] = t0;
#ifdef MMDFN_TUNING
    if (const char* e = getenv("MMDFN_LSM_ABL")) {
        switch (atoi(e)) {
            case 1: hipLaunchKernelGGL(linear_small_kernel<1>, dim3(t0), dim3(256), 0, (hipStream_t)stream, G); MMDFN_CHECK_LAUNCH(); return 0;
            case 2: hipLaunchKernelGGL(linear_small_kernel<2>, dim3(t0), dim3(256), 0, (hipStream_t)stream, G); MMDFN_CHECK_LAUNCH(); return 0;
            case 4: hipLaunchKernelGGL(linear_small_kernel<4>, dim3(t0), dim3(256), 0, (hipStream_t)stream, G); MMDFN_CHECK_LAUNCH(); return 0;
            case 6: hipLaunchKernelGGL(linear_small_kernel<6>, dim3(t0), dim3(256), 0, (hipStream_t)stream, G); MMDFN_CHECK_LAUNCH(); return 0;
            case 5: hipLaunchKernelGGL(linear_small_kernel<5>, dim3(t0), dim3(256), 0, (hipStream_t)stream, G); MMDFN_CHECK_LAUNCH(); return 0;
            case 7: hipLaunchKernelGGL(linear_small_kernel<7>, dim3(t0), dim3(256), 0, (hipStream_t)stream, G); MMDFN_CHECK_LAUNCH(); return 0;
            default: break;
        }
    }
#endif
    hipLaunchKernelGGL(linear_small_kernel<0>, dim3(t0), dim3(256), 0, (hipStream_t)stream, G);
    MMDFN_CHECK_LAUNCH();
    return 0;
}
