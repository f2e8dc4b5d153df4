// K1p: dense projections  Y = act(X W^T + b) (+ Y)  against a weight that arrives as bf16 PIECE PLANES.
//
// Replaces nn.Linear / F.linear for the hoisted GRU input contractions of both recurrent encoders and their input gradients
// (reference model.py:866,868,1082,1132: `nn.GRU` input products of `lstm_l` / `rnn_parties`; run_train_erc.py:512 is where the
// weights change) at the row counts of the BASELINE configs (1 760 .. 19 008 rows, K = 200 / 600, N = 600 / 200).
//
// Why another form.  The 128 x 128-tile kernels (linear_split.hip) cut BOTH operands in every workgroup, run 7 chunks of a
// software pipeline whose prologue and epilogue weigh as much as the chunks, and put 275 workgroups on 256 CUs: 33 us for a
// product whose matrix time is 4 us ("weak" 7).  A weight changes once per optimizer step, so its three bf16
// pieces are cut ONCE per step (mmdfn_cut_weight_planes, one grouped launch for all registered weights) and stored in MFMA
// B-fragment order; here
//   * a workgroup = 64 rows x 128 columns, wave = 64 rows x ONE 32-column tile (two 32 x 32 accumulators): 2 090 wave tasks of
//     168 MFMAs at 7 040 x 200 -> 600, four workgroups resident per CU (24 KB of LDS, < 128 VGPRs), so the chip fills evenly and
//     the phases of different workgroups overlap;
//   * B fragments (the weight pieces) go L2 -> registers with ONE coalesced 1 KB load per piece and k-step, one phase ahead:
//     no cutting, no LDS, no barrier on the B side;
//   * A (the X rows) is cut once per workgroup -- 1/4 per wave -- into LDS in A-fragment order (16 bytes per lane, conflict
//     free), two k-steps (32 k) per phase, double-buffered, one barrier per phase.
// Arithmetic: the six piece products of weight >= 2^-16 of propagate_split.hip (fp32-level error; exact pieces by truncation).
// Plane layout: planes_common.h; the main loop: planes_pipeline.h (shared with the GCN stack's plane kernels, gcn_planes.hip).
#include "planes_common.h"
#include "../../include/mmdfn_hip.h"

namespace {

constexpr int PL_MAXW = 16;        // weights per cut launch

struct CutTable {
    const float* w1[PL_MAXW];      // first block of the stored matrix
    const float* w2[PL_MAXW];      // second block (null: one block)
    u32x4* planes[PL_MAXW];
    int n1[PL_MAXW], ld[PL_MAXW];
    int N[PL_MAXW], K[PL_MAXW];    // of the B operand: N output columns, K contraction
    int mode[PL_MAXW];             // see mmdfn_cut_weight_planes
    int prefix[PL_MAXW + 1];       // fragment-lane tasks (NT * KS * 64) before weight i
    int n;
};

// element B[n][k] of weight i (0 outside N x K)
__device__ __forceinline__ float cut_element(const CutTable& T, int i, int n, int k) {
    const int N = T.N[i], K = T.K[i], ld = T.ld[i], n1 = T.n1[i];
    if (n >= N || k >= K) return 0.f;
    int row, col;
    const float* base;
    switch (T.mode[i]) {
        case 0:                    // B[n][k] = stored[n][k], stored rows split at n1
            row = n; col = k;
            base = row < n1 ? T.w1[i] : T.w2[i];
            if (row >= n1) row -= n1;
            break;
        case 1:                    // B[n][k] = stored[k][n], stored rows split at n1
            row = k; col = n;
            base = row < n1 ? T.w1[i] : T.w2[i];
            if (row >= n1) row -= n1;
            break;
        case 2:                    // B[n][k] = w1[k][n] (n < n1) | w2[k][n - n1]: two matrices side by side, transposed
            row = k; col = n;
            base = col < n1 ? T.w1[i] : T.w2[i];
            if (col >= n1) col -= n1;
            break;
        default: {                 // 3: as 2 with the contraction index gate-interleaved: k = 4 u + g  <->  stored row g (K / 4) + u
            const int H = K >> 2;
            row = (k & 3) * H + (k >> 2); col = n;
            base = col < n1 ? T.w1[i] : T.w2[i];
            if (col >= n1) col -= n1;
            break;
        }
    }
    return base[(int64_t)row * ld + col];
}

// one thread = one fragment lane (ct, ks, lane): 8 elements in, 3 x 16 bytes out
__global__ __launch_bounds__(256) void cut_planes_kernel(CutTable T) {
    const int g = blockIdx.x * 256 + threadIdx.x;
    if (g >= T.prefix[T.n]) return;
    int i = 0;
    while (i + 1 < T.n && g >= T.prefix[i + 1]) ++i;
    const int t = g - T.prefix[i];
    const int N = T.N[i], K = T.K[i], ld = T.ld[i], n1 = T.n1[i];
    const int KS = (K + 15) >> 4;
    const int lane = t & 63;
    const int f = t >> 6;                      // ct * KS + ks
    const int ct = f / KS, ks = f - ct * KS;
    const int n = 32 * ct + (lane & 31);
    const int k0 = 16 * ks + 8 * (lane >> 5);
    float x[8];
    if (T.mode[i] == 0 && n < N && k0 + 8 <= K && (ld & 3) == 0) {
        // eight consecutive k of one stored row: two 16-byte loads (rows are 16-byte aligned: checked by the launcher)
        const float* rp = (n < n1) ? T.w1[i] + (int64_t)n * ld + k0 : T.w2[i] + (int64_t)(n - n1) * ld + k0;
        const float4 a = *reinterpret_cast<const float4*>(rp), b = *reinterpret_cast<const float4*>(rp + 4);
        x[0] = a.x; x[1] = a.y; x[2] = a.z; x[3] = a.w; x[4] = b.x; x[5] = b.y; x[6] = b.z; x[7] = b.w;
    } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = cut_element(T, i, n, k0 + j);      // (transposed modes: lanes walk a stored row)
    }
    u32x4 p1, p2, p3;
    pl_cut8(x, p1, p2, p3);
    u32x4* dst = T.planes[i] + (int64_t)f * 3 * 64 + lane;
    dst[0] = p1;
    dst[64] = p2;
    dst[128] = p3;
}

// RH = row halves per wave: 2 -> wave = 64 rows x one 32-column tile, workgroup = 64 x 128 (many column tiles: the forward
// products, N = 600); 1 -> wave = 32 rows x one tile, workgroup = 64 x 64 (few column tiles: the input gradients, N = 200 --
// twice the workgroups, half the MFMA chain per wave).  A = the X rows: every wave cuts ONE (row half, k-step) task per phase.
// one workgroup's share of Y = act(X B^T + bias) (+ Y); `bid`: the workgroup's number inside ITS problem's block range
template <int RH>
__device__ __forceinline__ void linear_planes_body(
    u32x4* As, const float* __restrict__ X, const u32x4* __restrict__ planes, const float* __restrict__ bias,
    const float* __restrict__ bias2, int n1, float* __restrict__ Y, int R, int K, int N, int ldx, int ldy, int act,
    int accumulate, int nrb, int ncb, int bid, const float* __restrict__ mask = nullptr, float mscale = 1.0f) {
    constexpr int TPW = 4 / (3 - RH);                   // column tiles per workgroup: RH = 2 -> 4, RH = 1 -> 2
    constexpr int NACC = (RH == 2) ? 1 : 2;
    int rb, cb;
    if (!pl_decode(nrb, ncb, rb, cb, bid)) return;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = tid >> 6;
    const int KS = (K + 15) >> 4;
    const int NPH = (KS + PL_STG - 1) / PL_STG;
    const int NT = (N + 31) >> 5;
    const int ct = TPW * cb + (RH == 2 ? w : (w >> 1));
    const int myh = (RH == 2) ? 0 : (w & 1);
    const bool has_tile = ct < NT;                       // (a wave without a tile multiplies tile 0 and stores nothing)
    const int r0 = rb * PL_BM;

    f32x16 acc[RH][NACC];
#pragma unroll
    for (int h = 0; h < RH; ++h)
#pragma unroll
        for (int c = 0; c < NACC; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[h][c][r] = 0.f;

    const u32x4* bsrc = planes + ((int64_t)(has_tile ? ct : 0) * KS) * 3 * 64 + lane;
    f32x4 raw[2][2];                                     // [phase parity][float4 of the 8 k values]
    // staging task of this wave in every phase: row half w & 1, k-step w >> 1; lane -> (row of the half, k group)
    const int shf = w & 1, sksl = w >> 1;
    const int srow = r0 + 32 * shf + (lane & 31);
    const float* xrow = X + (int64_t)(srow < R ? srow : R - 1) * ldx;
    const int skofs = 16 * sksl + 8 * (lane >> 5);       // k of this lane's first value inside the phase

#define PL_ISSUE_X(PAR, PH)                                                                                 \
    do {                                                                                                    \
        const int k0_ = 16 * PL_STG * (PH) + skofs;                                                         \
        raw[PAR][0] = *reinterpret_cast<const f32x4*>(xrow + (k0_ < K ? k0_ : 0));                          \
        raw[PAR][1] = *reinterpret_cast<const f32x4*>(xrow + (k0_ + 4 < K ? k0_ + 4 : 0));                  \
    } while (0)
#define PL_PARK(PAR, PH, BUF)                                                                               \
    do {                                                                                                    \
        const int k0_ = 16 * PL_STG * (PH) + skofs;                                                         \
        const bool ok0_ = k0_ < K, ok1_ = k0_ + 4 < K;       /* (K % 4 == 0: a float4 below K is inside the row) */ \
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

    if (!has_tile) return;
    // ---- epilogue straight from the accumulators: C layout col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
    const int col = 32 * ct + (lane & 31);
    if (col >= N) return;
    float bv = 0.f;
    if (col < n1) { if (bias) bv = bias[col]; }
    else if (bias2) bv = bias2[col - n1];
    // keep flags of a folded dropout: requested for all the rows first (a load inside the store loop waits a round trip per row)
    float mv[RH][16];
#pragma unroll
    for (int h = 0; h < RH; ++h)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = r0 + 32 * (RH == 2 ? h : myh) + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            mv[h][r] = 1.0f;
            if (mask) mv[h][r] = mask[(int64_t)(row < R ? row : R - 1) * N + col] * mscale;
        }
#pragma unroll
    for (int h = 0; h < RH; ++h)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = r0 + 32 * (RH == 2 ? h : myh) + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            if (row < R) {
                float* yp = Y + (int64_t)row * ldy + col;
                float v = (NACC == 2 ? acc[h][0][r] + acc[h][NACC - 1][r] : acc[h][0][r]) + bv;
                if (act == 1) v = v > 0.f ? v : 0.f;
                v *= mv[h][r];
                if (accumulate) v += *yp;
                *yp = v;
            }
        }
}

template <int RH>
__global__ __launch_bounds__(256, 4) void linear_planes_kernel(
    const float* __restrict__ X, const u32x4* __restrict__ planes, const float* __restrict__ bias,
    const float* __restrict__ bias2, int n1, float* __restrict__ Y, int R, int K, int N, int ldx, int ldy, int act,
    int accumulate, int nrb, int ncb) {
    __shared__ u32x4 As[PL_LDS];
    linear_planes_body<RH>(As, X, planes, bias, bias2, n1, Y, R, K, N, ldx, ldy, act, accumulate, nrb, ncb, (int)blockIdx.x);
}

// Several independent projections in ONE launch (the context and the party encoder's input contractions of a GRU layer, their
// input gradients): problem p owns the blocks [blk0[p], blk0[p + 1]) and keeps the tile form it would take alone.
constexpr int PL_MAXG = 4;
struct PlGroup {
    const float* X[PL_MAXG];
    const u32x4* planes[PL_MAXG];
    const float* bias[PL_MAXG];
    const float* bias2[PL_MAXG];
    float* Y[PL_MAXG];
    const float* mask[PL_MAXG];
    int n1[PL_MAXG], R[PL_MAXG], K[PL_MAXG], N[PL_MAXG], ldx[PL_MAXG], ldy[PL_MAXG], nrb[PL_MAXG], ncb[PL_MAXG], rh[PL_MAXG];
    int blk0[PL_MAXG + 1];
    int n;
};

__global__ __launch_bounds__(256, 4) void linear_planes_group_kernel(PlGroup G, int act, int accumulate, float mscale) {
    __shared__ u32x4 As[PL_LDS];
    int p = 0;
    while (p + 1 < G.n && (int)blockIdx.x >= G.blk0[p + 1]) ++p;
    const int bid = (int)blockIdx.x - G.blk0[p];
    if (G.rh[p] == 2)
        linear_planes_body<2>(As, G.X[p], G.planes[p], G.bias[p], G.bias2[p], G.n1[p], G.Y[p], G.R[p], G.K[p], G.N[p], G.ldx[p],
                              G.ldy[p], act, accumulate, G.nrb[p], G.ncb[p], bid, G.mask[p], mscale);
    else
        linear_planes_body<1>(As, G.X[p], G.planes[p], G.bias[p], G.bias2[p], G.n1[p], G.Y[p], G.R[p], G.K[p], G.N[p], G.ldx[p],
                              G.ldy[p], act, accumulate, G.nrb[p], G.ncb[p], bid, G.mask[p], mscale);
}

// tile form of one problem: few column tiles (an input gradient, N = 200) or few rows -> 64 x 64 workgroups, so that the launch
// has workgroups for every CU
inline void pl_form(int R, int N, int* nrb, int* ncb, int* rh) {
    *nrb = (R + PL_BM - 1) / PL_BM;
    const int NT = (N + 31) / 32;
    const bool narrow = (int64_t)*nrb * ((NT + 3) / 4) < 400;
    *ncb = narrow ? (NT + 1) / 2 : (NT + 3) / 4;
    *rh = narrow ? 1 : 2;
}

}  // namespace

extern "C" {

int64_t mmdfn_weight_planes_workspace(int N, int K) {
    if (N <= 0 || K <= 0) return -1;
    const int64_t NT = (N + 31) / 32, KS = (K + 15) / 16;
    return NT * KS * 3 * 64 * 16;                 // bytes
}

int mmdfn_cut_weight_planes(int n, const float* const* w1, const float* const* w2, const int* n1, const int* ld,
                            const int* N, const int* K, const int* mode, void* const* planes, void* stream) {
    if (n <= 0 || n > PL_MAXW) return -1;
    CutTable T;
    T.n = n;
    T.prefix[0] = 0;
    for (int i = 0; i < n; ++i) {
        if (!w1[i] || !planes[i] || N[i] <= 0 || K[i] <= 0 || mode[i] < 0 || mode[i] > 3) return -1;
        if ((reinterpret_cast<uintptr_t>(planes[i]) & 15) != 0) return -1;
        if ((ld[i] & 3) == 0 && ((reinterpret_cast<uintptr_t>(w1[i]) & 15) || (w2[i] && (reinterpret_cast<uintptr_t>(w2[i]) & 15))))
            return -1;                             // (16-byte rows are fetched with 16-byte loads)
        // the index that is split at n1: stored rows (modes 0, 1) or output columns (modes 2, 3)
        const int split_extent = mode[i] == 0 ? N[i] : mode[i] == 1 ? K[i] : N[i];
        if (n1[i] < split_extent && !w2[i]) return -1;
        if (mode[i] == 3 && (K[i] & 3)) return -1;
        T.w1[i] = w1[i];
        T.w2[i] = w2[i];
        T.planes[i] = reinterpret_cast<u32x4*>(planes[i]);
        T.n1[i] = n1[i];
        T.ld[i] = ld[i];
        T.N[i] = N[i];
        T.K[i] = K[i];
        T.mode[i] = mode[i];
        const int64_t tasks = (int64_t)((N[i] + 31) / 32) * ((K[i] + 15) / 16) * 64;
        if (T.prefix[i] + tasks > (1ll << 30)) return -1;
        T.prefix[i + 1] = T.prefix[i] + (int)tasks;
    }
    hipLaunchKernelGGL(cut_planes_kernel, dim3((T.prefix[n] + 255) / 256), dim3(256), 0, (hipStream_t)stream, T);
    MMDFN_CHECK_LAUNCH();
    return 0;
}

int mmdfn_linear_planes(const float* X, const void* planes, const float* bias, const float* bias2, int n1, float* Y, int R,
                        int K, int N, int ldx, int ldy, int act, int accumulate, void* stream) {
    if (R <= 0) return 0;
    if (!X || !planes || !Y || K < 4 || (K & 3) || N <= 0 || (ldx & 3) || ldx < K || ldy < N) return -1;
    if ((reinterpret_cast<uintptr_t>(X) & 15) || (reinterpret_cast<uintptr_t>(planes) & 15)) return -1;
    int nrb, ncb, rh;
    pl_form(R, N, &nrb, &ncb, &rh);
    const bool narrow = rh == 1;
    const int64_t grid = pl_grid(nrb, ncb);
    if (grid > (1ll << 30)) return -1;
    if (narrow)
        hipLaunchKernelGGL(linear_planes_kernel<1>, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, X,
                           reinterpret_cast<const u32x4*>(planes), bias, bias2, n1, Y, R, K, N, ldx, ldy, act, accumulate, nrb, ncb);
    else
        hipLaunchKernelGGL(linear_planes_kernel<2>, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, X,
                           reinterpret_cast<const u32x4*>(planes), bias, bias2, n1, Y, R, K, N, ldx, ldy, act, accumulate, nrb, ncb);
    MMDFN_CHECK_LAUNCH();
    return 0;
}

int mmdfn_linear_planes_group(int n, const float* const* X, const void* const* planes, const float* const* bias,
                              const float* const* bias2, const int* n1, float* const* Y, const int* R, const int* K,
                              const int* N, const int* ldx, const int* ldy, int act, int accumulate,
                              const float* const* mask, float mask_scale, void* stream) {
    if (n <= 0 || n > PL_MAXG) return -1;
    PlGroup G;
    G.n = 0;
    G.blk0[0] = 0;
    for (int i = 0; i < n; ++i) {
        if (R[i] <= 0) continue;                   // (an empty problem takes no blocks)
        if (!X[i] || !planes[i] || !Y[i] || K[i] < 4 || (K[i] & 3) || N[i] <= 0 || (ldx[i] & 3) || ldx[i] < K[i] || ldy[i] < N[i])
            return -1;
        if ((reinterpret_cast<uintptr_t>(X[i]) & 15) || (reinterpret_cast<uintptr_t>(planes[i]) & 15)) return -1;
        const int p = G.n++;
        G.X[p] = X[i];
        G.planes[p] = reinterpret_cast<const u32x4*>(planes[i]);
        G.bias[p] = bias ? bias[i] : nullptr;
        G.bias2[p] = bias2 ? bias2[i] : nullptr;
        G.Y[p] = Y[i];
        G.mask[p] = mask ? mask[i] : nullptr;
        G.n1[p] = n1[i]; G.R[p] = R[i]; G.K[p] = K[i]; G.N[p] = N[i]; G.ldx[p] = ldx[i]; G.ldy[p] = ldy[i];
        pl_form(R[i], N[i], &G.nrb[p], &G.ncb[p], &G.rh[p]);
        const int64_t blocks = pl_grid(G.nrb[p], G.ncb[p]);
        if (G.blk0[p] + blocks > (1ll << 30)) return -1;
        G.blk0[p + 1] = G.blk0[p] + (int)blocks;
    }
    if (G.n == 0) return 0;
    for (int p = G.n; p < PL_MAXG; ++p) {
        G.X[p] = nullptr; G.planes[p] = nullptr; G.bias[p] = G.bias2[p] = nullptr; G.Y[p] = nullptr; G.mask[p] = nullptr;
        G.n1[p] = G.R[p] = G.K[p] = G.N[p] = G.ldx[p] = G.ldy[p] = G.nrb[p] = G.ncb[p] = G.rh[p] = 0;
        G.blk0[p + 1] = G.blk0[G.n];
    }
    hipLaunchKernelGGL(linear_planes_group_kernel, dim3((unsigned)G.blk0[G.n]), dim3(256), 0, (hipStream_t)stream, G, act,
                       accumulate, mask_scale);
    MMDFN_CHECK_LAUNCH();
    return 0;
}

}  // extern "C"
