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
//     156 MFMAs at 7 040 x 200 -> 600, 3-4 workgroups resident per CU (43 KB of LDS, < 128 VGPRs), so the chip fills evenly and
//     the phases of different workgroups overlap without a software pipeline;
//   * B fragments (the weight pieces) go L2 -> registers with ONE coalesced 1 KB load per piece and k-step, one k-step ahead:
//     no cutting, no LDS, no barrier on the B side;
//   * A (the X rows) is cut once per workgroup -- 1/4 per wave -- into LDS in A-fragment order (16 bytes per lane, conflict
//     free), 7 k-steps (112 k) per phase.
// Arithmetic: the six piece products of weight >= 2^-16 of propagate_split.hip (fp32-level error; exact pieces by truncation).
//
// Plane layout of a weight "B[n][k]" (n < N output columns, k < K contraction):  KS = ceil(K / 16) k-steps, NT = ceil(N / 32)
// column tiles;  planes[((ct * KS + ks) * 3 + piece) * 64 + lane] = 8 bf16 (16 bytes) = piece `piece` of
// B[32 ct + (lane & 31)][16 ks + 8 (lane >> 5) + 0..7], zero outside N x K: exactly the B operand of v_mfma_f32_32x32x16_bf16.
#include "mmdfn_internal.h"
#include "../../include/mmdfn_hip.h"
#include <stdlib.h>
#include <type_traits>

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int PL_BM = 64;          // rows per workgroup
constexpr int PL_STG = 2;          // k-steps per phase (32 k): 2 buffers x 2 x 3 pieces x 2 row halves x 1 KB = 24 KB of LDS
constexpr int PL_MAXW = 16;        // weights per cut launch

__device__ __forceinline__ float as_f(uint32_t u) { return __builtin_bit_cast(float, u); }
__device__ __forceinline__ uint32_t as_u(float f) { return __builtin_bit_cast(uint32_t, f); }
__device__ __forceinline__ f32x16 mfma_bf16(u32x4 a, u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// eight consecutive fp32 values -> three u32x4 of packed bf16 pieces (x = p1 + p2 + p3 exactly, by truncation)
__device__ __forceinline__ void cut8(const float (&x)[8], u32x4& p1, u32x4& p2, u32x4& p3) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float a = x[2 * j], b = x[2 * j + 1];
        p1[j] = __builtin_amdgcn_perm(as_u(b), as_u(a), 0x07060302u);
        a -= as_f(as_u(a) & 0xffff0000u);
        b -= as_f(as_u(b) & 0xffff0000u);
        p2[j] = __builtin_amdgcn_perm(as_u(b), as_u(a), 0x07060302u);
        a -= as_f(as_u(a) & 0xffff0000u);
        b -= as_f(as_u(b) & 0xffff0000u);
        p3[j] = __builtin_amdgcn_perm(as_u(b), as_u(a), 0x07060302u);
    }
}

struct CutTable {
    const float* w1[PL_MAXW];      // rows [0, n1) of the stored matrix
    const float* w2[PL_MAXW];      // rows [n1, ...) (null: one block)
    u32x4* planes[PL_MAXW];
    int n1[PL_MAXW], ld[PL_MAXW];
    int N[PL_MAXW], K[PL_MAXW];    // of the B operand: N output columns, K contraction
    int transposed[PL_MAXW];       // 0: B[n][k] = stored[n][k];  1: B[n][k] = stored[k][n]  (the input gradient's operand)
    int prefix[PL_MAXW + 1];       // fragment-lane tasks (NT * KS * 64) before weight i
    int n;
};

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
    if (!T.transposed[i] && n < N && k0 + 8 <= K && (ld & 3) == 0) {
        // eight consecutive k of one stored row: two 16-byte loads (rows are 16-byte aligned: checked by the launcher)
        const float* rp = (n < n1) ? T.w1[i] + (int64_t)n * ld + k0 : T.w2[i] + (int64_t)(n - n1) * ld + k0;
        const float4 a = *reinterpret_cast<const float4*>(rp), b = *reinterpret_cast<const float4*>(rp + 4);
        x[0] = a.x; x[1] = a.y; x[2] = a.z; x[3] = a.w; x[4] = b.x; x[5] = b.y; x[6] = b.z; x[7] = b.w;
    } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = k0 + j;
            float v = 0.f;
            if (n < N && k < K) {
                const int row = T.transposed[i] ? k : n;      // (transposed: lanes walk a stored row -> coalesced)
                const int col = T.transposed[i] ? n : k;
                v = (row < n1) ? T.w1[i][(int64_t)row * ld + col] : T.w2[i][(int64_t)(row - n1) * ld + col];
            }
            x[j] = v;
        }
    }
    u32x4 p1, p2, p3;
    cut8(x, p1, p2, p3);
    u32x4* dst = T.planes[i] + (int64_t)f * 3 * 64 + lane;
    dst[0] = p1;
    dst[64] = p2;
    dst[128] = p3;
}

// grid: 8 * ceil(row blocks / 8) * ncb workgroups; blockIdx % 8 (the XCD) owns the row blocks = its number (mod 8) and runs the
// ncb column blocks of a row block back to back, so the X rows of a row block are fetched into ONE L2.
// RH = row halves per wave: 2 -> wave = 64 rows x one 32-column tile, workgroup = 64 x 128 (many column tiles: the forward
// products, N = 600); 1 -> wave = 32 rows x one tile, workgroup = 64 x 64 (few column tiles: the input gradients, N = 200 --
// twice the workgroups, half the MFMA chain per wave).
//
// Phases of PL_STG = 2 k-steps, everything double-buffered with STATIC indices (the phase loop is unrolled by two):
//   * vector-memory results retire IN ORDER, so a B fragment requested behind an X load cannot be used before that X load has
//     landed: the B fragments of phase p + 1 are requested during phase p (one full phase ahead, never waited for inside the
//     phase that requests them), and the X rows of phase p + 2 are requested at the start of phase p, BEFORE them -- by the time a
//     fragment of phase p + 1 is needed, the X request in front of it is a whole phase old.  (The first version requested X rows
//     one phase ahead and fragments two k-steps ahead: every phase stalled on the X latency at its third k-step.)
//   * the X rows of phase p + 1 are cut and parked in the other LDS buffer behind phase p's last MFMA; ONE barrier per phase.
// HAND = true: the requests are asm statements with hand-counted waits (3 waves per SIMD: no spill may ever sit between a request
// and its wait -- checked at build time, mm_dfn_amd/build.py); HAND = false: plain loads, hipcc's own schedule and waits (it sinks
// the requests to their first use; 4 waves per SIMD).  The launcher picks by measurement (tools/bench_linear_planes.py).
template <int RH, bool HAND>
__global__ __launch_bounds__(256, HAND ? 3 : 4) void linear_planes_kernel(
    const float* __restrict__ X, const u32x4* __restrict__ planes, const float* __restrict__ bias,
    const float* __restrict__ bias2, int n1, float* __restrict__ Y, int R, int K, int N, int ldx, int ldy, int act,
    int accumulate, int nrb, int ncb) {
    __shared__ u32x4 As[2][PL_STG * 3 * 2 * 64];       // [buffer][k-step in phase][piece][row half][lane]
    constexpr int TPW = 4 / (3 - RH);                   // column tiles per workgroup: RH = 2 -> 4, RH = 1 -> 2
    constexpr int NACC = (RH == 2) ? 1 : 2;             // accumulators per row half: consecutive MFMAs never share one
    const int bid = blockIdx.x;
    const int yq = bid >> 3;
    const int rb = (yq / ncb) * 8 + (bid & 7);
    if (rb >= nrb) return;
    const int cb = yq % ncb;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = tid >> 6;
    const int KS = (K + 15) >> 4;
    const int NPH = (KS + PL_STG - 1) / PL_STG;
    const int NT = (N + 31) >> 5;
    const int ct = TPW * cb + (RH == 2 ? w : (w >> 1));
    const int myh = (RH == 2) ? 0 : (w & 1);             // (RH = 1) the row half this wave multiplies
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
    u32x4 bq[2][PL_STG][3];                              // [phase parity][k-step][piece]
    f32x4 raw[2][2];                                     // [phase parity][float4 of the 8 k values]: this wave's ONE staging task
    // staging task of this wave in every phase: row half w & 1, k-step w >> 1; lane -> (row of the half, k group)
    const int shf = w & 1, sksl = w >> 1;
    const int srow = r0 + 32 * shf + (lane & 31);
    const float* xrow = X + (int64_t)(srow < R ? srow : R - 1) * ldx;
    const int skofs = 16 * sksl + 8 * (lane >> 5);       // k of this lane's first value inside the phase

    // Every vector-memory request of the loop is an asm statement and every wait is hand-counted: left to hipcc the requests
    // sink to their first use (right in front of the phase's barrier) and each phase opens by waiting for them.  A phase
    // issues exactly PL_NLD requests (2 X + 6 B); vector-memory results return in order, so `vmcnt(PL_NLD)` behind a phase's own
    // requests means "everything requested in earlier phases has landed".
#define PL_GLOAD(DST, PTR)                                                                                  \
    do {                                                                                                    \
        if constexpr (HAND) asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(DST) : "v"(PTR) : "memory"); \
        else DST = *reinterpret_cast<const std::remove_reference_t<decltype(DST)>*>(PTR);                                            \
    } while (0)
#define PL_ISSUE_X(PAR, PH)                                                                                 \
    do {                                                                                                    \
        const int k0_ = 16 * PL_STG * (PH) + skofs;                                                         \
        const float* xa_ = xrow + (k0_ < K ? k0_ : 0);                                                      \
        const float* xb_ = xrow + (k0_ + 4 < K ? k0_ + 4 : 0);                                              \
        PL_GLOAD(raw[PAR][0], xa_);                                                                         \
        PL_GLOAD(raw[PAR][1], xb_);                                                                         \
    } while (0)
#define PL_PARK(PAR, PH, BUF)                                                                               \
    do {                                                                                                    \
        const int k0_ = 16 * PL_STG * (PH) + skofs;                                                         \
        const bool ok0_ = k0_ < K, ok1_ = k0_ + 4 < K;       /* (K % 4 == 0: a float4 below K is inside the row) */ \
        float x_[8] = {ok0_ ? raw[PAR][0].x : 0.f, ok0_ ? raw[PAR][0].y : 0.f, ok0_ ? raw[PAR][0].z : 0.f, ok0_ ? raw[PAR][0].w : 0.f, \
                       ok1_ ? raw[PAR][1].x : 0.f, ok1_ ? raw[PAR][1].y : 0.f, ok1_ ? raw[PAR][1].z : 0.f, ok1_ ? raw[PAR][1].w : 0.f}; \
        u32x4 p1_, p2_, p3_;                                                                                \
        cut8(x_, p1_, p2_, p3_);                                                                            \
        u32x4* dst_ = &As[BUF][((sksl * 3) * 2 + shf) * 64 + lane];                                         \
        dst_[0] = p1_;                                                                                      \
        dst_[2 * 64] = p2_;                                                                                 \
        dst_[4 * 64] = p3_;                                                                                 \
    } while (0)
#define PL_ISSUE_B(PAR, PH)                                                                                 \
    do {                                                                                                    \
        _Pragma("unroll") for (int j_ = 0; j_ < PL_STG; ++j_) {                                             \
            const int ks_ = PL_STG * (PH) + j_ < KS ? PL_STG * (PH) + j_ : KS - 1;                           \
            _Pragma("unroll") for (int p_ = 0; p_ < 3; ++p_) {                                              \
                const u32x4* bp_ = bsrc + ((int64_t)ks_ * 3 + p_) * 64;                                     \
                PL_GLOAD(bq[PAR][j_][p_], bp_);                                                             \
            }                                                                                               \
        }                                                                                                   \
    } while (0)
    // "everything requested before this phase's own PL_NLD requests has landed": the fragments of phase PAR and the X rows
    // parked at the end of this phase (tied to the statement so that no use is scheduled in front of it)
#define PL_WAIT_OLDER(PAR)                                                                                  \
    do {                                                                                                    \
        if constexpr (HAND)                                                                                 \
            asm volatile("s_waitcnt vmcnt(8)"                                                               \
                         : "+v"(bq[PAR][0][0]), "+v"(bq[PAR][0][1]), "+v"(bq[PAR][0][2]), "+v"(bq[PAR][1][0]), "+v"(bq[PAR][1][1]), \
                           "+v"(bq[PAR][1][2]), "+v"(raw[(PAR) ^ 1][0]), "+v"(raw[(PAR) ^ 1][1])            \
                         : : "memory");                                                                     \
    } while (0)
    // one phase: request X of phase PH + 2 and the fragments of phase PH + 1, multiply phase PH, park X of phase PH + 1
#define PL_PHASE(PAR, PH)                                                                                   \
    do {                                                                                                    \
        /* (unconditional, clamped: the loads of a phase past the end re-read the last one and are never used; a k-step past  */ \
        /* KS multiplies zero A pieces -- parked as zeros beyond K -- by the last real fragments; a wave without a column tile  */ \
        /* multiplies tile 0 and stores nothing: no branch inside the phase)                                                    */ \
        PL_ISSUE_X(PAR, (PH) + 2 < NPH ? (PH) + 2 : NPH - 1);                                               \
        PL_ISSUE_B((PAR) ^ 1, (PH) + 1 < NPH ? (PH) + 1 : NPH - 1);                                         \
        /* the requests lead the phase (hipcc otherwise sinks them behind the MFMAs, right in front of the barrier, and the   */ \
        /* next phase opens by waiting for them); then ALL A fragments of the phase are requested before its first MFMA        */ \
        u32x4 a_[PL_STG][RH][3];                                                                            \
        _Pragma("unroll") for (int j_ = 0; j_ < PL_STG; ++j_)                                               \
            _Pragma("unroll") for (int p_ = 0; p_ < 3; ++p_)                                                \
                _Pragma("unroll") for (int h_ = 0; h_ < RH; ++h_)                                           \
                    a_[j_][h_][p_] = As[PAR][((j_ * 3 + p_) * 2 + (RH == 2 ? h_ : myh)) * 64 + lane];       \
        PL_WAIT_OLDER(PAR);                                                                                 \
        _Pragma("unroll") for (int j_ = 0; j_ < PL_STG; ++j_) {                                             \
            /* products: against b1: a3 a2 a1;  against b2: a2 a1;  against b3: a1 */                      \
            _Pragma("unroll") for (int q_ = 0; q_ < 3; ++q_) {                                              \
                const int ahi_ = 2 - q_, alo_ = (q_ == 0) ? 1 : 0, blo_ = (q_ < 2) ? 1 : 2;                 \
                _Pragma("unroll") for (int h_ = 0; h_ < RH; ++h_)                                           \
                    acc[h_][0] = mfma_bf16(a_[j_][h_][ahi_], bq[PAR][j_][0], acc[h_][0]);                   \
                _Pragma("unroll") for (int h_ = 0; h_ < RH; ++h_)                                           \
                    acc[h_][NACC - 1] = mfma_bf16(a_[j_][h_][alo_], bq[PAR][j_][blo_], acc[h_][NACC - 1]);  \
            }                                                                                               \
        }                                                                                                   \
        __builtin_amdgcn_sched_barrier(0);                                                                  \
        if ((PH) + 1 < NPH) {                                                                               \
            PL_PARK((PAR) ^ 1, (PH) + 1, (PAR) ^ 1);    /* (the other buffer's readers passed the previous barrier) */ \
            __syncthreads();                                                                                \
        }                                                                                                   \
    } while (0)

    static_assert(PL_STG == 2, "PL_WAIT_OLDER counts 2 X + 3 * PL_STG B requests per phase");
    PL_ISSUE_X(0, 0);
    PL_ISSUE_X(1, NPH > 1 ? 1 : 0);
    PL_ISSUE_B(0, 0);
    if constexpr (HAND) asm volatile("s_waitcnt vmcnt(8)" : "+v"(raw[0][0]), "+v"(raw[0][1]) : : "memory");   // X of phase 0 (2 + 6 requests behind it)
    PL_PARK(0, 0, 0);
    __syncthreads();
    for (int ph = 0; ph < NPH; ph += 2) {
        PL_PHASE(0, ph);
        if (ph + 1 < NPH) PL_PHASE(1, ph + 1);
    }
#undef PL_PHASE
#undef PL_WAIT_OLDER
#undef PL_GLOAD
#undef PL_ISSUE_B
#undef PL_PARK
#undef PL_ISSUE_X
    if (!has_tile) return;
    // ---- epilogue straight from the accumulators: C layout col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
    const int col = 32 * ct + (lane & 31);
    if (col >= N) return;
    float bv = 0.f;
    if (col < n1) { if (bias) bv = bias[col]; }
    else if (bias2) bv = bias2[col - n1];
#pragma unroll
    for (int h = 0; h < RH; ++h)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = r0 + 32 * (RH == 2 ? h : myh) + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            if (row < R) {
                float* yp = Y + (int64_t)row * ldy + col;
                float v = (NACC == 2 ? acc[h][0][r] + acc[h][NACC - 1][r] : acc[h][0][r]) + bv;
                if (act == 1) v = v > 0.f ? v : 0.f;
                if (accumulate) v += *yp;
                *yp = v;
            }
        }
}

}  // namespace

extern "C" {

int64_t mmdfn_weight_planes_workspace(int N, int K) {
    if (N <= 0 || K <= 0) return -1;
    const int64_t NT = (N + 31) / 32, KS = (K + 15) / 16;
    return NT * KS * 3 * 64 * 16;                 // bytes
}

int mmdfn_cut_weight_planes(int n, const float* const* w1, const float* const* w2, const int* n1, const int* ld,
                            const int* N, const int* K, const int* transposed, void* const* planes, void* stream) {
    if (n <= 0 || n > PL_MAXW) return -1;
    CutTable T;
    T.n = n;
    T.prefix[0] = 0;
    for (int i = 0; i < n; ++i) {
        if (!w1[i] || !planes[i] || N[i] <= 0 || K[i] <= 0) return -1;
        if ((reinterpret_cast<uintptr_t>(planes[i]) & 15) != 0) return -1;
        if ((ld[i] & 3) == 0 && ((reinterpret_cast<uintptr_t>(w1[i]) & 15) || (w2[i] && (reinterpret_cast<uintptr_t>(w2[i]) & 15))))
            return -1;                             // (16-byte rows are fetched with 16-byte loads)
        const int rows = transposed[i] ? K[i] : N[i];
        if (n1[i] < rows && !w2[i]) return -1;
        T.w1[i] = w1[i];
        T.w2[i] = w2[i];
        T.planes[i] = reinterpret_cast<u32x4*>(planes[i]);
        T.n1[i] = n1[i];
        T.ld[i] = ld[i];
        T.N[i] = N[i];
        T.K[i] = K[i];
        T.transposed[i] = transposed[i];
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
    const int nrb = (R + PL_BM - 1) / PL_BM;
    const int NT = (N + 31) / 32;
    // few column tiles (an input gradient, N = 200) or few rows: 64 x 64 workgroups, so that the launch has workgroups for every CU
    const bool narrow = (int64_t)nrb * ((NT + 3) / 4) < 400;
    const int ncb = narrow ? (NT + 1) / 2 : (NT + 3) / 4;
    const int64_t grid = (int64_t)((nrb + 7) / 8) * 8 * ncb;
    if (grid > (1ll << 30)) return -1;
    // same-box A/B inside the cfg2 step (tools/bench_linear_planes.py, bench.py through the tuning library, three alternating
    // runs): 0.9640 / 0.9636 / 0.9630 ms with the hand-counted form, 0.9605 / 0.9674 / 0.9619 with hipcc's schedule (0.9816 /
    // 0.9825 / 0.9798 without the plane form); isolated launches: 25.2 vs 21.9 us forward, 22.4 vs 23.4 us input gradient at
    // 7 040 rows.  Equal in the step; hipcc's form keeps 4 waves per SIMD and needs no spill guard, so it ships.
    bool hand = false;
#ifdef MMDFN_TUNING
    if (const char* e = getenv("MMDFN_PLANES_HAND")) hand = e[0] != '0';      // A/B aid
#endif
#define PL_LAUNCH(RH_, HAND_)                                                                                 \
    hipLaunchKernelGGL((linear_planes_kernel<RH_, HAND_>), dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, X, \
                       reinterpret_cast<const u32x4*>(planes), bias, bias2, n1, Y, R, K, N, ldx, ldy, act, accumulate, nrb, ncb)
    if (narrow && hand) PL_LAUNCH(1, true);
    else if (narrow) PL_LAUNCH(1, false);
    else if (hand) PL_LAUNCH(2, true);
    else PL_LAUNCH(2, false);
#undef PL_LAUNCH
    MMDFN_CHECK_LAUNCH();
    return 0;
}

}  // extern "C"
