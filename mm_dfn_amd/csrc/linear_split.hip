// K1 (many-row variant): dense feature projections  Y = act(X W^T + b) (+ Y)  with fp32 results carried by the
// bf16 matrix path.
//
// Replaces nn.Linear / F.linear on the hot path (the hoisted GRU input contractions model.py:866,868, the GCN input
// layer and LSTM gate pre-activations model_GCN.py:454,466) where the row count is large enough to fill the chip
// with 128 x 128 output tiles.  Same arithmetic as propagate_split.hip: every fp32 operand is cut exactly into three
// bf16 pieces and the six piece products of weight >= 2^-16 are issued as v_mfma_f32_32x32x16_bf16 (fp32-level
// error, 2.7x less matrix-pipe time than exact-f32 MFMAs).  X (R, K) and W (N, K) are both k-contiguous, so
//   A = X rows go HBM/L2 -> registers in MFMA layout (four 16-byte loads per lane per 32-wide k chunk),
//   B = W rows (one per output column) are cut once per workgroup (two 16-byte loads per staging task) and parked in
//       LDS as three bf16 [col][k-slot] arrays whose 16-byte reads are the B fragments,
// and the chunk loop is the shared software pipeline of split_mfma_pipeline.h.  The epilogue adds the bias, applies
// ReLU / the accumulate addend and stores straight from the accumulators (a 32x32 tile row is 128 contiguous bytes).
#include "mmdfn_internal.h"
#include "../../include/mmdfn_hip.h"
#include <stdlib.h>

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int SBK = 32;
constexpr int SROW = 20;

__device__ __forceinline__ float as_f(uint32_t u) { return __builtin_bit_cast(float, u); }
__device__ __forceinline__ uint32_t as_u(float f) { return __builtin_bit_cast(uint32_t, f); }

__device__ __forceinline__ f32x16 mfma_bf16(u32x4 a, u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

__global__ __launch_bounds__(256, 2) void linear_split_kernel(const float* __restrict__ X, const float* __restrict__ W,
                                                              const float* __restrict__ bias, float* __restrict__ Y,
                                                              int R, int K, int N, int ldx, int ldy, int act,
                                                              int accumulate) {
    constexpr int NCT = 4;
    constexpr int ABLC = 0;
    constexpr int WROWS = 32;
    constexpr int BM = 4 * WROWS;
    constexpr int CB = 32 * NCT;
    constexpr int split_stride = 128 * SROW;
    constexpr int stage_stride = 3 * split_stride;
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];

    // column blocks fastest: the row blocks of one X strip run back to back and share it through L2
    const int nbn = (N + CB - 1) / CB;
    const int bm = blockIdx.x / nbn;
    const int bn = blockIdx.x - bm * nbn;
    const int r0 = bm * BM;
    const int c0 = bn * CB;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = tid >> 6;
    const int l32 = lane & 31;
    const int kg = lane >> 5;
    const int wrow0 = r0 + WROWS * w;

    f32x16 acc[NCT];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[ct][r] = 0.f;

    // B staging tasks: thread -> output column c0 + (tid & 127), slots (kh = 0 and 1, kg = tid >> 7)
    const int bcol = tid & 127;
    const bool bok = c0 + bcol < N;
    const int bkg = __builtin_amdgcn_readfirstlane(tid >> 7);
    const int blds = bcol * SROW + 4 * bkg;
    const float* w_lane = W + (int64_t)(bok ? c0 + bcol : N - 1) * K + 4 * bkg;

    const int arow = wrow0 + l32;
    const float* a_lane = X + (int64_t)(arow < R ? arow : R - 1) * ldx + 4 * kg;
    int boff[NCT];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) boff[ct] = (32 * ct + l32) * SROW + 4 * kg;

    const int nchunks = (K + SBK - 1) / SBK;
    const int klast = (nchunks - 1) * SBK;
    const int nfull = K / SBK;
    const int limA = K - 4 * kg;
    const int limB = bok ? K - 4 * bkg : -(1 << 30);

    // K % 4 == 0 (checked by the launcher): a float4 starting below K lies fully inside the row
#define SPLIT_ISSUE(SET, K0, SAFE)                                                                         \
    do {                                                                                                   \
        _Pragma("unroll") for (int e = 0; e < 2; ++e)                                                      \
            _Pragma("unroll") for (int h = 0; h < 2; ++h) {                                                \
                const int kb_ = (K0) + 16 * e + 8 * h;   /* + 4 bkg (folded into w_lane) */                \
                const int kc_ = (!(SAFE) || kb_ + 4 * bkg < K) ? kb_ : K - 4 - 4 * bkg;                    \
                const float4 v_ = *reinterpret_cast<const float4*>(w_lane + kc_);                          \
                braw[SET][e][4 * h + 0] = v_.x; braw[SET][e][4 * h + 1] = v_.y;                            \
                braw[SET][e][4 * h + 2] = v_.z; braw[SET][e][4 * h + 3] = v_.w;                            \
            }                                                                                              \
        _Pragma("unroll") for (int f = 0; f < 4; ++f) {                                                    \
            const int ka_ = (K0) + 8 * f;                /* + 4 kg (folded into a_lane) */                 \
            const int kc_ = (!(SAFE) || ka_ + 4 * kg < K) ? ka_ : K - 4 - 4 * kg;                          \
            araw[SET][f] = *reinterpret_cast<const float4*>(a_lane + kc_);                                 \
        }                                                                                                  \
    } while (0)

#include "split_mfma_pipeline.h"

    // ---- epilogue straight from the accumulators: C/D layout of the 32x32 tile: col = lane & 31,
    //      row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5); one row of a tile = 128 contiguous bytes
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) {
        const int c = c0 + 32 * ct + l32;
        if (c >= N) continue;
        const float bb = bias ? bias[c] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = wrow0 + (r & 3) + 8 * (r >> 2) + 4 * kg;
            if (row >= R) continue;
            float v = acc[ct][r] + bb;
            float* y = Y + (int64_t)row * ldy + c;
            if (accumulate) v += *y;
            if (act == 1) v = fmaxf(v, 0.f);
            *y = v;
        }
    }
}

}  // namespace

// -2: shape not covered (caller falls back to the f32-MFMA kernel)
int mmdfn_launch_linear_split(const float* X, const float* W, const float* bias, float* Y, int R, int K, int N, int ldx,
                              int ldy, int act, int accumulate, hipStream_t s) {
    if (K < 8 || (K & 3) || (ldx & 3)) return -2;
    const int lds_bytes = 2 * 3 * 128 * SROW * 4;
    dim3 grid(((R + 127) / 128) * ((N + 127) / 128));
    hipLaunchKernelGGL(linear_split_kernel, grid, dim3(256), lds_bytes, s, X, W, bias, Y, R, K, N, ldx, ldy, act, accumulate);
    MMDFN_CHECK_LAUNCH();
    return 0;
}
