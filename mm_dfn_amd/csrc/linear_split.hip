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
__device__ __forceinline__ f32x4 mfma_bf16_16(u32x4 a, u32x4 b, f32x4 c) {     // (the pipeline's tail tile)
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// One 128 x 128 output block:  Yb[r][c] = act( sum_k Xb[r][k] Wb[c][k] + bias[c] ) (+ Yb[r][c])  for r < Rv, c < Nstore;
// columns Nv <= c < Nstore are written as exact zeros (row padding of the block-tile layout).
// Block columns [0, n1) read rows of Wb / bias, columns [n1, Nv) rows (c - n1) of W2b / bias2 (n1 >= Nv: one block).
__device__ __forceinline__ void split_gemm_block(const float* __restrict__ Xb, const float* __restrict__ Wb,
                                                 const float* __restrict__ W2b, int n1,
                                                 const float* __restrict__ bias, const float* __restrict__ bias2,
                                                 float* __restrict__ Yb, int Rv, int Nv,
                                                 int Nstore, int K, int ldx, int ldw, int ldy, int act, int accumulate,
                                                 bool nt_store = false) {
    constexpr int NCT = 4;
    constexpr int ABLC = 0;
    constexpr int WROWS = 32;
    constexpr int split_stride = 128 * SROW;
    constexpr int stage_stride = 3 * split_stride;
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];

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

    // B staging tasks: thread -> output column (tid & 127), slots (kh = 0 and 1, kg = tid >> 7)
    const int bcol = tid & 127;
    const bool bok = bcol < Nv;
    const int bkg = __builtin_amdgcn_readfirstlane(tid >> 7);
    const int blds = bcol * SROW + 4 * bkg;
    const int bcolc = bok ? bcol : Nv - 1;
    const float* w_lane = ((bcolc < n1) ? Wb + (int64_t)bcolc * ldw : W2b + (int64_t)(bcolc - n1) * ldw) + 4 * bkg;

    const int arow = wrow0 + l32;
    const float* a_lane = Xb + (int64_t)(arow < Rv ? arow : Rv - 1) * ldx + 4 * kg;
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

constexpr bool SPLIT_TAIL = false;     // (four full 32-column tiles)
    f32x4 acct[1];
    const int tboff = 0;
    (void)acct; (void)tboff;
constexpr bool SPLIT_BPRE = false;
#include "split_mfma_pipeline.h"

    // ---- epilogue.  Vector path (16-byte aligned rows): accumulators -> LDS (64 rows per pass) -> four threads per
    // output row write whole float4s (a 32x32 accumulator tile by itself only offers 4-byte stores, 128 bytes per
    // row and instruction).  C/D layout of the tile: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5).
    const bool vec_ok = ((Nstore & 3) == 0) && ((ldy & 3) == 0) && ((reinterpret_cast<uintptr_t>(Yb) & 15) == 0);
    if (vec_ok) {
        constexpr int LDO = 128 + 8;
        float* Os = reinterpret_cast<float*>(smem);
        __syncthreads();                               // the last step's (unused) fragment reloads have retired
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
            if (pass) __syncthreads();
            if ((w >> 1) == pass) {
                const int lrow0 = WROWS * (w & 1) + 4 * kg;
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
                    for (int r = 0; r < 16; ++r) Os[(lrow0 + (r & 3) + 8 * (r >> 2)) * LDO + 32 * ct + l32] = acc[ct][r];
            }
            __syncthreads();
            // 32 consecutive lanes write one row's 128 columns (512 contiguous bytes, whole cache lines per instruction: with
            // four lanes per row a store instruction left 64-byte pieces, which nontemporal stores sent to memory one by one --
            // WRITE_SIZE 279 MB for 197 MB of tiles)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int idx = tid + 256 * j;
                const int er = idx >> 5, c = 4 * (idx & 31);
                const int row = 64 * pass + er;
                if (row >= Rv || c >= Nstore) continue;
                float* yrow = Yb + (int64_t)row * ldy;
                float4 v = *reinterpret_cast<const float4*>(&Os[er * LDO + c]);
                float o[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    o[e] = (c + e < Nv) ? o[e] + (bias ? (c + e < n1 ? bias[c + e] : bias2[c + e - n1]) : 0.f) : 0.f;
                if (accumulate) {
                    const float4 old = *reinterpret_cast<const float4*>(yrow + c);
                    o[0] += old.x; o[1] += old.y; o[2] += old.z; o[3] += old.w;
                }
                if (act == 1) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = fmaxf(o[e], 0.f);
                }
                // (nt_store: results nobody reads soon -- the 201 MB of adjacency-gradient tiles -- go around the L2)
                if (nt_store) __builtin_nontemporal_store((f32x4){o[0], o[1], o[2], o[3]}, reinterpret_cast<f32x4*>(yrow + c));
                else *reinterpret_cast<float4*>(yrow + c) = make_float4(o[0], o[1], o[2], o[3]);
            }
        }
        return;
    }
    // scalar path straight from the accumulators
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) {
        const int c = 32 * ct + l32;
        if (c >= Nstore) continue;
        const float bb = (bias && c < Nv) ? (c < n1 ? bias[c] : bias2[c - n1]) : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = wrow0 + (r & 3) + 8 * (r >> 2) + 4 * kg;
            if (row >= Rv) continue;
            // columns Nv <= c < Nstore: the fast chunk body stages a clamped row there, the value is not a product
            float v = (c < Nv) ? acc[ct][r] + bb : 0.f;
            float* y = Yb + (int64_t)row * ldy + c;
            if (accumulate) v += *y;
            if (act == 1) v = fmaxf(v, 0.f);
            *y = v;
        }
    }
}

__global__ __launch_bounds__(256, 2) void linear_split_kernel(const float* __restrict__ X, const float* __restrict__ W,
                                                              const float* __restrict__ W2, int N1,
                                                              const float* __restrict__ bias, const float* __restrict__ bias2,
                                                              float* __restrict__ Y, int R, int K, int N, int ldx, int ldy,
                                                              int act, int accumulate) {
    // column blocks fastest: the column blocks of one X strip run back to back and share it through L2
    const int nbn = (N + 127) / 128;
    const int bm = blockIdx.x / nbn;
    const int bn = blockIdx.x - bm * nbn;
    const int r0 = bm * 128, c0 = bn * 128;
    const int Rv = (R - r0 < 128) ? R - r0 : 128;
    const int Nv = (N - c0 < 128) ? N - c0 : 128;
    // the block's columns c0 .. c0+Nv-1 against the two weight blocks (rows [0, N1) of W, rows [0, N - N1) of W2)
    const int n1 = N1 - c0;                                   // block-relative boundary (<= 0: all of it is W2)
    const float* Wb = (n1 > 0) ? W + (int64_t)c0 * K : nullptr;
    const float* W2b = (n1 > 0) ? W2 : W2 + (int64_t)(-n1) * K;
    const float* bb = (bias && n1 > 0) ? bias + c0 : nullptr;
    const float* bb2 = bias ? ((n1 > 0) ? bias2 : bias2 + (-n1)) : nullptr;
    if (n1 <= 0)      // entirely in the second block: present it as a single block
        split_gemm_block(X + (int64_t)r0 * ldx, W2b, nullptr, 1 << 30, bb2, nullptr, Y + (int64_t)r0 * ldy + c0, Rv, Nv, Nv, K,
                         ldx, K, ldy, act, accumulate);
    else
        split_gemm_block(X + (int64_t)r0 * ldx, Wb, W2b, n1, bb, bb2, Y + (int64_t)r0 * ldy + c0, Rv, Nv, Nv, K, ldx, K, ldy,
                         act, accumulate);
}

// K6' on the same path: dtiles_{i,m}[p, q] (+)= X[(m,p), :] . Y[(m,q), :]  (the adjacency gradient dA = dOut . H^T on the
// block-tile pattern; replaces the dense (MN x MN) product of SpmmBackward, reference model_GCN.py:178).  One workgroup
// per 128 x 128 block of a tile; XCD mapping blockIdx % 8 == dialogue % 8 as in K6.
__global__ __launch_bounds__(256, 2) void tile_dot_split_kernel(const float* __restrict__ X, const float* __restrict__ Y,
                                                                float* __restrict__ out_tiles,
                                                                const int32_t* __restrict__ dia_len,
                                                                const int32_t* __restrict__ row_start,
                                                                const int64_t* __restrict__ tile_base, int B, int M, int N,
                                                                int K, int ldx, int ldy, int max_rb, int accumulate) {
    const int Rt = max_rb * max_rb;
    const int Rd = M * Rt;
    const int bid = blockIdx.x;
    const int yq = bid >> 3;
    const int i = (yq / Rd) * 8 + (bid & 7);
    if (i >= B) return;
    const int rho = yq % Rd;
    const int m = rho / Rt;
    const int rb = (rho - m * Rt) / max_rb;
    const int cb = (rho - m * Rt) - rb * max_rb;
    const int L = dia_len[i];
    const int r0 = rb * 128, c0 = cb * 128;
    if (r0 >= L || c0 >= L) return;
    const int ld = (L + 3) & ~3;
    const int rs = row_start[i];
    const float* Xm = X + ((int64_t)m * N + rs + r0) * ldx;
    const float* Ym = Y + ((int64_t)m * N + rs + c0) * ldy;
    float* T = out_tiles + tile_base[i] + (int64_t)m * L * ld + (int64_t)r0 * ld + c0;
    const int Rv = (L - r0 < 128) ? L - r0 : 128;
    const int Nv = (L - c0 < 128) ? L - c0 : 128;
    const int Ns = (ld - c0 < 128) ? ld - c0 : 128;
    split_gemm_block(Xm, Ym, nullptr, 1 << 30, nullptr, nullptr, T, Rv, Nv, Ns, K, ldx, ldy, ld, 0, accumulate, true);
}

}  // namespace

// -2: shape not covered (caller falls back to the f32-MFMA kernel)
int mmdfn_launch_linear_split(const float* X, const float* W, const float* W2, int N1, const float* bias,
                              const float* bias2, float* Y, int R, int K, int N, int ldx, int ldy, int act, int accumulate,
                              hipStream_t s) {
    if (K < 8 || (K & 3) || (ldx & 3)) return -2;
    const int lds_bytes = 2 * 3 * 128 * SROW * 4;
    dim3 grid(((R + 127) / 128) * ((N + 127) / 128));
    hipLaunchKernelGGL(linear_split_kernel, grid, dim3(256), lds_bytes, s, X, W, W2, N1, bias, bias2, Y, R, K, N, ldx, ldy, act,
                       accumulate);
    MMDFN_CHECK_LAUNCH();
    return 0;
}

// EPI 0 (dtiles (+)= X . Y^T) only; -2: shape not covered
int mmdfn_launch_tile_dot_split(const float* X, const float* Y, float* out_tiles, const int32_t* dia_len,
                                const int32_t* row_start, const int64_t* tile_base, int B, int M, int N, int K, int ldx,
                                int ldy, int max_len, int accumulate, hipStream_t s) {
    if (K < 8 || (K & 3) || (ldx & 3) || (ldy & 3)) return -2;
    const int max_rb = (max_len + 127) / 128;
    const int lds_bytes = 2 * 3 * 128 * SROW * 4;
    dim3 grid(((B + 7) / 8) * 8 * M * max_rb * max_rb);
    hipLaunchKernelGGL(tile_dot_split_kernel, grid, dim3(256), lds_bytes, s, X, Y, out_tiles, dia_len, row_start, tile_base,
                       B, M, N, K, ldx, ldy, max_rb, accumulate);
    MMDFN_CHECK_LAUNCH();
    return 0;
}
