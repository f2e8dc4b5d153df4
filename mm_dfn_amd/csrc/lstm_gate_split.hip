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
#include "../../include/mmdfn_hip.h"

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

// the gate non-linearities of gcn_stack.hip (hardware exp / rcp forms, |err| < 3e-7)
__device__ __forceinline__ float sigm(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float tanhf_(float x) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(2.0f * x)); }

// Piece planes of [W_ih | W_hh] for the BPRE form (the cell is shared by all layers of the stack and constant inside a step,
// model_GCN.py:466: cut ONCE per step by lstm_gate_cut_kernel instead of by every workgroup of every layer's launch).  Layout =
// what a staging thread stores to LDS: [unit block][chunk of 32 k][piece][k-slot 2 kh + kg][accumulator column (gate, unit)] x
// 16 bytes (the 8 k of the slot in the pipeline's order k = 32 c + 16 kh + 4 kg + (j & 3) + 8 (j >> 2)).
constexpr int GPL_SLOT = 128;                      // u32x4 per (chunk, piece, k-slot)
constexpr int GPL_CHUNK = 3 * 4 * GPL_SLOT;        // u32x4 per chunk

__global__ __launch_bounds__(256) void lstm_gate_cut_kernel(const float* __restrict__ Wih, const float* __restrict__ Whh,
                                                            u32x4* __restrict__ planes, int H, int nchunks, int total) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int bcol = idx & 127;
    const int slot = (idx >> 7) & 3;
    const int cc = (idx >> 9) % nchunks;
    const int ub = (idx >> 9) / nchunks;
    const int gate = bcol >> 5, unit = 32 * ub + (bcol & 31);
    const int e = slot >> 1, bkg = slot & 1;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int k = 32 * cc + 16 * e + 4 * bkg + (j & 3) + 8 * (j >> 2);
        const int64_t row = (int64_t)(gate * H + unit) * H;
        v[j] = (unit < H && k < 2 * H) ? (k < H ? Wih[row + k] : Whh[row + k - H]) : 0.f;
    }
    uint32_t pc[3][4];
    float x[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) x[j] = v[j];
#pragma unroll
    for (int q = 0; q < 3; ++q) {
#pragma unroll
        for (int p2 = 0; p2 < 4; ++p2) pc[q][p2] = __builtin_amdgcn_perm(as_u(x[2 * p2 + 1]), as_u(x[2 * p2]), 0x07060302u);
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] -= as_f(as_u(x[j]) & 0xffff0000u);
    }
    u32x4* dst = planes + ((int64_t)(ub * nchunks + cc) * 3 * 4 + slot) * GPL_SLOT + bcol;
#pragma unroll
    for (int q = 0; q < 3; ++q) dst[q * 4 * GPL_SLOT] = u32x4{pc[q][0], pc[q][1], pc[q][2], pc[q][3]};
}

// BPRE: the weight pieces come from the planes above (h != null only: K = 2 H)
template <bool BPRE>
__global__ __launch_bounds__(256, 2) void lstm_gate_fwd_split_kernel(
    const float* __restrict__ q, const float* __restrict__ h, const float* __restrict__ c, const float* __restrict__ Wih,
    const float* __restrict__ Whh, const float* __restrict__ bsum, const float* __restrict__ bsum2, float* __restrict__ gates,
    float* __restrict__ h_out, float* __restrict__ c_out, int R, int H, int ldh, const u32x4* __restrict__ planes) {
    constexpr bool SPLIT_BPRE = BPRE;
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
    const u32x4* planes_lane = BPRE ? planes + (int64_t)ub * ((2 * H + SBK - 1) / SBK) * GPL_CHUNK + bcol : nullptr;
#define SPLIT_ISSUE(SET, K0, SAFE)                                                                         \
    do {                                                                                                   \
        if constexpr (BPRE) {                                                                              \
            const u32x4* pp_ = planes_lane + (int64_t)((K0) / SBK) * GPL_CHUNK + bkg * GPL_SLOT;           \
            _Pragma("unroll") for (int e = 0; e < 2; ++e)                                                  \
                _Pragma("unroll") for (int q_ = 0; q_ < 3; ++q_)                                           \
                    bpre[SET][e][q_] = pp_[(q_ * 4 + 2 * e) * GPL_SLOT];                                   \
        } else                                                                                             \
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

constexpr bool SPLIT_TAIL = false;     // (four full 32-column tiles)
    f32x4 acct[1];
    const int tboff = 0;
    (void)acct; (void)tboff;
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

}  // namespace

// -2: shape not covered (the caller keeps the exact-f32 kernels of gcn_stack.hip)
int mmdfn_launch_lstm_gate_fwd_split(const float* q, const float* h, const float* c, const float* Wih, const float* Whh,
                                     const float* bsum, const float* bsum2, float* gates, float* h_out, float* c_out, int R,
                                     int H, int ldh, const void* planes, hipStream_t s) {
    if (H < 8 || (H & 3) || R <= 0) return -2;
    const int nub = (H + 31) / 32;
    const int lds_bytes = 2 * 3 * 128 * SROW * 4;
    dim3 grid(((R + 127) / 128) * nub);
    if (planes != nullptr && h != nullptr)
        hipLaunchKernelGGL((lstm_gate_fwd_split_kernel<true>), grid, dim3(256), lds_bytes, s, q, h, c, Wih, Whh, bsum, bsum2, gates,
                           h_out, c_out, R, H, ldh, reinterpret_cast<const u32x4*>(planes));
    else
        hipLaunchKernelGGL((lstm_gate_fwd_split_kernel<false>), grid, dim3(256), lds_bytes, s, q, h, c, Wih, Whh, bsum, bsum2, gates,
                           h_out, c_out, R, H, ldh, static_cast<const u32x4*>(nullptr));
    MMDFN_CHECK_LAUNCH();
    return 0;
}

// floats of the piece planes of one cell (K = 2 H)
int64_t mmdfn_lstm_gate_planes_floats(int H) {
    return (int64_t)((H + 31) / 32) * ((2 * H + SBK - 1) / SBK) * GPL_CHUNK * 4;
}

int mmdfn_launch_lstm_gate_cut(const float* Wih, const float* Whh, void* planes, int H, hipStream_t s) {
    if (H < 8 || (H & 3) || Wih == nullptr || Whh == nullptr || planes == nullptr) return -1;
    const int nchunks = (2 * H + SBK - 1) / SBK;
    const int total = ((H + 31) / 32) * nchunks * 4 * 128;
    hipLaunchKernelGGL(lstm_gate_cut_kernel, dim3((total + 255) / 256), dim3(256), 0, s, Wih, Whh, reinterpret_cast<u32x4*>(planes), H,
                       nchunks, total);
    MMDFN_CHECK_LAUNCH();
    return 0;
}
