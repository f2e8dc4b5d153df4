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
constexpr int LDS64_TILES = 400;  // grids of at least this many 64 x 64 tiles use them (LDS-staged form)

struct SmallGroup {
    int n;
    int act;
    const float* X[SG_MAX];
    const float* W[SG_MAX];
    const float* W2[SG_MAX];
    const float* b[SG_MAX];
    const float* b2[SG_MAX];
    float* Y[SG_MAX];
    const float* Z[SG_MAX];      // accumulate: the addend (Y itself, or another buffer: out-of-place y = x W + z), row stride ldz
    int ldz[SG_MAX];
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
        const float* z = G.Z[p] + (int64_t)row * G.ldz[p] + col;
        if (vec && ((G.ldz[p] & 3) == 0) && ((reinterpret_cast<uintptr_t>(G.Z[p]) & 15) == 0)) {
            const float4 old = *reinterpret_cast<const float4*>(z);
            o[0] += old.x; o[1] += old.y; o[2] += old.z; o[3] += old.w;
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (col + e < N) o[e] += z[e];
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

// ---------------------------------------------------------------------------------------------------------------------------
// The LDS-staged form (round 3, second design).  What decides the speed of a GEMM this small is how its operand bytes get
// through the compute units' L1s and how much memory latency each workgroup exposes (profiles/r03_linear_group_vs_hipblaslt.md:
// with the 16-byte fragment-shaped loads of the kernel above the loads were 7.5 of 14 us, the exact-f32 MFMAs 4 us; a
// register-staged LDS double buffer moved whole lines but still waited one memory round trip per 64 k: 13.4 us).  Here:
//   * workgroup = one 32 x 64 output tile (220 workgroups at 1 760 x 200, one per CU); its four waves = two 32-column halves x
//     two halves of the contraction (even / odd 32-wide chunks), so every SIMD carries a quarter of the tile's MFMAs and the two
//     partial tiles per half meet in LDS in a fixed order (bit-reproducible);
//   * operands go HBM/L2 -> LDS by LDS-DMA (global_load_lds_dwordx4: no staging registers, nothing to wait for until the chunk
//     is consumed) in WHOLE cache lines: per 64-wide chunk of k the 32 x 64 block of X and the 64 x 64 block of W are 24 pieces
//     of 1 KB (4 rows x 256 bytes, 16 consecutive lanes per row), 6 per wave, into a ring of NST stages: NST - 1 chunks are in
//     flight while one feeds the MFMAs;
//   * the DMA writes lane-linear, so the bank swizzle is applied on the SOURCE side: the lane that fills 16-byte slot `pos` of
//     row r fetches unit pos ^ (r & 15) (k-contiguous rows; fragment reads are conflict-free 16-byte reads) or, for KMAJOR
//     weights ([k][n] rows), unit pos ^ 4 ((r >> 2) & 1) (fragment reads are conflict-free column reads);
//   * k past the end (and KMAJOR columns past N) are fetched from a 16-byte block of zeros, so the MFMA loop has no masks.
constexpr int TN = 64, DK = 64;                 // tile columns, k per staged chunk
constexpr int B_ST = DK * TN;                   // floats of the W block of a stage (16 KB)

#define LDS_AS(T, p) ((__attribute__((address_space(3))) T*)(p))
__device__ float4 g_zero16;                     // (zero-initialised) what out-of-range operand units are fetched from

// NP pieces (1 KB each, consecutive in LDS from byte offset m) of one wave; M0 kept
template <int NP>
__device__ __forceinline__ void dma_pieces(uint32_t m, const void* const* src) {
    uint32_t keep;
    if (NP == 2)
        asm volatile(
            "s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\t"
            "global_load_lds_dwordx4 %2, off\n\ts_add_u32 m0, m0, 0x400\n\ts_nop 0\n\t"
            "global_load_lds_dwordx4 %3, off\n\ts_mov_b32 m0, %0"
            : "=&s"(keep) : "s"(m), "v"(src[0]), "v"(src[1]) : "memory", "scc");
    else
        asm volatile(
            "s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\t"
            "global_load_lds_dwordx4 %2, off\n\ts_add_u32 m0, m0, 0x400\n\ts_nop 0\n\t"
            "global_load_lds_dwordx4 %3, off\n\ts_add_u32 m0, m0, 0x400\n\ts_nop 0\n\t"
            "global_load_lds_dwordx4 %4, off\n\ts_add_u32 m0, m0, 0x400\n\ts_nop 0\n\t"
            "global_load_lds_dwordx4 %5, off\n\ts_mov_b32 m0, %0"
            : "=&s"(keep) : "s"(m), "v"(src[0]), "v"(src[1]), "v"(src[2]), "v"(src[3]) : "memory", "scc");
}

// TMR = 32: 32 x 64 tile, the four waves = two 32-column halves x two halves of every chunk's k (partial tiles added in the
//           epilogue) -- the form for grids that would otherwise leave compute units empty;
// TMR = 64: 64 x 64 tile, the four waves = 2 x 2 quadrants, each contracting the whole chunk: a third fewer operand bytes per
//           output and half the per-chunk overhead per MFMA -- the form for grids of a few hundred tiles and more.
// ABL (tuning build, timing only): 1 no MFMA, 2 no operand DMA, 4 no reduction / epilogue
template <int NST, int TMR, int ABL = 0>
__global__ __launch_bounds__(256) void linear_lds_kernel(SmallGroup G) {
    constexpr int A_ST = TMR * DK;                  // floats of the X block of a stage
    constexpr int STAGE = A_ST + B_ST;              // 24 KB (TMR 32) / 32 KB (TMR 64)
    constexpr int NA = TMR / 16;                    // X pieces per wave and chunk
    constexpr int NC = TMR == 32 ? 2 : 4;           // 16-wide k groups a wave contracts per chunk
    constexpr int NL = NA + 4;                      // DMA instructions per wave and chunk
    extern __shared__ __attribute__((aligned(16))) float smem[];     // NST stages; the epilogue's partial tiles overlay them

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
    const int row0 = TMR * tr, col0 = TN * tc;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ch = w & 1;                               // column half of the tile
    const int rh = TMR == 32 ? 0 : (w >> 1);            // row half of the tile (TMR 64)
    const int kp = TMR == 32 ? (w >> 1) : 0;            // which 32-wide half of every staged chunk this wave contracts (TMR 32)
    const int fi = lane & 15, g = lane >> 4;
    const uint32_t lds0 = (uint32_t)(uintptr_t)LDS_AS(void, smem);

    // the epilogue's bias values are requested now: their round trip hides under the whole contraction
    const float* bias = G.b[p];
    const float* bias2 = G.b2[p];
    const int ecol = col0 + 4 * (tid & 15);
    float bv4[4] = {0.f, 0.f, 0.f, 0.f};
    if (bias) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (ecol + e < N) bv4[e] = (ecol + e < N1) ? bias[ecol + e] : bias2[ecol + e - N1];
    }

    // ---- staging: the lane fills slot `pos` of row `prow` of each of its wave's pieces
    const int prow = lane >> 4, pos = lane & 15;
    const float* xsrc[NA];
    int xk[NA];                                                    // first k of the lane's unit inside a chunk
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int rt = 4 * (NA * w + i) + prow;                    // row of the tile
        const int u = pos ^ (rt & 15);
        const int r = row0 + rt;
        xk[i] = 4 * u;
        xsrc[i] = X + (int64_t)(r < R ? r : R - 1) * ldx + 4 * u;
    }
    const float* wsrc[4];
    int wk[4];                                                     // NK: first k of the unit;  KMAJOR: the k-row
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int bt = 4 * (4 * w + i) + prow;                     // row of the W image, 0..63
        if (kmajor) {
            const int u = pos ^ (((bt >> 2) & 1) << 2);
            const int c = col0 + 4 * u;
            const bool cok = c + 3 < N;                            // (N % 4 == 0 for KMAJOR: checked by the launcher)
            wk[i] = bt;
            wsrc[i] = W + (int64_t)bt * ldw + (cok ? c : 0);      // columns past N only feed outputs that are not stored
        } else {
            const int u = pos ^ (bt & 15);
            const int c = col0 + bt;
            const int cc = c < N ? c : N - 1;
            wk[i] = 4 * u;
            wsrc[i] = ((cc < N1) ? W + (int64_t)cc * ldw : W2 + (int64_t)(cc - N1) * ldw) + 4 * u;
        }
    }
    // running sources: every issue moves them one chunk along k; only the LAST chunk can hold units past K, fetched from the zero block
    const int64_t wstep = kmajor ? (int64_t)DK * ldw : DK;         // floats per chunk along the weight source
    const void* const zero = &g_zero16;
    const int nj = (K + DK - 1) / DK;
    int issued = 0;
    uint32_t wr = lds0;                                            // LDS byte address of the stage the next issue fills
    auto issue = [&]() {
        if (ABL & 2) return;
        const void* sa[NA];
        const void* sb[4];
        if (issued == nj - 1) {
            const int k0 = DK * issued;
#pragma unroll
            for (int i = 0; i < NA; ++i) sa[i] = (k0 + xk[i] < K) ? (const void*)xsrc[i] : zero;
#pragma unroll
            for (int i = 0; i < 4; ++i) sb[i] = (k0 + wk[i] < K) ? (const void*)wsrc[i] : zero;
        } else {
#pragma unroll
            for (int i = 0; i < NA; ++i) sa[i] = xsrc[i];
#pragma unroll
            for (int i = 0; i < 4; ++i) sb[i] = wsrc[i];
        }
        dma_pieces<NA>(wr + (NA * w) * 1024, sa);
        dma_pieces<4>(wr + A_ST * 4 + (4 * w) * 1024, sb);
#pragma unroll
        for (int i = 0; i < NA; ++i) xsrc[i] += DK;
#pragma unroll
        for (int i = 0; i < 4; ++i) wsrc[i] += wstep;
        wr += STAGE * 4;
        if (wr == lds0 + NST * STAGE * 4) wr = lds0;
        ++issued;
    };

    // ---- fragment read offsets (floats): in k group c of the wave, lane (fi, g) holds k = 16 (NC kp + c) + 4 g .. + 3 of the chunk
    int aoff[2][NC], boff[2][NC];        // [16-row / 16-column tile][k group]
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int ar = 32 * rh + 16 * q + fi;
            const int kg = NC * kp + c;
            const int u = 4 * kg + g;
            aoff[q][c] = ar * DK + 4 * (u ^ (ar & 15));
            const int bc = 32 * ch + 16 * q + fi;
            boff[q][c] = kmajor ? A_ST + (16 * kg + 4 * g) * TN + (bc ^ ((g & 1) << 4)) : A_ST + bc * DK + 4 * (u ^ (bc & 15));
        }

    f32x4 acc[2][2];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) acc[rt][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};

#pragma unroll
    for (int c = 0; c < NST - 1; ++c)
        if (c < nj) issue();
    const float* s = smem;                                         // the stage chunk j is read from
    for (int j = 0; j < nj; ++j) {
        // chunk j landed (this wave's pieces), then everybody's: the chunks behind it stay in flight
        const int ahead = issued - 1 - j;
        if (NST >= 4 && ahead >= 2) {
            if (NL == 6) asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        } else if (NST >= 3 && ahead >= 1) {
            if (NL == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
        if (issued < nj) issue();                                  // into the stage chunk j-1 was read from (all waves are past it)
        const int kw = DK * j + 16 * NC * kp;                      // first k of this wave's part of the chunk
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            if (kw + 16 * c >= K) continue;                        // (wave-uniform) nothing but staged zeros from here on
            float4 av[2], bv[2];
#pragma unroll
            for (int q = 0; q < 2; ++q) av[q] = *reinterpret_cast<const float4*>(s + aoff[q][c]);
            if (kmajor) {
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const float* bp = s + boff[q][c];
                    bv[q] = make_float4(bp[0], bp[TN], bp[2 * TN], bp[3 * TN]);
                }
            } else {
#pragma unroll
                for (int q = 0; q < 2; ++q) bv[q] = *reinterpret_cast<const float4*>(s + boff[q][c]);
            }
            const float aj[2][4] = {{av[0].x, av[0].y, av[0].z, av[0].w}, {av[1].x, av[1].y, av[1].z, av[1].w}};
            const float bj[2][4] = {{bv[0].x, bv[0].y, bv[0].z, bv[0].w}, {bv[1].x, bv[1].y, bv[1].z, bv[1].w}};
#pragma unroll
            for (int jj = 0; jj < 4; ++jj)
#pragma unroll
                for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                    for (int ct = 0; ct < 2; ++ct)
                        if (!(ABL & 1)) acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(aj[rt][jj], bj[ct][jj], acc[rt][ct], 0, 0, 0);
                        else acc[rt][ct][jj] += aj[rt][jj] * bj[ct][jj];
        }
        s += STAGE;
        if (s == smem + NST * STAGE) s = smem;
    }
    __syncthreads();
    if (ABL & 4) {
        if (acc[0][0][0] + acc[1][1][1] + acc[0][1][2] + acc[1][0][3] == 1.2345f) G.Y[p][tid] = 1.f;
        return;
    }

    // ---- the waves' tiles meet in LDS: part[wave][32][36] (C/D layout of a 16 x 16 tile: column fi, rows 4 g + r)
    float* part = smem;
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int r = 0; r < 4; ++r) part[(w * 32 + 16 * rt + 4 * g + r) * 36 + 16 * ct + fi] = acc[rt][ct][r];
    __syncthreads();
    // row epilogue: float4 per thread and pass; 16 consecutive lanes = one 256-byte row of the tile
    const bool yvec = ((ldy & 3) == 0) && ((reinterpret_cast<uintptr_t>(G.Y[p]) & 15) == 0);
#pragma unroll
    for (int i = 0; i < TMR / 16; ++i) {
        const int idx = tid + 256 * i;
        const int er = idx >> 4, ec = 4 * (idx & 15);              // row of the tile, column 0..60
        const int row = row0 + er, col = col0 + ec;
        if (row >= R || col >= N) continue;
        const int h = ec >> 5, lc = ec & 31;
        float o[4];
        if (TMR == 32) {
            const float4 p0 = *reinterpret_cast<const float4*>(&part[(h * 32 + er) * 36 + lc]);            // k half 0 (wave h)
            const float4 p1 = *reinterpret_cast<const float4*>(&part[((2 + h) * 32 + er) * 36 + lc]);      // k half 1 (wave 2 + h)
            o[0] = p0.x + p1.x; o[1] = p0.y + p1.y; o[2] = p0.z + p1.z; o[3] = p0.w + p1.w;
        } else {
            const float4 p0 = *reinterpret_cast<const float4*>(&part[((2 * (er >> 5) + h) * 32 + (er & 31)) * 36 + lc]);
            o[0] = p0.x; o[1] = p0.y; o[2] = p0.z; o[3] = p0.w;
        }
        float* y = G.Y[p] + (int64_t)row * ldy + col;
        const bool vec = yvec && (col + 3 < N);
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] += bv4[e];
        if (G.accumulate[p]) {
            const float* z = G.Z[p] + (int64_t)row * G.ldz[p] + col;
            if (vec && ((G.ldz[p] & 3) == 0) && ((reinterpret_cast<uintptr_t>(G.Z[p]) & 15) == 0)) {
                const float4 old = *reinterpret_cast<const float4*>(z);
                o[0] += old.x; o[1] += old.y; o[2] += old.z; o[3] += old.w;
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (col + e < N) o[e] += z[e];
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
}

template <int NST, int TMR, int ABL>
int launch_lds(const SmallGroup& G, int tiles, hipStream_t stream) {
    auto kern = linear_lds_kernel<NST, TMR, ABL>;
    if (int e = mmdfn_allow_big_lds(kern)) return e;
    hipLaunchKernelGGL(kern, dim3(tiles), dim3(256), NST * (TMR * DK + B_ST) * 4, stream, G);
    MMDFN_CHECK_LAUNCH();
    return 0;
}

}  // namespace

// 1: the shape is one the few-row kernel covers (16-byte-aligned operands; the register-ring fallback for operands that are
// not is limited to K <= 768 and refused at launch beyond)
extern "C" int mmdfn_linear_group_supported(int R, int K, int N) {
    return (R > 0 && N > 0 && K >= 4 && (K & 3) == 0) ? 1 : 0;
}

extern "C" int mmdfn_linear_group(int n, const float* const* X, const float* const* W, const float* const* W2, const int* N1,
                                  const float* const* bias, const float* const* bias2, float* const* Y, const int* R,
                                  const int* K, const int* N, const int* ldx, const int* ldw, const int* ldy,
                                  const int* kmajor, const int* accumulate, int act, void* stream) {
    return mmdfn_linear_group_addend(n, X, W, W2, N1, bias, bias2, Y, nullptr, nullptr, R, K, N, ldx, ldw, ldy, kmajor, accumulate,
                                     act, stream);
}

extern "C" int mmdfn_linear_group_addend(int n, const float* const* X, const float* const* W, const float* const* W2,
                                         const int* N1, const float* const* bias, const float* const* bias2, float* const* Y,
                                         const float* const* Z, const int* ldz, const int* R, const int* K, const int* N,
                                         const int* ldx, const int* ldw, const int* ldy, const int* kmajor, const int* accumulate,
                                         int act, void* stream) {
    if (n <= 0 || n > SG_MAX) return -1;
    SmallGroup G;
    G.n = n;
    G.act = act;
    // the LDS-staged form needs 16-byte-aligned operand rows (and N % 4 == 0 for KMAJOR weights)
    bool lds = true;
    int64_t t64 = 0;
    for (int p = 0; p < n; ++p) {
        if ((reinterpret_cast<uintptr_t>(X[p]) & 15) || (reinterpret_cast<uintptr_t>(W[p]) & 15)) lds = false;
        if (kmajor[p] && ((ldw[p] & 3) || (N[p] & 3))) lds = false;
        if (!kmajor[p] && N1[p] < N[p] && (reinterpret_cast<uintptr_t>(W2[p]) & 15)) lds = false;
        t64 += (int64_t)((R[p] + 63) / 64) * ((N[p] + TN - 1) / TN);
    }
    // 64-row tiles once they fill the chip on their own (2 workgroups per CU) AND the contraction is long (measured: 7 040 x 600
    // -> 200: 29.9 us against 33.3 us with 32-row tiles; at K = 200 the 32-row tiles win, 3 520 x 200 -> 600: 19.2 against 21.5 us);
    // 32-row tiles (k split across waves) otherwise
    int kmin = 1 << 30;
    for (int p = 0; p < n; ++p) kmin = K[p] < kmin ? K[p] : kmin;
    int tm = (lds && t64 >= LDS64_TILES && kmin >= 400) ? 64 : 32;
#ifdef MMDFN_TUNING
    if (const char* e = getenv("MMDFN_LSM_V1")) lds = lds && atoi(e) == 0;
    if (const char* e = getenv("MMDFN_LSM_TM")) tm = atoi(e);
#endif
    const int tn = lds ? TN : 32;
    int t0 = 0;
    for (int p = 0; p < n; ++p) {
        if (!mmdfn_linear_group_supported(R[p], K[p], N[p])) return -1;
        if (!lds && K[p] > 4 * 16 * MAXC) return -1;
        if ((ldx[p] & 3) || ldx[p] < K[p] || ldy[p] < N[p]) return -1;
        if (kmajor[p]) {
            if (ldw[p] < N[p] || N1[p] != N[p]) return -1;
        } else {
            if ((ldw[p] & 3) || ldw[p] < K[p] || N1[p] <= 0 || N1[p] > N[p] || (N1[p] < N[p] && W2[p] == nullptr)) return -1;
            if (bias[p] != nullptr && N1[p] < N[p] && bias2[p] == nullptr) return -1;
        }
        G.X[p] = X[p]; G.W[p] = W[p]; G.W2[p] = W2[p]; G.b[p] = bias[p]; G.b2[p] = bias2[p]; G.Y[p] = Y[p];
        G.R[p] = R[p]; G.K[p] = K[p]; G.N[p] = N[p]; G.N1[p] = N1[p]; G.ldx[p] = ldx[p]; G.ldw[p] = ldw[p]; G.ldy[p] = ldy[p];
        G.kmajor[p] = kmajor[p];
        // problem p adds Z[p] (row stride ldz[p]) when one is given, its own output when `accumulate` asks for that
        const bool ext = Z != nullptr && Z[p] != nullptr;
        if (ext && ldz[p] < N[p]) return -1;
        G.accumulate[p] = (ext || accumulate[p]) ? 1 : 0;
        G.Z[p] = ext ? Z[p] : Y[p];
        G.ldz[p] = ext ? ldz[p] : ldy[p];
        G.ntn[p] = (N[p] + tn - 1) / tn;
        G.tile0[p] = t0;
        t0 += ((R[p] + tm - 1) / tm) * G.ntn[p];
    }
    for (int p = n; p <= SG_MAX; ++p) G.tile0[p] = t0;
    hipStream_t hs = (hipStream_t)stream;
    if (lds) {
        // ring depth: one workgroup per CU anyway -> a deep ring; more -> shallow rings, co-resident workgroups hide each other's waits
        int nst = (tm == 64) ? 2 : ((t0 <= 256) ? 4 : 2);
#ifdef MMDFN_TUNING
        if (const char* e = getenv("MMDFN_LSM_NST")) nst = atoi(e);
        if (const char* e = getenv("MMDFN_LSM_ABL")) {
            const int a = atoi(e);
            if (tm == 64) {
                switch (a) {
                    case 1: return launch_lds<2, 64, 1>(G, t0, hs);
                    case 2: return launch_lds<2, 64, 2>(G, t0, hs);
                    case 4: return launch_lds<2, 64, 4>(G, t0, hs);
                    case 3: return launch_lds<2, 64, 3>(G, t0, hs);
                    case 7: return launch_lds<2, 64, 7>(G, t0, hs);
                    default: break;
                }
            } else if (nst == 4) {
                switch (a) {
                    case 1: return launch_lds<4, 32, 1>(G, t0, hs);
                    case 2: return launch_lds<4, 32, 2>(G, t0, hs);
                    case 4: return launch_lds<4, 32, 4>(G, t0, hs);
                    case 3: return launch_lds<4, 32, 3>(G, t0, hs);
                    case 7: return launch_lds<4, 32, 7>(G, t0, hs);
                    default: break;
                }
            } else {
                switch (a) {
                    case 1: return launch_lds<2, 32, 1>(G, t0, hs);
                    case 2: return launch_lds<2, 32, 2>(G, t0, hs);
                    case 4: return launch_lds<2, 32, 4>(G, t0, hs);
                    case 3: return launch_lds<2, 32, 3>(G, t0, hs);
                    case 7: return launch_lds<2, 32, 7>(G, t0, hs);
                    default: break;
                }
            }
        }
        if (tm == 64 && nst == 3) return launch_lds<3, 64, 0>(G, t0, hs);
        if (tm == 64 && nst == 4) return launch_lds<4, 64, 0>(G, t0, hs);
        if (tm == 32 && nst == 3) return launch_lds<3, 32, 0>(G, t0, hs);
#endif
        if (tm == 64) return launch_lds<2, 64, 0>(G, t0, hs);
        if (nst == 4) return launch_lds<4, 32, 0>(G, t0, hs);
        return launch_lds<2, 32, 0>(G, t0, hs);
    }
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
