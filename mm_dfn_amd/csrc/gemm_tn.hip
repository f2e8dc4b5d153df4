// Weight-gradient contraction  C (M x N) = A^T B  (+ column sums of A)  with A: (R, M), B: (R, N), R >> M, N.
//
// Replaces the dW = dY^T X and db = sum_r dY reductions of the dense layers on the hot path (autograd of
// nn.Linear / nn.GRU in the reference).  The contraction runs over the ROW index r, i.e. both operands are
// read "transposed": row chunks of A and B are staged through LDS with coalesced 16-byte loads (row stride
// = 4 mod 8 floats -> conflict-free ds_read_b32 fragment reads, 16-byte aligned ds_write_b128), fragments
// A[r][m] / B[r][n] feed exact-f32 MFMA 16x16x4 with k = r.  Split over r across workgroups (the output is
// tiny, the reduction long): partial 64x64 tiles go to a workspace and a second kernel sums the slabs
// (fp32 atomics would serialise in L2).  Row strides lda/ldb let it read column slices / time-shifted views
// of the GRU buffers in place.
#include "mmdfn_internal.h"
#include "../../include/mmdfn_hip.h"
#include <stdlib.h>
#include <algorithm>
#include <stdint.h>

namespace {

constexpr int BR = 32;          // rows per staged chunk
constexpr int TM = 64, TN = 64; // output tile per workgroup (TNW columns in the wide form)
constexpr int TNW = 112;
constexpr int LDA = TM + 4;     // = 4 (mod 8)
constexpr int LDB = TN + 4;
constexpr int BS_FLOATS = 32 * (TNW + 4);    // B stage of the wider form (BR rows)

// 1: the 112-column tile covers N better than the 64-column one
inline bool tn_wide(int N) {
    const double e64 = (double)N / (64.0 * ((N + 63) / 64)), e112 = (double)N / (112.0 * ((N + 111) / 112));
    return e112 > e64 + 0.02;
}

// One workgroup: output tile `tile` of split `split`.  B is read at row r + bshift (rows outside [0, R) count as
// zero): the recurrent-weight gradient  dW_hh = sum_t dgh_t (x) h_{t-1}  pairs row t of dgh with row t-1 (forward
// direction) or t+1 (reverse) of the output sequence, and with the shift inside the kernel A still covers every row,
// so its column sums are the complete bias gradient.
// WIDE: the tile is 64 x 112 (seven 16-column MFMA tiles per wave, the four waves take 16 rows each) instead of 64 x 64 (2 x 2
// per wave, waves 2 x 2): the hot-path outputs are 100 or 200 columns wide, which 64-column tiles cover at 78 % (64 + 36) and
// 112-column tiles at 89 %.
template <bool WIDE>
__device__ __forceinline__ void gemm_tn_body(float (*As)[BR * LDA], float* Bs0, const float* __restrict__ A,
                                             const float* __restrict__ B, float* __restrict__ part,
                                             float* __restrict__ colpart, int R, int M, int N, int lda, int ldb,
                                             int rows_per_split, int bshift, int tile, int split) {
    constexpr int TN = WIDE ? TNW : 64;
    constexpr int LDB = TN + 4;                         // = 4 (mod 8) for both widths
    constexpr int WMT = WIDE ? 1 : 2, WNT = WIDE ? 7 : 2;   // 16-row / 16-column MFMA tiles per wave
    constexpr int NSB = WIDE ? 4 : 2;                   // B staging slots per thread (32 x TN / 4 float4 over 256 threads)
    constexpr int BSL = BR * TN / 4;                    // B float4 per chunk (896 / 512)
    float(*Bs)[BR * LDB] = reinterpret_cast<float(*)[BR * LDB]>(Bs0);
    const int nbn = (N + TN - 1) / TN;
    const int bm = tile / nbn;
    const int bn = tile - bm * nbn;
    const int r_begin = split * rows_per_split;
    const int r_end = min(R, r_begin + rows_per_split);
    const int m0 = bm * TM, n0 = bn * TN;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int wm = WIDE ? w : (w & 1), wn = WIDE ? 0 : (w >> 1);
    const int mrow0 = 16 * WMT * wm, ncol0 = 16 * WNT * wn;    // the wave's corner inside the tile
    const int fi = lane & 15, g = lane >> 4;

    f32x4 acc[WMT][WNT];
#pragma unroll
    for (int i = 0; i < WMT; ++i)
#pragma unroll
        for (int j = 0; j < WNT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float csum[WMT];
#pragma unroll
    for (int i = 0; i < WMT; ++i) csum[i] = 0.f;

    // staging slots: A BR*16 float4 = 512 -> 2 per thread; B BR*TN/4 float4 -> 2 (512) or 4 (896: the fourth slot exists for
    // the first two waves only, a wave-uniform condition)
    int s_r[2], s_c[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const int idx = tid + e * 256;
        s_r[e] = idx >> 4;
        s_c[e] = (idx & 15) * 4;
    }
    int b_r[NSB], b_c[NSB];
    bool b_on[NSB];
#pragma unroll
    for (int e = 0; e < NSB; ++e) {
        const int idx = tid + e * 256;
        b_on[e] = idx < BSL;
        const int ic = b_on[e] ? idx : BSL - 1;
        b_r[e] = ic / (TN / 4);
        b_c[e] = (ic - b_r[e] * (TN / 4)) * 4;
    }
    // two register sets form a ring: the loads of chunk c + 2 are issued during chunk c and stored to LDS at the top
    // of chunk c + 2, i.e. two MFMA blocks later -- twice the bytes in flight of a distance-1 prefetch (the kernel is
    // latency-bound: a split is 10-20 chunks long).  Unrolled by two so that the ring index is static.
    float4 ra[2][2], rb[2][NSB];
#define TN_ISSUE(SET, R0)                                                                                     \
    do {                                                                                                      \
        _Pragma("unroll") for (int e = 0; e < 2; ++e) {                                                       \
            const int r = (R0) + s_r[e];                                                                      \
            const int rc = r < r_end ? r : r_end - 1;                                                         \
            const int ca = m0 + s_c[e];                                                                       \
            ra[SET][e] = *reinterpret_cast<const float4*>(A + (int64_t)rc * lda + (ca < M ? ca : 0));         \
        }                                                                                                     \
        _Pragma("unroll") for (int e = 0; e < NSB; ++e) {                                                     \
            const int r = (R0) + b_r[e];                                                                      \
            const int rc = r < r_end ? r : r_end - 1;                                                         \
            const int rbs = rc + bshift;                                                                      \
            const int rbc = rbs < 0 ? 0 : (rbs < R ? rbs : R - 1);                                            \
            const int cb = n0 + b_c[e];                                                                       \
            rb[SET][e] = *reinterpret_cast<const float4*>(B + (int64_t)rbc * ldb + (cb < N ? cb : 0));        \
        }                                                                                                     \
    } while (0)
#define TN_CHUNK(SET, C)                                                                                      \
    do {                                                                                                      \
        const int r0 = r_begin + (C) * BR;                                                                    \
        float* as = As[(C) & 1];                                                                              \
        float* bs = Bs[(C) & 1];                                                                              \
        _Pragma("unroll") for (int e = 0; e < 2; ++e) {                                                       \
            const bool aok = (r0 + s_r[e]) < r_end && (m0 + s_c[e] < M);                                      \
            /* M, N are multiples of 4 (checked by the launcher), so a float4 is fully inside or outside */   \
            const float4 va = ra[SET][e];                                                                     \
            *reinterpret_cast<float4*>(&as[s_r[e] * LDA + s_c[e]]) =                                          \
                make_float4(aok ? va.x : 0.f, aok ? va.y : 0.f, aok ? va.z : 0.f, aok ? va.w : 0.f);          \
        }                                                                                                     \
        _Pragma("unroll") for (int e = 0; e < NSB; ++e) {                                                     \
            const int rbs = r0 + b_r[e] + bshift;                                                             \
            const bool bok = (r0 + b_r[e]) < r_end && (n0 + b_c[e] < N) && rbs >= 0 && rbs < R;               \
            const float4 vb = rb[SET][e];                                                                     \
            if (NSB == 2 || e < 3 || b_on[e])                                                                 \
                *reinterpret_cast<float4*>(&bs[b_r[e] * LDB + b_c[e]]) =                                      \
                    make_float4(bok ? vb.x : 0.f, bok ? vb.y : 0.f, bok ? vb.z : 0.f, bok ? vb.w : 0.f);      \
        }                                                                                                     \
        __syncthreads();                                                                                      \
        if ((C) + 2 < nchunks) TN_ISSUE(SET, r0 + 2 * BR);                                                    \
        _Pragma("unroll") for (int ks = 0; ks < BR / 4; ++ks) {                                               \
            const int rr = 4 * ks + g; /* MFMA k index = row within the chunk */                              \
            float av[WMT], bv[WNT];                                                                           \
            _Pragma("unroll") for (int i = 0; i < WMT; ++i) av[i] = as[rr * LDA + mrow0 + 16 * i + fi];       \
            _Pragma("unroll") for (int j = 0; j < WNT; ++j) bv[j] = bs[rr * LDB + ncol0 + 16 * j + fi];       \
            _Pragma("unroll") for (int i = 0; i < WMT; ++i) csum[i] += av[i];                                 \
            _Pragma("unroll") for (int i = 0; i < WMT; ++i)                                                   \
                _Pragma("unroll") for (int j = 0; j < WNT; ++j)                                               \
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], bv[j], acc[i][j], 0, 0, 0);       \
        }                                                                                                     \
    } while (0)
    const int nchunks = (r_end - r_begin + BR - 1) / BR;
    if (nchunks > 0) TN_ISSUE(0, r_begin);
    if (nchunks > 1) TN_ISSUE(1, r_begin + BR);
    for (int c = 0; c < nchunks; c += 2) {
        TN_CHUNK(0, c);
        if (c + 1 < nchunks) TN_CHUNK(1, c + 1);
    }
#undef TN_ISSUE
#undef TN_CHUNK
    // partial tile -> workspace [split][M][N]; C/D layout: col (n) = lane&15, row (m) = 4g + r
    float* P = part + (int64_t)split * M * N;
#pragma unroll
    for (int i = 0; i < WMT; ++i)
#pragma unroll
        for (int j = 0; j < WNT; ++j) {
            const int n = n0 + ncol0 + 16 * j + fi;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + mrow0 + 16 * i + 4 * g + r;
                if (m < M && n < N) P[(int64_t)m * N + n] = acc[i][j][r];
            }
        }
    if (colpart != nullptr && bn == 0 && wn == 0) {
        // lanes with equal fi hold partial sums of the same column (different row residues g)
#pragma unroll
        for (int i = 0; i < WMT; ++i) {
            float v = csum[i];
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            const int m = m0 + mrow0 + 16 * i + fi;
            if (g == 0 && m < M) colpart[(int64_t)split * M + m] = v;
        }
    }
}

__global__ __launch_bounds__(256) void gemm_tn_kernel(const float* __restrict__ A, const float* __restrict__ B,
                                                      float* __restrict__ part, float* __restrict__ colpart,
                                                      int R, int M, int N, int lda, int ldb, int rows_per_split) {
    __shared__ __attribute__((aligned(16))) float As[2][BR * LDA];
    __shared__ __attribute__((aligned(16))) float Bs[2 * BR * LDB];
    gemm_tn_body<false>(As, Bs, A, B, part, colpart, R, M, N, lda, ldb, rows_per_split, 0, blockIdx.x, blockIdx.y);
}

// up to TN_MAXG independent problems in one launch (each too small to fill the chip on its own)
constexpr int TN_MAXG = 8;
struct TnGroups {
    const float* A[TN_MAXG];
    const float* B[TN_MAXG];
    float* part[TN_MAXG];
    float* colpart[TN_MAXG];
    float* C[TN_MAXG];
    float* colsum[TN_MAXG];
    int R[TN_MAXG], M[TN_MAXG], N[TN_MAXG], lda[TN_MAXG], ldb[TN_MAXG], ldc[TN_MAXG], bshift[TN_MAXG];
    int rows_per_split[TN_MAXG], splits[TN_MAXG], tiles[TN_MAXG];
    int wg_prefix[TN_MAXG + 1];    // workgroups of the contraction kernel
    int blk_prefix[TN_MAXG + 1];   // 256-thread blocks of the slab reduction
    int n;
};

__global__ __launch_bounds__(256) void gemm_tn_grouped_kernel(const TnGroups gq) {
    int p = 0;
    while (p + 1 < gq.n && (int)blockIdx.x >= gq.wg_prefix[p + 1]) ++p;
    const int local = blockIdx.x - gq.wg_prefix[p];
    const int split = local / gq.tiles[p];
    const int tile = local - split * gq.tiles[p];
    __shared__ __attribute__((aligned(16))) float As[2][BR * LDA];
    __shared__ __attribute__((aligned(16))) float Bs[2 * BR * LDB];
    gemm_tn_body<false>(As, Bs, gq.A[p], gq.B[p], gq.part[p], gq.colpart[p], gq.R[p], gq.M[p], gq.N[p], gq.lda[p], gq.ldb[p],
                        gq.rows_per_split[p], gq.bshift[p], tile, split);
}

__device__ __forceinline__ void gemm_tn_reduce_body(const float* __restrict__ part, const float* __restrict__ colpart,
                                                    float* __restrict__ C, float* __restrict__ colsum, int M, int N,
                                                    int ldc, int splits, int64_t first, int64_t stride) {
    const int64_t total = (int64_t)M * N;
    for (int64_t idx = first; idx < total + M; idx += stride) {
        if (idx < total) {
            float s = 0.f;
            for (int k = 0; k < splits; ++k) s += part[(int64_t)k * total + idx];
            const int m = (int)(idx / N);
            C[(int64_t)m * ldc + (idx - (int64_t)m * N)] = s;
        } else if (colsum != nullptr) {
            const int m = (int)(idx - total);
            float s = 0.f;
            for (int k = 0; k < splits; ++k) s += colpart[(int64_t)k * M + m];
            colsum[m] = s;
        }
    }
}

__global__ void gemm_tn_reduce_kernel(const float* __restrict__ part, const float* __restrict__ colpart,
                                      float* __restrict__ C, float* __restrict__ colsum, int M, int N, int ldc,
                                      int splits) {
    gemm_tn_reduce_body(part, colpart, C, colsum, M, N, ldc, splits, blockIdx.x * (int64_t)blockDim.x + threadIdx.x,
                        (int64_t)gridDim.x * blockDim.x);
}

__global__ void gemm_tn_grouped_reduce_kernel(const TnGroups gq) {
    int p = 0;
    while (p + 1 < gq.n && (int)blockIdx.x >= gq.blk_prefix[p + 1]) ++p;
    const int nblk = gq.blk_prefix[p + 1] - gq.blk_prefix[p];
    gemm_tn_reduce_body(gq.part[p], gq.colpart[p], gq.C[p], gq.colsum[p], gq.M[p], gq.N[p], gq.ldc[p], gq.splits[p],
                        (blockIdx.x - gq.blk_prefix[p]) * (int64_t)blockDim.x + threadIdx.x, (int64_t)nblk * blockDim.x);
}


// ---------------------------------------------------------------------------------------------------------------
// Batch form: EVERY weight-gradient contraction of a training step in one launch pair.  A step has ~25 of them (dense
// layers, GRU input / recurrent weights, LSTM gate, GCN layers), each far too small to fill the chip and none feeding
// anything but the optimizer, so the host queues them during backward and issues them together at its end.
//   segment s: (A_s, B_s, R_s rows, shift) contributes  A_s^T B_s  to output out[s]; segments of one output are extra
//   splits of the same slab stack (the layer-shared LSTM gate gets one segment per GCN layer, the reduction sums them:
//   no gradient-accumulation kernels).
constexpr int TN_MAXSEG = 40;
constexpr int TN_MAXOUT = 40;
struct TnSegs {
    const float* A[TN_MAXSEG];
    const float* B[TN_MAXSEG];
    float* part[TN_MAXSEG];
    float* colpart[TN_MAXSEG];
    int R[TN_MAXSEG], lda[TN_MAXSEG], ldb[TN_MAXSEG], bshift[TN_MAXSEG], rows_per_split[TN_MAXSEG], tiles[TN_MAXSEG];
    int M[TN_MAXSEG], N[TN_MAXSEG], wide[TN_MAXSEG];
    int wg_prefix[TN_MAXSEG + 1];
    int n;
};
struct TnOuts {
    const float* part[TN_MAXOUT];
    const float* colpart[TN_MAXOUT];
    float* C[TN_MAXOUT];
    float* colsum[TN_MAXOUT];
    float* colsum2[TN_MAXOUT];    // optional second destination of the column sums (b_ih and b_hh share one gradient)
    int M[TN_MAXOUT], N[TN_MAXOUT], ldc[TN_MAXOUT], splits[TN_MAXOUT], accumulate[TN_MAXOUT];
    int blk_prefix[TN_MAXOUT + 1];
    int n;
};

// Segments flagged `wide` (outputs 100 or 200 columns wide) use 64 x 112 tiles, the others 64 x 64.  Register budget of three
// waves per SIMD (136 VGPRs): with the default budget hipcc took 146 and the occupancy of two cost the small cfg2 batch 13 us.
__global__ __launch_bounds__(256, 3) void gemm_tn_batch_kernel(const TnSegs sq) {
    int p = 0;
    while (p + 1 < sq.n && (int)blockIdx.x >= sq.wg_prefix[p + 1]) ++p;
    const int local = blockIdx.x - sq.wg_prefix[p];
    const int split = local / sq.tiles[p];
    const int tile = local - split * sq.tiles[p];
    __shared__ __attribute__((aligned(16))) float As[2][BR * LDA];
    __shared__ __attribute__((aligned(16))) float Bs[2 * BS_FLOATS];
    if (sq.wide[p])
        gemm_tn_body<true>(As, Bs, sq.A[p], sq.B[p], sq.part[p], sq.colpart[p], sq.R[p], sq.M[p], sq.N[p], sq.lda[p], sq.ldb[p],
                           sq.rows_per_split[p], sq.bshift[p], tile, split);
    else
        gemm_tn_body<false>(As, Bs, sq.A[p], sq.B[p], sq.part[p], sq.colpart[p], sq.R[p], sq.M[p], sq.N[p], sq.lda[p], sq.ldb[p],
                            sq.rows_per_split[p], sq.bshift[p], tile, split);
}

// ---------------------------------------------------------------------------------------------------------------
// Tall form for LONG segments (BASELINE cfg5: 24 576 .. 98 304 rows into a 400 x 100 or 100 x 100 output).  At that length the
// 64 x 112 tiles above keep the matrix pipe 64 % busy, a fifth of it on padding (400 rows = 7 x 64 = 448, 100 columns on 112), LDS
// 43 % busy, and every tile moves (64 + 112) x 4 bytes per row through L1 for 64 x 112 multiply-adds (hot and cold operands time
// the same: tools/bench_gemm_tn_tall.py; counters: profiles/r03_stack_kernels_pmc.md).  Here ONE workgroup owns every output row
// of a segment (up to 448 = 4 waves x 7 row tiles; a 400-row output is issued as 25 row tiles) x all (<= 112) columns: (400 + 100)
// x 4 bytes per row for 400 x 100 multiply-adds, 2.5 x fewer bytes per MFMA, a third of the LDS cycles, 10 % fewer MFMA cycles.
// The 49 accumulator tiles of a wave leave no registers for staging, so operand rows go global -> LDS by LDS-DMA
// (global_load_lds_dwordx4: scalar chunk base + per-lane offset, destination lane-linear), rows stored back to back (row stride M
// or N floats: the fragment reads conflict ~40 % of the time at 400 / 100, which costs ~1 % here).  Columns past M / N of the last
// MFMA tile read the next row's data: they only reach accumulator rows / columns that are never stored.
constexpr int TBR = 16;                        // rows per staged chunk
constexpr int TALL_MAX_M = 448, TALL_MAX_N = 112;

__device__ __forceinline__ void tall_dma(uint32_t m0v, uint32_t off, const void* base) {
    uint32_t keep;
    asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "s"(m0v), "v"(off), "s"(base) : "memory");
}

// ABL (tuning build, timing only): 1 no operand DMA, 2 no wait / barrier, 4 no fragment reads, 8 no MFMAs
template <int RT, int ABL = 0>
__device__ __forceinline__ void gemm_tn_tall_body(float* smem, const float* __restrict__ A, const float* __restrict__ B,
                                                  float* __restrict__ part, float* __restrict__ colpart, int R, int M, int NT,
                                                  int lda, int ldb, int rows_per_split, int split, int col0, int N) {
    // (NT: columns of the whole output; this workgroup owns columns col0 .. col0 + N - 1 of it, N = the segment's block width --
    //  the last block of an output may hold fewer: its surplus columns re-fetch valid ones and are not stored)
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fi = lane & 15, g = lane >> 4;
    const int r_begin = split * rows_per_split;
    const int r_end = min(R, r_begin + rows_per_split);
    const int nchunks = (r_end - r_begin) / TBR;                     // R and rows_per_split are multiples of TBR
    const int a_st = (TBR * M * 4 + 1023) & ~1023, b_st = (TBR * N * 4 + 1023) & ~1023;     // bytes per stage
    const uint32_t lds0 = (uint32_t)(uintptr_t)((__attribute__((address_space(3))) void*)smem);
    const int nrt = (M + 15) >> 4;                                   // MFMA row tiles of the output
    int nvt = nrt - w * RT;                                          // row tiles of this wave
    nvt = nvt < 0 ? 0 : (nvt > RT ? RT : nvt);

    // DMA slots of this wave: instruction j = w + 4 i moves operand pieces 64 j .. 64 j + 63 of a chunk (piece = 16 bytes,
    // row-major over the TBR x M block); lanes past the last piece re-fetch the last one into the stage's padding
    const int pa = TBR * (M >> 2), pb = TBR * (N >> 2);
    uint32_t aoff[RT], boff[2];
#pragma unroll
    for (int i = 0; i < RT; ++i) {
        int q = (w + 4 * i) * 64 + lane;
        q = q < pa ? q : pa - 1;
        const int row = q / (M >> 2), cp = q - row * (M >> 2);
        aoff[i] = (uint32_t)(row * lda + cp * 4) * 4u;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        int q = (w + 4 * i) * 64 + lane;
        q = q < pb ? q : pb - 1;
        const int row = q / (N >> 2), cp = q - row * (N >> 2);
        const int col = col0 + cp * 4;
        boff[i] = (uint32_t)(row * ldb + (col < NT ? col : col0)) * 4u;
    }
    auto issue = [&](int c) {
        const int st = c & 1;
        const float* ab = A + (int64_t)(r_begin + c * TBR) * lda;
        const float* bb = B + (int64_t)(r_begin + c * TBR) * ldb;
#pragma unroll
        for (int i = 0; i < RT; ++i)
            if ((w + 4 * i) * 64 < pa) tall_dma(lds0 + st * a_st + (w + 4 * i) * 1024, aoff[i], ab);
#pragma unroll
        for (int i = 0; i < 2; ++i)
            if ((w + 4 * i) * 64 < pb) tall_dma(lds0 + 2 * a_st + st * b_st + (w + 4 * i) * 1024, boff[i], bb);
    };

    f32x4 acc[RT][7];
#pragma unroll
    for (int i = 0; i < RT; ++i)
#pragma unroll
        for (int j = 0; j < 7; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float csum[RT];
#pragma unroll
    for (int i = 0; i < RT; ++i) csum[i] = 0.f;

    const int a_lane = g * M + 16 * RT * w + fi, b_lane = g * N + fi;      // float offsets of the lane's fragment elements
    if (nchunks > 0) issue(0);
    for (int c = 0; c < nchunks; ++c) {
        if (!(ABL & 2)) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's pieces of chunk c are in LDS ...
            __syncthreads();                                      // ... and everybody's; stage (c + 1) & 1 is no longer being read
        }
        if (c + 1 < nchunks && !(ABL & 1)) issue(c + 1);
        const float* as = smem + (c & 1) * (a_st >> 2) + a_lane;
        const float* bs = smem + 2 * (a_st >> 2) + (c & 1) * (b_st >> 2) + b_lane;
#pragma unroll
        for (int ks = 0; ks < TBR / 4; ++ks) {
            const float* ak = as + 4 * ks * M;
            const float* bk = bs + 4 * ks * N;
            float av[RT], bv[7];
            if (ABL & 4) {
#pragma unroll
                for (int i = 0; i < RT; ++i) av[i] = 1.0f + i + (float)c;
#pragma unroll
                for (int j = 0; j < 7; ++j) bv[j] = 2.0f + j + (float)c;
            } else {
#pragma unroll
                for (int i = 0; i < RT; ++i) av[i] = ak[16 * i];
#pragma unroll
                for (int j = 0; j < 7; ++j) bv[j] = bk[16 * j];
            }
#pragma unroll
            for (int i = 0; i < RT; ++i) csum[i] += av[i];
#pragma unroll
            for (int i = 0; i < RT; ++i)
                if (i < nvt) {
#pragma unroll
                    for (int j = 0; j < 7; ++j) {
                        if (ABL & 8) acc[i][j][0] += av[i] * bv[j];
                        else acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], bv[j], acc[i][j], 0, 0, 0);
                    }
                }
        }
    }
    float* P = part + (int64_t)split * M * NT;
#pragma unroll
    for (int i = 0; i < RT; ++i)
#pragma unroll
        for (int j = 0; j < 7; ++j) {
            const int n = 16 * j + fi;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = 16 * (RT * w + i) + 4 * g + r;
                if (m < M && n < N && col0 + n < NT) P[(int64_t)m * NT + col0 + n] = acc[i][j][r];
            }
        }
    if (colpart != nullptr && col0 == 0) {
#pragma unroll
        for (int i = 0; i < RT; ++i) {
            float v = csum[i];
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            const int m = 16 * (RT * w + i) + fi;
            if (g == 0 && m < M) colpart[(int64_t)split * M + m] = v;
        }
    }
}

// The launch of a big batch: tall segments (TnSegs: tiles = column blocks of the output, wide = their width >= 64, bshift unused)
// first, the batch's other segments on the tiled bodies behind them in the SAME grid -- their short workgroups fill the CUs the
// last tall ones leave idle (as a launch of their own they were 93 us of mostly empty chip at cfg5).
template <int ABL>
__global__ __launch_bounds__(256, 2) void gemm_tn_tall_kernel(const TnSegs sq) {
    extern __shared__ __attribute__((aligned(16))) float tall_smem[];
    int p = 0;
    while (p + 1 < sq.n && (int)blockIdx.x >= sq.wg_prefix[p + 1]) ++p;
    const int local = blockIdx.x - sq.wg_prefix[p];
    const int split = local / sq.tiles[p];
    const int tile = local - split * sq.tiles[p];
    if (sq.wide[p] >= 64) {
#define TALL_ARGS tall_smem, sq.A[p], sq.B[p], sq.part[p], sq.colpart[p], sq.R[p], sq.M[p], sq.N[p], sq.lda[p], sq.ldb[p], \
                  sq.rows_per_split[p], split, tile * sq.wide[p], sq.wide[p]
        if (sq.M[p] > 128) gemm_tn_tall_body<7, ABL>(TALL_ARGS);
        else gemm_tn_tall_body<2, ABL>(TALL_ARGS);
#undef TALL_ARGS
        return;
    }
    float(*As)[BR * LDA] = reinterpret_cast<float(*)[BR * LDA]>(tall_smem);
    float* Bs = tall_smem + 2 * BR * LDA;
    if (sq.wide[p])
        gemm_tn_body<true>(As, Bs, sq.A[p], sq.B[p], sq.part[p], sq.colpart[p], sq.R[p], sq.M[p], sq.N[p], sq.lda[p], sq.ldb[p],
                           sq.rows_per_split[p], sq.bshift[p], tile, split);
    else
        gemm_tn_body<false>(As, Bs, sq.A[p], sq.B[p], sq.part[p], sq.colpart[p], sq.R[p], sq.M[p], sq.N[p], sq.lda[p], sq.ldb[p],
                            sq.rows_per_split[p], sq.bshift[p], tile, split);
}

constexpr int TN_REDUCE_WIDE = 40;      // stacks taller than this get 8 lanes per output element

__global__ void gemm_tn_batch_reduce_kernel(const TnOuts oq) {
    int p = 0;
    while (p + 1 < oq.n && (int)blockIdx.x >= oq.blk_prefix[p + 1]) ++p;
    const int nblk = oq.blk_prefix[p + 1] - oq.blk_prefix[p];
    const int M = oq.M[p], N = oq.N[p], ldc = oq.ldc[p], splits = oq.splits[p];
    const bool acc = oq.accumulate[p] != 0;
    const float* part = oq.part[p];
    const float* colpart = oq.colpart[p];
    float* C = oq.C[p];
    float* cs = oq.colsum[p];
    float* cs2 = oq.colsum2[p];
    const int64_t total = (int64_t)M * N;
    const int64_t stride = (int64_t)nblk * blockDim.x;
    if (splits > TN_REDUCE_WIDE) {
        // tall stacks (foreign ones: 256 slabs of the head's backward, 48 of the column-sum kernel): 8 lanes per output element,
        // each adds its share of the slabs in a fixed order with all its loads in flight, the partial sums meet in a fixed
        // shuffle tree (bit-reproducible; the order of head.hip's own reduction)
        const int sub = threadIdx.x & 7;
        for (int64_t t = (blockIdx.x - oq.blk_prefix[p]) * (int64_t)blockDim.x + threadIdx.x; t < ((total + M + 31) & ~31ll) * 8;
             t += stride) {
            const int64_t idx = t >> 3;
            const bool live = idx < total + M && (idx < total || cs != nullptr);
            const float* src = idx < total ? part + idx : colpart + (idx - total);
            const int64_t sstride = idx < total ? total : M;
            float s = 0.f;
            if (live) {
                int k = sub;
                for (; k + 56 < splits; k += 64) {
                    float v[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = src[(int64_t)(k + 8 * e) * sstride];
#pragma unroll
                    for (int e = 0; e < 8; ++e) s += v[e];
                }
                for (; k < splits; k += 8) s += src[(int64_t)k * sstride];
            }
            s += __shfl_xor(s, 1, 64);
            s += __shfl_xor(s, 2, 64);
            s += __shfl_xor(s, 4, 64);
            if (live && sub == 0) {
                if (idx < total) {
                    const int m = (int)(idx / N);
                    float* dst = C + (int64_t)m * ldc + (idx - (int64_t)m * N);
                    *dst = acc ? *dst + s : s;
                } else {
                    const int m = (int)(idx - total);
                    cs[m] = acc ? cs[m] + s : s;
                    if (cs2 != nullptr) cs2[m] = acc ? cs2[m] + s : s;
                }
            }
        }
        return;
    }
    for (int64_t idx = (blockIdx.x - oq.blk_prefix[p]) * (int64_t)blockDim.x + threadIdx.x; idx < total + M; idx += stride) {
        if (idx < total) {
            // fixed summation order, 8 independent loads in flight per thread (a plain loop pays the latency per slab)
            float s = 0.f;
            int k = 0;
            for (; k + 8 <= splits; k += 8) {
                float v[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = part[(int64_t)(k + e) * total + idx];
#pragma unroll
                for (int e = 0; e < 8; ++e) s += v[e];
            }
            for (; k < splits; ++k) s += part[(int64_t)k * total + idx];
            const int m = (int)(idx / N);
            float* dst = C + (int64_t)m * ldc + (idx - (int64_t)m * N);
            *dst = acc ? *dst + s : s;
        } else if (cs != nullptr) {
            const int m = (int)(idx - total);
            float s = 0.f;
#pragma unroll 8
            for (int k = 0; k < splits; ++k) s += colpart[(int64_t)k * M + m];
            cs[m] = acc ? cs[m] + s : s;
            if (cs2 != nullptr) cs2[m] = acc ? cs2[m] + s : s;
        }
    }
}

}  // namespace

// rows_target: rows per split aimed at while the problem is short enough for <= 16 splits
static int tn_splits_for(int R, int M, int N, int rows_target) {
#ifdef MMDFN_TUNING
    if (const char* e = getenv("MMDFN_TN_SPLITS")) {   // tools/bench_gemm_tn.py
        const int v = atoi(e);
        if (v > 0) return v;
    }
#endif
    // measured on MI355X (tools/bench_gemm_tn.py): ~330-660 rows per split is the sweet spot for every hot-path
    // shape (R = 1.7k .. 10.5k, outputs 100x200 .. 600x200) launched on its own; more splits only inflate the slab
    // reduction.  Long reductions (cfg5: R = 98 304 rows into a 100 x 200 output = 8 tiles) need far more than 16
    // splits to put a workgroup on every CU: ~512 rows per split, at most ~1536 workgroups per problem.
    int s = (R + rows_target - 1) / rows_target;
    if (s < 8) s = 8;
    if (s > 16) {
        const int tiles = ((M + TM - 1) / TM) * ((N + TN - 1) / TN);
        int cap = 1536 / (tiles > 0 ? tiles : 1);
        if (cap < 16) cap = 16;
        s = (R + 511) / 512;
        if (s > cap) s = cap;
        if (s < 16) s = 16;
    }
    const int max_s = (R + 2 * BR - 1) / (2 * BR);  // at least two staged chunks per split
    if (s > max_s) s = max_s;
    if (s < 1) s = 1;
    return s;
}

extern "C" int mmdfn_gemm_tn_splits(int R, int M, int N) { return tn_splits_for(R, M, N, 330); }

extern "C" int mmdfn_gemm_tn(const float* A, const float* B, float* C, float* colsum, float* workspace, int R, int M,
                             int N, int lda, int ldb, int ldc, int splits, void* stream) {
    if (R <= 0 || M <= 0 || N <= 0 || (M & 3) || (N & 3) || (lda & 3) || (ldb & 3) || lda < M || ldb < N || ldc < N ||
        splits < 1)
        return -1;
    hipStream_t s = (hipStream_t)stream;
    const int tiles = ((M + TM - 1) / TM) * ((N + TN - 1) / TN);
    const int rows_per_split = ((R + splits - 1) / splits + BR - 1) / BR * BR;
    const int eff_splits = (R + rows_per_split - 1) / rows_per_split;
    float* part = workspace;
    float* colpart = colsum ? workspace + (int64_t)splits * M * N : nullptr;
    hipLaunchKernelGGL(gemm_tn_kernel, dim3(tiles, eff_splits), dim3(256), 0, s, A, B, part, colpart, R, M, N, lda, ldb,
                       rows_per_split);
    MMDFN_CHECK_LAUNCH();
    int64_t total = (int64_t)M * N + M;
    int grid = (int)((total + 255) / 256);
    if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(gemm_tn_reduce_kernel, dim3(grid), dim3(256), 0, s, part, colpart, C, colsum, M, N, ldc,
                       eff_splits);
    MMDFN_CHECK_LAUNCH();
    return 0;
}

extern "C" int64_t mmdfn_gemm_tn_grouped_workspace(int n, const int* R, const int* M, const int* N) {
    int64_t total = 0;
    for (int p = 0; p < n; ++p) total += (int64_t)mmdfn_gemm_tn_splits(R[p], M[p], N[p]) * ((int64_t)M[p] * N[p] + M[p]);
    return total;
}

extern "C" int mmdfn_gemm_tn_grouped(int n, const float* const* A, const float* const* B, float* const* C,
                                     float* const* colsum, const int* R, const int* M, const int* N, const int* lda,
                                     const int* ldb, const int* ldc, const int* bshift, float* workspace, void* stream) {
    if (n < 1 || n > TN_MAXG) return -1;
    TnGroups gq;
    gq.n = n;
    gq.wg_prefix[0] = 0;
    gq.blk_prefix[0] = 0;
    float* ws = workspace;
    for (int p = 0; p < n; ++p) {
        if (R[p] <= 0 || M[p] <= 0 || N[p] <= 0 || (M[p] & 3) || (N[p] & 3) || (lda[p] & 3) || (ldb[p] & 3) ||
            lda[p] < M[p] || ldb[p] < N[p] || ldc[p] < N[p])
            return -1;
        const int splits = mmdfn_gemm_tn_splits(R[p], M[p], N[p]);
        const int tiles = ((M[p] + TM - 1) / TM) * ((N[p] + TN - 1) / TN);
        const int rps = ((R[p] + splits - 1) / splits + BR - 1) / BR * BR;
        const int eff = (R[p] + rps - 1) / rps;
        gq.A[p] = A[p]; gq.B[p] = B[p]; gq.C[p] = C[p]; gq.colsum[p] = colsum ? colsum[p] : nullptr;
        gq.R[p] = R[p]; gq.M[p] = M[p]; gq.N[p] = N[p]; gq.lda[p] = lda[p]; gq.ldb[p] = ldb[p]; gq.ldc[p] = ldc[p];
        gq.bshift[p] = bshift ? bshift[p] : 0;
        gq.rows_per_split[p] = rps; gq.splits[p] = eff; gq.tiles[p] = tiles;
        gq.part[p] = ws;
        gq.colpart[p] = gq.colsum[p] ? ws + (int64_t)splits * M[p] * N[p] : nullptr;
        ws += (int64_t)splits * ((int64_t)M[p] * N[p] + M[p]);
        gq.wg_prefix[p + 1] = gq.wg_prefix[p] + tiles * eff;
        int nblk = (int)(((int64_t)M[p] * N[p] + M[p] + 255) / 256);
        if (nblk > 512) nblk = 512;
        gq.blk_prefix[p + 1] = gq.blk_prefix[p] + nblk;
    }
    for (int p = n; p < TN_MAXG; ++p) {
        gq.A[p] = gq.B[p] = nullptr; gq.part[p] = gq.colpart[p] = gq.C[p] = gq.colsum[p] = nullptr;
        gq.R[p] = gq.M[p] = gq.N[p] = gq.lda[p] = gq.ldb[p] = gq.ldc[p] = gq.bshift[p] = 0;
        gq.rows_per_split[p] = gq.splits[p] = gq.tiles[p] = 0;
        gq.wg_prefix[p + 1] = gq.wg_prefix[n]; gq.blk_prefix[p + 1] = gq.blk_prefix[n];
    }
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(gemm_tn_grouped_kernel, dim3(gq.wg_prefix[n]), dim3(256), 0, s, gq);
    MMDFN_CHECK_LAUNCH();
    hipLaunchKernelGGL(gemm_tn_grouped_reduce_kernel, dim3(gq.blk_prefix[n]), dim3(256), 0, s, gq);
    MMDFN_CHECK_LAUNCH();
    return 0;
}

// ---- batch form (see TnSegs / TnOuts above) ----------------------------------------------------------------------
// In the batch form ~25 contractions share one launch, so no single one has to fill the chip: 8 splits each (~1000 rows
// per split at the hot-path sizes) instead of 8-16 keeps as many workgroups in flight with half the slabs to write and to
// reduce (cfg2 step 1.117 -> 1.109 ms; 4 or fewer splits lose again).
// Rows per split of a BATCH: 1 000 at the dialogue-graph sizes (above); when the batch as a whole already holds many more
// workgroups than that needs (BASELINE cfg5: 24 576-row segments, 15 000 workgroups of 16 chunks at 512 rows per split, 69
// TFLOP/s), longer splits -- about 12 workgroups per CU in total, at most 4 096 rows -- amortise each workgroup's prologue and
// write / reduce a quarter of the slabs.
#ifdef MMDFN_TUNING
static bool tn_no_wide() { const char* e = getenv("MMDFN_TN_NO_WIDE"); return e && atoi(e) != 0; }
#else
constexpr bool tn_no_wide() { return false; }
#endif

static bool tn_tall_shape1(int R, int M, int N);
static int batch_rows_target(int nseg, const int* R, const int* out, int nout, const int* M, const int* N, bool tall_on) {
    double units = 0.0;                                  // sum over segments of output tiles x rows (tiled form only)
    for (int s = 0; s < nseg; ++s) {
        const int o = out[s];
        if (o < 0 || o >= nout || (tall_on && tn_tall_shape1(R[s], M[o], N[o]))) continue;
        units += (double)(((M[o] + TM - 1) / TM) * ((N[o] + TN - 1) / TN)) * R[s];
    }
    double wgs = 3072.0;
    int rt_max = 4096;
#ifdef MMDFN_TUNING
    if (const char* e = getenv("MMDFN_TN_BATCH_WGS")) wgs = atof(e);
    if (const char* e = getenv("MMDFN_TN_BATCH_RTMAX")) rt_max = atoi(e);
#endif
    int rt = (int)(units / wgs);
    if (rt < 1000) rt = 1000;
    if (rt > rt_max) rt = rt_max;
    return rt;
}

// Tall form (gemm_tn_tall_kernel): the long segments of a BIG batch.  A segment qualifies by shape (rows a multiple of 16, at
// least 8192; 64 .. 128 or 336 .. 448 output rows = 2 or 7 row tiles per wave; 64 .. 112 columns -- the kernel also cuts wider
// outputs into column blocks, but the 100 x 200 / 200 x 512 segments of cfg5 measured slower that way than on the tiles: every
// block re-reads the A rows and a 4 096-row projection makes too few workgroups); the form is used when the qualifying segments together hold enough matrix work to
// put a 768-row workgroup of the 49-tile kind (or its equivalent) on most CUs -- the dialogue-graph batches (cfg2 .. cfg4: a few
// thousand rows per segment) stay on the tiles above, whose many small workgroups fill the chip there.
// Rows per workgroup (tools/bench_gemm_tn_tall.py, cfg5 with 8 and 32 dialogues): the batch time is flat between ~250 and ~1000
// workgroups of the 49-tile kind and rises on both sides (fewer: CUs idle behind the last ones; more: partial outputs to write and
// reduce), so aim at ~384 of them, 768 .. 3072 rows each; the lighter kinds 1536 .. 3072 rows.
constexpr int TALL_MIN_R = 8192;
constexpr double TALL_MIN_COST = 192.0 * 768 * 49;      // (rows x accumulator tiles per wave, summed over column blocks)
static int tall_kind(int M) { return M > 128 ? 7 : 2; }
static int tall_blocks(int N) { return (N + TALL_MAX_N - 1) / TALL_MAX_N; }
static int tall_block_width(int N) { const int nb = tall_blocks(N); return ((N + nb - 1) / nb + 3) / 4 * 4; }
static bool tn_tall_shape1(int R, int M, int N) {
    return R >= TALL_MIN_R && (R % TBR) == 0 && N >= 64 && N <= TALL_MAX_N && M >= 64 && M <= TALL_MAX_M && !(M > 128 && M < 336);
}
struct TallPlan { bool on; int rows7, rows2; };
// (from the segment shapes alone, so that the workspace query and the launch agree)
static TallPlan tall_plan(int nseg, const int* R, const int* out, int nout, const int* M, const int* N) {
    double tot7 = 0.0, tot2 = 0.0, cost = 0.0;
    for (int s = 0; s < nseg; ++s) {
        const int o = out[s];
        if (o < 0 || o >= nout || !tn_tall_shape1(R[s], M[o], N[o])) continue;
        const int k = tall_kind(M[o]), nb = tall_blocks(N[o]);
        (k == 7 ? tot7 : tot2) += (double)R[s] * nb;
        cost += (double)R[s] * nb * 7 * k;
    }
    TallPlan p;
    p.on = cost >= TALL_MIN_COST;
#ifdef MMDFN_TUNING
    if (const char* e = getenv("MMDFN_TN_NO_TALL")) if (atoi(e)) p.on = false;
#endif
    p.rows7 = (int)(tot7 / 384.0);
    p.rows2 = (int)(tot2 / 256.0);
    p.rows7 = p.rows7 < 768 ? 768 : (p.rows7 > 3072 ? 3072 : p.rows7);
    p.rows2 = p.rows2 < 1536 ? 1536 : (p.rows2 > 3072 ? 3072 : p.rows2);
#ifdef MMDFN_TUNING
    if (const char* e = getenv("MMDFN_TN_TALL_ROWS7")) p.rows7 = atoi(e);
    if (const char* e = getenv("MMDFN_TN_TALL_ROWS2")) p.rows2 = atoi(e);
#endif
    return p;
}
static int tall_eff_splits(int R, int M, const TallPlan& plan, int* rps_out) {
    const int target = M > 128 ? plan.rows7 : plan.rows2;
    int splits = (R + target - 1) / (target > 0 ? target : 1);
    if (splits < 1) splits = 1;
    const int rps = ((R + splits - 1) / splits + TBR - 1) / TBR * TBR;
    if (rps_out) *rps_out = rps;
    return (R + rps - 1) / rps;
}

static int batch_eff_splits(int R, int M, int N, int rows_target, int* rps_out) {
    int splits = tn_splits_for(R, M, N, 1000);
    if (rows_target > 1000) {
        splits = (R + rows_target - 1) / rows_target;
        if (splits < 1) splits = 1;
    }
    const int rps = ((R + splits - 1) / splits + BR - 1) / BR * BR;
    if (rps_out) *rps_out = rps;
    return (R + rps - 1) / rps;
}

// bf16-piece form (gemm_tn_split.hip): every segment of the batch on 128 x 112 tiles of v_mfma_f32_16x16x32_bf16.  One workgroup
// of 8 waves per CU (110 KB of LDS), so the batch is cut into about TNS_WGS workgroups of equal length: rows per split =
// (sum over segments of output tiles x rows) / TNS_WGS, between 256 and 4 096 rows.
#ifdef MMDFN_TUNING
static bool tns_enabled() { const char* e = getenv("MMDFN_TN_SPLIT"); return !e || atoi(e) != 0; }
#else
constexpr bool tns_enabled() { return true; }
#endif
// Column blocks of an output: 112 wide, or -- when that needs fewer of them by enough (a 224-column block costs about 1.5 of
// the narrow ones: the A planes are cut once for its two halves) -- 224 wide (N = 200, 400, 512, 600: yes; 100, 300: no).
static bool tns_wide(int N) {
#ifdef MMDFN_TUNING
    if (const char* e = getenv("MMDFN_TNS_WIDE")) { if (atoi(e) == 0) return false; }
#endif
    const int nn = (N + MMDFN_TNS_TN - 1) / MMDFN_TNS_TN, nw = (N + 2 * MMDFN_TNS_TN - 1) / (2 * MMDFN_TNS_TN);
    return 3 * nw < 2 * nn;
}
static int tns_tiles(int M, int N, int* nblocks, int* wide = nullptr) {
    const bool w = tns_wide(N);
    const int bw = w ? 2 * MMDFN_TNS_TN : MMDFN_TNS_TN;
    const int nb = (N + bw - 1) / bw;
    if (nblocks) *nblocks = nb;
    if (wide) *wide = w ? 1 : 0;
    return ((M + MMDFN_TNS_TM - 1) / MMDFN_TNS_TM) * nb;
}
static int tns_rows_target(int nseg, const int* R, const int* out, int nout, const int* M, const int* N, bool riders = false) {
    double units = 0.0;
    for (int s = 0; s < nseg; ++s) {
        const int o = out[s];
        if (o < 0 || o >= nout) continue;
        int wd = 0;
        const int tiles = tns_tiles(M[o], N[o], nullptr, &wd);
        units += (wd ? 1.5 : 1.0) * tiles * R[s];              // (a 224-column tile holds about 1.5 narrow tiles of work)
    }
    double wgs = 384.0;          // (round 5, with the 224-column tiles: 256 .. 768 swept on the cfg2-cfg5 batches)
    // a rider batch has a third of the chip for the length of a recurrence: fewer, longer workgroups (and fewer slabs for the
    // end-of-backward reduction launch, which sums the rider batches' stacks as well)
    if (riders) wgs = 160.0;     // (96 .. 192 measure the same on the cfg2 step, 224 +0.3 %, 384 +1.4 %)
#ifdef MMDFN_TUNING
    if (const char* e = getenv("MMDFN_TNS_WGS")) wgs = atof(e);
    if (riders) { if (const char* e = getenv("MMDFN_RIDER_WGS")) wgs = atof(e); }
#endif
    int rt = (int)(units / wgs);
    rt = rt < 256 ? 256 : (rt > 4096 ? 4096 : rt);
    return rt;
}
static int tns_eff_splits(int R, int rows_target, int* rps_out) {
    // a multiple of 8 (one split per XCD and round), nearest to rows / target; short segments: at least two chunks per split
    int splits = 8 * ((R + 4 * rows_target) / (8 * rows_target));
    if (splits < 8) splits = 8;
    const int max_s = (R + 2 * MMDFN_TNS_BK - 1) / (2 * MMDFN_TNS_BK);
    if (splits > max_s) splits = max_s;
    if (splits < 1) splits = 1;
    const int rps = ((R + splits - 1) / splits + MMDFN_TNS_BK - 1) / MMDFN_TNS_BK * MMDFN_TNS_BK;
    if (rps_out) *rps_out = rps;
    return (R + rps - 1) / rps;
}

extern "C" int64_t mmdfn_gemm_tn_batch_workspace(int nseg, const int* R, const int* out, int nout, const int* M,
                                                 const int* N) {
    int64_t total = 0;
    const TallPlan plan = tall_plan(nseg, R, out, nout, M, N);
    const int rt = batch_rows_target(nseg, R, out, nout, M, N, plan.on);
    const int rt_split = tns_rows_target(nseg, R, out, nout, M, N);
    for (int s = 0; s < nseg; ++s) {
        const int o = out[s];
        if (o < 0 || o >= nout) return -1;
        int eff = batch_eff_splits(R[s], M[o], N[o], rt, nullptr);
        if (plan.on && tn_tall_shape1(R[s], M[o], N[o])) eff = std::max(eff, tall_eff_splits(R[s], M[o], plan, nullptr));   // (either form may run)
        eff = std::max(eff, tns_eff_splits(R[s], rt_split, nullptr));
        total += (int64_t)eff * ((int64_t)M[o] * N[o] + M[o]);
    }
    return total;
}

// Slab stacks produced by OTHER kernels (the head's backward, the column-sum kernel of the project-then-gather node) that the
// batch's reduction launch sums as well -- the last launch of a backward pass, so handing them over costs no launch of their own.
struct TnExt {
    int n;
    const float* const* part;       // [splits][M][N] (may be null when N = 0)
    const float* const* colpart;    // [splits][M] or null
    float* const* C;
    float* const* colsum;
    const int *M, *N, *ldc, *splits, *accumulate;
};

// A batch staged for the GRU backward launch (see mmdfn_internal.h): its tile table and its reduction table.
struct RiderPlan {
    bool valid = false;
    TnSplitSegs tq;
    TnOuts oq;
    int nblk = 0;
};
static RiderPlan g_rider;

const TnSplitSegs* mmdfn_riders_pending() { return g_rider.valid ? &g_rider.tq : nullptr; }

// Slab stacks of rider batches whose reduction waits for the NEXT reduction launch of the backward pass (the end-of-backward
// batch's, normally): a reduction launch of their own behind every recurrence costs the chain ~12 us each.
struct DeferredOut {
    const float* part; const float* colpart; float* C; float* colsum; float* colsum2;
    int M, N, ldc, splits, accumulate;
};
static DeferredOut g_deferred[TN_MAXOUT];
static int g_ndeferred = 0;

static int reduce_blocks(int M, int N, int splits) {
    int nblk = (int)((((int64_t)M * N + M) * (splits > TN_REDUCE_WIDE ? 8 : 1) + 255) / 256);
    return nblk > 256 ? 256 : nblk;
}

static void put_out(TnOuts& oq, int o, const DeferredOut& d) {
    oq.part[o] = d.part; oq.colpart[o] = d.colpart; oq.C[o] = d.C; oq.colsum[o] = d.colsum; oq.colsum2[o] = d.colsum2;
    oq.M[o] = d.M; oq.N[o] = d.N; oq.ldc[o] = d.ldc; oq.splits[o] = d.splits; oq.accumulate[o] = d.accumulate;
    oq.blk_prefix[o + 1] = oq.blk_prefix[o] + reduce_blocks(d.M, d.N, d.splits);
}

static int launch_deferred(hipStream_t s) {
    if (g_ndeferred == 0) return 0;
    TnOuts oq;
    oq.blk_prefix[0] = 0;
    for (int o = 0; o < g_ndeferred; ++o) put_out(oq, o, g_deferred[o]);
    oq.n = g_ndeferred;
    for (int o = g_ndeferred; o < TN_MAXOUT; ++o) {
        oq.part[o] = oq.colpart[o] = nullptr; oq.C[o] = oq.colsum[o] = oq.colsum2[o] = nullptr;
        oq.M[o] = oq.N[o] = oq.ldc[o] = oq.splits[o] = oq.accumulate[o] = 0;
        oq.blk_prefix[o + 1] = oq.blk_prefix[g_ndeferred];
    }
    const int nblk = oq.blk_prefix[g_ndeferred];
    g_ndeferred = 0;
    hipLaunchKernelGGL(gemm_tn_batch_reduce_kernel, dim3(nblk), dim3(256), 0, s, oq);
    MMDFN_CHECK_LAUNCH();
    return 0;
}

static bool riders_defer_reduce() {
#ifdef MMDFN_TUNING
    if (const char* e = getenv("MMDFN_RIDER_DEFER")) return atoi(e) != 0;     // A/B aid
#endif
    return true;
}

int mmdfn_riders_launched(hipStream_t s) {
    if (!g_rider.valid) return -1;
    g_rider.valid = false;
    const TnOuts& oq = g_rider.oq;
    // (two waiting stacks may not share a destination either: an earlier rider batch that wrote one of these gradients is
    // reduced first, by a launch of its own)
    bool clash = false;
    for (int d = 0; d < g_ndeferred && !clash; ++d)
        for (int o = 0; o < oq.n && !clash; ++o) {
            const DeferredOut& q = g_deferred[d];
            clash = (q.C != nullptr && q.C == oq.C[o]) ||
                    (q.colsum != nullptr && (q.colsum == oq.colsum[o] || q.colsum == oq.colsum2[o])) ||
                    (q.colsum2 != nullptr && (q.colsum2 == oq.colsum[o] || q.colsum2 == oq.colsum2[o]));
        }
    if (clash)
        if (int e = launch_deferred(s)) return e;
    if (riders_defer_reduce() && g_ndeferred + oq.n <= TN_MAXOUT) {
        for (int o = 0; o < oq.n; ++o)
            g_deferred[g_ndeferred++] = DeferredOut{oq.part[o], oq.colpart[o], oq.C[o], oq.colsum[o], oq.colsum2[o],
                                                    oq.M[o], oq.N[o], oq.ldc[o], oq.splits[o], oq.accumulate[o]};
        return 0;
    }
    hipLaunchKernelGGL(gemm_tn_batch_reduce_kernel, dim3(g_rider.nblk), dim3(256), 0, s, g_rider.oq);
    MMDFN_CHECK_LAUNCH();
    return 0;
}

static int tn_batch_impl(int nseg, const float* const* A, const float* const* B, const int* R, const int* lda,
                         const int* ldb, const int* bshift, const int* out, int nout, float* const* C,
                         float* const* colsum, float* const* colsum2, const int* M, const int* N,
                         const int* ldc, const int* accumulate, float* workspace, const TnExt& ext, void* stream,
                         bool stage = false) {
    if (nseg < 0 || nseg > TN_MAXSEG || nout < 0 || nout + ext.n > TN_MAXOUT || (nseg == 0) != (nout == 0) ||
        (nseg == 0 && ext.n == 0)) return -1;
    for (int e = 0; e < ext.n; ++e)
        if (ext.M[e] <= 0 || ext.N[e] < 0 || ext.splits[e] <= 0 || (ext.N[e] > 0 && (ext.part[e] == nullptr || ext.C[e] == nullptr ||
            ext.ldc[e] < ext.N[e])) || (ext.colsum[e] != nullptr && ext.colpart[e] == nullptr)) return -1;
    TnSegs sq;
    TnOuts oq;
    // pass 1: splits per output (its segments stack their slabs)
    int out_splits[TN_MAXOUT], seg_eff[TN_MAXSEG], seg_rps[TN_MAXSEG];
    bool seg_tall[TN_MAXSEG];
    const TallPlan plan = tall_plan(nseg, R, out, nout, M, N);
    const int rt = batch_rows_target(nseg, R, out, nout, M, N, plan.on);
    for (int o = 0; o < nout; ++o) {
        out_splits[o] = 0;
        if (M[o] <= 0 || N[o] <= 0 || (M[o] & 3) || (N[o] & 3) || ldc[o] < N[o]) return -1;
    }
    // the bf16-piece form takes the whole batch when every operand is 16-byte aligned (its loads are float4 like the tiled
    // form's, which the launch checks of that form never enforced on the base pointers)
    bool split_form = tns_enabled();
    for (int s = 0; s < nseg && split_form; ++s)
        if ((((uintptr_t)A[s] | (uintptr_t)B[s]) & 15) != 0 || (int64_t)R[s] * lda[s] >= (1ll << 30) ||
            (int64_t)R[s] * ldb[s] >= (1ll << 30) || R[s] >= (1 << 24))
            split_form = false;               // (its loads address rows by 32-bit byte offsets from the operand bases)
    const int rt_split = tns_rows_target(nseg, R, out, nout, M, N, stage);
    for (int s = 0; s < nseg; ++s) {
        const int o = out[s];
        if (o < 0 || o >= nout || R[s] <= 0 || (lda[s] & 3) || (ldb[s] & 3) || lda[s] < M[o] || ldb[s] < N[o]) return -1;
        seg_tall[s] = !split_form && plan.on && tn_tall_shape1(R[s], M[o], N[o]) && (bshift == nullptr || bshift[s] == 0) &&
                      (((uintptr_t)A[s] | (uintptr_t)B[s]) & 15) == 0;
        seg_eff[s] = split_form ? tns_eff_splits(R[s], rt_split, &seg_rps[s])
                     : seg_tall[s] ? tall_eff_splits(R[s], M[o], plan, &seg_rps[s]) : batch_eff_splits(R[s], M[o], N[o], rt, &seg_rps[s]);
        out_splits[o] += seg_eff[s];
    }
    // workspace layout: per output [splits][M][N] then [splits][M]
    float* ws = workspace;
    float* part_base[TN_MAXOUT];
    float* col_base[TN_MAXOUT];
    oq.n = nout;
    oq.blk_prefix[0] = 0;
    for (int o = 0; o < nout; ++o) {
        if (out_splits[o] == 0) return -1;   // an output nobody contributes to
        part_base[o] = ws;
        ws += (int64_t)out_splits[o] * M[o] * N[o];
        col_base[o] = ws;
        ws += (int64_t)out_splits[o] * M[o];
        oq.part[o] = part_base[o];
        oq.colpart[o] = col_base[o];
        oq.C[o] = C[o];
        oq.colsum[o] = colsum ? colsum[o] : nullptr;
        oq.colsum2[o] = colsum2 ? colsum2[o] : nullptr;
        oq.M[o] = M[o]; oq.N[o] = N[o]; oq.ldc[o] = ldc[o]; oq.splits[o] = out_splits[o];
        oq.accumulate[o] = accumulate ? accumulate[o] : 0;
        int nblk = (int)((((int64_t)M[o] * N[o] + M[o]) * (out_splits[o] > TN_REDUCE_WIDE ? 8 : 1) + 255) / 256);
        if (nblk > 256) nblk = 256;
        oq.blk_prefix[o + 1] = oq.blk_prefix[o] + nblk;
    }
    int ntot = nout + ext.n;
    for (int e = 0; e < ext.n; ++e) {
        const int o = nout + e;
        oq.part[o] = ext.part[e]; oq.colpart[o] = ext.colpart[e];
        oq.C[o] = ext.C[e]; oq.colsum[o] = ext.colsum[e]; oq.colsum2[o] = nullptr;
        oq.M[o] = ext.M[e]; oq.N[o] = ext.N[e]; oq.ldc[o] = ext.ldc[e]; oq.splits[o] = ext.splits[e];
        oq.accumulate[o] = ext.accumulate ? ext.accumulate[e] : 0;
        const int lanes = ext.splits[e] > TN_REDUCE_WIDE ? 8 : 1;
        int nblk = (int)((((int64_t)ext.M[e] * ext.N[e] + ext.M[e]) * lanes + 255) / 256);
        if (nblk > 256) nblk = 256;
        oq.blk_prefix[o + 1] = oq.blk_prefix[o] + nblk;
    }
    // deferred slab stacks of rider batches join this launch's reduction -- unless a destination of theirs is also written here
    // (two entries of one launch may not touch the same gradient: theirs is reduced first, by a launch of its own)
    if (g_ndeferred > 0 && !stage) {
        bool clash = ntot + g_ndeferred > TN_MAXOUT;
        for (int d = 0; d < g_ndeferred && !clash; ++d)
            for (int o = 0; o < ntot && !clash; ++o) {
                const DeferredOut& q = g_deferred[d];
                clash = (q.C != nullptr && q.C == oq.C[o]) ||
                        (q.colsum != nullptr && (q.colsum == oq.colsum[o] || q.colsum == oq.colsum2[o])) ||
                        (q.colsum2 != nullptr && (q.colsum2 == oq.colsum[o] || q.colsum2 == oq.colsum2[o]));
            }
        if (clash) {
            if (int e = launch_deferred((hipStream_t)stream)) return e;
        } else {
            for (int d = 0; d < g_ndeferred; ++d) put_out(oq, ntot + d, g_deferred[d]);
            ntot += g_ndeferred;
            g_ndeferred = 0;
        }
    }
    oq.n = ntot;
    for (int o = ntot; o < TN_MAXOUT; ++o) {
        oq.part[o] = oq.colpart[o] = nullptr; oq.C[o] = oq.colsum[o] = oq.colsum2[o] = nullptr;
        oq.M[o] = oq.N[o] = oq.ldc[o] = oq.splits[o] = oq.accumulate[o] = 0;
        oq.blk_prefix[o + 1] = oq.blk_prefix[ntot];
    }
    int used[TN_MAXOUT];
    for (int o = 0; o < nout; ++o) used[o] = 0;
    hipStream_t st = (hipStream_t)stream;
    if (nseg == 0) {       // nothing but foreign slab stacks
        hipLaunchKernelGGL(gemm_tn_batch_reduce_kernel, dim3(oq.blk_prefix[ntot]), dim3(256), 0, st, oq);
        MMDFN_CHECK_LAUNCH();
        return 0;
    }
    if (split_form) {
        // longest workgroups first
        int ord[TN_MAXSEG];
        for (int s = 0; s < nseg; ++s) ord[s] = s;
        auto wg_len = [&](int q) { return seg_rps[q] * (tns_wide(N[out[q]]) ? 3 : 2); };
        std::stable_sort(ord, ord + nseg, [&](int a, int b) { return wg_len(a) > wg_len(b); });
        TnSplitSegs tq;
        tq.n = nseg;
        tq.wg_prefix[0] = 0;
        for (int k = 0; k < nseg; ++k) {
            const int s = ord[k], o = out[s];
            int nb = 1, wd = 0;
            const int tiles = tns_tiles(M[o], N[o], &nb, &wd);
            tq.wide[k] = wd;
            tq.A[k] = A[s]; tq.B[k] = B[s];
            tq.part[k] = part_base[o] + (int64_t)used[o] * M[o] * N[o];
            tq.colpart[k] = (oq.colsum[o] != nullptr) ? col_base[o] + (int64_t)used[o] * M[o] : nullptr;
            used[o] += seg_eff[s];
            tq.R[k] = R[s]; tq.lda[k] = lda[s]; tq.ldb[k] = ldb[s]; tq.bshift[k] = bshift ? bshift[s] : 0;
            tq.rows_per_split[k] = seg_rps[s]; tq.splits[k] = seg_eff[s]; tq.tiles[k] = tiles; tq.nblocks[k] = nb;
            tq.M[k] = M[o]; tq.N[k] = N[o];
            tq.wg_prefix[k + 1] = tq.wg_prefix[k] + tiles * 8 * ((seg_eff[s] + 7) / 8);
        }
        for (int k = nseg; k < MMDFN_TNS_MAXSEG; ++k) {
            tq.A[k] = tq.B[k] = nullptr; tq.part[k] = tq.colpart[k] = nullptr;
            tq.R[k] = tq.lda[k] = tq.ldb[k] = tq.bshift[k] = tq.rows_per_split[k] = tq.splits[k] = tq.tiles[k] = tq.nblocks[k] = 0;
            tq.M[k] = tq.N[k] = 0; tq.wide[k] = 0;
            tq.wg_prefix[k + 1] = tq.wg_prefix[nseg];
        }
        if (stage && nseg <= MMDFN_RIDER_MAXSEG && !g_rider.valid) {
            g_rider.tq = tq;
            g_rider.oq = oq;
            g_rider.nblk = oq.blk_prefix[ntot];
            g_rider.valid = true;
            return 0;
        }
        if (int e = mmdfn_launch_gemm_tn_split(tq, st)) return e;
        hipLaunchKernelGGL(gemm_tn_batch_reduce_kernel, dim3(oq.blk_prefix[ntot]), dim3(256), 0, st, oq);
        MMDFN_CHECK_LAUNCH();
        return 0;
    }
    const bool use_wide = !tn_no_wide();
    // order: tall segments first, the 49-tile kind before the 14-tile kind (longest workgroups first; alternating the two kinds
    // measured 15 % slower), then the segments on the tiled bodies
    int order[TN_MAXSEG], no = 0, ntall = 0;
    for (int pass = 0; pass < 3; ++pass)
        for (int s = 0; s < nseg; ++s) {
            const int kind = !seg_tall[s] ? 2 : (tall_kind(M[out[s]]) == 7 ? 0 : 1);
            if (kind != pass) continue;
            order[no++] = s;
            if (seg_tall[s]) ++ntall;
        }
    size_t tall_lds = ntall > 0 ? (size_t)(2 * BR * LDA + 2 * BS_FLOATS) * sizeof(float) : 0;    // (the tiled bodies' stages)
    sq.n = nseg;
    sq.wg_prefix[0] = 0;
    for (int k = 0; k < nseg; ++k) {
        const int s = order[k];
        const int o = out[s];
        const bool wide = !seg_tall[s] && use_wide && tn_wide(N[o]);
        const int tnw = wide ? TNW : TN;
        const int tiles = seg_tall[s] ? tall_blocks(N[o]) : ((M[o] + TM - 1) / TM) * ((N[o] + tnw - 1) / tnw);
        sq.wide[k] = seg_tall[s] ? tall_block_width(N[o]) : (wide ? 1 : 0);
        sq.A[k] = A[s]; sq.B[k] = B[s];
        sq.part[k] = part_base[o] + (int64_t)used[o] * M[o] * N[o];
        sq.colpart[k] = (oq.colsum[o] != nullptr) ? col_base[o] + (int64_t)used[o] * M[o] : nullptr;
        used[o] += seg_eff[s];
        sq.R[k] = R[s]; sq.lda[k] = lda[s]; sq.ldb[k] = ldb[s]; sq.bshift[k] = bshift ? bshift[s] : 0;
        sq.rows_per_split[k] = seg_rps[s]; sq.tiles[k] = tiles; sq.M[k] = M[o]; sq.N[k] = N[o];
        sq.wg_prefix[k + 1] = sq.wg_prefix[k] + tiles * seg_eff[s];
        if (seg_tall[s]) {
            const int bw = tall_block_width(N[o]);
            const size_t need = 2 * (size_t)(((TBR * M[o] * 4 + 1023) & ~1023) + ((TBR * bw * 4 + 1023) & ~1023)) + 1024;
            if (need > tall_lds) tall_lds = need;
        }
    }
    for (int s = nseg; s < TN_MAXSEG; ++s) {
        sq.A[s] = sq.B[s] = nullptr; sq.part[s] = sq.colpart[s] = nullptr; sq.wide[s] = 0;
        sq.R[s] = sq.lda[s] = sq.ldb[s] = sq.bshift[s] = sq.rows_per_split[s] = sq.tiles[s] = sq.M[s] = sq.N[s] = 0;
        sq.wg_prefix[s + 1] = sq.wg_prefix[nseg];
    }
    if (ntall > 0) {
        void (*kern)(const TnSegs) = gemm_tn_tall_kernel<0>;
#ifdef MMDFN_TUNING
        if (const char* e = getenv("MMDFN_TN_TALL_ABL")) {       // tools/bench_gemm_tn_tall.py
            switch (atoi(e)) {
                case 1: kern = gemm_tn_tall_kernel<1>; break;
                case 2: kern = gemm_tn_tall_kernel<2>; break;
                case 4: kern = gemm_tn_tall_kernel<4>; break;
                case 6: kern = gemm_tn_tall_kernel<6>; break;
                case 7: kern = gemm_tn_tall_kernel<7>; break;
                case 8: kern = gemm_tn_tall_kernel<8>; break;
                case 12: kern = gemm_tn_tall_kernel<12>; break;
                default: break;
            }
        }
#endif
        if (int e = mmdfn_allow_big_lds(kern)) return e;
        hipLaunchKernelGGL(kern, dim3(sq.wg_prefix[nseg]), dim3(256), tall_lds, st, sq);
    } else {
        hipLaunchKernelGGL(gemm_tn_batch_kernel, dim3(sq.wg_prefix[nseg]), dim3(256), 0, st, sq);
    }
    MMDFN_CHECK_LAUNCH();
    hipLaunchKernelGGL(gemm_tn_batch_reduce_kernel, dim3(oq.blk_prefix[ntot]), dim3(256), 0, st, oq);
    MMDFN_CHECK_LAUNCH();
    return 0;
}

extern "C" int mmdfn_gemm_tn_batch(int nseg, const float* const* A, const float* const* B, const int* R, const int* lda,
                                   const int* ldb, const int* bshift, const int* out, int nout, float* const* C,
                                   float* const* colsum, float* const* colsum2, const int* M, const int* N,
                                   const int* ldc, const int* accumulate, float* workspace, void* stream) {
    if (nseg < 1 || nout < 1) return -1;
    const TnExt none = {0, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    return tn_batch_impl(nseg, A, B, R, lda, ldb, bshift, out, nout, C, colsum, colsum2, M, N, ldc, accumulate, workspace, none, stream);
}

// The batch is planned as usual but NOT launched when it can ride in the next GRU backward launch (bf16-piece form, at most
// MMDFN_RIDER_MAXSEG segments, nothing staged yet); otherwise it is launched now.  mmdfn_wgrad_riders_flush launches whatever is
// still staged (the GRU launch that followed was of another kind, or there was none).
extern "C" int mmdfn_wgrad_riders_stage(int nseg, const float* const* A, const float* const* B, const int* R, const int* lda,
                                        const int* ldb, const int* bshift, const int* out, int nout, float* const* C,
                                        float* const* colsum, float* const* colsum2, const int* M, const int* N,
                                        const int* ldc, const int* accumulate, float* workspace, void* stream) {
    if (nseg < 1 || nout < 1) return -1;
    const TnExt none = {0, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    return tn_batch_impl(nseg, A, B, R, lda, ldb, bshift, out, nout, C, colsum, colsum2, M, N, ldc, accumulate, workspace, none, stream,
                         true);
}

extern "C" int mmdfn_wgrad_riders_staged() { return g_rider.valid ? 1 : 0; }

// End of the backward pass: slab stacks of rider batches that no later reduction launch took are reduced now (discard != 0:
// forgotten instead -- a backward pass that raised left them behind).
extern "C" int mmdfn_wgrad_riders_drain(void* stream, int discard) {
    if (discard) { g_ndeferred = 0; g_rider.valid = false; return 0; }
    return launch_deferred((hipStream_t)stream);
}

extern "C" int mmdfn_wgrad_riders_flush(void* stream) {
    if (!g_rider.valid) return 0;
    if (int e = mmdfn_launch_gemm_tn_split(g_rider.tq, (hipStream_t)stream)) { g_rider.valid = false; return e; }
    return mmdfn_riders_launched((hipStream_t)stream);
}

extern "C" int mmdfn_gemm_tn_batch_ext(int nseg, const float* const* A, const float* const* B, const int* R, const int* lda,
                                       const int* ldb, const int* bshift, const int* out, int nout, float* const* C,
                                       float* const* colsum, float* const* colsum2, const int* M, const int* N,
                                       const int* ldc, const int* accumulate, float* workspace, int next,
                                       const float* const* ext_part, const float* const* ext_colpart, float* const* ext_C,
                                       float* const* ext_colsum, const int* ext_M, const int* ext_N, const int* ext_ldc,
                                       const int* ext_splits, const int* ext_accumulate, void* stream) {
    if (next < 0 || (next > 0 && (!ext_part || !ext_colpart || !ext_C || !ext_colsum || !ext_M || !ext_N || !ext_ldc || !ext_splits)))
        return -1;
    const TnExt ext = {next, ext_part, ext_colpart, ext_C, ext_colsum, ext_M, ext_N, ext_ldc, ext_splits, ext_accumulate};
    return tn_batch_impl(nseg, A, B, R, lda, ldb, bshift, out, nout, C, colsum, colsum2, M, N, ldc, accumulate, workspace, ext, stream);
}
