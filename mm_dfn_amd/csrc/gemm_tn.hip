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

static int batch_rows_target(int nseg, const int* R, const int* out, int nout, const int* M, const int* N) {
    double units = 0.0;                                  // sum over segments of output tiles x rows
    for (int s = 0; s < nseg; ++s) {
        const int o = out[s];
        if (o < 0 || o >= nout) continue;
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

extern "C" int64_t mmdfn_gemm_tn_batch_workspace(int nseg, const int* R, const int* out, int nout, const int* M,
                                                 const int* N) {
    int64_t total = 0;
    const int rt = batch_rows_target(nseg, R, out, nout, M, N);
    for (int s = 0; s < nseg; ++s) {
        const int o = out[s];
        if (o < 0 || o >= nout) return -1;
        total += (int64_t)batch_eff_splits(R[s], M[o], N[o], rt, nullptr) * ((int64_t)M[o] * N[o] + M[o]);
    }
    return total;
}

extern "C" int mmdfn_gemm_tn_batch(int nseg, const float* const* A, const float* const* B, const int* R, const int* lda,
                                   const int* ldb, const int* bshift, const int* out, int nout, float* const* C,
                                   float* const* colsum, float* const* colsum2, const int* M, const int* N,
                                   const int* ldc, const int* accumulate, float* workspace, void* stream) {
    if (nseg < 1 || nseg > TN_MAXSEG || nout < 1 || nout > TN_MAXOUT) return -1;
    TnSegs sq;
    TnOuts oq;
    // pass 1: splits per output (its segments stack their slabs)
    int out_splits[TN_MAXOUT], seg_eff[TN_MAXSEG], seg_rps[TN_MAXSEG];
    const int rt = batch_rows_target(nseg, R, out, nout, M, N);
    for (int o = 0; o < nout; ++o) {
        out_splits[o] = 0;
        if (M[o] <= 0 || N[o] <= 0 || (M[o] & 3) || (N[o] & 3) || ldc[o] < N[o]) return -1;
    }
    for (int s = 0; s < nseg; ++s) {
        const int o = out[s];
        if (o < 0 || o >= nout || R[s] <= 0 || (lda[s] & 3) || (ldb[s] & 3) || lda[s] < M[o] || ldb[s] < N[o]) return -1;
        seg_eff[s] = batch_eff_splits(R[s], M[o], N[o], rt, &seg_rps[s]);
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
        int nblk = (int)(((int64_t)M[o] * N[o] + M[o] + 255) / 256);
        if (nblk > 256) nblk = 256;
        oq.blk_prefix[o + 1] = oq.blk_prefix[o] + nblk;
    }
    for (int o = nout; o < TN_MAXOUT; ++o) {
        oq.part[o] = oq.colpart[o] = nullptr; oq.C[o] = oq.colsum[o] = oq.colsum2[o] = nullptr;
        oq.M[o] = oq.N[o] = oq.ldc[o] = oq.splits[o] = oq.accumulate[o] = 0;
        oq.blk_prefix[o + 1] = oq.blk_prefix[nout];
    }
    int used[TN_MAXOUT];
    for (int o = 0; o < nout; ++o) used[o] = 0;
    const bool use_wide = !tn_no_wide();
    sq.n = nseg;
    sq.wg_prefix[0] = 0;
    for (int s = 0; s < nseg; ++s) {
        const int o = out[s];
        const bool wide = use_wide && tn_wide(N[o]);
        const int tnw = wide ? TNW : TN;
        const int tiles = ((M[o] + TM - 1) / TM) * ((N[o] + tnw - 1) / tnw);
        sq.wide[s] = wide ? 1 : 0;
        sq.A[s] = A[s]; sq.B[s] = B[s];
        sq.part[s] = part_base[o] + (int64_t)used[o] * M[o] * N[o];
        sq.colpart[s] = (oq.colsum[o] != nullptr) ? col_base[o] + (int64_t)used[o] * M[o] : nullptr;
        used[o] += seg_eff[s];
        sq.R[s] = R[s]; sq.lda[s] = lda[s]; sq.ldb[s] = ldb[s]; sq.bshift[s] = bshift ? bshift[s] : 0;
        sq.rows_per_split[s] = seg_rps[s]; sq.tiles[s] = tiles; sq.M[s] = M[o]; sq.N[s] = N[o];
        sq.wg_prefix[s + 1] = sq.wg_prefix[s] + tiles * seg_eff[s];
    }
    for (int s = nseg; s < TN_MAXSEG; ++s) {
        sq.A[s] = sq.B[s] = nullptr; sq.part[s] = sq.colpart[s] = nullptr;
        sq.R[s] = sq.lda[s] = sq.ldb[s] = sq.bshift[s] = sq.rows_per_split[s] = sq.tiles[s] = sq.M[s] = sq.N[s] = 0;
        sq.wg_prefix[s + 1] = sq.wg_prefix[nseg];
    }
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(gemm_tn_batch_kernel, dim3(sq.wg_prefix[nseg]), dim3(256), 0, st, sq);
    MMDFN_CHECK_LAUNCH();
    hipLaunchKernelGGL(gemm_tn_batch_reduce_kernel, dim3(oq.blk_prefix[nout]), dim3(256), 0, st, oq);
    MMDFN_CHECK_LAUNCH();
    return 0;
}
