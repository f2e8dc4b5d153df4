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
constexpr int TM = 64, TN = 64; // output tile per workgroup
constexpr int LDA = TM + 4;     // = 4 (mod 8)
constexpr int LDB = TN + 4;

// One workgroup: output tile `tile` of split `split`.  B is read at row r + bshift (rows outside [0, R) count as
// zero): the recurrent-weight gradient  dW_hh = sum_t dgh_t (x) h_{t-1}  pairs row t of dgh with row t-1 (forward
// direction) or t+1 (reverse) of the output sequence, and with the shift inside the kernel A still covers every row,
// so its column sums are the complete bias gradient.
__device__ __forceinline__ void gemm_tn_body(const float* __restrict__ A, const float* __restrict__ B,
                                             float* __restrict__ part, float* __restrict__ colpart, int R, int M, int N,
                                             int lda, int ldb, int rows_per_split, int bshift, int tile, int split) {
    __shared__ __attribute__((aligned(16))) float As[2][BR * LDA];
    __shared__ __attribute__((aligned(16))) float Bs[2][BR * LDB];
    const int nbn = (N + TN - 1) / TN;
    const int bm = tile / nbn;
    const int bn = tile - bm * nbn;
    const int r_begin = split * rows_per_split;
    const int r_end = min(R, r_begin + rows_per_split);
    const int m0 = bm * TM, n0 = bn * TN;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int wm = w & 1, wn = w >> 1;
    const int fi = lane & 15, g = lane >> 4;

    f32x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float csum[2] = {0.f, 0.f};

    // staging slots: BR*16 float4 per operand = 512 -> 2 per thread per operand
    int s_r[2], s_c[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const int idx = tid + e * 256;
        s_r[e] = idx >> 4;
        s_c[e] = (idx & 15) * 4;
    }
    float4 ra[2], rb[2];
    auto issue = [&](int r0) {
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int r = r0 + s_r[e];
            const int rc = r < r_end ? r : r_end - 1;
            const int rbs = rc + bshift;
            const int rbc = rbs < 0 ? 0 : (rbs < R ? rbs : R - 1);
            const int ca = m0 + s_c[e], cb = n0 + s_c[e];
            ra[e] = *reinterpret_cast<const float4*>(A + (int64_t)rc * lda + (ca < M ? ca : 0));
            rb[e] = *reinterpret_cast<const float4*>(B + (int64_t)rbc * ldb + (cb < N ? cb : 0));
        }
    };
    const int nchunks = (r_end - r_begin + BR - 1) / BR;
    if (nchunks > 0) issue(r_begin);
    for (int c = 0; c < nchunks; ++c) {
        const int r0 = r_begin + c * BR;
        float* as = As[c & 1];
        float* bs = Bs[c & 1];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const bool rok = (r0 + s_r[e]) < r_end;
            const int rbs = r0 + s_r[e] + bshift;
            const bool aok = rok && (m0 + s_c[e] < M), bok = rok && (n0 + s_c[e] < N) && rbs >= 0 && rbs < R;
            // M, N are multiples of 4 (checked by the launcher), so a float4 is fully inside or outside
            *reinterpret_cast<float4*>(&as[s_r[e] * LDA + s_c[e]]) =
                make_float4(aok ? ra[e].x : 0.f, aok ? ra[e].y : 0.f, aok ? ra[e].z : 0.f, aok ? ra[e].w : 0.f);
            *reinterpret_cast<float4*>(&bs[s_r[e] * LDB + s_c[e]]) =
                make_float4(bok ? rb[e].x : 0.f, bok ? rb[e].y : 0.f, bok ? rb[e].z : 0.f, bok ? rb[e].w : 0.f);
        }
        __syncthreads();
        if (c + 1 < nchunks) issue(r0 + BR);
#pragma unroll
        for (int ks = 0; ks < BR / 4; ++ks) {
            const int rr = 4 * ks + g;   // MFMA k index = row within the chunk
            float av[2], bv[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) av[i] = as[rr * LDA + 32 * wm + 16 * i + fi];
#pragma unroll
            for (int j = 0; j < 2; ++j) bv[j] = bs[rr * LDB + 32 * wn + 16 * j + fi];
            csum[0] += av[0];
            csum[1] += av[1];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], bv[j], acc[i][j], 0, 0, 0);
        }
    }
    // partial tile -> workspace [split][M][N]; C/D layout: col (n) = lane&15, row (m) = 4g + r
    float* P = part + (int64_t)split * M * N;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = n0 + 32 * wn + 16 * j + fi;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + 32 * wm + 16 * i + 4 * g + r;
                if (m < M && n < N) P[(int64_t)m * N + n] = acc[i][j][r];
            }
        }
    if (colpart != nullptr && bn == 0 && wn == 0) {
        // lanes with equal fi hold partial sums of the same column (different row residues g)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            float v = csum[i];
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            const int m = m0 + 32 * wm + 16 * i + fi;
            if (g == 0 && m < M) colpart[(int64_t)split * M + m] = v;
        }
    }
}

__global__ __launch_bounds__(256) void gemm_tn_kernel(const float* __restrict__ A, const float* __restrict__ B,
                                                      float* __restrict__ part, float* __restrict__ colpart,
                                                      int R, int M, int N, int lda, int ldb, int rows_per_split) {
    gemm_tn_body(A, B, part, colpart, R, M, N, lda, ldb, rows_per_split, 0, blockIdx.x, blockIdx.y);
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
    gemm_tn_body(gq.A[p], gq.B[p], gq.part[p], gq.colpart[p], gq.R[p], gq.M[p], gq.N[p], gq.lda[p], gq.ldb[p],
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

}  // namespace

extern "C" int mmdfn_gemm_tn_splits(int R, int M, int N) {
#ifdef MMDFN_TUNING
    if (const char* e = getenv("MMDFN_TN_SPLITS")) {   // tools/bench_gemm_tn.py
        const int v = atoi(e);
        if (v > 0) return v;
    }
#endif
    // measured on MI355X (tools/bench_gemm_tn.py): ~330-660 rows per split is the sweet spot for every hot-path
    // shape (R = 1.7k .. 10.5k, outputs 100x200 .. 600x200); more splits only inflate the slab reduction
    (void)M;
    (void)N;
    int s = (R + 329) / 330;
    if (s < 8) s = 8;
    if (s > 16) s = 16;
    const int max_s = (R + 2 * BR - 1) / (2 * BR);  // at least two staged chunks per split
    if (s > max_s) s = max_s;
    if (s < 1) s = 1;
    return s;
}

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
