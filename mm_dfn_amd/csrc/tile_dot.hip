// Tile-pattern dot products  tiles_{i,m}[p,q] = X[(m,p),:] . Y[(m,q),:].
//
//  EPI 0 (outer):  gradient of the propagate step w.r.t. the stored adjacency
//                  entries (dA = dOut . H^T restricted to the block-tile
//                  pattern; the reference's SpmmBackward produces the dense
//                  (MN x MN) matrix, model_GCN.py:178).
//  EPI 1 (gram):   cosine Gram of the unit feature rows + angular similarity
//                  + row degree (model_mm.py:145-151, 176).
//
// Both operands are k-contiguous, so both MFMA fragments are loaded straight
// from global memory with one 16-byte load per lane per 16-wide k-chunk
// (lane (row, g) holds k = k0+4g .. k0+4g+3; the same k-permutation on A and
// B).  The A strip (16 rows x K) of each wave stays in registers for the whole
// sweep over q; no LDS is needed.
#include "mmdfn_internal.h"
#include "../../include/mmdfn_hip.h"
#include <stdlib.h>

namespace {

template <int NW, int KC, int EPI>
__global__ __launch_bounds__(64 * NW) void tile_dot_kernel(
    const float* __restrict__ X, const float* __restrict__ Y, float* __restrict__ out_tiles,
    float* __restrict__ out_aux, float* __restrict__ deg, const int32_t* __restrict__ dia_len,
    const int32_t* __restrict__ row_start, const int64_t* __restrict__ tile_base, int M, int N, int K,
    int ldx, int ldy, int max_rb, int accumulate) {
    constexpr int BM = 16 * NW;
    const int i = blockIdx.x / max_rb;
    const int rb = blockIdx.x % max_rb;
    const int m = blockIdx.y;
    const int L = dia_len[i];
    const int r0 = rb * BM;
    if (r0 >= L) return;
    const int ld = (L + 3) & ~3;
    const int rs = row_start[i];
    const int64_t toff = tile_base[i] + (int64_t)m * L * ld;
    const float* Xm = X + ((int64_t)m * N + rs) * ldx;
    const float* Ym = Y + ((int64_t)m * N + rs) * ldy;

    const int lane = threadIdx.x & 63;
    const int w = threadIdx.x >> 6;
    const int frow = lane & 15;
    const int g = lane >> 4;
    const int prow = r0 + 16 * w + frow;
    if (r0 + 16 * w >= L) return;  // whole wave out of range (no barriers in this kernel)

    float4 a[KC];
#pragma unroll
    for (int kc = 0; kc < KC; ++kc) {
        const int k = 16 * kc + 4 * g;
        a[kc] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (prow < L && k < K) a[kc] = *reinterpret_cast<const float4*>(Xm + (int64_t)prow * ldx + k);
    }

    float rowsum[4] = {0.f, 0.f, 0.f, 0.f};
    for (int q0 = 0; q0 < ld; q0 += 16) {
        const int qrow = q0 + frow;
        f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kc = 0; kc < KC; ++kc) {
            const int k = 16 * kc + 4 * g;
            float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
            if (qrow < L && k < K) b = *reinterpret_cast<const float4*>(Ym + (int64_t)qrow * ldy + k);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[kc].x, b.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[kc].y, b.y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[kc].z, b.z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[kc].w, b.w, acc, 0, 0, 0);
        }
        // C/D layout: col = lane&15 (q), row = 4g + r (p)
        const int q = q0 + frow;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int p = r0 + 16 * w + 4 * g + r;
            const bool ok = (p < L) && (q < ld);
            const int64_t off = toff + (int64_t)p * ld + q;
            if (EPI == 0) {
                if (ok) {
                    float v = acc[r];
                    if (accumulate) v += out_tiles[off];
                    out_tiles[off] = v;
                }
            } else {
                float s = 0.f;
                if (ok) {
                    const float gq = (q < L) ? acc[r] : 0.f;
                    s = (q < L) ? mmdfn_sim(gq) : 0.f;
                    out_aux[off] = gq;    // raw cosine (saved for backward)
                    out_tiles[off] = s;   // raw similarity; normalised by a later kernel
                }
                // reduce over the 16 lanes (q) that share this row
                s += __shfl_xor(s, 1, 64);
                s += __shfl_xor(s, 2, 64);
                s += __shfl_xor(s, 4, 64);
                s += __shfl_xor(s, 8, 64);
                rowsum[r] += s;
            }
        }
    }
    if (EPI == 1 && frow == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int p = r0 + 16 * w + 4 * g + r;
            if (p < L) deg[(int64_t)m * N + rs + p] += rowsum[r];  // single writer per row
        }
    }
}

// dcross_{mn}[r] (+)= X[(m,r)].Y[(n,r)] + X[(n,r)].Y[(m,r)]
// Four rows per wave: a 16-lane group owns one row, lane j of the group holds elements j, j+16, ... of every
// modality's X and Y row in registers (each loaded ONCE), the M(M-1)/2 pair products are formed from registers and
// reduced over the 16 lanes with four DPP steps -- all four rows of the wave share every instruction.  (First
// version: one row per wave, rows re-read from L1 for every pair, 15 full-wave reductions per row: 40 us at L=512,
// M=6.)
template <int MMAX, int KSL>
__device__ __forceinline__ void cross_dot_block(int block, const float* __restrict__ X, const float* __restrict__ Y,
                                                float* __restrict__ dcross, int M, int N, int K, int ldx, int ldy,
                                                int accumulate) {
    const int lane = threadIdx.x & 63;
    const int j = lane & 15;
    const int row = (block * 4 + (threadIdx.x >> 6)) * 4 + (lane >> 4);
    const bool rok = row < N;
    const int rowc = rok ? row : N - 1;
    float xv[MMAX][KSL], yv[MMAX][KSL];
#pragma unroll
    for (int m = 0; m < MMAX; ++m)
#pragma unroll
        for (int sidx = 0; sidx < KSL; ++sidx) {
            const int k = j + 16 * sidx;
            const bool ok = (m < M) && (k < K);
            xv[m][sidx] = ok ? X[((int64_t)m * N + rowc) * ldx + k] : 0.f;
            yv[m][sidx] = ok ? Y[((int64_t)m * N + rowc) * ldy + k] : 0.f;
        }
#pragma unroll
    for (int m = 0; m < MMAX; ++m)
#pragma unroll
        for (int n = m + 1; n < MMAX; ++n) {
            if (n >= M) continue;
            float s = 0.f;
#pragma unroll
            for (int sidx = 0; sidx < KSL; ++sidx) s += xv[m][sidx] * yv[n][sidx] + xv[n][sidx] * yv[m][sidx];
            s += __shfl_xor(s, 1, 64);
            s += __shfl_xor(s, 2, 64);
            s += __shfl_xor(s, 4, 64);
            s += __shfl_xor(s, 8, 64);
            if (j == 0 && rok) {
                const int64_t o = (int64_t)mmdfn_pair_index(m, n, M) * N + row;
                dcross[o] = accumulate ? dcross[o] + s : s;
            }
        }
}

template <int MMAX, int KSL>
__global__ __launch_bounds__(256) void cross_dot_kernel(const float* __restrict__ X, const float* __restrict__ Y,
                                                        float* __restrict__ dcross, int M, int N, int K,
                                                        int ldx, int ldy, int accumulate) {
    cross_dot_block<MMAX, KSL>(blockIdx.x, X, Y, dcross, M, N, K, ldx, ldy, accumulate);
}

// ---------------------------------------------------------------------------------------------
// v2 (K <= 208): each wave sweeps q in groups of four 16-column tiles.
//   * B fragments of the next tile are prefetched into a second register set while the current tile's
//     MFMAs run;
//   * the 16 x 64 result block goes through a wave-private LDS patch so that it leaves as 16-byte,
//     row-contiguous stores (256 B per row instead of 64-byte column fragments) and the Gram epilogue
//     (similarity, saved cosine, row degree) works on whole float4 rows;
//   * blockIdx % 8 == dialogue % 8 (all tiles of a dialogue share one XCD's L2, like the propagate kernel).
template <int KC, int EPI>
__global__ __launch_bounds__(256) void tile_dot_v2_kernel(
    const float* __restrict__ X, const float* __restrict__ Y, float* __restrict__ out_tiles,
    float* __restrict__ out_aux, float* __restrict__ deg, const int32_t* __restrict__ dia_len,
    const int32_t* __restrict__ row_start, const int64_t* __restrict__ tile_base, int B, int M, int N, int K,
    int ldx, int ldy, int max_rb, int accumulate, float* __restrict__ dcross, int tile_blocks, int qsplit) {
    constexpr int BM = 64;
    constexpr int LDP = 68;                       // patch row stride (floats), 16-byte aligned rows
    __shared__ __attribute__((aligned(16))) float patch[4][16 * LDP];

    if (EPI == 0 && (int)blockIdx.x >= tile_blocks) {
        // the blocks behind the tiles: the cross-modal diagonals of the same outer product (M <= 3), so that the
        // adjacency gradient of a layer is ONE launch
        cross_dot_block<3, KC>((int)blockIdx.x - tile_blocks, X, Y, dcross, M, N, K, ldx, ldy, accumulate);
        return;
    }
    // qsplit (EPI 0 only): the column tiles of a row block are dealt to qsplit workgroups in groups of four -- a wave that sweeps
    // all seven tiles of a 110-column tile row alone is a serial chain of seven operand round trips (96 workgroups at cfg2)
    const int Rd = M * max_rb * qsplit;
    const int bid = blockIdx.x;
    const int yq = bid >> 3;
    const int i = (yq / Rd) * 8 + (bid & 7);
    if (i >= B) return;
    const int rho = yq % Rd;
    const int m = rho / (max_rb * qsplit);
    const int rq = rho - m * (max_rb * qsplit);
    const int rb = rq / qsplit;
    const int chunk = rq - rb * qsplit;
    const int L = dia_len[i];
    const int r0 = rb * BM;
    if (r0 >= L) return;
    const int ld = (L + 3) & ~3;
    const int rs = row_start[i];
    const int64_t toff = tile_base[i] + (int64_t)m * L * ld;
    const float* Xm = X + ((int64_t)m * N + rs) * ldx;
    const float* Ym = Y + ((int64_t)m * N + rs) * ldy;

    const int lane = threadIdx.x & 63;
    const int w = threadIdx.x >> 6;
    const int fi = lane & 15;
    const int g = lane >> 4;
    const int p0 = r0 + 16 * w;
    if (p0 >= L) return;                          // whole wave out of range (no workgroup barriers below)
    float* pw = patch[w];

    float4 a[KC];
    {
        const int prow = p0 + fi;
        const float* xp = Xm + (int64_t)(prow < L ? prow : L - 1) * ldx;
#pragma unroll
        for (int kc = 0; kc < KC; ++kc) {
            const int k = 16 * kc + 4 * g;
            const float4 v = *reinterpret_cast<const float4*>(xp + (k < K ? k : K - 4));
            const bool ok = (prow < L) && (k < K);
            a[kc] = make_float4(ok ? v.x : 0.f, ok ? v.y : 0.f, ok ? v.z : 0.f, ok ? v.w : 0.f);
        }
    }
    float4 bset[2][KC];
    auto load_b = [&](int set, int qt) {
        const int qrow = 16 * qt + fi;
        const float* yp = Ym + (int64_t)(qrow < L ? qrow : L - 1) * ldy;
#pragma unroll
        for (int kc = 0; kc < KC; ++kc) {
            const int k = 16 * kc + 4 * g;
            bset[set][kc] = *reinterpret_cast<const float4*>(yp + (k < K ? k : K - 4));
        }
    };
    const int nqt = (ld + 15) / 16;
    float rowsum[4] = {0.f, 0.f, 0.f, 0.f};

#define TD_TILE(SET, QT, SLOT)                                                                         \
    do {                                                                                               \
        if ((QT) + 1 < nqt) load_b(1 - (SET), (QT) + 1);                                               \
        const bool qok = (16 * (QT) + fi) < L;                                                         \
        f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};                                                       \
        __builtin_amdgcn_sched_barrier(0);                                                             \
        _Pragma("unroll") for (int kc = 0; kc < KC; ++kc) {                                            \
            const bool ok = qok && (16 * kc + 4 * g < K);                                              \
            const float4 b = bset[SET][kc];                                                            \
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[kc].x, ok ? b.x : 0.f, acc, 0, 0, 0);          \
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[kc].y, ok ? b.y : 0.f, acc, 0, 0, 0);          \
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[kc].z, ok ? b.z : 0.f, acc, 0, 0, 0);          \
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[kc].w, ok ? b.w : 0.f, acc, 0, 0, 0);          \
        }                                                                                              \
        _Pragma("unroll") for (int r = 0; r < 4; ++r) pw[(4 * g + r) * LDP + 16 * (SLOT) + fi] = acc[r]; \
    } while (0)

    if (4 * chunk >= nqt) return;                  // (no barriers below)
    load_b(0, 4 * chunk);
    for (int q0t = 4 * chunk; q0t < nqt; q0t += 4 * qsplit) {
        if (qsplit > 1 && q0t != 4 * chunk) load_b(0, q0t);      // (the next group of this chunk is not the neighbouring tile)
        TD_TILE(0, q0t, 0);
        if (q0t + 1 < nqt) TD_TILE(1, q0t + 1, 1);
        if (q0t + 2 < nqt) TD_TILE(0, q0t + 2, 2);
        if (q0t + 3 < nqt) TD_TILE(1, q0t + 3, 3);
        __builtin_amdgcn_wave_barrier();
        // 16 rows x 64 columns -> 256 float4: lane handles (row = g + 4e, c4 = fi)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int row = g + 4 * e;
            const int p = p0 + row;
            const int q = 16 * q0t + 4 * fi;
            const bool ok = (p < L) && (q < ld) && (4 * fi < 16 * (nqt - q0t));
            const float4 v = *reinterpret_cast<const float4*>(&pw[row * LDP + 4 * fi]);
            const int64_t off = toff + (int64_t)p * ld + q;
            float s4 = 0.f;
            if (EPI == 0) {
                if (ok) {
                    float4 o = v;
                    if (accumulate) {
                        const float4 old = *reinterpret_cast<const float4*>(out_tiles + off);
                        o.x += old.x; o.y += old.y; o.z += old.z; o.w += old.w;
                    }
                    *reinterpret_cast<float4*>(out_tiles + off) = o;
                }
            } else {
                if (ok) {
                    float4 c, sv;
                    c.x = (q + 0 < L) ? v.x : 0.f; c.y = (q + 1 < L) ? v.y : 0.f;
                    c.z = (q + 2 < L) ? v.z : 0.f; c.w = (q + 3 < L) ? v.w : 0.f;
                    sv.x = (q + 0 < L) ? mmdfn_sim(c.x) : 0.f; sv.y = (q + 1 < L) ? mmdfn_sim(c.y) : 0.f;
                    sv.z = (q + 2 < L) ? mmdfn_sim(c.z) : 0.f; sv.w = (q + 3 < L) ? mmdfn_sim(c.w) : 0.f;
                    *reinterpret_cast<float4*>(out_aux + off) = c;      // raw cosine (saved for backward)
                    *reinterpret_cast<float4*>(out_tiles + off) = sv;   // raw similarity; normalised later
                    s4 = (sv.x + sv.y) + (sv.z + sv.w);
                }
                s4 += __shfl_xor(s4, 1, 64);
                s4 += __shfl_xor(s4, 2, 64);
                s4 += __shfl_xor(s4, 4, 64);
                s4 += __shfl_xor(s4, 8, 64);
                rowsum[e] += s4;
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
#undef TD_TILE
    if (EPI == 1 && fi == 0) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int p = p0 + g + 4 * e;
            if (p < L) deg[(int64_t)m * N + rs + p] += rowsum[e];  // single writer per row
        }
    }
}

template <int KC>
int launch_v2(const float* X, const float* Y, float* out_tiles, float* out_aux, float* deg, const int32_t* dia_len,
              const int32_t* row_start, const int64_t* tile_base, int B, int M, int N, int K, int ldx, int ldy,
              int max_len, int epi, int accumulate, hipStream_t s, float* dcross) {
    const int max_rb = (max_len + 63) / 64;
    int qsplit = 1;
    if (epi == 0 && max_len > 64 && ((B + 7) / 8) * 8 * M * max_rb <= 512) qsplit = 2;      // few row blocks: two workgroups per row block
#ifdef MMDFN_TUNING
    if (const char* e = getenv("MMDFN_TILEDOT_QSPLIT")) { const int v = atoi(e); if (epi == 0 && (v == 1 || v == 2)) qsplit = v; }
#endif
    const int tile_blocks = ((B + 7) / 8) * 8 * M * max_rb * qsplit;
    if (epi == 0)
        hipLaunchKernelGGL((tile_dot_v2_kernel<KC, 0>), dim3(tile_blocks + (dcross ? (N + 15) / 16 : 0)), dim3(256), 0, s, X, Y,
                           out_tiles, out_aux, deg, dia_len, row_start, tile_base, B, M, N, K, ldx, ldy, max_rb, accumulate,
                           dcross, tile_blocks, qsplit);
    else
        hipLaunchKernelGGL((tile_dot_v2_kernel<KC, 1>), dim3(tile_blocks), dim3(256), 0, s, X, Y, out_tiles, out_aux, deg,
                           dia_len, row_start, tile_base, B, M, N, K, ldx, ldy, max_rb, accumulate, nullptr, tile_blocks, 1);
    MMDFN_CHECK_LAUNCH();
    return 0;
}

template <int NW, int KC>
int launch(const float* X, const float* Y, float* out_tiles, float* out_aux, float* deg, const int32_t* dia_len,
           const int32_t* row_start, const int64_t* tile_base, int B, int M, int N, int K, int ldx, int ldy,
           int max_len, int epi, int accumulate, hipStream_t s) {
    const int BM = 16 * NW;
    const int max_rb = (max_len + BM - 1) / BM;
    dim3 grid(B * max_rb, M);
    dim3 block(64 * NW);
    if (epi == 0)
        hipLaunchKernelGGL((tile_dot_kernel<NW, KC, 0>), grid, block, 0, s, X, Y, out_tiles, out_aux, deg, dia_len,
                           row_start, tile_base, M, N, K, ldx, ldy, max_rb, accumulate);
    else
        hipLaunchKernelGGL((tile_dot_kernel<NW, KC, 1>), grid, block, 0, s, X, Y, out_tiles, out_aux, deg, dia_len,
                           row_start, tile_base, M, N, K, ldx, ldy, max_rb, accumulate);
    MMDFN_CHECK_LAUNCH();
    return 0;
}

// any M (<= 9) / K: one wave per row, rows streamed per pair
__global__ __launch_bounds__(256) void cross_dot_generic_kernel(const float* __restrict__ X, const float* __restrict__ Y,
                                                                float* __restrict__ dcross, int M, int N, int K,
                                                                int ldx, int ldy, int accumulate) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= N) return;
    for (int m = 0; m < M; ++m)
        for (int n = m + 1; n < M; ++n) {
            const float* xm = X + ((int64_t)m * N + row) * ldx;
            const float* xn = X + ((int64_t)n * N + row) * ldx;
            const float* ym = Y + ((int64_t)m * N + row) * ldy;
            const float* yn = Y + ((int64_t)n * N + row) * ldy;
            float s = 0.f;
            for (int k = lane; k < K; k += 64) s += xm[k] * yn[k] + xn[k] * ym[k];
            s = wave_sum(s);
            if (lane == 0) {
                const int64_t o = (int64_t)mmdfn_pair_index(m, n, M) * N + row;
                dcross[o] = accumulate ? dcross[o] + s : s;
            }
        }
}

}  // namespace

// fused_cross (in/out, may be NULL): on entry the dcross buffer the caller would like filled by the same launch; set
// to NULL on return when that happened (short dialogues, M <= 3: the v2 kernel), left alone otherwise.
static int launch_tile_dot(const float* X, const float* Y, float* out_tiles, float* out_aux, float* deg,
                           const int32_t* dia_len, const int32_t* row_start, const int64_t* tile_base,
                           int B, int M, int N, int K, int ldx, int ldy, int max_len, int epi, int accumulate,
                           hipStream_t s, float** fused_cross) {
    if (B <= 0 || M <= 0 || N <= 0 || K <= 0 || (K & 3) || max_len <= 0) return -1;
    if (ldx < K || ldy < K || (ldx & 3) || (ldy & 3)) return -1;
#ifdef MMDFN_TUNING
    const char* sp = getenv("MMDFN_TILEDOT_SPLIT");   // 1 = always the bf16-piece kernel, 0 = never
#else
    constexpr const char* sp = nullptr;
#endif
    const int mrb = (max_len + 127) / 128;
    // (K > 512 has no f32-MFMA instantiation: the register-resident A strip would not fit; the piece kernel walks K)
    const bool want_split = (K > 512) || (sp ? (sp[0] == '1') : (max_len >= 128 && (long)B * M * mrb * mrb >= 48));
    if (epi == 0 && want_split) {
        // long dialogues: 128 x 128 blocks on the bf16 matrix path (three exact bf16 pieces per operand, linear_split.hip)
        const int rc = mmdfn_launch_tile_dot_split(X, Y, out_tiles, dia_len, row_start, tile_base, B, M, N, K, ldx, ldy,
                                                   max_len, accumulate, s);
        if (rc != -2) return rc;
    }
#ifdef MMDFN_TUNING
    const char* e = getenv("MMDFN_TILEDOT_V1");   // A/B aid: the first-generation kernel
#else
    constexpr const char* e = nullptr;
#endif
    if (e == nullptr || e[0] == '0') {
        float* dc = nullptr;
        if (fused_cross && *fused_cross && epi == 0 && M > 1 && M <= 3 && K <= 208) {
            dc = *fused_cross;
            *fused_cross = nullptr;
        }
        if (K <= 112) return launch_v2<7>(X, Y, out_tiles, out_aux, deg, dia_len, row_start, tile_base, B, M, N, K, ldx, ldy, max_len, epi, accumulate, s, dc);
        if (K <= 208) return launch_v2<13>(X, Y, out_tiles, out_aux, deg, dia_len, row_start, tile_base, B, M, N, K, ldx, ldy, max_len, epi, accumulate, s, dc);
    }
    if (K <= 112) return launch<4, 7>(X, Y, out_tiles, out_aux, deg, dia_len, row_start, tile_base, B, M, N, K, ldx, ldy, max_len, epi, accumulate, s);
    if (K <= 208) return launch<4, 13>(X, Y, out_tiles, out_aux, deg, dia_len, row_start, tile_base, B, M, N, K, ldx, ldy, max_len, epi, accumulate, s);
    if (K <= 512) return launch<4, 32>(X, Y, out_tiles, out_aux, deg, dia_len, row_start, tile_base, B, M, N, K, ldx, ldy, max_len, epi, accumulate, s);
    return -1;
}

int mmdfn_launch_tile_dot(const float* X, const float* Y, float* out_tiles, float* out_aux, float* deg,
                          const int32_t* dia_len, const int32_t* row_start, const int64_t* tile_base,
                          int B, int M, int N, int K, int ldx, int ldy, int max_len, int epi, int accumulate,
                          hipStream_t s) {
    return launch_tile_dot(X, Y, out_tiles, out_aux, deg, dia_len, row_start, tile_base, B, M, N, K, ldx, ldy, max_len, epi,
                           accumulate, s, nullptr);
}

// cross-modal diagonals of one contraction piece (d <= 208 takes the register-resident kernels)
static void cross_piece(const float* X, const float* Y, float* dcross, int M, int N, int d, int ldx, int ldy, int accumulate,
                        hipStream_t s) {
#define CROSS_DOT(MM, KS) \
    hipLaunchKernelGGL((cross_dot_kernel<MM, KS>), dim3((N + 15) / 16), dim3(256), 0, s, X, Y, dcross, M, N, d, ldx, ldy, accumulate)
    if (M <= 3 && d <= 112) CROSS_DOT(3, 7);
    else if (M <= 3 && d <= 208) CROSS_DOT(3, 13);
    else if (M <= 6 && d <= 112) CROSS_DOT(6, 7);
    else if (M <= 6 && d <= 208) CROSS_DOT(6, 13);
    else
        hipLaunchKernelGGL(cross_dot_generic_kernel, dim3((N + 3) / 4), dim3(256), 0, s, X, Y, dcross, M, N, d, ldx, ldy,
                           accumulate);
#undef CROSS_DOT
}

extern "C" int mmdfn_tile_outer(const float* X, const float* Y, float* dtiles, float* dcross,
                                const int32_t* dia_len, const int32_t* row_start, const int64_t* tile_base,
                                int B, int M, int N, int d, int ldx, int ldy, int max_len, int accumulate,
                                void* stream) {
    hipStream_t s = (hipStream_t)stream;
    if (d <= 0 || (d & 3)) return -1;
    // A wide contraction (the GCN stack hands over all its layers at once: d = nl H, gcn_stack.py) is ONE launch on the bf16-piece
    // kernel, which walks K; the exact-f32 kernels keep their A strip in registers (K <= 208), so there the contraction index
    // is cut into pieces of <= 200 columns that accumulate (any cut of the index is exact up to the order of the sum).  The
    // cross-modal diagonals are cut the same way in both cases.
    const int mrb = (max_len + 127) / 128;
    const bool piece_kernel = (d > 512) || (max_len >= 128 && (long)B * M * mrb * mrb >= 48);
    constexpr int CUT = 200;
    if (piece_kernel || d <= 208) {
        float* dc = dcross;
        int rc = launch_tile_dot(X, Y, dtiles, nullptr, nullptr, dia_len, row_start, tile_base, B, M, N, d, ldx, ldy, max_len, 0,
                                 accumulate, s, &dc);
        if (rc) return rc;
        if (M > 1 && dc) {              // (NULL by now if the tile launch took the cross diagonals along)
            for (int c0 = 0; c0 < d; c0 += CUT) {
                const int kc = d - c0 <= CUT + 8 ? d - c0 : CUT;          // (no sliver at the end: 208 stays one piece)
                cross_piece(X + c0, Y + c0, dc, M, N, kc, ldx, ldy, (accumulate || c0 > 0) ? 1 : 0, s);
                if (kc != CUT) break;
            }
            MMDFN_CHECK_LAUNCH();
        }
        return 0;
    }
    for (int c0 = 0; c0 < d; c0 += CUT) {
        const int kc = d - c0 <= CUT + 8 ? d - c0 : CUT;
        const int acc = (accumulate || c0 > 0) ? 1 : 0;
        float* dc = dcross;
        int rc = launch_tile_dot(X + c0, Y + c0, dtiles, nullptr, nullptr, dia_len, row_start, tile_base, B, M, N, kc, ldx, ldy,
                                 max_len, 0, acc, s, &dc);
        if (rc) return rc;
        if (M > 1 && dc) {
            cross_piece(X + c0, Y + c0, dc, M, N, kc, ldx, ldy, acc, s);
            MMDFN_CHECK_LAUNCH();
        }
        if (kc != CUT) break;
    }
    return 0;
}
