// Weight-gradient contraction  C (M x N) = A^T B  on the bf16 matrix path: the batch form of gemm_tn.hip (every dW = dY^T X of a
// training step in one launch: autograd of nn.Linear / nn.GRU / nn.LSTM in the reference) with each fp32 operand cut exactly
// into three bf16 pieces and the six piece products of weight >= 2^-16 issued as v_mfma_f32_16x16x32_bf16 (fp32-level error,
// see propagate_split.hip for the arithmetic) instead of exact-f32 16x16x4 MFMAs (1/16 of the bf16 rate: the tiled kernel of
// gemm_tn.hip runs the cfg2 batch at 64 TFLOP/s).
//
// The contraction index is the ROW index r of both operands, i.e. both are k-major: an MFMA fragment (8 consecutive k of one
// column) is a column walk through memory.  So the operands are transposed on chip:
//   * a workgroup (8 waves) owns a 128 x 112 output tile over the row range of its split.  Per chunk of 32 rows every thread
//     loads two float4 of A (32 rows x 128 columns) and two of B (32 x 112) -- full rows, coalesced --, cuts them in registers
//     (5.5 VALU per element: each element is cut ONCE per workgroup and chunk, then read by 7 or 8 fragment loads) and writes
//     the pieces as three ROW-MAJOR bf16 planes per operand to LDS (ds_write_b64, 288-byte rows);
//   * fragments come out of the planes with ds_read_b64_tr_b16, the hardware transpose read: a 16-lane group reads a
//     [4 rows][16 columns] block and each lane receives one column of it.  A chunk is exactly one K = 32 MFMA step; lane group
//     g takes rows 4 g .. 4 g + 3 and 16 + 4 g .. 16 + 4 g + 3 of the chunk for BOTH operands (any assignment works as long as
//     it is the same on both sides).  Row stride 288 bytes = 72 dwords = 8 (mod 64): the 8 rows the 32 lanes of one LDS cycle
//     read fall on 8 different 8-bank groups -- conflict-free;
//   * wave w multiplies its 16 output rows (A columns 16 w ..) by all 7 column tiles: 6 + 42 transpose reads and 42 MFMAs per
//     chunk, 28 accumulator registers;
//   * the LDS stage is double-buffered (one s_barrier per chunk), global loads run two chunks ahead in two register sets.
//     Waves w and w + 4 share a SIMD: one of them cuts chunk c + 1 before its MFMAs of chunk c, the other after, so that the
//     SIMD's vector and matrix work overlap instead of alternating in lockstep.
// Row shifts (recurrent weights: row r of A pairs with row r + shift of B), row ranges, column sums of A (bias gradients,
// accumulated in fp32 by the cutting threads before the cut) and the slab workspace are those of the tiled batch kernel; the
// slab reduction kernel is shared.
#include "mmdfn_internal.h"
#include <stdlib.h>
#include <type_traits>

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
#define LDS_AS(T, p) ((__attribute__((address_space(3))) T*)(p))

constexpr int SK = MMDFN_TNS_BK;          // rows per chunk = K of one MFMA step
constexpr int TSM = MMDFN_TNS_TM;         // 128 output rows: 8 waves x 16
constexpr int TSN = MMDFN_TNS_TN;         // 112 output columns: 7 MFMA tiles
constexpr int NCT = TSN / 16;
constexpr int ROWB = 288;                 // bytes per plane row (128 bf16 + pad): 72 dwords = 8 (mod 64)
constexpr int PLANE_B = SK * ROWB;        // 9 216
constexpr int IMG_B = 3 * PLANE_B;        // the three planes of one operand
constexpr int STAGE_B = 2 * IMG_B;        // A image, B image: 55 296
constexpr int LDS_B = 2 * STAGE_B;        // 110 592

__device__ __forceinline__ float as_f(uint32_t u) { return __builtin_bit_cast(float, u); }
__device__ __forceinline__ uint32_t as_u(float f) { return __builtin_bit_cast(uint32_t, f); }

// four consecutive columns of one row -> their three bf16 pieces (leading 8, next 8, next 8 significant bits, cut by
// truncation: x = p1 + p2 + p3 + O(2^-24 x) exactly representable), packed in column order
__device__ __forceinline__ void cut4(float4 v, uint32_t himask, u32x2& p1, u32x2& p2, u32x2& p3) {
    float x0 = v.x, x1 = v.y, x2 = v.z, x3 = v.w;
    p1 = u32x2{__builtin_amdgcn_perm(as_u(x1), as_u(x0), 0x07060302u), __builtin_amdgcn_perm(as_u(x3), as_u(x2), 0x07060302u)};
    x0 -= as_f(as_u(x0) & himask); x1 -= as_f(as_u(x1) & himask); x2 -= as_f(as_u(x2) & himask); x3 -= as_f(as_u(x3) & himask);
    p2 = u32x2{__builtin_amdgcn_perm(as_u(x1), as_u(x0), 0x07060302u), __builtin_amdgcn_perm(as_u(x3), as_u(x2), 0x07060302u)};
    x0 -= as_f(as_u(x0) & himask); x1 -= as_f(as_u(x1) & himask); x2 -= as_f(as_u(x2) & himask); x3 -= as_f(as_u(x3) & himask);
    p3 = u32x2{__builtin_amdgcn_perm(as_u(x1), as_u(x0), 0x07060302u), __builtin_amdgcn_perm(as_u(x3), as_u(x2), 0x07060302u)};
}

__device__ __forceinline__ u32x4 tr_frag(uint32_t addr) {
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_AS(s16x4, (uintptr_t)addr));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_AS(s16x4, (uintptr_t)(addr + 16 * ROWB)));
    const u32x2 l2 = __builtin_bit_cast(u32x2, lo), h2 = __builtin_bit_cast(u32x2, hi);
    return u32x4{l2.x, l2.y, h2.x, h2.y};
}

__device__ __forceinline__ f32x4 mfma16(u32x4 a, u32x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// ABL (tuning build, timing only): 1 no cut (pieces = raw bits), 2 no MFMAs, 4 no fragment reads, 8 no global loads, 16 cycle
// stamps of waves 0 and 4 into `trace` (32 floats per workgroup), 32 phase-separated periods only
template <int ABL>
__global__ __launch_bounds__(512, 2) void gemm_tn_split_kernel(const TnSplitSegs sq, float* __restrict__ trace) {
    extern __shared__ __attribute__((aligned(16))) unsigned char tns_smem[];
    int p = 0;
    while (p + 1 < sq.n && (int)blockIdx.x >= sq.wg_prefix[p + 1]) ++p;
    const int local = blockIdx.x - sq.wg_prefix[p];
    const int round = (local >> 3) / sq.tiles[p];
    const int tile = (local >> 3) - round * sq.tiles[p];
    const int split = 8 * round + (local & 7);
    if (split >= sq.splits[p]) return;
    const int nbn = sq.nblocks[p];
    const int bm = tile / nbn, bn = tile - bm * nbn;
    const int R = sq.R[p], M = sq.M[p], N = sq.N[p], lda = sq.lda[p], ldb = sq.ldb[p], bshift = sq.bshift[p];
    const float* __restrict__ A = sq.A[p];
    const float* __restrict__ B = sq.B[p];
    const int m0 = bm * TSM, n0 = bn * TSN;
    const int r_begin = split * sq.rows_per_split[p];
    const int r_end = min(R, r_begin + sq.rows_per_split[p]);
    const int nchunks = (r_end - r_begin + SK - 1) / SK;

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int quad = tid & 31, lrow = tid >> 5;                      // staging slot e: chunk row lrow + 16 e, columns 4 quad ..
    const bool wave_on = m0 + 16 * w < M;                            // (a wave past M still stages its share of the chunk)
    float* __restrict__ colpart = (bn == 0) ? sq.colpart[p] : nullptr;

    const int ca = (m0 + 4 * quad < M) ? m0 + 4 * quad : 0;          // columns past M / N re-fetch valid ones: they only reach
    const int cb = (quad < TSN / 4 && n0 + 4 * quad < N) ? n0 + 4 * quad : 0;   // accumulator rows / columns that are never stored
    const uint32_t lds0 = (uint32_t)(uintptr_t)((__attribute__((address_space(3))) void*)tns_smem);
    const uint32_t wr = lds0 + lrow * ROWB + quad * 8;
    const uint32_t rd = lds0 + (4 * (lane >> 4) + ((lane & 15) >> 2)) * ROWB + (lane & 3) * 8;
    uint32_t himask;
    asm volatile("s_mov_b32 %0, 0xffff0000" : "=s"(himask));

    f32x4 acc[NCT];
#pragma unroll
    for (int j = 0; j < NCT; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float4 cs = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 na[2], nb[2];           // the chunk on its way (loads in flight)
    float4 va_[2], vb_[2];         // the chunk being staged
    // chunk c fully inside the split and inside B's shifted row range: no row needs masking, no address clamping
    auto inside = [&](int c) {
        const int r0 = r_begin + c * SK;
        return r0 + SK <= r_end && r0 + bshift >= 0 && r0 + SK - 1 + bshift < R;
    };
    // The four loads of chunk min(c, last).  ONE site, at the top of every period: loads defined on two paths of the loop meet in
    // phi copies at the back edge, and a copy of a register with a load in flight waits for it -- as does hipcc's loop-header
    // merge of two register sets filled in alternate periods (a 2 x unrolled, two-chunks-ahead version of this loop waited with
    // vmcnt(0) for loads issued a few instructions earlier in every other period).  Rows are clamped into the split / into B's
    // range; offsets are 32-bit from the scalar operand bases.
    auto issue = [&](int c) {
        if (ABL & 8) return;
        const int r0 = r_begin + (c < nchunks ? c : nchunks - 1) * SK + lrow;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int rc = min(r0 + 16 * e, r_end - 1);
            const int rbc = min(max(rc + bshift, 0), R - 1);
            const uint32_t oa = (__umul24((uint32_t)rc, (uint32_t)lda) + (uint32_t)ca) << 2;
            const uint32_t ob = (__umul24((uint32_t)rbc, (uint32_t)ldb) + (uint32_t)cb) << 2;
            na[e] = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(A) + oa);
            nb[e] = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(B) + ob);
        }
    };
    auto take = [&]() {                     // (waits for the loads; nothing younger is in flight at this point)
#pragma unroll
        for (int e = 0; e < 2; ++e) { va_[e] = na[e]; vb_[e] = nb[e]; }
    };
    auto cut = [&](int c, uint32_t stage_off) {
        const int r0 = r_begin + c * SK;
        const bool edge = !inside(c);
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            float4 va = va_[e], vb = vb_[e];
            if (ABL & 8) { va = make_float4(1.f + c, 2.f, 3.f, 4.f); vb = va; }
            if (edge) {
                const int r = r0 + lrow + 16 * e;
                const bool aok = r < r_end;
                const bool bok = aok && r + bshift >= 0 && r + bshift < R;
                va = make_float4(aok ? va.x : 0.f, aok ? va.y : 0.f, aok ? va.z : 0.f, aok ? va.w : 0.f);
                vb = make_float4(bok ? vb.x : 0.f, bok ? vb.y : 0.f, bok ? vb.z : 0.f, bok ? vb.w : 0.f);
            }
            cs.x += va.x; cs.y += va.y; cs.z += va.z; cs.w += va.w;
            u32x2 a1, a2, a3, b1, b2, b3;
            if (ABL & 1) {
                a1 = u32x2{as_u(va.x), as_u(va.y)}; a2 = u32x2{as_u(va.z), as_u(va.w)}; a3 = a1;
                b1 = u32x2{as_u(vb.x), as_u(vb.y)}; b2 = u32x2{as_u(vb.z), as_u(vb.w)}; b3 = b1;
            } else {
                cut4(va, himask, a1, a2, a3);
                cut4(vb, himask, b1, b2, b3);
            }
            const uint32_t d = wr + stage_off + e * 16 * ROWB;
            *LDS_AS(u32x2, (uintptr_t)d) = a1;
            *LDS_AS(u32x2, (uintptr_t)(d + PLANE_B)) = a2;
            *LDS_AS(u32x2, (uintptr_t)(d + 2 * PLANE_B)) = a3;
            *LDS_AS(u32x2, (uintptr_t)(d + IMG_B)) = b1;
            *LDS_AS(u32x2, (uintptr_t)(d + IMG_B + PLANE_B)) = b2;
            *LDS_AS(u32x2, (uintptr_t)(d + IMG_B + 2 * PLANE_B)) = b3;
        }
    };
    // the chunk's K = 32 step: products a3 b1, a2 b1, a1 b1 | a2 b2, a1 b2 | a1 b3 (smallest first)
    auto mma = [&](uint32_t stage_off) {
        const uint32_t base = rd + stage_off;
        u32x4 af[3], bf[3][NCT];
        if (ABL & 4) {
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                af[q] = u32x4{stage_off + q, 1u, 2u, 3u};
#pragma unroll
                for (int j = 0; j < NCT; ++j) bf[q][j] = u32x4{stage_off + j, q + 1u, 2u, 3u};
            }
        } else {
#pragma unroll
            for (int q = 0; q < 3; ++q) af[q] = tr_frag(base + q * PLANE_B + 32 * w);
#pragma unroll
            for (int q = 0; q < 3; ++q)
#pragma unroll
                for (int j = 0; j < NCT; ++j) bf[q][j] = tr_frag(base + IMG_B + q * PLANE_B + 32 * j);
        }
        if (ABL & 2) {
#pragma unroll
            for (int j = 0; j < NCT; ++j) acc[j][0] += as_f(af[0].x ^ af[1].y ^ af[2].z ^ bf[0][j].x ^ bf[1][j].y ^ bf[2][j].z);
            return;
        }
#pragma unroll
        for (int pc = 0; pc < 6; ++pc) {
            const int ai = (pc == 0) ? 2 : (pc == 1 || pc == 3) ? 1 : 0;
            const int bi = (pc < 3) ? 0 : (pc < 5) ? 1 : 2;
#pragma unroll
            for (int j = 0; j < NCT; ++j) acc[j] = mfma16(af[ai], bf[bi][j], acc[j]);
        }
    };

    long long tm[6] = {0, 0, 0, 0, 0, 0};
    long long ts[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // (ABL & 16) fused period: fragment reads issued, then after slots 6, 13, .. 41          // (ABL & 16) cut, issue, mma, barrier, prologue, epilogue
    auto now = [&]() -> long long {
        if (!(ABL & 16)) return 0;
        __builtin_amdgcn_sched_barrier(0);
        const long long t = (long long)__builtin_readcyclecounter();
        __builtin_amdgcn_sched_barrier(0);
        return t;
    };
    // Interior period (chunk c + 1 exists and needs no row masking): the 42 MFMAs of chunk c with the staging of chunk c + 1 in
    // their shadow, one slot per MFMA pinned by sched_barrier (a
    // 16 x 16 x 32 MFMA keeps the matrix pipe for 16 cycles = 4 issue slots, and the SIMD's other wave fills what this one leaves):
    //   before slot 0      fragment reads a1 a2 a3, b1[0..6]
    //   slots 0 .. 13      two transpose reads each: b2[j] (first used by slot 21 + j), then b3[j] (slot 35 + j)
    //   slots 0 .. 23      one cutting unit each (5 VALU, or 1 for the last piece): float4 f = slot / 6 (A rows e = 0, 1, then B
    //                      rows e = 0, 1), column pair (slot % 6) / 3, piece slot % 3
    //   slots 6 f + 6 .. 8 the three plane writes of float4 f
    // Waves past M (ON false) run the same slots without fragment reads and MFMAs.
    auto fused = [&](auto on_tag, uint32_t st_rd, uint32_t st_wr) {
        constexpr bool ON = decltype(on_tag)::value;
        const long long te = now();
        const uint32_t base = rd + st_rd;
        const uint32_t wbase = wr + st_wr;
        u32x4 af[3], bf[3][NCT];
        if (ON) {
#pragma unroll
            for (int q = 0; q < 3; ++q) af[q] = tr_frag(base + q * PLANE_B + 32 * w);
#pragma unroll
            for (int j = 0; j < NCT; ++j) bf[0][j] = tr_frag(base + IMG_B + 32 * j);
        }
        float x[4][4];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            x[e][0] = va_[e].x; x[e][1] = va_[e].y; x[e][2] = va_[e].z; x[e][3] = va_[e].w;
            x[2 + e][0] = vb_[e].x; x[2 + e][1] = vb_[e].y; x[2 + e][2] = vb_[e].z; x[2 + e][3] = vb_[e].w;
        }
        uint32_t pk[4][3][2];
        __builtin_amdgcn_sched_barrier(0);
        long long tp = now();
        ts[0] += tp - te;
#pragma unroll
        for (int s = 0; s < 6 * NCT; ++s) {
            const int pc = s / NCT, j = s - pc * NCT;
            if (ON) {
                const int ai = (pc == 0) ? 2 : (pc == 1 || pc == 3) ? 1 : 0;
                const int bi = (pc < 3) ? 0 : (pc < 5) ? 1 : 2;
                acc[j] = mfma16(af[ai], bf[bi][j], acc[j]);
                if (s < NCT) bf[1][s] = tr_frag(base + IMG_B + PLANE_B + 32 * s);
                else if (s < 2 * NCT) bf[2][s - NCT] = tr_frag(base + IMG_B + 2 * PLANE_B + 32 * (s - NCT));
            }
            if (s < 24) {
                const int f = s / 6, pr = (s % 6) / 3, st = s % 3;
                float& x0 = x[f][2 * pr];
                float& x1 = x[f][2 * pr + 1];
                if (st == 0 && f < 2) {
                    if (pr == 0) { cs.x += x0; cs.y += x1; } else { cs.z += x0; cs.w += x1; }
                }
                pk[f][st][pr] = __builtin_amdgcn_perm(as_u(x1), as_u(x0), 0x07060302u);
                if (st < 2) {
                    x0 -= as_f(as_u(x0) & himask);
                    x1 -= as_f(as_u(x1) & himask);
                }
            }
            if (s >= 6 && s < 30 && (s % 6) < 3) {
                const int f = s / 6 - 1, q = s % 6;
                const uint32_t d = wbase + (f & 1) * 16 * ROWB + (f >> 1) * IMG_B + q * PLANE_B;
                *LDS_AS(u32x2, (uintptr_t)d) = u32x2{pk[f][q][0], pk[f][q][1]};
            }
            __builtin_amdgcn_sched_barrier(0);
            if ((ABL & 16) && (s % NCT) == NCT - 1) {
                const long long t2 = now();
                ts[1 + s / NCT] += t2 - tp;
                tp = t2;
            }
        }
    };

    const long long t_begin = now();
    // prologue: chunk 0 staged into stage 0, chunk 1 on its way
    issue(0);
    take();
    __builtin_amdgcn_sched_barrier(0);
    issue(1);
    cut(0, 0);
    __syncthreads();
    tm[4] = now() - t_begin;
    // period c: chunk c (stage c & 1) is multiplied while chunk c + 1 (landed: loads issued one period ago) is staged into the
    // other stage and chunk c + 2 is requested
    auto run = [&](auto tag) {
#pragma unroll 1
        for (int c = 0; c < nchunks; ++c) {
            const uint32_t st_rd = (c & 1) * STAGE_B, st_wr = STAGE_B - st_rd;
            const long long s0_ = now();
            take();
            __builtin_amdgcn_sched_barrier(0);
            issue(c + 2);
            __builtin_amdgcn_sched_barrier(0);
            const long long s1_ = now();
            if (!(ABL & 32) && c + 1 < nchunks && inside(c + 1)) {
                fused(tag, st_rd, st_wr);
            } else {
                if (c + 1 < nchunks) cut(c + 1, st_wr);
                __builtin_amdgcn_sched_barrier(0);
                if (decltype(tag)::value) mma(st_rd);
            }
            const long long s2_ = now();
            __syncthreads();
            tm[1] += s1_ - s0_; tm[2] += s2_ - s1_; tm[3] += now() - s2_;
        }
    };
    if (wave_on) run(std::true_type{});
    else run(std::false_type{});
    const long long t_loop = now();

    // partial tile -> slab [split][M][N]; C/D layout of a 16 x 16 tile: column = lane & 15, row = 4 (lane >> 4) + r
    if (wave_on) {
        float* P = sq.part[p] + (int64_t)split * M * N;
        const int fi = lane & 15, g = lane >> 4;
#pragma unroll
        for (int j = 0; j < NCT; ++j) {
            const int n = n0 + 16 * j + fi;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + 16 * w + 4 * g + r;
                if (m < M && n < N) P[(int64_t)m * N + n] = acc[j][r];
            }
        }
    }
    if (colpart) {
        // the 16 threads of a column quad (one per staged row residue) hold partial sums of the same 4 columns
        float* red = reinterpret_cast<float*>(tns_smem);               // (every LDS read of the loop is behind its last barrier)
        *reinterpret_cast<float4*>(red + lrow * TSM + 4 * quad) = cs;
        __syncthreads();
        if (tid < TSM && m0 + tid < M) {
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < 16; ++k) s += red[k * TSM + tid];
            colpart[(int64_t)split * M + m0 + tid] = s;
        }
    }
    if ((ABL & 16) && trace && (tid == 0 || tid == 256)) {
        float* o = trace + (int64_t)blockIdx.x * 32 + (tid ? 16 : 0);
        const long long t_end = now();
        o[0] = (float)tm[0]; o[1] = (float)tm[1]; o[2] = (float)tm[2]; o[3] = (float)tm[3]; o[4] = (float)tm[4];
        o[5] = (float)(t_end - t_loop); o[6] = (float)nchunks; o[7] = (float)(t_end - t_begin);
#pragma unroll
        for (int k = 0; k < 7; ++k) o[8 + k] = (float)ts[k];
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Second form: the same tile, planes and arithmetic with FOUR waves of 32 x 112 (two row tiles each: 54 instead of 2 x 48
// transpose reads per 84 MFMAs), ONE LDS stage (55 KB) and two barriers per chunk -- two workgroups per CU, which overlap
// each other's staging (vector unit, LDS writes) and products (matrix pipe, LDS reads) instead of one workgroup's eight waves
// meeting at the same unit in lockstep.  Period c: products of chunk c | barrier | chunk c + 1 (requested one period ago) cut
// and written, chunk c + 2 requested | barrier.
template <int ABL>
__global__ __launch_bounds__(256, 2) void gemm_tn_split2_kernel(const TnSplitSegs sq, float* __restrict__ trace) {
    extern __shared__ __attribute__((aligned(16))) unsigned char tns_smem[];
    int p = 0;
    while (p + 1 < sq.n && (int)blockIdx.x >= sq.wg_prefix[p + 1]) ++p;
    const int local = blockIdx.x - sq.wg_prefix[p];
    const int round = (local >> 3) / sq.tiles[p];
    const int tile = (local >> 3) - round * sq.tiles[p];
    const int split = 8 * round + (local & 7);
    if (split >= sq.splits[p]) return;
    const int nbn = sq.nblocks[p];
    const int bm = tile / nbn, bn = tile - bm * nbn;
    const int R = sq.R[p], M = sq.M[p], N = sq.N[p], lda = sq.lda[p], ldb = sq.ldb[p], bshift = sq.bshift[p];
    const float* __restrict__ A = sq.A[p];
    const float* __restrict__ B = sq.B[p];
    const int m0 = bm * TSM, n0 = bn * TSN;
    const int r_begin = split * sq.rows_per_split[p];
    const int r_end = min(R, r_begin + sq.rows_per_split[p]);
    const int nchunks = (r_end - r_begin + SK - 1) / SK;

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int quad = tid & 31, lrow = tid >> 5;                      // staging slot e (0..3): chunk row lrow + 8 e, columns 4 quad ..
    const int mrem = M - (m0 + 32 * w);
    const int ntm = mrem <= 0 ? 0 : (mrem <= 16 ? 1 : 2);           // row tiles of this wave that hold output rows
    float* __restrict__ colpart = (bn == 0) ? sq.colpart[p] : nullptr;

    const int ca = (m0 + 4 * quad < M) ? m0 + 4 * quad : 0;
    const int cb = (quad < TSN / 4 && n0 + 4 * quad < N) ? n0 + 4 * quad : 0;
    const uint32_t lds0 = (uint32_t)(uintptr_t)((__attribute__((address_space(3))) void*)tns_smem);
    const uint32_t wr = lds0 + lrow * ROWB + quad * 8;
    const uint32_t rd = lds0 + (4 * (lane >> 4) + ((lane & 15) >> 2)) * ROWB + (lane & 3) * 8;
    uint32_t himask;
    asm volatile("s_mov_b32 %0, 0xffff0000" : "=s"(himask));

    f32x4 acc[2][NCT];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NCT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float4 cs = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 na[4], nb[4];

    auto inside = [&](int c) {
        const int r0 = r_begin + c * SK;
        return r0 + SK <= r_end && r0 + bshift >= 0 && r0 + SK - 1 + bshift < R;
    };
    // one site (see the first form): rows clamped into the split / into B's range, 32-bit offsets from the scalar bases
    auto issue = [&](int c) {
        if (ABL & 8) return;
        const int r0 = r_begin + (c < nchunks ? c : nchunks - 1) * SK + lrow;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int rc = min(r0 + 8 * e, r_end - 1);
            const int rbc = min(max(rc + bshift, 0), R - 1);
            const uint32_t oa = (__umul24((uint32_t)rc, (uint32_t)lda) + (uint32_t)ca) << 2;
            const uint32_t ob = (__umul24((uint32_t)rbc, (uint32_t)ldb) + (uint32_t)cb) << 2;
            na[e] = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(A) + oa);
            nb[e] = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(B) + ob);
        }
    };
    // cut the landed chunk c and write its planes; the loads of chunk c + 2 are issued as soon as the registers are free
    auto stage = [&](int c) {
        float4 va[4], vb[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) { va[e] = na[e]; vb[e] = nb[e]; }
        __builtin_amdgcn_sched_barrier(0);
        issue(c + 1);
        __builtin_amdgcn_sched_barrier(0);
        const int r0 = r_begin + c * SK;
        const bool edge = !inside(c);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float4 xa = va[e], xb = vb[e];
            if (ABL & 8) { xa = make_float4(1.f + c, 2.f, 3.f, 4.f); xb = xa; }
            if (edge) {
                const int r = r0 + lrow + 8 * e;
                const bool aok = r < r_end;
                const bool bok = aok && r + bshift >= 0 && r + bshift < R;
                xa = make_float4(aok ? xa.x : 0.f, aok ? xa.y : 0.f, aok ? xa.z : 0.f, aok ? xa.w : 0.f);
                xb = make_float4(bok ? xb.x : 0.f, bok ? xb.y : 0.f, bok ? xb.z : 0.f, bok ? xb.w : 0.f);
            }
            cs.x += xa.x; cs.y += xa.y; cs.z += xa.z; cs.w += xa.w;
            u32x2 a1, a2, a3, b1, b2, b3;
            if (ABL & 1) {
                a1 = u32x2{as_u(xa.x), as_u(xa.y)}; a2 = u32x2{as_u(xa.z), as_u(xa.w)}; a3 = a1;
                b1 = u32x2{as_u(xb.x), as_u(xb.y)}; b2 = u32x2{as_u(xb.z), as_u(xb.w)}; b3 = b1;
            } else {
                cut4(xa, himask, a1, a2, a3);
                cut4(xb, himask, b1, b2, b3);
            }
            const uint32_t d = wr + e * 8 * ROWB;
            *LDS_AS(u32x2, (uintptr_t)d) = a1;
            *LDS_AS(u32x2, (uintptr_t)(d + PLANE_B)) = a2;
            *LDS_AS(u32x2, (uintptr_t)(d + 2 * PLANE_B)) = a3;
            *LDS_AS(u32x2, (uintptr_t)(d + IMG_B)) = b1;
            *LDS_AS(u32x2, (uintptr_t)(d + IMG_B + PLANE_B)) = b2;
            *LDS_AS(u32x2, (uintptr_t)(d + IMG_B + 2 * PLANE_B)) = b3;
        }
    };
    // products a3 b1, a2 b1, a1 b1 | a2 b2, a1 b2 | a1 b3 (smallest first), NTM row tiles of the wave
    auto mma = [&](auto ntm_tag) {
        constexpr int NTM = decltype(ntm_tag)::value;
        u32x4 af[NTM][3], bf[3][NCT];
        if (ABL & 4) {
#pragma unroll
            for (int q = 0; q < 3; ++q) {
#pragma unroll
                for (int i = 0; i < NTM; ++i) af[i][q] = u32x4{(uint32_t)q, 1u, 2u, (uint32_t)i};
#pragma unroll
                for (int j = 0; j < NCT; ++j) bf[q][j] = u32x4{(uint32_t)j, q + 1u, 2u, 3u};
            }
        } else {
#pragma unroll
            for (int q = 0; q < 3; ++q)
#pragma unroll
                for (int i = 0; i < NTM; ++i) af[i][q] = tr_frag(rd + q * PLANE_B + 64 * w + 32 * i);
#pragma unroll
            for (int q = 0; q < 3; ++q)
#pragma unroll
                for (int j = 0; j < NCT; ++j) bf[q][j] = tr_frag(rd + IMG_B + q * PLANE_B + 32 * j);
        }
        if (ABL & 2) {
#pragma unroll
            for (int i = 0; i < NTM; ++i)
#pragma unroll
                for (int j = 0; j < NCT; ++j)
                    acc[i][j][0] += as_f(af[i][0].x ^ af[i][1].y ^ af[i][2].z ^ bf[0][j].x ^ bf[1][j].y ^ bf[2][j].z);
            return;
        }
#pragma unroll
        for (int pc = 0; pc < 6; ++pc) {
            const int ai = (pc == 0) ? 2 : (pc == 1 || pc == 3) ? 1 : 0;
            const int bi = (pc < 3) ? 0 : (pc < 5) ? 1 : 2;
#pragma unroll
            for (int j = 0; j < NCT; ++j)
#pragma unroll
                for (int i = 0; i < NTM; ++i) acc[i][j] = mfma16(af[i][ai], bf[bi][j], acc[i][j]);
        }
    };

    issue(0);
    stage(0);                       // (requests chunk 1)
    __syncthreads();
#pragma unroll 1
    for (int c = 0; c < nchunks; ++c) {
        if (ntm == 2) mma(std::integral_constant<int, 2>{});
        else if (ntm == 1) mma(std::integral_constant<int, 1>{});
        __syncthreads();
        if (c + 1 < nchunks) stage(c + 1);
        __syncthreads();
    }

    if (ntm > 0) {
        float* P = sq.part[p] + (int64_t)split * M * N;
        const int fi = lane & 15, g = lane >> 4;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < NCT; ++j) {
                const int n = n0 + 16 * j + fi;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = m0 + 32 * w + 16 * i + 4 * g + r;
                    if (m < M && n < N) P[(int64_t)m * N + n] = acc[i][j][r];
                }
            }
    }
    if (colpart) {
        // the 8 threads of a column quad (one per staged row residue) hold partial sums of the same 4 columns
        float* red = reinterpret_cast<float*>(tns_smem);
        *reinterpret_cast<float4*>(red + lrow * TSM + 4 * quad) = cs;
        __syncthreads();
        if (tid < TSM && m0 + tid < M) {
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < 8; ++k) s += red[k * TSM + tid];
            colpart[(int64_t)split * M + m0 + tid] = s;
        }
    }
    (void)trace;
}

// ---------------------------------------------------------------------------------------------------------------
// Third form: two groups of four waves (32 x 112 each, as in the second form) in ONE workgroup, working in opposite phases on
// alternate chunks of the same output tile.  Waves w and w + 4 share a SIMD and belong to different groups, so at any time every
// SIMD has one wave multiplying (matrix pipe, LDS reads) and one staging (loads, vector unit, LDS writes) -- the two independent
// workgroups of the second form drift INTO phase with each other (their ablations add up: removing the MFMAs saves their whole
// pipe time), here the phase is fixed by the step barrier.  Step k: group k & 1 multiplies chunk k out of its own stage, the other
// group cuts and writes chunk k + 1 into its stage and requests chunk k + 3; one barrier per step.  Each group accumulates its
// own chunks; the second group's accumulators are added through LDS at the end.
template <int ABL>
__global__ __launch_bounds__(512, 2) void gemm_tn_split3_kernel(const TnSplitSegs sq, float* __restrict__ trace) {
    extern __shared__ __attribute__((aligned(16))) unsigned char tns_smem[];
    int p = 0;
    while (p + 1 < sq.n && (int)blockIdx.x >= sq.wg_prefix[p + 1]) ++p;
    const int local = blockIdx.x - sq.wg_prefix[p];
    const int round = (local >> 3) / sq.tiles[p];
    const int tile = (local >> 3) - round * sq.tiles[p];
    const int split = 8 * round + (local & 7);
    if (split >= sq.splits[p]) return;
    const int nbn = sq.nblocks[p];
    const int bm = tile / nbn, bn = tile - bm * nbn;
    const int R = sq.R[p], M = sq.M[p], N = sq.N[p], lda = sq.lda[p], ldb = sq.ldb[p], bshift = sq.bshift[p];
    const float* __restrict__ A = sq.A[p];
    const float* __restrict__ B = sq.B[p];
    const int m0 = bm * TSM, n0 = bn * TSN;
    const int r_begin = split * sq.rows_per_split[p];
    const int r_end = min(R, r_begin + sq.rows_per_split[p]);
    const int nchunks = (r_end - r_begin + SK - 1) / SK;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wv >> 2, w = wv & 3;                             // group (chunk parity), wave inside the group
    const int quad = tid & 31, lrow = (tid >> 5) & 7;                // staging slot e (0..3): chunk row lrow + 8 e, columns 4 quad ..
    const int mrem = M - (m0 + 32 * w);
    const int ntm = mrem <= 0 ? 0 : (mrem <= 16 ? 1 : 2);           // row tiles of this wave that hold output rows
    float* __restrict__ colpart = (bn == 0) ? sq.colpart[p] : nullptr;

    const int ca = (m0 + 4 * quad < M) ? m0 + 4 * quad : 0;
    const int cb = (quad < TSN / 4 && n0 + 4 * quad < N) ? n0 + 4 * quad : 0;
    const uint32_t lds0 = (uint32_t)(uintptr_t)((__attribute__((address_space(3))) void*)tns_smem) + grp * STAGE_B;
    const uint32_t wr = lds0 + lrow * ROWB + quad * 8;
    const uint32_t rd = lds0 + (4 * (lane >> 4) + ((lane & 15) >> 2)) * ROWB + (lane & 3) * 8;
    uint32_t himask;
    asm volatile("s_mov_b32 %0, 0xffff0000" : "=s"(himask));

    f32x4 acc[2][NCT];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NCT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float4 cs = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 na[4], nb[4];

    auto inside = [&](int c) {
        const int r0 = r_begin + c * SK;
        return r0 + SK <= r_end && r0 + bshift >= 0 && r0 + SK - 1 + bshift < R;
    };
    // one site (see the first form): rows clamped into the split / into B's range, 32-bit offsets from the scalar bases
    auto issue = [&](int c) {
        if (ABL & 8) return;
        const int r0 = r_begin + (c < nchunks ? c : nchunks - 1) * SK + lrow;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int rc = min(r0 + 8 * e, r_end - 1);
            const int rbc = min(max(rc + bshift, 0), R - 1);
            const uint32_t oa = (__umul24((uint32_t)rc, (uint32_t)lda) + (uint32_t)ca) << 2;
            const uint32_t ob = (__umul24((uint32_t)rbc, (uint32_t)ldb) + (uint32_t)cb) << 2;
            na[e] = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(A) + oa);
            nb[e] = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(B) + ob);
        }
    };
    long long tm[6] = {0, 0, 0, 0, 0, 0};          // (ABL & 16) wait for the loads, issue, cut + write, products, barrier
    auto now = [&]() -> long long {
        if (!(ABL & 16)) return 0;
        __builtin_amdgcn_sched_barrier(0);
        const long long t = (long long)__builtin_readcyclecounter();
        __builtin_amdgcn_sched_barrier(0);
        return t;
    };
    // cut the landed chunk c and write its planes; the loads of the group's next chunk, c + 2, leave as soon as the registers
    // are free
    auto stage = [&](int c) {
        const long long t0 = now();
        float4 va[4], vb[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) { va[e] = na[e]; vb[e] = nb[e]; }
        if (ABL & 16) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const long long t1 = now();
        __builtin_amdgcn_sched_barrier(0);
        issue(c + 2);
        __builtin_amdgcn_sched_barrier(0);
        const long long t2 = now();
        tm[0] += t1 - t0; tm[1] += t2 - t1;
        const int r0 = r_begin + c * SK;
        const bool edge = !inside(c);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float4 xa = va[e], xb = vb[e];
            if (ABL & 8) { xa = make_float4(1.f + c, 2.f, 3.f, 4.f); xb = xa; }
            if (edge) {
                const int r = r0 + lrow + 8 * e;
                const bool aok = r < r_end;
                const bool bok = aok && r + bshift >= 0 && r + bshift < R;
                xa = make_float4(aok ? xa.x : 0.f, aok ? xa.y : 0.f, aok ? xa.z : 0.f, aok ? xa.w : 0.f);
                xb = make_float4(bok ? xb.x : 0.f, bok ? xb.y : 0.f, bok ? xb.z : 0.f, bok ? xb.w : 0.f);
            }
            cs.x += xa.x; cs.y += xa.y; cs.z += xa.z; cs.w += xa.w;
            u32x2 a1, a2, a3, b1, b2, b3;
            if (ABL & 1) {
                a1 = u32x2{as_u(xa.x), as_u(xa.y)}; a2 = u32x2{as_u(xa.z), as_u(xa.w)}; a3 = a1;
                b1 = u32x2{as_u(xb.x), as_u(xb.y)}; b2 = u32x2{as_u(xb.z), as_u(xb.w)}; b3 = b1;
            } else {
                cut4(xa, himask, a1, a2, a3);
                cut4(xb, himask, b1, b2, b3);
            }
            const uint32_t d = wr + e * 8 * ROWB;
            *LDS_AS(u32x2, (uintptr_t)d) = a1;
            *LDS_AS(u32x2, (uintptr_t)(d + PLANE_B)) = a2;
            *LDS_AS(u32x2, (uintptr_t)(d + 2 * PLANE_B)) = a3;
            *LDS_AS(u32x2, (uintptr_t)(d + IMG_B)) = b1;
            *LDS_AS(u32x2, (uintptr_t)(d + IMG_B + PLANE_B)) = b2;
            *LDS_AS(u32x2, (uintptr_t)(d + IMG_B + 2 * PLANE_B)) = b3;
        }
        tm[2] += now() - t2;
    };
    // products a3 b1, a2 b1, a1 b1 | a2 b2, a1 b2 | a1 b3 (smallest first), NTM row tiles of the wave
    auto mma = [&](auto ntm_tag) {
        constexpr int NTM = decltype(ntm_tag)::value;
        u32x4 af[NTM][3], bf[3][NCT];
        if (ABL & 4) {
#pragma unroll
            for (int q = 0; q < 3; ++q) {
#pragma unroll
                for (int i = 0; i < NTM; ++i) af[i][q] = u32x4{(uint32_t)q, 1u, 2u, (uint32_t)i};
#pragma unroll
                for (int j = 0; j < NCT; ++j) bf[q][j] = u32x4{(uint32_t)j, q + 1u, 2u, 3u};
            }
        } else {
#pragma unroll
            for (int q = 0; q < 3; ++q)
#pragma unroll
                for (int i = 0; i < NTM; ++i) af[i][q] = tr_frag(rd + q * PLANE_B + 64 * w + 32 * i);
#pragma unroll
            for (int q = 0; q < 3; ++q)
#pragma unroll
                for (int j = 0; j < NCT; ++j) bf[q][j] = tr_frag(rd + IMG_B + q * PLANE_B + 32 * j);
        }
        if (ABL & 2) {
#pragma unroll
            for (int i = 0; i < NTM; ++i)
#pragma unroll
                for (int j = 0; j < NCT; ++j)
                    acc[i][j][0] += as_f(af[i][0].x ^ af[i][1].y ^ af[i][2].z ^ bf[0][j].x ^ bf[1][j].y ^ bf[2][j].z);
            return;
        }
#pragma unroll
        for (int pc = 0; pc < 6; ++pc) {
            const int ai = (pc == 0) ? 2 : (pc == 1 || pc == 3) ? 1 : 0;
            const int bi = (pc < 3) ? 0 : (pc < 5) ? 1 : 2;
#pragma unroll
            for (int j = 0; j < NCT; ++j)
#pragma unroll
                for (int i = 0; i < NTM; ++i) acc[i][j] = mfma16(af[i][ai], bf[bi][j], acc[i][j]);
        }
    };

    // group g owns chunks g, g + 2, ..: chunk k is staged at step k - 1 and multiplied at step k
    issue(grp);
    if (grp == 0) stage(0);         // (requests chunk 2)
    __syncthreads();
    const long long t_begin = now();
#pragma unroll 1
    for (int k = 0; k < nchunks; ++k) {
        if ((k & 1) == grp) {
            const long long t0 = now();
            if (ntm == 2) mma(std::integral_constant<int, 2>{});
            else if (ntm == 1) mma(std::integral_constant<int, 1>{});
            if (ABL & 16) asm volatile("s_nop 0" ::: "memory");
            tm[3] += now() - t0;
        } else if (k + 1 < nchunks) {
            stage(k + 1);
        }
        const long long t3 = now();
        __syncthreads();
        tm[4] += now() - t3;
    }
    if ((ABL & 16) && trace && (tid == 0 || tid == 256)) {
        float* o = trace + (int64_t)blockIdx.x * 32 + (tid ? 16 : 0);
        o[0] = (float)tm[0]; o[1] = (float)tm[1]; o[2] = (float)tm[2]; o[3] = (float)tm[3]; o[4] = (float)tm[4];
        o[6] = (float)nchunks; o[7] = (float)(now() - t_begin);
    }
    // the second group's accumulators -> LDS (56 KB at the end of the stages, free behind the last barrier) -> added by the first
    if (grp == 1 && ntm > 0) {
        float* xch = reinterpret_cast<float*>(tns_smem + (LDS_B - 2 * NCT * 4096)) + (w * 64 + lane) * 4;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < NCT; ++j) *reinterpret_cast<f32x4*>(xch + (i * NCT + j) * 1024) = acc[i][j];
    }
    __syncthreads();
    if (grp == 0 && ntm > 0) {
        const float* xch = reinterpret_cast<const float*>(tns_smem + (LDS_B - 2 * NCT * 4096)) + (w * 64 + lane) * 4;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < NCT; ++j) acc[i][j] += *reinterpret_cast<const f32x4*>(xch + (i * NCT + j) * 1024);
    }

    if (grp == 0 && ntm > 0) {
        float* P = sq.part[p] + (int64_t)split * M * N;
        const int fi = lane & 15, g = lane >> 4;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < NCT; ++j) {
                const int n = n0 + 16 * j + fi;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = m0 + 32 * w + 16 * i + 4 * g + r;
                    if (m < M && n < N) P[(int64_t)m * N + n] = acc[i][j][r];
                }
            }
    }
    if (colpart) {
        // the 16 threads of a column quad (8 staged row residues x 2 groups) hold partial sums of the same 4 columns
        float* red = reinterpret_cast<float*>(tns_smem);               // (the first group's stage: its reads are behind a barrier)
        *reinterpret_cast<float4*>(red + (tid >> 5) * TSM + 4 * quad) = cs;
        __syncthreads();
        if (tid < TSM && m0 + tid < M) {
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < 16; ++k) s += red[k * TSM + tid];
            colpart[(int64_t)split * M + m0 + tid] = s;
        }
    }
    (void)trace;
}

}  // namespace

int mmdfn_launch_gemm_tn_split(const TnSplitSegs& sq, hipStream_t s) {
    void (*kern)(const TnSplitSegs, float*) = gemm_tn_split_kernel<0>;
    float* trace = nullptr;
#ifdef MMDFN_TUNING
    if (const char* e = getenv("MMDFN_TRACE_PTR")) trace = reinterpret_cast<float*>(strtoull(e, nullptr, 0));
    if (const char* e = getenv("MMDFN_TNS_ABL")) {       // tools/bench_gemm_tn_split.py
        switch (atoi(e)) {
            case 1: kern = gemm_tn_split_kernel<1>; break;
            case 2: kern = gemm_tn_split_kernel<2>; break;
            case 4: kern = gemm_tn_split_kernel<4>; break;
            case 6: kern = gemm_tn_split_kernel<6>; break;
            case 7: kern = gemm_tn_split_kernel<7>; break;
            case 8: kern = gemm_tn_split_kernel<8>; break;
            case 15: kern = gemm_tn_split_kernel<15>; break;
            case 16: kern = gemm_tn_split_kernel<16>; break;
            case 32: kern = gemm_tn_split_kernel<32>; break;
            case 48: kern = gemm_tn_split_kernel<48>; break;
            default: break;
        }
    }
#endif
    int form = 3;
#ifdef MMDFN_TUNING
    if (const char* e = getenv("MMDFN_TNS_FORM")) form = atoi(e);
#endif
    if (form == 3) {
        void (*k3)(const TnSplitSegs, float*) = gemm_tn_split3_kernel<0>;
#ifdef MMDFN_TUNING
        if (const char* e = getenv("MMDFN_TNS_ABL")) {
            switch (atoi(e)) {
                case 1: k3 = gemm_tn_split3_kernel<1>; break;
                case 2: k3 = gemm_tn_split3_kernel<2>; break;
                case 4: k3 = gemm_tn_split3_kernel<4>; break;
                case 7: k3 = gemm_tn_split3_kernel<7>; break;
                case 8: k3 = gemm_tn_split3_kernel<8>; break;
                case 15: k3 = gemm_tn_split3_kernel<15>; break;
                case 16: k3 = gemm_tn_split3_kernel<16>; break;
                default: break;
            }
        }
#endif
        if (int e = mmdfn_allow_big_lds(k3)) return e;
        hipLaunchKernelGGL(k3, dim3(sq.wg_prefix[sq.n]), dim3(512), LDS_B, s, sq, trace);
        MMDFN_CHECK_LAUNCH();
        return 0;
    }
    if (form == 2) {
        void (*k2)(const TnSplitSegs, float*) = gemm_tn_split2_kernel<0>;
#ifdef MMDFN_TUNING
        if (const char* e = getenv("MMDFN_TNS_ABL")) {
            switch (atoi(e)) {
                case 1: k2 = gemm_tn_split2_kernel<1>; break;
                case 2: k2 = gemm_tn_split2_kernel<2>; break;
                case 4: k2 = gemm_tn_split2_kernel<4>; break;
                case 7: k2 = gemm_tn_split2_kernel<7>; break;
                case 8: k2 = gemm_tn_split2_kernel<8>; break;
                case 15: k2 = gemm_tn_split2_kernel<15>; break;
                default: break;
            }
        }
#endif
        hipLaunchKernelGGL(k2, dim3(sq.wg_prefix[sq.n]), dim3(256), STAGE_B, s, sq, trace);
        MMDFN_CHECK_LAUNCH();
        return 0;
    }
    if (int e = mmdfn_allow_big_lds(kern)) return e;
    hipLaunchKernelGGL(kern, dim3(sq.wg_prefix[sq.n]), dim3(512), LDS_B, s, sq, trace);
    MMDFN_CHECK_LAUNCH();
    return 0;
}
