// Device side of gemm_tn_split.hip (see its header comment): the bf16-piece weight-gradient contraction of ONE workgroup, as a
// function, so that the same tiles can also run as RIDER workgroups of the GRU backward recurrence launch (gru.hip: the
// recurrence occupies one CU per sequence and leaves the other CUs idle for its whole duration).  A code header, included by
// exactly those two translation units.
#pragma once
#include "mmdfn_internal.h"
#include <type_traits>

namespace tnsb {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
#define LDS_AS(T, p) ((__attribute__((address_space(3))) T*)(p))

constexpr int SK = MMDFN_TNS_BK;          // rows per chunk = K of one MFMA step
constexpr int TSM = MMDFN_TNS_TM;         // 128 output rows: 4 waves x 32
constexpr int TSN = MMDFN_TNS_TN;         // 112 output columns: 7 MFMA tiles
constexpr int NCT = TSN / 16;
constexpr int ROWB = 288;                 // bytes per plane row (128 bf16 + pad): 72 dwords = 8 (mod 64)
constexpr int PLANE_B = SK * ROWB;        // 9 216
constexpr int IMG_B = 3 * PLANE_B;        // the three planes of one operand
constexpr int STAGE_B = 2 * IMG_B;        // A image, B image: 55 296
constexpr int LDS_B = 2 * STAGE_B;        // one stage per group: 110 592

__device__ __forceinline__ float as_f(uint32_t u) { return __builtin_bit_cast(float, u); }
__device__ __forceinline__ uint32_t as_u(float f) { return __builtin_bit_cast(uint32_t, f); }

// four consecutive columns of one row -> their three bf16 pieces (leading 8, next 8, next 8 significant bits, cut by
// truncation: x = p1 + p2 + p3 + O(2^-24 x), every piece exactly representable), packed in column order
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

// ---- the 128 x 224 form (round 5) --------------------------------------------------------------------------------------------
// The cut (5.5 VALU per staged element) cannot hide behind the MFMAs on this chip, so what is left is MFMAs per cut element: for
// outputs more than 112 columns wide one workgroup owns a 128 x 224 tile -- wave (wr, wc) multiplies rows 32 wr .. by columns
// 112 wc .. (the 2 x 7 tiles of the narrow form), the A planes are cut ONCE for both column halves and all 512 threads stage: 6
// float4 per thread and chunk instead of 8.  The planes of a chunk (A 27.6 KB + B 52.2 KB) leave no room for a second stage, so
// the eight waves run in lockstep -- stage chunk k, barrier, multiply it, barrier -- which costs nothing here: a SIMD's staging
// wave and its multiplying wave never overlapped anyway (see the header).
constexpr int WTN = 2 * MMDFN_TNS_TN;     // 224 output columns
constexpr int ROWB_W = 544;               // bytes per B plane row (224 bf16 + pad): 136 dwords = 8 (mod 64)
constexpr int PLANE_W = SK * ROWB_W;      // 17 408
constexpr int IMG_W = 3 * PLANE_W;        // 52 224 (A image IMG_B = 27 648 in front of it)
static_assert(IMG_B + IMG_W <= LDS_B, "the wide form's single stage fits the narrow form's LDS");

__device__ __forceinline__ u32x4 tr_frag_w(uint32_t addr) {
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_AS(s16x4, (uintptr_t)addr));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_AS(s16x4, (uintptr_t)(addr + 16 * ROWB_W)));
    const u32x2 l2 = __builtin_bit_cast(u32x2, lo), h2 = __builtin_bit_cast(u32x2, hi);
    return u32x4{l2.x, l2.y, h2.x, h2.y};
}

template <class SQ>
__device__ __forceinline__ void tns_wide_body(const SQ& sq, const int p, const int local, unsigned char* smem) {
    const int round = (local >> 3) / sq.tiles[p];
    const int tile = (local >> 3) - round * sq.tiles[p];
    const int split = 8 * round + (local & 7);
    if (split >= sq.splits[p]) return;
    const int nbn = sq.nblocks[p];
    const int bm = tile / nbn, bn = tile - bm * nbn;
    const int R = sq.R[p], M = sq.M[p], N = sq.N[p], lda = sq.lda[p], ldb = sq.ldb[p], bshift = sq.bshift[p];
    const float* __restrict__ A = sq.A[p];
    const float* __restrict__ B = sq.B[p];
    const int m0 = bm * TSM, n0 = bn * WTN;
    const int r_begin = split * sq.rows_per_split[p];
    const int r_end = min(R, r_begin + sq.rows_per_split[p]);
    const int nchunks = (r_end - r_begin + SK - 1) / SK;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wv & 3, wc = wv >> 2;                             // row group, column half of this wave
    const int quad = tid & 31, lrow = tid >> 5;                      // A slot e (0, 1): chunk row lrow + 16 e, columns 4 quad ..
    const int qb = tid & 63, rb = tid >> 6;                          // B slot e (0..3): chunk row rb + 8 e, columns 4 qb .. (qb < 56)
    const int mrem = M - (m0 + 32 * wr);
    const int ntm = mrem <= 0 ? 0 : (mrem <= 16 ? 1 : 2);
    float* __restrict__ colpart = (bn == 0) ? sq.colpart[p] : nullptr;

    const int ca = (m0 + 4 * quad < M) ? m0 + 4 * quad : 0;
    const int cb = (qb < WTN / 4 && n0 + 4 * qb < N) ? n0 + 4 * qb : 0;
    const uint32_t lds0 = (uint32_t)(uintptr_t)((__attribute__((address_space(3))) void*)smem);
    const uint32_t wra = lds0 + lrow * ROWB + quad * 8;
    const uint32_t wrb = lds0 + IMG_B + rb * ROWB_W + qb * 8;        // (lanes 56..63 land in the row padding)
    const uint32_t rsel = 4 * (lane >> 4) + ((lane & 15) >> 2);
    const uint32_t rda = lds0 + rsel * ROWB + (lane & 3) * 8 + 64 * wr;
    const uint32_t rdb = lds0 + IMG_B + rsel * ROWB_W + (lane & 3) * 8 + 2 * MMDFN_TNS_TN * wc;
    uint32_t himask;
    asm volatile("s_mov_b32 %0, 0xffff0000" : "=s"(himask));

    f32x4 acc[2][NCT];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NCT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float4 cs = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 na[2], nb[4];

    auto inside = [&](int c) {
        const int r0 = r_begin + c * SK;
        return c < nchunks && r0 + SK <= r_end && r0 + bshift >= 0 && r0 + SK - 1 + bshift < R;
    };
    uint32_t offa[2], offb[4];
#pragma unroll
    for (int e = 0; e < 2; ++e) offa[e] = (uint32_t)((lrow + 16 * e) * lda + ca) << 2;
#pragma unroll
    for (int e = 0; e < 4; ++e) offb[e] = (uint32_t)((rb + 8 * e) * ldb + cb) << 2;
    auto issue = [&](int c) {
        uint32_t va[2], vb[4];
        if (inside(c)) {
            const int r0 = r_begin + c * SK;
            const uint32_t sa = (uint32_t)(r0 * lda) << 2, sb = (uint32_t)((r0 + bshift) * ldb) << 2;
#pragma unroll
            for (int e = 0; e < 2; ++e) va[e] = sa + offa[e];
#pragma unroll
            for (int e = 0; e < 4; ++e) vb[e] = sb + offb[e];
        } else {
            const int r0 = r_begin + (c < nchunks ? c : nchunks - 1) * SK;
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int rc = min(r0 + lrow + 16 * e, r_end - 1);
                va[e] = (__umul24((uint32_t)rc, (uint32_t)lda) + (uint32_t)ca) << 2;
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int rc = min(r0 + rb + 8 * e, r_end - 1);
                const int rbc = min(max(rc + bshift, 0), R - 1);
                vb[e] = (__umul24((uint32_t)rbc, (uint32_t)ldb) + (uint32_t)cb) << 2;
            }
        }
#pragma unroll
        for (int e = 0; e < 2; ++e) na[e] = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(A) + va[e]);
#pragma unroll
        for (int e = 0; e < 4; ++e) nb[e] = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(B) + vb[e]);
    };
    float4 ta[2], tb[4];
    auto stage_body = [&](auto edge_tag, int c) {
        constexpr bool EDGE = decltype(edge_tag)::value;
        const int r0 = r_begin + c * SK;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            float4 xa = ta[e];
            if (EDGE) {
                const bool aok = r0 + lrow + 16 * e < r_end;
                xa = make_float4(aok ? xa.x : 0.f, aok ? xa.y : 0.f, aok ? xa.z : 0.f, aok ? xa.w : 0.f);
            }
            if (colpart) { cs.x += xa.x; cs.y += xa.y; cs.z += xa.z; cs.w += xa.w; }
            u32x2 a1, a2, a3;
            cut4(xa, himask, a1, a2, a3);
            const uint32_t d = wra + e * 16 * ROWB;
            *LDS_AS(u32x2, (uintptr_t)d) = a1;
            *LDS_AS(u32x2, (uintptr_t)(d + PLANE_B)) = a2;
            *LDS_AS(u32x2, (uintptr_t)(d + 2 * PLANE_B)) = a3;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float4 xb = tb[e];
            if (EDGE) {
                const int r = r0 + rb + 8 * e;
                const bool bok = r < r_end && r + bshift >= 0 && r + bshift < R;
                xb = make_float4(bok ? xb.x : 0.f, bok ? xb.y : 0.f, bok ? xb.z : 0.f, bok ? xb.w : 0.f);
            }
            u32x2 b1, b2, b3;
            cut4(xb, himask, b1, b2, b3);
            const uint32_t d = wrb + e * 8 * ROWB_W;
            *LDS_AS(u32x2, (uintptr_t)d) = b1;
            *LDS_AS(u32x2, (uintptr_t)(d + PLANE_W)) = b2;
            *LDS_AS(u32x2, (uintptr_t)(d + 2 * PLANE_W)) = b3;
        }
    };
    auto stage = [&](int c) {
#pragma unroll
        for (int e = 0; e < 2; ++e) ta[e] = na[e];
#pragma unroll
        for (int e = 0; e < 4; ++e) tb[e] = nb[e];
        __builtin_amdgcn_sched_barrier(0);
        issue(c + 1);                       // one whole period (cut, barrier, products, barrier) ahead of its use
        __builtin_amdgcn_sched_barrier(0);
        if (inside(c)) stage_body(std::false_type{}, c);
        else stage_body(std::true_type{}, c);
    };
    // NL: column tiles of this wave that hold output columns (N = 200: 7 and 6; the last block of N = 512: 4 and 0)
    const int ncol = N - (n0 + MMDFN_TNS_TN * wc);
    const int nlive = ncol <= 0 ? 0 : (ncol >= MMDFN_TNS_TN ? NCT : (ncol + 15) / 16);
    auto mma = [&](auto ntm_tag, auto nl_tag) {
        constexpr int NTM = decltype(ntm_tag)::value;
        constexpr int NL = decltype(nl_tag)::value;
        u32x4 af[NTM][3], bf[3][NL];
#pragma unroll
        for (int q = 0; q < 3; ++q)
#pragma unroll
            for (int i = 0; i < NTM; ++i) af[i][q] = tr_frag(rda + q * PLANE_B + 32 * i);
#pragma unroll
        for (int q = 0; q < 3; ++q)
#pragma unroll
            for (int j = 0; j < NL; ++j) bf[q][j] = tr_frag_w(rdb + q * PLANE_W + 32 * j);
#pragma unroll
        for (int pc = 0; pc < 6; ++pc) {
            const int ai = (pc == 0) ? 2 : (pc == 1 || pc == 3) ? 1 : 0;
            const int bi = (pc < 3) ? 0 : (pc < 5) ? 1 : 2;
#pragma unroll
            for (int j = 0; j < NL; ++j)
#pragma unroll
                for (int i = 0; i < NTM; ++i) acc[i][j] = mfma16(af[i][ai], bf[bi][j], acc[i][j]);
        }
    };
    auto mma_nl = [&](auto ntm_tag) {
        switch (nlive) {
            case 7: mma(ntm_tag, std::integral_constant<int, 7>{}); break;
            case 6: mma(ntm_tag, std::integral_constant<int, 6>{}); break;
            case 5: mma(ntm_tag, std::integral_constant<int, 5>{}); break;
            case 4: mma(ntm_tag, std::integral_constant<int, 4>{}); break;
            case 3: mma(ntm_tag, std::integral_constant<int, 3>{}); break;
            case 2: mma(ntm_tag, std::integral_constant<int, 2>{}); break;
            case 1: mma(ntm_tag, std::integral_constant<int, 1>{}); break;
            default: break;
        }
    };

    issue(0);
#pragma unroll 1
    for (int k = 0; k < nchunks; ++k) {
        stage(k);
        __syncthreads();
        if (ntm == 2) mma_nl(std::integral_constant<int, 2>{});
        else if (ntm == 1) mma_nl(std::integral_constant<int, 1>{});
        __syncthreads();
    }
    if (ntm > 0) {
        float* P = sq.part[p] + (int64_t)split * M * N;
        const int fi = lane & 15, g = lane >> 4;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < NCT; ++j) {
                const int n = n0 + MMDFN_TNS_TN * wc + 16 * j + fi;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = m0 + 32 * wr + 16 * i + 4 * g + r;
                    if (m < M && n < N) P[(int64_t)m * N + n] = acc[i][j][r];
                }
            }
    }
    if (colpart) {
        // the 16 threads of a column quad (16 staged row residues) hold partial sums of the same 4 columns
        float* red = reinterpret_cast<float*>(smem);                   // (the planes: every read is behind the loop's last barrier)
        *reinterpret_cast<float4*>(red + (tid >> 5) * TSM + 4 * quad) = cs;
        __syncthreads();
        if (tid < TSM && m0 + tid < M) {
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < 16; ++k) s += red[k * TSM + tid];
            colpart[(int64_t)split * M + m0 + tid] = s;
        }
    }
}

// ABL (tuning build, timing only): 1 no cut (pieces = raw bits), 2 no MFMAs, 4 no fragment reads, 8 no global loads, 16 cycle
// stamps of waves 0 and 4 into `trace` (32 floats per workgroup)
// One workgroup (512 threads, LDS_B bytes of LDS at tns_smem) of the launch: block `bid` of the segment table `sq` (TnSplitSegs, or
// the shorter rider table of the GRU backward launch, gru.hip).
template <int ABL, class SQ>
__device__ __forceinline__ void tns_block(const SQ& sq, const int bid, unsigned char* tns_smem, float* __restrict__ trace) {
    int p = 0;
    while (p + 1 < sq.n && bid >= sq.wg_prefix[p + 1]) ++p;
    const int local = bid - sq.wg_prefix[p];
    if (sq.wide[p]) {
        if (ABL == 0) tns_wide_body(sq, p, local, tns_smem);
        return;
    }
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

    const int ca = (m0 + 4 * quad < M) ? m0 + 4 * quad : 0;          // columns past M / N re-fetch valid ones: they only reach
    const int cb = (quad < TSN / 4 && n0 + 4 * quad < N) ? n0 + 4 * quad : 0;   // accumulator rows / columns that are never stored
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
    float4 na[4], nb[4];                                             // the group's next chunk, on its way

    // chunk c fully inside the split and inside B's shifted row range: no row needs masking, no address clamping
    auto inside = [&](int c) {
        const int r0 = r_begin + c * SK;
        return c < nchunks && r0 + SK <= r_end && r0 + bshift >= 0 && r0 + SK - 1 + bshift < R;
    };
    // byte offsets of the thread's four slots from the first row of a chunk (A) / of its shifted row (B): an interior chunk adds
    // its scalar row offset (one VALU per load); a chunk at an edge computes per-thread clamped offsets instead.  The loads
    // themselves have ONE site (loads defined on two paths of the loop would meet in phi copies, and a copy of a register with a
    // load in flight waits for it).
    uint32_t offa[4], offb[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        offa[e] = (uint32_t)((lrow + 8 * e) * lda + ca) << 2;
        offb[e] = (uint32_t)((lrow + 8 * e) * ldb + cb) << 2;
    }
    auto issue = [&](int c) {
        if (ABL & 8) return;
        uint32_t va[4], vb[4];
        if (inside(c)) {
            const int r0 = r_begin + c * SK;
            const uint32_t sa = (uint32_t)(r0 * lda) << 2, sb = (uint32_t)((r0 + bshift) * ldb) << 2;
#pragma unroll
            for (int e = 0; e < 4; ++e) { va[e] = sa + offa[e]; vb[e] = sb + offb[e]; }
        } else {
            const int r0 = r_begin + (c < nchunks ? c : nchunks - 1) * SK + lrow;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int rc = min(r0 + 8 * e, r_end - 1);
                const int rbc = min(max(rc + bshift, 0), R - 1);
                va[e] = (__umul24((uint32_t)rc, (uint32_t)lda) + (uint32_t)ca) << 2;
                vb[e] = (__umul24((uint32_t)rbc, (uint32_t)ldb) + (uint32_t)cb) << 2;
            }
        }
        // (offsets from the operand bases on both paths: rows x stride x 4 < 2^32 is checked by the launcher)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            na[e] = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(A) + va[e]);
            nb[e] = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(B) + vb[e]);
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
    // cut the landed chunk c and write its planes (EDGE: rows outside the split / outside B's range count as zeros)
    float4 ta[4], tb[4];                                             // the chunk being staged
    auto stage_body = [&](auto edge_tag, int c) {
        constexpr bool EDGE = decltype(edge_tag)::value;
        const int r0 = r_begin + c * SK;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float4 xa = ta[e], xb = tb[e];
            if (ABL & 8) { xa = make_float4(1.f + c, 2.f, 3.f, 4.f); xb = xa; }
            if (EDGE) {
                const int r = r0 + lrow + 8 * e;
                const bool aok = r < r_end;
                const bool bok = aok && r + bshift >= 0 && r + bshift < R;
                xa = make_float4(aok ? xa.x : 0.f, aok ? xa.y : 0.f, aok ? xa.z : 0.f, aok ? xa.w : 0.f);
                xb = make_float4(bok ? xb.x : 0.f, bok ? xb.y : 0.f, bok ? xb.z : 0.f, bok ? xb.w : 0.f);
            }
            if (colpart) { cs.x += xa.x; cs.y += xa.y; cs.z += xa.z; cs.w += xa.w; }
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
    // The landed chunk moves to the staging registers, the group's next chunk, c + 2, is requested at once (two steps ahead of
    // its use), then the cut.  (Requesting it behind the cut instead saves the 16 register moves and measured 25 % slower.)
    auto stage = [&](int c) {
        const long long t0 = now();
#pragma unroll
        for (int e = 0; e < 4; ++e) { ta[e] = na[e]; tb[e] = nb[e]; }
        if (ABL & 16) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        const long long t1 = now();
        issue(c + 2);
        __builtin_amdgcn_sched_barrier(0);
        const long long t2 = now();
        if (inside(c)) stage_body(std::false_type{}, c);
        else stage_body(std::true_type{}, c);
        tm[0] += t1 - t0; tm[1] += t2 - t1; tm[2] += now() - t2;
    };
    // the chunk's K = 32 step: products a3 b1, a2 b1, a1 b1 | a2 b2, a1 b2 | a1 b3 (smallest first), NTM row tiles of the wave
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
        float* o = trace + (int64_t)bid * 32 + (tid ? 16 : 0);
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
        // partial tile -> slab [split][M][N]; C/D layout of a 16 x 16 tile: column = lane & 15, row = 4 (lane >> 4) + r
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
}
}  // namespace tnsb
