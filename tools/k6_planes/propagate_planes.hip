// K6 on producer-cut operands: out = A_hat . H with the H side arriving as three bf16 piece planes.
//
// Replaces torch.spmm(adj, input) (reference model_GCN.py:178) for the launches propagate_split.hip serves (dialogues
// of >= 128 utterances), when the kernel that produced H also wrote its three exact bf16 pieces (h = h1 + h2 + h3, 8 + 8 + 8
// significant bits, cut by truncation) as ROW-MAJOR planes
//     HP[piece][row][dp]   (bf16, dp = round_up(d, 8), columns d..dp-1 and rows >= M N are zeros)
// -- mmdfn_cut_planes below, or the epilogue of the producing kernel.  Same arithmetic as propagate_split.hip (six
// piece products of weight >= 2^-16 on v_mfma_f32_32x32x16_bf16, fp32-level error), but the H side costs the MFMA
// waves no VALU and no registers any more:
//   * B operand: each 32-row chunk of the three planes goes HBM -> LDS by LDS-DMA (global_load_lds_dwordx4, 6 per wave
//     and chunk, no VGPR landing zone) into a 3-stage ring; the image is row-major [row][128 columns] with the 16-byte
//     units of row r XOR-ed by 4 (r & 3) -- applied on the SOURCE address, the DMA destination is lane-linear -- and the
//     MFMA B fragments (8 consecutive k of one column) come out of it with ds_read_b64_tr_b16 (the hardware transpose
//     read), conflict-free: the 32 lanes served per LDS cycle read 4 rows x 64 bytes that the XOR spreads over all banks.
//   * A operand (the tile strip): HBM -> registers in MFMA layout as before (k = 16 kh + 8 kg + e, two 16-byte loads per
//     K=16 step), cut in registers, 24 cutting stages per chunk pinned behind every other MFMA.
//   * Every vector-memory operation of the main loop is issued from inline asm and retired with ONE hand-counted
//     s_waitcnt vmcnt(10) per chunk (hipcc would drain the queue in front of every LDS read that follows an LDS-DMA);
//     one raw s_barrier per chunk, placed mid-chunk as in the shared pipeline.
// Cross-modal diagonals are added in the LDS row epilogue from the fp32 rows of H.  XCD mapping: blockIdx % 8 ==
// dialogue % 8.
#include "../../mm_dfn_amd/csrc/mmdfn_internal.h"
#include "k6_planes.h"
#include <stdlib.h>

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define LDS_AS(T, p) ((__attribute__((address_space(3))) T*)(p))

constexpr int PBK = 32;                 // k per chunk = two K=16 MFMA steps
constexpr int NSTG = 3;                 // LDS ring depth (chunks)
constexpr int PLANE_B = PBK * 256;      // one piece plane of a stage: 32 rows x 128 columns bf16 = 8192 bytes
constexpr int STAGE_B = 3 * PLANE_B;    // 24576 bytes

__device__ __forceinline__ float as_f(uint32_t u) { return __builtin_bit_cast(float, u); }
__device__ __forceinline__ uint32_t as_u(float f) { return __builtin_bit_cast(uint32_t, f); }

__device__ __forceinline__ f32x16 mfma_bf16(u32x4 a, u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// six LDS-DMA pieces (1 KiB each, consecutive in LDS) from one scalar base: lane l of piece t fetches 16 bytes at
// sbase + voff[t] and they land at lds_dst + 1024 t + 16 l.  M0 (the DMA destination) is saved and restored: the
// compiler owns it outside this statement.  The leading s_nop covers a scalar base written just before the statement.
__device__ __forceinline__ void dma6(uint32_t lds_dst, const void* sbase, uint32_t v0, uint32_t v1, uint32_t v2,
                                     uint32_t v3, uint32_t v4, uint32_t v5) {
    uint32_t keep;
    asm volatile(
        "s_nop 4\n\t"
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %1\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %3, %2\n\t"
        "s_add_u32 m0, m0, 0x400\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %4, %2\n\t"
        "s_add_u32 m0, m0, 0x400\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %5, %2\n\t"
        "s_add_u32 m0, m0, 0x400\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %6, %2\n\t"
        "s_add_u32 m0, m0, 0x400\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %7, %2\n\t"
        "s_add_u32 m0, m0, 0x400\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %8, %2\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "s"(lds_dst), "s"(sbase), "v"(v0), "v"(v1), "v"(v2), "v"(v3), "v"(v4), "v"(v5)
        : "memory", "scc");
}

// the tile-strip loads of one chunk: four 16-byte loads per lane, hidden from hipcc's s_waitcnt bookkeeping (the
// destinations are read only behind wait_loads()).  Fast form: one per-lane offset + scalar chunk base.
__device__ __forceinline__ void aload_fast(f32x4& a0, f32x4& a1, f32x4& a2, f32x4& a3, uint32_t voff, const void* sbase) {
    asm volatile(
        "s_nop 4\n\t"
        "global_load_dwordx4 %0, %4, %5\n\t"
        "global_load_dwordx4 %1, %4, %5 offset:16\n\t"
        "global_load_dwordx4 %2, %4, %5 offset:64\n\t"
        "global_load_dwordx4 %3, %4, %5 offset:80"
        : "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(a3)
        : "v"(voff), "s"(sbase)
        : "memory");
}
__device__ __forceinline__ void aload_fast_nt(f32x4& a0, f32x4& a1, f32x4& a2, f32x4& a3, uint32_t voff, const void* sbase) {
    asm volatile(
        "s_nop 4\n\t"
        "global_load_dwordx4 %0, %4, %5 nt\n\t"
        "global_load_dwordx4 %1, %4, %5 offset:16 nt\n\t"
        "global_load_dwordx4 %2, %4, %5 offset:64 nt\n\t"
        "global_load_dwordx4 %3, %4, %5 offset:80 nt"
        : "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(a3)
        : "v"(voff), "s"(sbase)
        : "memory");
}
// clamped form (the ragged last chunk): four per-lane offsets from the tile base
__device__ __forceinline__ void aload_safe(f32x4& a0, f32x4& a1, f32x4& a2, f32x4& a3, uint32_t o0, uint32_t o1,
                                           uint32_t o2, uint32_t o3, const void* sbase) {
    asm volatile(
        "s_nop 4\n\t"
        "global_load_dwordx4 %0, %4, %8\n\t"
        "global_load_dwordx4 %1, %5, %8\n\t"
        "global_load_dwordx4 %2, %6, %8\n\t"
        "global_load_dwordx4 %3, %7, %8"
        : "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(a3)
        : "v"(o0), "v"(o1), "v"(o2), "v"(o3), "s"(sbase)
        : "memory");
}
// everything but the newest 10 vector-memory operations of this wave has completed; the registers named become
// readable here (no consumer of them can be scheduled above the statement)
__device__ __forceinline__ void wait_loads(f32x4& a0, f32x4& a1, f32x4& a2, f32x4& a3) {
    asm volatile("s_waitcnt vmcnt(10)" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : : "memory");
}
// timing experiment only (wrong results): the same register tie without the wait
__device__ __forceinline__ void nowait_loads(f32x4& a0, f32x4& a1, f32x4& a2, f32x4& a3) {
    asm volatile("; no wait" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : : "memory");
}

// ABL (profiling aid, tuning build only): 1 no cross-modal terms, 2 no MFMAs, 4 no cutting stages, 8 no tile-strip loads,
// 16 no LDS-DMA, 32 no fragment reads, 64 no epilogue at all, 128 s_memtime stamps of the phases into `trace`
template <int ABL, int EPI>
__global__ __launch_bounds__(256, 2) void propagate_planes_kernel(
    const float* __restrict__ tiles, const float* __restrict__ cross, const float* __restrict__ H,
    const uint16_t* __restrict__ HP, float* __restrict__ out, const int32_t* __restrict__ dia_len,
    const int32_t* __restrict__ row_start, const int64_t* __restrict__ tile_base, int B, int M, int N, int d, int dp,
    int64_t plane_elems, int ldh, int ldo, int max_rb, unsigned long long* __restrict__ trace) {
    constexpr int NCT = 4;                 // 32-column MFMA tiles
    constexpr int WROWS = 32;              // tile rows per wave
    constexpr int BM = 4 * WROWS;          // 128 tile rows per workgroup
    constexpr int CB = 32 * NCT;
    constexpr int LDO = CB + 8;            // epilogue row stride (floats)
    constexpr int OROWS = 64;              // output rows staged per epilogue pass
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];

    // XCD-aware decode: bid % 8 == dialogue % 8
    const int Rd = M * max_rb;
    const int bid = blockIdx.x;
    const int yq = bid >> 3;
    const int i = (yq / Rd) * 8 + (bid & 7);
    if (i >= B) return;
    const int rho = yq % Rd;
    const int m = rho / max_rb;
    const int rb = rho - m * max_rb;
    const int L = dia_len[i];
    const int r0 = rb * BM;
    if (r0 >= L) return;
    const int ld = (L + 3) & ~3;
    const int rs = row_start[i];
    const float* T = tiles + tile_base[i] + (int64_t)m * L * ld;
    const int64_t R0 = (int64_t)m * N + rs;            // flat row of this tile's k = 0

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
#define PL_STAMP(K)                                                                          \
    do {                                                                                     \
        if ((ABL & 128) && tid == 0) trace[(int64_t)blockIdx.x * 8 + (K)] = __builtin_readcyclecounter(); \
    } while (0)
    PL_STAMP(0);
    if (ABL & 768) {     // experiment: delay the second workgroup of every CU (blocks 256..511) by ~16k (256) / ~32k (512) cycles
        if (blockIdx.x >= 256 && blockIdx.x < 512)
            for (int z = 0; z < ((ABL & 512) ? 8 : 4); ++z) __builtin_amdgcn_s_sleep(64);
    }
    if ((ABL & 128) && tid == 0) {
        unsigned hwid;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        trace[(int64_t)blockIdx.x * 8 + 7] = ((unsigned long long)xcc << 32) | hwid;
    }
    const int l32 = lane & 31;
    const int kg = lane >> 5;
    const int wrow0 = r0 + WROWS * w;

    f32x16 acc[NCT];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[ct][r] = 0.f;

    // ---- A side: lane (row l32, kg) of MFMA step kh holds A[row][K0 + 16 kh + 8 kg + e], e = 0..7
    const int arow = wrow0 + l32;
    const int arowc = arow < L ? arow : L - 1;
    const uint32_t a_voff = (uint32_t)((arowc * ld + 8 * kg) * 4);     // bytes from the tile base (tile < 2^31 bytes)
    uint32_t a_coal[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int rr = wrow0 + 8 * q + (lane >> 3);
        a_coal[q] = (uint32_t)(((rr < L ? rr : L - 1) * ld + 4 * (lane & 7)) * 4);
    }
    const int nchunks = (L + PBK - 1) / PBK;
    const int klast = (nchunks - 1) * PBK;
    const int nfull = L / PBK;             // chunks whose 32 k values all lie inside the tile
    const int limA = L - 8 * kg;           // strip column K0 + 16 kh + e' is data iff K0 + 16 kh + e' < limA

    // ---- B side: the DMA pieces of this wave.  Piece q = 6 w + t of a chunk: plane q >> 3, rows 4 (q & 7) .. + 3;
    // lane l -> row 4 (q & 7) + (l >> 4), LDS unit (l & 15) holding source unit (l & 15) ^ 4 (l >> 4)
    const int ndu = dp >> 3;               // 16-byte units per plane row that exist
    uint32_t bvoff[6];
#pragma unroll
    for (int t = 0; t < 6; ++t) {
        const int q = 6 * w + t;
        const int row = 4 * (q & 7) + (lane >> 4);
        int cu = (lane & 15) ^ (4 * (lane >> 4));
        cu = cu < ndu ? cu : ndu - 1;      // columns >= dp: re-read the last unit (their accumulator columns are never stored)
        bvoff[t] = (uint32_t)(((int64_t)(q >> 3) * plane_elems + (int64_t)row * dp) * 2 + cu * 16);
    }
    const uint32_t lds0 = (uint32_t)(uintptr_t)LDS_AS(void, smem);
    const uint32_t dma_dst = lds0 + 6 * 1024 * w;       // + stage offset
    const char* hp_row0 = reinterpret_cast<const char*>(HP) + R0 * dp * 2;   // plane 0, row k = 0 of this tile

    // fragment reads: lane (t = lane & 15, g = lane >> 4) of (column tile ct, step kh, half r) supplies the address of
    // row 16 kh + 8 (g >> 1) + 4 r + (t >> 2), columns 32 ct + 16 (g & 1) + 4 (t & 3) .. + 3
    uint32_t trb[NCT];
    {
        const int t = lane & 15, g = lane >> 4;
        const int rho4 = (t >> 2) & 3;
        const uint32_t rowpart = (uint32_t)((8 * (g >> 1) + (t >> 2)) * 256 + (2 * (g & 1) + ((t & 3) >> 1)) * 16 + 8 * (t & 1));
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) trb[ct] = rowpart + 64 * (ct ^ rho4);
    }

    f32x4 araw[2][4];
    u32x4 ap1[2][2], ap2[2][2], ap3[2][2];     // [set][kh]
    u32x4 bf_[NCT][3];
    if (ABL & (4 | 8 | 32)) {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) { ap1[a][b] = u32x4{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u}; ap2[a][b] = ap1[a][b]; ap3[a][b] = ap1[a][b]; }
#pragma unroll
        for (int a = 0; a < NCT; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) bf_[a][b] = u32x4{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) araw[a][b] = f32x4{1.f, 1.f, 1.f, 1.f};
    }
    float cx0 = 0.f, cx1 = 0.f;
    uint32_t himask;
    asm volatile("s_mov_b32 %0, 0xffff0000" : "=s"(himask));

    // ---- issue the loads of the chunk at k = K0: 6 DMA pieces into LDS stage STG + 4 tile-strip loads into set SET
#define PL_ISSUE(SET, K0, STG, SAFE)                                                                       \
    do {                                                                                                   \
        if (!(ABL & 16))                                                                                   \
            dma6(dma_dst + (STG) * STAGE_B, hp_row0 + (int64_t)(K0) * dp * 2, bvoff[0], bvoff[1], bvoff[2], \
                 bvoff[3], bvoff[4], bvoff[5]);                                                            \
        if (ABL & 8) {                                                                                     \
        } else if (SAFE) {                                                                                        \
            uint32_t o_[4];                                                                                \
            _Pragma("unroll") for (int f = 0; f < 4; ++f) {                                                \
                const int ka_ = (K0) + 16 * (f >> 1) + 8 * kg + 4 * (f & 1);                               \
                o_[f] = (uint32_t)((arowc * ld + (ka_ < ld ? ka_ : ld - 4)) * 4);                          \
            }                                                                                              \
            aload_safe(araw[SET][0], araw[SET][1], araw[SET][2], araw[SET][3], o_[0], o_[1], o_[2], o_[3], T); \
        } else if (ABL & 2048) {   /* timing experiment (wrong data placement): fully coalesced 8 rows x 128 B per load */ \
            aload_safe(araw[SET][0], araw[SET][1], araw[SET][2], araw[SET][3], a_coal[0], a_coal[1], a_coal[2], \
                       a_coal[3], T + (K0));                                                               \
        } else if (ABL & 8192) {                                                                           \
            aload_fast_nt(araw[SET][0], araw[SET][1], araw[SET][2], araw[SET][3], a_voff, T + (K0));       \
        } else {                                                                                           \
            aload_fast(araw[SET][0], araw[SET][1], araw[SET][2], araw[SET][3], a_voff, T + (K0));          \
        }                                                                                                  \
    } while (0)

    // ---- cutting stage T (0..23) of the chunk at k = K0 held raw in set SET: unit u = T / 3 (kh = u >> 2, pair p = u & 3:
    // elements e = 2p, 2p+1 of the step), stage st = T % 3 (first / second / third piece)
#define PL_STAGE(SET, K0, T, SAFE)                                                                         \
    do {                                                                                                   \
        const int u_ = (T) / 3, st_ = (T) % 3, p_ = u_ & 3, h_ = u_ >> 2;                                  \
        const int kp_ = (K0) + 16 * h_ + 2 * p_;                 /* + 8 kg (folded into limA) */           \
        if (st_ == 0) {                                                                                    \
            const f32x4 v_ = araw[SET][2 * h_ + (p_ >> 1)];                                               \
            cx0 = (!(SAFE) || kp_ < limA) ? ((p_ & 1) ? v_.z : v_.x) : 0.f;                                \
            cx1 = (!(SAFE) || kp_ + 1 < limA) ? ((p_ & 1) ? v_.w : v_.y) : 0.f;                            \
        }                                                                                                  \
        const uint32_t w_ = __builtin_amdgcn_perm(as_u(cx1), as_u(cx0), 0x07060302u);                      \
        if (st_ == 0) ap1[SET][h_][p_] = w_; else if (st_ == 1) ap2[SET][h_][p_] = w_; else ap3[SET][h_][p_] = w_; \
        if (st_ < 2) {                                                                                     \
            cx0 = cx0 - as_f(as_u(cx0) & himask);                                                          \
            cx1 = cx1 - as_f(as_u(cx1) & himask);                                                          \
        }                                                                                                  \
    } while (0)

    // ---- B fragments of (stage byte offset SOFF, step KH) for one piece: two transpose reads per column tile
#define PL_LOADB(PIECE, SOFF, KH)                                                                          \
    do {                                                                                                   \
        _Pragma("unroll") for (int ct_ = 0; ct_ < NCT; ++ct_) {                                            \
            const uint32_t a_ = lds0 + (SOFF) + trb[ct_] + (PIECE) * PLANE_B + (KH) * 4096;                \
            const s16x4 lo_ = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_AS(s16x4, (uintptr_t)a_));                  \
            const s16x4 hi_ = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_AS(s16x4, (uintptr_t)(a_ + 1024)));           \
            const u32x2 l2_ = __builtin_bit_cast(u32x2, lo_), h2_ = __builtin_bit_cast(u32x2, hi_);        \
            bf_[ct_][PIECE] = u32x4{l2_.x, l2_.y, h2_.x, h2_.y};                                           \
        }                                                                                                  \
    } while (0)

    // ---- one K=16 step: 6 piece products x 4 column tiles, accumulators round-robin; a cutting stage of the NEXT chunk
    // behind every other MFMA; fragments of (NSOFF, NKH) reloaded piece by piece as the pieces retire
#define PL_STEP(P, KH, NSOFF, NKH, K1, SAFE)                                                               \
    do {                                                                                                   \
        _Pragma("unroll") for (int pc_ = 0; pc_ < 6; ++pc_) {                                              \
            const u32x4 av_ = (pc_ == 0) ? ap3[P][KH] : (pc_ == 1 || pc_ == 3) ? ap2[P][KH] : ap1[P][KH];  \
            const int bi_ = (pc_ < 3) ? 0 : (pc_ < 5) ? 1 : 2;                                             \
            _Pragma("unroll") for (int ct_ = 0; ct_ < NCT; ++ct_) {                                        \
                if (!(ABL & 2)) acc[ct_] = mfma_bf16(av_, bf_[ct_][bi_], acc[ct_]);                        \
                if (!(ABL & 4) && (ct_ & 1) == 0) PL_STAGE((P) ^ 1, K1, 12 * (KH) + 2 * pc_ + (ct_ >> 1), SAFE);         \
                __builtin_amdgcn_sched_barrier(0);                                                         \
            }                                                                                              \
            if (!(ABL & 32)) {                                                                             \
                if (pc_ == 2) PL_LOADB(0, NSOFF, NKH);                                                     \
                if (pc_ == 4) PL_LOADB(1, NSOFF, NKH);                                                     \
                if (pc_ == 5) PL_LOADB(2, NSOFF, NKH);                                                     \
            }                                                                                              \
            __builtin_amdgcn_sched_barrier(0);                                                             \
        }                                                                                                  \
    } while (0)

    // ---- period C (parity P): loads of chunk C+2 -> raw set P / stage (C+2) % 3; MFMAs of chunk C (piece set P, stage
    // C % 3) with the cutting of chunk C+1 (raw set P^1 -> piece set P^1) behind them; ONE barrier, mid-chunk: by then every
    // wave has waited for its DMA pieces of chunk C+1 (issued one period ago), so after it stage (C+1) % 3 may be read (the
    // fragments of chunk C+1 step 0 are fetched during step 1); and every wave has issued its last read of stage
    // (C-1) % 3 long ago, which the DMA of chunk C+2 (issued at the top of this period, i.e. behind the barrier of
    // period C-1) overwrites.
#define PL_BODY(P, C, SAFE)                                                                                \
    do {                                                                                                   \
        const int kn1_ = ((C) + 1) * PBK < klast ? ((C) + 1) * PBK : klast;                                \
        const int kn2_ = ((C) + 2) * PBK < klast ? ((C) + 2) * PBK : klast;                                \
        PL_ISSUE(P, kn2_, s2, SAFE);                                                                       \
        if (ABL & 1024) nowait_loads(araw[(P) ^ 1][0], araw[(P) ^ 1][1], araw[(P) ^ 1][2], araw[(P) ^ 1][3]); \
        else wait_loads(araw[(P) ^ 1][0], araw[(P) ^ 1][1], araw[(P) ^ 1][2], araw[(P) ^ 1][3]);           \
        __builtin_amdgcn_sched_barrier(0);                                                                 \
        PL_STEP(P, 0, s0 * STAGE_B, 1, kn1_, SAFE);                                                        \
        asm volatile("" ::: "memory");                                                                     \
        if (!(ABL & 4096)) __builtin_amdgcn_s_barrier();   /* 4096: timing experiment, racy */             \
        asm volatile("" ::: "memory");                                                                     \
        PL_STEP(P, 1, s1 * STAGE_B, 0, kn1_, SAFE);                                                        \
        { const int s_ = s0; s0 = s1; s1 = s2; s2 = s_; }                                                  \
    } while (0)

    PL_STAMP(1);
    int s0 = 0, s1 = 1, s2 = 2;            // LDS stages of chunks C, C+1, C+2 (scalar)
    {   // prologue: chunk 0 -> raw set 0 / stage 0 (cut here into piece set 0), chunk 1 -> raw set 1 / stage 1
        PL_ISSUE(0, 0, 0, 1);
        PL_ISSUE(1, (PBK < klast ? PBK : klast), 1, 1);
        wait_loads(araw[0][0], araw[0][1], araw[0][2], araw[0][3]);     // chunk 0 has landed (the 10 newest are chunk 1's)
#pragma unroll
        for (int t = 0; t < 24; ++t) PL_STAGE(0, 0, t, 1);
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        PL_LOADB(0, 0, 0);
        PL_LOADB(1, 0, 0);
        PL_LOADB(2, 0, 0);
    }
    PL_STAMP(2);
    {
        // Period C loads chunk min(C+2, last) and cuts chunk min(C+1, last); only the PARTIAL last chunk (L % 32 != 0)
        // needs clamps and masks: the clamp/mask-free body runs for C + 2 < nfull, and for every C when L % 32 == 0.
        const int nsafe0 = (nfull == nchunks) ? nchunks : (nfull > 2 ? nfull - 2 : 0);
        int c = 0;
        for (; c + 1 < nsafe0; c += 2) {
            PL_BODY(0, c, 0);
            PL_BODY(1, c + 1, 0);
        }
        for (; c + 1 < nchunks; c += 2) {
            PL_BODY(0, c, 1);
            PL_BODY(1, c + 1, 1);
        }
        if (nchunks & 1) PL_BODY(0, c, 1);
    }
#undef PL_BODY
#undef PL_STEP
#undef PL_LOADB
#undef PL_STAGE
#undef PL_ISSUE

    // ---- epilogue through LDS, OROWS rows per pass.  Every DMA piece must have landed before the region is reused
    // (the last two periods issued re-reads of the last chunk).
    PL_STAMP(3);
    // the raw sets still have loads in flight (the last periods re-read the last chunk): naming them here keeps their
    // registers out of the compiler's hands until the data has landed
    asm volatile("s_waitcnt vmcnt(0)"
                 : "+v"(araw[0][0]), "+v"(araw[0][1]), "+v"(araw[0][2]), "+v"(araw[0][3]), "+v"(araw[1][0]),
                   "+v"(araw[1][1]), "+v"(araw[1][2]), "+v"(araw[1][3])
                 :
                 : "memory");
    float* Os = reinterpret_cast<float*>(smem);
    if (ABL & 64) {          // keep the accumulators alive without the epilogue's traffic
        float sacc = 0.f;
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
            for (int r = 0; r < 16; ++r) sacc += acc[ct][r];
        if (sacc == 1.2345f) out[tid] = sacc;
        return;
    }
    if (EPI == 1) {
        // ---- direct epilogue: every lane finishes the 64 outputs it holds (C/D layout of a 32x32 tile: column
        // 32 ct + l32, rows wrow0 + (r & 3) + 8 (r >> 2) + 4 kg).  No LDS round trip, no barrier, and the M-1 cross-modal
        // rows are plain 4-byte loads (two rows x 128 contiguous bytes per wave instruction) that are all independent:
        // the whole epilogue is one memory round trip deep instead of one per pair of modalities and pass.
        PL_STAMP(4);
        // column tile ct exists entirely (32 ct + 31 < d), partly (lanes l32 < d - 32 ct) or not at all: the loads of the
        // partial tile read a clamped column (no branch around a load), its stores are masked per lane
        int coff[NCT];
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) coff[ct] = (32 * ct + l32 < d) ? 32 * ct + l32 : d - 1;
        const int nct = (d + 31) >> 5;                 // column tiles that hold data (uniform)
        const bool allrows = wrow0 + WROWS <= L;       // uniform: no row of this wave lies past the tile
        uint32_t roff[16];        // (row_start + row) of the 16 rows, clamped into the dialogue
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = wrow0 + (r & 3) + 8 * (r >> 2) + 4 * kg;
            roff[r] = (uint32_t)(rs + (row < L ? row : L - 1));
        }
        if (!(ABL & 1)) {
            for (int q = 0; q < M - 1; ++q) {
                const int n = q + (q >= m ? 1 : 0);
                const int pk = (m < n) ? mmdfn_pair_index(m, n, M) : mmdfn_pair_index(n, m, M);
                const float* cwp = cross + (int64_t)pk * N;
                const float* hn = H + (int64_t)n * N * ldh;
                float cwv[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) cwv[r] = cwp[roff[r]];
                float hv[NCT][16];
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct)
                    if (ct < nct) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) hv[ct][r] = hn[(int64_t)roff[r] * ldh + coff[ct]];
                    }
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct)
                    if (ct < nct) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[ct][r] = fmaf(cwv[r], hv[ct][r], acc[ct][r]);
                    }
            }
        }
        float* om = out + (int64_t)m * N * ldo;
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) {
            if (32 * ct + l32 >= d) continue;
            if (allrows) {
#pragma unroll
                for (int r = 0; r < 16; ++r) om[(int64_t)roff[r] * ldo + 32 * ct + l32] = acc[ct][r];
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (wrow0 + (r & 3) + 8 * (r >> 2) + 4 * kg < L) om[(int64_t)roff[r] * ldo + 32 * ct + l32] = acc[ct][r];
            }
        }
        if (ABL & 128) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            PL_STAMP(6);
        }
        return;
    }
    if (EPI == 2) {
        // ---- row epilogue: 32 lanes per output row (lane j owns float4 j of the row, j < d / 4), 8 rows per step of
        // the 256 threads; a wave instruction of the cross-modal loads reads two whole H rows (2 x 4 d bytes, <= 7 cache
        // lines) instead of 64-byte pieces of 16 different rows.
        __syncthreads();
        const int cw4 = d / 4;
        const int j4 = tid & 31;                  // float4 index inside the row
        const int rsub = tid >> 5;                // 0..7
        const bool cok = j4 < cw4;
        const int jc = cok ? j4 : cw4 - 1;
#pragma unroll
        for (int pass = 0; pass < BM / OROWS; ++pass) {
            if (pass) __syncthreads();
            if ((w >> 1) == pass) {
                const int lrow0 = WROWS * (w & 1) + 4 * kg;
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        Os[(lrow0 + (r & 3) + 8 * (r >> 2)) * LDO + 32 * ct + l32] = acc[ct][r];
            }
            __syncthreads();
            constexpr int NR = OROWS / 8;         // rows per thread and pass
            float4 v[NR];
            int64_t grow[NR];
            bool rok[NR];
#pragma unroll
            for (int t = 0; t < NR; ++t) {
                const int lrow = rsub + 8 * t;
                const int row = r0 + pass * OROWS + lrow;
                rok[t] = row < L;
                grow[t] = rs + (rok[t] ? row : L - 1);
                v[t] = *reinterpret_cast<const float4*>(&Os[lrow * LDO + 4 * jc]);
            }
            if (!(ABL & 1)) {
                for (int q = 0; q < M - 1; ++q) {
                    const int n = q + (q >= m ? 1 : 0);
                    const int pk = (m < n) ? mmdfn_pair_index(m, n, M) : mmdfn_pair_index(n, m, M);
                    const float* cwp = cross + (int64_t)pk * N;
                    const float* hn = H + (int64_t)n * N * ldh + 4 * jc;
                    float cwv[NR];
                    float4 h[NR];
#pragma unroll
                    for (int t = 0; t < NR; ++t) cwv[t] = cwp[grow[t]];
#pragma unroll
                    for (int t = 0; t < NR; ++t) h[t] = *reinterpret_cast<const float4*>(hn + grow[t] * ldh);
#pragma unroll
                    for (int t = 0; t < NR; ++t) {
                        v[t].x = fmaf(cwv[t], h[t].x, v[t].x);
                        v[t].y = fmaf(cwv[t], h[t].y, v[t].y);
                        v[t].z = fmaf(cwv[t], h[t].z, v[t].z);
                        v[t].w = fmaf(cwv[t], h[t].w, v[t].w);
                    }
                }
            }
            float* om = out + (int64_t)m * N * ldo + 4 * j4;
#pragma unroll
            for (int t = 0; t < NR; ++t)
                if (cok && rok[t]) *reinterpret_cast<float4*>(om + grow[t] * ldo) = v[t];
        }
        if (ABL & 128) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            PL_STAMP(6);
        }
        return;
    }
    __syncthreads();
    constexpr int NJ = CB / 16;
    const int cw4 = d / 4;
    const int erow = tid >> 2;
    const int eq = tid & 3;
#pragma unroll
    for (int pass = 0; pass < BM / OROWS; ++pass) {
        if (pass) __syncthreads();
        PL_STAMP(4 + pass);
        if ((w >> 1) == pass) {
            const int lrow0 = WROWS * (w & 1) + 4 * kg;
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    Os[(lrow0 + (r & 3) + 8 * (r >> 2)) * LDO + 32 * ct + l32] = acc[ct][r];
        }
        __syncthreads();
        const int row = r0 + pass * OROWS + erow;
        if (row < L) {
            const int64_t grow = rs + row;
            float4 v[NJ];
            int coff[NJ];
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int c4 = eq + 4 * j;
                coff[j] = 4 * (c4 < cw4 ? c4 : cw4 - 1);
                v[j] = *reinterpret_cast<const float4*>(&Os[erow * LDO + coff[j]]);
            }
#pragma unroll 2
            for (int q = 0; q < ((ABL & 1) ? 0 : M - 1); ++q) {
                const int n = q + (q >= m ? 1 : 0);
                const int pk = (m < n) ? mmdfn_pair_index(m, n, M) : mmdfn_pair_index(n, m, M);
                const float cwt = cross[(int64_t)pk * N + grow];
                const float* hrow = H + ((int64_t)n * N + grow) * ldh;
                float4 h[NJ];
#pragma unroll
                for (int j = 0; j < NJ; ++j) h[j] = *reinterpret_cast<const float4*>(hrow + coff[j]);
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    v[j].x = fmaf(cwt, h[j].x, v[j].x);
                    v[j].y = fmaf(cwt, h[j].y, v[j].y);
                    v[j].z = fmaf(cwt, h[j].z, v[j].z);
                    v[j].w = fmaf(cwt, h[j].w, v[j].w);
                }
            }
            float* orow = out + ((int64_t)m * N + grow) * ldo;
#pragma unroll
            for (int j = 0; j < NJ; ++j)
                if (eq + 4 * j < cw4) *reinterpret_cast<float4*>(orow + coff[j]) = v[j];
        }
    }
    if (ABL & 128) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        PL_STAMP(6);
    }
#undef PL_STAMP
}

// H (R rows of d fp32, row stride ldx) -> three bf16 piece planes [piece][rows_pad][dp]; rows >= R and columns >= d are
// written as zeros.  One thread per (row, 4 columns).
__global__ __launch_bounds__(256) void cut_planes_kernel(const float* __restrict__ X, uint16_t* __restrict__ P, int64_t R,
                                                         int64_t rows_pad, int d, int dp, int ldx) {
    const int q4 = dp >> 2;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= rows_pad * q4) return;
    const int64_t row = idx / q4;
    const int c = (int)(idx - row * q4) * 4;
    float x[4] = {0.f, 0.f, 0.f, 0.f};
    if (row < R && c < d) {                       // d % 4 == 0: the group is entirely inside or outside
        const float4 v = *reinterpret_cast<const float4*>(X + row * ldx + c);
        x[0] = v.x; x[1] = v.y; x[2] = v.z; x[3] = v.w;
    }
    uint32_t pc[3][2];
#pragma unroll
    for (int s = 0; s < 3; ++s) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            pc[s][h] = __builtin_amdgcn_perm(as_u(x[2 * h + 1]), as_u(x[2 * h]), 0x07060302u);
            if (s < 2) {
                x[2 * h] = x[2 * h] - as_f(as_u(x[2 * h]) & 0xffff0000u);
                x[2 * h + 1] = x[2 * h + 1] - as_f(as_u(x[2 * h + 1]) & 0xffff0000u);
            }
        }
        *reinterpret_cast<uint2*>(P + ((int64_t)s * rows_pad + row) * dp + c) = make_uint2(pc[s][0], pc[s][1]);
    }
}

}  // namespace

// rows of one piece plane for R feature rows: a whole number of 32-row chunks plus one chunk of zeros behind the last row
static inline int64_t mmdfn_planes_rows(int64_t R) { return ((R + 31) / 32) * 32 + 32; }

extern "C" int mmdfn_cut_planes(const float* X, void* planes, int64_t R, int d, int ldx, void* stream) {
    if (R <= 0 || d <= 0 || (d & 3) || ldx < d || (ldx & 3)) return -1;
    const int dp = (d + 7) & ~7;
    const int64_t rows_pad = mmdfn_planes_rows(R);
    const int64_t n = rows_pad * (dp >> 2);
    hipLaunchKernelGGL(cut_planes_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, X,
                       reinterpret_cast<uint16_t*>(planes), R, rows_pad, d, dp, ldx);
    MMDFN_CHECK_LAUNCH();
    return 0;
}

// -2: shape not covered by the plane kernel (caller uses mmdfn_propagate)
extern "C" int mmdfn_propagate_planes(const float* tiles, const float* cross, const float* H, const void* planes,
                                      float* out, const int32_t* dia_len, const int32_t* row_start,
                                      const int64_t* tile_base, int B, int M, int N, int d, int ldh, int ldo, int max_len,
                                      void* stream) {
    if (B <= 0 || M <= 0 || M > 9 || N <= 0 || d <= 0 || (d & 3) || max_len <= 0) return -1;
    if (ldh < d || ldo < d || (ldh & 3) || (ldo & 3)) return -1;
    if (d > 128) return -2;
    const int dp = (d + 7) & ~7;
    const int64_t rows_pad = mmdfn_planes_rows((int64_t)M * N);
    const int64_t plane_elems = rows_pad * dp;
    if (3 * plane_elems * 2 >= (int64_t)1 << 31) return -2;      // 32-bit DMA offsets
#if 1
    {
        const char* ve = getenv("MMDFN_PLANES_VER");
        const int ver = ve ? atoi(ve) : 1;
        if (ver >= 2) {
            const char* ne = getenv("MMDFN_PLANES_NT");
            const char* ae2 = getenv("MMDFN_PLANES_ABL");
            const char* ke = getenv("MMDFN_PLANES_KM");
            return mmdfn_launch_propagate_planes2(tiles, cross, H, planes, out, dia_len, row_start, tile_base, B, M, N, d, ldh,
                                                  ldo, max_len, ver == 3, ne ? atoi(ne) : 0, ae2 ? atoi(ae2) : 0, ke ? atoi(ke) : 0,
                                                  (hipStream_t)stream);
        }
    }
#endif
    const int max_rb = (max_len + 127) / 128;
    const int lds_bytes = NSTG * STAGE_B;           // 73728 B (>= the 64 x 136 float epilogue staging)
    dim3 grid(((B + 7) / 8) * 8 * M * max_rb);
#define PL_LAUNCH2(A, E)                                                                                            \
    do {                                                                                                         \
        if (mmdfn_allow_big_lds(propagate_planes_kernel<A, E>) != 0) return -1;                                  \
        hipLaunchKernelGGL((propagate_planes_kernel<A, E>), grid, dim3(256), lds_bytes, (hipStream_t)stream, tiles, \
                           cross, H, reinterpret_cast<const uint16_t*>(planes), out, dia_len, row_start,         \
                           tile_base, B, M, N, d, dp, plane_elems, ldh, ldo, max_rb, trace_ptr);                 \
    } while (0)
    unsigned long long* trace_ptr = nullptr;
#if 1
    const char* ee = getenv("MMDFN_PLANES_EPI");
    const int epi = ee ? atoi(ee) : 0;
#define PL_LAUNCH(A)                     \
    do {                                 \
        if (epi == 1) PL_LAUNCH2(A, 1);  \
        else if (epi == 2) PL_LAUNCH2(A, 2); \
        else PL_LAUNCH2(A, 0);           \
    } while (0)
    const char* te = getenv("MMDFN_TRACE_PTR");
    if (te) trace_ptr = reinterpret_cast<unsigned long long*>(strtoull(te, nullptr, 0));
    const char* ae = getenv("MMDFN_PLANES_ABL");
    switch (ae ? atoi(ae) : 0) {
        case 1: PL_LAUNCH(1); break;
        case 2: PL_LAUNCH(2); break;
        case 4: PL_LAUNCH(4); break;
        case 12: PL_LAUNCH(12); break;
        case 16: PL_LAUNCH(16); break;
        case 48: PL_LAUNCH(48); break;
        case 60: PL_LAUNCH(60); break;
        case 64: PL_LAUNCH(64); break;
        case 66: PL_LAUNCH(66); break;
        case 76: PL_LAUNCH(76); break;
        case 112: PL_LAUNCH(112); break;
        case 124: PL_LAUNCH(124); break;
        case 126: PL_LAUNCH(126); break;
        case 4096: PL_LAUNCH(4096); break;
        case 4160: PL_LAUNCH(4160); break;
        case 6208: PL_LAUNCH(6208); break;
        case 8192: PL_LAUNCH(8192); break;
        case 8256: PL_LAUNCH(8256); break;
        case 2048: PL_LAUNCH(2048); break;
        case 2112: PL_LAUNCH(2112); break;
        case 1024: PL_LAUNCH(1024); break;
        case 1088: PL_LAUNCH(1088); break;
        case 256: PL_LAUNCH(256); break;
        case 512: PL_LAUNCH(512); break;
        case 768: PL_LAUNCH(768); break;
        case 128: if (trace_ptr) { PL_LAUNCH(128); } else { PL_LAUNCH(0); } break;
        default: PL_LAUNCH(0); break;
    }
#undef PL_LAUNCH
#else
    PL_LAUNCH2(0, 0);
#endif
#undef PL_LAUNCH2
    MMDFN_CHECK_LAUNCH();
    return 0;
}
