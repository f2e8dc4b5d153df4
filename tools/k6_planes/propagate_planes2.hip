// K6 on producer-cut operands, second structure: both operands enter the compute units in whole cache lines.
//
// What bounds K6 at BASELINE cfg5 (rocprofv3 counters, profiles/r03_k6_memory_path.md): the sum over all vector-memory
// requests of their latency divided by the ~64 requests a CU's L1 keeps in flight equals the kernel time to 2 % --
// neither the matrix pipe (33 % busy) nor instruction issue.  So this version spends instructions to save requests:
//   * A (the tile strip, the only HBM stream that matters): each wave loads its 32 rows x 32 k in FULL 128-byte row
//     segments (lane = 8 row + unit: 8 rows x 128 B per instruction, every line requested exactly once), parks them in a
//     wave-private 4 KB LDS square (units XOR-swizzled by (row >> 1) & 7: conflict-free 16-byte writes and reads) and
//     reads them back in MFMA layout (lane = row, 8 consecutive k) to be cut into bf16 pieces in registers.
//     Fragment-shaped loads straight to registers (propagate_planes.hip) touch 32 lines per instruction, 16 bytes each.
//   * B (bf16 piece planes of H): LDS-DMA into a 2-stage ring, transpose reads, as in propagate_planes.hip.
//   * NW = 4 waves x 32 rows (128-row blocks, two workgroups per CU) or NW = 8 (256-row blocks: every H row is fetched
//     half as often).
// Pipeline, period C (parity P): top: the strip chunk C+1 (loaded during period C-1) -> LDS square -> fragments; loads
// of chunk C+2 issued; MFMA step 0 of chunk C with the cutting of chunk C+1 behind the MFMAs; mid: DMA pieces of chunk
// C+1 have landed (vmcnt), barrier, DMA of chunk C+2 into the stage chunk C just left; MFMA step 1.
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

constexpr int PBK = 32;
constexpr int PLANE_B = PBK * 256;      // 8192
constexpr int STAGE_B = 3 * PLANE_B;    // 24576
constexpr int ASQ_B = 32 * 128;         // one wave's strip square: 32 rows x 32 fp32

__device__ __forceinline__ float as_f(uint32_t u) { return __builtin_bit_cast(float, u); }
__device__ __forceinline__ uint32_t as_u(float f) { return __builtin_bit_cast(uint32_t, f); }
__device__ __forceinline__ f32x16 mfma_bf16(u32x4 a, u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// NP LDS-DMA pieces (1 KiB each, consecutive in LDS) from one scalar base; M0 saved and restored
template <int NP>
__device__ __forceinline__ void dma_pieces(uint32_t lds_dst, const void* sbase, const uint32_t (&v)[6]) {
    uint32_t keep;
    if (NP == 6)
        asm volatile(
            "s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\t"
            "global_load_lds_dwordx4 %3, %2\n\ts_add_u32 m0, m0, 0x400\n\ts_nop 0\n\t"
            "global_load_lds_dwordx4 %4, %2\n\ts_add_u32 m0, m0, 0x400\n\ts_nop 0\n\t"
            "global_load_lds_dwordx4 %5, %2\n\ts_add_u32 m0, m0, 0x400\n\ts_nop 0\n\t"
            "global_load_lds_dwordx4 %6, %2\n\ts_add_u32 m0, m0, 0x400\n\ts_nop 0\n\t"
            "global_load_lds_dwordx4 %7, %2\n\ts_add_u32 m0, m0, 0x400\n\ts_nop 0\n\t"
            "global_load_lds_dwordx4 %8, %2\n\ts_mov_b32 m0, %0"
            : "=&s"(keep)
            : "s"(lds_dst), "s"(sbase), "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5])
            : "memory", "scc");
    else
        asm volatile(
            "s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\t"
            "global_load_lds_dwordx4 %3, %2\n\ts_add_u32 m0, m0, 0x400\n\ts_nop 0\n\t"
            "global_load_lds_dwordx4 %4, %2\n\ts_add_u32 m0, m0, 0x400\n\ts_nop 0\n\t"
            "global_load_lds_dwordx4 %5, %2\n\ts_mov_b32 m0, %0"
            : "=&s"(keep)
            : "s"(lds_dst), "s"(sbase), "v"(v[0]), "v"(v[1]), "v"(v[2])
            : "memory", "scc");
}

// the four strip loads of a chunk (8 rows x 128 bytes each), hidden from hipcc's s_waitcnt bookkeeping
template <int NT>
__device__ __forceinline__ void aload4(f32x4& a0, f32x4& a1, f32x4& a2, f32x4& a3, uint32_t o0, uint32_t o1, uint32_t o2,
                                       uint32_t o3, const void* sbase) {
    if (NT)
        asm volatile(
            "s_nop 4\n\t"
            "global_load_dwordx4 %0, %4, %8 nt\n\tglobal_load_dwordx4 %1, %5, %8 nt\n\t"
            "global_load_dwordx4 %2, %6, %8 nt\n\tglobal_load_dwordx4 %3, %7, %8 nt"
            : "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(a3)
            : "v"(o0), "v"(o1), "v"(o2), "v"(o3), "s"(sbase)
            : "memory");
    else
        asm volatile(
            "s_nop 4\n\t"
            "global_load_dwordx4 %0, %4, %8\n\tglobal_load_dwordx4 %1, %5, %8\n\t"
            "global_load_dwordx4 %2, %6, %8\n\tglobal_load_dwordx4 %3, %7, %8"
            : "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(a3)
            : "v"(o0), "v"(o1), "v"(o2), "v"(o3), "s"(sbase)
            : "memory");
}

#define VM_WAIT(N, T)                                                                                      \
    asm volatile("s_waitcnt vmcnt(" #N ")" : "+v"(T[0]), "+v"(T[1]), "+v"(T[2]), "+v"(T[3]) : : "memory")

// ABL (tuning build): 1 no cross-modal terms, 64 no epilogue.  KM = 1: the strip is read K-MAJOR through the symmetry of
// the tile (A[r][k] = A[k][r]): chunk rows k, 128 contiguous bytes = this wave's 32 output rows; the four waves of a
// workgroup then read 512 contiguous bytes of every tile row of the chunk (one DRAM page visit per row and chunk instead of
// four rows' worth of 128-byte visits), and the fragments are column reads of the LDS square.
template <int NW, int NT, int ABL, int KM>
__global__ __launch_bounds__(64 * NW, 2) void propagate_planes2_kernel(
    const float* __restrict__ tiles, const float* __restrict__ cross, const float* __restrict__ H,
    const uint16_t* __restrict__ HP, float* __restrict__ out, const int32_t* __restrict__ dia_len,
    const int32_t* __restrict__ row_start, const int64_t* __restrict__ tile_base, int B, int M, int N, int d, int dp,
    int64_t plane_elems, int ldh, int ldo, int max_rb) {
    constexpr int NCT = 4;
    constexpr int WROWS = 32;
    constexpr int BM = NW * WROWS;
    constexpr int CB = 32 * NCT;
    constexpr int LDO = CB + 8;
    constexpr int OROWS = BM / 2;          // output rows staged per epilogue pass (two passes)
    constexpr int NP = 24 / NW;            // DMA pieces per wave and chunk
    constexpr int NTHR = 64 * NW;
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];

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
    const int64_t R0 = (int64_t)m * N + rs;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l32 = lane & 31;
    const int kg = lane >> 5;
    const int wrow0 = r0 + WROWS * w;

    f32x16 acc[NCT];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[ct][r] = 0.f;

    const int nchunks = (L + PBK - 1) / PBK;
    const int klast = (nchunks - 1) * PBK;
    const int nfull = L / PBK;
    const int limA = L - 8 * kg;

    // ---- LDS map: [2 B stages][NW strip squares]; the epilogue staging overlays it
    const uint32_t lds0 = (uint32_t)(uintptr_t)LDS_AS(void, smem);
    const uint32_t asq = lds0 + 2 * STAGE_B + ASQ_B * w;

    // ---- A side.  Load q (0..3): lane -> strip row 8 q + (lane >> 3), 16-byte unit (lane & 7) of the chunk's 128 bytes
    const int lr = lane >> 3, lu = lane & 7;
    uint32_t a_row[4];                      // byte offset of (row, unit) from the tile base, chunk offset added per chunk
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        if (KM) {       // tile row = chunk row 8 q + lr (chunk base added per chunk), columns wrow0 + 4 lu .. + 3
            const int cc = wrow0 + 4 * lu;
            a_row[q] = (uint32_t)(((8 * q + lr) * ld + (cc < ld ? cc : ld - 4)) * 4);
        } else {
            const int rr = wrow0 + 8 * q + lr;
            a_row[q] = (uint32_t)(((rr < L ? rr : L - 1) * ld + 4 * lu) * 4);
        }
    }
    // square writes: row 8 q + lr, unit lu ^ swz(row), swz(row) = (row >> 1) & 7  (8 q + lr >> 1 & 7 = 4 q + (lr >> 1) & 7)
    uint32_t a_wr[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) a_wr[q] = asq + (8 * q + lr) * 128 + ((KM ? lu : (lu ^ ((4 * q + (lr >> 1)) & 7))) << 4);
    // square reads: lane (row l32, kg), fragment f = 2 kh + f' -> unit 4 kh + 2 kg + f'
    uint32_t a_rd[4];
#pragma unroll
    for (int f = 0; f < 4; ++f) a_rd[f] = asq + l32 * 128 + (((4 * (f >> 1) + 2 * kg + (f & 1)) ^ ((l32 >> 1) & 7)) << 4);

    const uint32_t a_col = asq + 8 * kg * 128 + 4 * l32;       // KM: column l32 of the square, rows 8 kg ..
    // ---- B side: DMA pieces q = NP w + t: plane q >> 3, rows 4 (q & 7) + (lane >> 4), unit (lane & 15) ^ 4 (lane >> 4)
    const int ndu = dp >> 3;
    uint32_t bvoff[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int t = 0; t < NP; ++t) {
        const int q = NP * w + t;
        const int row = 4 * (q & 7) + (lane >> 4);
        int cu = (lane & 15) ^ (4 * (lane >> 4));
        cu = cu < ndu ? cu : ndu - 1;
        bvoff[t] = (uint32_t)(((int64_t)(q >> 3) * plane_elems + (int64_t)row * dp) * 2 + cu * 16);
    }
    const uint32_t dma_dst = lds0 + NP * 1024 * w;
    const char* hp_row0 = reinterpret_cast<const char*>(HP) + R0 * dp * 2;

    uint32_t trb[NCT];
    {
        const int t = lane & 15, g = lane >> 4;
        const int rho4 = (t >> 2) & 3;
        const uint32_t rowpart = (uint32_t)((8 * (g >> 1) + (t >> 2)) * 256 + (2 * (g & 1) + ((t & 3) >> 1)) * 16 + 8 * (t & 1));
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) trb[ct] = lds0 + rowpart + 64 * (ct ^ rho4);
    }

    f32x4 tr_[4];                           // transit registers of the strip loads
    f32x4 af_[4];                           // the chunk being cut, MFMA layout: af_[2 kh + f'] = A[row][16 kh + 8 kg + 4 f' ..]
    u32x4 ap1[2][2], ap2[2][2], ap3[2][2];
    u32x4 bf_[NCT][3];
    float cx0 = 0.f, cx1 = 0.f;
    uint32_t himask;
    asm volatile("s_mov_b32 %0, 0xffff0000" : "=s"(himask));

#define P2_ALOAD(K0, SAFE)                                                                                 \
    do {                                                                                                   \
        if (KM) {                                                                                          \
            if (SAFE) {   /* chunk rows past the tile: re-read the last row (masked when cut) */            \
                uint32_t o_[4];                                                                            \
                _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                            \
                    const int kr_ = (K0) + 8 * q + lr;                                                     \
                    o_[q] = a_row[q] + (uint32_t)(((kr_ < L ? kr_ : L - 1) - (8 * q + lr)) * ld * 4);      \
                }                                                                                          \
                aload4<NT>(tr_[0], tr_[1], tr_[2], tr_[3], o_[0], o_[1], o_[2], o_[3], T);                 \
            } else {                                                                                       \
                aload4<NT>(tr_[0], tr_[1], tr_[2], tr_[3], a_row[0], a_row[1], a_row[2], a_row[3],         \
                           T + (int64_t)(K0) * ld);                                                        \
            }                                                                                              \
        } else if (SAFE) {                                                                                        \
            const int kk_ = (K0) + 4 * lu;                                                                 \
            const uint32_t adj_ = (uint32_t)((kk_ < ld ? kk_ : ld - 4) * 4 - 16 * lu);                     \
            aload4<NT>(tr_[0], tr_[1], tr_[2], tr_[3], a_row[0] + adj_, a_row[1] + adj_, a_row[2] + adj_,  \
                       a_row[3] + adj_, T);                                                                \
        } else {                                                                                           \
            aload4<NT>(tr_[0], tr_[1], tr_[2], tr_[3], a_row[0], a_row[1], a_row[2], a_row[3], T + (K0));  \
        }                                                                                                  \
    } while (0)

    // transit registers -> LDS square -> fragments (same wave, LDS executes a wave's operations in order)
#define P2_SQUARE()                                                                                        \
    do {                                                                                                   \
        _Pragma("unroll") for (int q = 0; q < 4; ++q)                                                      \
            *LDS_AS(f32x4, (uintptr_t)a_wr[q]) = tr_[q];                                                   \
        if (KM) {   /* column reads: af_[2 kh + f'][j] = S[16 kh + 8 kg + 4 f' + j][l32] */                   \
            _Pragma("unroll") for (int f = 0; f < 4; ++f)                                                  \
                _Pragma("unroll") for (int j = 0; j < 4; ++j)                                              \
                    af_[f][j] = *LDS_AS(float, (uintptr_t)(a_col + (16 * (f >> 1) + 4 * (f & 1) + j) * 128)); \
        } else {                                                                                           \
            _Pragma("unroll") for (int f = 0; f < 4; ++f)                                                  \
                af_[f] = *LDS_AS(f32x4, (uintptr_t)a_rd[f]);                                               \
        }                                                                                                  \
    } while (0)

#define P2_STAGE(SET, K0, T_, SAFE)                                                                        \
    do {                                                                                                   \
        const int u_ = (T_) / 3, st_ = (T_) % 3, p_ = u_ & 3, h_ = u_ >> 2;                                \
        const int kp_ = (K0) + 16 * h_ + 2 * p_;                                                           \
        if (st_ == 0) {                                                                                    \
            const f32x4 v_ = af_[2 * h_ + (p_ >> 1)];                                                      \
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

// the two transpose reads of one (column tile, piece) fragment
#define P2_LOADB1(CT, PIECE, SOFF, KH)                                                                     \
    do {                                                                                                   \
        const uint32_t a_ = (SOFF) + trb[CT] + (PIECE) * PLANE_B + (KH) * 4096;                            \
        const s16x4 lo_ = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_AS(s16x4, (uintptr_t)a_));           \
        const s16x4 hi_ = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_AS(s16x4, (uintptr_t)(a_ + 1024)));  \
        const u32x2 l2_ = __builtin_bit_cast(u32x2, lo_), h2_ = __builtin_bit_cast(u32x2, hi_);            \
        bf_[CT][PIECE] = u32x4{l2_.x, l2_.y, h2_.x, h2_.y};                                                \
    } while (0)
#define P2_LOADB(PIECE, SOFF, KH)                                                                          \
    do {                                                                                                   \
        _Pragma("unroll") for (int ct_ = 0; ct_ < NCT; ++ct_) P2_LOADB1(ct_, PIECE, SOFF, KH);             \
    } while (0)

    // One K=16 step (SOFF, KH) of chunk parity P: 6 piece products x 4 column tiles, accumulators round-robin.  Product
    // order a3b1 a2b1 a1b1 | a2b2 a1b2 | a1b3.  Behind the MFMAs, at most ~6 instructions per gap (a gap of 8 or more
    // costs a whole extra MFMA slot on gfx950: tools/ubench/mfma_bf16_fillers.hip):
    //   even column tiles: one cutting stage of the next chunk (~5 VALU);
    //   odd column tiles : the fragment reloads, two (column tile, piece) pairs = 4 transpose reads, as pieces retire:
    //     during product 3 (a2b2): b1 of the NEXT step (b1 retired by product 2) <- (NSOFF, NKH)
    //     during product 5 (a1b3): b2 of the NEXT step (b2 retired by product 4)
    //   and b3 of the next step in one burst behind the last MFMA (b3 is in use until then; the burst must precede the
    //   mid-chunk barrier, after which the DMA of chunk C+2 overwrites the stage these reads come from).
#define P2_STEP(P, SOFF, KH, NSOFF, NKH, K1, SAFE)                                                         \
    do {                                                                                                   \
        _Pragma("unroll") for (int pc_ = 0; pc_ < 6; ++pc_) {                                              \
            const u32x4 av_ = (pc_ == 0) ? ap3[P][KH] : (pc_ == 1 || pc_ == 3) ? ap2[P][KH] : ap1[P][KH];  \
            const int bi_ = (pc_ < 3) ? 0 : (pc_ < 5) ? 1 : 2;                                             \
            _Pragma("unroll") for (int ct_ = 0; ct_ < NCT; ++ct_) {                                        \
                acc[ct_] = mfma_bf16(av_, bf_[ct_][bi_], acc[ct_]);                                        \
                if ((ct_ & 1) == 0) {                                                                      \
                    P2_STAGE((P) ^ 1, K1, 12 * (KH) + 2 * pc_ + (ct_ >> 1), SAFE);                         \
                } else {                                                                                   \
                    if (pc_ == 3) { P2_LOADB1(ct_ - 1, 0, NSOFF, NKH); P2_LOADB1(ct_, 0, NSOFF, NKH); }    \
                    if (pc_ == 5) { P2_LOADB1(ct_ - 1, 1, NSOFF, NKH); P2_LOADB1(ct_, 1, NSOFF, NKH); }    \
                }                                                                                          \
                __builtin_amdgcn_sched_barrier(0);                                                         \
            }                                                                                              \
        }                                                                                                  \
        P2_LOADB(2, NSOFF, NKH);                                                                           \
        __builtin_amdgcn_sched_barrier(0);                                                                 \
    } while (0)

    // vector-memory operations in flight, oldest first, at the top of period C: DMA(C+1) [NP], strip(C+1) [4] ... no:
    // issue order is strip(C+1) [top of C-1], DMA(C+1) [mid C-1], strip(C+2) [top of C], DMA(C+2) [mid C]:
    //   top of C : strip(C+1) must have landed, DMA(C+1) may be in flight      -> vmcnt(NP)
    //   mid of C : DMA(C+1) must have landed, strip(C+2) may be in flight     -> vmcnt(4)
#define P2_BODY(P, C, SAFE)                                                                                \
    do {                                                                                                   \
        const int kn1_ = ((C) + 1) * PBK < klast ? ((C) + 1) * PBK : klast;                                \
        const int kn2_ = ((C) + 2) * PBK < klast ? ((C) + 2) * PBK : klast;                                \
        if (NP == 6) VM_WAIT(6, tr_); else VM_WAIT(3, tr_);                                                \
        P2_SQUARE();                                                                                       \
        P2_ALOAD(kn2_, SAFE);                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                                 \
        P2_STEP(P, (P) * STAGE_B, 0, (P) * STAGE_B, 1, kn1_, SAFE);                                        \
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");                                                   \
        __builtin_amdgcn_s_barrier();                                                                      \
        asm volatile("" ::: "memory");                                                                     \
        dma_pieces<NP>(dma_dst + (P) * STAGE_B, hp_row0 + (int64_t)kn2_ * dp * 2, bvoff);                  \
        P2_STEP(P, (P) * STAGE_B, 1, ((P) ^ 1) * STAGE_B, 0, kn1_, SAFE);                                  \
    } while (0)

    {   // prologue: strip(0), DMA(0) -> stage 0, strip(1)... chunk 0 is cut here, chunk 1 enters the pipeline state
        P2_ALOAD(0, 1);
        dma_pieces<NP>(dma_dst, hp_row0, bvoff);
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(tr_[0]), "+v"(tr_[1]), "+v"(tr_[2]), "+v"(tr_[3]) : : "memory");
        P2_SQUARE();
        const int k1 = PBK < klast ? PBK : klast;
        P2_ALOAD(k1, 1);                                       // strip(1): waited for at the top of period 0
#pragma unroll
        for (int t = 0; t < 24; ++t) P2_STAGE(0, 0, t, 1);     // pieces of chunk 0 -> set 0
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_barrier();                          // DMA(0) of every wave has landed (vmcnt(0) above)
        asm volatile("" ::: "memory");
        dma_pieces<NP>(dma_dst + STAGE_B, hp_row0 + (int64_t)k1 * dp * 2, bvoff);    // DMA(1) -> stage 1
        P2_LOADB(0, 0, 0);
        P2_LOADB(1, 0, 0);
        P2_LOADB(2, 0, 0);
    }
    {
        const int nsafe0 = (nfull == nchunks) ? nchunks : (nfull > 2 ? nfull - 2 : 0);
        int c = 0;
        for (; c + 1 < nsafe0; c += 2) {
            P2_BODY(0, c, 0);
            P2_BODY(1, c + 1, 0);
        }
        for (; c + 1 < nchunks; c += 2) {
            P2_BODY(0, c, 1);
            P2_BODY(1, c + 1, 1);
        }
        if (nchunks & 1) P2_BODY(0, c, 1);
    }
#undef P2_BODY
#undef P2_STEP
#undef P2_LOADB
#undef P2_STAGE
#undef P2_SQUARE
#undef P2_ALOAD

    // ---- epilogue through LDS, two passes of OROWS rows (the transit registers still have loads in flight)
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(tr_[0]), "+v"(tr_[1]), "+v"(tr_[2]), "+v"(tr_[3]) : : "memory");
    float* Os = reinterpret_cast<float*>(smem);
    if (ABL & 64) {
        float sacc = 0.f;
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
            for (int r = 0; r < 16; ++r) sacc += acc[ct][r];
        if (sacc == 1.2345f) out[tid] = sacc;
        return;
    }
    __syncthreads();
    constexpr int NJ = CB / 16;
    const int cw4 = d / 4;
    const int erow = tid >> 2;
    const int eq = tid & 3;
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        if (pass) __syncthreads();
        if (w / (NW / 2) == pass) {
            const int lrow0 = WROWS * (w % (NW / 2)) + 4 * kg;
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
}

}  // namespace

// variant: 0 = 4 waves, 1 = 8 waves; nt: non-temporal strip loads; abl: tuning only
int mmdfn_launch_propagate_planes2(const float* tiles, const float* cross, const float* H, const void* planes, float* out,
                                   const int32_t* dia_len, const int32_t* row_start, const int64_t* tile_base, int B, int M,
                                   int N, int d, int ldh, int ldo, int max_len, int variant, int nt, int abl, int km, hipStream_t s) {
    const int dp = (d + 7) & ~7;
    const int64_t rows_pad = (((int64_t)M * N + 31) / 32) * 32 + 32;
    const int64_t plane_elems = rows_pad * dp;
    const int nw = variant ? 8 : 4;
    const int max_rb = (max_len + 32 * nw - 1) / (32 * nw);
    int lds_bytes = 2 * STAGE_B + nw * ASQ_B;
    const int epi_bytes = 16 * nw * (128 + 8) * 4;
    if (lds_bytes < epi_bytes) lds_bytes = epi_bytes;
    dim3 grid(((B + 7) / 8) * 8 * M * max_rb);
#define P2_LAUNCH3(NW_, NT_, A_, K_)                                                                               \
    do {                                                                                                          \
        if (mmdfn_allow_big_lds(propagate_planes2_kernel<NW_, NT_, A_, K_>) != 0) return -1;                          \
        hipLaunchKernelGGL((propagate_planes2_kernel<NW_, NT_, A_, K_>), grid, dim3(64 * NW_), lds_bytes, s, tiles,   \
                           cross, H, reinterpret_cast<const uint16_t*>(planes), out, dia_len, row_start,          \
                           tile_base, B, M, N, d, dp, plane_elems, ldh, ldo, max_rb);                             \
    } while (0)
#define P2_LAUNCH(NW_, NT_, A_) do { if (km) P2_LAUNCH3(NW_, NT_, A_, 1); else P2_LAUNCH3(NW_, NT_, A_, 0); } while (0)
#if 1
    if (abl == 1) { if (variant) { if (nt) P2_LAUNCH(8, 1, 1); else P2_LAUNCH(8, 0, 1); } else { if (nt) P2_LAUNCH(4, 1, 1); else P2_LAUNCH(4, 0, 1); } }
    else if (abl == 64) { if (variant) { if (nt) P2_LAUNCH(8, 1, 64); else P2_LAUNCH(8, 0, 64); } else { if (nt) P2_LAUNCH(4, 1, 64); else P2_LAUNCH(4, 0, 64); } }
    else
#endif
    {
        (void)abl;
        if (variant) { if (nt) P2_LAUNCH(8, 1, 0); else P2_LAUNCH(8, 0, 0); }
        else { if (nt) P2_LAUNCH(4, 1, 0); else P2_LAUNCH(4, 0, 0); }
    }
#undef P2_LAUNCH
#undef P2_LAUNCH3
    MMDFN_CHECK_LAUNCH();
    return 0;
}
