// K6 (large-dialogue variant, producer / consumer form): out = A_hat . H with the fp32 product carried by bf16 MFMAs.
//
// Replaces torch.spmm(adj, input) (reference model_GCN.py:178) for launches whose dialogues are long enough that the
// exact-f32 matrix rate would bound the product (BASELINE cfg5: L = 512, M = 6).  Same arithmetic as propagate_split.hip:
// every fp32 operand is cut exactly (truncation) into three bf16 pieces  x = x1 + x2 + x3  and the product is the six
// piece products of weight >= 2^-16, each exact in the fp32 accumulator (fp32-level error, checked against fp64).
//
// What is different is WHO does what.  In propagate_split.hip every wave loads, cuts, multiplies and finishes its rows,
// and on gfx950 the phases of such a wave add up (profiles/r03_k6_memory_path.md: MFMA 38 us + tile side 20 + H side 10 +
// epilogue 22).  Here a workgroup is PERSISTENT (one per compute unit, it walks its share of the (dialogue, modality,
// 128-row block) items) and its eight waves have two roles, one wave of each kind per SIMD:
//   * four CONSUMER waves issue nothing but LDS fragment reads and v_mfma_f32_32x32x16_bf16 (32 tile rows x 128 feature
//     columns each; at the end of an item they park the accumulators in an LDS row buffer and go on with the next item);
//   * four PRODUCER waves do everything else: two of them own the tile strip (whole 128-byte lines per tile row), two
//     the H rows (whole rows per load instruction); chunk c+3 is requested from HBM / L2 while chunk c+1 is cut into bf16
//     pieces and parked in LDS in fragment order (XOR swizzled 64-byte rows: conflict-free 16-byte fragment reads and
//     piece writes without padding) -- three chunks of 29 KB per compute unit are in flight, which is what it takes to
//     keep the HBM stream from draining while a chunk is cut (profiles/r04_k6_producer_consumer.md) -- and the PREVIOUS
//     item's rows are finished: cross-modal diagonals added from the other modalities' H rows, whole 400-byte rows per
//     load / store instruction, an eighth of the row block in each of the next item's first eight chunks.
// One s_barrier per 32-wide chunk separates "stage s is being written" from "stage s is being read" (two stages).  The
// matrix pipe therefore runs through prologue, cutting and epilogue of the neighbouring items instead of waiting for them.
//
// All vector-memory instructions of the producers are issued from inline asm and retired with hand-counted s_waitcnt:
//   * gfx950 has one counter for loads and stores and hipcc falls back to vmcnt(0) whenever both kinds are in flight in a
//     wave, which would serialise the row stores with the requests running four chunks ahead;
//   * with compiler-managed loads the register allocator recycles a request's destination registers as temporaries as
//     soon as the previous value is dead, and the waits it then has to insert keep barely one and a half chunks in
//     flight (measured: the tile-strip producers became latency-bound, 129 us per launch).
// Loads return in order among themselves, so "at most Y operations outstanding", Y = the number of loads issued after the
// ones needed, is sufficient whatever the stores do.  The price: between such a load and its wait the compiler believes
// the destination already holds the value, so it must never copy or touch it.  Every asm load therefore has exactly ONE
// site per period in straight-line code (no load inside a branch whose result would meet another definition in a phi),
// the register sets are never initialised (a phi with undef needs no copy), and tools/verify_pc_asm.py checks in the
// generated ISA that the registers named by every wait are the ones written by the matching load and that nothing in
// between mentions them (tests/test_pc_kernel_asm.py runs it on the library's own sources).
//
// Item order: item t <-> (dialogue, modality, row block) with t % 8 == dialogue % 8, workgroup g takes t = g, g + G, ...
// (G a multiple of 8), so every tile of a dialogue is served by one XCD's L2 as in the other K6 kernels.
#include "../../mm_dfn_amd/csrc/mmdfn_internal.h"
#include <stdlib.h>
#include <type_traits>

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int PC_BM = 128;                 // tile rows per item
constexpr int PC_BK = 32;                  // k per chunk (two K = 16 MFMA steps)
constexpr int PC_PIECE = 128 * 64;         // bytes of one bf16 piece array: 128 rows (A) / feature columns (B) x 32 k
constexpr int PC_OPER = 3 * PC_PIECE;      // three pieces
constexpr int PC_STAGE = 2 * PC_OPER;      // A pieces, then B pieces
constexpr int PC_EOFF = 2 * PC_STAGE;      // row buffer of the finished item behind the two stages
constexpr int PC_ES = 116;                 // its row stride in floats (d <= 112)
constexpr int PC_LDS = PC_EOFF + PC_BM * PC_ES * 4;   // 157 696 B
constexpr int PC_NE = 8;                   // the finished item's rows leave in PC_NE slices (16 rows per period)
constexpr int PC_MINP = PC_NE + 1;         // periods per item (>= the slices + the period that parks the next rows)
constexpr int PC_NS = 4;                   // register sets of chunk requests = how many periods they run ahead of the MFMAs
constexpr int PC_LEAD = PC_NS;             // lead-in periods

__device__ __forceinline__ float pc_as_f(uint32_t u) { return __builtin_bit_cast(float, u); }
__device__ __forceinline__ uint32_t pc_as_u(float f) { return __builtin_bit_cast(uint32_t, f); }

__device__ __forceinline__ f32x16 pc_mfma(u32x4 a, u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// all of a workgroup's waves meet here once per period; LDS traffic of the period has been retired
__device__ __forceinline__ void pc_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// ---- the stream of periods a workgroup walks through (identical, wave-uniform bookkeeping in all eight waves) ----
//   lead-in (4 periods: the producers fill the pipeline)  ->  items (max(chunks, 9) periods each)  ->  drain (8 periods:
//   the last item's rows are finished)  ->  end
struct PcCursor {
    int state;          // 0 lead-in, 1 item, 2 drain, 3 end
    int c, P, nch;      // period inside the state, periods of the state, chunks of the item
    int t;              // next item index to examine
    int m, r0, L, ld, rs;
    long long tb;       // float offset of the (dialogue, modality) tile
};

struct PcArgs {
    const int32_t* dia_len;
    const int32_t* row_start;
    const int64_t* tile_base;
    int B, M, max_rb, n_items, G;
};

__device__ __forceinline__ void pc_next_item(PcCursor& k, const PcArgs& a) {
    const int Rd = a.M * a.max_rb;
    while (k.t < a.n_items) {
        const int t = k.t;
        k.t += a.G;
        const int yq = t >> 3;
        const int i = (yq / Rd) * 8 + (t & 7);
        if (i >= a.B) continue;
        const int rho = yq % Rd;
        const int m = rho / a.max_rb;
        const int rb = rho - m * a.max_rb;
        const int L = a.dia_len[i];
        const int r0 = rb * PC_BM;
        if (r0 >= L) continue;
        const int ld = (L + 3) & ~3;
        k.state = 1;
        k.c = 0;
        k.nch = (L + PC_BK - 1) / PC_BK;
        k.P = k.nch > PC_MINP ? k.nch : PC_MINP;
        k.m = m;
        k.r0 = r0;
        k.L = L;
        k.ld = ld;
        k.rs = a.row_start[i];
        k.tb = a.tile_base[i] + (long long)m * L * ld;
        return;
    }
    k.state = 2;
    k.c = 0;
    k.P = PC_NE;
    k.nch = 0;
}

__device__ __forceinline__ void pc_advance(PcCursor& k, const PcArgs& a) {
    if (++k.c < k.P) return;
    if (k.state >= 2) {
        k.state = 3;
        k.c = 0;
        k.P = 1 << 30;
        k.nch = 0;
        return;
    }
    pc_next_item(k, a);
}

// (the comment behind an instruction names its registers for tools/verify_pc_asm.py)
#define PC_GLD4(DST, VOFF, SBASE) \
    asm volatile("global_load_dwordx4 %0, %1, %2 ; pc-load" : "=&v"(DST) : "v"(VOFF), "s"(SBASE) : "memory")
#define PC_GLD1(DST, VOFF, SBASE) \
    asm volatile("global_load_dword %0, %1, %2 ; pc-load" : "=&v"(DST) : "v"(VOFF), "s"(SBASE) : "memory")
// (s_nop: a wide store's data registers must not be overwritten in the two issue slots behind it; hipcc's hazard
// recogniser does not see instructions inside inline asm)
#define PC_GST4(VOFF, SRC, SBASE) \
    asm volatile("global_store_dwordx4 %0, %1, %2\n\ts_nop 1" : : "v"(VOFF), "v"(SRC), "s"(SBASE) : "memory")

template <int MX>   // MX = M - 1 other modalities (compile time: their rows are all in flight at once)
__global__ __launch_bounds__(1024, 4) void propagate_pc_kernel(
    const float* __restrict__ tiles, const float* __restrict__ cross, const float* __restrict__ H,
    float* __restrict__ out, const int32_t* __restrict__ dia_len, const int32_t* __restrict__ row_start,
    const int64_t* __restrict__ tile_base, int B, int M, int N, int d, int ldh, int ldo, int max_rb, int n_items,
    int abl_arg) {
#ifndef MMDFN_TUNING
    constexpr int abl = 0;                 // production build: no ablation paths
    (void)abl_arg;
#else
    const int abl = abl_arg;               // 1: no MFMAs, 2: no cutting, 4: no row epilogue, 64: cycle stamps (timing only)
#endif
    extern __shared__ __attribute__((aligned(16))) unsigned char pc_smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);

    PcArgs args{dia_len, row_start, tile_base, B, M, max_rb, n_items, (int)gridDim.x};
    PcCursor cons;
    cons.state = 0; cons.c = 0; cons.P = PC_LEAD; cons.nch = 0; cons.t = blockIdx.x;
    cons.m = 0; cons.r0 = 0; cons.L = 1; cons.ld = 4; cons.rs = 0; cons.tb = 0;
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    using I3 = std::integral_constant<int, 3>;
#ifdef MMDFN_TUNING
#define PC_STAMP(V) const long long V = (abl & 64) ? (long long)__builtin_readcyclecounter() : 0
#else
#define PC_STAMP(V) (void)0
#endif

#ifdef MMDFN_TUNING
    if (w < 8 ? (abl & 256) : (abl & 128)) __builtin_amdgcn_s_setprio(3);     // 128: producers, 256: consumers at priority 3
#endif
    if (w < 8) {
        // =========================== consumer: fragment reads + MFMAs ===========================
        // wave (rw = w & 3, ch = w >> 2): tile rows 32 rw .. + 31, feature columns 64 ch .. + 63 (two 32-column tiles).
        // Waves w and w + 4 share a SIMD (and their A fragments): two MFMA streams per SIMD keep its matrix pipe busy
        // (one in-order wave alone issues a 32x32x16 MFMA every ~44 cycles, two interleaved every ~34).
        const int l32 = lane & 31;
        const int kg = lane >> 5;
        const int rw = w & 3;
        const int ch = w >> 2;
        f32x16 acc[2];
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[ct][r] = 0.f;
        // fragment address of (row 32 x + l32, 16-byte unit 2 kh + kg) inside a piece array: unit ^ ((row >> 2) & 3);
        // the B arrays additionally keep feature column c in row slot (c & ~3) | ((c & 3) ^ ((c >> 2) & 3)) (so that the
        // H-row producers' piece writes of one instruction spread over all banks)
        const int sw = (l32 >> 2) & 3;
        const int bslot = (l32 & ~3) | ((l32 & 3) ^ sw);
        int foffa[2], foffb[2];
#pragma unroll
        for (int kh = 0; kh < 2; ++kh) {
            foffa[kh] = (32 * rw + l32) * 64 + (((2 * kh + kg) ^ sw) << 4);
            foffb[kh] = PC_OPER + (64 * ch + bslot) * 64 + (((2 * kh + kg) ^ sw) << 4);
        }
        // parked row (r) of accumulator tile ct: row 32 rw + 4 kg + (r & 3) + 8 (r >> 2), column 64 ch + 32 ct + l32
        float* const Ebuf = reinterpret_cast<float*>(pc_smem + PC_EOFF);
        const int ebase = (32 * rw + 4 * kg) * PC_ES + 64 * ch + l32;
        auto chunk = [&](auto par_) {
            constexpr int PAR = decltype(par_)::value;
            const unsigned char* S = pc_smem + PAR * PC_STAGE;
            u32x4 fa[2][3], fb[2][3][2];
            auto lda = [&](int kh, int x) { fa[kh][x] = *reinterpret_cast<const u32x4*>(S + x * PC_PIECE + foffa[kh]); };
            auto ldb = [&](int kh, int x) {
#pragma unroll
                for (int ct = 0; ct < 2; ++ct)
                    fb[kh][x][ct] = *reinterpret_cast<const u32x4*>(S + x * PC_PIECE + ct * 32 * 64 + foffb[kh]);
            };
            auto mm = [&](int kh, int xa, int xb) {
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) acc[ct] = pc_mfma(fa[kh][xa], fb[kh][xb][ct], acc[ct]);
            };
            // product order a3b1 a2b1 a1b1 | a2b2 a1b2 | a1b3 (small terms first), fragments requested in that order;
            // step 1's fragments are requested between the MFMA groups of step 0
            lda(0, 2); ldb(0, 0);
            __builtin_amdgcn_sched_barrier(0);      // (the first MFMA pair waits for these three reads only)
            lda(0, 1); lda(0, 0); ldb(0, 1); ldb(0, 2);
            __builtin_amdgcn_sched_barrier(0);
            mm(0, 2, 0);
            __builtin_amdgcn_sched_barrier(0);
            lda(1, 2); ldb(1, 0);
            __builtin_amdgcn_sched_barrier(0);
            mm(0, 1, 0);

            __builtin_amdgcn_sched_barrier(0);
            lda(1, 1); lda(1, 0); ldb(1, 1);
            __builtin_amdgcn_sched_barrier(0);
            mm(0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            ldb(1, 2);
            __builtin_amdgcn_sched_barrier(0);
            mm(0, 1, 1); mm(0, 0, 1); mm(0, 0, 2);
            mm(1, 2, 0); mm(1, 1, 0); mm(1, 0, 0); mm(1, 1, 1); mm(1, 0, 1); mm(1, 0, 2);
        };

#ifdef MMDFN_TUNING
        long long tm_chunk = 0, tm_bar = 0, tm_n = 0;
        const long long tm_begin = __builtin_readcyclecounter();
        const long long tw_begin = wall_clock64();
#endif
        auto period = [&](auto par_) {
            PC_STAMP(s0);
            if (cons.state == 1) {
                if (cons.c == 0) {
#pragma unroll
                    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[ct][r] = 0.f;
                }
                if (cons.c < cons.nch && !(abl & 1)) chunk(par_);
                if (cons.c == cons.P - 1) {
                    // park the finished rows (the producers read the previous item's last slice >= 1 period ago)
#pragma unroll
                    for (int ct = 0; ct < 2; ++ct)
                        if (64 * ch + 32 * ct + l32 < d) {
#pragma unroll
                            for (int r = 0; r < 16; ++r)
                                Ebuf[ebase + ((r & 3) + 8 * (r >> 2)) * PC_ES + 32 * ct] = acc[ct][r];
                        }
                }
            }
            PC_STAMP(s1);
            pc_barrier();
            PC_STAMP(s2);
#ifdef MMDFN_TUNING
            if (cons.state == 1 && cons.c < cons.nch) { tm_chunk += s1 - s0; tm_bar += s2 - s1; tm_n += 1; }
#endif
            pc_advance(cons, args);
        };
        while (cons.state != 3) {   // four periods per trip like the producers (two LDS stages, four request sets)
            period(I0{});
            period(I1{});
            period(I0{});
            period(I1{});
        }
#ifdef MMDFN_TUNING
        if ((abl & 64) && tid == 0) {   // timing-only run: per-workgroup stamps over the head of the output
            float* o = out + blockIdx.x * 16;
            o[0] = (float)tm_chunk; o[1] = (float)tm_bar; o[2] = (float)tm_n;
            o[3] = (float)((long long)__builtin_readcyclecounter() - tm_begin);
            o[4] = (float)((long long)wall_clock64() - tw_begin);
        }
#endif
        return;
    }

    // =========================== producer: loads, cutting, row epilogue ===========================
    const int pw = w - 8;                    // 0..3: H rows (B operand)   4..7: tile strip (A operand)
    uint32_t himask;
    asm volatile("s_mov_b32 %0, 0xffff0000" : "=s"(himask));

    // ---- A (tile strip), waves 4..7: load j (0..3) of a chunk = tile row 32 (pw - 4) + 8 j + (lane >> 3),
    //      strip columns k0 + 4 (lane & 7) .. + 3   (one instruction = 8 rows x one 128-byte line)
    const int rA = lane >> 3;
    const int qA = lane & 7;
    const int arow0 = 32 * (pw & 3) + rA;
    // LDS byte offset of this lane's 8 bytes in row (arow0 + 8 j): swizzle term (row >> 2) & 3 = (2 j + (rA >> 2)) & 3
    const int a_lds_e = arow0 * 64 + (((qA >> 1) ^ (rA >> 2)) << 4) + ((qA & 1) << 3);          // even j (+ 512 j)
    const int a_lds_o = arow0 * 64 + (((qA >> 1) ^ (2 + (rA >> 2))) << 4) + ((qA & 1) << 3);    // odd j  (+ 512 j)
    // ---- B (H rows), waves 0..3: lane = 16 g + 8 cb + h  ->  feature columns 4 cg .. 4 cg + 3 (cg = 8 pw + 2 g + cb), the
    //      four H rows k0 + 4 h + j: the lane ends up with half a 16-byte unit (4 consecutive k) of four columns.  (The 16
    //      lanes of a piece-write group are 2 column groups x 8 half units: all 32 banks.)
    const int hB = lane & 7;
    const int cg = 8 * (pw & 3) + 2 * (lane >> 4) + ((lane >> 3) & 1);
    const int cw4 = d >> 2;
    const int cgc = cg < cw4 ? cg : cw4 - 1;         // column groups past d: any finite data (never stored)
    const uint32_t b_voff = (uint32_t)cgc * 16u;
    // row slot of column 4 cg + comp: 4 cg + (comp ^ (cg & 3)); unit (hB >> 1) ^ (cg & 3); half hB & 1
    const int b_lds = PC_OPER + 4 * cg * 64 + (((hB >> 1) ^ (cg & 3)) << 4) + ((hB & 1) << 3);
    const int b_x = cg & 3;

    f32x4 raw[PC_NS][4];      // (never initialised: see the note on asm loads at the top)

    // the request cursor runs PC_NS periods ahead of the consumers; what it pointed at PC_NS - 1 periods ago is cut now
    PcCursor ldc = cons;
#pragma unroll
    for (int s = 0; s < PC_NS; ++s) pc_advance(ldc, args);
    bool cutq_real[PC_NS - 1];             // [0]: chunk p + 1 of the period p in progress, [1]: p + 2, ...
    int cutq_k0[PC_NS - 1], cutq_L[PC_NS - 1];
#pragma unroll
    for (int s = 0; s < PC_NS - 1; ++s) { cutq_real[s] = false; cutq_k0[s] = 0; cutq_L[s] = 1; }

    // the finished item whose rows are in the LDS row buffer
    bool has_prev = false;
    int prev_m = 0, prev_grow0 = 0, prev_rows = 0;

    // ---- epilogue task of a period: row 16 c + 2 pw + (lane >> 5) of the finished item, 16 bytes at column 4 (lane & 31)
    const int c4 = lane & 31;
    const int c4c = c4 < cw4 ? c4 : cw4 - 1;
    const float* const Ebuf = reinterpret_cast<const float*>(pc_smem + PC_EOFF);

    // ---- chunk requests, one load at a time (spread over the cutting work of a period: every load holds the wave at
    // issue for as long as the compute unit's address path takes to accept it, ~70 cycles with all producers loading)
    uint32_t ld_voff = 0, ld_step = 0;     // B: running row offset;  A: column offset of the chunk
    int ld_left = 0;
    const float* ld_base = H;
    auto begin_loads = [&](auto isb_) {
        constexpr bool isB = decltype(isb_)::value;
        // a bubble period re-reads the cursor's last item (its fields stay valid): unconditional loads keep the counts exact
        const int k0 = (ldc.state == 1 && ldc.c < ldc.nch) ? ldc.c * PC_BK : 0;
        if (isB) {
            ld_base = H + ((long long)ldc.m * N + ldc.rs) * ldh;
            // rows past the dialogue: the last one again (masked when they are cut)
            int kr = k0 + 4 * hB;
            kr = kr < ldc.L ? kr : ldc.L - 1;
            ld_voff = (uint32_t)kr * (uint32_t)ldh * 4u + b_voff;
            ld_step = (uint32_t)ldh * 4u;
            ld_left = ldc.L - 1 - kr;          // rows that follow inside the dialogue
        } else {
            ld_base = tiles + ldc.tb;
            int ka = k0 + 4 * qA;
            ka = ka < ldc.ld - 4 ? ka : ldc.ld - 4;
            ld_voff = (uint32_t)ka * 4u;
            ld_step = (uint32_t)ldc.ld * 4u;
            ld_left = ldc.L - 1;
        }
    };
    auto issue_load = [&](auto isb_, auto set_, auto j_) {
        constexpr bool isB = decltype(isb_)::value;
        constexpr int SET = decltype(set_)::value;
        constexpr int J = decltype(j_)::value;
        auto& raw_ = raw;      // (clang: operands of an asm statement inside a generic lambda do not capture by themselves)
        const float* const ld_base_ = ld_base;
        if (isB) {
            const uint32_t voff = ld_voff;
            PC_GLD4(raw_[SET][J], voff, ld_base_);
            ld_voff += (J < ld_left) ? ld_step : 0u;
        } else {
            int row = ldc.r0 + arow0 + 8 * J;
            row = row < ld_left ? row : ld_left;
            const uint32_t voff = (uint32_t)row * ld_step + ld_voff;
            PC_GLD4(raw_[SET][J], voff, ld_base_);
        }
    };

    // three exact bf16 pieces of four floats: p1 = hi16(x), x -= p1, p2 = hi16(x), x -= p2, p3 = hi16(x); the four chains
    // advance together (a dependent VALU pair costs more than two independent ones)
    auto cut4 = [&](float x0, float x1, float x2, float x3, u32x2& p1, u32x2& p2, u32x2& p3) {
        p1 = (u32x2){__builtin_amdgcn_perm(pc_as_u(x1), pc_as_u(x0), 0x07060302u),
                     __builtin_amdgcn_perm(pc_as_u(x3), pc_as_u(x2), 0x07060302u)};
        float t0 = pc_as_f(pc_as_u(x0) & himask), t1 = pc_as_f(pc_as_u(x1) & himask);
        float t2 = pc_as_f(pc_as_u(x2) & himask), t3 = pc_as_f(pc_as_u(x3) & himask);
        x0 -= t0; x1 -= t1; x2 -= t2; x3 -= t3;
        p2 = (u32x2){__builtin_amdgcn_perm(pc_as_u(x1), pc_as_u(x0), 0x07060302u),
                     __builtin_amdgcn_perm(pc_as_u(x3), pc_as_u(x2), 0x07060302u)};
        t0 = pc_as_f(pc_as_u(x0) & himask); t1 = pc_as_f(pc_as_u(x1) & himask);
        t2 = pc_as_f(pc_as_u(x2) & himask); t3 = pc_as_f(pc_as_u(x3) & himask);
        x0 -= t0; x1 -= t1; x2 -= t2; x3 -= t3;
        p3 = (u32x2){__builtin_amdgcn_perm(pc_as_u(x1), pc_as_u(x0), 0x07060302u),
                     __builtin_amdgcn_perm(pc_as_u(x3), pc_as_u(x2), 0x07060302u)};
    };

    // a quarter of a thread's cutting work per period: A -- the float4 of load J (row arow0 + 8 J);  B -- column component J
    // of the four rows (half a unit)
    int cut_lim = 0;       // elements (A: of a float4, B: of the four rows) that lie inside the dialogue
    auto cut_item = [&](auto isb_, auto set_, auto stage_, auto j_) {
        constexpr bool isB = decltype(isb_)::value;
        constexpr int SET = decltype(set_)::value;
        constexpr int J = decltype(j_)::value;
        unsigned char* const st = pc_smem + decltype(stage_)::value * PC_STAGE;
        float x[4];
        unsigned char* dst;
        if (isB) {
#pragma unroll
            for (int e = 0; e < 4; ++e) x[e] = (e < cut_lim) ? raw[SET][e][J] : 0.f;   // (ragged last chunk: rows >= L are zero)
            dst = st + b_lds + ((J ^ b_x) << 6);
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) x[e] = (e < cut_lim) ? raw[SET][J][e] : 0.f;   // (strip columns >= L are zero)
            dst = st + ((J & 1) ? a_lds_o : a_lds_e) + 512 * J;
        }
        u32x2 p1, p2, p3;
        cut4(x[0], x[1], x[2], x[3], p1, p2, p3);
        *reinterpret_cast<u32x2*>(dst) = p1;
        *reinterpret_cast<u32x2*>(dst + PC_PIECE) = p2;
        *reinterpret_cast<u32x2*>(dst + 2 * PC_PIECE) = p3;
    };

    // wait until at most Y vector-memory operations are outstanding (Y = loads issued after the ones tied here)
    auto wait_raw = [&](auto set_, auto y_) {
        constexpr int SET = decltype(set_)::value;
        constexpr int Y = decltype(y_)::value;
        auto& raw_ = raw;
        asm volatile("s_waitcnt vmcnt(%[y]) ; pc-wait %0 %1 %2 %3"
                     : "+v"(raw_[SET][0]), "+v"(raw_[SET][1]), "+v"(raw_[SET][2]), "+v"(raw_[SET][3])
                     : [y] "i"(Y) : "memory");
    };

#ifdef MMDFN_TUNING
    long long tp_issue = 0, tp_wait = 0, tp_cut = 0, tp_epi = 0, tp_bar = 0;
#endif
    // period p: PAR = p & 1 is the LDS stage the consumers read, SET = p % PC_NS the register set chunk p + PC_NS is
    // requested into.  Vector-memory order of a period: [item 0: chunk load 0, half of the slice loads] [item 1: chunk load 1,
    // the other half] [item 2: chunk load 2] [item 3: chunk load 3] [slice store]
    auto period = [&](auto isb_, auto par_, auto set_) {
        constexpr bool isB = decltype(isb_)::value;
        PC_STAMP(q0);
        constexpr int PAR = decltype(par_)::value;
        constexpr int SET = decltype(set_)::value;
        using CUTSET = std::integral_constant<int, (SET + 1) % PC_NS>;     // chunk p + 1, requested PC_NS - 1 periods ago
        using CUTSTAGE = std::integral_constant<int, PAR ^ 1>;
        // ---- a slice of the previous item's rows (16 bytes of one row per thread)
        const bool eact = has_prev && (cons.state == 1 || cons.state == 2) && cons.c < PC_NE && !(abl & 4);
        f32x4 eh[MX];
        float ew[MX];
        uint32_t eo = 0, ehoff = 0, ewoff = 0;
        int erow = 0;
        bool ev = false;
        uint32_t sH[MX], sW[MX];       // byte offsets of the other modalities' H blocks / pair diagonals (scalar)
#pragma unroll
        for (int q = 0; q < MX; ++q) { sH[q] = 0; sW[q] = 0; }
        if (eact) {
#pragma unroll
            for (int q = 0; q < MX; ++q) {
                const int n = q + (q >= prev_m ? 1 : 0);
                const int pk = (prev_m < n) ? mmdfn_pair_index(prev_m, n, M) : mmdfn_pair_index(n, prev_m, M);
                sH[q] = (uint32_t)n * (uint32_t)N * (uint32_t)ldh * 4u;
                sW[q] = (uint32_t)pk * (uint32_t)N * 4u;
            }
            const int rl = 16 * cons.c + 2 * pw + (lane >> 5);
            erow = rl;
            ev = (rl < prev_rows) && (c4 < cw4);
            const int rlc = rl < prev_rows ? rl : prev_rows - 1;
            const uint32_t grow = (uint32_t)(prev_grow0 + rlc);
            ehoff = (grow * (uint32_t)ldh + 4u * (uint32_t)c4c) * 4u;
            ewoff = grow * 4u;
            eo = (grow * (uint32_t)ldo + 4u * (uint32_t)c4c) * 4u;
        }
        // slice loads q, spread over items 0 and 1 (ONE asm site per load: the values meet nothing but undef at the join)
        auto slice_loads = [&](auto j_) {
            constexpr int J = decltype(j_)::value;
            auto& eh_ = eh;
            auto& ew_ = ew;
            const float* const H_ = H;
            const float* const cross_ = cross;
            if (J < 2 && eact) {
#pragma unroll
                for (int q = (MX * J) / 2; q < (MX * (J + 1)) / 2; ++q) {
                    const uint32_t ho = ehoff + sH[q];
                    const uint32_t wo = ewoff + sW[q];
                    PC_GLD4(eh_[q], ho, H_);
                    PC_GLD1(ew_[q], wo, cross_);
                }
            }
        };
        begin_loads(isb_);
        cut_lim = cutq_L[0] - cutq_k0[0] - (isB ? 4 * hB : 4 * qA);    // ragged last chunk: strip columns / H rows >= L are zero
        const bool docut = cutq_real[0] && !(abl & 2);
        PC_STAMP(q1);
        // ---- chunk p + 1 (requested PC_NS - 1 periods ago): the 4 (PC_NS - 2) requests of chunks p + 2 .. p + PC_NS - 1
        //      are younger
        wait_raw(CUTSET{}, std::integral_constant<int, 4 * (PC_NS - 2)>{});
        PC_STAMP(q2);
        // ---- it is cut and chunk p + PC_NS requested, one quarter at a time (the loads stand OUTSIDE the branch)
        auto item = [&](auto j_) {
            if (docut) cut_item(isb_, CUTSET{}, CUTSTAGE{}, j_);
            issue_load(isb_, set_, j_);
            slice_loads(j_);
        };
        item(I0{});
        item(I1{});
        item(I2{});
        item(I3{});
        PC_STAMP(q3);
        // ---- finish the slice: chunk loads 2 and 3 are younger than its operands
        if (eact) {
            asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
#pragma unroll
            for (int q = 0; q < MX; ++q) asm volatile("; pc-wait %0 %1" : "+v"(eh[q]), "+v"(ew[q]));
            const float* ob = out + (long long)prev_m * N * ldo;
            f32x4 v = *reinterpret_cast<const f32x4*>(Ebuf + erow * PC_ES + 4 * c4c);
#pragma unroll
            for (int q = 0; q < MX; ++q) {
                const float cwt = ew[q];
                const f32x4 h = eh[q];
                v.x = fmaf(cwt, h.x, v.x);
                v.y = fmaf(cwt, h.y, v.y);
                v.z = fmaf(cwt, h.z, v.z);
                v.w = fmaf(cwt, h.w, v.w);
            }
            if (ev && !(abl & 8)) PC_GST4(eo, v, ob);
        }
        PC_STAMP(q4);
        pc_barrier();
        PC_STAMP(q5);
#ifdef MMDFN_TUNING
        if (cons.state == 1 && cons.c < cons.nch) {
            tp_issue += q1 - q0; tp_wait += q2 - q1; tp_cut += q3 - q2; tp_epi += q4 - q3; tp_bar += q5 - q4;
        }
#endif
        // ---- bookkeeping (wave-uniform)
#pragma unroll
        for (int s = 0; s + 1 < PC_NS - 1; ++s) {
            cutq_real[s] = cutq_real[s + 1]; cutq_k0[s] = cutq_k0[s + 1]; cutq_L[s] = cutq_L[s + 1];
        }
        cutq_real[PC_NS - 2] = ldc.state == 1 && ldc.c < ldc.nch;
        cutq_k0[PC_NS - 2] = ldc.c * PC_BK;
        cutq_L[PC_NS - 2] = ldc.L;
        pc_advance(ldc, args);
        if (cons.state == 1 && cons.c == cons.P - 1) {
            has_prev = true;
            prev_m = cons.m;
            prev_grow0 = cons.rs + cons.r0;
            prev_rows = cons.L - cons.r0 < PC_BM ? cons.L - cons.r0 : PC_BM;
        }
        pc_advance(cons, args);
    };
    static_assert(PC_NS == 4, "the period loop below is unrolled for four request sets");
    if (pw < 4) {
        while (cons.state != 3) {
            period(std::true_type{}, I0{}, I0{});
            period(std::true_type{}, I1{}, I1{});
            period(std::true_type{}, I0{}, I2{});
            period(std::true_type{}, I1{}, I3{});
        }
    } else {
        while (cons.state != 3) {
            period(std::false_type{}, I0{}, I0{});
            period(std::false_type{}, I1{}, I1{});
            period(std::false_type{}, I0{}, I2{});
            period(std::false_type{}, I1{}, I3{});
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef MMDFN_TUNING
    if ((abl & 64) && (tid == 512 || tid == 768)) {
        float* o = out + blockIdx.x * 16 + (tid == 512 ? 5 : 10);
        o[0] = (float)tp_issue; o[1] = (float)tp_wait; o[2] = (float)tp_cut; o[3] = (float)tp_epi; o[4] = (float)tp_bar;
    }
#endif
#undef PC_STAMP
}

#ifdef MMDFN_TUNING
int pc_ablation() {
    const char* e = getenv("MMDFN_PC_ABL");
    return e ? atoi(e) : 0;
}
#else
constexpr int pc_ablation() { return 0; }
#endif

int pc_num_cus() {
    static int n = 0;
    if (n == 0) {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) == hipSuccess &&
            hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0)
            n = v;
        else
            n = 256;
    }
    return n;
}

}  // namespace

// returns -2 when the shape is not covered
static int launch_propagate_pc(const float* tiles, const float* cross, const float* H, float* out,
                              const int32_t* dia_len, const int32_t* row_start, const int64_t* tile_base,
                              int B, int M, int N, int d, int ldh, int ldo, int max_len, hipStream_t s) {
    if ((d & 3) || d > 112 || d < 4 || (M != 2 && M != 3 && M != 6)) return -2;
    // 32-bit byte offsets from the base pointers inside the kernel
    const long long big = (long long)M * N * (ldh > ldo ? ldh : ldo) * 4;
    const long long pairs = (long long)M * (M - 1) / 2 * N * 4;
    if (big >= (1LL << 31) || pairs >= (1LL << 31) || (long long)max_len * (max_len + 4) * 4 >= (1LL << 31)) return -2;
    const int max_rb = (max_len + PC_BM - 1) / PC_BM;
    const int n_items = ((B + 7) / 8) * 8 * M * max_rb;
    int G = pc_num_cus() & ~7;
    if (G < 8) G = 8;
    if (G > n_items) G = n_items;
#define PC_LAUNCH(MXV)                                                                                               \
    do {                                                                                                             \
        if (mmdfn_allow_big_lds(propagate_pc_kernel<MXV>)) return -2;                                                \
        hipLaunchKernelGGL((propagate_pc_kernel<MXV>), dim3(G), dim3(1024), PC_LDS, s, tiles, cross, H, out, dia_len, \
                           row_start, tile_base, B, M, N, d, ldh, ldo, max_rb, n_items, pc_ablation());              \
    } while (0)
    switch (M) {                 // (instantiated for the modality counts of the reference and of BASELINE cfg5)
        case 2: PC_LAUNCH(1); break;
        case 3: PC_LAUNCH(2); break;
        case 6: PC_LAUNCH(5); break;
        default: return -2;
    }
#undef PC_LAUNCH
    MMDFN_CHECK_LAUNCH();
    return 0;
}

// the experiment's C entry (tools/k6_pc/pc_ops.py binds it with ctypes; not part of libmmdfn_hip.so since round 5)
extern "C" int k6pc_propagate(const float* tiles, const float* cross, const float* H, float* out, const int32_t* dia_len,
                              const int32_t* row_start, const int64_t* tile_base, int B, int M, int N, int d, int ldh, int ldo,
                              int max_len, void* stream) {
    return launch_propagate_pc(tiles, cross, H, out, dia_len, row_start, tile_base, B, M, N, d, ldh, ldo, max_len,
                               (hipStream_t)stream);
}
