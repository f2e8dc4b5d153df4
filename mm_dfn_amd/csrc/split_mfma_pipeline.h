// Shared main loop of the bf16-piece MFMA kernels (propagate_split.hip, linear_split.hip): a CODE FRAGMENT included
// inside the kernel body, not a header of declarations.
//
// Computes, for one workgroup (4 waves x 32 rows x 4 column tiles of 32),  acc[ct] += A (128 x K) . B (K x 128)  with
// every fp32 operand cut exactly into three bf16 pieces and the six piece products of weight >= 2^-16 issued as
// v_mfma_f32_32x32x16_bf16 (see propagate_split.hip for the arithmetic).  K is walked in chunks of 32.
//
// k permutation inside a chunk: MFMA step kh (0,1), lane group kg (0,1), element e (0..7)  <->
//      k = 16 kh + 8 (e >> 2) + 4 kg + (e & 3)
//
// The includer provides, before the #include:
//   SPLIT_ISSUE(SET, K0, SAFE)  loads of the chunk starting at k = K0 into the raw register sets
//        araw[SET][f]     f = 0..3: float4 of this lane's A row at k = K0 + 8 f + 4 kg .. +3
//        braw[SET][e][j]  e = 0,1 (kh), j = 0..7: B[k = K0 + 16 e + 4 bkg + (j&3) + 8 (j>>2)][column bcol]
//        SAFE = 1: the chunk may reach past K (clamped addresses); SAFE = 0: fully inside
//   variables: smem, stage_stride, split_stride, blds, boff[NCT], acc[NCT], NCT (= 4; 3 with the tail tile), ABLC,
//        nchunks, klast = (nchunks-1)*32, nfull = K / 32, limA = K - 4 kg, limB = (column staged ? K - 4 bkg : -inf)
//   SPLIT_BPRE (constexpr bool): true = the B operand arrives as bf16 pieces (SPLIT_ISSUE fills bpre[SET][kh][piece] with the
//        three u32x4 of this thread's k-slot instead of braw): a WEIGHT operand is cut once per step by a small kernel, the 24
//        B-side cutting stages of every chunk vanish (lstm_gate_split.hip with its piece planes).
//   SPLIT_TAIL (constexpr bool): false = four 32-column tiles; true (propagate_split.hip, 96 < d <= 112) = three of them + a
//        16-column TAIL tile (columns 96 .. 111) on v_mfma_f32_16x16x32_bf16: the fourth 32-column tile of a d = 100 launch
//        carries 4 useful columns in 12 of the chunk's 48 MFMAs; the tail tile replaces them by 12 half-length ones (6 piece
//        products x two 16-row halves, K = 32 each).  The includer then also provides  acct[2] (f32x4, rows 16 half + 4 (lane
//        >> 4) + r, column 96 + (lane & 15))  and  tboff  (dword offset of this lane's tail fragment inside a piece array).
// and declares nothing with the names used below.  After the fragment every LDS read has been issued; the includer
// must __syncthreads() before reusing smem.
    float4 araw[2][4];
    float braw[2][2][8];
    u32x4 bpre[2][2][3];                       // (SPLIT_BPRE) [set][kh][piece]: B pieces that arrive cut (a weight cut once per step)
    u32x4 ap1[2][2], ap2[2][2], ap3[2][2];     // [set][kh]


    // The cutting work of one chunk = 16 units (U 0..7: B pair (kh = U>>2, p = U&3) of this thread's staging
    // tasks; U 8..15: A pair (kh = (U-8)>>2, p = U&3)) x 3 stages of ~5 VALU (mask + first piece, second
    // piece, third piece).  Stage T (0..47) is issued right behind the T-th MFMA of the chunk.
    float cx0 = 0.f, cx1 = 0.f;            // the pair in flight through the three stages
    uint32_t himask;                       // in an SGPR: a literal operand would double the v_and encoding size
    asm volatile("s_mov_b32 %0, 0xffff0000" : "=s"(himask));
#define SPLIT_STAGE(SET, K0, STG, T, SAFE)                                                                     \
    do {                                                                                                   \
        const int u_ = (T) / 3, st_ = (T) % 3, p_ = u_ & 3, h_ = (u_ >> 2) & 1;                            \
        const int kp_ = (K0) + 16 * h_ + 8 * (p_ >> 1) + 2 * (p_ & 1);   /* + 4 kg (folded into lim) */    \
        if (SPLIT_BPRE && u_ < 8) {                                                                        \
            /* B pieces arrive cut (SPLIT_ISSUE filled bpre): no VALU on this side, only the LDS stores */ \
            if (p_ == 3 && st_ == 2) {                                                                     \
                uint32_t* dst_ = smem + (STG) * stage_stride + blds + 8 * h_;                              \
                *reinterpret_cast<u32x4*>(dst_) = bpre[SET][h_][0];                                        \
                *reinterpret_cast<u32x4*>(dst_ + split_stride) = bpre[SET][h_][1];                         \
                *reinterpret_cast<u32x4*>(dst_ + 2 * split_stride) = bpre[SET][h_][2];                     \
            }                                                                                              \
            break;                                                                                         \
        }                                                                                                  \
        if (st_ == 0) {                                                                                    \
            if (u_ < 8) {                                                                                  \
                cx0 = (!(SAFE) || kp_ < limB) ? braw[SET][h_][2 * p_] : 0.f;                               \
                cx1 = (!(SAFE) || kp_ + 1 < limB) ? braw[SET][h_][2 * p_ + 1] : 0.f;                       \
            } else {                                                                                       \
                const float4 v_ = araw[SET][2 * h_ + (p_ >> 1)];                                           \
                cx0 = (!(SAFE) || kp_ < limA) ? ((p_ & 1) ? v_.z : v_.x) : 0.f;                            \
                cx1 = (!(SAFE) || kp_ + 1 < limA) ? ((p_ & 1) ? v_.w : v_.y) : 0.f;                        \
            }                                                                                              \
        }                                                                                                  \
        const uint32_t w_ = __builtin_amdgcn_perm(as_u(cx1), as_u(cx0), 0x07060302u);                      \
        if (u_ < 8) {                                                                                      \
            if (st_ == 0) bp1[h_][p_] = w_; else if (st_ == 1) bp2[h_][p_] = w_; else bp3[h_][p_] = w_;    \
        } else {                                                                                           \
            if (st_ == 0) ap1[SET][h_][p_] = w_; else if (st_ == 1) ap2[SET][h_][p_] = w_; else ap3[SET][h_][p_] = w_; \
        }                                                                                                  \
        if (st_ < 2) {                                                                                     \
            cx0 = cx0 - as_f(as_u(cx0) & himask);                                                          \
            cx1 = cx1 - as_f(as_u(cx1) & himask);                                                          \
        }                                                                                                  \
        if (u_ < 8 && p_ == 3 && st_ == 2) {                                                               \
            uint32_t* dst_ = smem + (STG) * stage_stride + blds + 8 * h_;                                  \
            *reinterpret_cast<u32x4*>(dst_) = bp1[h_];                                                     \
            *reinterpret_cast<u32x4*>(dst_ + split_stride) = bp2[h_];                                      \
            *reinterpret_cast<u32x4*>(dst_ + 2 * split_stride) = bp3[h_];                                  \
        }                                                                                                  \
    } while (0)

    // B fragments: bf_[column tile][piece] of ONE K=16 step, 48 registers, reloaded piece by piece as soon as a
    // piece's last MFMA of the step has been issued (product order below), so the next step's fragments are
    // in flight >= 12 MFMAs (384 cycles) before their first use without a second register set.
#define SPLIT_LOADB(PIECE, STG, KH)                                                                        \
    do {                                                                                                   \
        _Pragma("unroll") for (int ct_ = 0; ct_ < NCT; ++ct_)                                              \
            bf_[ct_][PIECE] = *reinterpret_cast<const u32x4*>(smem + (STG) * stage_stride + (PIECE) * split_stride + \
                                                              boff[ct_] + 8 * (KH));                       \
    } while (0)

    // one K=16 step: 6 piece products x 4 column tiles = 24 MFMAs, the four accumulators round-robin (a
    // 32x32x16 MFMA's result is needed again only 4 MFMAs = 128 cycles later).  Product order
    // a3b1 a2b1 a1b1 | a2b2 a1b2 | a1b3  frees b1, then b2, then b3 for the reload of step (NSTG, NKH).
// Cutting stages per MFMA slot.  Four tiles: 24 slots per step, one stage each (stage = 24 KH + slot).  Tail mode: 18 slots per
// step + 12 tail slots behind step 1; the 24 B-side stages must retire before the mid-chunk barrier, so the first six slots of
// step 0 carry two CONSECUTIVE stages (a pair flows through three consecutive stages), step 1 carries stages 24 .. 41 and the
// first six tail slots the rest.
#define SPLIT_SLOT_STAGES(P, KH, SLOT, K1, SAFE)                                                           \
    do {                                                                                                   \
        if constexpr (!SPLIT_TAIL) {                                                                       \
            SPLIT_STAGE((P) ^ 1, K1, (P) ^ 1, 24 * (KH) + (SLOT), SAFE);                                   \
        } else if ((KH) == 0) {                                                                            \
            if ((SLOT) < 6) {                                                                              \
                SPLIT_STAGE((P) ^ 1, K1, (P) ^ 1, 2 * (SLOT), SAFE);                                       \
                SPLIT_STAGE((P) ^ 1, K1, (P) ^ 1, 2 * (SLOT) + 1, SAFE);                                   \
            } else {                                                                                       \
                SPLIT_STAGE((P) ^ 1, K1, (P) ^ 1, (SLOT) + 6, SAFE);                                       \
            }                                                                                              \
        } else {                                                                                           \
            SPLIT_STAGE((P) ^ 1, K1, (P) ^ 1, 24 + (SLOT), SAFE);                                          \
        }                                                                                                  \
    } while (0)
#define SPLIT_STEP(P, KH, NSTG, NKH, K1, SAFE)                                                             \
    do {                                                                                                   \
        _Pragma("unroll") for (int pc_ = 0; pc_ < 6; ++pc_) {                                              \
            const u32x4 av_ = (pc_ == 0) ? ap3[P][KH] : (pc_ == 1 || pc_ == 3) ? ap2[P][KH] : ap1[P][KH];  \
            const int bi_ = (pc_ < 3) ? 0 : (pc_ < 5) ? 1 : 2;                                             \
            _Pragma("unroll") for (int ct_ = 0; ct_ < NCT; ++ct_) {                                        \
                if (!(ABLC & 2) && !((ABLC & 4) && ct_ == 3)) acc[ct_] = mfma_bf16(av_, bf_[ct_][bi_], acc[ct_]);                       \
                if (!(ABLC & 1)) SPLIT_SLOT_STAGES(P, KH, NCT * pc_ + ct_, K1, SAFE);                      \
                __builtin_amdgcn_sched_barrier(0);                                                         \
            }                                                                                              \
            if (pc_ == 2) SPLIT_LOADB(0, NSTG, NKH);                                                       \
            if (pc_ == 4) SPLIT_LOADB(1, NSTG, NKH);                                                       \
            if (pc_ == 5) SPLIT_LOADB(2, NSTG, NKH);                                                       \
            __builtin_amdgcn_sched_barrier(0);                                                             \
        }                                                                                                  \
    } while (0)

    // One barrier per chunk, in the MIDDLE: by then every wave has stored its B pieces of chunk C+1 (cutting
    // stages 0..23) and has issued its last fragment read of chunk C (step 1's fragments are reloaded during
    // step 0), so after it LDS stage P^1 may be read (fragments of chunk C+1, step 0) and stage P may be
    // overwritten (chunk C+2, during the first half of the next period).
// The tail tile of chunk C, behind its step 1 (the A pieces of set P are dead then): v_permlane16_swap turns the two K = 16
// operands of a piece (lane rows: [rows 0-15 | rows 16-31] x lane group kg, for kh = 0 and kh = 1) into two K = 32 operands of 16
// rows each (lane group j of rows 0-15: k slots {kh0 kg0, kh1 kg0, kh0 kg1, kh1 kg1}; tboff reads the B side in that order), six
// piece products per half, then the tail fragments of chunk C + 1 are requested from stage P ^ 1 (complete since the mid-chunk
// barrier, overwritten only behind the next one).
#define SPLIT_TAIL_SWAP(X)                                                                                 \
    do {                                                                                                   \
        _Pragma("unroll") for (int d_ = 0; d_ < 4; ++d_) {                                                 \
            const auto r_ = __builtin_amdgcn_permlane16_swap(X[P_][0][d_], X[P_][1][d_], false, false);    \
            X[P_][0][d_] = r_[0];                                                                          \
            X[P_][1][d_] = r_[1];                                                                          \
        }                                                                                                  \
    } while (0)
#define SPLIT_TAIL_BLOCK(P, K1, SAFE)                                                                      \
    do {                                                                                                   \
        constexpr int P_ = (P);                                                                            \
        SPLIT_TAIL_SWAP(ap1); SPLIT_TAIL_SWAP(ap2); SPLIT_TAIL_SWAP(ap3);                                  \
        _Pragma("unroll") for (int pc_ = 0; pc_ < 6; ++pc_) {                                              \
            const int bi_ = (pc_ < 3) ? 0 : (pc_ < 5) ? 1 : 2;                                             \
            _Pragma("unroll") for (int hf_ = 0; hf_ < 2; ++hf_) {                                          \
                const u32x4 av_ = (pc_ == 0) ? ap3[P][hf_] : (pc_ == 1 || pc_ == 3) ? ap2[P][hf_] : ap1[P][hf_]; \
                if (!(ABLC & 2)) acct[hf_] = mfma_bf16_16(av_, bft_[bi_], acct[hf_]);                      \
                if (!(ABLC & 1) && 2 * pc_ + hf_ < 6) SPLIT_STAGE((P) ^ 1, K1, (P) ^ 1, 42 + 2 * pc_ + hf_, SAFE); \
                __builtin_amdgcn_sched_barrier(0);                                                         \
            }                                                                                              \
        }                                                                                                  \
        _Pragma("unroll") for (int q_ = 0; q_ < 3; ++q_)                                                   \
            bft_[q_] = *reinterpret_cast<const u32x4*>(smem + ((P) ^ 1) * stage_stride + q_ * split_stride + tboff); \
    } while (0)

#define SPLIT_BODY(P, C, SAFE)                                                                             \
    do {                                                                                                   \
        const int kn1_ = ((C) + 1) * SBK < klast ? ((C) + 1) * SBK : klast;                                \
        const int kn2_ = ((C) + 2) * SBK < klast ? ((C) + 2) * SBK : klast;                                \
        SPLIT_ISSUE(P, kn2_, SAFE);                                                                        \
        __builtin_amdgcn_sched_barrier(0); /* the loads lead the period: one full chunk of MFMAs hides them */ \
        u32x4 bp1[2], bp2[2], bp3[2];                                                                      \
        /* one wait for the whole previous period's loads (only this period's 20 may stay in flight) */   \
        /* instead of a decreasing vmcnt in front of every cutting stage */                               \
        __builtin_amdgcn_s_waitcnt(0x4F74);  /* vmcnt(20) */                                               \
        SPLIT_STEP(P, 0, P, 1, kn1_, SAFE);                                                                \
        __syncthreads();                                                                                   \
        SPLIT_STEP(P, 1, (P) ^ 1, 0, kn1_, SAFE);                                                          \
        if constexpr (SPLIT_TAIL) SPLIT_TAIL_BLOCK(P, kn1_, SAFE);                                         \
    } while (0)

    // chunks C+1 and C+2 (cut / loaded during chunk C) lie entirely inside the tile <=> C + 2 < nfull.
    // (The fast body stages column 0 again in the LDS rows of columns >= d: those only feed accumulator
    // columns >= d, which are never stored.)
    u32x4 bf_[NCT][3];
    u32x4 bft_[3];                         // (tail mode) the tail tile's B fragments of one chunk, one per piece
    {   // prologue: chunk 0 -> pieces set 0 / LDS stage 0; chunk 1 raw -> set 1
        SPLIT_ISSUE(1, 0, 1);
        u32x4 bp1[2], bp2[2], bp3[2];
#pragma unroll
        for (int t = 0; t < 48; ++t) SPLIT_STAGE(1, 0, 0, t, 1);
#pragma unroll
        for (int kh = 0; kh < 2; ++kh) { ap1[0][kh] = ap1[1][kh]; ap2[0][kh] = ap2[1][kh]; ap3[0][kh] = ap3[1][kh]; }
        SPLIT_ISSUE(1, (SBK < klast ? SBK : klast), 1);
        __syncthreads();
        SPLIT_LOADB(0, 0, 0);
        SPLIT_LOADB(1, 0, 0);
        SPLIT_LOADB(2, 0, 0);
        if constexpr (SPLIT_TAIL) {
#pragma unroll
            for (int q_ = 0; q_ < 3; ++q_) bft_[q_] = *reinterpret_cast<const u32x4*>(smem + q_ * split_stride + tboff);
        }
    }
    {
        // Period C loads chunk min(C+2, last) and cuts chunk min(C+1, last); only the PARTIAL last chunk (K % 32 != 0)
        // needs clamps and masks, so the clamp/mask-free body runs for C + 2 < nfull, and for every C when K % 32 == 0
        // (the past-the-end loads then re-read the last full chunk).
        const int nsafe0 = (nfull == nchunks) ? nchunks : (nfull > 2 ? nfull - 2 : 0);
        int c = 0;
        for (; c + 1 < nsafe0; c += 2) {
            SPLIT_BODY(0, c, 0);
            SPLIT_BODY(1, c + 1, 0);
        }
        for (; c + 1 < nchunks; c += 2) {      // the last few chunks: clamped loads, masked cutting
            SPLIT_BODY(0, c, 1);
            SPLIT_BODY(1, c + 1, 1);
        }
        if (nchunks & 1) SPLIT_BODY(0, c, 1);
    }
#undef SPLIT_BODY
#undef SPLIT_TAIL_BLOCK
#undef SPLIT_TAIL_SWAP
#undef SPLIT_STEP
#undef SPLIT_SLOT_STAGES
#undef SPLIT_LOADB
#undef SPLIT_STAGE
#undef SPLIT_ISSUE
