// Shared main loop of the piece-plane kernels (linear_planes.hip, gcn_planes.hip): a CODE FRAGMENT included inside the kernel
// body, not a header of declarations.
//
// Computes, for one wave,  acc[h][.] += A (32 RH rows x K) . B^T (one 32-column tile)  where B arrives as bf16 piece planes (MFMA
// B-fragment order, planes_common.h) and A is produced by the includer, 16 k at a time, as bf16 pieces parked in LDS in MFMA
// A-fragment order:  As[buffer][((k-step in phase) * 3 + piece) * 2 + row half][lane] = the 16 bytes lane `lane` feeds
// v_mfma_f32_32x32x16_bf16 (row = lane & 31 of the half, k = 8 (lane >> 5) + 0..7 of the k-step).  Six piece products per MAC (the
// arithmetic of propagate_split.hip: fp32-level error).
//
// Phases of PL_STG = 2 k-steps, everything double-buffered with STATIC indices (the phase loop is unrolled by two): the B fragments
// of phase p + 1 are requested during phase p -- one coalesced 1 KB load per piece and k-step, L2 -> registers, no LDS, no barrier
// on the B side -- the A operands of phase p + 2 are requested at the start of phase p (PL_ISSUE_X), and those of phase p + 1 are cut
// and parked in the other LDS buffer behind phase p's last MFMA (PL_PARK); ONE barrier per phase.  (Vector-memory results
// retire in order: requesting A two phases ahead and B one phase ahead keeps every wait behind a whole phase of MFMAs.  A form with
// asm requests and hand-counted vmcnt waits measured the same inside the step and 10 % slower alone -- it needs 3 waves per SIMD
// where hipcc's schedule fits 4 -- and was removed; round 6)
//
// The includer provides, before the #include:
//   constexpr int RH (row halves per wave: 1 or 2), NACC (accumulators per half: consecutive MFMAs never share one; 2 for RH = 1)
//   f32x16 acc[RH][NACC] (zeroed), u32x4* As (PL_LDS elements of LDS), const u32x4* bsrc (this wave's tile: planes + ct KS 3 64 + lane),
//   int lane, myh (RH = 1: the row half this wave multiplies), KS, NPH = ceil(KS / PL_STG)
//   PL_ISSUE_X(PAR, PH)       request this wave's share of the A operands of phase PH into its raw register set PAR
//   PL_PARK(PAR, PH, BUF)     cut raw set PAR (phase PH; zeros beyond K) and park the pieces in LDS buffer BUF
// A phase index past the end (PL_ISSUE_X only) must be harmless: the fragment clamps it to NPH - 1 and never uses the result.
    u32x4 bq[2][PL_STG][3];                              // [phase parity][k-step][piece]
#define PL_AS(BUF, IDX) As[(BUF) * (PL_LDS / 2) + (IDX)]
#define PL_ISSUE_B(PAR, PH)                                                                                 \
    do {                                                                                                    \
        _Pragma("unroll") for (int j_ = 0; j_ < PL_STG; ++j_) {                                             \
            const int ks_ = PL_STG * (PH) + j_ < KS ? PL_STG * (PH) + j_ : KS - 1;                           \
            _Pragma("unroll") for (int p_ = 0; p_ < 3; ++p_) bq[PAR][j_][p_] = bsrc[((int64_t)ks_ * 3 + p_) * 64]; \
        }                                                                                                   \
    } while (0)
    // one phase: request A of phase PH + 2 and the fragments of phase PH + 1, multiply phase PH, park A of phase PH + 1.
    // (unconditional, clamped requests; a k-step past KS multiplies zero A pieces -- parked as zeros beyond K -- by the last real
    // fragments: no branch inside the phase)
#define PL_PHASE(PAR, PH)                                                                                   \
    do {                                                                                                    \
        PL_ISSUE_X(PAR, (PH) + 2 < NPH ? (PH) + 2 : NPH - 1);                                               \
        PL_ISSUE_B((PAR) ^ 1, (PH) + 1 < NPH ? (PH) + 1 : NPH - 1);                                         \
        u32x4 a_[PL_STG][RH][3];                                                                            \
        _Pragma("unroll") for (int j_ = 0; j_ < PL_STG; ++j_)                                               \
            _Pragma("unroll") for (int p_ = 0; p_ < 3; ++p_)                                                \
                _Pragma("unroll") for (int h_ = 0; h_ < RH; ++h_)                                           \
                    a_[j_][h_][p_] = PL_AS(PAR, ((j_ * 3 + p_) * 2 + (RH == 2 ? h_ : myh)) * 64 + lane);    \
        _Pragma("unroll") for (int j_ = 0; j_ < PL_STG; ++j_) {                                             \
            /* products: against b1: a3 a2 a1;  against b2: a2 a1;  against b3: a1 */                      \
            _Pragma("unroll") for (int q_ = 0; q_ < 3; ++q_) {                                              \
                const int ahi_ = 2 - q_, alo_ = (q_ == 0) ? 1 : 0, blo_ = (q_ < 2) ? 1 : 2;                 \
                _Pragma("unroll") for (int h_ = 0; h_ < RH; ++h_)                                           \
                    acc[h_][0] = pl_mfma(a_[j_][h_][ahi_], bq[PAR][j_][0], acc[h_][0]);                     \
                _Pragma("unroll") for (int h_ = 0; h_ < RH; ++h_)                                           \
                    acc[h_][NACC - 1] = pl_mfma(a_[j_][h_][alo_], bq[PAR][j_][blo_], acc[h_][NACC - 1]);    \
            }                                                                                               \
        }                                                                                                   \
        if ((PH) + 1 < NPH) {                                                                               \
            PL_PARK((PAR) ^ 1, (PH) + 1, (PAR) ^ 1);    /* (the other buffer's readers passed the previous barrier) */ \
            __syncthreads();                                                                                \
        }                                                                                                   \
    } while (0)

    static_assert(PL_STG == 2, "the phase loop below is unrolled for two register / LDS parities");
    PL_ISSUE_X(0, 0);
    PL_ISSUE_X(1, NPH > 1 ? 1 : 0);
    PL_ISSUE_B(0, 0);
    PL_PARK(0, 0, 0);
    __syncthreads();
    for (int ph_ = 0; ph_ < NPH; ph_ += 2) {
        PL_PHASE(0, ph_);
        if (ph_ + 1 < NPH) PL_PHASE(1, ph_ + 1);
    }
#undef PL_PHASE
#undef PL_ISSUE_B
#undef PL_AS
