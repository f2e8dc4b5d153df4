// Weight-gradient contraction  C (M x N) = A^T B  on the bf16 matrix path: the batch form of gemm_tn.hip (every dW = dY^T X of a
// training step in one launch: autograd of nn.Linear / nn.GRU / nn.LSTM in the reference) with each fp32 operand cut exactly
// into three bf16 pieces and the six piece products of weight >= 2^-16 issued as v_mfma_f32_16x16x32_bf16 (fp32-level error,
// see propagate_split.hip for the arithmetic) instead of exact-f32 16x16x4 MFMAs (1/16 of the bf16 rate).
//
// The contraction index is the ROW index r of both operands, i.e. both are k-major: an MFMA fragment (8 consecutive k of one
// column) is a column walk through memory.  So the operands are transposed on chip:
//   * a workgroup owns a 128 x 112 output tile over the row range of its split.  Per chunk of 32 rows the staging threads load
//     four float4 of A (32 rows x 128 columns) and four of B (32 x 112) each -- full rows, coalesced --, cut them in registers
//     (5.5 VALU per element: each element is cut ONCE per workgroup and chunk) and write the pieces as three ROW-MAJOR bf16
//     planes per operand to LDS (288-byte rows);
//   * fragments come out of the planes with ds_read_b64_tr_b16, the hardware transpose read: a 16-lane group reads a
//     [4 rows][16 columns] block and each lane receives one column of it.  A chunk is exactly one K = 32 MFMA step; lane group
//     g takes rows 4 g .. 4 g + 3 and 16 + 4 g .. 16 + 4 g + 3 of the chunk for BOTH operands (any assignment works as long as
//     it is the same on both sides).  Row stride 288 bytes = 72 dwords = 8 (mod 64): the 8 rows the 32 lanes of one LDS cycle
//     read fall on 8 different 8-bank groups -- conflict-free (SQ_LDS_BANK_CONFLICT = 0);
//   * a wave multiplies 32 output rows (two row tiles) by all 7 column tiles: 12 + 42 transpose reads and 84 MFMAs per chunk.
// Two groups of four waves work in opposite phases on alternate chunks of the same tile.  Waves w and w + 4 share a SIMD and
// belong to different groups: at any time every SIMD has one wave multiplying (matrix pipe, LDS reads) and one staging (loads,
// vector unit, LDS writes).  Step k: group k & 1 multiplies chunk k out of its own LDS stage, the other group stages chunk k + 1
// into its stage and requests chunk k + 3; one barrier per step.  Each group accumulates its own chunks; the second group's
// accumulators are added through LDS at the end.
// What bounds it (profiles/r04_wgrad_bf16_pieces.md): a SIMD issues about one instruction per 5.5 cycles whatever its two waves
// send (MFMA, VALU and LDS instructions of different waves do not pair up), so the step costs its instruction count: 84 MFMAs +
// 54 reads on one side, ~210 staging instructions on the other.  Forms that were measured and dropped: eight symmetric waves with
// the staging interleaved behind the MFMAs slot by slot, and two independent four-wave workgroups per CU (they drift into phase).
// Row shifts (recurrent weights: row r of A pairs with row r + shift of B), row ranges, column sums of A (bias gradients,
// accumulated in fp32 by the staging threads before the cut) and the slab workspace are those of the tiled batch kernel; the
// slab reduction kernel is shared.  Block index -> (split, tile): see TnSplitSegs (XCD-aware).
#include "gemm_tn_split_body.h"
#include <stdlib.h>


namespace {

using namespace tnsb;

// ABL: see tns_block
template <int ABL>
__global__ __launch_bounds__(512, 2) void gemm_tn_split_kernel(const TnSplitSegs sq, float* __restrict__ trace) {
    extern __shared__ __attribute__((aligned(16))) unsigned char tns_smem[];
    tns_block<ABL>(sq, (int)blockIdx.x, tns_smem, trace);
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
            case 7: kern = gemm_tn_split_kernel<7>; break;
            case 8: kern = gemm_tn_split_kernel<8>; break;
            case 15: kern = gemm_tn_split_kernel<15>; break;
            case 16: kern = gemm_tn_split_kernel<16>; break;
            default: break;
        }
    }
#endif
    if (int e = mmdfn_allow_big_lds(kern)) return e;
    hipLaunchKernelGGL(kern, dim3(sq.wg_prefix[sq.n]), dim3(512), LDS_B, s, sq, trace);
    MMDFN_CHECK_LAUNCH();
    return 0;
}
