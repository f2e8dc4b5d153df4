// Internal helpers shared by the gfx950 kernels of libmmdfn_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define MMDFN_CHECK_LAUNCH()                         \
    do {                                             \
        hipError_t e__ = hipGetLastError();          \
        if (e__ != hipSuccess) return (int)e__;      \
    } while (0)

// sim(c) = 1 - acos(0.99999 c) / pi          (model_mm.py:149-150, 166-167)
#define MMDFN_COS_SHRINK 0.99999f
#define MMDFN_PI_F 3.14159265358979323846f

__device__ __forceinline__ float mmdfn_sim(float c) {
    return 1.0f - acosf(c * MMDFN_COS_SHRINK) / MMDFN_PI_F;
}
// d sim / d c = a / (pi sqrt(1 - (a c)^2))
__device__ __forceinline__ float mmdfn_dsim(float c) {
    float ac = c * MMDFN_COS_SHRINK;
    return MMDFN_COS_SHRINK / (MMDFN_PI_F * sqrtf(1.0f - ac * ac));
}

// index of the unordered modality pair (m < n) in lexicographic order
__host__ __device__ __forceinline__ int mmdfn_pair_index(int m, int n, int M) {
    return m * (2 * M - m - 1) / 2 + (n - m - 1);
}

// Reductions on the DPP path (hipcc turns __shfl_xor into ds_bpermute_b32: an LDS round trip per step).  After the two quad
// steps every quad holds its sum in all four lanes, so the mirror steps are exchanges between equal halves: all 16 lanes of a
// row end with the same bits.
template <int CTRL>
__device__ __forceinline__ float mmdfn_dpp_mov(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
// sum over each row of 16 lanes (every lane of the row gets it)
__device__ __forceinline__ float row_sum16(float v) {
    v += mmdfn_dpp_mov<0xB1>(v);          // quad_perm [1,0,3,2]
    v += mmdfn_dpp_mov<0x4E>(v);          // quad_perm [2,3,0,1]
    v += mmdfn_dpp_mov<0x141>(v);         // row_half_mirror
    v += mmdfn_dpp_mov<0x140>(v);         // row_mirror
    return v;
}
__device__ __forceinline__ float lane_value(float v, int l) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l));
}
// sum over the whole wave (every lane gets it): four row sums met through scalar registers in a fixed order
__device__ __forceinline__ float wave_sum(float v) {
    v = row_sum16(v);
    return (lane_value(v, 0) + lane_value(v, 16)) + (lane_value(v, 32) + lane_value(v, 48));
}
// sum over the 32 lanes of each half of the wave (lanes 0-31 get the lower half's sum, 32-63 the upper's)
__device__ __forceinline__ float half_sum32(float v) {
    v = row_sum16(v);
    const float lo = lane_value(v, 0) + lane_value(v, 16);
    const float hi = lane_value(v, 32) + lane_value(v, 48);
    return (threadIdx.x & 32) ? hi : lo;
}

// more than 64 KB of dynamic LDS per workgroup needs the attribute raised once per kernel (gfx950: 160 KB per CU)
template <class Kern>
inline int mmdfn_allow_big_lds(Kern kern) {
    static thread_local const void* done[48] = {nullptr};      // (one table per kernel SIGNATURE: kernels of one type share it)
    const void* key = reinterpret_cast<const void*>(kern);
    for (int i = 0; i < 48; ++i)
        if (done[i] == key) return 0;
    // (a little below the 160 KB of a CU: kernels may also hold a few hundred bytes of static LDS)
    hipError_t e = hipFuncSetAttribute(key, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024);
    if (e != hipSuccess) {
        (void)hipGetLastError();      // do not leave the error for the next launch check to find
        return (int)e;
    }
    for (int i = 0; i < 48; ++i)
        if (done[i] == nullptr) { done[i] = key; break; }
    return 0;
}

// launchers implemented in propagate.hip / tile_dot.hip, used by adjacency.hip
int mmdfn_launch_propagate(const float* tiles, const float* cross, const float* H, float* out,
                           const int32_t* dia_len, const int32_t* row_start, const int64_t* tile_base,
                           int B, int M, int N, int d, int ldh, int ldo, int max_len, int transpose, hipStream_t s);

// one-workgroup-per-(dialogue, modality) form of the adjacency build for short dialogues (adjacency_small.hip); -2 = not covered
int mmdfn_launch_adj_small_fwd(const float* feats, float* unit, float* norm, float* cosg, float* cdot, float* rdeg,
                               float* tiles, float* cross, const int32_t* dia_len, const int32_t* row_start,
                               const int64_t* tile_base, int B, int M, int N, int D, int max_len, float modal_weight,
                               hipStream_t s);
int mmdfn_launch_adj_small_bwd(const float* dtiles, const float* dcross, const float* unit, const float* norm,
                               const float* cosg, const float* cdot, const float* rdeg, const float* tiles,
                               const float* cross, const float* addend, float* dfeats, const int32_t* dia_len,
                               const int32_t* row_start, const int64_t* tile_base, int B, int M, int N, int D, int max_len,
                               float modal_weight, hipStream_t s);

// bf16-piece variant of the forward product for large launches (propagate_split.hip); -2 = shape not covered
int mmdfn_launch_propagate_split(const float* tiles, const float* cross, const float* H, float* out,
                                 const int32_t* dia_len, const int32_t* row_start, const int64_t* tile_base,
                                 int B, int M, int N, int d, int ldh, int ldo, int max_len, hipStream_t s);

// bf16-piece variant of the dense projection for many-row launches (linear_split.hip); -2 = shape not covered
int mmdfn_launch_linear_split(const float* X, const float* W, const float* W2, int N1, const float* bias,
                              const float* bias2, float* Y, int R, int K, int N, int ldx, int ldy, int act, int accumulate,
                              hipStream_t s);

// bf16-piece form of the GCN stack's LSTM cell for many-row launches (lstm_gate_split.hip); -2 = shape not covered
int mmdfn_launch_lstm_gate_fwd_split(const float* q, const float* h, const float* c, const float* Wih, const float* Whh,
                                     const float* bsum, const float* bsum2, float* gates, float* h_out, float* c_out, int R,
                                     int H, int ldh, const void* planes, hipStream_t s);
// piece planes of the cell's weights for the form above (cut once per step; `planes` = null: every workgroup cuts its own)
int64_t mmdfn_lstm_gate_planes_floats(int H);
int mmdfn_launch_lstm_gate_cut(const float* Wih, const float* Whh, void* planes, int H, hipStream_t s);

int mmdfn_launch_tile_dot_split(const float* X, const float* Y, float* out_tiles, const int32_t* dia_len,
                                const int32_t* row_start, const int64_t* tile_base, int B, int M, int N, int K, int ldx,
                                int ldy, int max_len, int accumulate, hipStream_t s);

// EPI 0: dtiles (+)= X.Y^T ; EPI 1: cosine Gram + raw similarity + row degree
int mmdfn_launch_tile_dot(const float* X, const float* Y, float* out_tiles, float* out_aux, float* deg,
                          const int32_t* dia_len, const int32_t* row_start, const int64_t* tile_base,
                          int B, int M, int N, int K, int ldx, int ldy, int max_len, int epi, int accumulate, hipStream_t s);

// bf16-piece form of the weight-gradient batch (gemm_tn_split.hip): the segment table of one launch.  A workgroup owns a
// MMDFN_TNS_TM x MMDFN_TNS_TN output tile of segment p over the rows [split * rows_per_split, ...) of its split; tiles =
// row tiles x nblocks (column blocks).  part / colpart: the slab stacks of gemm_tn.hip's batch ([split][M][N], [split][M]).
// Block index -> (split, tile) inside a segment's range of 8 ceil(splits / 8) tiles blocks: XCD = block % 8 takes the splits
// = its number (mod 8), all tiles of a split back to back -- the workgroups that read the same operand rows run on one XCD
// at the same time and share them through its L2 (blocks of splits past the last one exit at once).
#define MMDFN_TNS_BK 32
#define MMDFN_TNS_TM 128
#define MMDFN_TNS_TN 112
constexpr int MMDFN_TNS_MAXSEG = 40;
struct TnSplitSegs {
    const float* A[MMDFN_TNS_MAXSEG];
    const float* B[MMDFN_TNS_MAXSEG];
    float* part[MMDFN_TNS_MAXSEG];
    float* colpart[MMDFN_TNS_MAXSEG];
    int R[MMDFN_TNS_MAXSEG], lda[MMDFN_TNS_MAXSEG], ldb[MMDFN_TNS_MAXSEG], bshift[MMDFN_TNS_MAXSEG];
    int rows_per_split[MMDFN_TNS_MAXSEG], splits[MMDFN_TNS_MAXSEG], tiles[MMDFN_TNS_MAXSEG], nblocks[MMDFN_TNS_MAXSEG];
    int M[MMDFN_TNS_MAXSEG], N[MMDFN_TNS_MAXSEG];
    int wide[MMDFN_TNS_MAXSEG];          // != 0: 128 x (2 MMDFN_TNS_TN) tiles, nblocks counts 224-column blocks (gemm_tn_split.hip)
    int wg_prefix[MMDFN_TNS_MAXSEG + 1];
    int n;
};
int mmdfn_launch_gemm_tn_split(const TnSplitSegs& sq, hipStream_t s);

// Riders of the GRU backward launch (gru.hip, gru_seq_bwd_riders_kernel): a weight-gradient batch STAGED by
// mmdfn_wgrad_riders_stage (gemm_tn.hip) instead of launched; the next plain GRU backward launch of the one-sequence-per-workgroup
// kind runs its tiles as extra workgroups on the CUs the recurrence leaves idle, then calls mmdfn_riders_launched (the slab
// reduction).  Whatever is still pending when the host asks (mmdfn_wgrad_riders_flush) is launched the ordinary way.
constexpr int MMDFN_RIDER_MAXSEG = 16;
struct TnRiderSegs {
    const float* A[MMDFN_RIDER_MAXSEG];
    const float* B[MMDFN_RIDER_MAXSEG];
    float* part[MMDFN_RIDER_MAXSEG];
    float* colpart[MMDFN_RIDER_MAXSEG];
    int R[MMDFN_RIDER_MAXSEG], lda[MMDFN_RIDER_MAXSEG], ldb[MMDFN_RIDER_MAXSEG], bshift[MMDFN_RIDER_MAXSEG];
    int rows_per_split[MMDFN_RIDER_MAXSEG], splits[MMDFN_RIDER_MAXSEG], tiles[MMDFN_RIDER_MAXSEG], nblocks[MMDFN_RIDER_MAXSEG];
    int M[MMDFN_RIDER_MAXSEG], N[MMDFN_RIDER_MAXSEG];
    int wide[MMDFN_RIDER_MAXSEG];
    int wg_prefix[MMDFN_RIDER_MAXSEG + 1];
    int n;
};
inline TnRiderSegs mmdfn_rider_table(const TnSplitSegs& t) {      // (t.n <= MMDFN_RIDER_MAXSEG)
    TnRiderSegs rq;
    for (int k = 0; k < MMDFN_RIDER_MAXSEG; ++k) {
        rq.A[k] = t.A[k]; rq.B[k] = t.B[k]; rq.part[k] = t.part[k]; rq.colpart[k] = t.colpart[k];
        rq.R[k] = t.R[k]; rq.lda[k] = t.lda[k]; rq.ldb[k] = t.ldb[k]; rq.bshift[k] = t.bshift[k];
        rq.rows_per_split[k] = t.rows_per_split[k]; rq.splits[k] = t.splits[k]; rq.tiles[k] = t.tiles[k];
        rq.nblocks[k] = t.nblocks[k]; rq.M[k] = t.M[k]; rq.N[k] = t.N[k]; rq.wide[k] = t.wide[k];
        rq.wg_prefix[k] = t.wg_prefix[k];
    }
    rq.wg_prefix[MMDFN_RIDER_MAXSEG] = t.wg_prefix[t.n];
    rq.n = t.n;
    return rq;
}
const TnSplitSegs* mmdfn_riders_pending();
int mmdfn_riders_launched(hipStream_t s);

// MFMA form of the GRU recurrence for launches with very many sequences (gru_mfma.hip): 16 sequences per workgroup, the
// recurrent products on bf16 pieces.  Same operands and layouts as mmdfn_gru_seq_fwd / _bwd; -2 = not covered.
int mmdfn_launch_gru_fwd_mfma(int ngroups, const float* const* gi, const float* const* w_hh, const float* const* b_hh,
                              float* const* y, float* const* gates, const int* rows, const int* T, hipStream_t s);
int mmdfn_launch_gru_bwd_mfma(int ngroups, const float* const* dy, const float* const* y, const float* const* gates,
                              const float* const* w_hh, float* const* dgi, float* const* dgh, const int* rows, const int* T,
                              hipStream_t s);
