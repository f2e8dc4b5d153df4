// K6 (large-dialogue variant): out = A_hat . H with the fp32 product carried by bf16 MFMAs.
//
// Replaces torch.spmm(adj, input) (reference model_GCN.py:178) like propagate.hip, for launches that are
// bound by the exact-f32 MFMA rate (v_mfma_f32_16x16x4_f32: 32 cycles per 2 kflop on a SIMD).  Each fp32
// operand is cut -- exactly, by truncation -- into three bf16 pieces  x = x1 + x2 + x3  (8 + 8 + 8
// significant bits) and the product is assembled from the six piece products whose weight is >= 2^-16:
//     a.b ~= a1b1 + a1b2 + a2b1 + a1b3 + a2b2 + a3b1          (dropped: a2b3 + a3b2 + a3b3 <= 2^-23 |a||b|)
// Every piece product is exact in the fp32 accumulator, so the result carries fp32-level error (the dropped
// terms are the size of one fp32 rounding of a.b; measured max |err| vs an fp64 product: 1.8e-7, the same as
// the f32-MFMA kernel) while six bf16 MFMAs per K=16 replace four f32 MFMAs per K=16 at 1/16 of the cycles
// per flop: 2.7x less matrix-pipe time.
//
// Work decomposition: workgroup = (dialogue, modality, 128 tile rows, 128 feature columns); 4 waves x 32 rows x
// 4 v_mfma_f32_32x32x16_bf16 column tiles.
//   A (the tile strip) goes HBM -> registers in MFMA layout (lane (row, kg) holds 8 consecutive k per K=16
//     step; the k permutation inside a 32-wide chunk makes each load instruction fetch 32 contiguous bytes per
//     row) and is cut in registers.
//   B (the H rows of the tile) is cut ONCE per workgroup and parked in LDS as three bf16 [col][k-slot] arrays
//     whose 16-byte reads ARE the MFMA B fragments (row stride 20 dwords: conflict-free per quarter wave);
//     double-buffered, one barrier per chunk.
//   Each of the 48 MFMAs of a chunk is followed by one of the 48 cutting stages (~4 VALU) of the NEXT chunk,
//     pinned there with scheduling barriers, so no wave ever sits in a VALU-only phase while the matrix pipe
//     idles.  (Measured on MI355X, tools/ubench/mfma_bf16_fillers.hip + profiles/r01_propagate_tuning.md: a
//     32x32x16 MFMA hides at most ~7 trailing VALU, and a second wave on the SIMD adds no VALU/MFMA overlap, so
//     the kernel's time is ~ MFMA time + 4 cycles per other instruction: the instruction count of the cutting is
//     what bounds it.)
// Cross-modal diagonals are added in the LDS row epilogue as in propagate.hip.  XCD mapping: blockIdx % 8 ==
// dialogue % 8.
#include "mmdfn_internal.h"
#include <stdlib.h>

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int SBK = 32;    // k per chunk = two K=16 MFMA steps
constexpr int SROW = 20;   // LDS row stride in dwords: 16 (32 bf16) + 4 pad; 16 consecutive rows -> 64 banks

__device__ __forceinline__ float as_f(uint32_t u) { return __builtin_bit_cast(float, u); }
__device__ __forceinline__ uint32_t as_u(float f) { return __builtin_bit_cast(uint32_t, f); }

__device__ __forceinline__ f32x16 mfma_bf16(u32x4 a, u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 mfma_bf16_16(u32x4 a, u32x4 b, f32x4 c) {     // (the pipeline's tail tile)
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// ABLC (profiling aid, compile time): 1 = no cutting stages, 2 = no MFMAs, 4 = no MFMAs of the fourth column tile (the
// upper bound of what a tail tile for columns 96.. can buy at d = 100; results wrong)
// TAIL: 96 < d <= 112 (the reference's d = 100): three 32-column tiles + a 16-column tail tile, see split_mfma_pipeline.h
template <int ABLC, bool TAIL = false>
__global__ __launch_bounds__(256, 2) void propagate_split_kernel(
    const float* __restrict__ tiles, const float* __restrict__ cross, const float* __restrict__ H,
    float* __restrict__ out, const int32_t* __restrict__ dia_len, const int32_t* __restrict__ row_start,
    const int64_t* __restrict__ tile_base, int B, int M, int N, int d, int ldh, int ldo, int max_rb, int ncb, int abl_arg,
    unsigned long long* trace_arg) {
#ifndef MMDFN_TUNING
    constexpr int abl = 0;                 // production build: no ablation paths
    (void)abl_arg;
    (void)trace_arg;
#define SPLIT_STAMP(K) do { } while (0)
#else
    const int abl = abl_arg;
    unsigned long long* trace = trace_arg;
    // per-workgroup timeline (tools/k6_trace.py): slots 0..5 = s_memtime stamps, 7 = (XCC_ID << 32) | HW_ID
#define SPLIT_STAMP(K)                                                                                     \
    do {                                                                                                   \
        if (trace && threadIdx.x == 0) trace[(int64_t)blockIdx.x * 8 + (K)] = __builtin_readcyclecounter(); \
    } while (0)
    SPLIT_STAMP(0);
    if (trace && threadIdx.x == 0) {
        unsigned hwid, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        trace[(int64_t)blockIdx.x * 8 + 7] = ((unsigned long long)xcc << 32) | hwid;
    }
#endif
    constexpr int NCT = TAIL ? 3 : 4;      // 32-column MFMA tiles
    constexpr bool SPLIT_TAIL = TAIL;
    constexpr int WROWS = 32;              // tile rows per wave
    constexpr int BM = 4 * WROWS;          // 128 tile rows per workgroup
    constexpr int CB = 128;                // feature columns staged per workgroup
    constexpr int LDO = CB + 8;            // epilogue row stride (floats): 4 rows apart -> 32 banks apart
    constexpr int OROWS = 64;              // output rows staged per epilogue pass
    constexpr int split_stride = 128 * SROW;   // dwords per bf16 piece array: one row per column (rows >= d hold zeros)
    constexpr int stage_stride = 3 * split_stride;
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];

    // XCD-aware decode: bid % 8 == dialogue % 8
    // (wider feature blocks, d > 128: one workgroup per 128-column block, column blocks of a row block adjacent)
    const int Rt = max_rb * ncb;
    const int Rd = M * Rt;
    const int bid = blockIdx.x;
    int yq = bid >> 3;
#ifdef MMDFN_TUNING
    // (experiment) co-residency pairing: if workgroups yq and yq + 32 of an XCD share a CU, make them neighbours v = 2a, 2a + 1
    const int dec = (abl >> 8) & 3;
    if (dec) {
        const int tot = (int)(gridDim.x >> 3);
        const int blk = yq >> 6, j = yq & 63;
        if ((blk + 1) * 64 <= tot) yq = blk * 64 + 2 * (j & 31) + (j >> 5);
    }
#else
    constexpr int dec = 0;
#endif
    const int i = (yq / Rd) * 8 + (bid & 7);
    if (i >= B) return;
    const int rho = yq % Rd;
    // dec 2: modality fastest (neighbours = two modalities of one row block: they share four of their five cross-modal rows);
    // dec 0 / 1: row block fastest (neighbours = two row blocks of one modality: they share the B operand stream)
    const int m = dec == 2 ? rho % M : rho / Rt;
    const int rb = dec == 2 ? (rho / M) / ncb : (rho - m * Rt) / ncb;
    const int c0 = dec == 2 ? ((rho / M) % ncb) * CB : ((rho - m * Rt) - rb * ncb) * CB;
    const int dloc = (d - c0 < CB) ? d - c0 : CB;      // columns of this block that exist
    const int L = dia_len[i];
    const int r0 = rb * BM;
    if (r0 >= L) return;
    const int ld = (L + 3) & ~3;
    const int rs = row_start[i];
    const float* T = tiles + tile_base[i] + (int64_t)m * L * ld;
    const float* Hm = H + ((int64_t)m * N + rs) * ldh + c0;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = tid >> 6;
    const int l32 = lane & 31;
    const int kg = lane >> 5;
    const int wrow0 = r0 + WROWS * w;

    f32x16 acc[NCT];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[ct][r] = 0.f;
    f32x4 acct[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};      // (TAIL) rows 16 half + 4 (lane >> 4) + r, column 96 + (lane & 15)
    const int tboff = (96 + (lane & 15)) * SROW + ((lane >> 4) == 0 ? 0 : (lane >> 4) == 1 ? 8 : (lane >> 4) == 2 ? 4 : 12);

    // ---- k permutation inside a chunk: MFMA step kh (0,1), lane group kg (0,1), element e (0..7)  <->
    //      k = 16 kh + 8 (e >> 2) + 4 kg + (e & 3)
    // B staging tasks: thread -> column (tid & 127), slots (kh = 0 and 1, kg = tid >> 7): 2 x 8 k values
    const int bcol = tid & 127;
    const bool bok = bcol < dloc;
    const int bcolc = bok ? bcol : 0;
    const int bkg = __builtin_amdgcn_readfirstlane(tid >> 7);   // wave-uniform -> scalar row arithmetic
    const int blds = bcol * SROW + 4 * bkg;                     // + 8 for the kh = 1 slot

    const int arow = wrow0 + l32;
    const float* a_lane = T + (int64_t)(arow < L ? arow : L - 1) * ld;
    int boff[NCT];                         // fragment read offsets (dwords) inside a piece array, + 8 kh
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) boff[ct] = (32 * ct + l32) * SROW + 4 * kg;

    const int nchunks = (L + SBK - 1) / SBK;
    const int klast = (nchunks - 1) * SBK;
    const int nfull = L / SBK;             // chunks whose 32 k values all lie inside the tile
    const int limA = L - 4 * kg;           // strip column k0 + kpos is data iff k0 + kpos - 4 kg < limA
    const int limB = bok ? L - 4 * bkg : -(1 << 30);   // threads of columns >= d stage zeros (no branch around the LDS stores)

    // Software pipeline, period = one chunk C (parity P = C & 1, static after unrolling by two):
    //   top     : issue the HBM/L2 loads of chunk C+2  -> raw set P        (unconditional, clamped: a branch
    //             around loads makes the s_waitcnt at the join conservative and serialises the pipeline)
    //   body    : 48 MFMAs of chunk C (pieces set P, LDS stage P), each followed by one cutting stage of chunk
    //             C+1 (raw set P^1, loaded one period ago): B pieces -> LDS stage P^1, A pieces -> set P^1
    //   barrier : stage P^1 complete, stage P free for chunk C+2
    // SAFE = 0: every row / strip column the loads touch is inside the tile -> no clamps, no masks, and the
    // addresses are (loop-invariant per-lane offset) + (scalar chunk base).  SAFE = 1: the ragged last chunk
    // is involved (clamped addresses, out-of-range k zeroed when the values are cut).
    int bsoff[2][8];                       // (scalar) element offsets of the 16 staged H rows inside a chunk
#pragma unroll
    for (int e = 0; e < 2; ++e)
#pragma unroll
        for (int j = 0; j < 8; ++j) bsoff[e][j] = (16 * e + 4 * bkg + (j & 3) + 8 * (j >> 2)) * ldh;
    const float* a_lane4 = a_lane + 4 * kg;

#define SPLIT_ISSUE(SET, K0, SAFE)                                                                         \
    do {                                                                                                   \
        if (SAFE) {                                                                                        \
            _Pragma("unroll") for (int e = 0; e < 2; ++e)                                                  \
                _Pragma("unroll") for (int j = 0; j < 8; ++j) {                                            \
                    const int k_ = (K0) + 16 * e + 4 * bkg + (j & 3) + 8 * (j >> 2);                       \
                    const float* rowp_ = Hm + (int64_t)(k_ < L ? k_ : L - 1) * ldh;                        \
                    braw[SET][e][j] = rowp_[bcolc];                                                        \
                }                                                                                          \
            _Pragma("unroll") for (int f = 0; f < 4; ++f) {                                                \
                const int ka_ = (K0) + 8 * f + 4 * kg;                                                     \
                araw[SET][f] = *reinterpret_cast<const float4*>(a_lane + (ka_ < ld ? ka_ : ld - 4));       \
            }                                                                                              \
        } else {                                                                                           \
            const float* hb_ = Hm + (int64_t)(K0) * ldh;                                                   \
            _Pragma("unroll") for (int e = 0; e < 2; ++e)                                                  \
                _Pragma("unroll") for (int j = 0; j < 8; ++j) braw[SET][e][j] = (hb_ + bsoff[e][j])[bcolc]; \
            const float* ab_ = a_lane4 + (K0);                                                             \
            _Pragma("unroll") for (int f = 0; f < 4; ++f)                                                  \
                araw[SET][f] = *reinterpret_cast<const float4*>(ab_ + 8 * f);                              \
        }                                                                                                  \
    } while (0)

    SPLIT_STAMP(1);
constexpr bool SPLIT_BPRE = false;
#include "split_mfma_pipeline.h"   // araw / braw / piece sets, cutting stages, MFMA steps, the chunk loops

    // ---- epilogue through LDS, OROWS rows per pass (the loop's last barrier has retired every fragment read).
    // Four threads per output row, each owning every 4th float4 of it: the M-1 cross-modal weights of the
    // row are loaded once per thread and all its H loads of one modality are in flight together.
    float* Os = reinterpret_cast<float*>(smem);
    SPLIT_STAMP(2);
    if (abl & 32) return;
    __syncthreads();   // the last step's (unused) fragment reloads have retired
    constexpr int NJ = TAIL ? 7 : CB / 16;   // float4 slots per thread and row (d <= 112: at most 28 float4 per row)
    const int cw4 = dloc / 4;
    const int erow = tid >> 2;
    const int eq = tid & 3;
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
            if constexpr (TAIL) {
#pragma unroll
                for (int hf = 0; hf < 2; ++hf)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        Os[(WROWS * (w & 1) + 16 * hf + 4 * (lane >> 4) + r) * LDO + 96 + (lane & 15)] = acct[hf][r];
            }
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
            const int nq = (abl & 1) ? 0 : (abl & 2) ? 1 : M - 1;   // (2: one cross-modal row instead of M - 1, timing only)
#pragma unroll 2
            for (int q = 0; q < nq; ++q) {
                const int n = q + (q >= m ? 1 : 0);
                const int pk = (m < n) ? mmdfn_pair_index(m, n, M) : mmdfn_pair_index(n, m, M);
                const float cwt = cross[(int64_t)pk * N + grow];
                // (abl & 8, timing only: the cross-modal rows of a workgroup alias 8 rows -> L1 hits)
                const float* hrow = H + ((int64_t)n * N + ((abl & 8) ? rs + (row & 7) : grow)) * ldh + c0;
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
            float* orow = out + ((int64_t)m * N + grow) * ldo + c0;
#pragma unroll
            for (int j = 0; j < NJ; ++j)
                if (eq + 4 * j < cw4) *reinterpret_cast<float4*>(orow + coff[j]) = v[j];
        }
        SPLIT_STAMP(3 + pass);
    }
#undef SPLIT_STAMP
}

#ifdef MMDFN_TUNING
int split_ablation() {
    const char* e = getenv("MMDFN_PROP_ABL");
    return e ? atoi(e) : 0;
}
unsigned long long* split_trace() {       // device address of a (workgroups x 8) uint64 buffer (tools/k6_trace.py)
    const char* e = getenv("MMDFN_TRACE_PTR");
    return e ? reinterpret_cast<unsigned long long*>(strtoull(e, nullptr, 10)) : nullptr;
}
#else
constexpr int split_ablation() { return 0; }
constexpr unsigned long long* split_trace() { return nullptr; }
#endif

}  // namespace

// returns -2 when the shape is not covered (caller falls back to the f32-MFMA kernel)
int mmdfn_launch_propagate_split(const float* tiles, const float* cross, const float* H, float* out,
                                 const int32_t* dia_len, const int32_t* row_start, const int64_t* tile_base,
                                 int B, int M, int N, int d, int ldh, int ldo, int max_len, hipStream_t s) {
    if (d & 3) return -2;
    const int max_rb = (max_len + 127) / 128;
    const int ncb = (d + 127) / 128;
    const int lds_bytes = 2 * 3 * 128 * SROW * 4;   // 61440 B (>= the 64 x 136 float epilogue staging)
    dim3 grid(((B + 7) / 8) * 8 * M * max_rb * ncb);
    bool tail = d > 96 && d <= 112;        // three 32-column tiles + the 16-column tail tile
#ifdef MMDFN_TUNING
    if (const char* e = getenv("MMDFN_SPLIT_TAIL")) tail = tail && atoi(e) != 0;      // A/B aid
#endif
#define SPLIT_LAUNCH(A)                                                                                          \
    do {                                                                                                         \
        if (tail && (A) == 0)                                                                                    \
            hipLaunchKernelGGL((propagate_split_kernel<0, true>), grid, dim3(256), lds_bytes, s, tiles, cross, H, out, dia_len, \
                               row_start, tile_base, B, M, N, d, ldh, ldo, max_rb, ncb, split_ablation(), split_trace()); \
        else                                                                                                     \
            hipLaunchKernelGGL((propagate_split_kernel<A>), grid, dim3(256), lds_bytes, s, tiles, cross, H, out, dia_len, \
                               row_start, tile_base, B, M, N, d, ldh, ldo, max_rb, ncb, split_ablation(), split_trace()); \
    } while (0)
#ifdef MMDFN_TUNING
    const char* ac = getenv("MMDFN_SPLIT_ABLC");  // compile-time ablations (1: no cutting, 2: no MFMA, 4: no fourth-tile MFMAs)
    const int ablc = ac ? atoi(ac) : 0;
    if (ablc == 1) SPLIT_LAUNCH(1);
    else if (ablc == 2) SPLIT_LAUNCH(2);
    else if (ablc == 4) SPLIT_LAUNCH(4);
    else
#endif
    SPLIT_LAUNCH(0);
#undef SPLIT_LAUNCH
    MMDFN_CHECK_LAUNCH();
    return 0;
}
