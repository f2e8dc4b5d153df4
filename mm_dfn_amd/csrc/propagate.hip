// K6: scatter-propagate  out = A_hat . H  over the block-tile dialogue graph.
// (v1 kernel below is kept for the TRANSPOSED product; the hot forward path is propagate_v2_kernel.)
//
// Replaces torch.spmm(adj, input) (reference model_GCN.py:178) without ever
// materialising the dense (MN x MN) matrix: per (dialogue i, modality m) the
// intra-modal L_i x L_i tile is a small dense contraction (exact-f32 MFMA
// 16x16x4), the M(M-1) cross-modal diagonals are a fused axpy in the epilogue.
//
// Work decomposition: one workgroup = (dialogue, modality, 16*NW tile rows,
// 16*NCT feature columns); each of the NW waves owns 16 output rows.
//   A operand (tile strip): read ONCE, straight from HBM into MFMA layout with
//     two 16-byte loads per lane per 32-wide k-chunk (lane (row, g) holds
//     T[row][k0+8g .. k0+8g+7]; the k-permutation is applied to B as well).
//   B operand (H rows of this dialogue/modality): staged through LDS with
//     coalesced 16-byte loads, shared by the NW waves; LDS row stride = 2 mod 4
//     so the ds_read_b32 fragment reads are bank-conflict free.
#include "mmdfn_internal.h"
#include "../../include/mmdfn_hip.h"
#include <stdlib.h>

namespace {

constexpr int BK = 32;

template <int NW, int NCT, bool TRANS>
__global__ __launch_bounds__(64 * NW) void propagate_kernel(
    const float* __restrict__ tiles, const float* __restrict__ cross, const float* __restrict__ H,
    float* __restrict__ out, const int32_t* __restrict__ dia_len, const int32_t* __restrict__ row_start,
    const int64_t* __restrict__ tile_base, int M, int N, int d, int ldh, int ldo, int max_rb) {
    constexpr int BM = 16 * NW;
    constexpr int CB = 16 * NCT;
    constexpr int LDH = CB + 2;
    constexpr int LDT = BM + 2;
    __shared__ float Hs[BK * LDH];
    __shared__ float Ts[TRANS ? BK * LDT : 1];

    const int i = blockIdx.x / max_rb;
    const int rb = blockIdx.x % max_rb;
    const int m = blockIdx.y;
    const int c0 = blockIdx.z * CB;
    const int L = dia_len[i];
    const int r0 = rb * BM;
    if (r0 >= L) return;
    const int ld = (L + 3) & ~3;
    const int rs = row_start[i];
    const float* T = tiles + tile_base[i] + (int64_t)m * L * ld;
    const float* Hm = H + ((int64_t)m * N + rs) * ldh;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = tid >> 6;
    const int frow = lane & 15;
    const int g = lane >> 4;
    const int arow = r0 + 16 * w + frow;

    f32x4 acc[NCT];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) acc[ct] = (f32x4){0.f, 0.f, 0.f, 0.f};

    for (int k0 = 0; k0 < L; k0 += BK) {
        __syncthreads();
        // ---- stage H[k0 .. k0+31][c0 .. c0+CB) into LDS
        for (int idx = tid; idx < BK * (CB / 4); idx += 64 * NW) {
            const int kk = idx / (CB / 4);
            const int c4 = idx - kk * (CB / 4);
            const int k = k0 + kk;
            const int c = c0 + 4 * c4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (k < L && c < d) v = *reinterpret_cast<const float4*>(Hm + (int64_t)k * ldh + c);
            float2* dst = reinterpret_cast<float2*>(&Hs[kk * LDH + 4 * c4]);
            dst[0] = make_float2(v.x, v.y);
            dst[1] = make_float2(v.z, v.w);
        }
        float a[8];
        if (TRANS) {
            // Ts[kk][qq] = T[k0+kk][r0+qq]
            for (int idx = tid; idx < BK * BM; idx += 64 * NW) {
                const int kk = idx / BM;
                const int qq = idx - kk * BM;
                const int k = k0 + kk;
                const int q = r0 + qq;
                Ts[kk * LDT + qq] = (k < L && q < L) ? T[(int64_t)k * ld + q] : 0.f;
            }
        } else {
            const int kbase = k0 + 8 * g;
            float4 v0 = make_float4(0.f, 0.f, 0.f, 0.f), v1 = v0;
            if (arow < L) {
                const float* p = T + (int64_t)arow * ld + kbase;
                if (kbase < ld) v0 = *reinterpret_cast<const float4*>(p);
                if (kbase + 4 < ld) v1 = *reinterpret_cast<const float4*>(p + 4);
            }
            a[0] = v0.x; a[1] = v0.y; a[2] = v0.z; a[3] = v0.w;
            a[4] = v1.x; a[5] = v1.y; a[6] = v1.z; a[7] = v1.w;
            // the row padding (columns L .. ld-1) is not data and may hold anything, NaN included
#pragma unroll
            for (int j = 0; j < 8; ++j) a[j] = (kbase + j < L) ? a[j] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int kk = 8 * g + j;
            const float av = TRANS ? Ts[kk * LDT + 16 * w + frow] : a[j];
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) {
                const float bv = Hs[kk * LDH + ct * 16 + frow];
                acc[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[ct], 0, 0, 0);
            }
        }
    }

    // ---- epilogue: C/D layout col = lane&15, row = 4*(lane>>4) + reg
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = r0 + 16 * w + 4 * g + r;
        if (row >= L) continue;
        const int64_t grow = rs + row;
        float cw[8];
        int cn[8];
        int nc = 0;
        for (int n = 0; n < M && nc < 8; ++n) {
            if (n == m) continue;
            const int pk = (m < n) ? mmdfn_pair_index(m, n, M) : mmdfn_pair_index(n, m, M);
            cw[nc] = cross[(int64_t)pk * N + grow];
            cn[nc] = n;
            ++nc;
        }
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) {
            const int col = c0 + ct * 16 + frow;
            if (col >= d) continue;
            float v = acc[ct][r];
            for (int e = 0; e < nc; ++e) v += cw[e] * H[((int64_t)cn[e] * N + grow) * ldh + col];
            out[((int64_t)m * N + grow) * ldo + col] = v;
        }
    }
}


// ---------------------------------------------------------------------------------------------
// v2 forward kernel: software-pipelined.
//   workgroup = (dialogue i, modality m, 16*NWR tile rows, 16*NCTW*NWC feature columns), NWR*NWC waves;
//   wave (wr, wc) owns 16 rows x NCTW column tiles.
//   * H rows stream through a DOUBLE-BUFFERED LDS ring in BKT-row chunks (one barrier per chunk); the
//     global loads of chunk c+1 are issued before the MFMAs of chunk c and land in registers meanwhile.
//   * the tile strip (A operand) never touches LDS: lane (row, g) loads T[row][k0+16h+4g .. +3] with
//     16-byte loads (the wave covers 16 rows x 64 contiguous bytes per load); the same k-permutation
//     is applied to the B fragments read from LDS (row stride = 4 mod 8 floats: conflict-free
//     ds_read_b32 and 16-byte aligned ds_write_b128).  B fragments are fetched one k-step ahead.
//   * epilogue: accumulators -> LDS -> whole rows; the cross-modal diagonal terms are added with
//     coalesced 16-byte loads of H[(n, row), :] and the result leaves as 16-byte row-contiguous stores.
//   * blockIdx -> work mapping keeps ALL work of one dialogue (every modality, row and column block)
//     on one XCD (blockIdx % 8): the H rows shared by the row blocks of a tile AND the other
//     modalities' rows read by the cross-modal terms are served by that XCD's L2.
//   * small LDS / VGPR footprint on purpose: with BKT = 16 six 4-wave workgroups fit per CU, so a
//     launch of <= 6 waves per SIMD runs in ONE round with the MFMA pipe shared by all of them.
template <int NWR, int NWC, int NCTW, int BKT, int RPW>
__global__ __launch_bounds__(64 * NWR * NWC, (RPW >= 4 ? 2 : 4)) void propagate_v2_kernel(
    const float* __restrict__ tiles, const float* __restrict__ cross, const float* __restrict__ H,
    float* __restrict__ out, const int32_t* __restrict__ dia_len, const int32_t* __restrict__ row_start,
    const int64_t* __restrict__ tile_base, int B, int M, int N, int d, int ldh, int ldo, int max_rb, int ncb, int abl_arg) {
    // RPW = 16-row tiles per wave: every B fragment read from LDS feeds RPW MFMAs, and the H rows are
    // staged once per 16*RPW*NWR output rows (L2 -> LDS traffic scales with 1 / (RPW*NWR)).
#ifndef MMDFN_TUNING
    constexpr int abl = 0;                            // production build: no ablation paths
    (void)abl_arg;
#else
    const int abl = abl_arg;
#endif
    constexpr int NT = 64 * NWR * NWC;
    constexpr int WROWS = 16 * RPW;                   // rows per wave
    constexpr int BM = WROWS * NWR;
    constexpr int CB = 16 * NCTW * NWC;
    constexpr int LDH = CB + 4;                       // = 4 (mod 8): conflict-free fragment reads
    constexpr int NSUB = BKT / 16;                    // 16-wide k sub-chunks per chunk
    constexpr int OROWS = (2 * BKT < BM) ? 2 * BKT : BM;  // output rows staged per epilogue pass (multiple of 16)
    constexpr int NH4 = (BKT * (CB / 4) + NT - 1) / NT;   // float4 staging loads per thread per chunk
    constexpr int HROWS = (NH4 * NT + CB / 4 - 1) / (CB / 4);  // rows touched by the unconditional staging stores (>= BKT)
    constexpr int HBUF = HROWS * LDH;
    __shared__ __attribute__((aligned(16))) float lds[2 * HBUF];

    // XCD-aware decode: bid % 8 == dialogue % 8
    const int Rt = max_rb * ncb;     // work items per (dialogue, modality) tile
    const int Rd = M * Rt;           // work items per dialogue
    const int bid = blockIdx.x;
    const int yq = bid >> 3;
    const int i = (yq / Rd) * 8 + (bid & 7);
    if (i >= B) return;
    const int rho = yq % Rd;
    const int m = rho / Rt;
    const int rb = (rho - m * Rt) / ncb;
    const int c0 = (rho - m * Rt - rb * ncb) * CB;
    const int L = dia_len[i];
    const int r0 = rb * BM;
    if (r0 >= L) return;
    const int ld = (L + 3) & ~3;
    const int rs = row_start[i];
    const float* T = tiles + tile_base[i] + (int64_t)m * L * ld;
    const float* Hm = H + ((int64_t)m * N + rs) * ldh;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = tid >> 6;
    const int wr = w % NWR;
    const int wc = w / NWR;
    const int frow = lane & 15;
    const int g = lane >> 4;
    const int wrow0 = r0 + WROWS * wr;                // first tile row of this wave
    // a wave whose rows or column tiles all lie outside the tile skips the MFMA loop (wave-uniform);
    // inside the loop every tile is computed unconditionally (pad rows/columns are zero) so the
    // MFMAs stay in one basic block and pipeline behind their ds_reads.
    const bool wave_active = (c0 + 16 * wc * NCTW < d) && (wrow0 < L);

    f32x4 acc[RPW][NCTW];
#pragma unroll
    for (int rp = 0; rp < RPW; ++rp)
#pragma unroll
        for (int ct = 0; ct < NCTW; ++ct) acc[rp][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // ---- loop-invariant staging slots: every load below is unconditional (clamped address) and the
    // out-of-range lanes are zeroed with selects, so a chunk's loads / LDS stores form ONE basic block
    // (predicated loads compile to a branch + s_waitcnt vmcnt(0) per load and serialise the pipeline).
    int s_kk[NH4], s_lds[NH4];
    bool s_cok[NH4];
    const float* s_ptr[NH4];
#pragma unroll
    for (int e = 0; e < NH4; ++e) {
        const int idx = tid + e * NT;
        const int kk = idx / (CB / 4);
        const int c4 = idx - kk * (CB / 4);
        const int c = c0 + 4 * c4;
        s_kk[e] = kk;
        s_lds[e] = kk * LDH + 4 * c4;
        s_cok[e] = (c < d) && !(abl & 4);
        s_ptr[e] = Hm + (c < d ? c : 0);
    }
    const float* a_ptr[RPW];
    bool a_ok[RPW];
#pragma unroll
    for (int rp = 0; rp < RPW; ++rp) {
        const int arow = wrow0 + 16 * rp + frow;
        a_ok[rp] = (arow < L) && !(abl & 8);
        a_ptr[rp] = T + (int64_t)(arow < L ? arow : L - 1) * ld;
    }
    // raw loads only: the zero-masking of out-of-range lanes is applied when the registers are CONSUMED
    // (next iteration), otherwise the selects would force a vmcnt(0) wait right behind the loads.
    // Two register sets form a ring: the loads of chunk c+2 are issued during chunk c (right after
    // its barrier) and consumed at the top of chunk c+2, i.e. two MFMA blocks later -- twice the bytes
    // in flight per wave of a distance-1 prefetch (the staging path is latency-bound: Little's law).
    // Raw loads only; the zero-masking happens when a set is consumed, otherwise the selects would
    // force a vmcnt(0) wait right behind the loads.  The loop is unrolled by two so that the ring
    // index is static (runtime-indexed register arrays would go to scratch).
    float4 hset[2][NH4] = {};
    float4 aset[2][RPW][NSUB] = {};
    const int nchunks = (L + BKT - 1) / BKT;

#define MMDFN_ISSUE(SET, K0)                                                                              \
    do {                                                                                                  \
        if (!(abl & 4)) {                                                                                 \
            _Pragma("unroll") for (int e = 0; e < NH4; ++e) {                                             \
                const int k_ = (K0) + s_kk[e];                                                            \
                hset[SET][e] = *reinterpret_cast<const float4*>(s_ptr[e] + (int64_t)(k_ < L ? k_ : L - 1) * ldh); \
            }                                                                                             \
        }                                                                                                 \
        if (!(abl & 8)) {                                                                                 \
            _Pragma("unroll") for (int rp = 0; rp < RPW; ++rp)                                            \
                _Pragma("unroll") for (int h = 0; h < NSUB; ++h) {                                        \
                    const int ka_ = (K0) + 16 * h + 4 * g;                                                \
                    aset[SET][rp][h] = *reinterpret_cast<const float4*>(a_ptr[rp] + (ka_ < ld ? ka_ : ld - 4)); \
                }                                                                                         \
        }                                                                                                 \
    } while (0)

#define MMDFN_CHUNK(SET, C)                                                                               \
    do {                                                                                                  \
        float* Hs = lds + ((C) & 1) * HBUF;                                                               \
        const int kc0 = (C) * BKT;                                                                        \
        _Pragma("unroll") for (int e = 0; e < NH4; ++e) {                                                 \
            const bool ok = s_cok[e] && (kc0 + s_kk[e] < L);                                              \
            const float4 v = hset[SET][e];                                                                \
            *reinterpret_cast<float4*>(&Hs[s_lds[e]]) =                                                   \
                make_float4(ok ? v.x : 0.f, ok ? v.y : 0.f, ok ? v.z : 0.f, ok ? v.w : 0.f);              \
        }                                                                                                 \
        float av[RPW][4 * NSUB];                                                                          \
        _Pragma("unroll") for (int rp = 0; rp < RPW; ++rp)                                                \
            _Pragma("unroll") for (int h = 0; h < NSUB; ++h) {                                            \
                /* per element: the row padding (columns L .. ld-1) is not data, it may hold NaN */      \
                const int ka_ = kc0 + 16 * h + 4 * g;                                                     \
                const float4 v = aset[SET][rp][h];                                                        \
                av[rp][4 * h + 0] = (a_ok[rp] && ka_ + 0 < L) ? v.x : 0.f;                                \
                av[rp][4 * h + 1] = (a_ok[rp] && ka_ + 1 < L) ? v.y : 0.f;                                \
                av[rp][4 * h + 2] = (a_ok[rp] && ka_ + 2 < L) ? v.z : 0.f;                                \
                av[rp][4 * h + 3] = (a_ok[rp] && ka_ + 3 < L) ? v.w : 0.f;                                \
            }                                                                                             \
        __syncthreads();                                                                                  \
        if ((C) + 2 < nchunks) MMDFN_ISSUE(SET, ((C) + 2) * BKT);                                         \
        if (wave_active && !(abl & 2)) {                                                                  \
            __builtin_amdgcn_s_setprio(3); /* MFMA phase outranks other waves' load/store phases */       \
            const float* hbase = &Hs[16 * wc * NCTW + frow];                                              \
            float bq[2][NCTW];                                                                            \
            _Pragma("unroll") for (int ct = 0; ct < NCTW; ++ct) bq[0][ct] = hbase[(4 * g) * LDH + 16 * ct]; \
            _Pragma("unroll") for (int j = 0; j < 4 * NSUB; ++j) {                                        \
                if (j + 1 < 4 * NSUB) {                                                                   \
                    const int kn = 16 * ((j + 1) >> 2) + 4 * g + ((j + 1) & 3);                           \
                    _Pragma("unroll") for (int ct = 0; ct < NCTW; ++ct)                                   \
                        bq[(j + 1) & 1][ct] = hbase[kn * LDH + 16 * ct];                                  \
                }                                                                                         \
                __builtin_amdgcn_sched_barrier(0); /* keep the B prefetch ahead of the MFMAs */           \
                _Pragma("unroll") for (int rp = 0; rp < RPW; ++rp)                                        \
                    _Pragma("unroll") for (int ct = 0; ct < NCTW; ++ct)                                   \
                        acc[rp][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[rp][j], bq[j & 1][ct],      \
                                                                           acc[rp][ct], 0, 0, 0);         \
                __builtin_amdgcn_sched_barrier(0);                                                        \
            }                                                                                             \
        }                                                                                                 \
        __builtin_amdgcn_s_setprio(0);                                                                    \
        __builtin_amdgcn_sched_barrier(0);                                                                \
    } while (0)

    MMDFN_ISSUE(0, 0);
    if (nchunks > 1) MMDFN_ISSUE(1, BKT);
    for (int c = 0; c < nchunks; c += 2) {
        MMDFN_CHUNK(0, c);
        if (c + 1 < nchunks) MMDFN_CHUNK(1, c + 1);
    }
#undef MMDFN_ISSUE
#undef MMDFN_CHUNK

    // ---- epilogue through LDS, OROWS rows per pass: Os[row][col], row stride LDH (conflict-free scatter)
    float* Os = lds;
    const int cw4 = ((d - c0 < CB) ? (d - c0) : CB) / 4;  // float4 columns of this block that exist
#pragma unroll
    for (int pass = 0; pass < BM / OROWS; ++pass) {
        __syncthreads();
#pragma unroll
        for (int rp = 0; rp < RPW; ++rp) {
            const int lrow0 = WROWS * wr + 16 * rp;       // first block-local row of this 16-row tile
            if (lrow0 >= pass * OROWS && lrow0 < (pass + 1) * OROWS) {
#pragma unroll
                for (int ct = 0; ct < NCTW; ++ct)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        Os[(lrow0 - pass * OROWS + 4 * g + r) * LDH + 16 * (wc * NCTW + ct) + frow] = acc[rp][ct][r];
            }
        }
        __syncthreads();
        for (int idx = tid; idx < OROWS * cw4; idx += NT) {
            const int rr = idx / cw4;
            const int c4 = idx - rr * cw4;
            const int row = r0 + pass * OROWS + rr;
            if (row >= L) continue;
            const int64_t grow = rs + row;
            float4 v = *reinterpret_cast<const float4*>(&Os[rr * LDH + 4 * c4]);
            const int nq = (abl & 1) ? 0 : M - 1;
#pragma unroll 5
            for (int q = 0; q < nq; ++q) {           // the M-1 other modalities; unrolled so the loads batch
                const int n = q + (q >= m ? 1 : 0);
                const int pk = (m < n) ? mmdfn_pair_index(m, n, M) : mmdfn_pair_index(n, m, M);
                const float cwt = cross[(int64_t)pk * N + grow];
                const float4 h = *reinterpret_cast<const float4*>(H + ((int64_t)n * N + grow) * ldh + c0 + 4 * c4);
                v.x = fmaf(cwt, h.x, v.x);
                v.y = fmaf(cwt, h.y, v.y);
                v.z = fmaf(cwt, h.z, v.z);
                v.w = fmaf(cwt, h.w, v.w);
            }
            *reinterpret_cast<float4*>(out + ((int64_t)m * N + grow) * ldo + c0 + 4 * c4) = v;
        }
    }
}

// Ablation / tuning switches exist only in the -DMMDFN_TUNING build (lib/libmmdfn_hip_tuning.so, used by tools/):
// the production library reads no environment variable and the ablation tests fold away at compile time.
#ifdef MMDFN_TUNING
int ablation() {
    const char* e = getenv("MMDFN_PROP_ABL");
    return e ? atoi(e) : 0;
}
#else
constexpr int ablation() { return 0; }
#endif

template <int NWR, int NWC, int NCTW, int BKT, int RPW>
int launch_v2(const float* tiles, const float* cross, const float* H, float* out, const int32_t* dia_len,
              const int32_t* row_start, const int64_t* tile_base, int B, int M, int N, int d, int ldh, int ldo,
              int max_len, hipStream_t s) {
    const int BM = 16 * NWR * RPW;
    const int CB = 16 * NCTW * NWC;
    const int max_rb = (max_len + BM - 1) / BM;
    const int ncb = (d + CB - 1) / CB;
    dim3 grid(((B + 7) / 8) * 8 * M * max_rb * ncb);
    dim3 block(64 * NWR * NWC);
    hipLaunchKernelGGL((propagate_v2_kernel<NWR, NWC, NCTW, BKT, RPW>), grid, block, 0, s, tiles, cross, H, out, dia_len,
                       row_start, tile_base, B, M, N, d, ldh, ldo, max_rb, ncb, ablation());
    MMDFN_CHECK_LAUNCH();
    return 0;
}

#ifdef MMDFN_TUNING
int tuning_override() {
    const char* e = getenv("MMDFN_PROP_CFG");  // tools/tune_propagate.py
    return e ? atoi(e) : -1;
}
#else
constexpr int tuning_override() { return -1; }
#endif

template <int NW, int NCT>
int launch(const float* tiles, const float* cross, const float* H, float* out, const int32_t* dia_len,
           const int32_t* row_start, const int64_t* tile_base, int B, int M, int N, int d, int ldh, int ldo,
           int max_len, int transpose, hipStream_t s) {
    const int BM = 16 * NW;
    const int max_rb = (max_len + BM - 1) / BM;
    dim3 grid(B * max_rb, M, (d + 16 * NCT - 1) / (16 * NCT));
    dim3 block(64 * NW);
    if (transpose)
        hipLaunchKernelGGL((propagate_kernel<NW, NCT, true>), grid, block, 0, s, tiles, cross, H, out, dia_len,
                           row_start, tile_base, M, N, d, ldh, ldo, max_rb);
    else
        hipLaunchKernelGGL((propagate_kernel<NW, NCT, false>), grid, block, 0, s, tiles, cross, H, out, dia_len,
                           row_start, tile_base, M, N, d, ldh, ldo, max_rb);
    MMDFN_CHECK_LAUNCH();
    return 0;
}

}  // namespace

#define V2(NWR, NWC, NCTW, BKT, RPW) \
    launch_v2<NWR, NWC, NCTW, BKT, RPW>(tiles, cross, H, out, dia_len, row_start, tile_base, B, M, N, d, ldh, ldo, max_len, s)

int mmdfn_launch_propagate(const float* tiles, const float* cross, const float* H, float* out,
                           const int32_t* dia_len, const int32_t* row_start, const int64_t* tile_base,
                           int B, int M, int N, int d, int ldh, int ldo, int max_len, int transpose, hipStream_t s) {
    if (B <= 0 || M <= 0 || M > 9 || N <= 0 || d <= 0 || (d & 3) || max_len <= 0) return -1;
    if (ldh < d || ldo < d || (ldh & 3) || (ldo & 3)) return -1;
    if (transpose) {
        const bool small_rows = max_len <= 48;
        if (d <= 112)
            return small_rows ? launch<2, 7>(tiles, cross, H, out, dia_len, row_start, tile_base, B, M, N, d, ldh, ldo, max_len, 1, s)
                              : launch<4, 7>(tiles, cross, H, out, dia_len, row_start, tile_base, B, M, N, d, ldh, ldo, max_len, 1, s);
        if (d <= 208)
            return small_rows ? launch<2, 13>(tiles, cross, H, out, dia_len, row_start, tile_base, B, M, N, d, ldh, ldo, max_len, 1, s)
                              : launch<4, 13>(tiles, cross, H, out, dia_len, row_start, tile_base, B, M, N, d, ldh, ldo, max_len, 1, s);
        return small_rows ? launch<2, 8>(tiles, cross, H, out, dia_len, row_start, tile_base, B, M, N, d, ldh, ldo, max_len, 1, s)
                          : launch<4, 8>(tiles, cross, H, out, dia_len, row_start, tile_base, B, M, N, d, ldh, ldo, max_len, 1, s);
    }
    // 0-7: f32-MFMA tilings, 8: bf16-piece kernel, 9: never a bf16-piece kernel, 10: producer / consumer bf16-piece kernel
    const int ov = tuning_override();
    if (ov >= 0) {
        switch (ov) {
            case 0: return V2(4, 1, 7, 16, 1);
            case 1: return V2(8, 1, 7, 16, 1);
            case 2: return V2(2, 4, 2, 16, 1);
            case 3: return V2(2, 4, 2, 32, 1);
            case 4: return V2(2, 2, 4, 16, 1);
            case 5: return V2(4, 2, 4, 16, 1);
            case 6: return V2(4, 1, 7, 16, 2);
            case 7: return V2(4, 2, 7, 16, 1);
            case 8: {
                const int rc = mmdfn_launch_propagate_split(tiles, cross, H, out, dia_len, row_start, tile_base, B, M, N, d,
                                                            ldh, ldo, max_len, s);
                if (rc != -2) return rc;
                break;
            }
            default: break;
        }
    }
    if (ov == 9) {
        if (d <= 112) return ((long)B * M * max_len <= 32768L) ? V2(2, 4, 2, 16, 1) : V2(8, 1, 7, 16, 1);
        if (d <= 224) return V2(4, 2, 7, 16, 1);
        return V2(4, 2, 4, 16, 1);
    }
    // measured on MI355X (profiles/r01_propagate_tuning.md): many small workgroups with the columns split
    // over 4 waves win while the launch is latency-bound; 8-wave row blocks win once the H-row staging
    // traffic (one pass over the H tile per row block) dominates.
    const long approx_rows = (long)B * M * max_len;
    const long split_wgs = (long)B * M * ((max_len + 127) / 128) * ((d + 127) / 128);
    if (max_len >= 128 && split_wgs >= 48) {
        // dialogues of >= 128 utterances: the exact-f32 MFMA rate bounds the kernel; carry the product on bf16 MFMAs
        // (three exact bf16 pieces per operand, fp32-level error; propagate_split.hip).  Measured: L=512 M=6 B=32
        // 124 -> 86 us, B=8 47 -> 32 us, d=512 152 -> 100 us; L=128..384 M=3 10-25 % faster
        // (profiles/r01_propagate_tuning.md)
        const int rc = mmdfn_launch_propagate_split(tiles, cross, H, out, dia_len, row_start, tile_base, B, M, N, d, ldh,
                                                    ldo, max_len, s);
        if (rc != -2) return rc;
    }
    if (d <= 112) {
        if (approx_rows <= 32768L) return V2(2, 4, 2, 16, 1);
        return V2(8, 1, 7, 16, 1);
    }
    if (d <= 224) return V2(4, 2, 7, 16, 1);
    return V2(4, 2, 4, 16, 1);
}
#undef V2

extern "C" int mmdfn_abi_version(void) { return 17; }

extern "C" int mmdfn_propagate(const float* tiles, const float* cross, const float* H, float* out,
                               const int32_t* dia_len, const int32_t* row_start, const int64_t* tile_base,
                               int B, int M, int N, int d, int ldh, int ldo, int max_len, int transpose,
                               void* stream) {
    return mmdfn_launch_propagate(tiles, cross, H, out, dia_len, row_start, tile_base, B, M, N, d, ldh, ldo, max_len,
                                  transpose, (hipStream_t)stream);
}
