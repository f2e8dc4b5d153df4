// K5s: the adjacency build of SHORT dialogues (L <= 128, M <= 3, D <= 256: IEMOCAP / MELD sizes) in few batches, one
// workgroup per (dialogue, modality, strip of SR rows): forward in TWO launches, backward in ONE.
//
// Replaces MM_GCN.create_big_adj (reference model_mm.py:122-180) and its autograd graph for the sizes the reference's
// datasets have.  adjacency.hip runs the same arithmetic as four + five launches whose grids cover all dialogues; at
// cfg2 (16 dialogues x 110 utterances) each of those launches holds a few microseconds of work and pays a launch, a
// ramp, a memory round trip per dependent stage and a tail (33 + 39 us per step against 15 us of floor).
// Here a strip of a (dialogue, modality) tile never leaves its compute unit between the stages; what a stage needs from
// OTHER strips it recomputes (the unit rows of the whole dialogue; d(degree) of every row), because that is cheaper than
// a launch boundary:
//
//   forward  : strip_fwd  -> unit rows of the dialogue (one modality) -> LDS; the strip's cross-modal cosines and degree
//                            seed; cosine Gram strip U_strip . U^T on exact-f32 MFMAs; angular similarity; row degrees
//                            through a fixed-order partial table (bit-reproducible); degree^-1/2; r_p S[p,q] from
//                            the accumulators
//              finish     -> T[p,q] = (r_p S[p,q]) r_q and the cross diagonals (they need the degrees of other strips /
//                            modalities: the one launch boundary of the forward pass)
//   backward : strip_bwd  -> d(degree) of EVERY modality's rows of the dialogue (row + column sums of dT o T: no
//                            transposed read, no acos -- S r_q = T / r_p); the strip of W = dT + dT^T from the same
//                            loads; E = (W r_p r_q + dd_p + dd_q) sim'(G) in LDS; d(unit) = E . U + the cross diagonals
//                            on exact-f32 MFMAs (a wave owns 16-column tiles of U, whose B fragments it requested
//                            before the first stage); (du - u (u.du)) / ||x|| + addend by whole rows.
// Every batch of global loads of a stage is requested before the first one is consumed (a workgroup is one serial
// chain: a dependent load costs it a full memory round trip).
#include "mmdfn_internal.h"
#include "../../include/mmdfn_hip.h"
#include <stdlib.h>

namespace {

constexpr int KS_MAXL = 128;                // rows of a tile this form covers
constexpr int KS_MMAX = 3;
constexpr int KS_NW = 8;                    // waves per workgroup
constexpr int KS_SE = 132;                  // row stride (floats) of the strips in LDS: 16-byte rows, 33 quads (odd)

#ifdef MMDFN_TUNING
#define KS_STOP(K) do { if (stop == (K)) return; } while (0)
#else
#define KS_STOP(K) do { } while (0)
#endif

__device__ __forceinline__ float sum16(float v) { return row_sum16(v); }        // (DPP reductions: mmdfn_internal.h)
__device__ __forceinline__ float sum32(float v) { return half_sum32(v); }
__device__ __forceinline__ float sum64(float v) { return wave_sum(v); }
// x / d with 1 / d at hand: quotient estimate + one residual correction (the core of the division expansion without its
// range scaling: operands here are feature values and their norm)
__device__ __forceinline__ float div_by(float x, float d, float inv) {
    const float q = x * inv;
    return fmaf(fmaf(-q, d, x), inv, q);
}
// d sim / d c = a / (pi sqrt(1 - (a c)^2)) on the hardware reciprocal square root (1 ulp; mmdfn_dsim's correctly rounded
// sqrt + division are ~60 instructions per element of a workgroup that is one serial chain)
__device__ __forceinline__ float ks_dsim(float c) {
    const float ac = c * MMDFN_COS_SHRINK;
    return (MMDFN_COS_SHRINK / MMDFN_PI_F) * __builtin_amdgcn_rsqf(1.0f - ac * ac);
}
__device__ __forceinline__ float dot4(const float4& a, const float4& b) { return (a.x * b.x + a.y * b.y) + (a.z * b.z + a.w * b.w); }

// blockIdx -> (dialogue, modality, strip): blockIdx % 8 == dialogue % 8 (everything of a dialogue shares one XCD's L2)
__device__ __forceinline__ bool ks_decode(int B, int M, int NS, int& i, int& m, int& st) {
    const int bid = blockIdx.x;
    const int yq = bid >> 3;
    const int per = M * NS;
    i = (yq / per) * 8 + (bid & 7);
    const int rem = yq % per;
    m = rem / NS;
    st = rem - m * NS;
    return i < B;
}

// ---------------------------------------------------------------------------------------------------------------------------
// forward, stage A.  LDS: U[lmax_p][SU] | part[8][SR] | seed[SR] | rr[SR]
template <int SR>
__global__ __launch_bounds__(64 * KS_NW) void adj_strip_fwd_kernel(
    const float* __restrict__ feats, float* __restrict__ unit, float* __restrict__ norm, float* __restrict__ cosg,
    float* __restrict__ cdot, float* __restrict__ rdeg, float* __restrict__ tiles, float* __restrict__ cross,
    const int32_t* __restrict__ dia_len, const int32_t* __restrict__ row_start, const int64_t* __restrict__ tile_base,
    int B, int M, int N, int D, int SU, int lmax_p, int NS, float modal_weight, int stop) {
    constexpr int RT = SR / 16;                          // 16-row tiles of a strip
    constexpr int RI = SR / 32;                          // 32-row groups of a strip
    extern __shared__ __attribute__((aligned(16))) float smem[];
    int i, m, st;
    if (!ks_decode(B, M, NS, i, m, st)) return;
    const int L = dia_len[i];
    const int r0 = st * SR;
    if (r0 >= L) return;
    KS_STOP(9);
    const int ld = (L + 3) & ~3;
    const int Lp = (L + 15) & ~15;
    const int nt = Lp >> 4;
    const int rs = row_start[i];
    const int64_t toff = tile_base[i] + (int64_t)m * L * ld;
    const int KP = (D + 15) & ~15;

    float* U = smem;
    float* part = smem + (size_t)lmax_p * SU;
    float* seed = part + 8 * SR;
    float* rr = seed + SR;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fi = lane & 15, g = lane >> 4;

    int o0 = -1, o1 = -1;                                // the other modalities
    for (int n = 0; n < M; ++n)
        if (n != m) { if (o0 < 0) o0 = n; else o1 = n; }

    // ---- every load of the stage first: this modality's rows of the whole dialogue (a 16-lane group owns a row, lane j
    // holds elements 64 s + 4 j .. + 3), the other modalities' rows of the strip
    // (one lane-masked region per 64-wide k slice; rows past the dialogue are clamped and zeroed / skipped where they are used)
    float4 xm[4][4];
    float4 xo[RI][2][4];
    {
        const float* fm = feats + ((int64_t)m * N + rs) * D;
        const float* f0 = feats + ((int64_t)(o0 >= 0 ? o0 : m) * N + rs) * D;
        const float* f1 = feats + ((int64_t)(o1 >= 0 ? o1 : m) * N + rs) * D;
        unsigned offm[4], offo[RI];
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int p = 32 * it + 4 * w + g;
            offm[it] = (unsigned)(p < L ? p : L - 1) * (unsigned)D + 4 * fi;
        }
#pragma unroll
        for (int j = 0; j < RI; ++j) {
            const int p = r0 + 32 * j + 4 * w + g;
            offo[j] = (unsigned)(p < L ? p : L - 1) * (unsigned)D + 4 * fi;
        }
#pragma unroll
        for (int sidx = 0; sidx < 4; ++sidx) {
#pragma unroll
            for (int it = 0; it < 4; ++it) xm[it][sidx] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int j = 0; j < RI; ++j) { xo[j][0][sidx] = make_float4(0.f, 0.f, 0.f, 0.f); xo[j][1][sidx] = xo[j][0][sidx]; }
            if (64 * sidx + 4 * fi < D) {
#pragma unroll
                for (int it = 0; it < 4; ++it)
                    if (32 * it < Lp) xm[it][sidx] = *reinterpret_cast<const float4*>(fm + offm[it] + 64 * sidx);
#pragma unroll
                for (int j = 0; j < RI; ++j) {
                    if (r0 + 32 * j >= Lp) continue;
                    if (o0 >= 0) xo[j][0][sidx] = *reinterpret_cast<const float4*>(f0 + offo[j] + 64 * sidx);
                    if (o1 >= 0) xo[j][1][sidx] = *reinterpret_cast<const float4*>(f1 + offo[j] + 64 * sidx);
                }
            }
        }
    }
    // ---- unit rows of the dialogue -> LDS; the strip's own rows -> unit / norm
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        if (32 * it >= Lp) continue;
        const int p = 32 * it + 4 * w + g;
        const bool rok = p < L;
        float ss = 0.f;
#pragma unroll
        for (int sidx = 0; sidx < 4; ++sidx) ss += dot4(xm[it][sidx], xm[it][sidx]);
        ss = sum16(ss);
        const float nv = sqrtf(ss);
        const float inv = 1.0f / nv;
        const bool mine = rok && p >= r0 && p < r0 + SR;
#pragma unroll
        for (int sidx = 0; sidx < 4; ++sidx) {
            const int k = 64 * sidx + 4 * fi;
            float4 u = xm[it][sidx];
            u.x = div_by(u.x, nv, inv); u.y = div_by(u.y, nv, inv); u.z = div_by(u.z, nv, inv); u.w = div_by(u.w, nv, inv);
            if (k < KP && p < Lp) *reinterpret_cast<float4*>(U + (size_t)p * SU + k) = rok ? u : make_float4(0.f, 0.f, 0.f, 0.f);
            if (mine && k < D) *reinterpret_cast<float4*>(unit + ((int64_t)m * N + rs + p) * D + k) = u;
        }
        if (mine && fi == 0) norm[(int64_t)m * N + rs + p] = nv;
    }
    __syncthreads();
    KS_STOP(1);

    // ---- the strip's cross-modal cosines (pairs with this modality) and degree seed
#pragma unroll
    for (int j = 0; j < RI; ++j) {
        if (r0 + 32 * j >= Lp) continue;
        const int pl = 32 * j + 4 * w + g;
        const int p = r0 + pl;
        const bool rok = p < L;
        float4 um[4];
#pragma unroll
        for (int sidx = 0; sidx < 4; ++sidx) {
            const int k = 64 * sidx + 4 * fi;
            um[sidx] = k < KP ? *reinterpret_cast<const float4*>(U + (size_t)(p < Lp ? p : 0) * SU + k) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        float dsum = 0.f;
#pragma unroll
        for (int oi = 0; oi < 2; ++oi) {
            const int o = oi == 0 ? o0 : o1;
            if (o < 0) continue;
            float ss = 0.f;
#pragma unroll
            for (int sidx = 0; sidx < 4; ++sidx) ss += dot4(xo[j][oi][sidx], xo[j][oi][sidx]);
            ss = sum16(ss);
            const float nv = sqrtf(ss);
            const float inv = 1.0f / nv;
            float sdot = 0.f;
#pragma unroll
            for (int sidx = 0; sidx < 4; ++sidx) {
                float4 u = xo[j][oi][sidx];
                u.x = div_by(u.x, nv, inv); u.y = div_by(u.y, nv, inv); u.z = div_by(u.z, nv, inv); u.w = div_by(u.w, nv, inv);
                sdot += dot4(um[sidx], u);
            }
            sdot = sum16(sdot);
            const float c = mmdfn_sim(sdot) * modal_weight;
            dsum += c;
            if (m < o && fi == 0 && rok) {               // every pair has one writer: the workgroup of its lower modality
                const int64_t oo = (int64_t)mmdfn_pair_index(m, o, M) * N + rs + p;
                cdot[oo] = sdot;
                cross[oo] = c;                           // raw; scaled by adj_finish_kernel once both degrees exist
            }
        }
        if (fi == 0) seed[pl] = dsum;
    }
    KS_STOP(2);

    // ---- Gram strip: wave w owns tile column b = w, all RT row tiles of the strip
    float sv[RT][4];
    const int b = w;
    if (b < nt) {
        f32x4 acc[RT];
#pragma unroll
        for (int a = 0; a < RT; ++a) acc[a] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const float* ub = U + (size_t)(16 * b + fi) * SU + 4 * g;
        const float* ua = U + (size_t)(r0 + fi) * SU + 4 * g;
        // fragments of the next 16-wide k group are read while this one's MFMAs run
        float4 bn = *reinterpret_cast<const float4*>(ub);
        float4 an[RT];
#pragma unroll
        for (int a = 0; a < RT; ++a) an[a] = *reinterpret_cast<const float4*>(ua + (r0 + 16 * a < Lp ? (size_t)16 * a * SU : 0));
        for (int kc = 0; kc < KP; kc += 16) {
            const float4 bv = bn;
            float4 av[RT];
#pragma unroll
            for (int a = 0; a < RT; ++a) av[a] = an[a];
            const int kn = kc + 16 < KP ? kc + 16 : kc;
            bn = *reinterpret_cast<const float4*>(ub + kn);
#pragma unroll
            for (int a = 0; a < RT; ++a) an[a] = *reinterpret_cast<const float4*>(ua + (r0 + 16 * a < Lp ? (size_t)16 * a * SU : 0) + kn);
#pragma unroll
            for (int a = 0; a < RT; ++a) {
                if (r0 + 16 * a >= Lp) continue;
                acc[a] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[a].x, bv.x, acc[a], 0, 0, 0);
                acc[a] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[a].y, bv.y, acc[a], 0, 0, 0);
                acc[a] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[a].z, bv.z, acc[a], 0, 0, 0);
                acc[a] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[a].w, bv.w, acc[a], 0, 0, 0);
            }
        }
        // C layout: acc[a][r] = G[p = r0 + 16 a + 4 g + r][q = 16 b + fi]
        const int q = 16 * b + fi;
#pragma unroll
        for (int a = 0; a < RT; ++a) {
            if (r0 + 16 * a >= Lp) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int pl = 16 * a + 4 * g + r;
                const int p = r0 + pl;
                const bool in = (p < L) && (q < L);
                if (p < L && q < ld) cosg[toff + (int64_t)p * ld + q] = in ? acc[a][r] : 0.f;     // raw cosine (saved for backward)
                sv[a][r] = in ? mmdfn_sim(acc[a][r]) : 0.f;
                const float rsum = sum16(sv[a][r]);
                if (fi == 0) part[b * SR + pl] = rsum;
            }
        }
    }
    __syncthreads();
    KS_STOP(3);

    // ---- degrees: seed + the partial sums in a fixed order
    if (tid < SR && r0 + tid < L) {
        float deg = seed[tid];
        for (int y = 0; y < nt; ++y) deg += part[y * SR + tid];
        const float r = powf(deg, -0.5f);
        rr[tid] = r;
        rdeg[(int64_t)m * N + rs + r0 + tid] = r;
    }
    __syncthreads();

    // ---- r_p S[p,q]  (adj_finish_kernel multiplies by r_q)
    if (b < nt) {
        const int q = 16 * b + fi;
#pragma unroll
        for (int a = 0; a < RT; ++a) {
            if (r0 + 16 * a >= Lp) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int pl = 16 * a + 4 * g + r;
                const int p = r0 + pl;
                if (p < L && q < ld) tiles[toff + (int64_t)p * ld + q] = (q < L) ? rr[pl] * sv[a][r] : 0.f;
            }
        }
    }
}

// forward, stage B: T[p,q] = (r_p S[p,q]) r_q -- one wave per tile row; the blocks behind the tile rows (y = 0) do the
// cross diagonals: cross[k][r] = (r_m cross_raw[k][r]) r_n
__global__ __launch_bounds__(256) void adj_finish_kernel(float* __restrict__ tiles, const float* __restrict__ rdeg,
                                                         float* __restrict__ cross, const int32_t* __restrict__ dia_len,
                                                         const int32_t* __restrict__ row_start,
                                                         const int64_t* __restrict__ tile_base, int M, int N, int max_len,
                                                         int tile_blocks) {
    if ((int)blockIdx.x >= tile_blocks) {
        const int row = ((int)blockIdx.x - tile_blocks) * 256 + threadIdx.x;
        if (blockIdx.y != 0 || row >= N) return;
        float r[KS_MMAX];
#pragma unroll
        for (int m = 0; m < KS_MMAX; ++m) r[m] = m < M ? rdeg[(int64_t)m * N + row] : 0.f;
#pragma unroll
        for (int m = 0; m < KS_MMAX; ++m)
#pragma unroll
            for (int n = m + 1; n < KS_MMAX; ++n) {
                if (n >= M) continue;
                const int64_t o = (int64_t)mmdfn_pair_index(m, n, M) * N + row;
                cross[o] = (r[m] * cross[o]) * r[n];
            }
        return;
    }
    const int rowblocks = (max_len + 3) / 4;
    const int i = blockIdx.x / rowblocks;
    const int p = (blockIdx.x % rowblocks) * 4 + (threadIdx.x >> 6);
    const int m = blockIdx.y;
    const int L = dia_len[i];
    if (p >= L) return;
    const int lane = threadIdx.x & 63;
    const int ld = (L + 3) & ~3;
    const int rs = row_start[i];
    float* t = tiles + tile_base[i] + (int64_t)m * L * ld + (int64_t)p * ld;
    const float* r = rdeg + (int64_t)m * N + rs;
    const int q0 = lane, q1 = lane + 64;
    float t0 = 0.f, t1 = 0.f, ra = 0.f, rb = 0.f;
    if (q0 < L) { t0 = t[q0]; ra = r[q0]; }
    if (q1 < L) { t1 = t[q1]; rb = r[q1]; }
    if (q0 < L) t[q0] = t0 * ra;
    if (q1 < L) t[q1] = t1 * rb;
}

// ---------------------------------------------------------------------------------------------------------------------------
// backward.  LDS: Rw[SR][KS_SE] | Ct[SR][KS_SE] | colp[3][16][128] | zrow[3][128] | dd[3][128] | rl[3][128] | ec[3][SR] | dUl[SR][SD]
template <int SR>
__global__ __launch_bounds__(64 * KS_NW) void adj_strip_bwd_kernel(
    const float* __restrict__ dtiles, const float* __restrict__ dcross, const float* __restrict__ unit,
    const float* __restrict__ norm, const float* __restrict__ cosg, const float* __restrict__ cdot,
    const float* __restrict__ rdeg, const float* __restrict__ tiles, const float* __restrict__ cross,
    const float* __restrict__ addend, float* __restrict__ dfeats, const int32_t* __restrict__ dia_len,
    const int32_t* __restrict__ row_start, const int64_t* __restrict__ tile_base, int B, int M, int N, int D, int SD,
    int NS, float modal_weight, int stop) {
    constexpr int RT = SR / 16;
    constexpr int RJ = SR / KS_NW;                       // strip rows per wave in the row-per-wave stages
    extern __shared__ __attribute__((aligned(16))) float smem[];
    int i, m, st;
    if (!ks_decode(B, M, NS, i, m, st)) return;
    const int L = dia_len[i];
    const int r0 = st * SR;
    if (r0 >= L) return;
    KS_STOP(9);
    const int ld = (L + 3) & ~3;
    const int Lp = (L + 15) & ~15;
    const int nt = Lp >> 4;
    const int rs = row_start[i];
    const int nct = (D + 15) >> 4;                      // 16-column tiles of U (<= 16)
    const int64_t toff_m = tile_base[i] + (int64_t)m * L * ld;

    float* Rw = smem;
    float* Ct = Rw + SR * KS_SE;
    float* colp = Ct + SR * KS_SE;                       // [3][16][128]: column sums per (wave, half-wave)
    float* zrow = colp + 3 * 16 * KS_MAXL;
    float* dd = zrow + 3 * KS_MAXL;
    float* rl = dd + 3 * KS_MAXL;
    float* ec = rl + 3 * KS_MAXL;
    float* dUl = ec + 3 * SR;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fi = lane & 15, g = lane >> 4;

    // ---- requested now, used last: the B fragments of d(unit) = E . U for this wave's column tiles (ct = w, w + 8).
    // MFMA step j of the 16-wide k group kc contracts q = 16 kc + 4 g + j (the A side reads E[row][16 kc + 4 g .. + 3]).
    // Rows q >= L and columns >= D are clamped, not masked: they meet E columns that are zero / feed outputs nobody stores.
    float bfr[2][32];
    {
        const float* um = unit + ((int64_t)m * N + rs) * D;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
            for (int kk = 0; kk < 32; ++kk) bfr[h][kk] = 0.f;
            if (w + KS_NW * h >= nct) continue;
            const int col = 16 * (w + KS_NW * h) + fi;
            const unsigned colc = col < D ? col : D - 1;
#pragma unroll
            for (int kk = 0; kk < 32; ++kk) {
                if (16 * (kk >> 2) >= Lp) continue;                    // (uniform)
                const int q = 16 * (kk >> 2) + 4 * g + (kk & 3);
                const unsigned qc = q < L ? q : L - 1;
                bfr[h][kk] = um[qc * (unsigned)D + colc];
            }
        }
    }
    // ---- the saved cosines of the strip (stage 3)
    float cg[RJ][2];
#pragma unroll
    for (int j = 0; j < RJ; ++j) {
        const int p = r0 + w + KS_NW * j;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int q = lane + 64 * e;
            cg[j][e] = cosg[toff_m + (unsigned)((p < L ? p : L - 1) * ld + (q < L ? q : L - 1))];      // (clamped; used for p, q < L only)
        }
    }

    // ---- the cross-diagonal operands of stage 3 (thread e < 3 SR: modality e / SR, strip row e % SR)
    float pre_dc = 0.f, pre_cd = 0.f;
    {
        const int n = tid / SR, pl = tid - n * SR;
        const int p = r0 + pl;
        if (n < M && n != m && p < L) {
            const int pk = (m < n) ? mmdfn_pair_index(m, n, M) : mmdfn_pair_index(n, m, M);
            const int64_t o = (int64_t)pk * N + rs + p;
            pre_dc = dcross[o];
            pre_cd = cdot[o];
        }
    }

    // ---- stage 1: Z = dT o T: row sums, column sums (this modality: the whole tile; the others: what the strip's rows need);
    // this modality's strip rows / strip columns of dT -> LDS
    {
        const int c4 = lane & 31, sub = lane >> 5;
        const int q = 4 * c4;
        for (int n = 0; n < M; ++n) {
            const int64_t toff = tile_base[i] + (int64_t)n * L * ld;
            if (n != m) {
                // another modality: only d(degree) of the STRIP's rows is needed (the cross diagonals): row sums of the strip's
                // rows, column sums of the strip's columns (CQ lanes per row, RPW rows per wave and request)
                constexpr int CQ = SR / 4, RPW = 64 / CQ, NI = KS_MAXL / (KS_NW * RPW);
                const float* dtn = dtiles + toff;
                const float* ttn = tiles + toff;
                const unsigned qc = q < ld ? q : ld - 4;
                float4 dr[RT], tr[RT], dc[NI], tc[NI];
#pragma unroll
                for (int j = 0; j < RT; ++j) {
                    const int p = r0 + 16 * j + 2 * w + sub;
                    const unsigned off = (unsigned)(p < L ? p : L - 1) * (unsigned)ld + qc;
                    dr[j] = *reinterpret_cast<const float4*>(dtn + off);
                    tr[j] = *reinterpret_cast<const float4*>(ttn + off);
                }
                const int cq = lane % CQ, rq = lane / CQ;
                const int colq = r0 + 4 * cq;
#pragma unroll
                for (int it = 0; it < NI; ++it) {
                    const int p = KS_NW * RPW * it + RPW * w + rq;
                    const unsigned off = (unsigned)(p < L ? p : L - 1) * (unsigned)ld + (unsigned)(colq < ld ? colq : ld - 4);
                    dc[it] = *reinterpret_cast<const float4*>(dtn + off);
                    tc[it] = *reinterpret_cast<const float4*>(ttn + off);
                }
#pragma unroll
                for (int j = 0; j < RT; ++j) {
                    const int p = r0 + 16 * j + 2 * w + sub;
                    const bool rowin = p < L;
                    float zs = 0.f;
                    zs += (rowin && q + 0 < L) ? dr[j].x * tr[j].x : 0.f;
                    zs += (rowin && q + 1 < L) ? dr[j].y * tr[j].y : 0.f;
                    zs += (rowin && q + 2 < L) ? dr[j].z * tr[j].z : 0.f;
                    zs += (rowin && q + 3 < L) ? dr[j].w * tr[j].w : 0.f;
                    const float zr = sum32(zs);
                    if (c4 == 0 && p < KS_MAXL) zrow[n * KS_MAXL + p] = zr;
                }
                float4 cacc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int it = 0; it < NI; ++it) {
                    const int p = KS_NW * RPW * it + RPW * w + rq;
                    const bool rowin = p < L;
                    cacc.x += (rowin && colq + 0 < L) ? dc[it].x * tc[it].x : 0.f;
                    cacc.y += (rowin && colq + 1 < L) ? dc[it].y * tc[it].y : 0.f;
                    cacc.z += (rowin && colq + 2 < L) ? dc[it].z * tc[it].z : 0.f;
                    cacc.w += (rowin && colq + 3 < L) ? dc[it].w * tc[it].w : 0.f;
                }
                // colp slot of modality n: [KS_NW * RPW partial sums][SR strip columns]
                *reinterpret_cast<float4*>(colp + n * 16 * KS_MAXL + (RPW * w + rq) * SR + 4 * cq) = cacc;
                continue;
            }
            float4 dt[8], tt[8];
            const float* dtn = dtiles + toff;
            const float* ttn = tiles + toff;
            const unsigned qc = q < ld ? q : ld - 4;                  // (lanes past the row: clamped, zeroed below)
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int p = 16 * it + 2 * w + sub;
                dt[it] = make_float4(0.f, 0.f, 0.f, 0.f);
                tt[it] = dt[it];
                if (16 * it < L) {                                     // (uniform; rows past the tile: clamped, zeroed below)
                    const unsigned off = (unsigned)(p < L ? p : L - 1) * (unsigned)ld + qc;
                    dt[it] = *reinterpret_cast<const float4*>(dtn + off);
                    tt[it] = *reinterpret_cast<const float4*>(ttn + off);
                }
            }
            float4 cacc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                if (16 * it >= Lp) continue;
                const int p = 16 * it + 2 * w + sub;
                float4 d4 = dt[it];
                const bool rowin = p < L;
                d4.x = (rowin && q + 0 < L) ? d4.x : 0.f;
                d4.y = (rowin && q + 1 < L) ? d4.y : 0.f;
                d4.z = (rowin && q + 2 < L) ? d4.z : 0.f;
                d4.w = (rowin && q + 3 < L) ? d4.w : 0.f;
                const float4 z = make_float4(d4.x * tt[it].x, d4.y * tt[it].y, d4.z * tt[it].z, d4.w * tt[it].w);
                const float zr = sum32((z.x + z.y) + (z.z + z.w));
                if (c4 == 0) zrow[n * KS_MAXL + p] = zr;
                cacc.x += z.x; cacc.y += z.y; cacc.z += z.z; cacc.w += z.w;
                {
                    const int pl = p - r0;
                    if (pl >= 0 && pl < SR) *reinterpret_cast<float4*>(Rw + pl * KS_SE + q) = d4;
                    const int cl = q - r0;                       // (r0, q multiples of 4: the four columns are in or out together)
                    if (cl >= 0 && cl < SR) {
                        Ct[(cl + 0) * KS_SE + p] = d4.x;
                        Ct[(cl + 1) * KS_SE + p] = d4.y;
                        Ct[(cl + 2) * KS_SE + p] = d4.z;
                        Ct[(cl + 3) * KS_SE + p] = d4.w;
                    }
                }
            }
            *reinterpret_cast<float4*>(colp + (n * 16 + 2 * w + sub) * KS_MAXL + q) = cacc;
        }
    }
    __syncthreads();
    KS_STOP(1);

    // ---- stage 2: d(degree) of every modality's rows of this dialogue:  dd = -1/2 r^2 (Z row + Z column + cross part)
    for (int e = tid; e < M * KS_MAXL; e += 64 * KS_NW) {
        const int n = e >> 7, p = e & 127;
        if (p >= L) continue;
        if (n != m && (p < r0 || p >= r0 + SR)) continue;       // (other modalities: the strip's rows only)
        const int64_t grow = rs + p;
        float zz = zrow[n * KS_MAXL + p];
        float zc = 0.f;
        if (n == m) {
            for (int ww = 0; ww < 2 * KS_NW; ++ww) zc += colp[(n * 16 + ww) * KS_MAXL + p];
        } else {
            constexpr int NP = KS_NW * (64 / (SR / 4));
            for (int ww = 0; ww < NP; ++ww) zc += colp[n * 16 * KS_MAXL + ww * SR + (p - r0)];
        }
        zz += zc;
        for (int k = 0; k < M; ++k) {
            if (k == n) continue;
            const int pk = (n < k) ? mmdfn_pair_index(n, k, M) : mmdfn_pair_index(k, n, M);
            zz += dcross[(int64_t)pk * N + grow] * cross[(int64_t)pk * N + grow];
        }
        const float r = rdeg[(int64_t)n * N + grow];
        rl[n * KS_MAXL + p] = r;
        dd[n * KS_MAXL + p] = -0.5f * r * r * zz;
    }
    __syncthreads();
    KS_STOP(2);

    // ---- stage 3: the strip of E = (W r_p r_q + dd_p + dd_q) sim'(G), in place of the strip rows; cross diagonals
    {
        const float* rm = rl + m * KS_MAXL;
        const float* dm = dd + m * KS_MAXL;
#pragma unroll
        for (int j = 0; j < RJ; ++j) {
            const int pl = w + KS_NW * j;
            const int p = r0 + pl;
            if (p >= L) continue;
            const float rp = rm[p], ddp = dm[p];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int q = lane + 64 * e;
                if (q >= L) continue;
                const float wv = Rw[pl * KS_SE + q] + Ct[pl * KS_SE + q];
                Rw[pl * KS_SE + q] = (wv * rp * rm[q] + ddp + dm[q]) * ks_dsim(cg[j][e]);
            }
        }
        {
            const int n = tid / SR, pl = tid - n * SR;       // (3 SR <= 512 threads)
            const int p = r0 + pl;
            if (n < M && n != m && p < L)
                ec[n * SR + pl] = (pre_dc * rm[p] * rl[n * KS_MAXL + p] + dm[p] + dd[n * KS_MAXL + p]) * modal_weight * ks_dsim(pre_cd);
        }
    }
    // ---- requested now, used after the MFMAs: the unit rows / addend rows of the epilogue (lane = four columns)
    int o0 = -1, o1 = -1;
    for (int n = 0; n < M; ++n)
        if (n != m) { if (o0 < 0) o0 = n; else o1 = n; }
    float4 eu[RJ], e0[RJ], e1[RJ], ea[RJ];
    float einv[RJ];
    const int k4 = 4 * lane;
    {
        // (rows / columns past the end are clamped, not masked: their results are not stored; absent operands read `unit`
        // and are not used -- every request stays a plain global load)
        const unsigned kc4 = k4 < D ? k4 : D - 4;
        const float* pu = unit + ((int64_t)m * N + rs) * D;
        const float* p0 = unit + ((int64_t)(o0 >= 0 ? o0 : m) * N + rs) * D;
        const float* p1 = unit + ((int64_t)(o1 >= 0 ? o1 : m) * N + rs) * D;
        const float* pa = addend ? addend + ((int64_t)m * N + rs) * D : pu;
        const float* pn = norm + (int64_t)m * N + rs;
#pragma unroll
        for (int j = 0; j < RJ; ++j) {
            const int p = r0 + w + KS_NW * j;
            const unsigned off = (unsigned)(p < L ? p : L - 1) * (unsigned)D + kc4;
            eu[j] = *reinterpret_cast<const float4*>(pu + off);
            e0[j] = *reinterpret_cast<const float4*>(p0 + off);
            e1[j] = *reinterpret_cast<const float4*>(p1 + off);
            ea[j] = *reinterpret_cast<const float4*>(pa + off);
            einv[j] = pn[p < L ? p : L - 1];                 // (||x||: the division waits for the load, so it is done in the epilogue)
        }
    }
    __syncthreads();
    KS_STOP(3);

    // ---- stage 4: d(unit) strip = E strip . U on exact-f32 MFMAs -> LDS
    // per row tile: all eight A fragments are read first (columns past the tile hold zeros), then the two column tiles'
    // accumulator chains run interleaved
    if (w < nct) {
        const bool two = w + KS_NW < nct;
#pragma unroll
        for (int a = 0; a < RT; ++a) {
            if (r0 + 16 * a >= Lp) continue;
            const float* ea_ = Rw + (16 * a + fi) * KS_SE + 4 * g;
            float4 av[8];
#pragma unroll
            for (int kc = 0; kc < 8; ++kc) av[kc] = *reinterpret_cast<const float4*>(ea_ + 16 * kc);
            f32x4 acc0 = (f32x4){0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
#pragma unroll
            for (int kc = 0; kc < 8; ++kc) {
                if (kc >= nt) continue;
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[kc].x, bfr[0][4 * kc + 0], acc0, 0, 0, 0);
                if (two) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[kc].x, bfr[1][4 * kc + 0], acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[kc].y, bfr[0][4 * kc + 1], acc0, 0, 0, 0);
                if (two) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[kc].y, bfr[1][4 * kc + 1], acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[kc].z, bfr[0][4 * kc + 2], acc0, 0, 0, 0);
                if (two) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[kc].z, bfr[1][4 * kc + 2], acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[kc].w, bfr[0][4 * kc + 3], acc0, 0, 0, 0);
                if (two) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[kc].w, bfr[1][4 * kc + 3], acc1, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                dUl[(16 * a + 4 * g + r) * SD + 16 * w + fi] = acc0[r];
                if (two) dUl[(16 * a + 4 * g + r) * SD + 16 * (w + KS_NW) + fi] = acc1[r];
            }
        }
    }
    __syncthreads();
    KS_STOP(4);

    // ---- dX = (du - u (u.du)) / ||x||  (+ addend), one wave per strip row
#pragma unroll
    for (int j = 0; j < RJ; ++j) {
        const int pl = w + KS_NW * j;
        const int p = r0 + pl;
        if (p >= L) continue;
        const bool ok = k4 < D;
        float4 du = ok ? *reinterpret_cast<const float4*>(dUl + pl * SD + k4) : make_float4(0.f, 0.f, 0.f, 0.f);
        const float4 uu = ok ? eu[j] : make_float4(0.f, 0.f, 0.f, 0.f);
        const float4 ad = addend ? ea[j] : make_float4(0.f, 0.f, 0.f, 0.f);
        if (o0 >= 0) { const float c = ec[o0 * SR + pl]; du.x += c * e0[j].x; du.y += c * e0[j].y; du.z += c * e0[j].z; du.w += c * e0[j].w; }
        if (o1 >= 0) { const float c = ec[o1 * SR + pl]; du.x += c * e1[j].x; du.y += c * e1[j].y; du.z += c * e1[j].z; du.w += c * e1[j].w; }
        const float s = sum64(dot4(uu, du));
        if (ok) {
            const float inv = 1.0f / einv[j];
            float4 v;
            v.x = (du.x - uu.x * s) * inv + ad.x;
            v.y = (du.y - uu.y * s) * inv + ad.y;
            v.z = (du.z - uu.z * s) * inv + ad.z;
            v.w = (du.w - uu.w * s) * inv + ad.w;
            *reinterpret_cast<float4*>(dfeats + ((int64_t)m * N + rs + p) * D + k4) = v;
        }
    }
}

struct KsPlan {
    int SR, NS, SU, SD, lmax_p, grid;
    size_t lds_fwd, lds_bwd;
};

// the form is chosen from the SHAPE alone (B, M, D, max_len): forward and backward of one tensor set agree
bool ks_plan(int B, int M, int D, int max_len, KsPlan* pl) {
    if (M > KS_MMAX || max_len > KS_MAXL || D > 256 || (D & 3)) return false;
    int force_sr = 0;
#ifdef MMDFN_TUNING
    if (const char* e = getenv("MMDFN_ADJ_SMALL")) if (atoi(e) == 0) return false;
    if (const char* e = getenv("MMDFN_ADJ_SR")) force_sr = atoi(e);
#endif
    const int bg = ((B + 7) / 8) * 8 * M;
    int SR = (bg * ((max_len + 31) / 32) <= 304) ? 32 : 64;
    if (force_sr == 32 || force_sr == 64) SR = force_sr;
    const int NS = (max_len + SR - 1) / SR;
    if (!force_sr && bg * NS > 1024) return false;       // many dialogues: the many-launch form fills the chip by itself
    const int KP = (D + 15) & ~15;
    int SU = KP + 4;
    if (((SU >> 2) & 1) == 0) SU += 4;                   // rows of an odd number of 16-byte units: conflict-free fragment reads
    pl->SR = SR;
    pl->NS = NS;
    pl->SU = SU;
    pl->SD = KP + 4;
    pl->lmax_p = (max_len + 15) & ~15;
    pl->grid = bg * NS;
    pl->lds_fwd = ((size_t)pl->lmax_p * SU + 8 * SR + 2 * SR) * sizeof(float);
    pl->lds_bwd = ((size_t)2 * SR * KS_SE + 3 * 16 * KS_MAXL + 3 * 3 * KS_MAXL + 3 * SR + (size_t)SR * pl->SD) * sizeof(float);
    return pl->lds_fwd <= 156 * 1024 && pl->lds_bwd <= 156 * 1024;
}

int ks_stop() {
#ifdef MMDFN_TUNING
    if (const char* e = getenv("MMDFN_ADJ_STOP")) return atoi(e);
#endif
    return 0;
}

}  // namespace

// -2: shape not covered by this form (the caller runs the many-launch form of adjacency.hip)
int mmdfn_launch_adj_small_fwd(const float* feats, float* unit, float* norm, float* cosg, float* cdot, float* rdeg,
                               float* tiles, float* cross, const int32_t* dia_len, const int32_t* row_start,
                               const int64_t* tile_base, int B, int M, int N, int D, int max_len, float modal_weight,
                               hipStream_t s) {
    KsPlan pl;
    if (!ks_plan(B, M, D, max_len, &pl)) return -2;
#define KS_FWD(SRV)                                                                                                          \
    do {                                                                                                                     \
        if (pl.lds_fwd > 64 * 1024 && mmdfn_allow_big_lds(adj_strip_fwd_kernel<SRV>)) return -2;                            \
        hipLaunchKernelGGL(adj_strip_fwd_kernel<SRV>, dim3(pl.grid), dim3(64 * KS_NW), pl.lds_fwd, s, feats, unit, norm, cosg, \
                           cdot, rdeg, tiles, cross, dia_len, row_start, tile_base, B, M, N, D, pl.SU, pl.lmax_p, pl.NS,     \
                           modal_weight, ks_stop());                                                                         \
    } while (0)
    if (pl.SR == 32) KS_FWD(32); else KS_FWD(64);
#undef KS_FWD
    MMDFN_CHECK_LAUNCH();
    const int rowblocks = (max_len + 3) / 4;
    const int tile_blocks = B * rowblocks;
    hipLaunchKernelGGL(adj_finish_kernel, dim3(tile_blocks + (M > 1 ? (N + 255) / 256 : 0), M), dim3(256), 0, s, tiles, rdeg,
                       cross, dia_len, row_start, tile_base, M, N, max_len, tile_blocks);
    MMDFN_CHECK_LAUNCH();
    return 0;
}

int mmdfn_launch_adj_small_bwd(const float* dtiles, const float* dcross, const float* unit, const float* norm,
                               const float* cosg, const float* cdot, const float* rdeg, const float* tiles,
                               const float* cross, const float* addend, float* dfeats, const int32_t* dia_len,
                               const int32_t* row_start, const int64_t* tile_base, int B, int M, int N, int D, int max_len,
                               float modal_weight, hipStream_t s) {
    KsPlan pl;
    if (!ks_plan(B, M, D, max_len, &pl)) return -2;
#define KS_BWD(SRV)                                                                                                          \
    do {                                                                                                                     \
        if (pl.lds_bwd > 64 * 1024 && mmdfn_allow_big_lds(adj_strip_bwd_kernel<SRV>)) return -2;                            \
        hipLaunchKernelGGL(adj_strip_bwd_kernel<SRV>, dim3(pl.grid), dim3(64 * KS_NW), pl.lds_bwd, s, dtiles, dcross, unit,   \
                           norm, cosg, cdot, rdeg, tiles, cross, addend, dfeats, dia_len, row_start, tile_base, B, M, N, D,  \
                           pl.SD, pl.NS, modal_weight, ks_stop());                                                           \
    } while (0)
    if (pl.SR == 32) KS_BWD(32); else KS_BWD(64);
#undef KS_BWD
    MMDFN_CHECK_LAUNCH();
    return 0;
}
