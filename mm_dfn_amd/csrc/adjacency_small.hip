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

// Reductions on the DPP path (hipcc turns __shfl_xor into ds_bpermute_b32: an LDS round trip per step, and a workgroup
// here is one serial chain).  After the two quad steps every quad holds its sum in all four lanes, so the mirror steps are
// exchanges between equal halves: all 16 lanes of a row end with the same bits.
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float sum16(float v) {
    v += dpp_mov<0xB1>(v);          // quad_perm [1,0,3,2]
    v += dpp_mov<0x4E>(v);          // quad_perm [2,3,0,1]
    v += dpp_mov<0x141>(v);         // row_half_mirror
    v += dpp_mov<0x140>(v);         // row_mirror
    return v;
}
__device__ __forceinline__ float lane_value(float v, int l) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l));
}
// sum over the 32 lanes of each half of the wave (lanes 0-31 get the lower half's sum, 32-63 the upper's)
__device__ __forceinline__ float sum32(float v) {
    v = sum16(v);
    const float lo = lane_value(v, 0) + lane_value(v, 16);
    const float hi = lane_value(v, 32) + lane_value(v, 48);
    return (threadIdx.x & 32) ? hi : lo;
}
__device__ __forceinline__ float sum64(float v) {
    v = sum16(v);
    return (lane_value(v, 0) + lane_value(v, 16)) + (lane_value(v, 32) + lane_value(v, 48));
}
// x / d with 1 / d at hand: quotient estimate + one residual correction (the core of the division expansion without its
// range scaling: operands here are feature values and their norm)
__device__ __forceinline__ float div_by(float x, float d, float inv) {
    const float q = x * inv;
    return fmaf(fmaf(-q, d, x), inv, q);
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
    float4 xm[4][4];
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int p = 32 * it + 4 * w + g;
        const int64_t grow = rs + (p < L ? p : L - 1);
#pragma unroll
        for (int sidx = 0; sidx < 4; ++sidx) {
            const int k = 64 * sidx + 4 * fi;
            xm[it][sidx] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (32 * it < Lp && k < D) xm[it][sidx] = *reinterpret_cast<const float4*>(feats + ((int64_t)m * N + grow) * D + k);
        }
    }
    float4 xo[RI][2][4];
#pragma unroll
    for (int j = 0; j < RI; ++j) {
        const int p = r0 + 32 * j + 4 * w + g;
        const int64_t grow = rs + (p < L ? p : L - 1);
#pragma unroll
        for (int oi = 0; oi < 2; ++oi) {
            const int o = oi == 0 ? o0 : o1;
#pragma unroll
            for (int sidx = 0; sidx < 4; ++sidx) {
                const int k = 64 * sidx + 4 * fi;
                xo[j][oi][sidx] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (o >= 0 && r0 + 32 * j < Lp && k < D)
                    xo[j][oi][sidx] = *reinterpret_cast<const float4*>(feats + ((int64_t)o * N + grow) * D + k);
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
    float bfr[2][32];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int col = 16 * (w + KS_NW * h) + fi;
#pragma unroll
        for (int kk = 0; kk < 32; ++kk) {
            const int q = 16 * (kk >> 2) + 4 * g + (kk & 3);
            bfr[h][kk] = 0.f;
            if (w + KS_NW * h < nct && q < L && col < D) bfr[h][kk] = unit[((int64_t)m * N + rs + q) * D + col];
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
            cg[j][e] = (p < L && q < L) ? cosg[toff_m + (int64_t)p * ld + q] : 0.f;
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

    // ---- stage 1: Z = dT o T of every modality: row sums, column sums; this modality's strip rows / strip columns of dT -> LDS
    {
        const int c4 = lane & 31, sub = lane >> 5;
        const int q = 4 * c4;
        for (int n = 0; n < M; ++n) {
            const int64_t toff = tile_base[i] + (int64_t)n * L * ld;
            float4 dt[8], tt[8];
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int p = 16 * it + 2 * w + sub;
                dt[it] = make_float4(0.f, 0.f, 0.f, 0.f);
                tt[it] = dt[it];
                if (p < L && q < ld) {
                    dt[it] = *reinterpret_cast<const float4*>(dtiles + toff + (int64_t)p * ld + q);
                    tt[it] = *reinterpret_cast<const float4*>(tiles + toff + (int64_t)p * ld + q);
                }
            }
            float4 cacc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                if (16 * it >= Lp) continue;
                const int p = 16 * it + 2 * w + sub;
                float4 d4 = dt[it];
                if (q + 0 >= L) d4.x = 0.f;
                if (q + 1 >= L) d4.y = 0.f;
                if (q + 2 >= L) d4.z = 0.f;
                if (q + 3 >= L) d4.w = 0.f;
                const float4 z = make_float4(d4.x * tt[it].x, d4.y * tt[it].y, d4.z * tt[it].z, d4.w * tt[it].w);
                const float zr = sum32((z.x + z.y) + (z.z + z.w));
                if (c4 == 0) zrow[n * KS_MAXL + p] = zr;
                cacc.x += z.x; cacc.y += z.y; cacc.z += z.z; cacc.w += z.w;
                if (n == m) {
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
        const int64_t grow = rs + p;
        float zz = zrow[n * KS_MAXL + p];
        float zc = 0.f;
        for (int ww = 0; ww < 2 * KS_NW; ++ww) zc += colp[(n * 16 + ww) * KS_MAXL + p];
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
                Rw[pl * KS_SE + q] = (wv * rp * rm[q] + ddp + dm[q]) * mmdfn_dsim(cg[j][e]);
            }
        }
        {
            const int n = tid / SR, pl = tid - n * SR;       // (3 SR <= 512 threads)
            const int p = r0 + pl;
            if (n < M && n != m && p < L)
                ec[n * SR + pl] = (pre_dc * rm[p] * rl[n * KS_MAXL + p] + dm[p] + dd[n * KS_MAXL + p]) * modal_weight * mmdfn_dsim(pre_cd);
        }
    }
    // ---- requested now, used after the MFMAs: the unit rows / addend rows of the epilogue (lane = four columns)
    int o0 = -1, o1 = -1;
    for (int n = 0; n < M; ++n)
        if (n != m) { if (o0 < 0) o0 = n; else o1 = n; }
    float4 eu[RJ], e0[RJ], e1[RJ], ea[RJ];
    float einv[RJ];
    const int k4 = 4 * lane;
#pragma unroll
    for (int j = 0; j < RJ; ++j) {
        const int p = r0 + w + KS_NW * j;
        const bool ok = p < L && k4 < D;
        const int64_t grow = rs + (p < L ? p : 0);
        const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
        eu[j] = ok ? *reinterpret_cast<const float4*>(unit + ((int64_t)m * N + grow) * D + k4) : z4;
        e0[j] = (ok && o0 >= 0) ? *reinterpret_cast<const float4*>(unit + ((int64_t)o0 * N + grow) * D + k4) : z4;
        e1[j] = (ok && o1 >= 0) ? *reinterpret_cast<const float4*>(unit + ((int64_t)o1 * N + grow) * D + k4) : z4;
        ea[j] = (ok && addend) ? *reinterpret_cast<const float4*>(addend + ((int64_t)m * N + grow) * D + k4) : z4;
        einv[j] = norm[(int64_t)m * N + grow];               // (||x||: the division waits for the load, so it is done in the epilogue)
    }
    __syncthreads();
    KS_STOP(3);

    // ---- stage 4: d(unit) strip = E strip . U on exact-f32 MFMAs -> LDS
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int ct = w + KS_NW * h;
        if (ct >= nct) continue;
#pragma unroll
        for (int a = 0; a < RT; ++a) {
            if (r0 + 16 * a >= Lp) continue;
            f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
            const float* ea_ = Rw + (16 * a + fi) * KS_SE + 4 * g;
#pragma unroll
            for (int kc = 0; kc < 8; ++kc) {
                if (kc >= nt) continue;
                const float4 av = *reinterpret_cast<const float4*>(ea_ + 16 * kc);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, bfr[h][4 * kc + 0], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, bfr[h][4 * kc + 1], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, bfr[h][4 * kc + 2], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, bfr[h][4 * kc + 3], acc, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) dUl[(16 * a + 4 * g + r) * SD + 16 * ct + fi] = acc[r];
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
        if (o0 >= 0) { const float c = ec[o0 * SR + pl]; du.x += c * e0[j].x; du.y += c * e0[j].y; du.z += c * e0[j].z; du.w += c * e0[j].w; }
        if (o1 >= 0) { const float c = ec[o1 * SR + pl]; du.x += c * e1[j].x; du.y += c * e1[j].y; du.z += c * e1[j].z; du.w += c * e1[j].w; }
        const float s = sum64(dot4(eu[j], du));
        if (ok) {
            const float inv = 1.0f / einv[j];
            float4 v;
            v.x = (du.x - eu[j].x * s) * inv + ea[j].x;
            v.y = (du.y - eu[j].y * s) * inv + ea[j].y;
            v.z = (du.z - eu[j].z * s) * inv + ea[j].z;
            v.w = (du.w - eu[j].w * s) * inv + ea[j].w;
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
