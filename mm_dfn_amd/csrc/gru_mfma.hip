// GRU recurrence for launches with MANY sequences (BASELINE cfg3: 1 900 sequence-directions of <= 33 steps): the
// recurrent products of SIXTEEN sequences per workgroup on the matrix pipe.
//
// Replaces the time loop of nn.GRU (reference model.py:866,868,1076-1087: 3 P separate party-GRU calls plus the context
// GRUs) like gru.hip, whose kernels give every sequence-direction a workgroup of its own: there the step is a dependent
// chain of packed FMAs on the weights held in registers, ~0.65-1.2 us per step whatever the batch, and a launch lasts as long
// as ceil(sequences / 512) rounds of T steps.  At cfg3 that is 3-4 rounds.  Here a workgroup owns 16 rows of one (group,
// direction):   gh^T (300 x 16) = W_hh (300 x 100) . h_{t-1}^T (100 x 16)   per step as v_mfma_f32_16x16x32_bf16 with every
// fp32 operand cut exactly into three bf16 pieces and the six piece products of weight >= 2^-16 (fp32-level error, see
// propagate_split.hip): the launch lasts T steps of ~2.4 us for up to 16 x 256 sequences.
//   forward : wave w (7 waves) owns hidden units 16 w .. 16 w + 15: its three gate tiles (r, z, n) x 4 K-steps of W_hh pieces
//             stay in registers (144 VGPRs) as the MFMA's A operand (rows = units); the B operand (k x 16 sequences) is
//             h_{t-1} as three bf16 planes [sequence][unit] in LDS, written by the lanes that produce h_t: a lane's D fragment
//             holds FOUR consecutive units of one sequence -- its r, z, n pre-activations sit in the same lane (gate math
//             without any exchange), h_t leaves as one 8-byte LDS write per piece and the five saved tensors (y, r, z, n,
//             W_hn h + b_hn) as 16-byte global stores nobody waits for.  One barrier per step, planes double-buffered.
//   backward: dh_{t+1}'s recurrent term  rec^T (100 x 16) = W_hh^T (100 x 300) . dgh^T (300 x 16): wave w owns units 16 w ..,
//             one tile x 10 K-steps of W_hh^T pieces (120 VGPRs); dgh (r, z, n pre-activation gradients) of the step just done
//             lives in LDS as three bf16 planes [sequence][gate row].  Same lane ownership, same barrier.
// The operands of a step (gi; dy, the saved gates, h_prev) reach the recurrence waves through LDS slots filled by an eighth
// wave that does nothing but load (see the kernels).  Used by mmdfn_gru_seq_fwd / _bwd when a launch holds more than
// mfma_min_chains() = 1 024 sequence-directions (gru.hip); H = 100 like every kernel of this path.  Measurements and the tuning
// log: profiles/r05_gru_mfma_form.md.
#include "mmdfn_internal.h"
#include "gemm_tn_split_body.h"
#include "keep_flags_body.h"
#include <stdlib.h>

namespace {

constexpr int GH = 100;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
#define LDS_AS(T, p) ((__attribute__((address_space(3))) T*)(p))

constexpr int MS = 16;                  // sequences per workgroup
constexpr int NWV = 7;                  // waves: 7 x 16 = 112 >= 100 hidden units
constexpr int NTH = 64 * NWV;
constexpr int HROWB = 272;              // bytes per sequence row of an h plane: 128 bf16 + 16 (68 dwords = 4 mod 64: a quarter
                                        // wave's 16-byte reads of 16 rows at one offset cover all 64 banks)
constexpr int HPLANE = MS * HROWB;      // 4 352
constexpr int DROWB = 656;              // bytes per sequence row of a dgh plane: 320 bf16 + 16 (164 dwords = 4 x 9 mod 64)
constexpr int DPLANE = MS * DROWB;      // 10 496

__device__ __forceinline__ float as_f(uint32_t u) { return __builtin_bit_cast(float, u); }
__device__ __forceinline__ uint32_t as_u(float f) { return __builtin_bit_cast(uint32_t, f); }

// four values -> their three bf16 pieces (8 + 8 + 8 significant bits, cut by truncation, each exactly representable)
__device__ __forceinline__ void cut4(float x0, float x1, float x2, float x3, u32x2& p1, u32x2& p2, u32x2& p3) {
    const uint32_t hm = 0xffff0000u;
    p1 = u32x2{__builtin_amdgcn_perm(as_u(x1), as_u(x0), 0x07060302u), __builtin_amdgcn_perm(as_u(x3), as_u(x2), 0x07060302u)};
    x0 -= as_f(as_u(x0) & hm); x1 -= as_f(as_u(x1) & hm); x2 -= as_f(as_u(x2) & hm); x3 -= as_f(as_u(x3) & hm);
    p2 = u32x2{__builtin_amdgcn_perm(as_u(x1), as_u(x0), 0x07060302u), __builtin_amdgcn_perm(as_u(x3), as_u(x2), 0x07060302u)};
    x0 -= as_f(as_u(x0) & hm); x1 -= as_f(as_u(x1) & hm); x2 -= as_f(as_u(x2) & hm); x3 -= as_f(as_u(x3) & hm);
    p3 = u32x2{__builtin_amdgcn_perm(as_u(x1), as_u(x0), 0x07060302u), __builtin_amdgcn_perm(as_u(x3), as_u(x2), 0x07060302u)};
}

__device__ __forceinline__ f32x4 mfma16(u32x4 a, u32x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

__device__ __forceinline__ float sigmoidf_(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float tanhf_(float x) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(2.0f * x)); }

// The per-step barrier: the planes written this step must be complete (lgkmcnt), the step's GLOBAL stores need not be --
// __syncthreads() also waits for vmcnt(0), i.e. for a store round trip per step (0.6 us of the first version's 2.5 us step).
__device__ __forceinline__ void step_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// the six piece products of one K = 32 step, smallest first
__device__ __forceinline__ f32x4 six(const u32x4 (&a)[3], const u32x4 (&b)[3], f32x4 c) {
    c = mfma16(a[2], b[0], c);
    c = mfma16(a[1], b[0], c);
    c = mfma16(a[1], b[1], c);
    c = mfma16(a[0], b[2], c);
    c = mfma16(a[0], b[1], c);
    c = mfma16(a[0], b[0], c);
    return c;
}

constexpr int MAXG = 8;
struct MfFwd {
    int n;
    const float* gi[MAXG];
    const float* w_hh[2 * MAXG];
    const float* b_hh[2 * MAXG];
    float* y[MAXG];
    float* gates[MAXG];
    int rows[MAXG], T[MAXG], slice0[MAXG + 1];
    int abl;        // tuning build (MMDFN_GRU_MF_ABL): 1 no operand loads in the loop, 2 no result stores, 4 no MFMAs (timing only)
};
struct MfBwd {
    int n;
    const float* dy[MAXG];
    const float* y[MAXG];
    const float* gates[MAXG];
    const float* w_hh[2 * MAXG];
    float* dgi[MAXG];
    float* dgh[MAXG];
    int rows[MAXG], T[MAXG], slice0[MAXG + 1];
    int abl;
};

// eight consecutive k of one weight row -> the three pieces of an A fragment register set
__device__ __forceinline__ void cut8(const float (&v)[8], u32x4& p1, u32x4& p2, u32x4& p3) {
    u32x2 a1, a2, a3, b1, b2, b3;
    cut4(v[0], v[1], v[2], v[3], a1, a2, a3);
    cut4(v[4], v[5], v[6], v[7], b1, b2, b3);
    p1 = u32x4{a1.x, a1.y, b1.x, b1.y};
    p2 = u32x4{a2.x, a2.y, b2.x, b2.y};
    p3 = u32x4{a3.x, a3.y, b3.x, b3.y};
}

// LDS staging of the steps' operands (filled by the I/O waves): one slot per step parity
constexpr int GROW = 3 * GH;                    // forward: gi of one sequence (r | z | n), 300 floats = 44 (mod 64) dwords apart
constexpr int GSLOT = MS * GROW * 4;            // 19 200 B
constexpr int BROW = 6 * GH + 4;                // backward: dy | r z n ghn | h_prev + pad, 604 floats = 28 (mod 64) dwords apart
constexpr int BSLOT = MS * BROW * 4;            // 38 656 B
constexpr int FWD_THREADS = NTH + 64;           // 7 recurrence waves + 1 I/O wave
constexpr int BWD_THREADS = NTH + 64;           // 7 recurrence waves + 1 I/O wave (a ninth wave would cap the kernel at 168 VGPRs)
constexpr int FWD_LDS = 2 * 3 * HPLANE + 2 * GSLOT;      // 64 512 B
constexpr int BWD_LDS = 2 * 3 * DPLANE + 2 * BSLOT;      // 140 288 B

// Forward.  Loads and stores of one wave retire out of order with each other, so a wave that does both can only wait for
// "everything" (vmcnt counts both): the recurrence waves therefore ONLY store (results, never waited for) and a separate wave
// only loads: the gi rows of step k + 3 are requested at step k into one of two register sets and dropped into the LDS slot of
// their step parity at step k + 2 (two steps of cover; its waits count loads only).
// (the body of workgroup (bx, by) = (slot of 16 sequences, direction); gm_smem: FWD_LDS bytes of LDS)
__device__ __forceinline__ void gru_fwd_mfma_body(const MfFwd& G, const int bx, const int by, unsigned char* gm_smem) {
    unsigned char* hp = gm_smem;
    float* gs = reinterpret_cast<float*>(gm_smem + 2 * 3 * HPLANE);
    int gidx = 0;
    while (gidx + 1 < G.n && bx >= G.slice0[gidx + 1]) ++gidx;
    const int dir = by;
    const int rows = G.rows[gidx], T = G.T[gidx];
    const int row0 = (bx - G.slice0[gidx]) * MS;
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef MMDFN_TUNING
    const int abl = G.abl;
#else
    constexpr int abl = 0;
#endif
    for (int i = tid; i < 2 * 3 * HPLANE / 16; i += FWD_THREADS) reinterpret_cast<float4*>(hp)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    const int64_t gstep = (int64_t)rows * (6 * GH);

    if (w == NWV) {
        // ---------------- the I/O wave ----------------
        constexpr int NLD = (MS * GROW / 4 + 63) / 64;      // 19 float4 per lane and step
        const float* __restrict__ gi = G.gi[gidx] + dir * 3 * GH;
        int64_t off[NLD];
        const bool live_last = lane + 64 * (NLD - 1) < MS * GROW / 4;      // (1 200 float4 per step: the 19th load of lanes 48.. is idle)
#pragma unroll
        for (int e = 0; e < NLD; ++e) {
            const int idx = lane + 64 * e;
            const int sq = idx / (GROW / 4), c4 = idx - sq * (GROW / 4);
            const int rw = row0 + sq < rows ? row0 + sq : rows - 1;
            off[e] = (idx < MS * GROW / 4) ? (int64_t)rw * (6 * GH) + 4 * c4 : 0;
        }
        f32x4 qa[NLD], qb[NLD];
        f32x4* slot0 = reinterpret_cast<f32x4*>(gs);
        f32x4* slot1 = reinterpret_cast<f32x4*>(gs + GSLOT / 4);
#define GM_ISSUE(SIDX, Q)                                                                                            \
        do {                                                                                                         \
            const int sc_ = (SIDX) < T ? (SIDX) : T - 1;                                                             \
            const float* base_ = gi + (dir ? T - 1 - sc_ : sc_) * gstep;                                             \
            _Pragma("unroll") for (int e = 0; e < NLD; ++e) Q[e] = *reinterpret_cast<const f32x4*>(base_ + off[e]);  \
        } while (0)
#define GM_DROP(DST, Q)                                                                                              \
        do {                                                                                                         \
            _Pragma("unroll") for (int e = 0; e < NLD; ++e)                                                          \
                if (e < NLD - 1 || live_last) DST[lane + 64 * e] = Q[e];                                             \
        } while (0)
        GM_ISSUE(0, qa);
        GM_DROP(slot0, qa);
        GM_ISSUE(1, qb);
        GM_ISSUE(2, qa);
        __syncthreads();
        int sidx = 0;
#pragma unroll 1
        for (; sidx + 1 < T; sidx += 2) {
            GM_DROP(slot1, qb);            // step sidx + 1
            GM_ISSUE(sidx + 3, qb);
            step_barrier();
            GM_DROP(slot0, qa);            // step sidx + 2
            GM_ISSUE(sidx + 4, qa);
            step_barrier();
        }
        if (sidx < T) step_barrier();
#undef GM_ISSUE
#undef GM_DROP
        return;
    }

    const int s = lane & 15, g4 = lane >> 4;
    const int u0 = 16 * w + 4 * g4;                  // D fragment: units u0 .. u0 + 3 of sequence s
    const bool uok = u0 < GH;
    const int u0c = uok ? u0 : 0;
    const int row = row0 + s;
    const bool rok = row < rows;
    const float* __restrict__ w_hh = G.w_hh[2 * gidx + dir];
    const float* __restrict__ b_hh = G.b_hh[2 * gidx + dir];

    // A fragments: lane (s, g4) holds unit row 16 w + s, k = 32 ks + 8 g4 .. + 7
    u32x4 wf[3][4][3];
    {
        const int ua = 16 * w + s;
#pragma unroll
        for (int gate = 0; gate < 3; ++gate)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                float v[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int k = 32 * ks + 8 * g4 + e;
                    v[e] = (ua < GH && k < GH) ? w_hh[(int64_t)(gate * GH + ua) * GH + k] : 0.f;
                }
                cut8(v, wf[gate][ks][0], wf[gate][ks][1], wf[gate][ks][2]);
            }
    }
    float bhr[4], bhz[4], bhn[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        bhr[r] = b_hh[u0c + r]; bhz[r] = b_hh[GH + u0c + r]; bhn[r] = b_hh[2 * GH + u0c + r];
    }
    const uint32_t lds0 = (uint32_t)(uintptr_t)((__attribute__((address_space(3))) void*)hp);
    const uint32_t rd = lds0 + s * HROWB + 16 * g4;              // + 64 ks + plane
    const uint32_t wr = lds0 + s * HROWB + 2 * u0c;
    const float* gsl = gs + s * GROW + u0c;                      // this lane's gi quad inside a slot (+ gate * GH)

    float* __restrict__ y = G.y[gidx] + (int64_t)row * (2 * GH) + dir * GH + u0c;
    float* __restrict__ gates = G.gates[gidx] + ((int64_t)row * 2 + dir) * (4 * GH) + u0c;
    const int64_t ystep = (int64_t)rows * (2 * GH), sstep = (int64_t)rows * (8 * GH);

    float hprev[4] = {0.f, 0.f, 0.f, 0.f};
    __syncthreads();
#pragma unroll 1
    for (int sidx = 0; sidx < T; ++sidx) {
        const int t = dir ? T - 1 - sidx : sidx;
        const int cur = sidx & 1;
        // (two accumulators per gate: six independent chains of dependent MFMAs instead of three)
        f32x4 ar = {0.f, 0.f, 0.f, 0.f}, az = ar, an = ar, ar2 = ar, az2 = ar, an2 = ar;
        const uint32_t rb = rd + cur * 3 * HPLANE;
        u32x4 b[4][3];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int q = 0; q < 3; ++q) b[ks][q] = *LDS_AS(u32x4, (uintptr_t)(rb + q * HPLANE + 64 * ks));
        const float* gq = gsl + cur * (GSLOT / 4);
        const float4 gr = *reinterpret_cast<const float4*>(gq), gz = *reinterpret_cast<const float4*>(gq + GH),
                     gn = *reinterpret_cast<const float4*>(gq + 2 * GH);
#pragma unroll
        for (int ks = 0; ks < 4; ks += 2) {
            if (abl & 4) { ar[0] += as_f(b[ks][0].x ^ wf[0][ks][0].x ^ b[ks + 1][1].y); az[0] += as_f(b[ks][1].x ^ wf[1][ks][1].y); an[0] += as_f(b[ks + 1][2].x ^ wf[2][ks][2].z); continue; }
            ar = six(wf[0][ks], b[ks], ar);
            az = six(wf[1][ks], b[ks], az);
            an = six(wf[2][ks], b[ks], an);
            ar2 = six(wf[0][ks + 1], b[ks + 1], ar2);
            az2 = six(wf[1][ks + 1], b[ks + 1], az2);
            an2 = six(wf[2][ks + 1], b[ks + 1], an2);
        }
        ar += ar2; az += az2; an += an2;
        const float grv[4] = {gr.x, gr.y, gr.z, gr.w}, gzv[4] = {gz.x, gz.y, gz.z, gz.w}, gnv[4] = {gn.x, gn.y, gn.z, gn.w};
        float hn[4], rr[4], zz[4], nn[4], ghn[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            ghn[r] = an[r] + bhn[r];
            rr[r] = sigmoidf_(grv[r] + ar[r] + bhr[r]);
            zz[r] = sigmoidf_(gzv[r] + az[r] + bhz[r]);
            nn[r] = tanhf_(gnv[r] + rr[r] * ghn[r]);
            hn[r] = (1.0f - zz[r]) * nn[r] + zz[r] * hprev[r];
            hprev[r] = hn[r];
        }
        if (uok) {
            u32x2 p1, p2, p3;
            cut4(hn[0], hn[1], hn[2], hn[3], p1, p2, p3);
            const uint32_t wb = wr + (cur ^ 1) * 3 * HPLANE;
            *LDS_AS(u32x2, (uintptr_t)wb) = p1;
            *LDS_AS(u32x2, (uintptr_t)(wb + HPLANE)) = p2;
            *LDS_AS(u32x2, (uintptr_t)(wb + 2 * HPLANE)) = p3;
        }
        step_barrier();
        // results: behind the barrier (nobody waits for them: these waves issue no loads)
        if (uok && rok && !(abl & 2)) {
            *reinterpret_cast<float4*>(y + t * ystep) = make_float4(hn[0], hn[1], hn[2], hn[3]);
            float* gp = gates + t * sstep;
            if (abl & 8) {      // (A/B, tuning build: plain stores -- the L2 may combine the waves' 64-byte pieces of a line)
                *reinterpret_cast<f32x4*>(gp) = f32x4{rr[0], rr[1], rr[2], rr[3]};
                *reinterpret_cast<f32x4*>(gp + GH) = f32x4{zz[0], zz[1], zz[2], zz[3]};
                *reinterpret_cast<f32x4*>(gp + 2 * GH) = f32x4{nn[0], nn[1], nn[2], nn[3]};
                *reinterpret_cast<f32x4*>(gp + 3 * GH) = f32x4{ghn[0], ghn[1], ghn[2], ghn[3]};
            } else {
            __builtin_nontemporal_store(f32x4{rr[0], rr[1], rr[2], rr[3]}, reinterpret_cast<f32x4*>(gp));
            __builtin_nontemporal_store(f32x4{zz[0], zz[1], zz[2], zz[3]}, reinterpret_cast<f32x4*>(gp + GH));
            __builtin_nontemporal_store(f32x4{nn[0], nn[1], nn[2], nn[3]}, reinterpret_cast<f32x4*>(gp + 2 * GH));
            __builtin_nontemporal_store(f32x4{ghn[0], ghn[1], ghn[2], ghn[3]}, reinterpret_cast<f32x4*>(gp + 3 * GH));
            }
        }
    }
}

__global__ __launch_bounds__(FWD_THREADS) void gru_seq_fwd_mfma_kernel(const MfFwd G) {
    extern __shared__ __attribute__((aligned(16))) unsigned char gm_smem[];
    gru_fwd_mfma_body(G, (int)blockIdx.x, (int)blockIdx.y, gm_smem);
}

// The forward launch with the step's dropout-flag draw aboard (see gru.hip gru_seq_fwd_io_flags_kernel): the workgroups behind the
// recurrences draw the keep flags of the whole step (keep_flags_body.h), one per idle CU at most.
__global__ __launch_bounds__(FWD_THREADS) void gru_seq_fwd_mfma_flags_kernel(const MfFwd G, const kfb::FlagJob J, const int nslots) {
    extern __shared__ __attribute__((aligned(16))) unsigned char gm_smem[];
    const int bid = (int)blockIdx.x;
    if (bid < 2 * nslots) gru_fwd_mfma_body(G, bid % nslots, bid / nslots, gm_smem);
    else kfb::keep_flags_block<FWD_THREADS>(J, bid - 2 * nslots, (int)gridDim.x - 2 * nslots);
}

// Backward: same split.  The I/O wave moves 38 float4 per lane and step (dy | r z n ghn | h_prev of 16 sequences) through ONE
// register set: requested at step k for step k + 2, dropped at step k + 1.
// (the body of workgroup (bx, by) = (slot of 16 sequences, direction); gm_smem: BWD_LDS bytes of LDS)
__device__ __forceinline__ void gru_bwd_mfma_body(const MfBwd& G, const int bx, const int by, unsigned char* gm_smem) {
    unsigned char* dp = gm_smem;
    float* bs = reinterpret_cast<float*>(gm_smem + 2 * 3 * DPLANE);
    int gidx = 0;
    while (gidx + 1 < G.n && bx >= G.slice0[gidx + 1]) ++gidx;
    const int dir = by;
    const int rows = G.rows[gidx], T = G.T[gidx];
    const int row0 = (bx - G.slice0[gidx]) * MS;
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < 2 * 3 * DPLANE / 16; i += BWD_THREADS) reinterpret_cast<float4*>(dp)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    const int64_t ystep = (int64_t)rows * (2 * GH), sstep = (int64_t)rows * (8 * GH), gstep = (int64_t)rows * (6 * GH);

    if (w >= NWV) {
        // ---------------- the I/O wave ----------------
        // three sections with a uniform base pointer each (scalar base + one 32-bit offset register per load): dy (16 x 25
        // float4: 7 loads per lane), the saved gates (16 x 100: 25), h_prev (16 x 25: 7)
        // lane -> (sequence lane / 4, float4 (lane & 3) + 4 e of its row): every offset is one base register + an immediate
        constexpr int ND = 7;                                // 25 float4 per row: e = 6 is live for (lane & 3) == 0 only
        constexpr int NG = 25;                               // 100 float4 per row
        const float* __restrict__ dy = G.dy[gidx] + dir * GH;
        const float* __restrict__ yv = G.y[gidx] + dir * GH;
        const float* __restrict__ gates = G.gates[gidx] + dir * (4 * GH);
        const int sq = lane >> 2, l4 = lane & 3;
        const int rw = row0 + sq < rows ? row0 + sq : rows - 1;
        const int od0 = rw * (2 * GH) + 4 * l4, og0 = rw * (8 * GH) + 4 * l4, ls0 = sq * BROW + 4 * l4;
        const bool tail_on = l4 == 0;
        f32x4 qd[ND], qg[NG], qh[ND];
#define GM_ISSUE(SIDX)                                                                                               \
        do {                                                                                                         \
            const int sc_ = (SIDX) < T ? (SIDX) : T - 1;                                                             \
            const int t_ = dir ? sc_ : T - 1 - sc_;                                                                  \
            const int tp_ = dir ? t_ + 1 : t_ - 1;     /* the step that produced h_prev in the forward pass */       \
            const bool hp_ok_ = tp_ >= 0 && tp_ < T;                                                                 \
            const float* pdy_ = dy + t_ * ystep;                                                                     \
            const float* pg_ = gates + t_ * sstep;                                                                   \
            const float* ph_ = yv + (hp_ok_ ? tp_ : t_) * ystep;                                                     \
            _Pragma("unroll") for (int e = 0; e < ND; ++e)                                                           \
                qd[e] = *reinterpret_cast<const f32x4*>(pdy_ + od0 + ((e < ND - 1 || tail_on) ? 16 * e : 0));       \
            _Pragma("unroll") for (int e = 0; e < NG; ++e) qg[e] = *reinterpret_cast<const f32x4*>(pg_ + og0 + 16 * e); \
            _Pragma("unroll") for (int e = 0; e < ND; ++e) {                                                         \
                qh[e] = *reinterpret_cast<const f32x4*>(ph_ + od0 + ((e < ND - 1 || tail_on) ? 16 * e : 0));       \
                if (!hp_ok_) qh[e] = f32x4{0.f, 0.f, 0.f, 0.f};                                                      \
            }                                                                                                        \
        } while (0)
#define GM_DROP(SLOT)                                                                                                \
        do {                                                                                                         \
            float* dst_ = bs + (SLOT) * (BSLOT / 4) + ls0;                                                           \
            _Pragma("unroll") for (int e = 0; e < ND; ++e)                                                           \
                if (e < ND - 1 || tail_on) *reinterpret_cast<f32x4*>(dst_ + 16 * e) = qd[e];                         \
            _Pragma("unroll") for (int e = 0; e < NG; ++e) *reinterpret_cast<f32x4*>(dst_ + GH + 16 * e) = qg[e];     \
            _Pragma("unroll") for (int e = 0; e < ND; ++e)                                                           \
                if (e < ND - 1 || tail_on) *reinterpret_cast<f32x4*>(dst_ + 5 * GH + 16 * e) = qh[e];                \
        } while (0)
        GM_ISSUE(0);
        GM_DROP(0);
        GM_ISSUE(1);
        __syncthreads();
#pragma unroll 1
        for (int sidx = 0; sidx < T; ++sidx) {
            GM_DROP((sidx + 1) & 1);       // step sidx + 1 (its slot was read at step sidx - 1)
            GM_ISSUE(sidx + 2);
            step_barrier();
        }
#undef GM_ISSUE
#undef GM_DROP
        return;
    }

    const int s = lane & 15, g4 = lane >> 4;
    const int u0 = 16 * w + 4 * g4;
    const bool uok = u0 < GH;
    const int u0c = uok ? u0 : 0;
    const int row = row0 + s;
    const bool rok = row < rows;
    const float* __restrict__ w_hh = G.w_hh[2 * gidx + dir];

    // A fragments of W_hh^T: lane (s, g4) holds unit row 16 w + s, gate rows j = 32 ks + 8 g4 .. + 7
    u32x4 wf[10][3];
    {
        const int ua = 16 * w + s;
#pragma unroll
        for (int ks = 0; ks < 10; ++ks) {
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int j = 32 * ks + 8 * g4 + e;
                v[e] = (ua < GH && j < 3 * GH) ? w_hh[(int64_t)j * GH + ua] : 0.f;
            }
            cut8(v, wf[ks][0], wf[ks][1], wf[ks][2]);
        }
    }
    const uint32_t lds0 = (uint32_t)(uintptr_t)((__attribute__((address_space(3))) void*)dp);
    const uint32_t rd = lds0 + s * DROWB + 16 * g4;               // + 64 ks + plane
    const uint32_t wr = lds0 + s * DROWB + 2 * u0c;               // + 200 gate + plane
    const float* bsl = bs + s * BROW + u0c;

    float* __restrict__ dgi = G.dgi[gidx] + (int64_t)row * (6 * GH) + dir * 3 * GH + u0c;
    float* __restrict__ dgh = G.dgh[gidx] + (int64_t)row * (6 * GH) + dir * 3 * GH + u0c;

    float carry[4] = {0.f, 0.f, 0.f, 0.f};
    __syncthreads();
#pragma unroll 1
    for (int sidx = 0; sidx < T; ++sidx) {
        const int t = dir ? sidx : T - 1 - sidx;      // the reverse of the forward order
        const int cur = sidx & 1;
        // recurrent term from the step processed just before (its dgh pieces sit in planes cur ^ 1)
        f32x4 rec = {0.f, 0.f, 0.f, 0.f}, rec2 = rec, rec3 = rec, rec4 = rec;     // four independent chains
        const uint32_t rb = rd + (cur ^ 1) * 3 * DPLANE;
        const float* bq = bsl + cur * (BSLOT / 4);
        const float4 c_dy = *reinterpret_cast<const float4*>(bq), c_r = *reinterpret_cast<const float4*>(bq + GH),
                     c_z = *reinterpret_cast<const float4*>(bq + 2 * GH), c_n = *reinterpret_cast<const float4*>(bq + 3 * GH),
                     c_g = *reinterpret_cast<const float4*>(bq + 4 * GH), c_h = *reinterpret_cast<const float4*>(bq + 5 * GH);
#pragma unroll
        for (int ks = 0; ks < 10; ks += 2) {
            u32x4 b[3], c[3];
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                b[q] = *LDS_AS(u32x4, (uintptr_t)(rb + q * DPLANE + 64 * ks));
                c[q] = *LDS_AS(u32x4, (uintptr_t)(rb + q * DPLANE + 64 * ks + 64));
            }
            if (ks & 2) { rec3 = six(wf[ks], b, rec3); rec4 = six(wf[ks + 1], c, rec4); }
            else { rec = six(wf[ks], b, rec); rec2 = six(wf[ks + 1], c, rec2); }
        }
        rec = (rec + rec2) + (rec3 + rec4);
        const float dyv[4] = {c_dy.x, c_dy.y, c_dy.z, c_dy.w}, rv[4] = {c_r.x, c_r.y, c_r.z, c_r.w},
                    zv[4] = {c_z.x, c_z.y, c_z.z, c_z.w}, nv[4] = {c_n.x, c_n.y, c_n.z, c_n.w},
                    gv[4] = {c_g.x, c_g.y, c_g.z, c_g.w}, hv[4] = {c_h.x, c_h.y, c_h.z, c_h.w};
        float drp[4], dzp[4], dnp[4], dgn[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float dh = dyv[r] + carry[r] + rec[r];
            const float dn = dh * (1.0f - zv[r]);
            const float dz = dh * (hv[r] - nv[r]);
            carry[r] = dh * zv[r];
            dnp[r] = dn * (1.0f - nv[r] * nv[r]);
            drp[r] = dnp[r] * gv[r] * rv[r] * (1.0f - rv[r]);
            dzp[r] = dz * zv[r] * (1.0f - zv[r]);
            dgn[r] = dnp[r] * rv[r];
        }
        if (uok) {
            const uint32_t wb = wr + cur * 3 * DPLANE;
            u32x2 p1, p2, p3;
            cut4(drp[0], drp[1], drp[2], drp[3], p1, p2, p3);
            *LDS_AS(u32x2, (uintptr_t)wb) = p1;
            *LDS_AS(u32x2, (uintptr_t)(wb + DPLANE)) = p2;
            *LDS_AS(u32x2, (uintptr_t)(wb + 2 * DPLANE)) = p3;
            cut4(dzp[0], dzp[1], dzp[2], dzp[3], p1, p2, p3);
            *LDS_AS(u32x2, (uintptr_t)(wb + 2 * GH)) = p1;
            *LDS_AS(u32x2, (uintptr_t)(wb + 2 * GH + DPLANE)) = p2;
            *LDS_AS(u32x2, (uintptr_t)(wb + 2 * GH + 2 * DPLANE)) = p3;
            cut4(dgn[0], dgn[1], dgn[2], dgn[3], p1, p2, p3);
            *LDS_AS(u32x2, (uintptr_t)(wb + 4 * GH)) = p1;
            *LDS_AS(u32x2, (uintptr_t)(wb + 4 * GH + DPLANE)) = p2;
            *LDS_AS(u32x2, (uintptr_t)(wb + 4 * GH + 2 * DPLANE)) = p3;
        }
        step_barrier();
        if (uok && rok) {
            float* o = dgi + t * gstep;
            *reinterpret_cast<float4*>(o) = make_float4(drp[0], drp[1], drp[2], drp[3]);
            *reinterpret_cast<float4*>(o + GH) = make_float4(dzp[0], dzp[1], dzp[2], dzp[3]);
            *reinterpret_cast<float4*>(o + 2 * GH) = make_float4(dnp[0], dnp[1], dnp[2], dnp[3]);
            float* h = dgh + t * gstep;           // (read by the weight-gradient batch only)
            __builtin_nontemporal_store(f32x4{drp[0], drp[1], drp[2], drp[3]}, reinterpret_cast<f32x4*>(h));
            __builtin_nontemporal_store(f32x4{dzp[0], dzp[1], dzp[2], dzp[3]}, reinterpret_cast<f32x4*>(h + GH));
            __builtin_nontemporal_store(f32x4{dgn[0], dgn[1], dgn[2], dgn[3]}, reinterpret_cast<f32x4*>(h + 2 * GH));
        }
    }
}

__global__ __launch_bounds__(BWD_THREADS) void gru_seq_bwd_mfma_kernel(const MfBwd G) {
    extern __shared__ __attribute__((aligned(16))) unsigned char gm_smem[];
    gru_bwd_mfma_body(G, (int)blockIdx.x, (int)blockIdx.y, gm_smem);
}

// The backward launch WITH weight-gradient RIDERS (see gru.hip gru_seq_bwd_riders_kernel): the launch holds sequences / 16
// workgroups per direction (cfg3: 112 on 256 CUs) for T x ~2.4 us; workgroups behind them run tiles of a staged weight-gradient
// batch (gemm_tn_split_body.h: 512 threads and 108 KB of LDS, both inside this kernel's 512 / 137 KB), one workgroup per CU.
static_assert(BWD_THREADS == 512 && BWD_LDS >= tnsb::LDS_B, "the rider tiles run in this launch's workgroup shape");
__global__ __launch_bounds__(BWD_THREADS) void gru_seq_bwd_mfma_riders_kernel(const MfBwd G, const TnRiderSegs rq, const int nslots,
                                                                               const int ngru8) {
    extern __shared__ __attribute__((aligned(16))) unsigned char gm_smem[];
    const int bid = (int)blockIdx.x;
    if (bid < 2 * nslots) {
        gru_bwd_mfma_body(G, bid % nslots, bid / nslots, gm_smem);
    } else if (bid >= ngru8) {
        tnsb::tns_block<0>(rq, bid - ngru8, gm_smem, nullptr);
    }
}

}  // namespace

int mmdfn_launch_gru_fwd_mfma(int ngroups, const float* const* gi, const float* const* w_hh, const float* const* b_hh,
                              float* const* y, float* const* gates, const int* rows, const int* T, hipStream_t s) {
    if (ngroups <= 0 || ngroups > MAXG) return -2;
    MfFwd G;
    G.n = ngroups;
    int sl = 0;
    for (int g = 0; g < MAXG; ++g) {
        const bool on = g < ngroups;
        G.gi[g] = on ? gi[g] : nullptr; G.y[g] = on ? y[g] : nullptr; G.gates[g] = on ? gates[g] : nullptr;
        G.w_hh[2 * g] = on ? w_hh[2 * g] : nullptr; G.w_hh[2 * g + 1] = on ? w_hh[2 * g + 1] : nullptr;
        G.b_hh[2 * g] = on ? b_hh[2 * g] : nullptr; G.b_hh[2 * g + 1] = on ? b_hh[2 * g + 1] : nullptr;
        G.rows[g] = on ? rows[g] : 0; G.T[g] = on ? T[g] : 0; G.slice0[g] = sl;
        if (on) sl += (rows[g] + MS - 1) / MS;
    }
    G.slice0[MAXG] = sl;
    for (int g = ngroups; g < MAXG; ++g) G.slice0[g] = sl;
    G.abl = 0;
#ifdef MMDFN_TUNING
    if (const char* e = getenv("MMDFN_GRU_MF_ABL")) G.abl = atoi(e);
#endif
    if (G.abl == 0 && 2 * sl < 256) {
        if (const kfb::FlagJob* fj = mmdfn_flag_job_pending()) {
            // a staged dropout-flag draw rides on the CUs this launch leaves idle
            int64_t nr = (fj->n8 + FWD_THREADS - 1) / FWD_THREADS;
            if (nr > 256 - 2 * sl) nr = 256 - 2 * sl;
            const kfb::FlagJob J = *fj;
            mmdfn_flag_job_taken();
            if (int e = mmdfn_allow_big_lds(gru_seq_fwd_mfma_flags_kernel)) return e;
            hipLaunchKernelGGL(gru_seq_fwd_mfma_flags_kernel, dim3(2 * sl + (int)nr), dim3(FWD_THREADS), FWD_LDS, s, G, J, sl);
            MMDFN_CHECK_LAUNCH();
            return 0;
        }
    }
    if (int e = mmdfn_allow_big_lds(gru_seq_fwd_mfma_kernel)) return e;
    hipLaunchKernelGGL(gru_seq_fwd_mfma_kernel, dim3(sl, 2), dim3(FWD_THREADS), FWD_LDS, s, G);
    MMDFN_CHECK_LAUNCH();
    return 0;
}

int mmdfn_launch_gru_bwd_mfma(int ngroups, const float* const* dy, const float* const* y, const float* const* gates,
                              const float* const* w_hh, float* const* dgi, float* const* dgh, const int* rows, const int* T,
                              hipStream_t s) {
    if (ngroups <= 0 || ngroups > MAXG) return -2;
    MfBwd G;
    G.n = ngroups;
    int sl = 0;
    for (int g = 0; g < MAXG; ++g) {
        const bool on = g < ngroups;
        G.dy[g] = on ? dy[g] : nullptr; G.y[g] = on ? y[g] : nullptr; G.gates[g] = on ? gates[g] : nullptr;
        G.w_hh[2 * g] = on ? w_hh[2 * g] : nullptr; G.w_hh[2 * g + 1] = on ? w_hh[2 * g + 1] : nullptr;
        G.dgi[g] = on ? dgi[g] : nullptr; G.dgh[g] = on ? dgh[g] : nullptr;
        G.rows[g] = on ? rows[g] : 0; G.T[g] = on ? T[g] : 0; G.slice0[g] = sl;
        if (on) sl += (rows[g] + MS - 1) / MS;
    }
    G.slice0[MAXG] = sl;
    for (int g = ngroups; g < MAXG; ++g) G.slice0[g] = sl;
    G.abl = 0;
    if (const TnSplitSegs* rp = mmdfn_riders_pending()) {
        // a staged weight-gradient batch rides on the CUs this launch leaves idle
        if (rp->n <= MMDFN_RIDER_MAXSEG && 2 * sl < 256) {
            const TnRiderSegs rq = mmdfn_rider_table(*rp);
            const int ngru8 = (2 * sl + 7) & ~7;
            if (int e = mmdfn_allow_big_lds(gru_seq_bwd_mfma_riders_kernel)) return e;
            hipLaunchKernelGGL(gru_seq_bwd_mfma_riders_kernel, dim3(ngru8 + rp->wg_prefix[rp->n]), dim3(BWD_THREADS), BWD_LDS, s, G, rq,
                               sl, ngru8);
            MMDFN_CHECK_LAUNCH();
            return mmdfn_riders_launched(s);
        }
    }
    if (int e = mmdfn_allow_big_lds(gru_seq_bwd_mfma_kernel)) return e;
    hipLaunchKernelGGL(gru_seq_bwd_mfma_kernel, dim3(sl, 2), dim3(BWD_THREADS), BWD_LDS, s, G);
    MMDFN_CHECK_LAUNCH();
    return 0;
}
