// K2: fused GRU recurrence (forward and backward through time) for the context encoder
// ``lstm_l`` and the shared speaker-party encoder ``rnn_parties`` -- both
// nn.GRU(200, 100, num_layers=2, bidirectional=True) in the reference (model.py:866,868), which
// runs them step by step through ATen/cuDNN (MIOpen on ROCm: ~10^4 tiny launches per step).
//
// Split of one GRU layer:
//   * the input contraction  GI = X W_ih^T + b_ih  for ALL timesteps and both directions is one
//     dense GEMM outside this file (a true contraction -> MFMA);
//   * this kernel is the serial part: one PERSISTENT workgroup owns R batch rows of one direction
//     for the whole sequence, so there is no inter-workgroup synchronisation and no launch per step.
//     W_hh (3H x H fp32 = 120 KB per direction) lives in REGISTERS, sliced over the workgroup:
//     the lane pair (2u, 2u+1) keeps the r/z/n rows of hidden unit u, each lane one half of the
//     contraction index (3 x 52 floats); h_{t-1} is broadcast from LDS; the two partial sums meet
//     with one __shfl_xor and the owning lane applies the sigmoid/tanh gate math, writes y_t, the
//     saved gates and h_t: ONE barrier per timestep.  GI for step t+1 is prefetched meanwhile.
//   * several independent GRUs (text context + party batch) share ONE launch ("groups").
//
// Gate order r, z, n;  n = tanh(gi_n + r * (W_hn h + b_hn));  h = (1-z) n + z h_prev.
#include "mmdfn_internal.h"
#include "gemm_tn_split_body.h"
#include "keep_flags_body.h"
#include "../../include/mmdfn_hip.h"
#include <stdlib.h>

namespace {

constexpr int GH = 100;          // hidden size (the reference hard-codes D_e = 100, model.py:847-849)
constexpr int NT = 256;
constexpr int MAXG = 4;

// Valid-length truncation of the speaker-party sequences (exact; model.py:1076-1087 fills rows [:k_bp] of a zero (L, H)
// buffer per (dialogue b, speaker p), runs the party GRU over all L steps and scatters rows [:k_bp] back).  Rows [k_bp, L)
// of every party sequence carry the same input (zeros -> gi = b_ih), so
//   * a direction whose OUTPUTS at the padding positions nobody reads and whose padding lies BEHIND the data in its
//     processing order (layer 2, forward direction) stops after k_bp steps;
//   * a direction that meets the padding FIRST (layer 1, reverse direction: t = L-1 .. 0) starts at t = k_bp - 1 from the
//     state the all-padding ("silent") sequence has reached at position k_bp -- one extra sequence per launch, y_tab --
//     and its outputs at the padding positions are copies of that sequence's outputs;
//   * a speaker who never talks in a dialogue (k_bp = 0) needs nothing at all.
// The P party sequences of one (modality, dialogue) then make Sum_p k_bp = L_b <= L steps in total, so a truncated
// direction runs them back to back in ONE workgroup ("merged chain": P segments, state re-initialised at each
// segment start) instead of in P workgroups of L steps each.  k_bp is read from the gather's rank array (max + 1).
constexpr int SCHED_MAX = 2048;    // flattened steps of one chain (host checks P * T <= SCHED_MAX)
constexpr int MAXSEG = 16;         // segments (speakers) per merged chain

struct SegInfo {
    const int32_t* rank[MAXG];     // (T, BP) int32: rank[t][b*P + p] >= 0 iff speaker p talks at t (its position among p's
                                   // utterances), or nullptr: every row of the group runs all T steps in both directions
    int P[MAXG];                   // speakers; row r of the group is party (r % BP), rows r..r+P-1 (r % P == 0) share a dialogue
    int BP[MAXG];
    int tdir[MAXG];                // direction that runs as merged, truncated chains (-1: none)
    int nslot;                     // launch order of the (group, direction) slots: full-length slots first
    int slot_g[2 * MAXG];
    int slot_d[2 * MAXG];
    int slot_start[2 * MAXG + 1];
};

struct Chain {
    int gidx, dir, row0, nseg, S;
    bool trunc, has_rank;
};

// blockIdx.x -> chain (arithmetic only: the weight loads that depend on the group / direction can start before the counting)
__device__ __forceinline__ Chain seg_decode(const SegInfo& Sg) {
    Chain c;
    int sl = 0;
    while (sl + 1 < Sg.nslot && (int)blockIdx.x >= Sg.slot_start[sl + 1]) ++sl;
    c.gidx = Sg.slot_g[sl];
    c.dir = Sg.slot_d[sl];
    const int ci = (int)blockIdx.x - Sg.slot_start[sl];
    c.has_rank = Sg.rank[c.gidx] != nullptr;
    c.trunc = c.has_rank && Sg.tdir[c.gidx] == c.dir;
    c.row0 = c.trunc ? ci * Sg.P[c.gidx] : ci;
    c.nseg = c.trunc ? Sg.P[c.gidx] : 1;
    c.S = 0;
    return c;
}

// count the steps of the chain's segments and lay out the flattened schedule in LDS:
//   sched[s] = t | segment << 12 | (first step of its segment) << 16,   s = 0 .. S-1 in FORWARD processing order
// seg_k[p] = steps of segment p (a full-length chain: T, or 0 for a silent party row).  Ends with a barrier.
// Uniform (scalar-register) walk over the segments that have steps: which segment a flattened step belongs to and where the
// next one starts.  The per-step checks of the time loops are integer compares on these (an LDS lookup per step, or a
// readfirstlane of one, costs the 5-wave forward kernel 10 us per 110 steps: tools/ablate_gru_seg.py).
struct SegCursor {
    int p, off, k;      // current segment, its first flattened step, its step count (k = 0: no segment left)
    int next;           // first flattened step of the next segment with steps (INT_MAX: none)
};

__device__ __forceinline__ int seg_next_with_steps(const int* seg_k, int nseg, int p) {
    ++p;
    while (p < nseg && seg_k[p] == 0) ++p;
    return p;
}

__device__ __forceinline__ SegCursor seg_cursor_begin(const int* seg_k, int nseg) {
    SegCursor c;
    c.p = seg_next_with_steps(seg_k, nseg, -1);
    c.off = 0;
    c.k = c.p < nseg ? seg_k[c.p] : 0;
    const int q = seg_next_with_steps(seg_k, nseg, c.p);
    c.next = q < nseg ? c.off + c.k : 0x7fffffff;
    c.p = __builtin_amdgcn_readfirstlane(c.p);
    c.k = __builtin_amdgcn_readfirstlane(c.k);
    c.next = __builtin_amdgcn_readfirstlane(c.next);
    return c;
}

__device__ __forceinline__ void seg_cursor_advance(SegCursor& c, const int* seg_k, int nseg) {
    c.off += c.k;
    c.p = seg_next_with_steps(seg_k, nseg, c.p);
    c.k = c.p < nseg ? seg_k[c.p] : 0;
    const int q = seg_next_with_steps(seg_k, nseg, c.p);
    c.next = q < nseg ? c.off + c.k : 0x7fffffff;
    c.p = __builtin_amdgcn_readfirstlane(c.p);
    c.k = __builtin_amdgcn_readfirstlane(c.k);
    c.next = __builtin_amdgcn_readfirstlane(c.next);
}

template <int NTH>
__device__ __forceinline__ void seg_schedule(const SegInfo& Sg, const int* Tg, Chain& c, uint32_t* sched, int* seg_k) {
    const int tid = threadIdx.x;
    const int32_t* __restrict__ rk = Sg.rank[c.gidx];
    const int BP = Sg.BP[c.gidx], T = Tg[c.gidx];
    if (tid < MAXSEG) seg_k[tid] = 0;
    __syncthreads();
    if (c.has_rank) {
        for (int idx = tid; idx < c.nseg * T; idx += NTH) {
            const int p = idx / T;
            const int t = idx - p * T;
            const int v = rk[(int64_t)t * BP + (c.row0 + p) % BP];
            if (v >= 0) atomicMax(&seg_k[p], v + 1);
        }
        __syncthreads();
        if (!c.trunc && tid == 0) seg_k[0] = seg_k[0] > 0 ? T : 0;
    } else if (tid == 0) {
        seg_k[0] = T;
    }
    __syncthreads();
    int S = 0;
    for (int p = 0; p < c.nseg; ++p) S += seg_k[p];
    c.S = S;
    for (int idx = tid; idx < c.nseg * T; idx += NTH) {
        const int p = idx / T;
        const int j = idx - p * T;
        const int k = seg_k[p];
        if (j < k) {
            int off = 0;
            for (int q = 0; q < p; ++q) off += seg_k[q];
            const int t = c.dir ? k - 1 - j : j;
            sched[off + j] = (uint32_t)t | ((uint32_t)p << 12) | ((j == 0 ? 1u : 0u) << 16);
        }
    }
    __syncthreads();
}

template <int NTH>
__device__ __forceinline__ Chain seg_setup(const SegInfo& Sg, const int* Tg, uint32_t* sched, int* seg_k) {
    Chain c = seg_decode(Sg);
    seg_schedule<NTH>(Sg, Tg, c, sched, seg_k);
    return c;
}

struct FwdGroups {
    int n;
    const float* gi[MAXG];
    const float* w_hh[2 * MAXG];   // [2g + direction]: (3H, H), the module's weight_hh_l*[_reverse] as stored
    const float* b_hh[2 * MAXG];
    float* y[MAXG];
    float* gates[MAXG];
    int rows[MAXG];
    int T[MAXG];
    int slice0[MAXG + 1];
    int abl;                       // tuning build only (MMDFN_GRU_ABL): timing ablations of the forward step, wrong results
    SegInfo seg;                   // segmented (valid-length) launches only
    const float* ytab[MAXG];       // (T, 1, 2H) outputs of the all-padding sequence, or nullptr (zero start / zero fill)
};

struct BwdGroups {
    int n;
    const float* dy[MAXG];
    const float* y[MAXG];
    const float* gates[MAXG];
    const float* w_hh[2 * MAXG];
    float* dgi[MAXG];
    float* dgh[MAXG];
    int rows[MAXG];
    int T[MAXG];
    int slice0[MAXG + 1];
    SegInfo seg;                   // segmented (valid-length) launches only
    float* dhinit[MAXG];           // (rows, H): gradient wrt the start state of every truncated row (nullptr: not wanted)
    int32_t* kout[MAXG];           // (rows): steps run per truncated row (with dhinit)
};

// Gate non-linearities on the hardware transcendental units (v_exp_f32 / v_rcp_f32, ~1 ulp each): the
// accurate libm expf/tanhf are ~600 cycles of dependent scalar code per timestep on the serial critical
// path of the recurrence (tools/ubench/step_latency.hip).  Absolute error < 3e-7, far inside the 1e-5 parity
// budget of the encoders (logits 1e-4).
__device__ __forceinline__ float sigmoidf_(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float tanhf_(float x) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(2.0f * x)); }

typedef float f32x2 __attribute__((ext_vector_type(2)));

// A use of a freshly loaded register BEFORE the time loop.  Without it the compiler sinks loop-invariant global loads to
// their first use and plants their s_waitcnt vmcnt(N) inside the step loop, where -- vector memory operations retire in
// order and the counter also counts stores -- every later trip would wait for the previous block's result stores.
__device__ __forceinline__ void pin_loaded(float& v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ void pin_loaded(f32x2& v) {
    float a = v[0], b = v[1];
    asm volatile("" : "+v"(a), "+v"(b));
    v[0] = a;
    v[1] = b;
}

// exchange with the other lane of the pair (lane ^ 1): one DPP quad_perm [1,0,3,2] move instead of a
// ds_bpermute round trip through the LDS crossbar
__device__ __forceinline__ float pair_swap(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true));
}

// Thread mapping of both kernels: 256 threads, lane pair (2u, 2u+1) owns hidden unit u (u < 100; the
// last 56 lanes idle).  The pair splits the contraction index in two 16-byte aligned halves
// [0, 52) and [52, 100); each lane keeps its 3 x 52 weights in registers, the two partial sums meet with
// ONE cross-lane exchange (__shfl_xor 1), and the lane that owns (row, unit) applies the gate math itself,
// so a timestep needs a single workgroup barrier (h is double-buffered in LDS).
constexpr int KH0 = 52;                 // split point of the contraction index
constexpr int KW = 52;                  // weights held per gate per lane (second half: 48 used)

// Global memory traffic of the time loop is BLOCKED: gfx950 retires vector memory operations in order and
// vmcnt counts stores as well as loads, so a per-step "wait for the next step's operands" would also wait
// for the previous step's result stores (a full store round trip on the serial critical path).  Instead the
// operands of TB = 8/R consecutive timesteps are copied global -> registers one block ahead (coalesced
// 16-byte loads by all 256 threads), dropped into LDS at the block boundary, and the results of a block are
// collected in LDS and written out with coalesced 16-byte stores that nobody waits for.  Inside a block the
// recurrence touches LDS only.

// SEG = 1 (R = 1 only): segmented launch, see SegInfo and gru_seq_fwd_io_kernel.
template <int R, int SEG>
__global__ __launch_bounds__(NT) void gru_seq_fwd_kernel(FwdGroups G) {
    static_assert(!SEG || R == 1, "segmented launches run one row per workgroup");
    constexpr int TB = 8 / R;
    constexpr int IN4 = 3 * GH / 4;                 // float4 per (step, row) of staged input  (gi: r, z, n)
    constexpr int OUT4 = 5 * GH / 4;                // float4 per (step, row) of staged output (y, r, z, n, ghn)
    constexpr int NIN = (TB * R * IN4 + NT - 1) / NT;
    __shared__ __attribute__((aligned(16))) float hs[2][R][GH + 4];
    __shared__ __attribute__((aligned(16))) float in_s[2][TB][R][3 * GH];
    __shared__ __attribute__((aligned(16))) float out_s[TB][R][5 * GH];
    __shared__ uint32_t sched[SEG ? SCHED_MAX : 1];
    __shared__ int seg_k[MAXSEG];
    __shared__ __attribute__((aligned(16))) float seg_init[SEG ? MAXSEG : 1][GH];

    int gidx = 0;
    int dir_ = 0, row_ = 0, T_ = 0;
    Chain ch{};
    const int tid = threadIdx.x;
    const int u = tid >> 1;
    const int half = tid & 1;
    const bool active = u < GH;
    const int uu = active ? u : GH - 1;
    const int kbase = half ? KH0 : 0;
    const int klen = half ? GH - KH0 : KH0;
    // weights as packed pairs: v_pk_fma_f32 retires two FMAs per issue slot (a single wave per SIMD issues
    // one VALU instruction every ~4 cycles, so halving the instruction count halves the matvec time)
    f32x2 wr[KW / 2], wz[KW / 2], wn[KW / 2];
    auto load_weights = [&](const float* __restrict__ wsrc) {
#pragma unroll
        for (int k = 0; k < KW; ++k) {
            const bool ok = k < klen;
            const int kk = kbase + (ok ? k : 0);
            wr[k >> 1][k & 1] = ok ? wsrc[(int64_t)(0 * GH + uu) * GH + kk] : 0.f;
            wz[k >> 1][k & 1] = ok ? wsrc[(int64_t)(1 * GH + uu) * GH + kk] : 0.f;
            wn[k >> 1][k & 1] = ok ? wsrc[(int64_t)(2 * GH + uu) * GH + kk] : 0.f;
        }
    };
    if constexpr (SEG) {
        ch = seg_decode(G.seg);
        gidx = ch.gidx;
        dir_ = ch.dir;
        row_ = ch.row0;
        load_weights(G.w_hh[2 * gidx + dir_]);     // in flight while the steps are counted
        seg_schedule<NT>(G.seg, G.T, ch, sched, seg_k);
        T_ = ch.S;                                 // the time loop runs over the flattened schedule
    } else {
        while (gidx + 1 < G.n && (int)blockIdx.x >= G.slice0[gidx + 1]) ++gidx;
        dir_ = blockIdx.y;
        row_ = ((int)blockIdx.x - G.slice0[gidx]) * R;
        T_ = G.T[gidx];
    }
    const int dir = dir_;
    const int rows = G.rows[gidx];
    const int T = T_;
    const int row0 = row_;
    // (row, t) of flattened step sidx (SEG), or of step sidx of row r
    auto step_row = [&](int sidx, int r, int& t) {
        if constexpr (SEG) {
            const uint32_t e = sched[sidx];
            t = (int)(e & 0xFFFu);
            return row0 + (int)((e >> 12) & 0xFu);
        } else {
            t = dir ? T - 1 - sidx : sidx;
            return row0 + r;
        }
    };
    const float* __restrict__ gi = G.gi[gidx];
    const float* __restrict__ w_hh = G.w_hh[2 * gidx + dir];
    const float* __restrict__ b_hh = G.b_hh[2 * gidx + dir];
    float* __restrict__ y = G.y[gidx];
    float* __restrict__ gates = G.gates[gidx];

    // SEG: positions this chain's rows do not visit get the all-padding sequence's outputs, or zeros (see gru_seq_fwd_io_kernel)
    auto fill_unvisited = [&]() {
        if (!ch.has_rank) return;
        const float* __restrict__ yt = ch.trunc ? G.ytab[gidx] : nullptr;
        const int Tfull = G.T[gidx];
        const int per = Tfull * (GH / 4);
        for (int idx = tid; idx < ch.nseg * per; idx += NT) {
            const int p = idx / per;
            const int rem = idx - p * per;
            const int t = rem / (GH / 4);
            const int c4 = rem - t * (GH / 4);
            if (t < seg_k[p]) continue;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (yt != nullptr) v = *reinterpret_cast<const float4*>(yt + (int64_t)t * (2 * GH) + dir * GH + 4 * c4);
            *reinterpret_cast<float4*>(y + ((int64_t)t * rows + row0 + p) * (2 * GH) + dir * GH + 4 * c4) = v;
        }
    };
    if constexpr (SEG) {
        fill_unvisited();              // (up front: the stores drain under the prologue)
        if (T == 0) return;            // a silent row / a dialogue without a party utterance: nothing to run
    }
#ifdef MMDFN_TUNING
    const int abl = G.abl;         // 1 no matvec, 2 no transcendental gate math, 4 no block traffic (stash / flush / prefetch),
                                   // 8 no deferred result writes, 16 no per-step barrier
#else
    constexpr int abl = 0;
#endif

    if constexpr (!SEG) load_weights(w_hh);
#pragma unroll
    for (int k = 0; k < KW / 2; ++k) {
        pin_loaded(wr[k]);
        pin_loaded(wz[k]);
        pin_loaded(wn[k]);
    }
    float bhr = b_hh[uu], bhz = b_hh[GH + uu], bhn = b_hh[2 * GH + uu];
    pin_loaded(bhr);
    pin_loaded(bhz);
    pin_loaded(bhn);

    float hprev[R];
    bool mine[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        mine[r] = active && ((r & 1) == half) && (row0 + r < rows);
        hprev[r] = 0.f;
    }

    const int nblocks = (T + TB - 1) / TB;
    float4 stage[NIN];
    auto load_block = [&](int b) {            // global -> registers (raw; out-of-range slots are never consumed)
#pragma unroll
        for (int e = 0; e < NIN; ++e) {
            const int idx = tid + e * NT;
            const int sl = idx / (R * IN4);
            const int rem = idx - sl * (R * IN4);
            const int r = rem / IN4;
            const int c4 = rem - r * IN4;
            const int sidx = b * TB + sl;
            stage[e] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (sl < TB && sidx < T && row0 + r < rows) {
                int t;
                const int rw = step_row(sidx, r, t);
                stage[e] = *reinterpret_cast<const float4*>(gi + ((int64_t)t * rows + rw) * (6 * GH) + dir * 3 * GH + 4 * c4);
            }
        }
    };
    auto stash_block = [&](int buf) {          // registers -> LDS
#pragma unroll
        for (int e = 0; e < NIN; ++e) {
            const int idx = tid + e * NT;
            if (idx < TB * R * IN4) *reinterpret_cast<float4*>(&in_s[buf][0][0][0] + 4 * idx) = stage[e];
        }
    };
    auto flush_block = [&](int b) {            // LDS -> global, nobody waits for these stores
        for (int idx = tid; idx < TB * R * OUT4; idx += NT) {
            const int sl = idx / (R * OUT4);
            const int rem = idx - sl * (R * OUT4);
            const int r = rem / OUT4;
            const int c4 = rem - r * OUT4;
            const int sidx = b * TB + sl;
            if (sidx >= T || row0 + r >= rows) continue;
            int t;
            const int rw = step_row(sidx, r, t);
            const float4 v = *reinterpret_cast<const float4*>(&out_s[sl][r][4 * c4]);
            const int64_t o = ((int64_t)t * rows + rw);
            if (c4 < GH / 4)
                *reinterpret_cast<float4*>(y + o * (2 * GH) + dir * GH + 4 * c4) = v;
            else
                __builtin_nontemporal_store(__builtin_bit_cast(f32x4, v), reinterpret_cast<f32x4*>(gates + (o * 2 + dir) * (4 * GH) + 4 * (c4 - GH / 4)));   // (saved for the backward pass)
        }
    };

    uint32_t e_cur = 0;
    if constexpr (SEG) {
        // start state of every segment: the all-padding sequence's state at position k (reverse direction), zero otherwise
        const float* __restrict__ yt = ch.trunc ? G.ytab[gidx] : nullptr;
        const int Tfull = G.T[gidx];
        for (int idx = tid; idx < ch.nseg * GH; idx += NT) {
            const int p = idx / GH;
            const int uq = idx - p * GH;
            const int kq = seg_k[p];
            seg_init[p][uq] = (yt != nullptr && kq > 0 && kq < Tfull) ? yt[(int64_t)kq * (2 * GH) + dir * GH + uq] : 0.f;
        }
        __syncthreads();
        e_cur = T > 0 ? sched[0] : 0u;
        const int p0 = (e_cur >> 12) & 0xFu;
        for (int i = tid; i < 2 * R * (GH + 4); i += NT) (&hs[0][0][0])[i] = (T > 0 && i < GH) ? seg_init[p0][i] : 0.f;
        hprev[0] = T > 0 ? seg_init[p0][uu] : 0.f;
    } else {
        for (int i = tid; i < 2 * R * (GH + 4); i += NT) (&hs[0][0][0])[i] = 0.f;
    }
    load_block(0);
    stash_block(0);
    if (nblocks > 1) load_block(1);
    __syncthreads();

    // LDS traffic of a step that is NOT the recurrence itself is kept off its critical path (timing ablations,
    // profiles/r02_gru_kernels.md: these eight accesses cost as much as the 78 packed FMAs when they sit between the matvec
    // and the barrier):
    //   * the step's three input-gate operands are read at the TOP of the step, under the matvec, not behind it;
    //   * the five result values (y, r, z, n, W_hn h) of step s are written AFTER the barrier of step s, at the top of
    //     step s + 1 (they stay in registers across the barrier), so the barrier only waits for the one write it exists
    //     for (h -> LDS).  The block's last step writes them before the block-boundary barrier.
    float pend[R][5];
    bool have_pend = false;
    int pend_sl = 0;
    auto write_pending = [&]() {
#pragma unroll
        for (int r = 0; r < R; ++r)
            if (mine[r]) {
                float* op = &out_s[pend_sl][r][u];
                op[0] = pend[r][0];
                op[GH] = pend[r][1];
                op[2 * GH] = pend[r][2];
                op[3 * GH] = pend[r][3];
                op[4 * GH] = pend[r][4];
            }
    };
    int step = 0;
    for (int b = 0; b < nblocks; ++b) {
        const int buf = b & 1;
        for (int sl = 0; sl < TB && step < T; ++sl, ++step) {
            const int cur = step & 1;
            if constexpr (SEG) {
                // a new segment: the state (LDS copy and register) becomes the segment's start state (e_cur was fetched during
                // the previous step)
                if (step > 0 && __builtin_amdgcn_readfirstlane((e_cur >> 16) & 1u)) {
                    hprev[0] = seg_init[(e_cur >> 12) & 0xFu][uu];
                    if (mine[0]) hs[cur][0][u] = hprev[0];
                    __syncthreads();
                }
            }
            if (have_pend && !(abl & 8)) write_pending();
            float g0[R], g1[R], g2[R];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const float* gp = &in_s[buf][sl][r][uu];
                g0[r] = gp[0];
                g1[r] = gp[GH];
                g2[r] = gp[2 * GH];
            }
#pragma unroll
            for (int r = 0; r < R; ++r) {
                f32x2 ar0 = {0.f, 0.f}, az0 = {0.f, 0.f}, an0 = {0.f, 0.f}, ar1 = {0.f, 0.f}, az1 = {0.f, 0.f}, an1 = {0.f, 0.f};
                const float4* hv = reinterpret_cast<const float4*>(&hs[cur][r][kbase]);
                if (abl & 1) {
                    ar0[0] = hs[cur][r][u & 63];
                } else if constexpr (R == 1) {
                    // The 13 broadcast loads of h_{t-1} run HLOOK groups ahead of their six packed FMAs, through a ring of
                    // HLOOK float4 registers filled by inline-asm ds_read_b128 and retired with hand-counted lgkmcnt waits.
                    // Left to itself hipcc keeps two loads in flight (the weights hold 156 of the 256 VGPRs), so every
                    // group of FMAs (~25 cycles of issue) waited for an LDS round trip: ~9 exposed waits per step, the largest
                    // single item of the step (profiles/r03_gru_kernels.md).  "memory" keeps the compiler's own LDS
                    // operations (the gate operands above, the result writes) out of the counted window.
                    constexpr int HLOOK = 6;
                    const uint32_t haddr = (uint32_t)(uintptr_t)((__attribute__((address_space(3))) const float*)(&hs[cur][0][kbase]));
                    f32x4 hb[HLOOK];
#define GRU_LDS_RD(J)                                                                                        \
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(hb[(J) % HLOOK]) : "v"(haddr), "i"(16 * (J)) : "memory")
#define GRU_LDS_WAIT(N, J) asm volatile("s_waitcnt lgkmcnt(" #N ")" : "+v"(hb[(J) % HLOOK]) : : "memory")
                    GRU_LDS_RD(0); GRU_LDS_RD(1); GRU_LDS_RD(2); GRU_LDS_RD(3); GRU_LDS_RD(4); GRU_LDS_RD(5);
#define GRU_GROUP(K4)                                                                                        \
    do {                                                                                                     \
        const f32x4 h4 = hb[(K4) % HLOOK];                                                                   \
        const f32x2 ha = {h4.x, h4.y}, hb2 = {h4.z, h4.w};                                                   \
        ar0 = __builtin_elementwise_fma(wr[2 * (K4)], ha, ar0);                                              \
        az0 = __builtin_elementwise_fma(wz[2 * (K4)], ha, az0);                                              \
        an0 = __builtin_elementwise_fma(wn[2 * (K4)], ha, an0);                                              \
        ar1 = __builtin_elementwise_fma(wr[2 * (K4) + 1], hb2, ar1);                                         \
        az1 = __builtin_elementwise_fma(wz[2 * (K4) + 1], hb2, az1);                                         \
        an1 = __builtin_elementwise_fma(wn[2 * (K4) + 1], hb2, an1);                                         \
    } while (0)
                    // group k waits until at most min(5, 12 - k) younger loads are outstanding, then its slot is reloaded
                    GRU_LDS_WAIT(5, 0); GRU_GROUP(0); GRU_LDS_RD(6);
                    GRU_LDS_WAIT(5, 1); GRU_GROUP(1); GRU_LDS_RD(7);
                    GRU_LDS_WAIT(5, 2); GRU_GROUP(2); GRU_LDS_RD(8);
                    GRU_LDS_WAIT(5, 3); GRU_GROUP(3); GRU_LDS_RD(9);
                    GRU_LDS_WAIT(5, 4); GRU_GROUP(4); GRU_LDS_RD(10);
                    GRU_LDS_WAIT(5, 5); GRU_GROUP(5); GRU_LDS_RD(11);
                    GRU_LDS_WAIT(5, 6); GRU_GROUP(6); GRU_LDS_RD(12);
                    GRU_LDS_WAIT(5, 7); GRU_GROUP(7);
                    GRU_LDS_WAIT(4, 8); GRU_GROUP(8);
                    GRU_LDS_WAIT(3, 9); GRU_GROUP(9);
                    GRU_LDS_WAIT(2, 10); GRU_GROUP(10);
                    GRU_LDS_WAIT(1, 11); GRU_GROUP(11);
                    GRU_LDS_WAIT(0, 12); GRU_GROUP(12);
#undef GRU_GROUP
#undef GRU_LDS_WAIT
#undef GRU_LDS_RD
                } else {
#pragma unroll
                    for (int k4 = 0; k4 < KW / 4; ++k4) {
                        // the second half has 12 real float4 (48 floats); its 13th reads the 4 zero pad floats
                        const float4 h4 = hv[k4];
                        const f32x2 ha = {h4.x, h4.y}, hb = {h4.z, h4.w};
                        ar0 = __builtin_elementwise_fma(wr[2 * k4], ha, ar0);
                        az0 = __builtin_elementwise_fma(wz[2 * k4], ha, az0);
                        an0 = __builtin_elementwise_fma(wn[2 * k4], ha, an0);
                        ar1 = __builtin_elementwise_fma(wr[2 * k4 + 1], hb, ar1);
                        az1 = __builtin_elementwise_fma(wz[2 * k4 + 1], hb, az1);
                        an1 = __builtin_elementwise_fma(wn[2 * k4 + 1], hb, an1);
                    }
                }
                float ar = (ar0.x + ar0.y) + (ar1.x + ar1.y);
                float az = (az0.x + az0.y) + (az1.x + az1.y);
                float an = (an0.x + an0.y) + (an1.x + an1.y);
                ar += pair_swap(ar);
                az += pair_swap(az);
                an += pair_swap(an);
                {   // (every lane does the gate math -- a conditional would pull the operand reads back behind the matvec)
                    const float ghn = an + bhn;
                    const float rr = (abl & 2) ? (g0[r] + ar + bhr) * 0.01f : sigmoidf_(g0[r] + ar + bhr);
                    const float zz = (abl & 2) ? (g1[r] + az + bhz) * 0.01f : sigmoidf_(g1[r] + az + bhz);
                    const float nn = (abl & 2) ? (g2[r] + rr * ghn) * 0.01f : tanhf_(g2[r] + rr * ghn);
                    const float hnew = (1.0f - zz) * nn + zz * hprev[r];
                    if (mine[r]) hs[cur ^ 1][r][u] = hnew;
                    hprev[r] = hnew;
                    pend[r][0] = hnew;
                    pend[r][1] = rr;
                    pend[r][2] = zz;
                    pend[r][3] = nn;
                    pend[r][4] = ghn;
                }
            }
            have_pend = true;
            pend_sl = sl;
            if constexpr (SEG) e_cur = step + 1 < T ? sched[step + 1] : 0u;
            if (!(abl & 16)) __syncthreads();
        }
        if (have_pend) write_pending();
        have_pend = false;
        __syncthreads();                        // the block's results are complete in out_s
        // block boundary: next block's operands (loaded a block ago) drop into LDS, this block's results
        // leave, and the loads of block b+2 are issued
        if (!(abl & 4)) {
            if (b + 1 < nblocks) stash_block(buf ^ 1);
            flush_block(b);
            if (b + 2 < nblocks) load_block(b + 2);
        }
        __syncthreads();
    }
}

// =====================================================================================================
// Forward pass, one sequence per workgroup, with a FIFTH wave that does all the global-memory traffic.
//
// Ablations of the 4-wave kernel above (tools/ablate_gru_fwd.py, profiles/r03_gru_kernels.md): of a 0.90 us step the matvec
// is 35 %, the block traffic (every 8 steps all four waves stop to move the next operands registers -> LDS, drain the
// results LDS -> global and issue the next prefetch: two more barriers, index arithmetic, LDS round trips) 22 %, the
// per-step barrier 14 %, the five scattered result writes per step 8 %, the transcendental gate math 6 %.  Here the four
// recurrence waves never touch global memory inside the time loop: wave 4 prefetches the gate pre-activations of block
// b+1 (TB steps) into the other half of a double-buffered LDS array and drains the results of block b-1 from the other half
// of a double-buffered result array, one step's worth per step, and joins the same per-step barrier; the block boundary
// needs no barrier of its own (the last step of a block writes its results before its barrier instead of after it).  A unit's
// five results go to LDS as one 16-byte + one 4-byte write ([unit][8] layout) instead of five 4-byte writes.
// =====================================================================================================
// ABL (tuning build, timing only): 1 no matvec, 2 no transcendental gate math, 4 no result writes, 8 no h exchange wait
// (the LDS write of h stays, the barrier goes), 16 no gate-operand reads
// SEG = 1: segmented launch (see SegInfo): 1-D grid of chains, the time loop runs over the chain's flattened schedule
// (S steps, each naming its row and t), the state is re-initialised at segment starts (one extra barrier there), and the
// positions a truncated / silent row does not visit are filled afterwards (copies of the all-padding sequence, or zeros).
// (the body of workgroup (bx, by) = (sequence slot, direction) of the plain launch; SEG launches decode their chain themselves)
template <int SCALAR_FMA, int ABL, int SEG>
__device__ __forceinline__ void gru_fwd_io_body(const FwdGroups& G, const int bx, const int by) {
    constexpr int TB = 4;                          // steps per block
    constexpr int NLD = (TB * 3 * GH / 4 + 63) / 64;   // float4 loads per I/O lane and block (5)
    __shared__ __attribute__((aligned(16))) float hs[2][GH + 4];
    __shared__ __attribute__((aligned(16))) float in_s[2][TB][3 * GH];
    __shared__ __attribute__((aligned(16))) float out_s[2][TB][GH][8];     // y r z n | W_hn h + b_hn, 3 pad
    __shared__ uint32_t sched[SEG ? SCHED_MAX : 1];
    __shared__ int seg_k[MAXSEG];
    __shared__ __attribute__((aligned(16))) float seg_init[SEG ? MAXSEG : 1][GH];

    int gidx = 0;
    int dir_ = 0, row_ = 0, T_ = 0;
    Chain ch{};
    // lane pair (2u, 2u+1) = hidden unit u x half of the contraction (the recurrence waves; the I/O wave's lanes map to unit GH-1)
    const int tid = threadIdx.x;
    const int u = tid >> 1;
    const int half = tid & 1;
    const bool active = u < GH;
    const int uu = active ? u : GH - 1;
    const int kbase = half ? KH0 : 0;
    const int klen = half ? GH - KH0 : KH0;
    f32x2 wr[KW / 2], wz[KW / 2], wn[KW / 2];
    auto load_weights = [&](const float* __restrict__ wsrc) {
#pragma unroll
        for (int k = 0; k < KW; ++k) {
            const bool ok = k < klen;
            const int kk = kbase + (ok ? k : 0);
            wr[k >> 1][k & 1] = ok ? wsrc[(int64_t)(0 * GH + uu) * GH + kk] : 0.f;
            wz[k >> 1][k & 1] = ok ? wsrc[(int64_t)(1 * GH + uu) * GH + kk] : 0.f;
            wn[k >> 1][k & 1] = ok ? wsrc[(int64_t)(2 * GH + uu) * GH + kk] : 0.f;
        }
    };
    if constexpr (SEG) {
        ch = seg_decode(G.seg);
        gidx = ch.gidx;
        dir_ = ch.dir;
        row_ = ch.row0;
        // the weight loads (they need the group and the direction only) are in flight while the steps are counted
        if (tid < NT) load_weights(G.w_hh[2 * gidx + dir_]);
        seg_schedule<320>(G.seg, G.T, ch, sched, seg_k);
        T_ = ch.S;                                 // the time loop runs over the flattened schedule
    } else {
        while (gidx + 1 < G.n && bx >= G.slice0[gidx + 1]) ++gidx;
        dir_ = by;
        row_ = bx - G.slice0[gidx];
        T_ = G.T[gidx];
    }
    const int dir = dir_;
    const int rows = G.rows[gidx];
    const int T = T_;
    const int row = row_;
    const float* __restrict__ gi = G.gi[gidx];
    const float* __restrict__ w_hh = G.w_hh[2 * gidx + dir];
    const float* __restrict__ b_hh = G.b_hh[2 * gidx + dir];
    float* __restrict__ y = G.y[gidx];
    float* __restrict__ gates = G.gates[gidx];
    const int nblocks = (T + TB - 1) / TB;
    if constexpr (SEG) {
        // start state of every segment: the all-padding sequence's state at position k (reverse direction: it has consumed
        // positions T-1 .. k), zero otherwise
        const float* __restrict__ yt = ch.trunc ? G.ytab[gidx] : nullptr;
        const int Tfull = G.T[gidx];
        for (int idx = tid; idx < ch.nseg * GH; idx += 320) {
            const int p = idx / GH;
            const int uq = idx - p * GH;
            const int k = seg_k[p];
            seg_init[p][uq] = (yt != nullptr && k > 0 && k < Tfull) ? yt[(int64_t)k * (2 * GH) + dir * GH + uq] : 0.f;
        }
        __syncthreads();
    }
    // SEG: positions this chain's rows do not visit get the all-padding sequence's outputs (a truncated direction that starts
    // from them) or zeros (never read as values; they are operands of dense products whose other factor is an exact zero there)
    auto fill_unvisited = [&](int first, int nthreads) {
        if (!ch.has_rank) return;
        const float* __restrict__ yt = ch.trunc ? G.ytab[gidx] : nullptr;
        const int Tfull = G.T[gidx];
        const int per = Tfull * (GH / 4);
        for (int idx = first; idx < ch.nseg * per; idx += nthreads) {
            const int p = idx / per;
            const int rem = idx - p * per;
            const int t = rem / (GH / 4);
            const int c4 = rem - t * (GH / 4);
            if (t < seg_k[p]) continue;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (yt != nullptr) v = *reinterpret_cast<const float4*>(yt + (int64_t)t * (2 * GH) + dir * GH + 4 * c4);
            *reinterpret_cast<float4*>(y + ((int64_t)t * rows + row + p) * (2 * GH) + dir * GH + 4 * c4) = v;
        }
    };
    if constexpr (SEG) {
        if (T == 0) {                  // a silent row / a dialogue without a party utterance: nothing to run, no weights needed
            fill_unvisited(tid, 320);
            return;
        }
    }
    // (row, t) of flattened step sidx
    auto step_row = [&](int sidx, int& t) {
        if constexpr (SEG) {
            const uint32_t e = sched[sidx];
            t = (int)(e & 0xFFFu);
            return row + (int)((e >> 12) & 0xFu);
        } else {
            t = dir ? T - 1 - sidx : sidx;
            return row;
        }
    };

    if (tid >= NT) {
        // ---------------- the I/O wave ----------------
        const int lane = tid - NT;
        float4 gq[NLD];
        auto load_block = [&](int b) {             // gi of block b: TB steps x 300 floats, 16 bytes per lane and load
#pragma unroll
            for (int e = 0; e < NLD; ++e) {
                const int idx = lane + 64 * e;
                const int sl = idx / (3 * GH / 4);
                const int c4 = idx - sl * (3 * GH / 4);
                const int sidx = b * TB + sl;
                gq[e] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (sl < TB && sidx < T) {
                    int t;
                    const int rw = step_row(sidx, t);
                    gq[e] = *reinterpret_cast<const float4*>(gi + ((int64_t)t * rows + rw) * (6 * GH) + dir * 3 * GH + 4 * c4);
                }
            }
        };
        auto stash_block = [&](int buf) {
#pragma unroll
            for (int e = 0; e < NLD; ++e) {
                const int idx = lane + 64 * e;
                if (idx < TB * 3 * GH / 4) *reinterpret_cast<float4*>(&in_s[buf][0][0] + 4 * idx) = gq[e];
            }
        };
        // SEG: the (row, t) of the step to drain comes from a scalar segment cursor (an LDS lookup in front of every drain makes
        // this wave late for the per-step barrier)
        SegCursor cf{};
        if constexpr (SEG) cf = seg_cursor_begin(seg_k, ch.nseg);
        auto flush_step = [&](int b, int sl) {     // results of step (b, sl): lanes = consecutive units (coalesced rows)
            const int sidx = b * TB + sl;
            if (sidx >= T) return;
            int t;
            int rw;
            if constexpr (SEG) {
                if (sidx >= cf.next) seg_cursor_advance(cf, seg_k, ch.nseg);      // (steps are drained in order)
                const int j = sidx - cf.off;
                t = dir ? cf.k - 1 - j : j;
                rw = row + cf.p;
            } else {
                rw = step_row(sidx, t);
            }
            const int64_t o = (int64_t)t * rows + rw;
            float* yp = y + o * (2 * GH) + dir * GH;
            float* gp = gates + (o * 2 + dir) * (4 * GH);
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int u = lane + 64 * q;
                if (u < GH) {
                    const float4 v = *reinterpret_cast<const float4*>(&out_s[b & 1][sl][u][0]);
                    const float g4 = out_s[b & 1][sl][u][4];
                    yp[u] = v.x;
                    // (the saved gate values are read again by the backward pass only: nontemporal, around the L2)
                    __builtin_nontemporal_store(v.y, &gp[u]);
                    __builtin_nontemporal_store(v.z, &gp[GH + u]);
                    __builtin_nontemporal_store(v.w, &gp[2 * GH + u]);
                    __builtin_nontemporal_store(g4, &gp[3 * GH + u]);
                }
            }
        };
        load_block(0);
        stash_block(0);
        __syncthreads();                           // (A) block 0 operands in LDS, h_0 = 0 written by the recurrence waves
        int step = 0;
        int fl = 0;                                // next step whose results are still to be drained
        SegCursor cs{};                            // SEG: where the next segment starts (the recurrence waves' extra barrier)
        if constexpr (SEG) cs = seg_cursor_begin(seg_k, ch.nseg);
        for (int b = 0; b < nblocks; ++b) {
            if (b + 1 < nblocks) load_block(b + 1);
            for (int sl = 0; sl < TB && step < T; ++sl, ++step) {
                if constexpr (SEG) {
                    if (!(ABL & 32) && step == cs.next) {      // a segment starts: the state is re-initialised
                        seg_cursor_advance(cs, seg_k, ch.nseg);
                        __syncthreads();
                    }
                }
                if (fl < b * TB) {                 // one finished step of an earlier block per step (its block is complete)
                    flush_step(fl / TB, fl % TB);
                    ++fl;
                }
                const bool last = (sl == TB - 1) || (step == T - 1);
                if (last && b + 1 < nblocks) stash_block((b + 1) & 1);
                if (!(ABL & 8)) __syncthreads();   // the recurrence waves' per-step barrier
            }
        }
        for (; fl < T; ++fl) flush_step(fl / TB, fl % TB);
        return;
    }

    // ---------------- the four recurrence waves (lane pair (2u, 2u+1) = hidden unit u x half of the contraction) ----------------
    if constexpr (!SEG) load_weights(w_hh);
#pragma unroll
    for (int k = 0; k < KW / 2; ++k) {
        pin_loaded(wr[k]);
        pin_loaded(wz[k]);
        pin_loaded(wn[k]);
    }
    float bhr = b_hh[uu], bhz = b_hh[GH + uu], bhn = b_hh[2 * GH + uu];
    pin_loaded(bhr);
    pin_loaded(bhz);
    pin_loaded(bhn);
    const bool mine = active && half == 0;
    float hprev = 0.f;
    SegCursor cs{};
    if constexpr (SEG) {
        cs = seg_cursor_begin(seg_k, ch.nseg);
        const int p0 = cs.p < ch.nseg ? cs.p : 0;
        for (int i = tid; i < 2 * (GH + 4); i += NT) (&hs[0][0])[i] = (T > 0 && i < GH) ? seg_init[p0][i] : 0.f;
        hprev = T > 0 ? seg_init[p0][uu] : 0.f;
        fill_unvisited(tid, NT);       // stores nobody waits for (these waves touch no global memory in the time loop)
    } else {
        for (int i = tid; i < 2 * (GH + 4); i += NT) (&hs[0][0])[i] = 0.f;
    }
    __syncthreads();                               // (A)

    __shared__ __attribute__((aligned(16))) float scratch[NT][8];      // write target of lanes that own no unit
    const uint32_t scratch_addr = (uint32_t)(uintptr_t)((__attribute__((address_space(3))) float*)(&scratch[tid][0]));
    f32x4 pend4 = {0.f, 0.f, 0.f, 0.f};
    float pend1 = 0.f;
    uint32_t pend_addr = scratch_addr;
    int step = 0;
    for (int b = 0; b < nblocks; ++b) {
        for (int sl = 0; sl < TB && step < T; ++sl, ++step) {
            const int cur = step & 1;
            if constexpr (SEG) {
                // a new segment: the state (LDS copy and register) becomes the segment's start state; the test is a scalar compare
                if (!(ABL & 32) && step == cs.next) {
                    seg_cursor_advance(cs, seg_k, ch.nseg);
                    hprev = seg_init[cs.p][uu];
                    if (mine) hs[cur][u] = hprev;
                    __syncthreads();
                }
            }
            // LDS operations of a step in ISSUE order (a wave's LDS operations execute in order, so anything queued in front of
            // the h loads delays the first FMA by its whole service time): six h loads, then the matvec with its reloads, and
            // only behind the LAST h load the three gate operands of this step and the two result writes of the previous
            // one -- all from inline asm, so the hand-counted lgkmcnt waits below are exact.  Lanes that own no unit write to
            // a scratch slot (no exec-mask branch in the step).
            const uint32_t gaddr = (uint32_t)(uintptr_t)((__attribute__((address_space(3))) const float*)(&in_s[b & 1][sl][uu]));
            float g0, g1, g2;
            f32x2 ar0 = {0.f, 0.f}, az0 = {0.f, 0.f}, an0 = {0.f, 0.f}, ar1 = {0.f, 0.f}, az1 = {0.f, 0.f}, an1 = {0.f, 0.f};
            if (ABL & 1) {
                ar0[0] = hs[cur][u & 63];
                const float* gp = &in_s[b & 1][sl][uu];
                g0 = gp[0]; g1 = gp[GH]; g2 = gp[2 * GH];
            } else {
                // 13 broadcast loads of h_{t-1} through a ring of six float4 registers, six groups ahead of their FMAs
                constexpr int HLOOK = 6;
                const uint32_t haddr = (uint32_t)(uintptr_t)((__attribute__((address_space(3))) const float*)(&hs[cur][kbase]));
                f32x4 hb[HLOOK];
#define GRU_LDS_RD(J)                                                                                        \
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(hb[(J) % HLOOK]) : "v"(haddr), "i"(16 * (J)) : "memory")
#define GRU_LDS_WAIT(N, J) asm volatile("s_waitcnt lgkmcnt(" #N ")" : "+v"(hb[(J) % HLOOK]) : : "memory")
#define GRU_FMA1(ACC, W, H) asm("v_fma_f32 %0, %1, %2, %0" : "+v"(ACC) : "v"(W), "v"(H))
#define GRU_GROUP(K4)                                                                                        \
    do {                                                                                                     \
        const f32x4 h4 = hb[(K4) % HLOOK];                                                                   \
        if (SCALAR_FMA) {   /* plain v_fma_f32 (not SLP-packed): A/B against v_pk_fma_f32 */                 \
            GRU_FMA1(ar0[0], wr[2 * (K4)][0], h4.x); GRU_FMA1(az0[0], wz[2 * (K4)][0], h4.x); GRU_FMA1(an0[0], wn[2 * (K4)][0], h4.x); \
            GRU_FMA1(ar0[1], wr[2 * (K4)][1], h4.y); GRU_FMA1(az0[1], wz[2 * (K4)][1], h4.y); GRU_FMA1(an0[1], wn[2 * (K4)][1], h4.y); \
            GRU_FMA1(ar1[0], wr[2 * (K4) + 1][0], h4.z); GRU_FMA1(az1[0], wz[2 * (K4) + 1][0], h4.z); GRU_FMA1(an1[0], wn[2 * (K4) + 1][0], h4.z); \
            GRU_FMA1(ar1[1], wr[2 * (K4) + 1][1], h4.w); GRU_FMA1(az1[1], wz[2 * (K4) + 1][1], h4.w); GRU_FMA1(an1[1], wn[2 * (K4) + 1][1], h4.w); \
        } else {                                                                                             \
            const f32x2 ha = {h4.x, h4.y}, hb2 = {h4.z, h4.w};                                               \
            ar0 = __builtin_elementwise_fma(wr[2 * (K4)], ha, ar0);                                          \
            az0 = __builtin_elementwise_fma(wz[2 * (K4)], ha, az0);                                          \
            an0 = __builtin_elementwise_fma(wn[2 * (K4)], ha, an0);                                          \
            ar1 = __builtin_elementwise_fma(wr[2 * (K4) + 1], hb2, ar1);                                     \
            az1 = __builtin_elementwise_fma(wz[2 * (K4) + 1], hb2, az1);                                     \
            an1 = __builtin_elementwise_fma(wn[2 * (K4) + 1], hb2, an1);                                     \
        }                                                                                                    \
    } while (0)
                GRU_LDS_RD(0); GRU_LDS_RD(1); GRU_LDS_RD(2); GRU_LDS_RD(3); GRU_LDS_RD(4); GRU_LDS_RD(5);
                GRU_LDS_WAIT(5, 0); GRU_GROUP(0); GRU_LDS_RD(6);
                GRU_LDS_WAIT(5, 1); GRU_GROUP(1); GRU_LDS_RD(7);
                GRU_LDS_WAIT(5, 2); GRU_GROUP(2); GRU_LDS_RD(8);
                GRU_LDS_WAIT(5, 3); GRU_GROUP(3); GRU_LDS_RD(9);
                GRU_LDS_WAIT(5, 4); GRU_GROUP(4); GRU_LDS_RD(10);
                GRU_LDS_WAIT(5, 5); GRU_GROUP(5); GRU_LDS_RD(11);
                GRU_LDS_WAIT(5, 6); GRU_GROUP(6); GRU_LDS_RD(12);
                // five more LDS operations behind the last h load: g0 g1 g2 of this step, the previous step's results
                if (ABL & 16) { g0 = 0.1f; g1 = 0.2f; g2 = 0.3f; }
                asm volatile("ds_read_b32 %0, %3\n\tds_read_b32 %1, %3 offset:400\n\tds_read_b32 %2, %3 offset:800"
                             : "=&v"(g0), "=&v"(g1), "=&v"(g2) : "v"(gaddr) : "memory");
                asm volatile("ds_write_b128 %0, %1\n\tds_write_b32 %0, %2 offset:16" : : "v"(pend_addr), "v"(pend4), "v"(pend1) : "memory");
                GRU_LDS_WAIT(10, 7); GRU_GROUP(7);
                GRU_LDS_WAIT(9, 8); GRU_GROUP(8);
                GRU_LDS_WAIT(8, 9); GRU_GROUP(9);
                GRU_LDS_WAIT(7, 10); GRU_GROUP(10);
                GRU_LDS_WAIT(6, 11); GRU_GROUP(11);
                GRU_LDS_WAIT(5, 12); GRU_GROUP(12);
                asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(g0), "+v"(g1), "+v"(g2) : : "memory");     // the two writes may still drain
#undef GRU_GROUP
#undef GRU_FMA1
#undef GRU_LDS_WAIT
#undef GRU_LDS_RD
            }
            float ar = (ar0.x + ar0.y) + (ar1.x + ar1.y);
            float az = (az0.x + az0.y) + (az1.x + az1.y);
            float an = (an0.x + an0.y) + (an1.x + an1.y);
            ar += pair_swap(ar);
            az += pair_swap(az);
            an += pair_swap(an);
            // (ABL & 64, timing only: the step as it would be with -log2(e) folded into W_hr / W_hz, 2 log2(e) into W_hn and the
            // recurrent biases pre-added to the staged operands -- the upper bound of the "constant folding" lead)
            const float ghn = (ABL & 64) ? an : an + bhn;
            const float rr = (ABL & 64) ? __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(g0 + ar))
                           : (ABL & 2) ? (g0 + ar + bhr) * 0.01f : sigmoidf_(g0 + ar + bhr);
            const float zz = (ABL & 64) ? __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(g1 + az))
                           : (ABL & 2) ? (g1 + az + bhz) * 0.01f : sigmoidf_(g1 + az + bhz);
            const float nn = (ABL & 64) ? 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(fmaf(rr, ghn, g2)))
                           : (ABL & 2) ? (g2 + rr * ghn) * 0.01f : tanhf_(g2 + rr * ghn);
            const float hnew = (1.0f - zz) * nn + zz * hprev;
            if (mine) hs[cur ^ 1][u] = hnew;
            hprev = hnew;
            // results: owners -> out_s[block][step][unit][0..4]; every other lane -> its private scratch slot.  Kept in
            // registers across the barrier and written from the middle of the next step's matvec; the last step of a block
            // (whose results the I/O wave drains right behind this barrier) writes them now.
            const uint32_t oaddr = mine ? (uint32_t)(uintptr_t)((__attribute__((address_space(3))) float*)(&out_s[b & 1][sl][u][0]))
                                        : scratch_addr;
            const bool last = (sl == TB - 1) || (step == T - 1);
            pend4 = f32x4{hnew, rr, zz, nn};
            pend1 = ghn;
            pend_addr = oaddr;
            if (last) {
                asm volatile("ds_write_b128 %0, %1\n\tds_write_b32 %0, %2 offset:16" : : "v"(pend_addr), "v"(pend4), "v"(pend1) : "memory");
                pend_addr = scratch_addr;          // (the next step's deferred write then lands in the scratch slot)
            }
            if (!(ABL & 8)) __syncthreads();
        }
    }
}

template <int SCALAR_FMA, int ABL, int SEG>
__global__ __launch_bounds__(320) void gru_seq_fwd_io_kernel(FwdGroups G) {
    gru_fwd_io_body<SCALAR_FMA, ABL, SEG>(G, (int)blockIdx.x, (int)blockIdx.y);
}

// The plain forward launch WITH the step's dropout-flag draw aboard: workgroups [0, 2 x slots) run the recurrences, the
// workgroups behind them (at most one per idle CU: the kernel's register budget keeps a CU to one workgroup) draw the keep flags
// of the whole step (keep_flags_body.h) -- the flags' first consumer is the dropout BEHIND this recurrence (nn.GRU's
// inter-layer dropout, reference model.py:866), so the generator launch that used to sit between the two GRU layers (7.7 us at
// cfg2) is gone.  Same flags as the launch of its own: which counter yields which flag depends on neither the grid nor the block size.
__global__ __launch_bounds__(320) void gru_seq_fwd_io_flags_kernel(const FwdGroups G, const kfb::FlagJob J, const int nslots) {
    const int bid = (int)blockIdx.x;
    if (bid < 2 * nslots) gru_fwd_io_body<0, 0, 0>(G, bid % nslots, bid / nslots);
    else kfb::keep_flags_block<320>(J, bid - 2 * nslots, (int)gridDim.x - 2 * nslots);
}

// Backward through time.  dh_prev[u] = sum_j dgh[j] W_hh[j][u] + dh z: lane (u, half) keeps W_hh[j][u] for
// j in its half of the 3H gate rows ([0,152) / [152,300)), dgh of the current step is broadcast from LDS.
constexpr int JH0 = 152;
constexpr int JW = 152;

// SEG = 1 (R = 1 only): segmented launch, see SegInfo and gru_seq_bwd_kpart_kernel.  The recurrent term of a step is formed
// from the previous step's dgh anyway: where a segment ended it IS the gradient wrt that segment's start state (written to
// dhinit when wanted) and the step itself starts from zero.
template <int R, int SEG>
__global__ __launch_bounds__(NT) void gru_seq_bwd_kernel(BwdGroups G) {
    static_assert(!SEG || R == 1, "segmented launches run one row per workgroup");
    constexpr int TB = 8 / R;
    constexpr int IN4 = 6 * GH / 4;     // staged per (step,row): dy (GH) | r z n ghn (4 GH) | h_prev (GH)
    constexpr int OUT4 = 6 * GH / 4;    // dgi (3 GH) | dgh (3 GH)
    constexpr int NIN = (TB * R * IN4 + NT - 1) / NT;
    __shared__ __attribute__((aligned(16))) float dghs[2][R][3 * GH + 4];
    __shared__ __attribute__((aligned(16))) float in_s[2][TB][R][6 * GH];
    __shared__ __attribute__((aligned(16))) float out_s[TB][R][6 * GH];
    __shared__ uint32_t sched[SEG ? SCHED_MAX : 1];
    __shared__ int seg_k[MAXSEG];

    int gidx = 0;
    int dir_ = 0, row_ = 0, T_ = 0;
    Chain ch{};
    const int tid = threadIdx.x;
    const int u = tid >> 1;
    const int half = tid & 1;
    const bool active = u < GH;
    const int uu = active ? u : GH - 1;
    const int jbase = half ? JH0 : 0;
    const int jlen = half ? 3 * GH - JH0 : JH0;
    f32x2 w[JW / 2];
    auto load_weights = [&](const float* __restrict__ wsrc) {
#pragma unroll
        for (int j = 0; j < JW; ++j) w[j >> 1][j & 1] = (j < jlen) ? wsrc[(int64_t)(jbase + (j < jlen ? j : 0)) * GH + uu] : 0.f;
    };
    if constexpr (SEG) {
        ch = seg_decode(G.seg);
        gidx = ch.gidx;
        dir_ = ch.dir;
        row_ = ch.row0;
        load_weights(G.w_hh[2 * gidx + dir_]);     // in flight while the steps are counted
        seg_schedule<NT>(G.seg, G.T, ch, sched, seg_k);
        T_ = ch.S;                                 // the time loop runs over the flattened schedule, backwards
    } else {
        while (gidx + 1 < G.n && (int)blockIdx.x >= G.slice0[gidx + 1]) ++gidx;
        dir_ = blockIdx.y;
        row_ = ((int)blockIdx.x - G.slice0[gidx]) * R;
        T_ = G.T[gidx];
    }
    const int dir = dir_;
    const int rows = G.rows[gidx];
    const int T = T_;
    const int row0 = row_;
    const int Tfull = G.T[gidx];
    float* __restrict__ dhinit = SEG ? G.dhinit[gidx] : nullptr;
    // (row, t) of backward step sidx
    auto step_row = [&](int sidx, int r, int& t) {
        if constexpr (SEG) {
            const uint32_t e = sched[T - 1 - sidx];
            t = (int)(e & 0xFFFu);
            return row0 + (int)((e >> 12) & 0xFu);
        } else {
            t = dir ? sidx : T - 1 - sidx;
            return row0 + r;
        }
    };
    const float* __restrict__ dy = G.dy[gidx];
    const float* __restrict__ y = G.y[gidx];
    const float* __restrict__ gates = G.gates[gidx];
    const float* __restrict__ w_hh = G.w_hh[2 * gidx + dir];
    float* __restrict__ dgi = G.dgi[gidx];
    float* __restrict__ dgh = G.dgh[gidx];

    // SEG: step counts / zero start-state gradients of rows without steps, zero dgi / dgh at the positions never visited
    auto finish_unvisited = [&]() {
        if (!ch.has_rank) return;
        int32_t* __restrict__ kout = G.kout[gidx];
        if (ch.trunc && dhinit != nullptr) {
            for (int idx = tid; idx < ch.nseg * GH; idx += NT) {
                const int p = idx / GH;
                if (seg_k[p] == 0) dhinit[(int64_t)(row0 + p) * GH + (idx - p * GH)] = 0.f;
            }
        }
        if (ch.trunc && kout != nullptr && tid < ch.nseg) kout[row0 + tid] = seg_k[tid];
        const int per = Tfull * (3 * GH / 4);
        for (int idx = tid; idx < ch.nseg * per; idx += NT) {
            const int p = idx / per;
            const int rem = idx - p * per;
            const int t = rem / (3 * GH / 4);
            const int c4 = rem - t * (3 * GH / 4);
            if (t < seg_k[p]) continue;
            const int64_t o = ((int64_t)t * rows + row0 + p) * (6 * GH) + dir * 3 * GH + 4 * c4;
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
            *reinterpret_cast<f32x4*>(dgi + o) = z;
            __builtin_nontemporal_store(z, reinterpret_cast<f32x4*>(dgh + o));
        }
    };
    if constexpr (SEG) {
        finish_unvisited();            // (up front: the stores drain under the prologue and the first steps instead of at the end)
        if (T == 0) return;            // nothing to run
    }
    if constexpr (!SEG) load_weights(w_hh);
#pragma unroll
    for (int j = 0; j < JW / 2; ++j) pin_loaded(w[j]);

    bool mine[R];
    float carry[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        mine[r] = active && ((r & 1) == half) && (row0 + r < rows);
        carry[r] = 0.f;
    }

    // step index s runs in the REVERSE of the forward order: t(s) = dir ? s : T-1-s
    const int nblocks = (T + TB - 1) / TB;
    float4 stage[NIN];
    auto load_block = [&](int b) {
#pragma unroll
        for (int e = 0; e < NIN; ++e) {
            const int idx = tid + e * NT;
            const int sl = idx / (R * IN4);
            const int rem = idx - sl * (R * IN4);
            const int r = rem / IN4;
            const int c4 = rem - r * IN4;
            const int sidx = b * TB + sl;
            stage[e] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (sl < TB && sidx < T && row0 + r < rows) {
                int t;
                const int rw = step_row(sidx, r, t);
                const int64_t o = ((int64_t)t * rows + rw);
                if (c4 < GH / 4) {
                    stage[e] = *reinterpret_cast<const float4*>(dy + o * (2 * GH) + dir * GH + 4 * c4);
                } else if (c4 < 5 * GH / 4) {
                    stage[e] = *reinterpret_cast<const float4*>(gates + (o * 2 + dir) * (4 * GH) + 4 * (c4 - GH / 4));
                } else {
                    const int tp = dir ? t + 1 : t - 1;   // the step that produced h_prev in the forward pass
                    if (tp >= 0 && tp < Tfull)
                        stage[e] = *reinterpret_cast<const float4*>(y + ((int64_t)tp * rows + rw) * (2 * GH) + dir * GH +
                                                                    4 * (c4 - 5 * GH / 4));
                }
            }
        }
    };
    auto stash_block = [&](int buf) {
#pragma unroll
        for (int e = 0; e < NIN; ++e) {
            const int idx = tid + e * NT;
            if (idx < TB * R * IN4) *reinterpret_cast<float4*>(&in_s[buf][0][0][0] + 4 * idx) = stage[e];
        }
    };
    auto flush_block = [&](int b) {
        for (int idx = tid; idx < TB * R * OUT4; idx += NT) {
            const int sl = idx / (R * OUT4);
            const int rem = idx - sl * (R * OUT4);
            const int r = rem / OUT4;
            const int c4 = rem - r * OUT4;
            const int sidx = b * TB + sl;
            if (sidx >= T || row0 + r >= rows) continue;
            int t;
            const int rw = step_row(sidx, r, t);
            const float4 v = *reinterpret_cast<const float4*>(&out_s[sl][r][4 * c4]);
            const int64_t o = ((int64_t)t * rows + rw) * (6 * GH) + dir * 3 * GH;
            if (c4 < 3 * GH / 4)
                *reinterpret_cast<float4*>(dgi + o + 4 * c4) = v;
            else
                __builtin_nontemporal_store(__builtin_bit_cast(f32x4, v), reinterpret_cast<f32x4*>(dgh + o + 4 * (c4 - 3 * GH / 4)));   // (read by the weight-gradient batch only)
        }
    };

    for (int i = tid; i < 2 * R * (3 * GH + 4); i += NT) (&dghs[0][0][0])[i] = 0.f;
    load_block(0);
    stash_block(0);
    if (nblocks > 1) load_block(1);
    __syncthreads();

    int step = 0;
    uint32_t e_succ = 0;
    for (int b = 0; b < nblocks; ++b) {
        const int buf = b & 1;
        for (int sl = 0; sl < TB && step < T; ++sl, ++step) {
            const int cur = step & 1;
            // SEG: sched[T - step] is this step's forward-order successor; if it opened a segment, that segment ended with the
            // previous backward step
            bool bnd = false;
            int prow = 0;
            if constexpr (SEG) {
                bnd = step > 0 && ((e_succ >> 16) & 1u);       // (e_succ = sched[T - step], fetched during the previous step)
                prow = row0 + (int)((e_succ >> 12) & 0xFu);
            }
#pragma unroll
            for (int r = 0; r < R; ++r) {
                // recurrent contribution from the step processed just before (its dgh sits in dghs[cur^1])
                f32x2 a0 = {0.f, 0.f}, a1 = {0.f, 0.f}, a2 = {0.f, 0.f}, a3 = {0.f, 0.f};
                const float4* dv = reinterpret_cast<const float4*>(&dghs[cur ^ 1][r][jbase]);
#pragma unroll
                for (int j8 = 0; j8 < JW / 8; ++j8) {
                    const float4 d4 = dv[2 * j8], e4 = dv[2 * j8 + 1];   // second half: 148 real values + 4 zero pad floats
                    const f32x2 da = {d4.x, d4.y}, db = {d4.z, d4.w}, dc = {e4.x, e4.y}, dd = {e4.z, e4.w};
                    a0 = __builtin_elementwise_fma(w[4 * j8 + 0], da, a0);
                    a1 = __builtin_elementwise_fma(w[4 * j8 + 1], db, a1);
                    a2 = __builtin_elementwise_fma(w[4 * j8 + 2], dc, a2);
                    a3 = __builtin_elementwise_fma(w[4 * j8 + 3], dd, a3);
                }
                float rec = ((a0.x + a0.y) + (a1.x + a1.y)) + ((a2.x + a2.y) + (a3.x + a3.y));
                rec += pair_swap(rec);
                float dh0 = 0.f;
                if constexpr (SEG) {
                    // branch-free in the step's body (a branch here splits the block the operand reads are scheduled in); the
                    // rare store of the finished segment's start-state gradient follows at the end of the step
                    dh0 = carry[r] + rec;
                    rec = bnd ? 0.f : rec;
                    carry[r] = bnd ? 0.f : carry[r];
                }
                if (mine[r]) {
                    const float* ip = &in_s[buf][sl][r][u];
                    const float dh = ip[0] + carry[r] + rec;
                    const float rr = ip[GH], zz = ip[2 * GH], nn = ip[3 * GH], ghn = ip[4 * GH], hprev = ip[5 * GH];
                    const float dn = dh * (1.0f - zz);
                    const float dz = dh * (hprev - nn);
                    carry[r] = dh * zz;
                    const float dnpre = dn * (1.0f - nn * nn);
                    const float drpre = dnpre * ghn * rr * (1.0f - rr);
                    const float dzpre = dz * zz * (1.0f - zz);
                    const float dghn = dnpre * rr;
                    float* op = &out_s[sl][r][u];
                    op[0] = drpre;
                    op[GH] = dzpre;
                    op[2 * GH] = dnpre;
                    op[3 * GH] = drpre;
                    op[4 * GH] = dzpre;
                    op[5 * GH] = dghn;
                    dghs[cur][r][u] = drpre;
                    dghs[cur][r][GH + u] = dzpre;
                    dghs[cur][r][2 * GH + u] = dghn;
                }
                if constexpr (SEG) {
                    if (bnd && mine[r] && dhinit != nullptr && ch.trunc) dhinit[(int64_t)prow * GH + u] = dh0;
                }
            }
            if constexpr (SEG) e_succ = sched[T - 1 - step];
            __syncthreads();
        }
        if (b + 1 < nblocks) stash_block(buf ^ 1);
        flush_block(b);
        if (b + 2 < nblocks) load_block(b + 2);
        __syncthreads();
    }
    if constexpr (SEG) {
        if (T > 0 && dhinit != nullptr && ch.trunc) {
            // the segment the forward pass ran first: its start-state gradient is the recurrent term one more step would use
            const int cur = T & 1;
            f32x2 a0 = {0.f, 0.f}, a1 = {0.f, 0.f}, a2 = {0.f, 0.f}, a3 = {0.f, 0.f};
            const float4* dv = reinterpret_cast<const float4*>(&dghs[cur ^ 1][0][jbase]);
#pragma unroll
            for (int j8 = 0; j8 < JW / 8; ++j8) {
                const float4 d4 = dv[2 * j8], e4 = dv[2 * j8 + 1];
                const f32x2 da = {d4.x, d4.y}, db = {d4.z, d4.w}, dc = {e4.x, e4.y}, dd = {e4.z, e4.w};
                a0 = __builtin_elementwise_fma(w[4 * j8 + 0], da, a0);
                a1 = __builtin_elementwise_fma(w[4 * j8 + 1], db, a1);
                a2 = __builtin_elementwise_fma(w[4 * j8 + 2], dc, a2);
                a3 = __builtin_elementwise_fma(w[4 * j8 + 3], dd, a3);
            }
            float rec = ((a0.x + a0.y) + (a1.x + a1.y)) + ((a2.x + a2.y) + (a3.x + a3.y));
            rec += pair_swap(rec);
            if (mine[0]) dhinit[(int64_t)(row0 + (int)((sched[0] >> 12) & 0xFu)) * GH + u] = carry[0] + rec;
        }
    }
}

// =====================================================================================================
// Backward pass, one sequence per workgroup: gate rows PARTITIONED OVER THE WAVES, operands broadcast through SGPRs.
//
// The lane-pair kernel above delivers the 300-entry dgh vector to every lane through LDS: 38 ds_read_b128 per lane and
// step, i.e. 4 waves x 38 KB = 152 KB through the CU's 128 B/clk LDS return path -- half of the measured step time -- and
// with the weight slices taking 152 VGPRs the compiler keeps only two of those loads in flight.  Here each wave owns a
// slice of the gate rows and produces its operands ITSELF: wave w owns rows j in [38w, 38w+38); lane l accumulates the
// partial dh_prev of units l and l + 64 over those rows -- dgh[j] comes from lane (j - 38w) of the same wave through
// v_readlane (an SGPR operand of the FMA: no LDS traffic, no load latency) -- and drops the two partials in LDS; after ONE
// barrier lane l < 38 rebuilds dh of unit (38w + l) mod 100 from the eight partials and computes the one pre-activation
// gradient dgh[38w + l] its wave needs next (the carry dh z is replicated in the up to three lanes that share a unit:
// same inputs, same order, bit-identical).  LDS traffic per step: 0.5 KB of partials per wave; weight slices 76 VGPRs;
// eight waves = two per SIMD (one wave per SIMD runs at ~10 cycles per instruction on this chain whatever the mix).
// Same time-blocked global staging as above.  Used when every sequence gets its own workgroup in one round
// (pick_r() == 1); larger batches keep the R = 2 / 4 kernels.  Variants and measurements: profiles/r02_gru_kernels.md.
// =====================================================================================================
constexpr int PU = 128;              // partial-sum row length, backward (100 units, padded)

__device__ __forceinline__ float lane_bcast(float v, int lane) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}


// NW waves per workgroup (8: two per SIMD, half the instructions per wave and a second wave to issue from while the
// first waits on a dependent result); GRP operands are fetched per group of FMAs
// SEG = 1: segmented launch (see SegInfo).  The flattened schedule is walked backwards; where a segment ends in that order the
// recurrent gradient is cut (and, for a direction that started from the all-padding sequence's state, the gradient wrt that
// start state is written to dhinit: one more matvec + exchange per segment); the (t, row) positions a truncated / silent row
// never visited get zero dgi / dgh afterwards (they are operands of the weight-gradient contractions and column sums).
// The body of one workgroup (bx, by) = (sequence slot, direction).  DYN = false: the kernel's own static LDS; DYN = true: the
// arrays are carved from `dyn` (the launch's dynamic LDS) -- the form the rider launch needs, whose LDS is shared with the
// weight-gradient tiles of the rider workgroups (gru_seq_bwd_riders_kernel).
template <int NW, int GRP, int SEG, bool DYN>
__device__ __forceinline__ void gru_bwd_kpart_body(const BwdGroups& G, const int bx, const int by, unsigned char* dyn) {
    constexpr int NTK = 64 * NW;
    constexpr int JPW = (3 * GH + NW - 1) / NW;     // gate rows per wave (75 / 38)
    constexpr int NGR = (JPW + 63) / 64;            // gate rows per lane in the gate stage (2 / 1)
    constexpr int TB = 8;
    constexpr int IN4 = 6 * GH / 4;     // staged per step: dy (GH) | r z n ghn (4 GH) | h_prev (GH)
    constexpr int OUT4 = 6 * GH / 4;    // dgi (3 GH) | dgh (3 GH)
    constexpr int NIN = (TB * IN4 + NTK - 1) / NTK;
    float (*in_s)[TB][6 * GH];
    float (*out_s)[6 * GH];
    float (*part)[NW][PU];
    uint32_t* sched;
    int* seg_k;
    if constexpr (DYN) {
        static_assert(!SEG, "the rider launch carries plain (full-length) recurrences only");
        in_s = reinterpret_cast<float (*)[TB][6 * GH]>(dyn);
        out_s = reinterpret_cast<float (*)[6 * GH]>(dyn + sizeof(float) * 2 * TB * 6 * GH);
        part = reinterpret_cast<float (*)[NW][PU]>(dyn + sizeof(float) * 3 * TB * 6 * GH);
        sched = nullptr;
        seg_k = nullptr;
    } else {
        __shared__ __attribute__((aligned(16))) float in_static[2][TB][6 * GH];
        __shared__ __attribute__((aligned(16))) float out_static[TB][6 * GH];
        __shared__ __attribute__((aligned(16))) float part_static[SEG ? 3 : 2][NW][PU];
        __shared__ uint32_t sched_static[SEG ? SCHED_MAX : 1];
        __shared__ int seg_k_static[MAXSEG];
        in_s = in_static; out_s = out_static; part = part_static; sched = sched_static; seg_k = seg_k_static;
    }

    int gidx = 0;
    int dir_ = 0, row_ = 0, T_ = 0;
    Chain ch{};
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int j0 = JPW * wv;
    // W_hh[j0 + jj][u] for the lane's two units u = lane, lane + 64 (< 100 for lane < 36); rows past 3H-1 carry zeros
    const bool has1 = lane + 64 < GH;
    f32x2 w[JPW];
    auto load_weights = [&](const float* __restrict__ wsrc) {
#pragma unroll
        for (int jj = 0; jj < JPW; ++jj) {
            const bool jok = j0 + jj < 3 * GH;
            const int jc = jok ? j0 + jj : 3 * GH - 1;
            w[jj][0] = jok ? wsrc[(int64_t)jc * GH + lane] : 0.f;
            w[jj][1] = (jok && has1) ? wsrc[(int64_t)jc * GH + lane + 64] : 0.f;
        }
    };
    if constexpr (SEG) {
        ch = seg_decode(G.seg);
        gidx = ch.gidx;
        dir_ = ch.dir;
        row_ = ch.row0;
        load_weights(G.w_hh[2 * gidx + dir_]);     // in flight while the steps are counted
        seg_schedule<NTK>(G.seg, G.T, ch, sched, seg_k);
        T_ = ch.S;                                 // the time loop runs over the flattened schedule, backwards
    } else {
        while (gidx + 1 < G.n && bx >= G.slice0[gidx + 1]) ++gidx;
        dir_ = by;
        row_ = bx - G.slice0[gidx];
        T_ = G.T[gidx];
    }
    const int dir = dir_;
    const int rows = G.rows[gidx];
    const int T = T_;
    const int row = row_;
    const int Tfull = G.T[gidx];
    // (row, t) of backward step sidx
    auto step_row = [&](int sidx, int& t) {
        if constexpr (SEG) {
            const uint32_t e = sched[T - 1 - sidx];
            t = (int)(e & 0xFFFu);
            return row + (int)((e >> 12) & 0xFu);
        } else {
            t = dir ? sidx : T - 1 - sidx;
            return row;
        }
    };
    const float* __restrict__ dy = G.dy[gidx];
    const float* __restrict__ y = G.y[gidx];
    const float* __restrict__ gates = G.gates[gidx];
    const float* __restrict__ w_hh = G.w_hh[2 * gidx + dir];
    float* __restrict__ dgi = G.dgi[gidx];
    float* __restrict__ dgh = G.dgh[gidx];

    float* __restrict__ dhinit = SEG ? G.dhinit[gidx] : nullptr;
    // SEG: step counts / zero start-state gradients of rows without steps, zero dgi / dgh at the positions never visited
    auto finish_unvisited = [&]() {
        if (!ch.has_rank) return;
        int32_t* __restrict__ kout = G.kout[gidx];
        if (ch.trunc && dhinit != nullptr) {
            for (int idx = tid; idx < ch.nseg * GH; idx += NTK) {
                const int p = idx / GH;
                if (seg_k[p] == 0) dhinit[(int64_t)(row + p) * GH + (idx - p * GH)] = 0.f;
            }
        }
        if (ch.trunc && kout != nullptr && tid < ch.nseg) kout[row + tid] = seg_k[tid];
        const int per = Tfull * (3 * GH / 4);
        for (int idx = tid; idx < ch.nseg * per; idx += NTK) {
            const int p = idx / per;
            const int rem = idx - p * per;
            const int t = rem / (3 * GH / 4);
            const int c4 = rem - t * (3 * GH / 4);
            if (t < seg_k[p]) continue;
            const int64_t o = ((int64_t)t * rows + row + p) * (6 * GH) + dir * 3 * GH + 4 * c4;
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
            *reinterpret_cast<f32x4*>(dgi + o) = z;
            __builtin_nontemporal_store(z, reinterpret_cast<f32x4*>(dgh + o));
        }
    };
    if constexpr (SEG) {
        finish_unvisited();            // (up front: the stores drain under the prologue and the first steps instead of at the end)
        if (T == 0) return;            // nothing to run
    }
    if constexpr (!SEG) load_weights(w_hh);
#pragma unroll
    for (int jj = 0; jj < JPW; ++jj) pin_loaded(w[jj]);
    // gate stage: lane l handles gate row j0 + l (and j0 + 64 + l when a wave owns more than 64 rows)
    int jr[NGR], gg[NGR], uu[NGR];
    bool has[NGR];
    float dv[NGR], carry[NGR];          // dgh[jr] of the previously processed step; dh z of unit uu
#pragma unroll
    for (int q = 0; q < NGR; ++q) {
        const int jl = 64 * q + lane;
        has[q] = jl < JPW && j0 + jl < 3 * GH;
        jr[q] = has[q] ? j0 + jl : 0;
        gg[q] = jr[q] / GH;
        uu[q] = jr[q] - gg[q] * GH;
        dv[q] = 0.f;
        carry[q] = 0.f;
    }

    const int nblocks = (T + TB - 1) / TB;
    float4 stage[NIN];
    auto load_block = [&](int b) {
#pragma unroll
        for (int e = 0; e < NIN; ++e) {
            const int idx = tid + e * NTK;
            const int sl = idx / IN4;
            const int c4 = idx - sl * IN4;
            const int sidx = b * TB + sl;
            stage[e] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (sl < TB && sidx < T) {
                int t;
                const int rw = step_row(sidx, t);
                const int64_t o = (int64_t)t * rows + rw;
                if (c4 < GH / 4) {
                    stage[e] = *reinterpret_cast<const float4*>(dy + o * (2 * GH) + dir * GH + 4 * c4);
                } else if (c4 < 5 * GH / 4) {
                    stage[e] = *reinterpret_cast<const float4*>(gates + (o * 2 + dir) * (4 * GH) + 4 * (c4 - GH / 4));
                } else {
                    // the step that produced h_prev in the forward pass (a truncated reverse row: position k holds the copy of
                    // the all-padding sequence's output it started from; a truncated forward row starts from 0 at t = 0)
                    const int tp = dir ? t + 1 : t - 1;
                    if (tp >= 0 && tp < Tfull)
                        stage[e] = *reinterpret_cast<const float4*>(y + ((int64_t)tp * rows + rw) * (2 * GH) + dir * GH +
                                                                    4 * (c4 - 5 * GH / 4));
                }
            }
        }
    };
    auto stash_block = [&](int buf) {
#pragma unroll
        for (int e = 0; e < NIN; ++e) {
            const int idx = tid + e * NTK;
            if (idx < TB * IN4) *reinterpret_cast<float4*>(&in_s[buf][0][0] + 4 * idx) = stage[e];
        }
    };
    auto flush_block = [&](int b) {
        for (int idx = tid; idx < TB * OUT4; idx += NTK) {
            const int sl = idx / OUT4;
            const int c4 = idx - sl * OUT4;
            const int sidx = b * TB + sl;
            if (sidx >= T) continue;
            int t;
            const int rw = step_row(sidx, t);
            const float4 v = *reinterpret_cast<const float4*>(&out_s[sl][4 * c4]);
            const int64_t o = ((int64_t)t * rows + rw) * (6 * GH) + dir * 3 * GH;
            if (c4 < 3 * GH / 4)
                *reinterpret_cast<float4*>(dgi + o + 4 * c4) = v;
            else
                __builtin_nontemporal_store(__builtin_bit_cast(f32x4, v), reinterpret_cast<f32x4*>(dgh + o + 4 * (c4 - 3 * GH / 4)));   // (read by the weight-gradient batch only)
        }
    };

    load_block(0);
    stash_block(0);
    if (nblocks > 1) load_block(1);
    __syncthreads();

    // the segment of row prow is complete: its gradient wrt the start state is carry + W_hh^T dgh(last step) -- the
    // recurrent term the next step would have used -- written out when wanted; then the recurrence is cut
    auto seg_end = [&](int prow) {
        if (dhinit != nullptr && ch.trunc) {
            f32x2 acc0 = {0.f, 0.f}, acc1 = {0.f, 0.f};
#pragma unroll
            for (int jj = 0; jj < JPW; ++jj) {
                const float dj = lane_bcast(dv[jj / 64], jj % 64);
                const f32x2 dd = {dj, dj};
                if (jj & 1) acc1 = __builtin_elementwise_fma(w[jj], dd, acc1);
                else acc0 = __builtin_elementwise_fma(w[jj], dd, acc0);
            }
            constexpr int PX = SEG ? 2 : 0;
            part[PX][wv][lane] = acc0[0] + acc1[0];
            part[PX][wv][lane + 64] = acc0[1] + acc1[1];
            __syncthreads();
#pragma unroll
            for (int q = 0; q < NGR; ++q) {
                float pr[NW];
#pragma unroll
                for (int w2 = 0; w2 < NW; ++w2) pr[w2] = part[PX][w2][uu[q]];
#pragma unroll
                for (int sp = 1; sp < NW; sp *= 2)
#pragma unroll
                    for (int w2 = 0; w2 + sp < NW; w2 += 2 * sp) pr[w2] += pr[w2 + sp];
                if (has[q] && gg[q] == 0) dhinit[(int64_t)prow * GH + uu[q]] = carry[q] + pr[0];
            }
        }
#pragma unroll
        for (int q = 0; q < NGR; ++q) {
            dv[q] = 0.f;
            carry[q] = 0.f;
        }
    };

    int step = 0;
    int seg_p = 0, bnd = 0x7fffffff;        // SEG: the segment being walked (last first) and the backward step at which it ends
    if constexpr (SEG) {
        int q = ch.nseg - 1;
        while (q >= 0 && seg_k[q] == 0) --q;
        int q2 = q - 1;
        while (q2 >= 0 && seg_k[q2] == 0) --q2;
        seg_p = __builtin_amdgcn_readfirstlane(q < 0 ? 0 : q);
        bnd = __builtin_amdgcn_readfirstlane(q2 >= 0 ? seg_k[q] : 0x7fffffff);
    }
    for (int b = 0; b < nblocks; ++b) {
        const int buf = b & 1;
        for (int sl = 0; sl < TB && step < T; ++sl, ++step) {
            const int pb = step & 1;
            if constexpr (SEG) {
                // the segments are walked last to first; seg_p's steps end where bnd says (scalar compare, see SegCursor)
                if (step == bnd) {
                    seg_end(row + seg_p);
                    int q = seg_p - 1;
                    while (q >= 0 && seg_k[q] == 0) --q;
                    seg_p = __builtin_amdgcn_readfirstlane(q);
                    int q2 = q - 1;
                    while (q2 >= 0 && seg_k[q2] == 0) --q2;
                    bnd = __builtin_amdgcn_readfirstlane(q2 >= 0 ? bnd + seg_k[q] : 0x7fffffff);
                }
            }
            // dh_prev partials of units (lane, lane + 64) over this wave's gate rows; dgh[j0 + jj] sits in lane jj % 64 of
            // THIS wave (register dv[jj / 64])
            // the step's staged operands of this lane's gate rows, read under the matvec instead of behind the barrier
            const float* ip = &in_s[buf][sl][0];
            float iv[NGR][6];
#pragma unroll
            for (int q = 0; q < NGR; ++q)
#pragma unroll
                for (int e = 0; e < 6; ++e) iv[q][e] = ip[e * GH + uu[q]];
            // (operands fetched four at a time: back-to-back v_readlane into distinct SGPRs, then the four FMAs -- a
            // readlane directly followed by its consumer costs two wait states and serialises on one SGPR)
            f32x2 acc0 = {0.f, 0.f}, acc1 = {0.f, 0.f};
#pragma unroll
            for (int j4 = 0; j4 < JPW; j4 += GRP) {
                float dj[GRP];
#pragma unroll
                for (int e = 0; e < GRP; ++e) dj[e] = (j4 + e < JPW) ? lane_bcast(dv[(j4 + e) / 64], (j4 + e) % 64) : 0.f;
                if (GRP == 4) asm volatile("" : "+s"(dj[0]), "+s"(dj[GRP > 1 ? 1 : 0]), "+s"(dj[GRP > 2 ? 2 : 0]), "+s"(dj[GRP > 3 ? 3 : 0]));
#pragma unroll
                for (int e = 0; e < GRP; ++e) {
                    if (j4 + e >= JPW) continue;
                    const f32x2 dd = {dj[e], dj[e]};
                    if ((j4 + e) & 1) acc1 = __builtin_elementwise_fma(w[j4 + e], dd, acc1);
                    else acc0 = __builtin_elementwise_fma(w[j4 + e], dd, acc0);
                }
            }
            part[pb][wv][lane] = acc0[0] + acc1[0];
            part[pb][wv][lane + 64] = acc0[1] + acc1[1];       // (units >= 100 are padding)
            __syncthreads();
            float* orow = &out_s[sl][0];
#pragma unroll
            for (int q = 0; q < NGR; ++q) {
                const int un = uu[q];
                float pr[NW];
#pragma unroll
                for (int w2 = 0; w2 < NW; ++w2) pr[w2] = part[pb][w2][un];
#pragma unroll
                for (int sp = 1; sp < NW; sp *= 2)                         // fixed tree: bit-reproducible
#pragma unroll
                    for (int w2 = 0; w2 + sp < NW; w2 += 2 * sp) pr[w2] += pr[w2 + sp];
                const float rec = pr[0];
                // the pre-activation gradient of gate row (g, u) from dh of unit u (every lane that shares the unit
                // rebuilds the same dh and carries the same dh z)
                const float dh = iv[q][0] + carry[q] + rec;
                const float rr = iv[q][1], zz = iv[q][2], nn = iv[q][3], ghn = iv[q][4], hprev = iv[q][5];
                const float dn = dh * (1.0f - zz);
                const float dz = dh * (hprev - nn);
                carry[q] = dh * zz;
                const float dnpre = dn * (1.0f - nn * nn);
                const float drpre = dnpre * ghn * rr * (1.0f - rr);
                const float dzpre = dz * zz * (1.0f - zz);
                const float dghn = dnpre * rr;
                const float gi_v = gg[q] == 0 ? drpre : (gg[q] == 1 ? dzpre : dnpre);
                const float gh_v = gg[q] == 0 ? drpre : (gg[q] == 1 ? dzpre : dghn);
                if (has[q]) {                 // (the math runs on every lane: a branch would pull the reads back down)
                    orow[jr[q]] = gi_v;
                    orow[3 * GH + jr[q]] = gh_v;
                }
                dv[q] = gh_v;
            }
        }
        __syncthreads();
        if (b + 1 < nblocks) stash_block(buf ^ 1);
        flush_block(b);
        if (b + 2 < nblocks) load_block(b + 2);
        __syncthreads();
    }
    if constexpr (SEG) {
        if (T > 0) seg_end(row + seg_p);       // the segment the forward pass ran first
    }
}

template <int NW, int GRP, int SEG>
__global__ __launch_bounds__(64 * NW) void gru_seq_bwd_kpart_kernel(BwdGroups G) {
    gru_bwd_kpart_body<NW, GRP, SEG, false>(G, (int)blockIdx.x, (int)blockIdx.y, nullptr);
}

// The recurrence launch WITH RIDERS: workgroups [0, ngru) run the recurrence (bid -> (slot, direction) as the plain launch's
// grid would), workgroups [ngru8, ...) run weight-gradient tiles of the step's queue (gemm_tn_split_body.h) that do not depend on
// this recurrence.  The recurrence needs a CU per sequence for ~T x 0.75 us and leaves the other CUs idle (cfg2: 160 of 256 busy);
// the launch's 108 KB of LDS per workgroup keeps every CU to ONE workgroup, so a rider never shares a CU with a recurrence
// (whose latency chain a co-resident matrix kernel slows down, profiles/r03_wgrad_overlap.md), and the recurrences, having the
// lowest block indices, are placed first.

__global__ __launch_bounds__(512) void gru_seq_bwd_riders_kernel(const BwdGroups G, const TnRiderSegs rq, const int nslots,
                                                                 const int ngru8) {
    extern __shared__ __attribute__((aligned(16))) unsigned char rider_smem[];
    const int bid = (int)blockIdx.x;
    if (bid < 2 * nslots) {
        gru_bwd_kpart_body<8, 4, 0, true>(G, bid % nslots, bid / nslots, rider_smem);
    } else if (bid >= ngru8) {
        tnsb::tns_block<0>(rq, bid - ngru8, rider_smem, nullptr);
    }
}

// one sequence per workgroup, backward pass: the wave-partitioned kernel (8 waves) unless the tuning build asks for the
// lane-pair one (MMDFN_GRU_KPART_BWD=0) for A/B runs.  Measured at cfg2 (profiles/r02_gru_kernels.md): lane-pair 110 us,
// partitioned over 4 waves 126, over 8 waves 84, over 16 waves ~125.  The forward pass keeps the lane-pair kernel (80 us;
// its wave-partitioned forms 102-104 us, a lane-quad form on 8 waves 80 us).
// Gradient that reaches the all-padding sequence (the extra row behind y_tab) from the rows that were truncated against it:
//   dyt[t][dir half] = sum over rows with k_row <= t of dy[t][row]     (their output at t is a copy of y_tab[t])
//                    + sum over rows with k_row == t >= 1 of dhinit[row] (they started from y_tab[t]),   the other half = 0.
// One workgroup per t, fixed summation order (bit-reproducible).
__global__ __launch_bounds__(1024) void gru_tab_reduce_kernel(const float* __restrict__ dy, const int32_t* __restrict__ kout,
                                                              const float* __restrict__ dhinit, float* __restrict__ dyt,
                                                              int rows, int T, int dir) {
    constexpr int C4 = GH / 4;        // 25 float4 columns on 32 lanes, 32 row groups
    constexpr int RG = 32;
    __shared__ float4 red[RG][32];
    const int t = blockIdx.x;
    const int c4 = threadIdx.x & 31, rg = threadIdx.x >> 5;
    const int cc = c4 < C4 ? c4 : C4 - 1;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    // eight rows per trip, every load issued before the first use (the row count per thread is what bounds this kernel)
    for (int r0 = rg; r0 < rows; r0 += 8 * RG) {
        float4 v[8], h[8];
        float m[8], mh[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int r = r0 + e * RG;
            const int rc = r < rows ? r : rows - 1;
            const int k = kout[rc];
            m[e] = (r < rows && k <= t) ? 1.f : 0.f;
            mh[e] = (r < rows && k == t && k >= 1) ? 1.f : 0.f;
            v[e] = *reinterpret_cast<const float4*>(dy + ((int64_t)t * rows + rc) * (2 * GH) + dir * GH + 4 * cc);
            h[e] = *reinterpret_cast<const float4*>(dhinit + (int64_t)rc * GH + 4 * cc);
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            // (select, not multiply: a row the sum excludes may hold anything)
            if (m[e] != 0.f) { acc.x += v[e].x; acc.y += v[e].y; acc.z += v[e].z; acc.w += v[e].w; }
            if (mh[e] != 0.f) { acc.x += h[e].x; acc.y += h[e].y; acc.z += h[e].z; acc.w += h[e].w; }
        }
    }
    red[rg][c4] = acc;
    __syncthreads();
    if (rg == 0 && c4 < C4) {
        float4 a = red[0][c4];
#pragma unroll
        for (int g = 1; g < RG; ++g) {
            const float4 v = red[g][c4];
            a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
        }
        *reinterpret_cast<float4*>(dyt + (int64_t)t * (2 * GH) + dir * GH + 4 * c4) = a;
        *reinterpret_cast<float4*>(dyt + (int64_t)t * (2 * GH) + (1 - dir) * GH + 4 * c4) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
}

// chain enumeration of a segmented launch: full-length (group, direction) slots first, merged truncated slots last (the
// hardware hands out workgroups in index order: longest first)
int seg_slots(SegInfo& S, int ngroups, const int* rows, const int* T, const int32_t* const* rank, const int* P, const int* BP,
              const int* tdir) {
    for (int g = 0; g < ngroups; ++g) {
        S.rank[g] = rank ? rank[g] : nullptr;
        S.P[g] = 1; S.BP[g] = 1; S.tdir[g] = -1;
        if (S.rank[g] != nullptr) {
            if (P[g] <= 0 || P[g] > MAXSEG || BP[g] <= 0 || BP[g] % P[g] || rows[g] % BP[g] || T[g] >= 4096 ||
                (int64_t)P[g] * T[g] > SCHED_MAX || tdir[g] < -1 || tdir[g] > 1) return -1;
            S.P[g] = P[g]; S.BP[g] = BP[g]; S.tdir[g] = tdir[g];
        } else if (T[g] > SCHED_MAX) {
            return -1;
        }
    }
    int ns = 0, start = 0;
    for (int pass = 0; pass < 2; ++pass)
        for (int g = 0; g < ngroups; ++g)
            for (int d = 0; d < 2; ++d) {
                const bool trunc = S.rank[g] != nullptr && S.tdir[g] == d;
                if ((int)trunc != pass) continue;
                S.slot_g[ns] = g; S.slot_d[ns] = d; S.slot_start[ns] = start;
                start += trunc ? rows[g] / S.P[g] : rows[g];
                ++ns;
            }
    S.nslot = ns;
    S.slot_start[ns] = start;
    return start;
}

// Launches with more sequence-directions than this run the MFMA form (gru_mfma.hip: 16 sequences per workgroup, ~1.2 us per
// step whatever the batch): beyond two rounds of the one-sequence-per-workgroup kernels (512 slots each) it is the shorter launch.
int mfma_min_chains() {
#ifdef MMDFN_TUNING
    if (const char* e = getenv("MMDFN_GRU_MFMA_MIN")) return atoi(e);      // tests / A-B: 0 forces the form, a huge value disables it
#endif
    return 1024;
}

bool kpart_any_size() {
#ifdef MMDFN_TUNING
    const char* e = getenv("MMDFN_GRU_KPART_BWD");
    return e != nullptr && e[0] == '2';        // A/B aid: the 8-wave kernel also for batches of several rounds
#else
    return false;
#endif
}

bool use_kpart_bwd() {
#ifdef MMDFN_TUNING
    const char* e = getenv("MMDFN_GRU_KPART_BWD");
    if (e != nullptr) return e[0] != '0';
#endif
    return true;
}

int pick_r(int ngroups, const int* rows) {
#ifdef MMDFN_TUNING
    if (const char* e = getenv("MMDFN_GRU_R")) {        // A/B aid: force the rows per workgroup
        const int v = atoi(e);
        if (v == 1 || v == 2 || v == 4) return v;
    }
#endif
    // One sequence per workgroup wins far beyond one round of workgroups: two lane-pair workgroups share a CU (two
    // independent latency chains per SIMD), which hides more than putting R sequences behind each other in one wave
    // (cfg3, ~900 sequences of <= 33 steps: R = 1 1.45 ms per step, R = 2 1.50, R = 4 1.73).  More rows per workgroup only
    // amortise the 120 KB weight prologue, which matters for very many short sequences.
    for (int R : {1, 2, 4}) {
        int wg = 0;
        for (int g = 0; g < ngroups; ++g) wg += (rows[g] + R - 1) / R;
        if (2 * wg <= 4096) return R;
    }
    return 4;
}

}  // namespace

extern "C" int mmdfn_gru_seq_fwd(int ngroups, const float* const* gi, const float* const* w_hh,
                                 const float* const* b_hh, float* const* y, float* const* gates, const int* rows,
                                 const int* T, int H, void* stream) {
    if (ngroups <= 0 || ngroups > MAXG || H != GH) return -1;
    FwdGroups G;
    G.n = ngroups;
    G.abl = 0;
#ifdef MMDFN_TUNING
    if (const char* e = getenv("MMDFN_GRU_ABL")) G.abl = atoi(e);
#endif
    const int R = pick_r(ngroups, rows);
    int sl = 0;
    for (int g = 0; g < ngroups; ++g) {
        if (rows[g] <= 0 || T[g] <= 0) return -1;
        G.gi[g] = gi[g]; G.y[g] = y[g]; G.gates[g] = gates[g];
        G.w_hh[2 * g] = w_hh[2 * g]; G.w_hh[2 * g + 1] = w_hh[2 * g + 1];
        G.b_hh[2 * g] = b_hh[2 * g]; G.b_hh[2 * g + 1] = b_hh[2 * g + 1];
        G.rows[g] = rows[g]; G.T[g] = T[g]; G.slice0[g] = sl;
        sl += (rows[g] + R - 1) / R;
    }
    G.slice0[ngroups] = sl;
    dim3 grid(sl, 2), block(NT);
    hipStream_t s = (hipStream_t)stream;
    {
        int chains = 0;
        for (int g = 0; g < ngroups; ++g) chains += 2 * rows[g];
        if (chains > mfma_min_chains() && G.abl == 0) return mmdfn_launch_gru_fwd_mfma(ngroups, gi, w_hh, b_hh, y, gates, rows, T, s);
    }
    // the 5-wave kernel runs one workgroup per CU (its fifth wave shares a SIMD with a recurrence wave at ~210 VGPRs each):
    // it wins while every sequence gets a CU of its own in one round (cfg2: 160 workgroups); beyond that the 4-wave kernel,
    // two workgroups per CU, needs fewer rounds (cfg3 / cfg4: +4-5 % step time with the 5-wave kernel, measured)
    bool io_wave = (R == 1) && (2 * sl <= 256);
#ifdef MMDFN_TUNING
    if (const char* e = getenv("MMDFN_GRU_IO")) io_wave = io_wave && e[0] != '0';      // A/B aid
#endif
    bool scalar_fma = false;
#ifdef MMDFN_TUNING
    if (const char* e = getenv("MMDFN_GRU_SCALAR_FMA")) scalar_fma = e[0] == '1';
#endif
#ifdef MMDFN_TUNING
#define GRU_IO_ABL(A) if (io_wave && G.abl == A) { hipLaunchKernelGGL((gru_seq_fwd_io_kernel<0, A, 0>), grid, dim3(320), 0, s, G); MMDFN_CHECK_LAUNCH(); return 0; }
    GRU_IO_ABL(1) GRU_IO_ABL(2) GRU_IO_ABL(4) GRU_IO_ABL(8) GRU_IO_ABL(16) GRU_IO_ABL(3) GRU_IO_ABL(11) GRU_IO_ABL(31) GRU_IO_ABL(23) GRU_IO_ABL(64)
#undef GRU_IO_ABL
#endif
    if (io_wave && !scalar_fma && 2 * sl < 256) {
        if (const kfb::FlagJob* fj = mmdfn_flag_job_pending()) {
            // a staged dropout-flag draw rides on the CUs this launch leaves idle (gru_seq_fwd_io_flags_kernel)
            int64_t nr = (fj->n8 + 319) / 320;
            if (nr > 256 - 2 * sl) nr = 256 - 2 * sl;
            const kfb::FlagJob J = *fj;
            mmdfn_flag_job_taken();
            hipLaunchKernelGGL(gru_seq_fwd_io_flags_kernel, dim3(2 * sl + (int)nr), dim3(320), 0, s, G, J, sl);
            MMDFN_CHECK_LAUNCH();
            return 0;
        }
    }
    if (io_wave && scalar_fma) hipLaunchKernelGGL((gru_seq_fwd_io_kernel<1, 0, 0>), grid, dim3(320), 0, s, G);
    else if (io_wave) hipLaunchKernelGGL((gru_seq_fwd_io_kernel<0, 0, 0>), grid, dim3(320), 0, s, G);
    else if (R == 1) hipLaunchKernelGGL((gru_seq_fwd_kernel<1, 0>), grid, block, 0, s, G);
    else if (R == 2) hipLaunchKernelGGL((gru_seq_fwd_kernel<2, 0>), grid, block, 0, s, G);
    else hipLaunchKernelGGL((gru_seq_fwd_kernel<4, 0>), grid, block, 0, s, G);
    MMDFN_CHECK_LAUNCH();
    return 0;
}

extern "C" int mmdfn_gru_seq_bwd(int ngroups, const float* const* dy, const float* const* y,
                                 const float* const* gates, const float* const* w_hh, float* const* dgi,
                                 float* const* dgh, const int* rows, const int* T, int H, void* stream) {
    if (ngroups <= 0 || ngroups > MAXG || H != GH) return -1;
    BwdGroups G;
    G.n = ngroups;
    const int R = pick_r(ngroups, rows);
    int sl = 0;
    for (int g = 0; g < ngroups; ++g) {
        if (rows[g] <= 0 || T[g] <= 0) return -1;
        G.dy[g] = dy[g]; G.y[g] = y[g]; G.gates[g] = gates[g]; G.dgi[g] = dgi[g];
        G.w_hh[2 * g] = w_hh[2 * g]; G.w_hh[2 * g + 1] = w_hh[2 * g + 1];
        G.dgh[g] = dgh[g]; G.rows[g] = rows[g]; G.T[g] = T[g]; G.slice0[g] = sl;
        sl += (rows[g] + R - 1) / R;
    }
    G.slice0[ngroups] = sl;
    dim3 grid(sl, 2), block(NT);
    hipStream_t s = (hipStream_t)stream;
    {
        int chains = 0;
        for (int g = 0; g < ngroups; ++g) chains += 2 * rows[g];
        if (chains > mfma_min_chains()) return mmdfn_launch_gru_bwd_mfma(ngroups, dy, y, gates, w_hh, dgi, dgh, rows, T, s);
    }
    // (the 8-wave kernel runs one workgroup per CU: it wins while all sequences fit in one round; beyond that the lane-pair
    // kernel, two workgroups per CU, keeps the batch in one round -- cfg4: 320 workgroups, 1.75 vs 1.69 ms per step)
    if (R == 1 && (2 * sl <= 256 || kpart_any_size()) && use_kpart_bwd()) {
        if (const TnSplitSegs* rp = mmdfn_riders_pending()) {
            // a staged weight-gradient batch rides on the CUs this launch leaves idle (gru_seq_bwd_riders_kernel)
            if (rp->n <= MMDFN_RIDER_MAXSEG && 2 * sl < 256) {
                const TnRiderSegs rq = mmdfn_rider_table(*rp);
                const int ngru8 = (2 * sl + 7) & ~7;
                if (int e = mmdfn_allow_big_lds(gru_seq_bwd_riders_kernel)) return e;
                hipLaunchKernelGGL(gru_seq_bwd_riders_kernel, dim3(ngru8 + rp->wg_prefix[rp->n]), dim3(512), tnsb::LDS_B, s, G, rq, sl,
                                   ngru8);
                MMDFN_CHECK_LAUNCH();
                return mmdfn_riders_launched(s);
            }
        }
        hipLaunchKernelGGL((gru_seq_bwd_kpart_kernel<8, 4, 0>), grid, dim3(512), 0, s, G);
    }
    else if (R == 1) hipLaunchKernelGGL((gru_seq_bwd_kernel<1, 0>), grid, block, 0, s, G);
    else if (R == 2) hipLaunchKernelGGL((gru_seq_bwd_kernel<2, 0>), grid, block, 0, s, G);
    else hipLaunchKernelGGL((gru_seq_bwd_kernel<4, 0>), grid, block, 0, s, G);
    MMDFN_CHECK_LAUNCH();
    return 0;
}

// CUs the plain backward launch of these groups would leave idle IF it is of the kind that takes weight-gradient riders (one
// sequence per workgroup on the wave-partitioned kernel, fewer workgroups than CUs); 0 otherwise.  The host stages a rider
// batch (mmdfn_wgrad_riders_stage) only in front of such a launch, and sizes it by this number.
extern "C" int mmdfn_gru_seq_bwd_idle_cus(int ngroups, const int* rows) {
    if (ngroups <= 0 || ngroups > MAXG) return 0;
    const int R = pick_r(ngroups, rows);
    int sl = 0, chains = 0;
    for (int g = 0; g < ngroups; ++g) {
        if (rows[g] <= 0) return 0;
        sl += (rows[g] + R - 1) / R;
        chains += 2 * rows[g];
    }
    if (chains > mfma_min_chains()) {           // the MFMA form: 16 sequences per workgroup (gru_mfma.hip)
        int slm = 0;
        for (int g = 0; g < ngroups; ++g) slm += (rows[g] + 15) / 16;
        return 2 * slm < 256 ? 256 - 2 * slm : 0;
    }
    if (R == 1 && 2 * sl < 256 && use_kpart_bwd()) return 256 - 2 * sl;
    return 0;
}

// 1 if the plain FORWARD launch of these groups is of the kind that carries a staged dropout-flag draw (mmdfn_keep_flags_stage)
extern "C" int mmdfn_gru_seq_fwd_takes_flags(int ngroups, const int* rows) {
    if (ngroups <= 0 || ngroups > MAXG) return 0;
    const int R = pick_r(ngroups, rows);
    int sl = 0, chains = 0;
    for (int g = 0; g < ngroups; ++g) {
        if (rows[g] <= 0) return 0;
        sl += (rows[g] + R - 1) / R;
        chains += 2 * rows[g];
    }
    if (chains > mfma_min_chains()) {           // the MFMA form: 16 sequences per workgroup (gru_mfma.hip)
        int slm = 0;
        for (int g = 0; g < ngroups; ++g) slm += (rows[g] + 15) / 16;
        return 2 * slm < 256 ? 1 : 0;
    }
    return (R == 1 && 2 * sl < 256) ? 1 : 0;
}

// nanoseconds per recurrence step of that launch (what the rider batch's size is priced with)
extern "C" int mmdfn_gru_seq_bwd_step_ns(int ngroups, const int* rows) {
    int chains = 0;
    for (int g = 0; g < ngroups && g < MAXG; ++g) chains += 2 * rows[g];
    return chains > mfma_min_chains() ? 2300 : 750;
}

extern "C" int mmdfn_gru_seq_fwd_seg(int ngroups, const float* const* gi, const float* const* w_hh,
                                     const float* const* b_hh, float* const* y, float* const* gates, const int* rows,
                                     const int* T, int H, const int32_t* const* rank, const int* P, const int* BP,
                                     const int* tdir, const float* const* ytab, void* stream) {
    if (ngroups <= 0 || ngroups > MAXG || H != GH) return -1;
    FwdGroups G;
    G.n = ngroups;
    G.abl = 0;
    for (int g = 0; g < ngroups; ++g) {
        if (rows[g] <= 0 || T[g] <= 0) return -1;
        G.gi[g] = gi[g]; G.y[g] = y[g]; G.gates[g] = gates[g];
        G.w_hh[2 * g] = w_hh[2 * g]; G.w_hh[2 * g + 1] = w_hh[2 * g + 1];
        G.b_hh[2 * g] = b_hh[2 * g]; G.b_hh[2 * g + 1] = b_hh[2 * g + 1];
        G.rows[g] = rows[g]; G.T[g] = T[g]; G.slice0[g] = 0;
        G.ytab[g] = ytab ? ytab[g] : nullptr;
        if (G.ytab[g] != nullptr && (rank == nullptr || rank[g] == nullptr || tdir[g] != 1)) return -1;   // a start table serves the reverse direction
    }
    const int nchains = seg_slots(G.seg, ngroups, rows, T, rank, P, BP, tdir);
    if (nchains <= 0) return -1;
    // one chain per CU: the 5-wave kernel (one workgroup per CU); more chains: the 4-wave kernel, two workgroups per CU
    bool io_wave = nchains <= 256;
#ifdef MMDFN_TUNING
    if (const char* e = getenv("MMDFN_GRU_IO")) io_wave = e[0] != '0';      // A/B aid
#endif
#ifdef MMDFN_TUNING
    if (const char* e = getenv("MMDFN_GRU_ABL")) {         // timing ablations of the segmented bookkeeping (wrong results)
        const int a = atoi(e);
#define GRU_SEG_ABL(A) if (io_wave && a == A) { hipLaunchKernelGGL((gru_seq_fwd_io_kernel<0, A, 1>), dim3(nchains), dim3(320), 0, (hipStream_t)stream, G); MMDFN_CHECK_LAUNCH(); return 0; }
        GRU_SEG_ABL(32)
#undef GRU_SEG_ABL
    }
#endif
    if (io_wave) hipLaunchKernelGGL((gru_seq_fwd_io_kernel<0, 0, 1>), dim3(nchains), dim3(320), 0, (hipStream_t)stream, G);
    else hipLaunchKernelGGL((gru_seq_fwd_kernel<1, 1>), dim3(nchains), dim3(NT), 0, (hipStream_t)stream, G);
    MMDFN_CHECK_LAUNCH();
    return 0;
}

extern "C" int mmdfn_gru_seq_bwd_seg(int ngroups, const float* const* dy, const float* const* y,
                                     const float* const* gates, const float* const* w_hh, float* const* dgi,
                                     float* const* dgh, const int* rows, const int* T, int H,
                                     const int32_t* const* rank, const int* P, const int* BP, const int* tdir,
                                     float* const* dhinit, int32_t* const* kout, void* stream) {
    if (ngroups <= 0 || ngroups > MAXG || H != GH) return -1;
    BwdGroups G;
    G.n = ngroups;
    for (int g = 0; g < ngroups; ++g) {
        if (rows[g] <= 0 || T[g] <= 0) return -1;
        G.dy[g] = dy[g]; G.y[g] = y[g]; G.gates[g] = gates[g]; G.dgi[g] = dgi[g];
        G.w_hh[2 * g] = w_hh[2 * g]; G.w_hh[2 * g + 1] = w_hh[2 * g + 1];
        G.dgh[g] = dgh[g]; G.rows[g] = rows[g]; G.T[g] = T[g]; G.slice0[g] = 0;
        G.dhinit[g] = dhinit ? dhinit[g] : nullptr;
        G.kout[g] = kout ? kout[g] : nullptr;
        if ((G.dhinit[g] != nullptr) != (G.kout[g] != nullptr)) return -1;
        if (G.dhinit[g] != nullptr && (rank == nullptr || rank[g] == nullptr || tdir[g] != 1)) return -1;
    }
    const int nchains = seg_slots(G.seg, ngroups, rows, T, rank, P, BP, tdir);
    if (nchains <= 0) return -1;
    bool kpart = nchains <= 256;
#ifdef MMDFN_TUNING
    if (const char* e = getenv("MMDFN_GRU_KPART_BWD")) kpart = e[0] != '0';      // A/B aid
#endif
    if (kpart) hipLaunchKernelGGL((gru_seq_bwd_kpart_kernel<8, 4, 1>), dim3(nchains), dim3(512), 0, (hipStream_t)stream, G);
    else hipLaunchKernelGGL((gru_seq_bwd_kernel<1, 1>), dim3(nchains), dim3(NT), 0, (hipStream_t)stream, G);
    MMDFN_CHECK_LAUNCH();
    return 0;
}

extern "C" int mmdfn_gru_tab_reduce(const float* dy, const int32_t* kout, const float* dhinit, float* dyt, int rows, int T,
                                    int H, int dir, void* stream) {
    if (H != GH || rows <= 0 || T <= 0 || dir < 0 || dir > 1) return -1;
    hipLaunchKernelGGL(gru_tab_reduce_kernel, dim3(T), dim3(1024), 0, (hipStream_t)stream, dy, kout, dhinit, dyt, rows, T, dir);
    MMDFN_CHECK_LAUNCH();
    return 0;
}
