// K3 / K4: speaker-party gather / scatter and pad-strip of the encoders.
//
// Reference (model.py:1070-1090, 1101-1121, 1134-1154, 553-565): per modality, dialogue b and speaker p a
// Python loop compacts speaker p's utterances to the front of a zero (L, H) buffer
// (`U_parties_[p][b][:k] = U_[b][index_i]`), runs the party GRU, scatters the first k outputs back
// (`U_p_[b][index_i] = E_parties_[p][b][:k]`), adds `w_m * U_p` to the base encoding and finally strips the
// padding dialogue by dialogue.  B*P*2*3 slice-assigns (each with a CopySlices backward) become:
//   party_gather       : one workgroup per (dialogue, speaker): ballot/popcount prefix scan over qmask gives
//                        the compaction order, then a coalesced 16-byte copy of all modalities' rows;
//   party_combine      : out[m][n] = base_m[t,b] + w_m * E[rank[t,b,p*], (m,b,p*)]  written directly in the
//                        dialogue-major (M, N, H) order the graph kernels consume (p* = last flagged speaker,
//                        the reference scatters speaker by speaker so the last one wins);
//   and their backward counterparts (single writer per output element, no atomics).
#include "mmdfn_internal.h"
#include "keep_flags_body.h"
#include "../../include/mmdfn_hip.h"

namespace {

constexpr int MAXMOD = 4;
constexpr int MAXL = 2048;
constexpr int COLSUM_SLABS = 48;   // row slabs of the column-sum kernel (x 64-column blocks: ~480 workgroups at H = 600)

struct ModPtrs {
    const float* p[MAXMOD];
};
struct ModPtrsW {
    float* p[MAXMOD];
};

// S: (L, Mn*B*P, H) with column ((m*B + b)*P + p);  rank: (L, B, P) int32, -1 = not this speaker
__global__ __launch_bounds__(256) void party_gather_kernel(ModPtrs X, const float* __restrict__ qmask,
                                                           const float* __restrict__ bias, float* __restrict__ S,
                                                           int32_t* __restrict__ rank, int L, int B, int P, int Mn,
                                                           int H) {
    __shared__ int sel[MAXL];
    __shared__ int cnt_s;
    const int b = blockIdx.x / P;
    const int p = blockIdx.x - b * P;
    const int tid = threadIdx.x;
    if (tid < 64) {
        int base = 0;
        for (int t0 = 0; t0 < L; t0 += 64) {
            const int t = t0 + tid;
            const bool f = (t < L) && (qmask[((int64_t)t * B + b) * P + p] != 0.f);
            const unsigned long long bal = __ballot(f);
            const int pre = __popcll(bal & ((1ull << tid) - 1ull));
            if (t < L && blockIdx.y == 0) rank[((int64_t)t * B + b) * P + p] = f ? base + pre : -1;
            if (f) sel[base + pre] = t;
            base += __popcll(bal);
        }
        if (tid == 0) cnt_s = base;
    }
    __syncthreads();
    const int cnt = cnt_s;
    const int H4 = H / 4;
    const int per_k = Mn * H4;
    const int64_t cols = (int64_t)Mn * B * P;
    // the copy is cut into gridDim.y row ranges (B*P workgroups alone leave most of the chip idle; every slice
    // redoes the cheap scan above)
    const int kper = (L + gridDim.y - 1) / gridDim.y;
    const int k_lo = blockIdx.y * kper;
    const int k_hi = (k_lo + kper < L) ? k_lo + kper : L;
    // thread -> fixed (modality, 16-byte column) slots, rows k in the inner loop: no integer division per element, the bias
    // slice is loaded once per slot
    for (int j = tid; j < per_k; j += 256) {
        const int m = j / H4;
        const int c4 = j - m * H4;
        float4 bb = make_float4(0.f, 0.f, 0.f, 0.f);
        if (bias != nullptr) bb = *reinterpret_cast<const float4*>(bias + 4 * c4);   // rows gathered AFTER a bias-free projection: every row, padding included, gets the bias
        const float* xm = X.p[m] + (int64_t)b * H + 4 * c4;
        float* sm = S + (((int64_t)m * B + b) * P + p) * H + 4 * c4;
#pragma unroll 4
        for (int k = k_lo; k < k_hi; ++k) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (k < cnt) v = *reinterpret_cast<const float4*>(xm + (int64_t)sel[k] * B * H);
            v.x += bb.x; v.y += bb.y; v.z += bb.z; v.w += bb.w;
            *reinterpret_cast<float4*>(sm + (int64_t)k * cols * H) = v;
        }
    }
}

// dX_m[t,b,:] = addend_m[t,b,:] + sum_p [rank[t,b,p] >= 0] dS[rank, (m,b,p), :]
__device__ __forceinline__ void party_gather_bwd_body(const float* __restrict__ dS, const int32_t* __restrict__ rank,
                                                      const ModPtrsW& dX, const ModPtrs& addend, int L, int B, int P, int Mn, int H,
                                                      int bid, int nblocks) {
    const int H4 = H / 4;
    const int64_t total = (int64_t)Mn * L * B * H4;
    const int64_t cols = (int64_t)Mn * B * P;
    for (int64_t idx = bid * (int64_t)blockDim.x + threadIdx.x; idx < total; idx += (int64_t)nblocks * blockDim.x) {
        const int c4 = (int)(idx % H4);
        int64_t r = idx / H4;
        const int b = (int)(r % B);
        r /= B;
        const int t = (int)(r % L);
        const int m = (int)(r / L);
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        // the gradient that reaches X_m on its other path (X_m is also the base of the combine stage): added here
        // instead of by an autograd accumulation launch
        if (addend.p[m]) acc = *reinterpret_cast<const float4*>(addend.p[m] + ((int64_t)t * B + b) * H + 4 * c4);
        for (int p = 0; p < P; ++p) {
            const int k = rank[((int64_t)t * B + b) * P + p];
            if (k >= 0) {
                const float4 v =
                    *reinterpret_cast<const float4*>(dS + ((int64_t)k * cols + ((int64_t)m * B + b) * P + p) * H + 4 * c4);
                acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
            }
        }
        *reinterpret_cast<float4*>(dX.p[m] + ((int64_t)t * B + b) * H + 4 * c4) = acc;
    }
}

__global__ void party_gather_bwd_kernel(const float* __restrict__ dS, const int32_t* __restrict__ rank, ModPtrsW dX,
                                        ModPtrs addend, int L, int B, int P, int Mn, int H) {
    party_gather_bwd_body(dS, rank, dX, addend, L, B, P, Mn, H, (int)blockIdx.x, (int)gridDim.x);
}

// out[m][n][:] = base_m[t,b,:] + w_m * E[rank[t,b,p*], (m,b,p*), :],  flat_idx[n] = t*B + b
__global__ void party_combine_kernel(ModPtrs base, const float* __restrict__ E, const int32_t* __restrict__ rank,
                                     const int64_t* __restrict__ flat_idx, float* __restrict__ out, float w0,
                                     float w1, float w2, float w3, int L, int B, int P, int Mn, int N, int H) {
    const int H4 = H / 4;
    const int64_t total = (int64_t)Mn * N * H4;
    const float wv[MAXMOD] = {w0, w1, w2, w3};
    // E holds one column block per modality whose weight is non-zero (a zero weight means the reference computes the
    // party encoding and multiplies it by 0, model.py:1121: that block is simply not produced here)
    int slot[MAXMOD], nact = 0;
#pragma unroll
    for (int q = 0; q < MAXMOD; ++q) { slot[q] = nact; nact += (q < Mn && wv[q] != 0.f) ? 1 : 0; }
    const int64_t cols = (int64_t)nact * B * P;
    for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int c4 = (int)(idx % H4);
        int64_t r = idx / H4;
        const int n = (int)(r % N);
        const int m = (int)(r / N);
        const int64_t tb = flat_idx[n];
        const int b = (int)(tb % B);
        float4 v = *reinterpret_cast<const float4*>(base.p[m] + tb * H + 4 * c4);
        int ps = -1, ks = -1;
        for (int p = 0; p < P; ++p) {
            const int k = rank[tb * P + p];
            if (k >= 0) { ps = p; ks = k; }
        }
        if (ps >= 0 && E != nullptr && wv[m] != 0.f) {
            const float4 e =
                *reinterpret_cast<const float4*>(E + ((int64_t)ks * cols + ((int64_t)slot[m] * B + b) * P + ps) * H + 4 * c4);
            const float w = wv[m];
            v.x = fmaf(w, e.x, v.x); v.y = fmaf(w, e.y, v.y); v.z = fmaf(w, e.z, v.z); v.w = fmaf(w, e.w, v.w);
        }
        *reinterpret_cast<float4*>(out + ((int64_t)m * N + n) * H + 4 * c4) = v;
    }
}

// dbase_m[t,b,:] = dout[m][n];  dE[rank, (m,b,p*), :] = w_m dout[m][n]   (dbase / dE are pre-zeroed by the caller)
__global__ void party_combine_bwd_kernel(const float* __restrict__ dout, const int32_t* __restrict__ rank,
                                         const int64_t* __restrict__ flat_idx, ModPtrsW dbase,
                                         float* __restrict__ dE, float w0, float w1, float w2, float w3, int L,
                                         int B, int P, int Mn, int N, int H) {
    const int H4 = H / 4;
    const int64_t total = (int64_t)Mn * N * H4;
    const float wv[MAXMOD] = {w0, w1, w2, w3};
    int slot[MAXMOD], nact = 0;
#pragma unroll
    for (int q = 0; q < MAXMOD; ++q) { slot[q] = nact; nact += (q < Mn && wv[q] != 0.f) ? 1 : 0; }
    const int64_t cols = (int64_t)nact * B * P;
    for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int c4 = (int)(idx % H4);
        int64_t r = idx / H4;
        const int n = (int)(r % N);
        const int m = (int)(r / N);
        const int64_t tb = flat_idx[n];
        const int b = (int)(tb % B);
        const float4 g = *reinterpret_cast<const float4*>(dout + ((int64_t)m * N + n) * H + 4 * c4);
        *reinterpret_cast<float4*>(dbase.p[m] + tb * H + 4 * c4) = g;
        int ps = -1, ks = -1;
        for (int p = 0; p < P; ++p) {
            const int k = rank[tb * P + p];
            if (k >= 0) { ps = p; ks = k; }
        }
        if (ps >= 0 && dE != nullptr && wv[m] != 0.f) {
            const float w = wv[m];
            *reinterpret_cast<float4*>(dE + ((int64_t)ks * cols + ((int64_t)slot[m] * B + b) * P + ps) * H + 4 * c4) =
                make_float4(w * g.x, w * g.y, w * g.z, w * g.w);
        }
    }
}

// The same gradients written DESTINATION by destination, so that nothing has to be zeroed first (the caller's fill was a launch
// of its own): workgroup (b, p, row slice) rebuilds the party's time list from `rank`, writes dE[k, (m, b, p), :] = w_m dout[m][n]
// for k below the party's count and zeros above it; the p = 0 workgroups also write dbase_m[t, b, :] = dout[m][n] or zero (padding).
// inv[t * B + b] = n, the row of (t, b) in the stripped order, or -1.
__global__ __launch_bounds__(256) void party_combine_bwd_dst_kernel(const float* __restrict__ dout, const int32_t* __restrict__ rank,
                                                                    const int64_t* __restrict__ inv, ModPtrsW dbase,
                                                                    float* __restrict__ dE, float w0, float w1, float w2, float w3,
                                                                    int L, int B, int P, int Mn, int N, int H) {
    __shared__ int sel[MAXL];
    __shared__ int cnt_s;
    const int b = blockIdx.x / P;
    const int p = blockIdx.x - b * P;
    const int tid = threadIdx.x;
    for (int t = tid; t < L; t += 256) {
        const int k = rank[((int64_t)t * B + b) * P + p];
        if (k >= 0) sel[k] = t;
    }
    if (tid < 64) {
        int c = 0;
        for (int t0 = 0; t0 < L; t0 += 64) {
            const int t = t0 + tid;
            c += __popcll(__ballot(t < L && rank[((int64_t)t * B + b) * P + p] >= 0));
        }
        if (tid == 0) cnt_s = c;
    }
    __syncthreads();
    const int cnt = cnt_s;
    const float wv[MAXMOD] = {w0, w1, w2, w3};
    int slot[MAXMOD], nact = 0;
#pragma unroll
    for (int q = 0; q < MAXMOD; ++q) { slot[q] = nact; nact += (q < Mn && wv[q] != 0.f) ? 1 : 0; }
    const int64_t cols = (int64_t)nact * B * P;
    const int H4 = H / 4;
    const int kper = (L + gridDim.y - 1) / gridDim.y;
    const int k_lo = blockIdx.y * kper;
    const int k_hi = (k_lo + kper < L) ? k_lo + kper : L;
    const int nk = k_hi - k_lo;
    // source rows of the slice, looked up once (the element loop below then has no dependent load): sel is reused for them
    __shared__ int src_e[MAXL / 8 + 8], src_b[MAXL / 8 + 8];
    int my_e = -1, my_b = -1;
    if (tid < nk) {
        const int k = k_lo + tid;
        if (k < cnt) my_e = (int)inv[(int64_t)sel[k] * B + b];
        if (p == 0) my_b = (int)inv[(int64_t)k * B + b];
    }
    __syncthreads();
    if (tid < nk) { src_e[tid] = my_e; src_b[tid] = my_b; }
    __syncthreads();
    // elements (modality, row of the slice, 16-byte column) spread over the threads: independent loads
    const int total = Mn * nk * H4;
    for (int e = tid; e < total; e += 256) {
        const int c4 = e % H4;
        const int r = e / H4;
        const int kk = r % nk;
        const int m = r / nk;
        const float w = wv[m];
        const float* dm = dout + (int64_t)m * N * H + 4 * c4;
        if (dE != nullptr && w != 0.f) {
            const int n = src_e[kk];
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (n >= 0) {
                const float4 g = *reinterpret_cast<const float4*>(dm + (int64_t)n * H);
                v = make_float4(w * g.x, w * g.y, w * g.z, w * g.w);
            }
            *reinterpret_cast<float4*>(dE + ((int64_t)(k_lo + kk) * cols + ((int64_t)slot[m] * B + b) * P + p) * H + 4 * c4) = v;
        }
        if (p == 0) {
            const int n = src_b[kk];
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (n >= 0) v = *reinterpret_cast<const float4*>(dm + (int64_t)n * H);
            *reinterpret_cast<float4*>(dbase.p[m] + ((int64_t)(k_lo + kk) * B + b) * H + 4 * c4) = v;
        }
    }
}

inline int grid_for(int64_t total) {
    int64_t b = (total + 255) / 256;
    if (b > 4096) b = 4096;
    if (b < 1) b = 1;
    return (int)b;
}

struct MaskScaleGroups {
    const float* x[MAXMOD];
    const float* mask[MAXMOD];
    float* out[MAXMOD];
    int64_t start4[MAXMOD + 1];      // prefix sums of n[g] / 4
};

// out = x * mask * scale over the concatenation of up to MAXMOD tensors (16-byte chunks, grid-stride)
__global__ void mask_scale_kernel(MaskScaleGroups G, int ngroups, float scale) {
    const int64_t total = G.start4[ngroups];
    for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        // (selects, not indexed loads: a per-thread index into the kernel-argument arrays would go through scratch)
        const int g = (idx >= G.start4[1]) + (idx >= G.start4[2]) + (idx >= G.start4[3]);
        const int64_t o = idx - (g == 0 ? G.start4[0] : g == 1 ? G.start4[1] : g == 2 ? G.start4[2] : G.start4[3]);
        const float* xp = g == 0 ? G.x[0] : g == 1 ? G.x[1] : g == 2 ? G.x[2] : G.x[3];
        const float* mp = g == 0 ? G.mask[0] : g == 1 ? G.mask[1] : g == 2 ? G.mask[2] : G.mask[3];
        float* op = g == 0 ? G.out[0] : g == 1 ? G.out[1] : g == 2 ? G.out[2] : G.out[3];
        const float4 v = reinterpret_cast<const float4*>(xp)[o];
        const float4 m = reinterpret_cast<const float4*>(mp)[o];
        reinterpret_cast<float4*>(op)[o] = make_float4(v.x * m.x * scale, v.y * m.y * scale, v.z * m.z * scale, v.w * m.w * scale);
    }
}

// out[c] = sum_r A[r][c] (the bias gradient of gate pre-activations that were gathered AFTER a bias-free projection: every
// gathered row, padding included, received the bias, so its gradient is the column sum of ALL rows of dS -- not of the
// scattered dG the weight gradient contracts).  Grid: 64-column blocks x row slabs; a slab's partial sums go to `ws`
// [slab][H] and a second small launch adds them in slab order (bit-reproducible).  (A single launch whose last workgroup
// finishes the sum needs a device-scope release fence per workgroup; on this chip that writes back the L2's dirty lines --
// the 17 MB of dS the GRU backward has just produced -- and took 43 us against 8 us for the two launches.)
__device__ __forceinline__ void colsum_body(float4 (*part)[16], const float* __restrict__ A, int64_t R, int H, int lda,
                                            float* __restrict__ ws, int cb, int sl, int nsl) {
    const int c4 = threadIdx.x & 15, rg = threadIdx.x >> 4;           // 16 lanes x 16 bytes = the 64 columns; 16 row groups
    const int col = 64 * cb + 4 * c4;
    const int64_t rper = (R + nsl - 1) / nsl;
    const int64_t r0 = sl * rper;
    const int64_t r1 = (r0 + rper < R) ? r0 + rper : R;
    float4 acc[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) acc[u] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (col < H) {
        const float* base = A + col;
        for (int64_t r = r0 + rg; r < r1; r += 64) {                  // four independent 16-byte loads in flight per thread
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int64_t rr = r + 16 * u;
                if (rr < r1) {
                    const float4 v = *reinterpret_cast<const float4*>(base + rr * lda);
                    acc[u].x += v.x; acc[u].y += v.y; acc[u].z += v.z; acc[u].w += v.w;
                }
            }
        }
    }
    float4 t = make_float4((acc[0].x + acc[1].x) + (acc[2].x + acc[3].x), (acc[0].y + acc[1].y) + (acc[2].y + acc[3].y),
                           (acc[0].z + acc[1].z) + (acc[2].z + acc[3].z), (acc[0].w + acc[1].w) + (acc[2].w + acc[3].w));
    part[rg][c4] = t;
    __syncthreads();
    if (rg == 0 && col < H) {
        float4 s4 = part[0][c4];
#pragma unroll
        for (int q = 1; q < 16; ++q) { const float4 v = part[q][c4]; s4.x += v.x; s4.y += v.y; s4.z += v.z; s4.w += v.w; }
        *reinterpret_cast<float4*>(ws + (int64_t)sl * H + col) = s4;
    }
}

__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ A, int64_t R, int H, int lda,
                                                     float* __restrict__ ws) {
    __shared__ float4 part[16][16];
    colsum_body(part, A, R, H, lda, ws, (int)blockIdx.x, (int)blockIdx.y, (int)gridDim.y);
}

// The scatter of the party gradient and the column-sum slabs of the SAME dS (the party GRU's bias gradient) in one launch:
// blocks [0, ngather) scatter, the ncb x nsl blocks behind them sum (both only read dS; nothing one half writes is read by
// the other)
__global__ __launch_bounds__(256) void party_gather_bwd_colsum_kernel(const float* __restrict__ dS, const int32_t* __restrict__ rank,
                                                                      ModPtrsW dX, ModPtrs addend, int L, int B, int P, int Mn,
                                                                      int H, int ngather, int64_t R, int ncb, int nsl,
                                                                      float* __restrict__ ws) {
    __shared__ float4 part[16][16];
    const int bid = blockIdx.x;
    if (bid < ngather) {
        party_gather_bwd_body(dS, rank, dX, addend, L, B, P, Mn, H, bid, ngather);
        return;
    }
    const int j = bid - ngather;
    colsum_body(part, dS, R, H, H, ws, j % ncb, j / ncb, nsl);
}

// out[c] = sum over slabs (in slab order) of ws[slab][c]
__global__ __launch_bounds__(64) void colsum_final_kernel(const float* __restrict__ ws, int nsl, int H, float* __restrict__ out) {
    const int c = blockIdx.x * 64 + threadIdx.x;
    if (c >= H) return;
    float v[COLSUM_SLABS];                       // all slabs requested before the first add (one round trip, not nsl)
#pragma unroll
    for (int s2 = 0; s2 < COLSUM_SLABS; ++s2) v[s2] = (s2 < nsl) ? ws[(int64_t)s2 * H + c] : 0.f;
    float t = 0.f;
#pragma unroll
    for (int s2 = 0; s2 < COLSUM_SLABS; ++s2) t += v[s2];
    out[c] = t;
}

// ---------------------------------------------------------------------------------------------------------------------------
// Dropout keep flags: out[i] = 1.0f with probability keep, else 0.0f, from Philox4x32-10 (the counter-based generator torch's
// dropout uses) keyed by state[0] (seed) at counter state[1] + i / 8.  The generator state lives in DEVICE memory and the launch
// advances it itself (the last workgroup to finish -- a relaxed atomic counter, no fence: every workgroup has read the offset
// before it increments the counter -- adds the number of counters consumed), so a captured graph draws fresh flags at every
// replay without the per-replay seed / offset fill launches of the framework generator, and one 6 M-flag draw costs 5 us instead
// of 11 us for `bernoulli_` + 9 us for those fills.
// ---------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void keep_flags_kernel(const kfb::FlagJob J) {
    kfb::keep_flags_block<1024>(J, (int)blockIdx.x, (int)gridDim.x);
}

}  // namespace

extern "C" int mmdfn_party_gather(int Mn, const float* const* X, const float* qmask, const float* bias, float* S,
                                  int32_t* rank, int L, int B, int P, int H, void* stream) {
    if (Mn <= 0 || Mn > MAXMOD || L <= 0 || L > MAXL || B <= 0 || P <= 0 || H <= 0 || (H & 3)) return -1;
    ModPtrs x;
    for (int m = 0; m < MAXMOD; ++m) x.p[m] = m < Mn ? X[m] : nullptr;
    int ny = 1024 / (B * P);                 // aim at ~4 workgroups per CU
    if (ny > (L + 7) / 8) ny = (L + 7) / 8;  // at least 8 rows per slice
    if (ny < 1) ny = 1;
    hipLaunchKernelGGL(party_gather_kernel, dim3(B * P, ny), dim3(256), 0, (hipStream_t)stream, x, qmask, bias, S, rank,
                       L, B, P, Mn, H);
    MMDFN_CHECK_LAUNCH();
    return 0;
}

extern "C" int mmdfn_party_gather_bwd(int Mn, const float* dS, const int32_t* rank, float* const* dX,
                                      const float* const* addend, int L, int B, int P, int H, void* stream) {
    if (Mn <= 0 || Mn > MAXMOD || L <= 0 || B <= 0 || P <= 0 || H <= 0 || (H & 3)) return -1;
    ModPtrsW x;
    ModPtrs a;
    for (int m = 0; m < MAXMOD; ++m) {
        x.p[m] = m < Mn ? dX[m] : nullptr;
        a.p[m] = (addend && m < Mn) ? addend[m] : nullptr;
    }
    hipLaunchKernelGGL(party_gather_bwd_kernel, dim3(grid_for((int64_t)Mn * L * B * (H / 4))), dim3(256), 0,
                       (hipStream_t)stream, dS, rank, x, a, L, B, P, Mn, H);
    MMDFN_CHECK_LAUNCH();
    return 0;
}

// mmdfn_party_gather_bwd + mmdfn_colsum_partial(dS as (L * Mn*B*P) x H, workspace) in ONE launch; returns the number of [H]
// slabs in `workspace` (> 0) like mmdfn_colsum_partial, negative = rejected
extern "C" int mmdfn_party_gather_bwd_colsum(int Mn, const float* dS, const int32_t* rank, float* const* dX,
                                             const float* const* addend, int L, int B, int P, int H, float* workspace,
                                             void* stream) {
    if (Mn <= 0 || Mn > MAXMOD || L <= 0 || B <= 0 || P <= 0 || H <= 0 || (H & 3) || workspace == nullptr) return -1;
    if (reinterpret_cast<uintptr_t>(dS) & 15) return -1;
    ModPtrsW x;
    ModPtrs a;
    for (int m = 0; m < MAXMOD; ++m) {
        x.p[m] = m < Mn ? dX[m] : nullptr;
        a.p[m] = (addend && m < Mn) ? addend[m] : nullptr;
    }
    const int64_t R = (int64_t)L * Mn * B * P;
    const int ncb = (H + 63) / 64;
    if (ncb > 64) return -1;
    int nsl = COLSUM_SLABS;
    if ((int64_t)nsl * 64 > R) nsl = (int)((R + 63) / 64);
    const int ngather = grid_for((int64_t)Mn * L * B * (H / 4));
    hipLaunchKernelGGL(party_gather_bwd_colsum_kernel, dim3(ngather + ncb * nsl), dim3(256), 0, (hipStream_t)stream, dS, rank, x, a, L,
                       B, P, Mn, H, ngather, R, ncb, nsl, workspace);
    if (hipGetLastError() != hipSuccess) return -2;
    return nsl;
}

extern "C" int mmdfn_party_combine(int Mn, const float* const* base, const float* E, const int32_t* rank,
                                   const int64_t* flat_idx, float* out, const float* weights, int L, int B, int P,
                                   int N, int H, void* stream) {
    if (Mn <= 0 || Mn > MAXMOD || L <= 0 || B <= 0 || P <= 0 || N <= 0 || H <= 0 || (H & 3)) return -1;
    ModPtrs x;
    float w[MAXMOD] = {0, 0, 0, 0};
    for (int m = 0; m < MAXMOD; ++m) {
        x.p[m] = m < Mn ? base[m] : nullptr;
        if (m < Mn) w[m] = weights[m];
    }
    hipLaunchKernelGGL(party_combine_kernel, dim3(grid_for((int64_t)Mn * N * (H / 4))), dim3(256), 0,
                       (hipStream_t)stream, x, E, rank, flat_idx, out, w[0], w[1], w[2], w[3], L, B, P, Mn, N, H);
    MMDFN_CHECK_LAUNCH();
    return 0;
}

extern "C" int mmdfn_party_combine_bwd(int Mn, const float* dout, const int32_t* rank, const int64_t* flat_idx,
                                       float* const* dbase, float* dE, const float* weights, int L, int B, int P,
                                       int N, int H, void* stream) {
    if (Mn <= 0 || Mn > MAXMOD || L <= 0 || B <= 0 || P <= 0 || N <= 0 || H <= 0 || (H & 3)) return -1;
    ModPtrsW x;
    float w[MAXMOD] = {0, 0, 0, 0};
    for (int m = 0; m < MAXMOD; ++m) {
        x.p[m] = m < Mn ? dbase[m] : nullptr;
        if (m < Mn) w[m] = weights[m];
    }
    hipLaunchKernelGGL(party_combine_bwd_kernel, dim3(grid_for((int64_t)Mn * N * (H / 4))), dim3(256), 0,
                       (hipStream_t)stream, dout, rank, flat_idx, x, dE, w[0], w[1], w[2], w[3], L, B, P, Mn, N, H);
    MMDFN_CHECK_LAUNCH();
    return 0;
}

// mmdfn_party_combine_bwd without the caller's zero fill: dbase / dE need NOT be initialised.  inv: (L * B) int64, the row of
// (t, b) in the stripped order or -1 (the inverse of flat_idx).  -2: shape not covered (L > 2048): use the pre-zeroed form.
extern "C" int mmdfn_party_combine_bwd_dst(int Mn, const float* dout, const int32_t* rank, const int64_t* inv,
                                           float* const* dbase, float* dE, const float* weights, int L, int B, int P, int N,
                                           int H, void* stream) {
    if (Mn <= 0 || Mn > MAXMOD || L <= 0 || B <= 0 || P <= 0 || N <= 0 || H <= 0 || (H & 3) || inv == nullptr) return -1;
    if (L > MAXL) return -2;
    ModPtrsW x;
    float w[MAXMOD] = {0, 0, 0, 0};
    for (int m = 0; m < MAXMOD; ++m) {
        x.p[m] = m < Mn ? dbase[m] : nullptr;
        if (m < Mn) w[m] = weights[m];
    }
    int ny = 1024 / (B * P);                 // row slices: B * P workgroups alone leave most of the chip idle
    if (ny > (L + 7) / 8) ny = (L + 7) / 8;
    if (ny < (L + 255) / 256) ny = (L + 255) / 256;      // (a slice's rows are looked up by one thread each)
    if (ny < 1) ny = 1;
    hipLaunchKernelGGL(party_combine_bwd_dst_kernel, dim3(B * P, ny), dim3(256), 0, (hipStream_t)stream, dout, rank, inv, x, dE,
                       w[0], w[1], w[2], w[3], L, B, P, Mn, N, H);
    MMDFN_CHECK_LAUNCH();
    return 0;
}

extern "C" int mmdfn_mask_scale(int ngroups, const float* const* x, const float* const* mask, float* const* out,
                                const int64_t* n, float scale, void* stream) {
    if (ngroups <= 0 || ngroups > MAXMOD) return -1;
    MaskScaleGroups G;
    int64_t acc = 0;
    for (int g = 0; g < MAXMOD; ++g) {
        G.start4[g] = acc;
        G.x[g] = G.mask[g] = nullptr;
        G.out[g] = nullptr;
        if (g < ngroups) {
            if (n[g] <= 0 || (n[g] & 3)) return -1;
            G.x[g] = x[g]; G.mask[g] = mask[g]; G.out[g] = out[g];
            acc += n[g] / 4;
        }
    }
    for (int g = ngroups; g <= MAXMOD; ++g) G.start4[g] = acc;
    hipLaunchKernelGGL(mask_scale_kernel, dim3(grid_for(acc)), dim3(256), 0, (hipStream_t)stream, G, ngroups, scale);
    MMDFN_CHECK_LAUNCH();
    return 0;
}

extern "C" int64_t mmdfn_colsum_workspace(int H) { return (int64_t)COLSUM_SLABS * H; }

extern "C" int mmdfn_colsum(const float* A, int64_t R, int H, int lda, float* out, float* workspace, void* stream) {
    if (R <= 0 || H <= 0 || (H & 3) || lda < H || (lda & 3) || (reinterpret_cast<uintptr_t>(A) & 15) ||
        (reinterpret_cast<uintptr_t>(out) & 15)) return -1;
    const int ncb = (H + 63) / 64;
    if (ncb > 64) return -1;
    int nsl = COLSUM_SLABS;
    if ((int64_t)nsl * 64 > R) nsl = (int)((R + 63) / 64);
    hipLaunchKernelGGL(colsum_kernel, dim3(ncb, nsl), dim3(256), 0, (hipStream_t)stream, A, R, H, lda, workspace);
    MMDFN_CHECK_LAUNCH();
    hipLaunchKernelGGL(colsum_final_kernel, dim3((H + 63) / 64), dim3(64), 0, (hipStream_t)stream, workspace, nsl, H, out);
    MMDFN_CHECK_LAUNCH();
    return 0;
}

// The first launch of mmdfn_colsum alone: workspace then holds the returned number (> 0) of [H] slabs for
// mmdfn_gemm_tn_batch_ext to sum; negative = rejected.
extern "C" int mmdfn_colsum_partial(const float* A, int64_t R, int H, int lda, float* workspace, void* stream) {
    if (R <= 0 || H <= 0 || (H & 3) || lda < H || (lda & 3) || (reinterpret_cast<uintptr_t>(A) & 15)) return -1;
    const int ncb = (H + 63) / 64;
    if (ncb > 64) return -1;
    int nsl = COLSUM_SLABS;
    if ((int64_t)nsl * 64 > R) nsl = (int)((R + 63) / 64);
    hipLaunchKernelGGL(colsum_kernel, dim3(ncb, nsl), dim3(256), 0, (hipStream_t)stream, A, R, H, lda, workspace);
    if (hipGetLastError() != hipSuccess) return -2;
    return nsl;
}

// A draw STAGED for the next plain GRU forward launch (gru.hip: its tiles of counters run as rider workgroups of the recurrence
// launch) instead of launched; mmdfn_keep_flags_flush launches a staged draw the ordinary way.
static kfb::FlagJob g_flag_job;
static bool g_flag_job_valid = false;
const kfb::FlagJob* mmdfn_flag_job_pending() { return g_flag_job_valid ? &g_flag_job : nullptr; }
void mmdfn_flag_job_taken() { g_flag_job_valid = false; }

static int flag_job(kfb::FlagJob& J, float* out, int64_t n, float keep, void* state) {
    if (n <= 0 || (n & 3) || (reinterpret_cast<uintptr_t>(out) & 15) || state == nullptr || !(keep >= 0.f) || keep > 1.f) return -1;
    J.out = out;
    J.n4 = n >> 2;
    J.n8 = (((J.n4 + 1) >> 1) + 63) & ~(int64_t)63;      // Philox counters consumed: whole waves
    // keep = 1: every flag is 1 whatever the draw
    J.all = keep >= 1.f ? 1 : 0;
    J.threshold = J.all ? 65536u : (uint32_t)((double)keep * 65536.0 + 0.5);
    J.state = reinterpret_cast<unsigned long long*>(state);
    return 0;
}

static int launch_flag_job(const kfb::FlagJob& J, hipStream_t s) {
    // one workgroup per CU at most: the end-of-launch counter is one L2 atomic per workgroup on ONE address (2 048 of them
    // serialised into 20 us); 1 024 threads each, so that four waves per SIMD hide the 10-round dependent chain of a Philox call
    int64_t grid = (J.n8 + 1023) / 1024;
    if (grid > 256) grid = 256;
    hipLaunchKernelGGL(keep_flags_kernel, dim3((unsigned)grid), dim3(1024), 0, s, J);
    MMDFN_CHECK_LAUNCH();
    return 0;
}

extern "C" int mmdfn_keep_flags(float* out, int64_t n, float keep, void* state, void* stream) {
    kfb::FlagJob J;
    if (int e = flag_job(J, out, n, keep, state)) return e;
    return launch_flag_job(J, (hipStream_t)stream);
}

extern "C" int mmdfn_keep_flags_stage(float* out, int64_t n, float keep, void* state, void* stream) {
    kfb::FlagJob J;
    if (int e = flag_job(J, out, n, keep, state)) return e;
    if (g_flag_job_valid) return launch_flag_job(J, (hipStream_t)stream);       // one staged draw at a time: this one goes now
    g_flag_job = J;
    g_flag_job_valid = true;
    return 0;
}

extern "C" int mmdfn_keep_flags_flush(void* stream) {
    if (!g_flag_job_valid) return 0;
    g_flag_job_valid = false;
    return launch_flag_job(g_flag_job, (hipStream_t)stream);
}
