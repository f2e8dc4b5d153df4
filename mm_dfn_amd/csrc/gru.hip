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
//     thread (u, s) keeps the r/z/n rows of hidden unit u restricted to the k-slice s
//     (3 x 20 floats); h_{t-1} is broadcast from LDS; the KS partial sums meet in LDS and the
//     R*H "gate threads" apply the sigmoid/tanh gate math, write y_t, the saved gates and h_t.
//     GI for step t+1 is prefetched into registers while step t computes.
//   * several independent GRUs (text context + party batch) share ONE launch ("groups").
//
// Gate order r, z, n;  n = tanh(gi_n + r * (W_hn h + b_hn));  h = (1-z) n + z h_prev.
#include "mmdfn_internal.h"
#include "../../include/mmdfn_hip.h"

namespace {

constexpr int GH = 100;          // hidden size (the reference hard-codes D_e = 100, model.py:847-849)
constexpr int KS = 5;            // k-slices
constexpr int KL = GH / KS;      // 20 hidden inputs per slice
constexpr int JL = 3 * GH / KS;  // 60 gate rows per slice (backward)
constexpr int NT = 512;
constexpr int MAXG = 4;

struct FwdGroups {
    int n;
    const float* gi[MAXG];
    const float* w_hh[MAXG];
    const float* b_hh[MAXG];
    float* y[MAXG];
    float* gates[MAXG];
    int rows[MAXG];
    int T[MAXG];
    int slice0[MAXG + 1];
};

struct BwdGroups {
    int n;
    const float* dy[MAXG];
    const float* y[MAXG];
    const float* gates[MAXG];
    const float* w_hh[MAXG];
    float* dgi[MAXG];
    float* dgh[MAXG];
    int rows[MAXG];
    int T[MAXG];
    int slice0[MAXG + 1];
};

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

template <int R>
__global__ __launch_bounds__(NT) void gru_seq_fwd_kernel(FwdGroups G) {
    __shared__ float hs[R][GH];
    __shared__ float part[KS][3][R][GH];

    int gidx = 0;
    while (gidx + 1 < G.n && (int)blockIdx.x >= G.slice0[gidx + 1]) ++gidx;
    const int dir = blockIdx.y;
    const int rows = G.rows[gidx];
    const int T = G.T[gidx];
    const int row0 = ((int)blockIdx.x - G.slice0[gidx]) * R;
    const float* __restrict__ gi = G.gi[gidx];
    const float* __restrict__ w_hh = G.w_hh[gidx] + (int64_t)dir * 3 * GH * GH;
    const float* __restrict__ b_hh = G.b_hh[gidx] + dir * 3 * GH;
    float* __restrict__ y = G.y[gidx];
    float* __restrict__ gates = G.gates[gidx];

    const int tid = threadIdx.x;
    const int u = tid % GH;
    const int s = tid / GH;
    const bool mv = tid < KS * GH;

    float wr[KL], wz[KL], wn[KL];
    if (mv) {
#pragma unroll
        for (int k = 0; k < KL; ++k) {
            wr[k] = w_hh[(int64_t)(0 * GH + u) * GH + s * KL + k];
            wz[k] = w_hh[(int64_t)(1 * GH + u) * GH + s * KL + k];
            wn[k] = w_hh[(int64_t)(2 * GH + u) * GH + s * KL + k];
        }
    }
    // gate-thread role: (r, u) for tid < R*GH
    const int gr = tid / GH;
    const bool gate = (tid < R * GH) && (row0 + gr < rows);
    const int row = row0 + gr;
    float bhr = 0.f, bhz = 0.f, bhn = 0.f, hprev = 0.f;
    if (gate) {
        bhr = b_hh[u];
        bhz = b_hh[GH + u];
        bhn = b_hh[2 * GH + u];
    }
    for (int i = tid; i < R * GH; i += NT) (&hs[0][0])[i] = 0.f;

    float gir = 0.f, giz = 0.f, gin = 0.f;
    auto load_gi = [&](int t) {
        const float* p = gi + ((int64_t)t * rows + row) * (6 * GH) + dir * 3 * GH + u;
        gir = p[0];
        giz = p[GH];
        gin = p[2 * GH];
    };
    if (gate && T > 0) load_gi(dir ? T - 1 : 0);
    __syncthreads();

    for (int step = 0; step < T; ++step) {
        const int t = dir ? T - 1 - step : step;
        if (mv) {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                float ar = 0.f, az = 0.f, an = 0.f;
                const float4* hv = reinterpret_cast<const float4*>(&hs[r][s * KL]);
#pragma unroll
                for (int k4 = 0; k4 < KL / 4; ++k4) {
                    const float4 h4 = hv[k4];
                    ar = fmaf(wr[4 * k4 + 0], h4.x, ar); az = fmaf(wz[4 * k4 + 0], h4.x, az); an = fmaf(wn[4 * k4 + 0], h4.x, an);
                    ar = fmaf(wr[4 * k4 + 1], h4.y, ar); az = fmaf(wz[4 * k4 + 1], h4.y, az); an = fmaf(wn[4 * k4 + 1], h4.y, an);
                    ar = fmaf(wr[4 * k4 + 2], h4.z, ar); az = fmaf(wz[4 * k4 + 2], h4.z, az); an = fmaf(wn[4 * k4 + 2], h4.z, an);
                    ar = fmaf(wr[4 * k4 + 3], h4.w, ar); az = fmaf(wz[4 * k4 + 3], h4.w, az); an = fmaf(wn[4 * k4 + 3], h4.w, an);
                }
                part[s][0][r][u] = ar;
                part[s][1][r][u] = az;
                part[s][2][r][u] = an;
            }
        }
        __syncthreads();
        if (gate) {
            float ghr = bhr, ghz = bhz, ghn = bhn;
#pragma unroll
            for (int q = 0; q < KS; ++q) {
                ghr += part[q][0][gr][u];
                ghz += part[q][1][gr][u];
                ghn += part[q][2][gr][u];
            }
            const float rr = sigmoidf_(gir + ghr);
            const float zz = sigmoidf_(giz + ghz);
            const float nn = tanhf(gin + rr * ghn);
            const float hnew = (1.0f - zz) * nn + zz * hprev;
            const int64_t o = ((int64_t)t * rows + row);
            y[o * (2 * GH) + dir * GH + u] = hnew;
            float* gp = gates + (o * 2 + dir) * (4 * GH) + u;
            gp[0] = rr;
            gp[GH] = zz;
            gp[2 * GH] = nn;
            gp[3 * GH] = ghn;
            hs[gr][u] = hnew;
            hprev = hnew;
            if (step + 1 < T) load_gi(dir ? t - 1 : t + 1);
        }
        __syncthreads();
    }
}

template <int R>
__global__ __launch_bounds__(NT) void gru_seq_bwd_kernel(BwdGroups G) {
    __shared__ float dghs[R][3 * GH];
    __shared__ float part[KS][R][GH];

    int gidx = 0;
    while (gidx + 1 < G.n && (int)blockIdx.x >= G.slice0[gidx + 1]) ++gidx;
    const int dir = blockIdx.y;
    const int rows = G.rows[gidx];
    const int T = G.T[gidx];
    const int row0 = ((int)blockIdx.x - G.slice0[gidx]) * R;
    const float* __restrict__ dy = G.dy[gidx];
    const float* __restrict__ y = G.y[gidx];
    const float* __restrict__ gates = G.gates[gidx];
    const float* __restrict__ w_hh = G.w_hh[gidx] + (int64_t)dir * 3 * GH * GH;
    float* __restrict__ dgi = G.dgi[gidx];
    float* __restrict__ dgh = G.dgh[gidx];

    const int tid = threadIdx.x;
    const int u = tid % GH;
    const int s = tid / GH;
    const bool mv = tid < KS * GH;
    // dh_prev[u] = sum_j dgh[j] W_hh[j][u]; this thread covers j in [s*JL, s*JL+JL)
    float w[JL];
    if (mv) {
#pragma unroll
        for (int j = 0; j < JL; ++j) w[j] = w_hh[(int64_t)(s * JL + j) * GH + u];
    }
    const int gr = tid / GH;
    const bool gate = (tid < R * GH) && (row0 + gr < rows);
    const int row = row0 + gr;
    float carry = 0.f;  // dh * z carried to the previous timestep

    float p_dy = 0.f, p_r = 0.f, p_z = 0.f, p_n = 0.f, p_ghn = 0.f, p_hprev = 0.f;
    auto prefetch = [&](int t) {
        const int64_t o = ((int64_t)t * rows + row);
        p_dy = dy[o * (2 * GH) + dir * GH + u];
        const float* gp = gates + (o * 2 + dir) * (4 * GH) + u;
        p_r = gp[0];
        p_z = gp[GH];
        p_n = gp[2 * GH];
        p_ghn = gp[3 * GH];
        const int tp = dir ? t + 1 : t - 1;  // the step that produced h_prev in the forward pass
        p_hprev = (tp >= 0 && tp < T) ? y[((int64_t)tp * rows + row) * (2 * GH) + dir * GH + u] : 0.f;
    };
    if (gate && T > 0) prefetch(dir ? 0 : T - 1);

    for (int step = 0; step < T; ++step) {
        const int t = dir ? step : T - 1 - step;  // reverse of the forward order
        if (gate) {
            float dh = p_dy + carry;
            if (step > 0) {
#pragma unroll
                for (int q = 0; q < KS; ++q) dh += part[q][gr][u];
            }
            const float rr = p_r, zz = p_z, nn = p_n, ghn = p_ghn, hprev = p_hprev;
            const float dn = dh * (1.0f - zz);
            const float dz = dh * (hprev - nn);
            carry = dh * zz;
            const float dnpre = dn * (1.0f - nn * nn);
            const float drpre = dnpre * ghn * rr * (1.0f - rr);
            const float dzpre = dz * zz * (1.0f - zz);
            const float dghn = dnpre * rr;
            const int64_t o = ((int64_t)t * rows + row) * (6 * GH) + dir * 3 * GH + u;
            dgi[o] = drpre;
            dgi[o + GH] = dzpre;
            dgi[o + 2 * GH] = dnpre;
            dgh[o] = drpre;
            dgh[o + GH] = dzpre;
            dgh[o + 2 * GH] = dghn;
            dghs[gr][u] = drpre;
            dghs[gr][GH + u] = dzpre;
            dghs[gr][2 * GH + u] = dghn;
            if (step + 1 < T) prefetch(dir ? t + 1 : t - 1);
        }
        __syncthreads();
        if (mv) {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                float a = 0.f;
                const float4* dv = reinterpret_cast<const float4*>(&dghs[r][s * JL]);
#pragma unroll
                for (int j4 = 0; j4 < JL / 4; ++j4) {
                    const float4 d4 = dv[j4];
                    a = fmaf(w[4 * j4 + 0], d4.x, a);
                    a = fmaf(w[4 * j4 + 1], d4.y, a);
                    a = fmaf(w[4 * j4 + 2], d4.z, a);
                    a = fmaf(w[4 * j4 + 3], d4.w, a);
                }
                part[s][r][u] = a;
            }
        }
        __syncthreads();
    }
}

int pick_r(int ngroups, const int* rows) {
    for (int R : {1, 2, 4}) {
        int wg = 0;
        for (int g = 0; g < ngroups; ++g) wg += (rows[g] + R - 1) / R;
        if (2 * wg <= 512) return R;
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
    const int R = pick_r(ngroups, rows);
    int sl = 0;
    for (int g = 0; g < ngroups; ++g) {
        if (rows[g] <= 0 || T[g] <= 0) return -1;
        G.gi[g] = gi[g]; G.w_hh[g] = w_hh[g]; G.b_hh[g] = b_hh[g]; G.y[g] = y[g]; G.gates[g] = gates[g];
        G.rows[g] = rows[g]; G.T[g] = T[g]; G.slice0[g] = sl;
        sl += (rows[g] + R - 1) / R;
    }
    G.slice0[ngroups] = sl;
    dim3 grid(sl, 2), block(NT);
    hipStream_t s = (hipStream_t)stream;
    if (R == 1) hipLaunchKernelGGL(gru_seq_fwd_kernel<1>, grid, block, 0, s, G);
    else if (R == 2) hipLaunchKernelGGL(gru_seq_fwd_kernel<2>, grid, block, 0, s, G);
    else hipLaunchKernelGGL(gru_seq_fwd_kernel<4>, grid, block, 0, s, G);
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
        G.dy[g] = dy[g]; G.y[g] = y[g]; G.gates[g] = gates[g]; G.w_hh[g] = w_hh[g]; G.dgi[g] = dgi[g];
        G.dgh[g] = dgh[g]; G.rows[g] = rows[g]; G.T[g] = T[g]; G.slice0[g] = sl;
        sl += (rows[g] + R - 1) / R;
    }
    G.slice0[ngroups] = sl;
    dim3 grid(sl, 2), block(NT);
    hipStream_t s = (hipStream_t)stream;
    if (R == 1) hipLaunchKernelGGL(gru_seq_bwd_kernel<1>, grid, block, 0, s, G);
    else if (R == 2) hipLaunchKernelGGL(gru_seq_bwd_kernel<2>, grid, block, 0, s, G);
    else hipLaunchKernelGGL(gru_seq_bwd_kernel<4>, grid, block, 0, s, G);
    MMDFN_CHECK_LAUNCH();
    return 0;
}
