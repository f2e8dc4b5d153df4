// K9: classifier head  log_softmax( relu(dropout(F)) W^T + b )  as one launch each way.
//
// Replaces  dropout_ -> ReLU -> smax_fc -> log_softmax  (reference model.py:1328-1337) and its autograd (soft-max
// backward, two GEMMs, a bias reduction, the ReLU and dropout masks: ten library launches).  NB the reference applies
// the ReLU to the whole fused feature row, including the raw residual x.
//   z = relu(F (.) m * mscale)            F: (N, W) rows (stride ldf), m: 0/1 keep-mask (N, W) or NULL, mscale = 1/(1-p)
//   logit_c = z . W_c + b_c ;  logp = logit - logsumexp(logit)                                   C <= 8 classes
// The weight (C x W, a few tens of KB) lives in LDS; one wave per row walks the row in 16-byte chunks, the C dot
// products are reduced across the wave; C is far too small for the matrix cores to matter (2 N W C flop = 19 MFLOP at
// cfg2) -- the kernel is a single pass over F at HBM / L2 speed.
// Backward:  g = dlogp - exp(logp) * sum_c dlogp_c;   dF = (g W) (.) [z > 0] m mscale;   dW = g^T z;   db = sum_rows g.
//   dW / db: every wave keeps its partial sums in registers over the rows it owns, waves of a workgroup meet in LDS,
//   workgroups write partial slabs and a second tiny kernel sums them in a fixed order (bit-reproducible, no atomics).
#include "mmdfn_internal.h"
#include "../../include/mmdfn_hip.h"

namespace {

constexpr int HC_MAX = 8;        // classes (IEMOCAP 6, MELD 7); wider heads stay on the library path
constexpr int HB_COLS = 1024;    // feature columns per backward column block (4 float4 per lane)
constexpr int HB_GROUPS = 256;   // row groups (workgroups per column block) of the backward pass: a wave walks ~5 rows at N = 5 280 (each row is a memory round trip)

// Element (row, col) of the feature matrix.  split = 0: plain rows of stride ld.  split = Wm > 0: the matrix is the
// column-wise concatenation of W / Wm blocks that live one after the other as (N, Wm) matrices of row stride ld -- the
// (M, N, Wm) output of the graph stack read as cat([F[0], F[1], ...], -1) (model_mm.py:113-117) without the copy.
// Wm is a multiple of 4, so a 16-byte chunk never straddles two blocks.
__device__ __forceinline__ int64_t feat_off(int64_t row, int col, int ld, int split, int64_t N) {
    if (split <= 0) return row * ld + col;
    const int blk = col / split;
    return ((int64_t)blk * N + row) * ld + (col - blk * split);
}

__global__ __launch_bounds__(256) void head_fwd_kernel(const float* __restrict__ Fm, const float* __restrict__ mask,
                                                       const float* __restrict__ Wt, const float* __restrict__ bias,
                                                       float* __restrict__ logp, int64_t N, int W, int C, int ldf,
                                                       int split, float mscale) {
    extern __shared__ __attribute__((aligned(16))) float sW[];       // [C][W]
    {   // batches of 8 independent 16-byte loads per thread (one load in flight per thread would pay the latency 6x)
        const int total = C * W / 4;
        for (int base = threadIdx.x; base < total; base += 256 * 8) {
            // (unconditional loads from a clamped index: a guarded load into v[e] makes hipcc keep v[] in scratch and wait
            // for every load before the next -- 8 serial round trips and a private segment per launch, 15 us for this kernel)
            float4 v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int i = base + 256 * e;
                v[e] = reinterpret_cast<const float4*>(Wt)[i < total ? i : total - 1];
            }
            // (the stores go to the clamped slot as well: behind a guard hipcc sinks each load into its store's branch again)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int i = base + 256 * e;
                reinterpret_cast<float4*>(sW)[i < total ? i : total - 1] = v[e];
            }
        }
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int W4 = W / 4;
    for (int64_t row = (int64_t)blockIdx.x * 4 + w; row < N; row += (int64_t)gridDim.x * 4) {
        float acc[HC_MAX];
#pragma unroll
        for (int c = 0; c < HC_MAX; ++c) acc[c] = 0.f;
        // all of the row's 16-byte loads (features and keep flags) go out before the first use: one memory round trip per
        // row instead of one per 64-lane slice (W = 900: 4 slices; the kernel is a 19 MB stream, 16.5 -> 8 us at N = 5 280)
        constexpr int NJ = 4;
        for (int j0 = 0; j0 < W4; j0 += 64 * NJ) {
            float4 zv[NJ], mv[NJ];
#pragma unroll
            for (int u = 0; u < NJ; ++u) {                        // unconditional loads from clamped columns (selected at use)
                const int j = j0 + 64 * u + lane;
                const int jc = j < W4 ? j : W4 - 1;
                zv[u] = *reinterpret_cast<const float4*>(Fm + feat_off(row, 4 * jc, ldf, split, N));
                mv[u] = *reinterpret_cast<const float4*>((mask ? mask : Fm) + (mask ? row * W + 4 * jc : feat_off(row, 4 * jc, ldf, split, N)));
            }
#pragma unroll
            for (int u = 0; u < NJ; ++u) {
                const int j = j0 + 64 * u + lane;
                if (j >= W4) continue;
                float4 z = zv[u];
                if (mask) {
                    const float4 m = mv[u];
                    z.x *= m.x * mscale; z.y *= m.y * mscale; z.z *= m.z * mscale; z.w *= m.w * mscale;
                }
                z.x = fmaxf(z.x, 0.f); z.y = fmaxf(z.y, 0.f); z.z = fmaxf(z.z, 0.f); z.w = fmaxf(z.w, 0.f);
#pragma unroll
                for (int c = 0; c < HC_MAX; ++c) {
                    if (c < C) {
                        const float4 wv = *reinterpret_cast<const float4*>(sW + c * W + 4 * j);
                        acc[c] += z.x * wv.x + z.y * wv.y + z.z * wv.z + z.w * wv.w;
                    }
                }
            }
        }
        float mx = -3.0e38f;
#pragma unroll
        for (int c = 0; c < HC_MAX; ++c) {
            if (c < C) {
                acc[c] = wave_sum(acc[c]) + bias[c];
                mx = fmaxf(mx, acc[c]);
            }
        }
        float se = 0.f;
#pragma unroll
        for (int c = 0; c < HC_MAX; ++c)
            if (c < C) se += expf(acc[c] - mx);
        const float lse = mx + logf(se);
        if (lane < C) {
            float v = 0.f;
#pragma unroll
            for (int c = 0; c < HC_MAX; ++c)
                if (c == lane) v = acc[c];
            logp[row * C + lane] = v - lse;
        }
    }
}

// grid (HB_GROUPS, column blocks).  part: [groups][C][W] slabs of dW, bpart: [groups][C] slabs of db (column block 0).
__global__ __launch_bounds__(256) void head_bwd_kernel(const float* __restrict__ dlogp, const float* __restrict__ logp,
                                                       const float* __restrict__ Fm, const float* __restrict__ mask,
                                                       const float* __restrict__ Wt, float* __restrict__ dF,
                                                       float* __restrict__ part, float* __restrict__ bpart, int64_t N, int W,
                                                       int C, int ldf, int lddf, int split, float mscale) {
    extern __shared__ __attribute__((aligned(16))) float sm[];       // [C][cols] weight block, later the wave partials
    const int col0 = blockIdx.y * HB_COLS;
    const int cols = min(HB_COLS, W - col0);
    const int cols4 = cols / 4;
    for (int base = threadIdx.x; base < C * cols4; base += 256 * 8) {
        float4 v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int i0 = base + 256 * e, i = i0 < C * cols4 ? i0 : C * cols4 - 1, c = i / cols4, j = i - c * cols4;
            v[e] = *reinterpret_cast<const float4*>(Wt + (int64_t)c * W + col0 + 4 * j);      // (clamped, unconditional: see head_fwd)
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int i0 = base + 256 * e, i = i0 < C * cols4 ? i0 : C * cols4 - 1, c = i / cols4, j = i - c * cols4;
            reinterpret_cast<float4*>(sm)[c * (HB_COLS / 4) + j] = v[e];                              // (clamped slot: no guard)
        }
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    float4 dw[HC_MAX][HB_COLS / 256];
    float db[HC_MAX];
#pragma unroll
    for (int c = 0; c < HC_MAX; ++c) {
        db[c] = 0.f;
#pragma unroll
        for (int s = 0; s < HB_COLS / 256; ++s) dw[c][s] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    // A wave walks ~10 rows (5 280 rows on 512 waves) and every row is one memory round trip, so the NEXT row's operands are
    // requested before the current row is worked on (20.7 -> 12 us at N = 5 280)
    constexpr int NSL = HB_COLS / 256;
    const int64_t rstep = (int64_t)gridDim.x * 4;
    float gq[HC_MAX], lq[HC_MAX];
    float4 zq[NSL], mq[NSL];
    auto fetch = [&](int64_t row) {
#pragma unroll
        for (int c = 0; c < HC_MAX; ++c) {
            gq[c] = (c < C && row < N) ? dlogp[row * C + c] : 0.f;
            lq[c] = (c < C && row < N) ? logp[row * C + c] : 0.f;
        }
#pragma unroll
        for (int s = 0; s < NSL; ++s) {
            const int j = lane + 64 * s;
            const bool ok = j < cols4 && row < N;
            zq[s] = ok ? *reinterpret_cast<const float4*>(Fm + feat_off(row, col0 + 4 * j, ldf, split, N)) : make_float4(0.f, 0.f, 0.f, 0.f);
            mq[s] = (ok && mask) ? *reinterpret_cast<const float4*>(mask + row * W + col0 + 4 * j) : make_float4(1.f, 1.f, 1.f, 1.f);
        }
    };
    fetch((int64_t)blockIdx.x * 4 + w);
    for (int64_t row = (int64_t)blockIdx.x * 4 + w; row < N; row += rstep) {
        float g[HC_MAX], lp[HC_MAX];
        float4 zc[NSL], mc[NSL];
#pragma unroll
        for (int c = 0; c < HC_MAX; ++c) { g[c] = gq[c]; lp[c] = lq[c]; }
#pragma unroll
        for (int s = 0; s < NSL; ++s) { zc[s] = zq[s]; mc[s] = mq[s]; }
        fetch(row + rstep);
        float sd = 0.f;
#pragma unroll
        for (int c = 0; c < HC_MAX; ++c) sd += g[c];
#pragma unroll
        for (int c = 0; c < HC_MAX; ++c) {
            if (c < C) {
                g[c] -= expf(lp[c]) * sd;
                db[c] += g[c];                        // (every lane holds the same value; lane 0 reports it)
            }
        }
#pragma unroll
        for (int s = 0; s < NSL; ++s) {
            const int j = lane + 64 * s;
            if (j >= cols4) continue;
            float4 z = zc[s];
            float4 m = mc[s];
            if (mask) { m.x *= mscale; m.y *= mscale; m.z *= mscale; m.w *= mscale; }
            z.x = fmaxf(z.x * m.x, 0.f); z.y = fmaxf(z.y * m.y, 0.f); z.z = fmaxf(z.z * m.z, 0.f); z.w = fmaxf(z.w * m.w, 0.f);
            float4 d = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int c = 0; c < HC_MAX; ++c) {
                if (c < C) {
                    const float4 wv = reinterpret_cast<const float4*>(sm)[c * (HB_COLS / 4) + j];
                    d.x += g[c] * wv.x; d.y += g[c] * wv.y; d.z += g[c] * wv.z; d.w += g[c] * wv.w;
                    dw[c][s].x += g[c] * z.x; dw[c][s].y += g[c] * z.y; dw[c][s].z += g[c] * z.z; dw[c][s].w += g[c] * z.w;
                }
            }
            d.x = z.x > 0.f ? d.x * m.x : 0.f; d.y = z.y > 0.f ? d.y * m.y : 0.f;
            d.z = z.z > 0.f ? d.z * m.z : 0.f; d.w = z.w > 0.f ? d.w * m.w : 0.f;
            *reinterpret_cast<float4*>(dF + feat_off(row, col0 + 4 * j, lddf, split, N)) = d;
        }
    }
    // waves -> LDS -> one slab per workgroup (fixed order)
    __syncthreads();                                     // the weight block is no longer needed
    float4* sp = reinterpret_cast<float4*>(sm);          // [4 waves][C][HB_COLS / 4]
#pragma unroll
    for (int c = 0; c < HC_MAX; ++c) {
        if (c < C) {
#pragma unroll
            for (int s = 0; s < HB_COLS / 256; ++s) sp[(w * C + c) * (HB_COLS / 4) + lane + 64 * s] = dw[c][s];
        }
    }
    __shared__ float sdb[4][HC_MAX];
    if (lane == 0) {
#pragma unroll
        for (int c = 0; c < HC_MAX; ++c) sdb[w][c] = db[c];
    }
    __syncthreads();
    float* pout = part + (int64_t)blockIdx.x * C * W;
    for (int i = threadIdx.x; i < C * cols4; i += 256) {
        const int c = i / cols4, j = i - c * cols4;
        float4 s = sp[(0 * C + c) * (HB_COLS / 4) + j];
#pragma unroll
        for (int ww = 1; ww < 4; ++ww) {
            const float4 v = sp[(ww * C + c) * (HB_COLS / 4) + j];
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
        *reinterpret_cast<float4*>(pout + (int64_t)c * W + col0 + 4 * j) = s;
    }
    if (blockIdx.y == 0 && threadIdx.x < C)
        bpart[blockIdx.x * C + threadIdx.x] = sdb[0][threadIdx.x] + sdb[1][threadIdx.x] + sdb[2][threadIdx.x] + sdb[3][threadIdx.x];
}

// sums the per-workgroup slabs: 8 lanes per output element, each adds its share of the slabs in a fixed order (all
// its loads in flight at once), the 8 partial sums meet with shuffles in a fixed tree -> bit-reproducible
__global__ __launch_bounds__(256) void head_reduce_kernel(const float* __restrict__ part, const float* __restrict__ bpart,
                                                          float* __restrict__ dW, float* __restrict__ db, int groups, int CW,
                                                          int C) {
    const int idx = (blockIdx.x * blockDim.x + threadIdx.x) >> 3;
    const int sub = threadIdx.x & 7;
    float s = 0.f;
    if (idx < CW + C) {
        const float* src = idx < CW ? part + idx : bpart + (idx - CW);
        const int stride = idx < CW ? CW : C;
        float v[HB_GROUPS / 8];
#pragma unroll
        for (int e = 0; e < HB_GROUPS / 8; ++e) {
            const int gidx = sub + 8 * e;
            v[e] = gidx < groups ? src[(int64_t)gidx * stride] : 0.f;
        }
#pragma unroll
        for (int e = 0; e < HB_GROUPS / 8; ++e) s += v[e];
    }
    s += __shfl_xor(s, 1, 64);
    s += __shfl_xor(s, 2, 64);
    s += __shfl_xor(s, 4, 64);
    if (sub == 0 && idx < CW + C) {
        if (idx < CW) dW[idx] = s;
        else db[idx - CW] = s;
    }
}

}  // namespace

static bool bad_split(int Wd, int ld, int split) {
    if (split == 0) return ld < Wd || (ld & 3);
    return split < 4 || (split & 3) || Wd % split || ld < split || (ld & 3);
}

extern "C" int mmdfn_head_fwd(const float* F, const float* mask, const float* W, const float* bias, float* logp, int64_t N,
                              int Wd, int C, int ldf, int split, float mscale, void* stream) {
    if (N <= 0 || Wd < 4 || (Wd & 3) || C < 1 || C > HC_MAX || bad_split(Wd, ldf, split) || (int64_t)C * Wd * 4 > 150 * 1024) return -1;
    int64_t grid = (N + 3) / 4;
    if (grid > 1024) grid = 1024;
    if (int e = mmdfn_allow_big_lds(head_fwd_kernel)) return e;
    hipLaunchKernelGGL(head_fwd_kernel, dim3((unsigned)grid), dim3(256), (size_t)C * Wd * sizeof(float), (hipStream_t)stream, F, mask,
                       W, bias, logp, N, Wd, C, ldf, split, mscale);
    MMDFN_CHECK_LAUNCH();
    return 0;
}

extern "C" int64_t mmdfn_head_bwd_workspace(int Wd, int C) { return (int64_t)HB_GROUPS * ((int64_t)C * Wd + C); }

static int head_bwd_impl(const float* dlogp, const float* logp, const float* F, const float* mask, const float* W, float* dF,
                         float* dW, float* db, float* workspace, int64_t N, int Wd, int C, int ldf, int lddf, int split,
                         float mscale, bool reduce, void* stream) {
    if (N <= 0 || Wd < 4 || (Wd & 3) || C < 1 || C > HC_MAX || bad_split(Wd, ldf, split) || bad_split(Wd, lddf, split)) return -1;
    hipStream_t s = (hipStream_t)stream;
    float* part = workspace;
    float* bpart = workspace + (int64_t)HB_GROUPS * C * Wd;
    const int nblk = (Wd + HB_COLS - 1) / HB_COLS;
    const size_t lds = (size_t)4 * C * HB_COLS * sizeof(float);       // wave partials (>= the C x HB_COLS weight block)
    if (lds > 150 * 1024) return -1;
    if (int e = mmdfn_allow_big_lds(head_bwd_kernel)) return e;
    hipLaunchKernelGGL(head_bwd_kernel, dim3(HB_GROUPS, nblk), dim3(256), lds, s, dlogp, logp, F, mask, W, dF, part, bpart, N, Wd, C,
                       ldf, lddf, split, mscale);
    MMDFN_CHECK_LAUNCH();
    if (!reduce) return 0;
    const int total = (C * Wd + C) * 8;               // 8 lanes per output element
    hipLaunchKernelGGL(head_reduce_kernel, dim3((total + 255) / 256), dim3(256), 0, s, part, bpart, dW, db, HB_GROUPS, C * Wd, C);
    MMDFN_CHECK_LAUNCH();
    return 0;
}

extern "C" int mmdfn_head_bwd(const float* dlogp, const float* logp, const float* F, const float* mask, const float* W, float* dF,
                              float* dW, float* db, float* workspace, int64_t N, int Wd, int C, int ldf, int lddf, int split,
                              float mscale, void* stream) {
    return head_bwd_impl(dlogp, logp, F, mask, W, dF, dW, db, workspace, N, Wd, C, ldf, lddf, split, mscale, true, stream);
}

// The same launch without the slab reduction: workspace then holds mmdfn_head_bwd_groups() slabs of dW ([groups][C][Wd]) followed
// by as many of db ([groups][C]) for mmdfn_gemm_tn_batch_ext to sum with the step's other weight gradients.
extern "C" int mmdfn_head_bwd_groups(void) { return HB_GROUPS; }
extern "C" int mmdfn_head_bwd_partial(const float* dlogp, const float* logp, const float* F, const float* mask, const float* W,
                                      float* dF, float* workspace, int64_t N, int Wd, int C, int ldf, int lddf, int split,
                                      float mscale, void* stream) {
    return head_bwd_impl(dlogp, logp, F, mask, W, dF, nullptr, nullptr, workspace, N, Wd, C, ldf, lddf, split, mscale, false, stream);
}
