// K6 + K7 forward of one GCN layer for SHORT dialogues (L <= 128, M <= 3, H <= 100) in ONE launch, one workgroup per (dialogue,
// modality, strip of 32 rows):
//
//   hi  = A_hat . z                                              (GraphConvolution's torch.spmm(adj, input), model_GCN.py:178)
//   pre = theta [hi | h0] W + (1 - theta)((1 - alpha) hi + alpha h0),  out = relu(pre) (.) m ms + q     (model_GCN.py:180-189)
//
// propagate.hip + gcn_stack.hip run this as two launches (9 + 14 us at cfg2's 5 280 rows, work for ~2 us); the layer update is
// row-local, so a strip that has just produced its rows of hi can go on with them: hi never leaves the compute unit between the
// two products (it is still written out once: the backward pass contracts against it).  Both products run on exact-f32 MFMAs with
// the A side in LDS (the adjacency strip; then [hi | h0]) and the B side in REGISTERS: a wave owns one 16-column tile and requests
// its fragments of z (28-32 values per lane) and of W (2H / 4 k-steps) straight from L2 before anything else -- like every other
// operand of the launch (a workgroup is one serial chain: adjacency_small.hip).
#include "mmdfn_internal.h"
#include "../../include/mmdfn_hip.h"
#include <stdlib.h>

namespace {

constexpr int GS_SR = 32;                   // strip rows
constexpr int GS_NW = 8;                    // waves
constexpr int GS_MAXL = 128;
constexpr int GS_SA = 132;                  // row stride (floats) of the adjacency strip in LDS (33 quads: conflict-free b128 reads)

#ifdef MMDFN_TUNING
#define GS_STOP(K) do { if (stop == (K)) return; } while (0)
#else
#define GS_STOP(K) do { } while (0)
#endif

// LDS: As[32][GS_SA] | A2[32][S2] ([hi | h0 | zero pad]) | cw[2][32]
__global__ __launch_bounds__(64 * GS_NW) void prop_layer_strip_fwd_kernel(
    const float* __restrict__ tiles, const float* __restrict__ cross, const float* __restrict__ zin, int ldz,
    const int32_t* __restrict__ dia_len, const int32_t* __restrict__ row_start, const int64_t* __restrict__ tile_base,
    int B, int M, int N, int NS, const float* __restrict__ h0, const float* __restrict__ W, const float* __restrict__ q,
    const float* __restrict__ mk, float* __restrict__ hi_out, float* __restrict__ out, float* __restrict__ gmask,
    float theta, float alpha, int H, int ldo, float ms, int S2, int stop) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int bid = blockIdx.x;
    const int yq = bid >> 3;
    const int per = M * NS;
    const int i = (yq / per) * 8 + (bid & 7);
    if (i >= B) return;
    const int rem = yq % per;
    const int m = rem / NS;
    const int st = rem - m * NS;
    const int L = dia_len[i];
    const int r0 = st * GS_SR;
    if (r0 >= L) return;
    GS_STOP(9);
    const int ld = (L + 3) & ~3;
    const int Lp = (L + 15) & ~15;
    const int nt = Lp >> 4;
    const int rs = row_start[i];
    const int64_t toff = tile_base[i] + (int64_t)m * L * ld;
    const int K2 = 2 * H;
    const int nkc2 = (K2 + 15) >> 4;                     // 16-wide k groups of the second product (<= 16)
    const int nct = (H + 15) >> 4;                       // 16-column tiles (<= 8: one per wave)

    float* As = smem;
    float* A2 = As + GS_SR * GS_SA;
    float* cw = A2 + GS_SR * S2;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fi = lane & 15, g = lane >> 4;
    const bool tile = w < nct;
    const int col = 16 * w + fi;
    const unsigned colc = col < H ? col : H - 1;         // (columns past H: clamped, never stored)

    int o0 = -1, o1 = -1;                                // the other modalities
    for (int n = 0; n < M; ++n)
        if (n != m) { if (o0 < 0) o0 = n; else o1 = n; }

    // ---- every request of the launch first, in the order of use (requests return in order: the LDS fill waits for the first few)
    // the adjacency strip (a half-wave = one row of 128 columns) and the strip's h0 rows, for the LDS
    const int c4 = lane & 31, sub = lane >> 5;
    float4 ar[2];
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int p = r0 + 16 * it + 2 * w + sub;
        const int qc = 4 * c4 < ld ? 4 * c4 : ld - 4;
        ar[it] = *reinterpret_cast<const float4*>(tiles + toff + (unsigned)((p < L ? p : L - 1) * ld + qc));
    }
    const int H4 = H >> 2;
    float4 hr[2];
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
        const int e = tid + 64 * GS_NW * s2;             // (32 rows x H / 4 quads <= 1 024)
        const int r = e / H4, k = (e - r * H4) * 4;
        const int p = r0 + r;
        hr[s2] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (r < GS_SR && p < L) hr[s2] = *reinterpret_cast<const float4*>(h0 + ((int64_t)m * N + rs + p) * H + k);
    }
    float cwv = 0.f;                                     // cross-modal weights of the strip rows (thread = (other modality, row))
    if (tid < 2 * GS_SR) {
        const int n = tid >> 5, pl = tid & 31;
        const int o = n == 0 ? o0 : o1;
        const int p = r0 + pl;
        if (o >= 0 && p < L) {
            const int pk = (m < o) ? mmdfn_pair_index(m, o, M) : mmdfn_pair_index(o, m, M);
            cwv = cross[(int64_t)pk * N + rs + p];
        }
    }
    // C layout of this wave's tile: element (a, r) = row r0 + 16 a + 4 g + r, column col.  Addresses = a UNIFORM row base (scalar
    // arithmetic) + one lane offset per array (computed once); a strip that ends inside the dialogue clamps its rows instead.
    float zo0[2][4], zo1[2][4], qv[2][4], mv[2][4];
    const bool whole = r0 + GS_SR <= L;                   // (uniform) every row of the strip exists
    const int64_t row0 = (int64_t)m * N + rs + r0;        // first row of the strip in the (M N) row space of this modality
    const unsigned lane_z = (unsigned)(4 * g) * (unsigned)ldz + colc, lane_h = (unsigned)(4 * g) * (unsigned)H + colc;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            zo0[a][r] = zo1[a][r] = qv[a][r] = 0.f;
            mv[a][r] = 1.0f;
        }
    if (tile) {
        const float* z0 = zin + ((int64_t)(o0 >= 0 ? o0 : m) * N + rs + r0) * ldz;
        const float* z1 = zin + ((int64_t)(o1 >= 0 ? o1 : m) * N + rs + r0) * ldz;
        const float* qb = (q ? q : h0) + row0 * H;        // (absent operands read h0 and are not used: plain global loads)
        const float* mb = (mk ? mk : h0) + row0 * H;
        if (whole) {
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int rr = 16 * a + r;
                    zo0[a][r] = (z0 + (size_t)rr * ldz)[lane_z];
                    zo1[a][r] = (z1 + (size_t)rr * ldz)[lane_z];
                    qv[a][r] = (qb + (size_t)rr * H)[lane_h];
                    mv[a][r] = (mb + (size_t)rr * H)[lane_h];
                }
        } else {
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int pl = 16 * a + 4 * g + r;
                    const unsigned pc = r0 + pl < L ? pl : L - 1 - r0;
                    zo0[a][r] = z0[pc * (unsigned)ldz + colc];
                    zo1[a][r] = z1[pc * (unsigned)ldz + colc];
                    qv[a][r] = qb[pc * (unsigned)H + colc];
                    mv[a][r] = mb[pc * (unsigned)H + colc];
                }
        }
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (!q) qv[a][r] = 0.f;
                mv[a][r] = mk ? mv[a][r] * ms : 1.0f;
            }
    }

    // B fragments: MFMA step j of the 16-wide k group kc contracts k = 16 kc + 4 g + j (the A side reads 16-byte units).  Rows
    // past L / k past 2H are clamped (they meet zeros of the A side); only the LAST k group of either product can hold such
    // rows: all the others are requested at a uniform row base + the lane offset, no per-request vector arithmetic.
    float zfr[32], wfr[64];
#pragma unroll
    for (int kk = 0; kk < 32; ++kk) zfr[kk] = 0.f;
#pragma unroll
    for (int kk = 0; kk < 64; ++kk) wfr[kk] = 0.f;
    if (tile) {
        const float* zm = zin + ((int64_t)m * N + rs) * ldz;
        const int zfull = L >> 4;
#pragma unroll
        for (int kc = 0; kc < 8; ++kc) {
            if (kc < zfull) {
#pragma unroll
                for (int j = 0; j < 4; ++j) zfr[4 * kc + j] = (zm + (size_t)(16 * kc + j) * ldz)[lane_z];
            } else if (kc < nt) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int qq = 16 * kc + 4 * g + j;
                    zfr[4 * kc + j] = zm[(unsigned)(qq < L ? qq : L - 1) * (unsigned)ldz + colc];
                }
            }
        }
        const int wfull = K2 >> 4;
#pragma unroll
        for (int kc = 0; kc < 16; ++kc) {
            if (kc < wfull) {
#pragma unroll
                for (int j = 0; j < 4; ++j) wfr[4 * kc + j] = (W + (size_t)(16 * kc + j) * H)[lane_h];
            } else if (kc < nkc2) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int k = 16 * kc + 4 * g + j;
                    wfr[4 * kc + j] = W[(unsigned)(k < K2 ? k : K2 - 1) * (unsigned)H + colc];
                }
            }
        }
    }
    // ---- LDS: adjacency strip (columns past L and rows past L are zero), h0 strip, zero pad of [hi | h0], cross weights
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int pl = 16 * it + 2 * w + sub;
        const int p = r0 + pl;
        const int qq = 4 * c4;
        float4 v = ar[it];
        const bool rowin = p < L;
        v.x = (rowin && qq + 0 < L) ? v.x : 0.f;
        v.y = (rowin && qq + 1 < L) ? v.y : 0.f;
        v.z = (rowin && qq + 2 < L) ? v.z : 0.f;
        v.w = (rowin && qq + 3 < L) ? v.w : 0.f;
        *reinterpret_cast<float4*>(As + pl * GS_SA + qq) = v;
    }
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
        const int e = tid + 64 * GS_NW * s2;
        const int r = e / H4, k = (e - r * H4) * 4;
        if (r < GS_SR) *reinterpret_cast<float4*>(A2 + r * S2 + H + k) = hr[s2];
    }
    for (int e = tid; e < GS_SR * (S2 - K2); e += 64 * GS_NW) {        // (the zero pad behind 2H)
        const int r = e / (S2 - K2), k = K2 + (e - r * (S2 - K2));
        A2[r * S2 + k] = 0.f;
    }
    if (tid < 2 * GS_SR) cw[tid] = cwv;
    __syncthreads();
    GS_STOP(1);

    // ---- hi strip = A strip . z (+ the cross-modal rows); -> global (the backward pass's operand) and -> LDS
    float hv[2][4];
    if (tile) {
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const float* ea = As + (16 * a + fi) * GS_SA + 4 * g;
            float4 av[8];
#pragma unroll
            for (int kc = 0; kc < 8; ++kc) av[kc] = *reinterpret_cast<const float4*>(ea + 16 * kc);
            f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kc = 0; kc < 8; ++kc) {
                if (kc >= nt) continue;
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[kc].x, zfr[4 * kc + 0], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[kc].y, zfr[4 * kc + 1], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[kc].z, zfr[4 * kc + 2], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[kc].w, zfr[4 * kc + 3], acc, 0, 0, 0);
            }
            float* hb = hi_out + (row0 + 16 * a) * H;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int pl = 16 * a + 4 * g + r;
                float v = acc[r];
                if (o0 >= 0) v += cw[pl] * zo0[a][r];
                if (o1 >= 0) v += cw[GS_SR + pl] * zo1[a][r];
                hv[a][r] = v;
                if (col < H) {
                    A2[pl * S2 + col] = v;
                    if (r0 + pl < L) (hb + (size_t)r * H)[lane_h] = v;
                }
            }
        }
    }
    __syncthreads();
    GS_STOP(2);

    // ---- P = [hi | h0] W, then the update, from the accumulators
    if (tile) {
        f32x4 acc0 = (f32x4){0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
        const float* e0 = A2 + fi * S2 + 4 * g;
        const float* e1 = A2 + (16 + fi) * S2 + 4 * g;
#pragma unroll
        for (int kc = 0; kc < 16; ++kc) {
            if (kc >= nkc2) continue;
            const float4 a0 = *reinterpret_cast<const float4*>(e0 + 16 * kc);
            const float4 a1 = *reinterpret_cast<const float4*>(e1 + 16 * kc);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.x, wfr[4 * kc + 0], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.x, wfr[4 * kc + 0], acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.y, wfr[4 * kc + 1], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.y, wfr[4 * kc + 1], acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.z, wfr[4 * kc + 2], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.z, wfr[4 * kc + 2], acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.w, wfr[4 * kc + 3], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.w, wfr[4 * kc + 3], acc1, 0, 0, 0);
        }
        if (col < H) {
            const unsigned lane_o = (unsigned)(4 * g) * (unsigned)ldo + col;
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int pl = 16 * a + 4 * g + r;
                    if (r0 + pl >= L) continue;
                    const float pv = a == 0 ? acc0[r] : acc1[r];
                    const float v0 = A2[pl * S2 + H + col];
                    const float pre = theta * pv + (1.0f - theta) * ((1.0f - alpha) * hv[a][r] + alpha * v0);
                    (out + (row0 + 16 * a + r) * ldo)[lane_o] = fmaxf(pre, 0.f) * mv[a][r] + qv[a][r];
                    (gmask + (row0 + 16 * a + r) * H)[lane_h] = pre > 0.f ? mv[a][r] : 0.f;        // (saved for the backward pass)
                }
        }
    }
}

}  // namespace

// -2: shape not covered (the caller runs mmdfn_propagate + mmdfn_gcnii_layer_fwd)
extern "C" int mmdfn_prop_layer_fwd(const float* tiles, const float* cross, const float* zin, int ldz, const int32_t* dia_len,
                                    const int32_t* row_start, const int64_t* tile_base, int B, int M, int N, int max_len,
                                    const float* h0, const float* W, const float* q, const float* m, float* hi, float* out,
                                    float* gmask, float theta, float alpha, int H, int ldo, float mscale, void* stream) {
    if (B <= 0 || M <= 0 || N <= 0 || H <= 0 || max_len <= 0 || !tiles || !zin || !h0 || !W || !hi || !out || !gmask) return -1;
    if (M > 3 || max_len > GS_MAXL || H > 100 || (H & 3) || ldz < H || ldo < H) return -2;     // (H: the K7 kernels' own limit)
    if (M > 1 && !cross) return -1;
    int stop = 0;
#ifdef MMDFN_TUNING
    if (const char* e = getenv("MMDFN_PROP_LAYER")) if (atoi(e) == 0) return -2;
    if (const char* e = getenv("MMDFN_PROP_LAYER_STOP")) stop = atoi(e);
#endif
    const int NS = (max_len + GS_SR - 1) / GS_SR;
    const int64_t grid = (int64_t)((B + 7) / 8) * 8 * M * NS;
    if (grid > 4096) return -2;                          // many dialogues: the two-launch form fills the chip by itself
    const int S2 = ((2 * H + 15) & ~15) + 4;             // [hi | h0] row stride: a multiple of 16 k + one 16-byte unit (odd quad count)
    const size_t lds = ((size_t)GS_SR * GS_SA + (size_t)GS_SR * S2 + 2 * GS_SR) * sizeof(float);
    hipLaunchKernelGGL(prop_layer_strip_fwd_kernel, dim3((unsigned)grid), dim3(64 * GS_NW), lds, (hipStream_t)stream, tiles, cross,
                       zin, ldz, dia_len, row_start, tile_base, B, M, N, NS, h0, W, q, m, hi, out, gmask, theta, alpha, H, ldo,
                       mscale, S2, stop);
    MMDFN_CHECK_LAUNCH();
    return 0;
}
