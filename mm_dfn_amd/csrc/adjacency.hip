// K5: multimodal dialogue-graph adjacency, forward and backward.
//
// Replaces MM_GCN.create_big_adj (reference model_mm.py:122-180), which fills
// a dense (MN x MN) fp32 matrix with per-dialogue Python slice-assigns and
// normalises it with two dense (MN)^3 products.  Here the matrix only exists
// in the block-tile layout of include/mmdfn_hip.h:
//
//   forward : unit_cross  -> x/||x||, cross-modal cosines, degree seed
//             tile_dot<1> -> cosine Gram, angular similarity, row degrees   (tile_dot.hip)
//             rdeg_cross  -> degree^-1/2, normalised cross diagonals
//             scale_tiles -> T[p,q] = S[p,q] * r_p * r_q
//   backward: symmetrize  -> W = dT + dT^T
//             bwd_rowsum  -> d(degree)
//             bwd_etile   -> E = (W r_p r_q + dd_p + dd_q) * sim'(G)   (+ the cross diagonals, same launch)
//             propagate   -> d(unit) = E . unit                         (propagate.hip)
//             unit_bwd    -> dX = (du - u (u.du)) / ||x||
#include "mmdfn_internal.h"
#include "../../include/mmdfn_hip.h"

namespace {

constexpr int MAXM = 9;

// ---- one wave per utterance row: unit vectors for every modality, cross cosines
__global__ __launch_bounds__(256) void unit_cross_kernel(const float* __restrict__ feats, float* __restrict__ unit,
                                                         float* __restrict__ norm, float* __restrict__ cdot,
                                                         float* __restrict__ cross_raw, float* __restrict__ deg,
                                                         int M, int N, int D, float modal_weight) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= N) return;
    float dsum[MAXM];
    for (int m = 0; m < M; ++m) {
        dsum[m] = 0.f;
        const float* x = feats + ((int64_t)m * N + row) * D;
        float* u = unit + ((int64_t)m * N + row) * D;
        float ss = 0.f;
        for (int k = lane; k < D; k += 64) ss += x[k] * x[k];
        ss = wave_sum(ss);
        const float nv = sqrtf(ss);
        for (int k = lane; k < D; k += 64) u[k] = x[k] / nv;
        if (lane == 0) norm[(int64_t)m * N + row] = nv;
    }
    // each lane re-reads only the elements it wrote itself
    for (int m = 0; m < M; ++m)
        for (int n = m + 1; n < M; ++n) {
            const float* um = unit + ((int64_t)m * N + row) * D;
            const float* un = unit + ((int64_t)n * N + row) * D;
            float s = 0.f;
            for (int k = lane; k < D; k += 64) s += um[k] * un[k];
            s = wave_sum(s);
            const float c = mmdfn_sim(s) * modal_weight;
            dsum[m] += c;
            dsum[n] += c;
            if (lane == 0) {
                const int64_t o = (int64_t)mmdfn_pair_index(m, n, M) * N + row;
                cdot[o] = s;
                cross_raw[o] = c;
            }
        }
    if (lane == 0)
        for (int m = 0; m < M; ++m) deg[(int64_t)m * N + row] = dsum[m];
}

// Same stage with every modality's row held in registers (M <= MMAX, D <= 64 * KSL): ONE round of loads per wave
// instead of a load -> reduce -> store -> reload chain per modality and per pair (the chain costs a memory round trip per
// link: 12 us at cfg2 against 5 us for this form).  Same arithmetic in the same order as the kernel above.
template <int MMAX, int KSL>
__global__ __launch_bounds__(256) void unit_cross_reg_kernel(const float* __restrict__ feats, float* __restrict__ unit,
                                                             float* __restrict__ norm, float* __restrict__ cdot,
                                                             float* __restrict__ cross_raw, float* __restrict__ deg,
                                                             int M, int N, int D, float modal_weight) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= N) return;
    float x[MMAX][KSL];
#pragma unroll
    for (int m = 0; m < MMAX; ++m)
#pragma unroll
        for (int sidx = 0; sidx < KSL; ++sidx) {
            const int k = lane + 64 * sidx;
            x[m][sidx] = (m < M && k < D) ? feats[((int64_t)m * N + row) * D + k] : 0.f;
        }
    float dsum[MMAX];
#pragma unroll
    for (int m = 0; m < MMAX; ++m) {
        dsum[m] = 0.f;
        if (m >= M) continue;
        float ss = 0.f;
#pragma unroll
        for (int sidx = 0; sidx < KSL; ++sidx) ss += x[m][sidx] * x[m][sidx];
        ss = wave_sum(ss);
        const float nv = sqrtf(ss);
#pragma unroll
        for (int sidx = 0; sidx < KSL; ++sidx) {
            const int k = lane + 64 * sidx;
            x[m][sidx] = x[m][sidx] / nv;
            if (k < D) unit[((int64_t)m * N + row) * D + k] = x[m][sidx];
        }
        if (lane == 0) norm[(int64_t)m * N + row] = nv;
    }
#pragma unroll
    for (int m = 0; m < MMAX; ++m)
#pragma unroll
        for (int n = m + 1; n < MMAX; ++n) {
            if (n >= M) continue;
            float sdot = 0.f;
#pragma unroll
            for (int sidx = 0; sidx < KSL; ++sidx) sdot += x[m][sidx] * x[n][sidx];
            sdot = wave_sum(sdot);
            const float c = mmdfn_sim(sdot) * modal_weight;
            dsum[m] += c;
            dsum[n] += c;
            if (lane == 0) {
                const int64_t o = (int64_t)mmdfn_pair_index(m, n, M) * N + row;
                cdot[o] = sdot;
                cross_raw[o] = c;
            }
        }
    if (lane == 0) {
#pragma unroll
        for (int m = 0; m < MMAX; ++m)
            if (m < M) deg[(int64_t)m * N + row] = dsum[m];
    }
}

// rdeg = deg^-1/2 (in place), cross[k][r] *= rdeg[m][r] * rdeg[n][r]
__global__ void rdeg_cross_kernel(float* __restrict__ rdeg, float* __restrict__ cross, int M, int N) {
    const int row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= N) return;
    float r[MAXM];
    for (int m = 0; m < M; ++m) {
        r[m] = powf(rdeg[(int64_t)m * N + row], -0.5f);
        rdeg[(int64_t)m * N + row] = r[m];
    }
    for (int m = 0; m < M; ++m)
        for (int n = m + 1; n < M; ++n) {
            const int64_t o = (int64_t)mmdfn_pair_index(m, n, M) * N + row;
            cross[o] = (r[m] * cross[o]) * r[n];
        }
}

// T[p,q] = (r_p * S[p,q]) * r_q  -- one wave per tile row
__global__ __launch_bounds__(256) void scale_tiles_kernel(float* __restrict__ tiles, const float* __restrict__ rdeg,
                                                          const int32_t* __restrict__ dia_len,
                                                          const int32_t* __restrict__ row_start,
                                                          const int64_t* __restrict__ tile_base, int N, int max_len) {
    const int i = blockIdx.x / ((max_len + 3) / 4);
    const int p = (blockIdx.x % ((max_len + 3) / 4)) * 4 + (threadIdx.x >> 6);
    const int m = blockIdx.y;
    const int L = dia_len[i];
    if (p >= L) return;
    const int lane = threadIdx.x & 63;
    const int ld = (L + 3) & ~3;
    const int rs = row_start[i];
    float* t = tiles + tile_base[i] + (int64_t)m * L * ld + (int64_t)p * ld;
    const float* r = rdeg + (int64_t)m * N + rs;
    const float rp = r[p];
    for (int q = lane; q < L; q += 64) t[q] = (rp * t[q]) * r[q];
}

// W = dT + dT^T on every tile, 32x32 blocks through LDS
__global__ __launch_bounds__(256) void symmetrize_kernel(const float* __restrict__ dT, float* __restrict__ W,
                                                         const int32_t* __restrict__ dia_len,
                                                         const int64_t* __restrict__ tile_base, int nb) {
    __shared__ float tr[32][33];
    const int i = blockIdx.x / (nb * nb);
    const int bb = blockIdx.x % (nb * nb);
    const int bp = bb / nb, bq = bb % nb;
    const int m = blockIdx.y;
    const int L = dia_len[i];
    if (bp * 32 >= L || bq * 32 >= L) return;
    const int ld = (L + 3) & ~3;
    const int64_t toff = tile_base[i] + (int64_t)m * L * ld;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    for (int rr = ty; rr < 32; rr += 8) {
        const int q = bq * 32 + rr, p = bp * 32 + tx;  // read dT[q][p] (row q, col p)
        tr[rr][tx] = (q < L && p < L) ? dT[toff + (int64_t)q * ld + p] : 0.f;
    }
    __syncthreads();
    for (int rr = ty; rr < 32; rr += 8) {
        const int p = bp * 32 + rr, q = bq * 32 + tx;
        if (p < L && q < ld) {
            const float v = (q < L) ? dT[toff + (int64_t)p * ld + q] + tr[tx][rr] : 0.f;
            W[toff + (int64_t)p * ld + q] = v;
        }
    }
}

// ddeg[m][row] = -1/2 r^3 * ( sum_q W[p,q] S[p,q] r_q + cross terms )   -- one wave per tile row
__global__ __launch_bounds__(256) void bwd_rowsum_kernel(const float* __restrict__ W, const float* __restrict__ cosg,
                                                         const float* __restrict__ rdeg,
                                                         const float* __restrict__ dcross,
                                                         const float* __restrict__ cdot, float* __restrict__ ddeg,
                                                         const int32_t* __restrict__ dia_len,
                                                         const int32_t* __restrict__ row_start,
                                                         const int64_t* __restrict__ tile_base, int M, int N,
                                                         int max_len, float modal_weight) {
    const int i = blockIdx.x / ((max_len + 3) / 4);
    const int p = (blockIdx.x % ((max_len + 3) / 4)) * 4 + (threadIdx.x >> 6);
    const int m = blockIdx.y;
    const int L = dia_len[i];
    if (p >= L) return;
    const int lane = threadIdx.x & 63;
    const int ld = (L + 3) & ~3;
    const int rs = row_start[i];
    const int64_t off = tile_base[i] + (int64_t)m * L * ld + (int64_t)p * ld;
    const float* r = rdeg + (int64_t)m * N + rs;
    float s = 0.f;
    for (int q = lane; q < L; q += 64) s += W[off + q] * mmdfn_sim(cosg[off + q]) * r[q];
    s = wave_sum(s);
    if (lane == 0) {
        const int64_t grow = rs + p;
        for (int n = 0; n < M; ++n) {
            if (n == m) continue;
            const int pk = (m < n) ? mmdfn_pair_index(m, n, M) : mmdfn_pair_index(n, m, M);
            const float c = mmdfn_sim(cdot[(int64_t)pk * N + grow]) * modal_weight;
            s += dcross[(int64_t)pk * N + grow] * c * rdeg[(int64_t)n * N + grow];
        }
        const float rp = r[p];
        ddeg[(int64_t)m * N + grow] = -0.5f * rp * rp * rp * s;
    }
}

// E[p,q] = (W[p,q] r_p r_q + dd_p + dd_q) * sim'(G[p,q])   -- one wave per tile row; the blocks behind the tile rows
// (blockIdx.x >= tile_blocks, y = 0) do the cross diagonals: ecross[k][r] = (dcross r_m r_n + dd_m + dd_n) * w * sim'(cdot)
__global__ __launch_bounds__(256) void bwd_etile_kernel(const float* __restrict__ W, const float* __restrict__ cosg,
                                                        const float* __restrict__ rdeg, const float* __restrict__ ddeg,
                                                        float* __restrict__ E, const float* __restrict__ dcross,
                                                        const float* __restrict__ cdot, float* __restrict__ ecross,
                                                        const int32_t* __restrict__ dia_len,
                                                        const int32_t* __restrict__ row_start,
                                                        const int64_t* __restrict__ tile_base, int M, int N, int max_len,
                                                        int tile_blocks, float modal_weight) {
    if ((int)blockIdx.x >= tile_blocks) {
        const int row = ((int)blockIdx.x - tile_blocks) * 256 + threadIdx.x;
        if (blockIdx.y != 0 || row >= N) return;
        for (int m = 0; m < M; ++m)
            for (int n = m + 1; n < M; ++n) {
                const int64_t o = (int64_t)mmdfn_pair_index(m, n, M) * N + row;
                const float rm = rdeg[(int64_t)m * N + row], rn = rdeg[(int64_t)n * N + row];
                ecross[o] = (dcross[o] * rm * rn + ddeg[(int64_t)m * N + row] + ddeg[(int64_t)n * N + row]) *
                            modal_weight * mmdfn_dsim(cdot[o]);
            }
        return;
    }
    const int i = blockIdx.x / ((max_len + 3) / 4);
    const int p = (blockIdx.x % ((max_len + 3) / 4)) * 4 + (threadIdx.x >> 6);
    const int m = blockIdx.y;
    const int L = dia_len[i];
    if (p >= L) return;
    const int lane = threadIdx.x & 63;
    const int ld = (L + 3) & ~3;
    const int rs = row_start[i];
    const int64_t off = tile_base[i] + (int64_t)m * L * ld + (int64_t)p * ld;
    const float* r = rdeg + (int64_t)m * N + rs;
    const float* dd = ddeg + (int64_t)m * N + rs;
    const float rp = r[p], ddp = dd[p];
    for (int q = lane; q < ld; q += 64) {
        float e = 0.f;
        if (q < L) e = (W[off + q] * rp * r[q] + ddp + dd[q]) * mmdfn_dsim(cosg[off + q]);
        E[off + q] = e;
    }
}

// dX = (du - u (u.du)) / ||x||  (+ addend)   -- one wave per (m, row)
__global__ __launch_bounds__(256) void unit_bwd_kernel(const float* __restrict__ unit, const float* __restrict__ norm,
                                                       const float* __restrict__ dunit, const float* __restrict__ addend,
                                                       float* __restrict__ dfeats, int64_t rows, int D) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* u = unit + row * D;
    const float* du = dunit + row * D;
    float s = 0.f;
    for (int k = lane; k < D; k += 64) s += u[k] * du[k];
    s = wave_sum(s);
    const float inv = 1.0f / norm[row];
    // addend: the gradient that reaches the same features on their other path (they are also the input of the GCN
    // stack); added here instead of by an autograd accumulation launch
    if (addend)
        for (int k = lane; k < D; k += 64) dfeats[row * D + k] = (du[k] - u[k] * s) * inv + addend[row * D + k];
    else
        for (int k = lane; k < D; k += 64) dfeats[row * D + k] = (du[k] - u[k] * s) * inv;
}

}  // namespace

extern "C" int mmdfn_adj_build(const float* feats, float* unit, float* norm, float* cosg, float* cdot, float* rdeg,
                               float* tiles, float* cross, const int32_t* dia_len, const int32_t* row_start,
                               const int64_t* tile_base, int B, int M, int N, int D, int max_len, float modal_weight,
                               void* stream) {
    if (B <= 0 || M <= 0 || M > MAXM || N <= 0 || D <= 0 || (D & 3) || max_len <= 0) return -1;
    hipStream_t s = (hipStream_t)stream;
    {
        // short dialogues: one workgroup per (dialogue, modality), one launch + the cross diagonals (adjacency_small.hip)
        const int rc = mmdfn_launch_adj_small_fwd(feats, unit, norm, cosg, cdot, rdeg, tiles, cross, dia_len, row_start,
                                                  tile_base, B, M, N, D, max_len, modal_weight, s);
        if (rc != -2) return rc;
    }
#define UNIT_CROSS(KERN) hipLaunchKernelGGL(KERN, dim3((N + 3) / 4), dim3(256), 0, s, feats, unit, norm, cdot, cross, rdeg, M, N, D, modal_weight)
    if (M <= 3 && D <= 256) UNIT_CROSS((unit_cross_reg_kernel<3, 4>));
    else if (M <= 6 && D <= 256) UNIT_CROSS((unit_cross_reg_kernel<6, 4>));
    else UNIT_CROSS(unit_cross_kernel);
#undef UNIT_CROSS
    MMDFN_CHECK_LAUNCH();
    int rc = mmdfn_launch_tile_dot(unit, unit, tiles, cosg, rdeg, dia_len, row_start, tile_base, B, M, N, D, D, D,
                                   max_len, 1, 0, s);
    if (rc) return rc;
    hipLaunchKernelGGL(rdeg_cross_kernel, dim3((N + 255) / 256), dim3(256), 0, s, rdeg, cross, M, N);
    MMDFN_CHECK_LAUNCH();
    hipLaunchKernelGGL(scale_tiles_kernel, dim3(B * ((max_len + 3) / 4), M), dim3(256), 0, s, tiles, rdeg, dia_len,
                       row_start, tile_base, N, max_len);
    MMDFN_CHECK_LAUNCH();
    return 0;
}

extern "C" int mmdfn_adj_build_bwd(const float* dtiles, const float* dcross, const float* unit, const float* norm,
                                   const float* cosg, const float* cdot, const float* rdeg, const float* tiles,
                                   const float* cross, float* wsym, float* etile, float* ecross, float* ddeg,
                                   float* dunit, float* dfeats, const float* addend, const int32_t* dia_len,
                                   const int32_t* row_start, const int64_t* tile_base, int B, int M, int N, int D,
                                   int max_len, float modal_weight, void* stream) {
    if (B <= 0 || M <= 0 || M > MAXM || N <= 0 || D <= 0 || (D & 3) || max_len <= 0) return -1;
    hipStream_t s = (hipStream_t)stream;
    {
        // (the form the forward pass of these tensors took: the choice depends on the shape only)
        const int rc = mmdfn_launch_adj_small_bwd(dtiles, dcross, unit, norm, cosg, cdot, rdeg, tiles, cross, addend, dfeats,
                                                  dia_len, row_start, tile_base, B, M, N, D, max_len, modal_weight, s);
        if (rc != -2) return rc;
    }
    const int nb = (max_len + 31) / 32;
    const int rowblocks = (max_len + 3) / 4;
    hipLaunchKernelGGL(symmetrize_kernel, dim3(B * nb * nb, M), dim3(256), 0, s, dtiles, wsym, dia_len, tile_base, nb);
    MMDFN_CHECK_LAUNCH();
    hipLaunchKernelGGL(bwd_rowsum_kernel, dim3(B * rowblocks, M), dim3(256), 0, s, wsym, cosg, rdeg, dcross, cdot,
                       ddeg, dia_len, row_start, tile_base, M, N, max_len, modal_weight);
    MMDFN_CHECK_LAUNCH();
    hipLaunchKernelGGL(bwd_etile_kernel, dim3(B * rowblocks + (N + 255) / 256, M), dim3(256), 0, s, wsym, cosg, rdeg, ddeg,
                       etile, dcross, cdot, ecross, dia_len, row_start, tile_base, M, N, max_len, B * rowblocks,
                       modal_weight);
    MMDFN_CHECK_LAUNCH();
    int rc = mmdfn_launch_propagate(etile, ecross, unit, dunit, dia_len, row_start, tile_base, B, M, N, D, D, D, max_len,
                                    0, s);
    if (rc) return rc;
    hipLaunchKernelGGL(unit_bwd_kernel, dim3((unsigned)(((int64_t)M * N + 3) / 4)), dim3(256), 0, s, unit, norm,
                       dunit, addend, dfeats, (int64_t)M * N, D);
    MMDFN_CHECK_LAUNCH();
    return 0;
}
