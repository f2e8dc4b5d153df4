// K6: scatter-propagate  out = A_hat . H  over the block-tile dialogue graph.
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

namespace {

constexpr int BK = 32;

template <int NW, int NCT, bool TRANS>
__global__ __launch_bounds__(64 * NW) void propagate_kernel(
    const float* __restrict__ tiles, const float* __restrict__ cross, const float* __restrict__ H,
    float* __restrict__ out, const int32_t* __restrict__ dia_len, const int32_t* __restrict__ row_start,
    const int64_t* __restrict__ tile_base, int M, int N, int d, int max_rb) {
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
    const float* Hm = H + ((int64_t)m * N + rs) * d;

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
            if (k < L && c < d) v = *reinterpret_cast<const float4*>(Hm + (int64_t)k * d + c);
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
            for (int e = 0; e < nc; ++e) v += cw[e] * H[((int64_t)cn[e] * N + grow) * d + col];
            out[((int64_t)m * N + grow) * d + col] = v;
        }
    }
}

template <int NW, int NCT>
int launch(const float* tiles, const float* cross, const float* H, float* out, const int32_t* dia_len,
           const int32_t* row_start, const int64_t* tile_base, int B, int M, int N, int d, int max_len,
           int transpose, hipStream_t s) {
    const int BM = 16 * NW;
    const int max_rb = (max_len + BM - 1) / BM;
    dim3 grid(B * max_rb, M, (d + 16 * NCT - 1) / (16 * NCT));
    dim3 block(64 * NW);
    if (transpose)
        hipLaunchKernelGGL((propagate_kernel<NW, NCT, true>), grid, block, 0, s, tiles, cross, H, out, dia_len,
                           row_start, tile_base, M, N, d, max_rb);
    else
        hipLaunchKernelGGL((propagate_kernel<NW, NCT, false>), grid, block, 0, s, tiles, cross, H, out, dia_len,
                           row_start, tile_base, M, N, d, max_rb);
    MMDFN_CHECK_LAUNCH();
    return 0;
}

}  // namespace

int mmdfn_launch_propagate(const float* tiles, const float* cross, const float* H, float* out,
                           const int32_t* dia_len, const int32_t* row_start, const int64_t* tile_base,
                           int B, int M, int N, int d, int max_len, int transpose, hipStream_t s) {
    if (B <= 0 || M <= 0 || M > 9 || N <= 0 || d <= 0 || (d & 3) || max_len <= 0) return -1;
    // rows per workgroup: 64 for long dialogues, 32 for short ones (less padding waste)
    const bool small_rows = max_len <= 48;
    if (d <= 112) {
        return small_rows ? launch<2, 7>(tiles, cross, H, out, dia_len, row_start, tile_base, B, M, N, d, max_len, transpose, s)
                          : launch<4, 7>(tiles, cross, H, out, dia_len, row_start, tile_base, B, M, N, d, max_len, transpose, s);
    } else if (d <= 208) {
        return small_rows ? launch<2, 13>(tiles, cross, H, out, dia_len, row_start, tile_base, B, M, N, d, max_len, transpose, s)
                          : launch<4, 13>(tiles, cross, H, out, dia_len, row_start, tile_base, B, M, N, d, max_len, transpose, s);
    }
    return small_rows ? launch<2, 8>(tiles, cross, H, out, dia_len, row_start, tile_base, B, M, N, d, max_len, transpose, s)
                      : launch<4, 8>(tiles, cross, H, out, dia_len, row_start, tile_base, B, M, N, d, max_len, transpose, s);
}

extern "C" int mmdfn_abi_version(void) { return 1; }

extern "C" int mmdfn_propagate(const float* tiles, const float* cross, const float* H, float* out,
                               const int32_t* dia_len, const int32_t* row_start, const int64_t* tile_base,
                               int B, int M, int N, int d, int max_len, int transpose, void* stream) {
    return mmdfn_launch_propagate(tiles, cross, H, out, dia_len, row_start, tile_base, B, M, N, d, max_len,
                                  transpose, (hipStream_t)stream);
}
