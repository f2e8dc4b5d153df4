// Shared declarations of the kernels that multiply against bf16 PIECE PLANES (linear_planes.hip, gcn_planes.hip).
//
// Plane layout of a B operand "B[n][k]" (n < N output columns, k < K contraction):  KS = ceil(K / 16) k-steps, NT = ceil(N / 32)
// column tiles;  planes[((ct * KS + ks) * 3 + piece) * 64 + lane] = 8 bf16 (16 bytes) = piece `piece` of
// B[32 ct + (lane & 31)][16 ks + 8 (lane >> 5) + 0..7], zero outside N x K: exactly the B operand of v_mfma_f32_32x32x16_bf16.
#pragma once
#include "mmdfn_internal.h"

namespace {

typedef __bf16 pl_bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int PL_BM = 64;          // rows per workgroup
constexpr int PL_STG = 2;          // k-steps per phase (32 k): 2 buffers x 2 x 3 pieces x 2 row halves x 1 KB = 24 KB of LDS
constexpr int PL_LDS = 2 * PL_STG * 3 * 2 * 64;     // u32x4 elements of the A-fragment double buffer

__device__ __forceinline__ float pl_as_f(uint32_t u) { return __builtin_bit_cast(float, u); }
__device__ __forceinline__ uint32_t pl_as_u(float f) { return __builtin_bit_cast(uint32_t, f); }
__device__ __forceinline__ f32x16 pl_mfma(u32x4 a, u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(pl_bf16x8, a), __builtin_bit_cast(pl_bf16x8, b), c, 0, 0, 0);
}

// two consecutive fp32 values -> one packed bf16 pair per piece (x = p1 + p2 + p3 exactly, by truncation)
__device__ __forceinline__ void pl_cut2(float a, float b, uint32_t& p1, uint32_t& p2, uint32_t& p3) {
    p1 = __builtin_amdgcn_perm(pl_as_u(b), pl_as_u(a), 0x07060302u);
    a -= pl_as_f(pl_as_u(a) & 0xffff0000u);
    b -= pl_as_f(pl_as_u(b) & 0xffff0000u);
    p2 = __builtin_amdgcn_perm(pl_as_u(b), pl_as_u(a), 0x07060302u);
    a -= pl_as_f(pl_as_u(a) & 0xffff0000u);
    b -= pl_as_f(pl_as_u(b) & 0xffff0000u);
    p3 = __builtin_amdgcn_perm(pl_as_u(b), pl_as_u(a), 0x07060302u);
}
// eight consecutive fp32 values -> three u32x4 of packed bf16 pieces
__device__ __forceinline__ void pl_cut8(const float (&x)[8], u32x4& p1, u32x4& p2, u32x4& p3) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        uint32_t q1, q2, q3;
        pl_cut2(x[2 * j], x[2 * j + 1], q1, q2, q3);
        p1[j] = q1;
        p2[j] = q2;
        p3[j] = q3;
    }
}

// workgroup decode shared by the launches: grid = 8 ceil(row blocks / 8) ncb; blockIdx % 8 (the XCD) owns the row blocks = its
// number (mod 8) and runs the ncb column blocks of a row block back to back, so the A rows of a row block are fetched into ONE L2
__device__ __forceinline__ bool pl_decode(int nrb, int ncb, int& rb, int& cb, int bid) {
    const int yq = bid >> 3;
    rb = (yq / ncb) * 8 + (bid & 7);
    cb = yq % ncb;
    return rb < nrb;
}
inline int64_t pl_grid(int nrb, int ncb) { return (int64_t)((nrb + 7) / 8) * 8 * ncb; }

}  // namespace
