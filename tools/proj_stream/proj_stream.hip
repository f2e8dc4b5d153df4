// EXPERIMENT (round 5, outside the product library): many-row projection  y = x W^T + b  with the weight cut ONCE per step.
//
// The projections of the encoder side (model.py:1065,1082,1094,1129,1132) run at a third of the exact-fp32 matrix rate on both
// hand-written forms (tools/bench_proj_forms.py: 7 040 x 200 -> 600 in 33-34 us): the few-row kernel pays operand DMA and a
// fixed part per 64 x 64 tile, the many-row bf16-piece kernel re-cuts W in every workgroup.  Weights are constant inside a
// training step, so here
//   ps_cut_weight  cuts W (N, K) into three bf16 piece planes stored in MFMA FRAGMENT ORDER: block (tile t, K-step ks, piece q) =
//                  1 KB, lane l holds W[32 t + (l & 31)][16 ks + 8 (l >> 5) .. + 7]  (one launch per weight and step);
//   ps_project     workgroup = 32 rows x ALL columns (4 waves x NT 32-column tiles): x is cut once into LDS planes, the weight
//                  fragments stream from L2 straight into registers (16-byte coalesced loads, one K-step ahead), six piece
//                  products per fp32 product on v_mfma_f32_32x32x16_bf16 (fp32-level error, see propagate_split.hip).
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
#define LDS_AS(T, p) ((__attribute__((address_space(3))) T*)(p))

namespace {

__device__ __forceinline__ float as_f(uint32_t u) { return __builtin_bit_cast(float, u); }
__device__ __forceinline__ uint32_t as_u(float f) { return __builtin_bit_cast(uint32_t, f); }

__device__ __forceinline__ void cut4(float x0, float x1, float x2, float x3, u32x2& p1, u32x2& p2, u32x2& p3) {
    const uint32_t hm = 0xffff0000u;
    p1 = u32x2{__builtin_amdgcn_perm(as_u(x1), as_u(x0), 0x07060302u), __builtin_amdgcn_perm(as_u(x3), as_u(x2), 0x07060302u)};
    x0 -= as_f(as_u(x0) & hm); x1 -= as_f(as_u(x1) & hm); x2 -= as_f(as_u(x2) & hm); x3 -= as_f(as_u(x3) & hm);
    p2 = u32x2{__builtin_amdgcn_perm(as_u(x1), as_u(x0), 0x07060302u), __builtin_amdgcn_perm(as_u(x3), as_u(x2), 0x07060302u)};
    x0 -= as_f(as_u(x0) & hm); x1 -= as_f(as_u(x1) & hm); x2 -= as_f(as_u(x2) & hm); x3 -= as_f(as_u(x3) & hm);
    p3 = u32x2{__builtin_amdgcn_perm(as_u(x1), as_u(x0), 0x07060302u), __builtin_amdgcn_perm(as_u(x3), as_u(x2), 0x07060302u)};
}

__device__ __forceinline__ f32x16 mfma32(u32x4 a, u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// planes: [tile][ks][piece][64 lanes] u32x4.  kmajor = 0: W is (N, K), element (n, k) at n * ldw + k;  1: W is (K, N), k * ldw + n
__global__ __launch_bounds__(256) void ps_cut_kernel(const float* __restrict__ W, u32x4* __restrict__ planes, int N, int K, int ldw,
                                                     int nks, int kmajor, int total) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int lane = idx & 63, blk = idx >> 6;
    const int ks = blk % nks, t = blk / nks;
    const int n = 32 * t + (lane & 31), k0 = 16 * ks + 8 * (lane >> 5);
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int k = k0 + e;
        v[e] = (n < N && k < K) ? (kmajor ? W[(int64_t)k * ldw + n] : W[(int64_t)n * ldw + k]) : 0.f;
    }
    u32x2 a1, a2, a3, b1, b2, b3;
    cut4(v[0], v[1], v[2], v[3], a1, a2, a3);
    cut4(v[4], v[5], v[6], v[7], b1, b2, b3);
    u32x4* dst = planes + (int64_t)blk * 3 * 64 + lane;
    dst[0] = u32x4{a1.x, a1.y, b1.x, b1.y};
    dst[64] = u32x4{a2.x, a2.y, b2.x, b2.y};
    dst[128] = u32x4{a3.x, a3.y, b3.x, b3.y};
}

// NT: 32-column tiles per wave; a workgroup covers 128 NT columns (column group blockIdx.y) of 32 rows.  The weight fragments
// of K-step ks + 4 are requested right behind the products of K-step ks (a ring of four register sets, the loop unrolled by
// four so that no set is ever copied): ~1 us of products between a request and its use, two workgroups per CU.
template <int NT>
__global__ __launch_bounds__(256, 2) void ps_project_kernel(const float* __restrict__ X, const u32x4* __restrict__ planes,
                                                            const float* __restrict__ bias, float* __restrict__ Y, int R, int K,
                                                            int N, int ldx, int ldy, int nks, int relu) {
    extern __shared__ __attribute__((aligned(16))) unsigned char ps_smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int row0 = blockIdx.x * 32;
    const int tile0 = (blockIdx.y * 4 + w) * NT;
    const int Kp = 16 * nks;
    const int rstride = 2 * Kp + 16;                       // bytes per row of an x plane (+ 16: rows 4 banks apart)
    const int pstride = 32 * rstride;
    const uint32_t lds0 = (uint32_t)(uintptr_t)LDS_AS(void, ps_smem);

    const int ntiles = (N + 31) >> 5;
    const u32x4* bsrc[NT];
    bool ton[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int tile = tile0 + t;
        ton[t] = tile < ntiles;
        bsrc[t] = planes + (int64_t)(ton[t] ? tile : 0) * nks * 3 * 64 + lane;
    }
    u32x4 b0[NT][3], b1[NT][3], b2[NT][3], b3[NT][3];
#define PS_ISSUE(SET, KS)                                                                      \
    do {                                                                                       \
        const int ks_ = (KS) < nks ? (KS) : nks - 1;                                           \
        _Pragma("unroll") for (int t = 0; t < NT; ++t)                                         \
            _Pragma("unroll") for (int q = 0; q < 3; ++q) SET[t][q] = bsrc[t][(ks_ * 3 + q) * 64]; \
    } while (0)
    PS_ISSUE(b0, 0); PS_ISSUE(b1, 1); PS_ISSUE(b2, 2); PS_ISSUE(b3, 3);

    // ---- x: 32 rows x Kp, cut once, three bf16 planes [row][k]
    const int k4n = Kp >> 2;                               // float4 per row
    for (int i = tid; i < 32 * k4n; i += 256) {
        const int r = i / k4n, c4 = i - r * k4n;
        const int row = row0 + r;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (row < R && 4 * c4 < K) v = *reinterpret_cast<const float4*>(X + (int64_t)row * ldx + 4 * c4);   // (K % 4 == 0)
        u32x2 p1, p2, p3;
        cut4(v.x, v.y, v.z, v.w, p1, p2, p3);
        const uint32_t d = lds0 + r * rstride + 8 * c4;
        *LDS_AS(u32x2, (uintptr_t)d) = p1;
        *LDS_AS(u32x2, (uintptr_t)(d + pstride)) = p2;
        *LDS_AS(u32x2, (uintptr_t)(d + 2 * pstride)) = p3;
    }
    __syncthreads();

    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    const uint32_t ard = lds0 + (lane & 31) * rstride + 16 * (lane >> 5);

#define PS_STEP(SET, KS)                                                                               \
    do {                                                                                               \
        if ((KS) < nks) {                                                                              \
            u32x4 a_[3];                                                                               \
            _Pragma("unroll") for (int q = 0; q < 3; ++q) a_[q] = *LDS_AS(u32x4, (uintptr_t)(ard + q * pstride + 32 * (KS))); \
            _Pragma("unroll") for (int pc = 0; pc < 6; ++pc) {                                         \
                const int ai = (pc == 0) ? 2 : (pc == 1 || pc == 3) ? 1 : 0;                           \
                const int bi = (pc < 3) ? 0 : (pc < 5) ? 1 : 2;                                        \
                _Pragma("unroll") for (int t = 0; t < NT; ++t) acc[t] = mfma32(a_[ai], SET[t][bi], acc[t]); \
            }                                                                                          \
            if ((KS) + 4 < nks) PS_ISSUE(SET, (KS) + 4);                                               \
        }                                                                                              \
    } while (0)
#pragma unroll 1
    for (int ks = 0; ks < nks; ks += 4) {
        PS_STEP(b0, ks); PS_STEP(b1, ks + 1); PS_STEP(b2, ks + 2); PS_STEP(b3, ks + 3);
    }
#undef PS_STEP
#undef PS_ISSUE

    // ---- epilogue: D layout of a 32 x 32 tile: column lane & 31, rows (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
    const int l32 = lane & 31, kg = lane >> 5;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int n = 32 * (tile0 + t) + l32;
        if (!ton[t] || n >= N) continue;
        const float bv = bias ? bias[n] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = row0 + (r & 3) + 8 * (r >> 2) + 4 * kg;
            if (row < R) {
                float o = acc[t][r] + bv;
                if (relu) o = fmaxf(o, 0.f);
                Y[(int64_t)row * ldy + n] = o;
            }
        }
    }
}

}  // namespace

extern "C" int64_t ps_planes_bytes(int N, int K) { return (int64_t)((N + 31) / 32) * ((K + 15) / 16) * 3 * 1024; }

extern "C" int ps_cut_weight(const float* W, void* planes, int N, int K, int ldw, int kmajor, void* stream) {
    const int nks = (K + 15) / 16, ntiles = (N + 31) / 32;
    const int total = ntiles * nks * 64;
    hipLaunchKernelGGL(ps_cut_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, W, (u32x4*)planes, N, K, ldw, nks,
                       kmajor, total);
    return (int)hipGetLastError();
}

extern "C" int ps_project(const float* X, const void* planes, const float* bias, float* Y, int R, int K, int N, int ldx, int ldy,
                          int relu, void* stream) {
    if (K & 3) return -1;
    const int nks = (K + 15) / 16;
    const int ntiles = (N + 31) / 32;
    const size_t lds = (size_t)3 * 32 * (2 * 16 * nks + 16);
    if (lds > 64 * 1024) return -2;
    // column groups of 128 NT columns: NT = 3 (384 columns per group) for wide outputs, NT = 2 (256) otherwise
    const int NTsel = ntiles > 8 ? 3 : 2;
    const int ncg = (ntiles + 4 * NTsel - 1) / (4 * NTsel);
    const dim3 grid((R + 31) / 32, ncg);
    if (NTsel == 2) hipLaunchKernelGGL((ps_project_kernel<2>), grid, dim3(256), lds, (hipStream_t)stream, X, (const u32x4*)planes, bias, Y, R, K, N, ldx, ldy, nks, relu);
    else hipLaunchKernelGGL((ps_project_kernel<3>), grid, dim3(256), lds, (hipStream_t)stream, X, (const u32x4*)planes, bias, Y, R, K, N, ldx, ldy, nks, relu);
    return (int)hipGetLastError();
}
