// Device side of the dropout keep-flag generator (encoder_glue.hip: see the comment there), as a function of (block, blocks,
// threads per block), so that the draw of a step's flags can also run as RIDER workgroups of the first GRU layer's forward
// recurrence launch (gru.hip: 96 of 256 CUs idle for 69 us at cfg2; the flags' first consumer is the dropout behind that layer).
// A code header, included by exactly those two translation units.
#pragma once
#include "mmdfn_internal.h"

namespace kfb {

struct FlagJob {
    float* out;
    int64_t n8, n4;
    uint32_t threshold;
    int all;
    unsigned long long* state;
};

__device__ __forceinline__ uint4 philox4x32_10(uint4 ctr, uint2 key) {
    constexpr uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned long long p0 = (unsigned long long)M0 * ctr.x, p1 = (unsigned long long)M1 * ctr.z;   // one v_mad_u64_u32 each
        const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0, hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
        ctr = make_uint4(hi1 ^ ctr.y ^ key.x, lo1, hi0 ^ ctr.w ^ key.y, lo0);
        key.x += W0;
        key.y += W1;
    }
    return ctr;
}

// block `bid` of `nblocks`, NT threads each (a multiple of 64): which flag lands where depends on neither
template <int NT>
__device__ __forceinline__ void keep_flags_block(const FlagJob& J, const int bid, const int nblocks) {
    float* __restrict__ out = J.out;
    unsigned long long* __restrict__ state = J.state;
    const int64_t n8 = J.n8, n4 = J.n4;
    const uint32_t threshold = J.threshold;
    const int all = J.all;
    const unsigned long long seed = state[0], offset = state[1];
    const uint2 key = make_uint2((uint32_t)seed, (uint32_t)(seed >> 32));
    // one Philox call = 128 random bits = EIGHT flags (16 bits each: the keep rate is exact to 2^-16, the generator is the
    // multiplier-bound part of the kernel: 19 v_mad_u64_u32 per call)
    // (whole waves: n8 is a multiple of 64 -- the launcher rounds the counter range up, the slot guards below cut the tail)
    for (int64_t i = (int64_t)bid * NT + threadIdx.x; i < n8; i += (int64_t)nblocks * NT) {
        const unsigned long long c = offset + (unsigned long long)i;
        const uint4 r = philox4x32_10(make_uint4((uint32_t)c, (uint32_t)(c >> 32), 0u, 0u), key);
        float4 v, u;
        v.x = (all || (r.x & 0xFFFFu) < threshold) ? 1.f : 0.f; v.y = (all || (r.x >> 16) < threshold) ? 1.f : 0.f;
        v.z = (all || (r.y & 0xFFFFu) < threshold) ? 1.f : 0.f; v.w = (all || (r.y >> 16) < threshold) ? 1.f : 0.f;
        u.x = (all || (r.z & 0xFFFFu) < threshold) ? 1.f : 0.f; u.y = (all || (r.z >> 16) < threshold) ? 1.f : 0.f;
        u.z = (all || (r.w & 0xFFFFu) < threshold) ? 1.f : 0.f; u.w = (all || (r.w >> 16) < threshold) ? 1.f : 0.f;
        // the wave's 128 float4 slots as two contiguous 1 KB stores (which flag lands where is immaterial)
        const int lane = threadIdx.x & 63;
        const int64_t sa = 2 * (i - lane) + lane, sb = sa + 64;
        if (sa < n4) __builtin_nontemporal_store(__builtin_bit_cast(f32x4, v), reinterpret_cast<f32x4*>(out + 4 * sa));
        if (sb < n4) __builtin_nontemporal_store(__builtin_bit_cast(f32x4, u), reinterpret_cast<f32x4*>(out + 4 * sb));
    }
    __shared__ int last_s;
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned done = atomicAdd(reinterpret_cast<unsigned*>(&state[2]), 1u);
        last_s = (done == (unsigned)nblocks - 1u) ? 1 : 0;
    }
    __syncthreads();
    if (last_s && threadIdx.x == 0) {
        state[1] = offset + (unsigned long long)n8;
        state[2] = 0ull;
    }
}

}  // namespace kfb

// (encoder_glue.hip) the draw staged by mmdfn_keep_flags_stage, or nullptr; the launch that ran it says so
const kfb::FlagJob* mmdfn_flag_job_pending();
void mmdfn_flag_job_taken();
