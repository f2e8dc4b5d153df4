// Experiment library (NOT part of libmmdfn_hip.so): K6 on producer-cut bf16 piece planes.  See README.md here and
// profiles/r03_k6_memory_path.md for what was measured and why none of it ships.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
extern "C" int mmdfn_cut_planes(const float* X, void* planes, int64_t R, int d, int ldx, void* stream);
extern "C" int mmdfn_propagate_planes(const float* tiles, const float* cross, const float* H, const void* planes,
                                      float* out, const int32_t* dia_len, const int32_t* row_start,
                                      const int64_t* tile_base, int B, int M, int N, int d, int ldh, int ldo, int max_len,
                                      void* stream);
int mmdfn_launch_propagate_planes2(const float* tiles, const float* cross, const float* H, const void* planes, float* out,
                                   const int32_t* dia_len, const int32_t* row_start, const int64_t* tile_base, int B, int M,
                                   int N, int d, int ldh, int ldo, int max_len, int variant, int nt, int abl, int km, hipStream_t s);
