// exp_wide2.h -- the experimental 128-row x two-sample forward conv tiling (tools/exp_wide2.hip; convbench only)
#pragma once
#include <hip/hip_runtime.h>
namespace rau {
bool conv_wide2_ok(int M, int K, int S, long w_rs);
hipError_t conv_wide2(hipStream_t st, int nB, int M, int K, int S, const float* Wt, long w_rs, const float* X,
                      long x_bs, float* C, long c_bs, const float* bias, int act, int per_cu = 1);
}  // namespace rau
