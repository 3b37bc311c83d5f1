// tools/exp_fwd16.hip (experiment, see its header)
#pragma once
#include <hip/hip_runtime.h>
namespace rau {
bool fwd16_ok(int nB, int D, int S, int M);
hipError_t fwd16(hipStream_t st, int nB, int D, int S, int M, const void* X16, const void* WD,
                 const float* bi, float* I);
hipError_t weights_kblocked(hipStream_t st, int M, int D, const float* W, void* WD);
}  // namespace rau
