// common.h -- shared device/host helpers for librau.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rau {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kWave = 64;  // CDNA wavefront width

// Kernels of the latency-bound recurrence chain run next to the bulk GEMMs of the
// other stream; instruction arbitration on a SIMD is by priority, then age, and the
// bulk waves are always older.  Chain kernels therefore raise their wave priority.
#define RAU_CHAIN_PRIO() __builtin_amdgcn_s_setprio(3)

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }
// tanh to ~1e-7 absolute: odd polynomial below 1/8 (no cancellation), else
// 1 - 2/(exp(2|x|)+1) on the hardware exp; saturates cleanly to +-1.
__device__ __forceinline__ float tanh_fast(float x) {
  const float ax = fabsf(x), x2 = x * x;
  const float p = x * (1.f + x2 * (-0.33333334f + x2 * (0.13333334f + x2 * -0.053968254f)));
  const float t = 1.f - 2.f / (__expf(2.f * ax) + 1.f);
  return ax < 0.125f ? p : copysignf(t, x);
}

// Dropout keep-bit of flat element `e` of a bit-packed mask (bit e&31 of word e>>5).
__device__ __forceinline__ uint32_t mask_bit(const uint32_t* __restrict__ bits, size_t e) {
  return (bits[e >> 5] >> (e & 31)) & 1u;
}
// Four keep bits of elements e..e+3 (e % 4 == 0), in bits 0..3.
__device__ __forceinline__ uint32_t mask_nib(const uint32_t* __restrict__ bits, size_t e) {
  return (bits[e >> 5] >> (e & 31)) & 0xFu;
}

// XCD-aware remap of a linear workgroup id: blocks id and id+8 share an XCD, so
// give each XCD a contiguous chunk of the logical grid (bijective for any nwg).
__device__ __forceinline__ int xcd_remap(int id, int nwg) {
  const int xcd = id & 7, q = nwg >> 3, r = nwg & 7;
  const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + (id >> 3);
}

// x rounded to bf16 (round to nearest even), as an f32 value: what a bf16 MFMA operand holds.  A product
// of two such values is exact in f32, so an f32 MFMA on rounded operands and a bf16 MFMA differ only in
// the order of their f32 additions.
__device__ __forceinline__ float rb16(float x) { return (float)(__bf16)x; }
__device__ __forceinline__ float4 rb16(const float4& v) {
  return make_float4(rb16(v.x), rb16(v.y), rb16(v.z), rb16(v.w));
}

}  // namespace rau
