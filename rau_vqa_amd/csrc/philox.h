// philox.h -- Philox4x32-10 counter-based generator (host + device).
// Dropout masks are a pure function of (seed, site, step, element index), so the
// backward pass and the CPU oracle (oracle/rau_cpu.cc: rau_oracle_fill_mask)
// regenerate exactly the same bits without any stored RNG state.
#pragma once
#include <stdint.h>
#ifdef __HIPCC__
#include <hip/hip_runtime.h>
#define RAU_HD __host__ __device__ __forceinline__
#else
#define RAU_HD inline
#endif

namespace rau {

struct Philox4 {
  uint32_t v[4];
};

RAU_HD Philox4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                             uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    const uint32_t n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    const uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  Philox4 o;
  o.v[0] = c0; o.v[1] = c1; o.v[2] = c2; o.v[3] = c3;
  return o;
}

// 16 keep bits for elements [16*blk, 16*blk+16) of mask `site` at `step`:
// byte j of the 128-bit output (little-endian in each word) >= thr  -> keep.
RAU_HD uint32_t philox_keep16(uint64_t seed, uint32_t site, uint32_t step, uint64_t blk,
                              uint32_t thr) {
  const Philox4 o = philox4x32_10((uint32_t)blk, (uint32_t)(blk >> 32), site, step,
                                  (uint32_t)seed, (uint32_t)(seed >> 32));
  uint32_t bits = 0;
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const uint32_t byte = (o.v[j >> 2] >> (8 * (j & 3))) & 0xFFu;
    bits |= (byte >= thr ? 1u : 0u) << j;
  }
  return bits;
}

}  // namespace rau
