// philox.h - Philox4x32-10 counter RNG.  counter = (residue, global patch, step, stream), key = seed.
// Bit-identical (integer part) to oracle/diffab_oracle.py::philox4x32; the noise a residue sees
// depends only on (seed, global patch id, residue, step, stream), never on the launch geometry or
// on how a batch is sharded over ranks.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace diffab {

enum PhiloxStream : uint32_t {
  STREAM_SEQ = 0,     // categorical draw of s_{t-1}
  STREAM_TRANS = 1,   // translation noise z
  STREAM_AXIS = 2,    // rotation axis
  STREAM_ANGLE = 3,   // rotation angle (u_bin, u_in, z via Box-Muller of lanes 2,3)
  STREAM_INIT_X = 4,
  STREAM_INIT_O = 5,
  STREAM_INIT_S = 6,
};

struct u32x4 {
  uint32_t x, y, z, w;
};

__host__ __device__ inline u32x4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
  constexpr uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = static_cast<uint64_t>(M0) * c0;
    const uint64_t p1 = static_cast<uint64_t>(M1) * c2;
    const uint32_t hi0 = static_cast<uint32_t>(p0 >> 32), lo0 = static_cast<uint32_t>(p0);
    const uint32_t hi1 = static_cast<uint32_t>(p1 >> 32), lo1 = static_cast<uint32_t>(p1);
    const uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += W0;
    k1 += W1;
  }
  return {c0, c1, c2, c3};
}

// (0,1) from the top 24 bits: (x >> 8) 2^-24 + 2^-25, never 0.  For x >> 8 == 2^24 - 1 the sum 1 - 2^-25 ties to even = 1.0f in fp32,
// so the largest value is clamped to 1 - 2^-24: never 1 either (-log(u) > 0: the exponential race of the IGSO3 bin draw divides by it).
__host__ __device__ inline float u32_to_unit(uint32_t x) {
  const float u = static_cast<float>(x >> 8) * 5.9604644775390625e-8f + 2.98023223876953125e-8f;
  return u < 0.99999994f ? u : 0.99999994f;
}

struct f32x4 {
  float x, y, z, w;
};

__device__ inline f32x4 philox_uniform4(uint64_t seed, uint32_t patch, uint32_t residue, uint32_t step, uint32_t stream) {
  const u32x4 r = philox4x32_10(residue, patch, step, stream, static_cast<uint32_t>(seed), static_cast<uint32_t>(seed >> 32));
  return {u32_to_unit(r.x), u32_to_unit(r.y), u32_to_unit(r.z), u32_to_unit(r.w)};
}

// Box-Muller on (u0,u1) and (u2,u3).
__device__ inline f32x4 normals_from_uniforms(const f32x4 u) {
  const float two_pi = 6.283185307179586f;
  const float r0 = sqrtf(-2.0f * logf(u.x));
  const float r1 = sqrtf(-2.0f * logf(u.z));
  float s0, c0, s1, c1;
  sincosf(two_pi * u.y, &s0, &c0);
  sincosf(two_pi * u.w, &s1, &c1);
  return {r0 * c0, r0 * s0, r1 * c1, r1 * s1};
}

__device__ inline f32x4 philox_normal4(uint64_t seed, uint32_t patch, uint32_t residue, uint32_t step, uint32_t stream) {
  return normals_from_uniforms(philox_uniform4(seed, patch, residue, step, stream));
}

}  // namespace diffab
