// attn_planes_tile.h - the VALUE side of one patch as fp16 planes for the attention tile's P x V product on the f16 matrix cores (round 6).
//
// Phase 3 of the attention tile (ipa_attn_tile.h, VPL) contracts the probabilities with the values of a head over the keys.  On the f32
// matrix cores that is 32 MFMAs of 32 cycles per (head, 32 keys) which also block the vector ALU; as three exact partial products of two
// fp16 planes per operand it is 12 MFMAs of 16 cycles, the B operand arrives as linear 1 KiB wave loads and needs no LDS staging.  The
// planes are cut HERE, once per (patch, layer), from the fp32 rows the projection tile has just written (L2-hot) - not inside the
// projection tile: its 204-214 of 256 VGPRs leave no room for a plane epilogue (three forms measured: +20 us of scattered stores or 31-40
// spilled VGPRs per launch, profiles/r06_attention.md; experiments/patches/r06_value_planes_in_projection_tile.patch).
//
// Layout: per (patch, head, key step T of 32 keys) 8 KiB = [tile u 0..3][plane h1, h2][lane n + 16 g][8 k slots]; k slot e of lane group g
// <-> key 32 T + 16 (e >> 2) + 4 g + (e & 3) (the order in which phase 3 reads P from the logits image); tile 0 / 1: v_s dims n / 16 + n;
// tile 2: x of point n (n < 8), y of point n - 8; tile 3: z of point n (n < 8), ONES in column 8 (the product then returns the
// probability mass as the split planes see it), zeros above.  Points are in the global frame RELATIVE to the patch's first translation
// (the planes resolve 2^-22 of a patch's spread, not of its distance from the origin; phase 3 adds (c - t_i) x mass back).  Scale: one
// power of two per (patch, head, step) and kind (v_s | points) that puts the largest magnitude into [2^13, 2^14) - both pieces of a value
// stay normal fp16 numbers down to 2^-27 of that maximum; osc[((patch * 8 + head) * K / 32 + T) * 2 + kind] = 1 / s.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace diffab {
namespace aplanes {
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
constexpr int NP = 1344, OFF_VS = 512, OFF_GV = 1152;  // column map of the projection buffer (ipa_attn_tile.h)
constexpr int LD = 60;                                   // floats per staged row: 32 v_s + 24 point coordinates + 4 pad (conflict-free reads)
constexpr int LDS_BYTES = 8 * 32 * LD * 4;               // one [32 rows][LD] slab per wave: 61 440 bytes
constexpr int STEP_HALVES = 4 * 2 * 64 * 8;              // fp16 per (patch, head, key step): 8 KiB

// power-of-two scale that puts m into [2^13, 2^14); is = 1 / s (both exact); m = 0, tiny or huge: no scaling
__device__ __forceinline__ void plane_scale(float m, float& s, float& is) {
  const int e = static_cast<int>((__float_as_uint(m) >> 23) & 255u);
  const bool ok = e >= 16 && e < 240;
  s = ok ? __uint_as_float(static_cast<unsigned>(127 + 13 + 127 - e) << 23) : 1.0f;
  is = ok ? __uint_as_float(static_cast<unsigned>(e - 13) << 23) : 1.0f;
}

// 512 threads (wave = head); lds: LDS_BYTES, 16-byte aligned; proj: [B K][1344] fp32; t: [B K][3]; vpl: [B][8][K / 32][STEP_HALVES] fp16;
// osc: [B][8][K / 32][2] floats.  K a multiple of 32.
__device__ __forceinline__ void attn_value_planes_tile(float* __restrict__ lds, const int tid, const int b, const float* __restrict__ proj,
                                                       const float* __restrict__ t, const int K, _Float16* __restrict__ vpl,
                                                       float* __restrict__ osc) {
  const int lane = tid & 63, h = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = lane & 15, g = lane >> 4;
  float* slab = lds + h * (32 * LD);
  const int64_t prow0 = static_cast<int64_t>(b) * K;
  const float* tc = t + prow0 * 3;  // the patch's first translation: the reference point of the point planes (wave-uniform: scalar loads)
  const float c0 = tc[0], c1 = tc[1], c2 = tc[2];
  const int steps = K >> 5;
  // the lane's column of each of the four tiles inside a staged row, and the centre to subtract (0 for v_s)
  int col[4];
  float cen[4];
  col[0] = n; cen[0] = 0.0f;
  col[1] = 16 + n; cen[1] = 0.0f;
  col[2] = 32 + 3 * (n & 7) + (n >> 3); cen[2] = n < 8 ? c0 : c1;
  col[3] = 32 + 3 * (n & 7) + 2; cen[3] = c2;
#pragma unroll 1
  for (int T = 0; T < steps; ++T) {
    const float* src = proj + (prow0 + 32 * T) * NP;  // 32 key rows
    // line-shaped loads (consecutive lanes on consecutive 16-byte chunks of a row): v_s 32 rows x 128 B, points 32 rows x 96 B
    f32x4 vs[4], gp[3];
#pragma unroll
    for (int i = 0; i < 4; ++i) vs[i] = *reinterpret_cast<const f32x4*>(src + (8 * i + (lane >> 3)) * NP + OFF_VS + 32 * h + 4 * (lane & 7));
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int id = 64 * i + lane;  // chunk id 0..191: row id / 6, chunk id % 6
      gp[i] = *reinterpret_cast<const f32x4*>(src + (id / 6) * NP + OFF_GV + 24 * h + 4 * (id % 6));
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4*>(slab + (8 * i + (lane >> 3)) * LD + 4 * (lane & 7)) = vs[i];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int id = 64 * i + lane;
      *reinterpret_cast<f32x4*>(slab + (id / 6) * LD + 32 + 4 * (id % 6)) = gp[i];
    }
    // (a wave reads only what it wrote, and LDS operations of one wave complete in order: no barrier)
    float x[4][8];
    float mv = 0.0f, mp = 0.0f;
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float v = slab[(16 * (e >> 2) + 4 * g + (e & 3)) * LD + col[u]] - cen[u];
        x[u][e] = v;
        if (u < 2) mv = fmaxf(mv, fabsf(v));
        else if (u == 2 || n < 8) mp = fmaxf(mp, fabsf(v));
      }
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      mv = fmaxf(mv, __shfl_xor(mv, o));
      mp = fmaxf(mp, __shfl_xor(mp, o));
    }
    float sv, isv, sp, isp;
    plane_scale(mv, sv, isv);
    plane_scale(mp, sp, isp);
    const int64_t blk = (static_cast<int64_t>(b) * 8 + h) * steps + T;
    if (lane == 0) {
      osc[blk * 2] = isv;
      osc[blk * 2 + 1] = isp;
    }
    _Float16* dst = vpl + blk * STEP_HALVES + lane * 8;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      f16x8 h1, h2;
      const float s = u < 2 ? sv : sp;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float xs = x[u][e] * s;
        if (u == 3 && n >= 8) xs = n == 8 ? 1.0f : 0.0f;  // the ones column (unscaled) and the unused columns of the z tile
        const _Float16 a1 = static_cast<_Float16>(xs);
        h1[e] = a1;
        h2[e] = static_cast<_Float16>(xs - static_cast<float>(a1));
      }
      *reinterpret_cast<f16x8*>(dst + (2 * u) * 512) = h1;
      *reinterpret_cast<f16x8*>(dst + (2 * u + 1) * 512) = h2;
    }
  }
}

}  // namespace aplanes
}  // namespace diffab
