// rowgemm_h3_tile.h - one ROWS x 128 output tile of Y = act(X W^T + b) on the f16 matrix cores as a THREE-term split product.
//
// gemm_bf16x6.hip writes an fp32 operand as three bf16 pieces (8 + 8 + 8 mantissa bits, fp32's exponent range) and needs six partial
// products.  With a power-of-two scale that puts the largest magnitude of a vector into [128, 256), TWO fp16 pieces (11 + 11 bits) hold
// it to 2^-22 of that maximum - what the attention kernel's pair planes have been doing since round 2 - and three partial products
// (h2 w1, h1 w2, h1 w1; h2 w2 < 2^-22 is dropped) reproduce the fp32 product: half the matrix-pipe time, two thirds of the LDS traffic
// of the six-term form (ablation with three of the six bf16 terms: to_out 57 -> 43 us, projections 67 -> 51 us per layer at B = 256).
// Scales: W one per output row n (all of K; fixed when the planes are written: wsplit128_h3), X one per (row, 64 k = two chunks), found by
// the eight lanes that stage that row from the operand ring (the pair's second chunk was requested three chunks earlier) - no pass
// over X in advance.  A pair is accumulated from zero on the matrix cores and joined on the VALU: tot = fma(acc, 2^-sx(row, pair), tot),
// pairs in ascending order, y = fma(tot, 2^-sw(n), bias) - the same operations for every launch shape (the sampler's bitwise shard
// invariance; PARTIAL: one pair per work-group as acc x 2^-sx, which is exact, added up in the same order by the parts-sum kernel).
#pragma once
#include <hip/hip_runtime.h>

namespace diffab {
namespace h3tile {
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int BK = 32;           // k per weight chunk
constexpr int PART_CHUNKS = 2;   // chunks per part (64 k): the unit of the X scale and of the fixed summation order

__device__ __forceinline__ int h3_off(int row, int slot) { return row * BK + 8 * (slot ^ ((row >> 2) & 3)); }  // fp16 elements
// power-of-two scale for a vector whose largest magnitude is m: s puts it into [128, 256); is = 1 / s (both exact); m = 0, below 2^-119
// (biased exponent < 8: 1 / s = 2^(e - 7 - 127) would not be a normal number, s = 2^(261 - e - 127) would overflow) or huge: no scaling
__device__ __forceinline__ void h3_scale(float m, float& s, float& is) {
  const int e = static_cast<int>((__float_as_uint(m) >> 23) & 255u);
  const bool ok = e >= 8 && e < 231;
  s = ok ? __uint_as_float(static_cast<unsigned>(127 + 7 + 127 - e) << 23) : 1.0f;
  is = ok ? __uint_as_float(static_cast<unsigned>(e - 7) << 23) : 1.0f;
}
__device__ __forceinline__ void split2(float x, _Float16& h1, _Float16& h2) {
  h1 = static_cast<_Float16>(x);
  h2 = static_cast<_Float16>(x - static_cast<float>(h1));
}
template <int ROWS>
constexpr int lds_bytes() { return 2 * 2 * (128 + ROWS) * BK * 2 + 2 * ROWS * 4; }  // W and X planes (two buffers), X pair scales (two pairs)

// h3_lds: lds_bytes<ROWS>() bytes, 16-byte aligned; ROWS * 4 threads; Wc: planes of wsplit128_h3 ([chunk][2][128][32] fp16), wis[128]:
// 1 / scale of the weight rows.  PARTIAL: ONE part (chunks [c_begin, c_begin + PART_CHUNKS)) as raw sums (before 2^-sw and the bias)
// into part_out[ROWS][128].
template <bool RELU, int ROWS, bool PARTIAL = false>
__device__ __forceinline__ void rowgemm128_h3_tile(_Float16* h3_lds, int tid, int tile_m, const float* __restrict__ X, int ldx,
                                                   const _Float16* __restrict__ Wc, const float* __restrict__ wis, const float* __restrict__ bias,
                                                   const int64_t* __restrict__ bias_idx, int bias_div, float* __restrict__ Y, int ldy, int M, int Kd,
                                                   int c_begin = 0, float* __restrict__ part_out = nullptr) {
#define H3TILE_FENCE() asm volatile("" ::: "memory")
  constexpr int T = ROWS * 4, NRW = ROWS / 32;
  _Float16* Ws = h3_lds;                          // [2 buf][2 planes][128][32]
  _Float16* As = h3_lds + 2 * 2 * 128 * BK;       // [2 buf][2 planes][ROWS][32]
  float* Sx = reinterpret_cast<float*>(As + 2 * 2 * ROWS * BK);  // [2 buf][ROWS]: 2^-sx of the staged row chunks
  const int lane = tid & 63, wv = tid >> 6;
  const int l31 = lane & 31, hk = lane >> 5, rw = wv % NRW, cw = wv / NRW;  // v_mfma_f32_32x32x16_f16: wave tile 32 rows x 64 columns
  const int m0 = tile_m * ROWS;
  const int nchunk = PARTIAL ? min(Kd / BK, c_begin + PART_CHUNKS) : Kd / BK;
  // weight staging: a chunk is 2 planes x 128 rows x 64 bytes = 1024 16-byte pieces, a straight copy of 16 KiB
  constexpr int WP = 1024 / T;  // pieces per thread (2 or 4)
  int w_dst[WP];
#pragma unroll
  for (int i = 0; i < WP; ++i) {
    const int idx = tid + T * i, p = idx / 512, row = (idx % 512) >> 2, part = idx & 3;
    w_dst[i] = (p * 128) * BK + h3_off(row, part);
  }
  f32x4 wreg[2][WP];
  auto load_w = [&](int slot, int ch) {
    ch = ch < nchunk ? ch : nchunk - 1;  // unconditional prefetch (a branch around it makes the compiler wait for it at once)
    const _Float16* src = Wc + static_cast<size_t>(ch) * (2 * 128 * BK) + tid * 8;
#pragma unroll
    for (int i = 0; i < WP; ++i) wreg[slot][i] = *reinterpret_cast<const f32x4*>(src + T * 8 * i);
  };
  auto store_w = [&](int slot, int buf) {
#pragma unroll
    for (int i = 0; i < WP; ++i) *reinterpret_cast<f32x4*>(Ws + buf * (2 * 128 * BK) + w_dst[i]) = wreg[slot][i];
  };
  // X staging: thread -> rows tid / 8 and ROWS / 2 + tid / 8, 16-byte part tid % 8 (8 lanes = one 128-byte line = one row chunk)
  const int xa_row = tid >> 3, xa_part = tid & 7;
  const float* xsrc[2];
  int x_dst[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int lrow = xa_row + (ROWS / 2) * j;
    int row = m0 + lrow;
    row = row < M ? row : M - 1;  // clamped (never stored)
    xsrc[j] = X + static_cast<int64_t>(row) * ldx + 4 * xa_part;
    x_dst[j] = h3_off(lrow, xa_part >> 1) + 4 * (xa_part & 1);
  }
  f32x4 xreg[4][2];  // ring of four chunks: slot = chunk % 4 (relative to c_begin)
  auto load_x = [&](int slot, int ch) {
    ch = ch < nchunk ? ch : nchunk - 1;
#pragma unroll
    for (int j = 0; j < 2; ++j) xreg[slot][j] = *reinterpret_cast<const f32x4*>(xsrc[j] + ch * BK);
  };
  float sx_cur[2];  // scale of the pair being staged, for the thread's two rows
  auto pair_scale = [&](int slot_a, int slot_b, int pp) {  // maximum of the row over the pair's two chunks (8 lanes x 2 x 4 values)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const f32x4 va = xreg[slot_a][j], vb = xreg[slot_b][j];
      float m = fmaxf(fmaxf(fmaxf(fabsf(va[0]), fabsf(va[1])), fmaxf(fabsf(va[2]), fabsf(va[3]))),
                      fmaxf(fmaxf(fabsf(vb[0]), fabsf(vb[1])), fmaxf(fabsf(vb[2]), fabsf(vb[3]))));
      m = fmaxf(m, __shfl_xor(m, 1));
      m = fmaxf(m, __shfl_xor(m, 2));
      m = fmaxf(m, __shfl_xor(m, 4));
      float is;
      h3_scale(m, sx_cur[j], is);
      if (xa_part == 0) Sx[pp * ROWS + xa_row + (ROWS / 2) * j] = is;
    }
  };
  auto store_x = [&](int slot, int buf) {  // fp32 x scale -> two fp16 planes -> LDS
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const f32x4 v = xreg[slot][j];
      f16x4 h1, h2;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        _Float16 a, b;
        split2(v[c] * sx_cur[j], a, b);
        h1[c] = a; h2[c] = b;
      }
      _Float16* dst = As + buf * (2 * ROWS * BK) + x_dst[j];
      *reinterpret_cast<f16x4*>(dst) = h1;
      *reinterpret_cast<f16x4*>(dst + ROWS * BK) = h2;
    }
  };
  f32x16 acc[2], tot[2];  // wave tile 32 x 64 = two 32 x 32 tiles: the running pair | the pairs before it
#pragma unroll
  for (int tt = 0; tt < 2; ++tt)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[tt][r] = tot[tt][r] = 0.f;

#pragma unroll
  for (int c = 0; c < 4; ++c) load_x(c, c_begin + c);
  load_w(0, c_begin);
  load_w(1, c_begin + 1);
  H3TILE_FENCE();
  store_w(0, 0);
  pair_scale(0, 1, 0);
  store_x(0, 0);
  load_w(0, c_begin + 2);
  load_x(0, c_begin + 4);
  H3TILE_FENCE();
  __syncthreads();
  const int fx = (l31 >> 2) & 3;
  const int a_off = (32 * rw + l31) * BK, w_off = (64 * cw + l31) * BK;
  for (int ch0 = c_begin; ch0 < nchunk; ch0 += 4) {  // ring slots are compile-time indices: four chunks (two pairs) per trip
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int ch = ch0 + u, buf = u & 1;
      if (ch >= nchunk) break;
      const _Float16* al = As + buf * (2 * ROWS * BK) + a_off;
      const _Float16* wl = Ws + buf * (2 * 128 * BK) + w_off;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const int so = 8 * ((2 * ks + hk) ^ fx);
        f16x8 a[2], b[2][2];
#pragma unroll
        for (int p = 0; p < 2; ++p) {
          a[p] = *reinterpret_cast<const f16x8*>(al + (p * ROWS) * BK + so);
#pragma unroll
          for (int tt = 0; tt < 2; ++tt) b[tt][p] = *reinterpret_cast<const f16x8*>(wl + (p * 128 + 32 * tt) * BK + so);
        }
        // (h2 w1) (h1 w2) (h1 w1): smallest first
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) acc[tt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[1], b[tt][0], acc[tt], 0, 0, 0);
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) acc[tt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0], b[tt][1], acc[tt], 0, 0, 0);
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) acc[tt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0], b[tt][0], acc[tt], 0, 0, 0);
      }
      // stage chunk ch + 1 into the other buffers (the first chunk of a pair brings the pair's scale), refill the ring slots
      store_w((u + 1) & 1, buf ^ 1);
      if (u & 1) pair_scale((u + 1) & 3, (u + 2) & 3, ((u + 1) >> 1) & 1);
      store_x((u + 1) & 3, buf ^ 1);
      load_w((u + 1) & 1, ch + 3);
      load_x((u + 1) & 3, ch + 5);
      H3TILE_FENCE();
      if (u & 1) {  // the pair is complete: D 32x32 row = (r & 3) + 8 (r >> 2) + 4 hk (+ 32 rw): four consecutive rows per r >> 2
        f32x4 sx[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) sx[g] = *reinterpret_cast<const f32x4*>(Sx + (u >> 1) * ROWS + 32 * rw + 8 * g + 4 * hk);
#pragma unroll
        for (int tt = 0; tt < 2; ++tt)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            tot[tt][r] = __builtin_fmaf(acc[tt][r], sx[r >> 2][r & 3], tot[tt][r]);
            acc[tt][r] = 0.f;
          }
      }
      __syncthreads();
    }
  }
  if (PARTIAL) {
#pragma unroll
    for (int tt = 0; tt < 2; ++tt)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        part_out[(32 * rw + (r & 3) + 8 * (r >> 2) + 4 * hk) * 128 + 64 * cw + 32 * tt + l31] = tot[tt][r];
    return;
  }
  const bool table = bias_idx != nullptr || bias_div > 0;
#pragma unroll
  for (int tt = 0; tt < 2; ++tt) {
    const int col = 64 * cw + 32 * tt + l31;
    const float wi = wis[col];
    float bv = (bias && !table) ? bias[col] : 0.0f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = m0 + 32 * rw + (r & 3) + 8 * (r >> 2) + 4 * hk;
      if (row >= M) continue;
      if (table) {
        const int64_t bi = bias_idx ? bias_idx[row] : row / bias_div;
        bv = bias[bi * 128 + col];
      }
      float o = __builtin_fmaf(tot[tt][r], wi, bv);
      if (RELU) o = fmaxf(o, 0.f);
      Y[static_cast<int64_t>(row) * ldy + col] = o;
    }
  }
#undef H3TILE_FENCE
}
}  // namespace h3tile
}  // namespace diffab
