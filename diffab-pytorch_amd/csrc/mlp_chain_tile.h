// mlp_chain_tile.h - a chain of two or three 128-wide dense layers for ONE 128-row tile (bf16 x 6 split products), as a device function:
// the body of mlp_chain_b6_kernel (gemm_bf16x6.hip: one tile per work-group) and of the embedding / head phases of the patch-resident
// module kernel (ipa_persistent.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

#include "rowgemm_b6_tile.h"

namespace diffab {
// A chain of two or three 128-wide dense layers without leaving the CU:
// Y = L3(relu(L2(relu(L1(X))))) (or two layers) for 128 rows per work-group, every layer K = 128 -> 128 columns (the last one n_out <=
// 128 wide, its missing weight rows zero planes).  The denoiser's MLPs (embedding: 2 layers; the three heads: 3 layers each) were
// nine + two launches of 10-19 us, each latency-bound (four 32-k chunks per work-group between a cold start and a 64 KiB store);
// here the activations never leave the CU: X is staged ONCE as split bf16 planes into a [4 chunks][3 planes][128 rows][32 k] LDS image
// (96 KiB, the same swizzled rows as rowgemm128_b6_kernel), a layer reads its A fragments from the image and streams its weight planes
// through the usual two-buffer ring, and its output (bias, ReLU) is split and written back INTO the image for the next layer.
struct MlpChain {
  const __bf16* planes[3];  // wsplit128 planes of each layer (K = 128)
  const float* bias[3];     // layer 0: vector, or table rows of 128 selected by bias_idx0[row] / row / bias_div0; layers 1, 2: vectors
  const int64_t* bias_idx0;
  int bias_div0;
  int nlayers;  // 2 or 3
  int n_out;    // columns of the last layer that exist
};
constexpr int kChainLdsBytes = 2 * 3 * 128 * b6tile::BK * 2 + 4 * 3 * 128 * b6tile::BK * 2;  // 49 152 + 98 304
// up to three chains over the SAME input rows in one launch (blockIdx.y = chain: the denoiser's three heads - at one patch three
// work-groups side by side instead of three latency-bound launches one after the other)
struct MlpChainSet {
  MlpChain c[3];
  float* Y[3];
  int ldy[3];
};


namespace chaintile {
using b6tile::BK;
using b6tile::b6_off;
using b6tile::split3;
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CHAIN_FENCE() asm volatile("" ::: "memory")

// cl: kChainLdsBytes of LDS, 16-byte aligned; 512 threads; rows tile_m * 128 .. of X[M x 128] (ldx floats apart) -> Y[M x n_out]
__device__ __forceinline__ void mlp_chain_tile(__bf16* cl, const int tid, const int tile_m, const float* __restrict__ X, int ldx,
                                               const __bf16* pl0, const __bf16* pl1, const __bf16* pl2, const float* bs0, const float* bs1,
                                               const float* bs2, const int64_t* bias_idx0, int bias_div0, int nlayers, int n_out,
                                               float* __restrict__ Y, int ldy, int M) {
  struct { const int64_t* bias_idx0; int bias_div0, nlayers, n_out; } ch{bias_idx0, bias_div0, nlayers, n_out};
  auto planes_of = [&](int L) { return L == 0 ? pl0 : L == 1 ? pl1 : pl2; };
  auto bias_of = [&](int L) { return L == 0 ? bs0 : L == 1 ? bs1 : bs2; };
  __bf16* Ws = cl;                        // [2][3][128][32]
  __bf16* img = cl + 2 * 3 * 128 * BK;    // [4][3][128][32]
  const int lane = tid & 63, wv = tid >> 6;
  const int l31 = lane & 31, hk = lane >> 5, rw = wv & 3, cw = wv >> 2;  // wave tile 32 rows x 64 columns (32x32x16 MFMA)
  const int m0 = tile_m * 128;
  typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
  // ---- X -> image: thread (rows tid / 8 and 64 + tid / 8, 16-byte part tid % 8) of each of the four 32-k chunks
  {
    const int xa_row = tid >> 3, xa_part = tid & 7;
    f32x4 xr[4][2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      int row = m0 + xa_row + 64 * j;
      row = row < M ? row : M - 1;  // clamped (never stored)
      const float* src = X + static_cast<int64_t>(row) * ldx + 4 * xa_part;
#pragma unroll
      for (int c = 0; c < 4; ++c) xr[c][j] = *reinterpret_cast<const f32x4*>(src + c * BK);
    }
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        bf16x4 h, m, l;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          __bf16 hh, mm, ll;
          split3(xr[c][j][e], hh, mm, ll);
          h[e] = hh; m[e] = mm; l[e] = ll;
        }
        __bf16* dst = img + c * (3 * 128 * BK) + b6_off(xa_row + 64 * j, xa_part >> 1) + 4 * (xa_part & 1);
        *reinterpret_cast<bf16x4*>(dst) = h;
        *reinterpret_cast<bf16x4*>(dst + 128 * BK) = m;
        *reinterpret_cast<bf16x4*>(dst + 2 * 128 * BK) = l;
      }
  }
  // weight staging: a chunk is 1536 16-byte pieces = 3 per thread (plane p = pass, row tid / 4, part tid % 4)
  int w_dst[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) w_dst[i] = (i * 128) * BK + b6_off(tid >> 2, tid & 3);
  f32x4 wreg[3];
  auto load_w = [&](const __bf16* Wc, int c) {
    const __bf16* src = Wc + static_cast<size_t>(c < 4 ? c : 3) * (3 * 128 * BK) + tid * 8;
#pragma unroll
    for (int i = 0; i < 3; ++i) wreg[i] = *reinterpret_cast<const f32x4*>(src + 512 * 8 * i);
  };
  auto store_w = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 3; ++i) *reinterpret_cast<f32x4*>(Ws + buf * (3 * 128 * BK) + w_dst[i]) = wreg[i];
  };
  const int fx = (l31 >> 2) & 3;
  const int a_off = (32 * rw + l31) * BK, w_off = (64 * cw + l31) * BK;
  constexpr int TA[6] = {1, 2, 0, 1, 0, 0}, TB[6] = {1, 0, 2, 0, 1, 0};  // (mid,mid) (lo,hi) (hi,lo) (mid,hi) (hi,mid) (hi,hi)
  load_w(pl0, 0);
  for (int L = 0; L < ch.nlayers; ++L) {
    const __bf16* Wc = planes_of(L);
    const bool last = L == ch.nlayers - 1;
    f32x16 acc[2];
#pragma unroll
    for (int tt = 0; tt < 2; ++tt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[tt][r] = 0.f;
    store_w(0);
    load_w(Wc, 1);
    CHAIN_FENCE();
    __syncthreads();  // the image (X, or the previous layer's output) and the first weight chunk are in LDS
#pragma unroll 1  // (fully unrolled the kernel needs 260 VGPRs: 4 spilled)
    for (int c = 0; c < 4; ++c) {
      const int buf = c & 1;
      const __bf16* al = img + c * (3 * 128 * BK) + a_off;
      const __bf16* wl = Ws + buf * (3 * 128 * BK) + w_off;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const int so = 8 * ((2 * ks + hk) ^ fx);
        bf16x8 a[3], b[2][3];
#pragma unroll
        for (int p = 0; p < 3; ++p) {
          a[p] = *reinterpret_cast<const bf16x8*>(al + (p * 128) * BK + so);
#pragma unroll
          for (int tt = 0; tt < 2; ++tt) b[tt][p] = *reinterpret_cast<const bf16x8*>(wl + (p * 128 + 32 * tt) * BK + so);
        }
#pragma unroll
        for (int term = 0; term < 6; ++term)
#pragma unroll
          for (int tt = 0; tt < 2; ++tt)
            // (weights as the A operand: D is the TRANSPOSED tile - a lane ends with 16 columns of ONE row, four consecutive ones per
            // r >> 2, and the epilogue writes the next layer's input as 8-byte pieces instead of 2-byte ones)
            acc[tt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[tt][TB[term]], a[TA[term]], acc[tt], 0, 0, 0);
      }
      if (c < 3) {  // next chunk of this layer into the other buffer, then request the one after it (or the next layer's first chunk)
        store_w(buf ^ 1);
        if (c < 2) load_w(Wc, c + 2);
        else if (!last) load_w(planes_of(L + 1), 0);
        CHAIN_FENCE();
      }
      __syncthreads();
    }
    // ---- epilogue.  D^T 32x32: row = lane & 31 (+ 32 rw), column = (r & 3) + 8 (r >> 2) + 4 hk (+ 32 tt + 64 cw)
    const bool table = L == 0 && (ch.bias_idx0 != nullptr || ch.bias_div0 > 0);
    const int lrow = 32 * rw + l31, row = m0 + lrow;
    const float* bl = bias_of(L);
    if (table) {
      const int rc = row < M ? row : M - 1;
      const int64_t bi = ch.bias_idx0 ? ch.bias_idx0[rc] : rc / ch.bias_div0;
      bl = bs0 + bi * 128;
    }
#pragma unroll
    for (int tt = 0; tt < 2; ++tt)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const int col = 64 * cw + 32 * tt + 8 * g4 + 4 * hk;  // four consecutive columns col .. col + 3
        f32x4 bv = {0.f, 0.f, 0.f, 0.f};
        if (bl && (table || !last || col + 3 < ch.n_out)) bv = *reinterpret_cast<const f32x4*>(bl + col);
        else if (bl && col < ch.n_out) {  // the narrow last layer: the columns that exist
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (col + e < ch.n_out) bv[e] = bl[col + e];
        }
        float o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = acc[tt][4 * g4 + e] + bv[e];
        if (last) {
          if (row < M) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (col + e < ch.n_out) Y[static_cast<int64_t>(row) * ldy + col + e] = o[e];
          }
        } else {
          bf16x4 h, m, l;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            __bf16 hh, mm, ll;
            split3(fmaxf(o[e], 0.f), hh, mm, ll);  // every layer but the last is followed by a ReLU
            h[e] = hh; m[e] = mm; l[e] = ll;
          }
          // elements (row lrow, k = col .. col + 3) of the next layer's input: chunk col / 32, slot (col % 32) / 8, elements col % 8 ..
          // All reads of the image by this layer are behind the last barrier of the chunk loop.
          __bf16* dst = img + (col >> 5) * (3 * 128 * BK) + b6_off(lrow, (col & 31) >> 3) + (col & 7);
          *reinterpret_cast<bf16x4*>(dst) = h;
          *reinterpret_cast<bf16x4*>(dst + 128 * BK) = m;
          *reinterpret_cast<bf16x4*>(dst + 2 * 128 * BK) = l;
        }
      }
  }
}
#undef CHAIN_FENCE
}  // namespace chaintile
}  // namespace diffab
