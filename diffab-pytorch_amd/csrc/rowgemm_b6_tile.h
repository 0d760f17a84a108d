// rowgemm_b6_tile.h - one ROWS x 128 output tile of Y = act(X W^T + b) on the bf16 matrix cores (six-term split, see gemm_bf16x6.hip).
// A device function so that more than one kernel can run it: rowgemm128_b6_kernel (one tile per work-group) and the queue-driven
// attention kernel of the EXPERIMENTAL build (to_out tiles behind the attention tiles of the same launch).
#pragma once
#include <hip/hip_runtime.h>

namespace diffab {
namespace b6tile {
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int BK = 32;  // k per weight chunk

__device__ __forceinline__ void split3(float x, __bf16& h, __bf16& m, __bf16& l) {
  h = static_cast<__bf16>(x);
  const float r = x - static_cast<float>(h);
  m = static_cast<__bf16>(r);
  l = static_cast<__bf16>(r - static_cast<float>(m));
}
__device__ __forceinline__ int b6_off(int row, int slot) { return row * BK + 8 * (slot ^ ((row >> 2) & 3)); }  // bf16 elements
template <int ROWS>
constexpr int lds_bytes() { return 2 * 3 * (128 + ROWS) * BK * 2; }

// Summation order over k (the same for every launch shape, so that a result does not depend on how many work-groups shared a row tile:
// the sampler's bitwise shard invariance): k is cut into PARTS of PART_CHUNKS chunks (128 k); a part is accumulated from zero on the
// matrix cores, the parts are added in ascending order on the VALU (0 + p0 + p1 + ...), the bias last.
constexpr int PART_CHUNKS = 4;

// b6_lds: lds_bytes<ROWS>() bytes of LDS, 16-byte aligned; ROWS * 4 threads (tid = threadIdx.x); tile_m: index of the row tile.
// PARTIAL: the work-group computes ONE part (chunks [c_begin, c_begin + PART_CHUNKS) of the Kd / 32) and stores the raw accumulators to
// part_out[ROWS][128] (no bias, no activation, rows past M included: they are never read); the caller adds the parts up.
template <bool RELU, int ROWS, bool PARTIAL = false>
__device__ __forceinline__ void rowgemm128_tile(__bf16* b6_lds, int tid, int tile_m, const float* __restrict__ X, int ldx, const __bf16* __restrict__ Wc,
                                                const float* __restrict__ bias, const int64_t* __restrict__ bias_idx, int bias_div,
                                                float* __restrict__ Y, int ldy, int M, int Kd, int c_begin = 0,
                                                float* __restrict__ part_out = nullptr) {
#define B6TILE_FENCE() asm volatile("" ::: "memory")

  // bias: one vector (bias_idx == nullptr, bias_div == 0), or a table of 128-wide rows indexed by bias_idx[row] or row / bias_div
  constexpr int T = ROWS * 4, NRW = ROWS / 32;  // threads; row waves
  __bf16* Ws = b6_lds;
  __bf16* As = b6_lds + 2 * 3 * 128 * BK;
  const int lane = tid & 63, wv = tid >> 6;  // tid = threadIdx.x (a parameter so that a caller inside a loop can pass an opaque copy)
  const int l31 = lane & 31, hk = lane >> 5, rw = wv % NRW, cw = wv / NRW;  // v_mfma_f32_32x32x16_bf16: wave tile 32 rows x 64 columns
  const int m0 = tile_m * ROWS;
  const int nchunk = PARTIAL ? min(Kd / BK, c_begin + PART_CHUNKS) : Kd / BK;  // (one past the last chunk of this work-group)
  // weight staging: a chunk is 3 planes x 128 rows x 64 bytes = 1536 16-byte pieces, a straight copy of 24 KiB
  constexpr int WP = 1536 / T;  // pieces per thread (3 or 6)
  int w_dst[WP];
#pragma unroll
  for (int i = 0; i < WP; ++i) {
    const int idx = tid + T * i, p = idx / 512, row = (idx % 512) >> 2, part = idx & 3;
    w_dst[i] = (p * 128) * BK + b6_off(row, part);
  }
  f32x4 wreg[2][WP];
  auto load_w = [&](int slot, int ch) {
    ch = ch < nchunk ? ch : nchunk - 1;  // unconditional prefetch (a branch around it makes the compiler wait for it at once)
    const __bf16* src = Wc + static_cast<size_t>(ch) * (3 * 128 * BK) + tid * 8;
#pragma unroll
    for (int i = 0; i < WP; ++i) wreg[slot][i] = *reinterpret_cast<const f32x4*>(src + T * 8 * i);
  };
  auto store_w = [&](int slot, int buf) {
#pragma unroll
    for (int i = 0; i < WP; ++i) *reinterpret_cast<f32x4*>(Ws + buf * (3 * 128 * BK) + w_dst[i]) = wreg[slot][i];
  };
  // X staging: a chunk is ROWS rows x 128 bytes; thread -> rows tid / 8 and ROWS / 2 + tid / 8, 16-byte part tid % 8 (8 lanes = one
  // line); rows past M are clamped (their results are never stored).  Requested four chunks ahead (HBM), ring slots are compile-time.
  const int xa_row = tid >> 3, xa_part = tid & 7;
  const float* xsrc[2];
  int x_dst[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int lrow = xa_row + (ROWS / 2) * j;
    int row = m0 + lrow;
    row = row < M ? row : M - 1;
    xsrc[j] = X + static_cast<int64_t>(row) * ldx + 4 * xa_part;
    x_dst[j] = b6_off(lrow, xa_part >> 1) + 4 * (xa_part & 1);
  }
  f32x4 xreg[4][2];
  auto load_x = [&](int slot, int ch) {
    ch = ch < nchunk ? ch : nchunk - 1;
#pragma unroll
    for (int j = 0; j < 2; ++j) xreg[slot][j] = *reinterpret_cast<const f32x4*>(xsrc[j] + ch * BK);
  };
  typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
  auto store_x = [&](int slot, int buf) {  // fp32 -> three bf16 planes -> LDS
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      bf16x4 h, m, l;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        __bf16 hh, mm, ll;
        split3(xreg[slot][j][c], hh, mm, ll);
        h[c] = hh; m[c] = mm; l[c] = ll;
      }
      __bf16* dst = As + buf * (3 * ROWS * BK) + x_dst[j];
      *reinterpret_cast<bf16x4*>(dst) = h;
      *reinterpret_cast<bf16x4*>(dst + ROWS * BK) = m;
      *reinterpret_cast<bf16x4*>(dst + 2 * ROWS * BK) = l;
    }
  };
  f32x16 acc[2], tot[2];  // wave tile: 32 rows x 64 columns = two 32 x 32 accumulators (the running part | the parts before it)
#pragma unroll
  for (int tt = 0; tt < 2; ++tt)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[tt][r] = tot[tt][r] = 0.f;

#pragma unroll
  for (int c = 0; c < 4; ++c) load_x(c, c_begin + c);
  load_w(0, c_begin);
  load_w(1, c_begin + 1);
  B6TILE_FENCE();
  store_w(0, 0);
  store_x(0, 0);
  load_w(0, c_begin + 2);  // slot s holds chunk c with c % 2 == s: chunk 0 is staged, its slot takes chunk 2
  load_x(0, c_begin + 4);
  B6TILE_FENCE();
  __syncthreads();
  // fragments: lane (row or column l31, k half hk) of k-step ks reads the 16-byte slot 2 ks + hk of its row
  const int fx = (l31 >> 2) & 3;
  const int a_off = (32 * rw + l31) * BK, w_off = (64 * cw + l31) * BK;
  static_assert(PART_CHUNKS == 4, "a trip of the chunk loop is one part");
  for (int ch0 = c_begin; ch0 < nchunk; ch0 += 4) {  // ring slots are compile-time indices: four chunks per trip, the tail guarded (uniform)
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int ch = ch0 + u, buf = u & 1;
      if (ch >= nchunk) break;
      const __bf16* al = As + buf * (3 * ROWS * BK) + a_off;
      const __bf16* wl = Ws + buf * (3 * 128 * BK) + w_off;
      constexpr int TA[6] = {1, 2, 0, 1, 0, 0}, TB[6] = {1, 0, 2, 0, 1, 0};  // (mid,mid) (lo,hi) (hi,lo) (mid,hi) (hi,mid) (hi,hi): smallest first
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const int so = 8 * ((2 * ks + hk) ^ fx);
        bf16x8 a[3], b[2][3];
#pragma unroll
        for (int p = 0; p < 3; ++p) {
          a[p] = *reinterpret_cast<const bf16x8*>(al + (p * ROWS) * BK + so);
#pragma unroll
          for (int tt = 0; tt < 2; ++tt) b[tt][p] = *reinterpret_cast<const bf16x8*>(wl + (p * 128 + 32 * tt) * BK + so);
        }
#pragma unroll
        for (int term = 0; term < 6; ++term)
#pragma unroll
          for (int tt = 0; tt < 2; ++tt)
            acc[tt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[TA[term]], b[tt][TB[term]], acc[tt], 0, 0, 0);
      }
      // stage chunk ch + 1 into the other buffers (its loads were issued two / four iterations ago), then refill the ring slots
      store_w((u + 1) & 1, buf ^ 1);
      store_x((u + 1) & 3, buf ^ 1);
      load_w((u + 1) & 1, ch + 3);
      load_x((u + 1) & 3, ch + 5);
      B6TILE_FENCE();
      __syncthreads();
    }
    if (!PARTIAL) {  // the part is complete: onto the sum of the parts before it
#pragma unroll
      for (int tt = 0; tt < 2; ++tt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          tot[tt][r] += acc[tt][r];
          acc[tt][r] = 0.f;
        }
    }
  }
  if (PARTIAL) {
#pragma unroll
    for (int tt = 0; tt < 2; ++tt)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        part_out[(32 * rw + (r & 3) + 8 * (r >> 2) + 4 * hk) * 128 + 64 * cw + 32 * tt + l31] = acc[tt][r];
    return;
  }
  // D 32x32: column = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5); 32 lanes = 128 contiguous bytes of a row
  const bool table = bias_idx != nullptr || bias_div > 0;
#pragma unroll
  for (int tt = 0; tt < 2; ++tt) {
    const int col = 64 * cw + 32 * tt + l31;
    float bv = (bias && !table) ? bias[col] : 0.0f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = m0 + 32 * rw + (r & 3) + 8 * (r >> 2) + 4 * hk;
      if (row >= M) continue;
      if (table) {
        const int64_t bi = bias_idx ? bias_idx[row] : row / bias_div;
        bv = bias[bi * 128 + col];
      }
      float o = tot[tt][r] + bv;
      if (RELU) o = fmaxf(o, 0.f);
      Y[static_cast<int64_t>(row) * ldy + col] = o;
    }
  }
#undef B6TILE_FENCE
}
}  // namespace b6tile
}  // namespace diffab
