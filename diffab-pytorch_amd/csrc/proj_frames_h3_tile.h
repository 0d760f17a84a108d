// proj_frames_h3_tile.h - the six IPA projections + local->global frames of one 128-row tile on the f16 matrix cores as a three-term
// split product (rowgemm_h3_tile.h: two fp16 pieces per operand under a power-of-two scale, h2 w1 + h1 w2 + h1 w1, fp32 accumulation).
// Same decomposition as proj_frames_b6_tile.h - x-stationary (a wave keeps its 32 x 128 slab of x as split A fragments: 64 VGPRs
// instead of 96), 14 blocks of 96 output columns, the MFMA column index permuted so that a lane ends with one (x, y, z) point - with
// half the matrix-pipe work and two thirds of the weight staging.  Scales: x one per ROW (the whole row is in the wave's registers:
// a maximum over a lane's 32 values and two shuffles), the weights one per output column (pjsplit_h3); both are undone exactly in the
// epilogue, in front of the frame transform.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

#include "rowgemm_h3_tile.h"

namespace diffab {
namespace pjh3 {
using h3tile::f16x8;
using h3tile::f32x4;
using h3tile::h3_scale;
using h3tile::split2;
#define PJH3_FENCE() asm volatile("" ::: "memory")
constexpr int PJ_NP = 1344, PJ_GQ = 768;  // column map of the projection buffer (ipa_attn_tile.h: ANP, OFF_GQ)
constexpr int PJ_B = 96, PJ_NB = PJ_NP / PJ_B, PJ_ROWS = 128;
constexpr int PJ_LD = 80;                       // fp16 per staged row: 64 k + 16 pad (160 bytes)
constexpr int PJ_STAGE_ELEMS = 2 * PJ_B * 64;   // fp16 per stage in global memory (24 576 bytes)
constexpr int PJ_STAGE_LDS = 2 * PJ_B * PJ_LD;  // fp16 per stage in LDS (30 720 bytes)
constexpr int PJ_LDS_BYTES = 2 * PJ_STAGE_LDS * 2 + PJ_ROWS * 12 * 4 + PJ_ROWS * 4;  // two stages | frames [128][12] | 1 / row scale [128]
struct __attribute__((packed, aligned(4))) pj_f3 { float x, y, z; };
// a * b as ONE multiply that is never contracted into an fma (the epilogue's operation order is part of the shard invariance).  NOT inline
// assembly (rounds 5-6 spelled it as a v_mul_f32 assembly statement against the packed-op hazard of profiles/r05_pk_opsel_hazard.md): the compiler
// does not guard an assembly statement's READ of a matrix-core result - the ragged x-stationary instantiation scheduled the first
// epilogue slice three instructions behind the MFMA that writes its z accumulator and stored a stale register (round 6,
// test_x_stationary_backward_product_is_fp32_accurate).  The packed forms cannot appear: the library is built without the SLP
// vectoriser and tools/isa_hazard_lint.py checks every kernel.
__device__ __forceinline__ float mul1(float a, float b) {
#pragma clang fp contract(off)
  return a * b;
}

// One 128-row tile (rows tile_m * 128 ..).  512 threads; pj_lds: PJ_LDS_BYTES, 16-byte aligned; Wc: stage-ordered planes of pjsplit_h3
// ([(block 2 + k half)][2 planes][96][64] fp16), wis[1344]: 1 / scale of every output column; split_i of split_n work-groups share the
// column blocks of the tile (each re-reads the x rows).
// PROJ = false (the backward's d feat = d y W_out, gemm_f16x3.hip launch_xstat_h3): the same x-stationary product for ANY number of output
// columns - nb_rt blocks of 96 (planes and wis padded with zeros / ones), the first n_rt columns stored into rows of ldy_rt floats, no frames.
template <bool FULL, bool SPLIT = false, bool PROJ = true>  // FULL: M is a multiple of 128, no row guards
__device__ __forceinline__ void proj_frames_h3_tile(_Float16* __restrict__ pj_lds, const int tid, const int tile_m, const int split_i,
                                                    const int split_n, const float* __restrict__ X, const _Float16* __restrict__ Wc,
                                                    const float* __restrict__ wis, const float* __restrict__ R, const float* __restrict__ t,
                                                    float* __restrict__ Y, int M, const int nb_rt = 0, const int n_rt = 0, const int ldy_rt = 0) {
  const int NB = PROJ ? PJ_NB : nb_rt, ldy = PROJ ? PJ_NP : ldy_rt, frames_from = PROJ ? PJ_GQ / PJ_B : NB;
  const int blk0 = SPLIT ? (NB * split_i) / split_n : 0;
  const int blk1 = SPLIT ? (NB * (split_i + 1)) / split_n : NB;
  float* Rt = reinterpret_cast<float*>(pj_lds + 2 * PJ_STAGE_LDS);  // [128][12]
  float* Sx = Rt + PJ_ROWS * 12;                                    // [128]
  const int lane = tid & 63, wv = tid >> 6;
  const int l15 = lane & 15, g = lane >> 4, rw = wv & 3, cw = wv >> 2;
  const int m0 = tile_m * PJ_ROWS;
  float* ybase = Y + static_cast<int64_t>(m0 + 32 * rw + 4 * g) * ldy + 48 * cw + 3 * l15;
  const int col0 = 48 * cw + 3 * l15;  // first of this lane's three columns inside a block

  // weight staging: a stage is 1536 16-byte pieces, piece idx -> (plane idx / 768, row (idx % 768) / 8, part idx % 8): three per thread
  int st_dst[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int idx = tid + 512 * i, pl = idx / 768, rem = idx % 768;
    st_dst[i] = (pl * PJ_B + (rem >> 3)) * PJ_LD + (rem & 7) * 8;
  }
  f32x4 wreg[3];
  const int NSTAGE = 2 * blk1;
  auto load_w = [&](int stg) {
    stg = stg < NSTAGE ? stg : NSTAGE - 1;
    const _Float16* src = Wc + static_cast<size_t>(stg) * PJ_STAGE_ELEMS + tid * 8;
#pragma unroll
    for (int i = 0; i < 3; ++i) wreg[i] = *reinterpret_cast<const f32x4*>(src + 512 * 8 * i);
  };
  auto store_w = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 3; ++i) *reinterpret_cast<f32x4*>(pj_lds + buf * PJ_STAGE_LDS + st_dst[i]) = wreg[i];
  };
  load_w(2 * blk0);
  // A fragments (v_mfma_f32_16x16x32_f16: lane = row l15, k group g): a[mt][q][piece] of x[m0 + 32 rw + 16 mt + l15][32 q + 8 g .. + 7] s_row
  f16x8 a[2][4][2];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
    const int row = m0 + 32 * rw + 16 * mt + l15;
    f32x4 v[4][2];
    float m = 0.0f;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      v[q][0] = v[q][1] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (FULL || row < M) {
        const float* xp = X + static_cast<int64_t>(row) * 128 + 32 * q + 8 * g;
        v[q][0] = *reinterpret_cast<const f32x4*>(xp);
        v[q][1] = *reinterpret_cast<const f32x4*>(xp + 4);
      }
#pragma unroll
      for (int c = 0; c < 8; ++c) m = fmaxf(m, fabsf(v[q][c >> 2][c & 3]));
    }
    m = fmaxf(m, __shfl_xor(m, 16));  // the row's other three k groups
    m = fmaxf(m, __shfl_xor(m, 32));
    float s, is;
    h3_scale(m, s, is);
    if (g == 0) Sx[32 * rw + 16 * mt + l15] = is;
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        _Float16 h1, h2;
        split2(v[q][c >> 2][c & 3] * s, h1, h2);
        a[mt][q][0][c] = h1;
        a[mt][q][1][c] = h2;
      }
  }
  for (int idx = tid; PROJ && idx < PJ_ROWS * 12; idx += 512) {
    const int row = idx / 12, cc = idx % 12, gr = m0 + row;
    float v = 0.0f;
    if (FULL || gr < M) v = cc < 9 ? R[static_cast<int64_t>(gr) * 9 + cc] : t[static_cast<int64_t>(gr) * 3 + (cc - 9)];
    Rt[idx] = v;
  }
  PJH3_FENCE();
  store_w(0);
  load_w(2 * blk0 + 1);
  PJH3_FENCE();
  __syncthreads();

  // one (mt, r) slice of a finished block: 3 consecutive columns of one row per lane; wsc: 1 / scale of the lane's three columns
  auto epilogue_piece = [&](const f32x4 (&acc)[2][3], const pj_f3 wsc, int blk, int piece) {
    const int mt = piece >> 2, r = piece & 3;
    const int lrow = 32 * rw + 16 * mt + 4 * g + r;
    const float isx = Sx[lrow];
    // (scalar multiplies; a `v_pk_mul_f32 ... op_sel:[0,1]` here - what the SLP vectoriser made of `isx * wsc.xy` - came out 0 in lanes 48-63
    // with the MFMAs of the next block in flight: profiles/r05_pk_opsel_hazard.md; the library is built without that pass)
    const float vx = mul1(acc[mt][0][r], mul1(isx, wsc.x)), vy = mul1(acc[mt][1][r], mul1(isx, wsc.y)), vz0 = mul1(acc[mt][2][r], mul1(isx, wsc.z));
    float ox = vx, oy = vy, oz = vz0;
    if (PROJ && blk >= frames_from) {  // point columns: local -> global frame (diffab_pytorch.py:324)
      const f32x4* F = reinterpret_cast<const f32x4*>(Rt + lrow * 12);
      const f32x4 f0 = F[0], f1 = F[1], f2 = F[2];  // R row-major 0..8, t 9..11
      // (explicit operation order: left to the compiler's contraction the three products associate differently in the SPLIT and the
      // plain instantiation, and a patch's bits would depend on the batch it came in)
      const float vz = vz0;
      ox = __builtin_fmaf(vz, f1[2], __builtin_fmaf(vy, f0[3], mul1(vx, f0[0]))) + f2[1];
      oy = __builtin_fmaf(vz, f1[3], __builtin_fmaf(vy, f1[0], mul1(vx, f0[1]))) + f2[2];
      oz = __builtin_fmaf(vz, f2[0], __builtin_fmaf(vy, f1[1], mul1(vx, f0[2]))) + f2[3];
    }
    if (FULL || m0 + lrow < M) {
      float* yp = ybase + (16 * mt + r) * ldy + PJ_B * blk;
      if constexpr (PROJ) {
        pj_f3 o{ox, oy, oz};
        *reinterpret_cast<pj_f3*>(yp) = o;
      } else {  // (the last block may be partial)
        const int c = PJ_B * blk + col0;
        if (c < n_rt) yp[0] = ox;
        if (c + 1 < n_rt) yp[1] = oy;
        if (c + 2 < n_rt) yp[2] = oz;
      }
    }
  };
  // block `blk` into `cur`; the previous block's epilogue (`prev`, `wprev`) is issued between the MFMA groups of the first k half
  auto run_block = [&](f32x4 (&cur)[2][3], const f32x4 (&prev)[2][3], const pj_f3 wprev, int blk) {
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int tt = 0; tt < 3; ++tt) cur[mt][tt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kh = 0; kh < 2; ++kh) {
      // stage s = 2 blk + kh is in buffer kh; stage s + 1 (loaded during stage s - 1) goes to buffer kh ^ 1, then s + 2 is requested
      // - before this stage issues any global store (a wait for loads behind stores in flight degenerates to vmcnt(0))
      store_w(kh ^ 1);
      load_w(2 * blk + kh + 2);
      PJH3_FENCE();
      const _Float16* wl = pj_lds + kh * PJ_STAGE_LDS + (48 * cw + l15) * PJ_LD + 8 * g;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        f16x8 b[3][2];
#pragma unroll
        for (int pl = 0; pl < 2; ++pl)
#pragma unroll
          for (int tt = 0; tt < 3; ++tt) b[tt][pl] = *reinterpret_cast<const f16x8*>(wl + (pl * PJ_B + 16 * tt) * PJ_LD + 32 * ks);
        // (h2 w1) (h1 w2) (h1 w1): smallest first
#pragma unroll
        for (int term = 0; term < 3; ++term) {
          const int pa = term == 0 ? 1 : 0, pb = term == 1 ? 1 : 0;
#pragma unroll
          for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int tt = 0; tt < 3; ++tt)
              cur[mt][tt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[mt][2 * kh + ks][pa], b[tt][pb], cur[mt][tt], 0, 0, 0);
          if (kh == 0 && blk > blk0) {  // the previous block's eight epilogue slices: 1 + 1 + 2 per k step
            epilogue_piece(prev, wprev, blk - 1, 4 * ks + term);
            if (term == 2) epilogue_piece(prev, wprev, blk - 1, 4 * ks + 3);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      }
      __syncthreads();
    }
  };
  auto wsc_of = [&](int blk) { return *reinterpret_cast<const pj_f3*>(wis + PJ_B * blk + col0); };
  f32x4 accA[2][3], accB[2][3];
  pj_f3 wA = wsc_of(blk0), wB = wA;
  int blk = blk0;
  for (; blk + 1 < blk1; blk += 2) {
    run_block(accA, accB, wB, blk);
    wB = wsc_of(blk + 1);
    run_block(accB, accA, wA, blk + 1);
    if (blk + 2 < blk1) wA = wsc_of(blk + 2);
  }
  if ((SPLIT || !PROJ) && blk < blk1) {  // odd share (the 14 projection blocks over two groups: 7 each)
    run_block(accA, accB, wB, blk);
#pragma unroll
    for (int piece = 0; piece < 8; ++piece) epilogue_piece(accA, wA, blk, piece);
  } else {
#pragma unroll
    for (int piece = 0; piece < 8; ++piece) epilogue_piece(accB, wB, blk1 - 1, piece);
  }
}
#undef PJH3_FENCE
}  // namespace pjh3
}  // namespace diffab
