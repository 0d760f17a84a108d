// proj_frames_b6_tile.h - the six IPA projections + local->global frames of one 128-row tile on the bf16 matrix cores (six-term split,
// gemm_bf16x6.hip) as a device function, so that more than one kernel can run it.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

#include "rowgemm_b6_tile.h"

namespace diffab {
namespace pjtile {
using b6tile::f32x4;
using b6tile::bf16x8;
using b6tile::split3;
#define PJTILE_FENCE() asm volatile("" ::: "memory")
constexpr int PJ_NP = 1344, PJ_GQ = 768, PJ_GK = 960, PJ_GV = 1152;  // column map of the projection buffer (ipa_attn_tile.h: ANP, OFF_*)
constexpr int PJ_B = 96, PJ_NB = PJ_NP / PJ_B, PJ_ROWS = 128;
constexpr int PJ_LD = 80;                                 // bf16 per staged row: 64 k + 16 pad (160 bytes)
constexpr int PJ_STAGE_ELEMS = 3 * PJ_B * 64;             // bf16 per stage in global memory (36 864 bytes)
constexpr int PJ_STAGE_LDS = 3 * PJ_B * PJ_LD;            // bf16 per stage in LDS (46 080 bytes)
constexpr int PJ_LDS_BYTES = 2 * PJ_STAGE_LDS * 2 + PJ_ROWS * 12 * 4;
struct __attribute__((packed, aligned(4))) pjb_f3 { float x, y, z; };

// Geometry at run time: N output columns in NB (even) blocks of 96 (columns past N are zero planes, never stored), rows of Y ldy
// floats apart, frames applied to blocks >= frames_from (NB: none; R, t may then be null).  The same kernel is the input-gradient
// product of to_out: dfeat[M x 1024] = dy[M x 128] Wo, with the planes of Wo^T (xsplit_kernel).
// PROJ: the geometry of the six projections at compile time (runtime geometry costs this kernel 11 %: 70 vs 63 us)
// SPLIT: the blocks are shared by gridDim.y work-groups per row tile (each re-reads the x rows and streams its share of the weights):
// twice the groups of half the length when 128-row tiles alone would leave CUs idle (B <= 128 patches of 128 residues per GPU)
// One 128-row tile (rows tile_m * 128 ..) as a device function: the body of proj_frames_b6_kernel (gemm_bf16x6.hip) and of the
// patch-resident module kernel (ipa_persistent.hip).  512 threads (tid: the thread index, a parameter so that a caller inside a loop
// can hand over an opaque copy); pj_lds: PJ_LDS_BYTES of LDS, 16-byte aligned; split_i of split_n work-groups share the column blocks.
template <bool FULL, bool PROJ, bool SPLIT = false>  // FULL: M is a multiple of 128, no row guards
__device__ __forceinline__ void proj_frames_b6_tile(__bf16* __restrict__ pj_lds, const int tid, const int tile_m, const int split_i,
                                                    const int split_n, const float* __restrict__ X, const __bf16* __restrict__ Wc,
                                                    const float* __restrict__ R, const float* __restrict__ t, float* __restrict__ Y,
                                                    int M, int N_, int NB_, int ldy_, int frames_from_) {
  const int N = PROJ ? PJ_NP : N_, NB = PROJ ? PJ_NB : NB_, ldy = PROJ ? PJ_NP : ldy_, frames_from = PROJ ? PJ_GQ / PJ_B : frames_from_;
  const int blk0 = SPLIT ? (NB * split_i) / split_n : 0;
  const int blk1 = SPLIT ? (NB * (split_i + 1)) / split_n : NB;
  // pj_lds: [2][3][96][PJ_LD] weights, then [128][12] frames (fp32)
  float* Rt = reinterpret_cast<float*>(pj_lds + 2 * PJ_STAGE_LDS);
  const int lane = tid & 63, wv = tid >> 6;
  const int l15 = lane & 15, g = lane >> 4, rw = wv & 3, cw = wv >> 2;
  const int m0 = tile_m * PJ_ROWS;
  float* ybase = Y + static_cast<int64_t>(m0 + 32 * rw + 4 * g) * ldy + 48 * cw + 3 * l15;
  const int col0 = 48 * cw + 3 * l15;  // first of this lane's three columns inside a block

  // weight staging: a stage is 2304 16-byte pieces, piece idx -> (plane idx / 768, row (idx % 768) / 8, part idx % 8); thread tid takes
  // idx = tid + 512 i (i = 0..3) and 2048 + (tid & 255) - the two halves of the work-group duplicate the last 256 (no branch)
  int st_src[5], st_dst[5];
#pragma unroll
  for (int i = 0; i < 5; ++i) {
    const int idx = i < 4 ? tid + 512 * i : 2048 + (tid & 255);
    const int pl = idx / 768, rem = idx % 768;
    st_src[i] = idx * 8;
    st_dst[i] = (pl * PJ_B + (rem >> 3)) * PJ_LD + (rem & 7) * 8;
  }
  f32x4 wreg[5];
  const int NSTAGE = 2 * blk1;
  auto load_w = [&](int stg) {
    stg = stg < NSTAGE ? stg : NSTAGE - 1;
    const __bf16* src = Wc + static_cast<size_t>(stg) * PJ_STAGE_ELEMS;
#pragma unroll
    for (int i = 0; i < 5; ++i) wreg[i] = *reinterpret_cast<const f32x4*>(src + st_src[i]);
  };
  auto store_w = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 5; ++i) *reinterpret_cast<f32x4*>(pj_lds + buf * PJ_STAGE_LDS + st_dst[i]) = wreg[i];
  };
  load_w(2 * blk0);
  // A fragments (v_mfma_f32_16x16x32_bf16: lane = row l15, k group g): a[mt][q][plane] = split(x[m0 + 32 rw + 16 mt + l15][32 q + 8 g .. + 7])
  bf16x8 a[2][4][3];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
    const int row = m0 + 32 * rw + 16 * mt + l15;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      f32x4 v0 = {0.f, 0.f, 0.f, 0.f}, v1 = v0;
      if (FULL || row < M) {
        const float* xp = X + static_cast<int64_t>(row) * 128 + 32 * q + 8 * g;
        v0 = *reinterpret_cast<const f32x4*>(xp);
        v1 = *reinterpret_cast<const f32x4*>(xp + 4);
      }
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        __bf16 hh, mm, ll;
        split3(c < 4 ? v0[c & 3] : v1[c & 3], hh, mm, ll);
        a[mt][q][0][c] = hh; a[mt][q][1][c] = mm; a[mt][q][2][c] = ll;
      }
    }
  }
  if (frames_from < NB)
    for (int idx = tid; idx < PJ_ROWS * 12; idx += 512) {
      const int row = idx / 12, cc = idx % 12, gr = m0 + row;
      float v = 0.0f;
      if (FULL || gr < M) v = cc < 9 ? R[static_cast<int64_t>(gr) * 9 + cc] : t[static_cast<int64_t>(gr) * 3 + (cc - 9)];
      Rt[idx] = v;
    }
  PJTILE_FENCE();
  store_w(0);
  load_w(2 * blk0 + 1);
  PJTILE_FENCE();
  __syncthreads();

  constexpr int TA[6] = {1, 2, 0, 1, 0, 0}, TB[6] = {1, 0, 2, 0, 1, 0};  // (mid,mid) (lo,hi) (hi,lo) (mid,hi) (hi,mid) (hi,hi)
  // one (mt, r) slice of a finished block: 3 consecutive columns of one row per lane
  auto epilogue_piece = [&](const f32x4 (&acc)[2][3], int blk, int piece) {
    const int mt = piece >> 2, r = piece & 3;
    const int lrow = 32 * rw + 16 * mt + 4 * g + r;
    float vx = acc[mt][0][r], vy = acc[mt][1][r], vz = acc[mt][2][r];
    if (blk >= frames_from) {  // point columns: local -> global frame
      const f32x4* F = reinterpret_cast<const f32x4*>(Rt + lrow * 12);
      const f32x4 f0 = F[0], f1 = F[1], f2 = F[2];  // R row-major 0..8, t 9..11
      const float ox = (vx * f0[0] + vy * f0[3] + vz * f1[2]) + f2[1];
      const float oy = (vx * f0[1] + vy * f1[0] + vz * f1[3]) + f2[2];
      const float oz = (vx * f0[2] + vy * f1[1] + vz * f2[0]) + f2[3];
      vx = ox; vy = oy; vz = oz;
    }
    if (FULL || m0 + lrow < M) {
      float* yp = ybase + (16 * mt + r) * ldy + PJ_B * blk;
      const int c = PJ_B * blk + col0;
      if (c + 2 < N) {
        pjb_f3 o{vx, vy, vz};
        *reinterpret_cast<pjb_f3*>(yp) = o;
      } else {  // the block that straddles N (N % 96 != 0)
        if (c < N) yp[0] = vx;
        if (c + 1 < N) yp[1] = vy;
      }
    }
  };
  // block `blk` into `cur`; the previous block's epilogue (`prev`) is issued between the MFMA groups of the first k half, so the
  // stores drain while the matrix pipe works (all waves storing at once in front of the barrier left it idle)
  auto run_block = [&](f32x4 (&cur)[2][3], const f32x4 (&prev)[2][3], int blk) {
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
      PJTILE_FENCE();
      const __bf16* wl = pj_lds + kh * PJ_STAGE_LDS + (48 * cw + l15) * PJ_LD + 8 * g;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        bf16x8 b[3][3];
        constexpr int PORD[3] = {1, 0, 2};  // planes in the order the terms need them (mid, hi, lo): the first MFMAs wait for 3 reads, not 9
#pragma unroll
        for (int pi = 0; pi < 3; ++pi)
#pragma unroll
          for (int tt = 0; tt < 3; ++tt)
            b[tt][PORD[pi]] = *reinterpret_cast<const bf16x8*>(wl + (PORD[pi] * PJ_B + 16 * tt) * PJ_LD + 32 * ks);
#pragma unroll
        for (int term = 0; term < 6; ++term) {
#pragma unroll
          for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int tt = 0; tt < 3; ++tt)
              cur[mt][tt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[mt][2 * kh + ks][TA[term]], b[tt][TB[term]], cur[mt][tt], 0, 0, 0);
          if (kh == 0 && blk > blk0 && term >= 1 && term <= 4) {
            epilogue_piece(prev, blk - 1, 4 * ks + term - 1);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      }
      __syncthreads();
    }
  };
  f32x4 accA[2][3], accB[2][3];
  int blk = blk0;
  for (; blk + 1 < blk1; blk += 2) {
    run_block(accA, accB, blk);
    run_block(accB, accA, blk + 1);
  }
  if (SPLIT && blk < blk1) {  // odd share (the 14 projection blocks over two groups: 7 each)
    run_block(accA, accB, blk);
#pragma unroll
    for (int piece = 0; piece < 8; ++piece) epilogue_piece(accA, blk, piece);
  } else {
#pragma unroll
    for (int piece = 0; piece < 8; ++piece) epilogue_piece(accB, blk1 - 1, piece);
  }
}
#undef PJTILE_FENCE
}  // namespace pjtile
}  // namespace diffab
