// proj_planes.hip - the six IPA projections written as MFMA OPERANDS of the attention kernel's logits product.
//
// The attention kernel (denoiser_fast.hip, B6L form) computes, per head, the scalar logits q.k AND the point-distance logits as ONE
// split-precision dot product over 64 "slots" on the bf16 matrix cores.  With gq_p = t_i + a_p, gk_p = t_j + b_p (a, b = the
// projected points rotated into the global orientation, diffab_pytorch.py:324 without the translation), D = t_i - t_j and
// t' = t - patch centroid, the squared distances of diffab_pytorch.py:426-436 are
//     sum_p |gq_p - gk_p|^2 = 8 |D|^2                                         [direct differences, fp32 VALU, shared by the heads]
//                           + (beta_j + 2 t'_j . w_j)                          [per key and head: rides in a slot against a 1.0]
//                           - 2 (u_i . t'_j + t'_i . w_j + sum_p a_p . b_p)    [30-dim dot product]
//                           + terms of (i, h) only                              [dropped: softmax over j does not see them]
// with u = sum_p a_p, w = sum_p b_p, beta = sum_p |b_p|^2.  Unlike the plain expansion |gq|^2 + |gk|^2 - 2 gq.gk (1e-5 in the
// logits) nothing large cancels: the only O(|t|^2) term, |D|^2, is formed from direct differences.
// tools/expanded_logits_numerics.py: against the float64 oracle this form is as close as the fp32 oracle itself (1.1e-6 vs
// 9.7e-7 on the aa logits at the benchmark geometry, 1.6e-5 vs 1.4e-5 at a 150 A offset).
//
// Slots of one (residue, head), query side | key side (their products summed over the 64 slots give
// ds^-1/2 q.k + coef (sum_p|.|^2 - 8 |D|^2) up to the dropped row terms; coef = -gamma_h / 12, c2 = -2 coef):
//     k-step 0, slot d = 0..31:        ds^-1/2 q_s[d]                 | k_s[d]
//     k-step 1, lane quarter g = 0..3: c2 a_g (3), E1_g               | b_g (3), F1_g           E1 = (c2 u, 1)    F1 = (t', ck)
//                                      c2 a_{4+g} (3), E2_g           | b_{4+g} (3), F2_g       E2 = (c2 t', 0)   F2 = (w, 0)
// ck = coef (beta + 2 t'.w).  Every slot is split into three bf16 planes (hi / mid / lo: exact) and stored in the FRAGMENT ORDER
// of v_mfma_f32_16x16x32_bf16: [patch][head][16-residue tile][k-step][plane][lane = residue % 16 + 16 g][8 slots 8 g ..] - each
// 1 KiB block is one A (query side) or B (key side) operand, loaded by the attention kernel with linear 1 KiB wave loads and no LDS
// staging.  The producer gets that layout for free by running the MFMAs of these columns with the operands SWAPPED (weights as A,
// x as B): the accumulator then holds 4 consecutive slots of ONE residue per lane - the transpose of the usual output - which is
// the consumer's operand layout (the accumulator-as-next-operand identity of the CDNA programming guide, section 3).
//
// The value side (v_s, global value points) keeps the fp32 row-major layout of the projection buffer (phase 3 of the attention
// kernel reads it as before); those column blocks run in the usual orientation with the usual epilogue.
//
// Structure (x-stationary, as proj_frames_b6_kernel in gemm_bf16x6.hip): a work-group owns 128 rows of x, every wave keeps its
// 32 x 128 slab as split A fragments, the weights stream through a two-stage LDS ring in 96-row blocks: 16 blocks = 11 of
// query/key slot tiles + 5 of value columns.
// Reference: InvariantPointAttentionLayer.forward, diffab_pytorch.py:391-436.
#include <type_traits>

#include "common.h"
#include "denoiser_internal.h"

namespace diffab {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
#define MEM_FENCE() asm volatile("" ::: "memory")

namespace {
constexpr int PP_NB = 16, PP_QKB = 11;       // 96-row weight blocks: [0, 11) query/key slot tiles (swapped operands), [11, 16) value side
constexpr int PP_B = 96, PP_ROWS = 128;
constexpr int PP_STAGE_ELEMS = 3 * PP_B * 64;  // bf16 per stage = (block, k half): 3 planes x 96 rows x 64 k = 36 x 1 KiB, the same image
                                               // in global memory and in LDS (copied by LDS-DMA, no registers)
constexpr int PP_STAGE_BYTES = PP_STAGE_ELEMS * 2;
constexpr int PP_PIECES = PP_STAGE_BYTES / 1024;  // 36 wave-sized DMA pieces
constexpr int PP_RT = 16;                    // floats per row of the frame table: R (9), t (3), t - centroid (3), pad
constexpr int PP_XF = 2 * 4 * 64 * 8;         // bf16 per (row wave, plane) of the x fragments parked in LDS: [mt][q][lane][8]
constexpr int PP_LDS_BYTES = 2 * PP_STAGE_BYTES + PP_ROWS * PP_RT * 4 + 4 * 2 * PP_XF * 2;
// Rows are 128 bytes, unpadded (an LDS-DMA instruction writes 1 KiB contiguously: 8 whole rows); the 16-byte chunk c of row r sits
// at position c ^ ((r >> 1) & 7), applied by ppsplit_kernel when it writes the planes: the 16-row ds_read_b128 fragment reads
// (four 16-lane groups, each 16 rows at one or two chunk indices) then touch 16 different bank quads - conflict-free.
__host__ __device__ inline int pp_swz(int row, int chunk) { return chunk ^ ((row >> 1) & 7); }
constexpr int PP_NP = 1344, PP_VS = 512, PP_GV = 1152;  // column map of the fp32 projection buffer (denoiser_fast.hip: ANP, OFF_VS, OFF_GV)
constexpr int PP_VPAD0 = 256, PP_VPAD1 = 288;           // virtual value columns: [0, 256) v_s | [256, 288) padding | [288, 480) value points
struct __attribute__((packed, aligned(4))) pp_f3 { float x, y, z; };

__device__ __forceinline__ void split3(float x, __bf16& h, __bf16& m, __bf16& l) {
  h = static_cast<__bf16>(x);
  const float r = x - static_cast<float>(h);
  m = static_cast<__bf16>(r);
  l = static_cast<__bf16>(r - static_cast<float>(m));
}

// Which weight row sits at row L (0 .. 1535) of the staged operand.  mat: 0 wq_s, 1 wk_s, 2 wv_s, 3 wq_p, 4 wk_p, 5 wv_p, -1: zero row.
// A slot tile is 16 rows = 4 slots x 4 lane quarters; 8 slots = one 16-byte operand piece per lane.  Tiles 0 and 1 of a (block,
// column wave) form a PAIR (one piece, 16-byte stores); tile 2 is a SINGLE: half a piece of a scalar k-step (8-byte stores; the
// geometry k-step needs both of its tiles in one lane for the sums over the eight points, so it is always a pair).
// -> head side hs (0..7 query heads, 8..15 key heads), k-step ks, half (which 4 slots of the piece a single holds)
struct PPTile { int hs, ks, half; };  // hs < 0: spare tile (zero weights, nothing stored)
__host__ __device__ inline PPTile pp_tile_of(int blk, int cw, int tt) {
  if (tt == 2) {  // singles: blocks 0..9 give 20 half pieces = scalar k-steps of head sides 0..9
    if (blk >= 10) return PPTile{-1, 0, 0};
    return PPTile{(blk >> 1) * 2 + cw, 0, blk & 1};
  }
  const int n = blk * 2 + cw;  // pairs 0..21: geometry k-steps of head sides 0..15, then scalar k-steps of head sides 10..15
  if (n < 16) return PPTile{n, 1, tt};
  return PPTile{10 + (n - 16), 0, tt};
}
struct PPSrc { int mat, row; };
__host__ __device__ inline PPSrc pp_source(int L) {
  const int blk = L / PP_B, rem = L % PP_B, cw = rem / 48, r48 = rem % 48, tt = r48 / 16, rho = r48 % 16;
  if (blk < PP_QKB) {
    const PPTile ti = pp_tile_of(blk, cw, tt);
    if (ti.hs < 0) return PPSrc{-1, 0};
    const int g = rho >> 2, r = rho & 3, side = ti.hs >> 3, h = ti.hs & 7;
    if (ti.ks == 0) return PPSrc{side, h * 32 + 8 * g + 4 * ti.half + r};
    if (r == 3) return PPSrc{-1, 0};
    return PPSrc{3 + side, h * 24 + (4 * ti.half + g) * 3 + r};
  }
  const int vc = PP_B * (blk - PP_QKB) + 48 * cw + 3 * rho + tt;  // lane column rho holds virtual columns 3 rho + tt (tt = 0..2)
  if (vc < PP_VPAD0) return PPSrc{2, vc};
  if (vc < PP_VPAD1) return PPSrc{-1, 0};
  return PPSrc{5, vc - PP_VPAD1};
}
}  // namespace

// stage-ordered split weights: out[((blk * 2 + kh) * 3 + plane) * 96 + l][8 swz(l, kk / 8) + kk % 8], l = row of the block, k = 64 kh + kk
__global__ void ppsplit_kernel(const float* __restrict__ W0, const float* __restrict__ W1, const float* __restrict__ W2,
                               const float* __restrict__ W3, const float* __restrict__ W4, const float* __restrict__ W5,
                               __bf16* __restrict__ out) {
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;  // (staged row L, k)
  if (gid >= PP_NB * PP_B * 128) return;
  const int L = gid >> 7, k = gid & 127;
  const PPSrc s = pp_source(L);
  float v = 0.0f;
  if (s.mat >= 0) {
    const float* Wp = s.mat == 0 ? W0 : (s.mat == 1 ? W1 : (s.mat == 2 ? W2 : (s.mat == 3 ? W3 : (s.mat == 4 ? W4 : W5))));
    v = Wp[s.row * 128 + k];
  }
  __bf16 h, m, l;
  split3(v, h, m, l);
  const int blk = L / PP_B, lrow = L % PP_B, kh = k >> 6, kk = k & 63;
  const size_t base = (static_cast<size_t>(blk * 2 + kh) * 3 * PP_B + lrow) * 64 + 8 * pp_swz(lrow, kk >> 3) + (kk & 7);
  out[base] = h;
  out[base + PP_B * 64] = m;
  out[base + 2 * PP_B * 64] = l;
}

// cent[b] = mean_k t[b, k]: the origin of the t' slots (any point near the patch works: it cancels exactly in the logits, its only
// role is to keep |t'| - and with it the rounding of the t'.u and t'.w products - small)
__global__ void patch_centroid_kernel(const float* __restrict__ t, int K, float* __restrict__ cent) {
  const int b = blockIdx.x, lane = threadIdx.x;  // one wave per patch
  float sx = 0.f, sy = 0.f, sz = 0.f;
  for (int k = lane; k < K; k += 64) {
    const float* p = t + (static_cast<int64_t>(b) * K + k) * 3;
    sx += p[0]; sy += p[1]; sz += p[2];
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    sx += __shfl_xor(sx, o); sy += __shfl_xor(sy, o); sz += __shfl_xor(sz, o);
  }
  if (lane == 0) {
    const float inv = 1.0f / static_cast<float>(K);
    cent[b * 4 + 0] = sx * inv; cent[b * 4 + 1] = sy * inv; cent[b * 4 + 2] = sz * inv; cent[b * 4 + 3] = 0.0f;
  }
}

// qk: query-side operands, then (side_stride bf16 further) key-side operands, layout in the header comment; Y: fp32 projection
// buffer (only the v_s and value-point columns are written).  gridDim.y work-groups share the 16 blocks of a row tile (even shares).
template <bool FULL>
__global__ __launch_bounds__(512) void proj_planes_b6_kernel(const float* __restrict__ X, const __bf16* __restrict__ Wc,
                                                             const float* __restrict__ R, const float* __restrict__ t,
                                                             const float* __restrict__ cent, const float* __restrict__ gamma,
                                                             __bf16* __restrict__ qk, int64_t side_stride, float* __restrict__ Y, int M,
                                                             int K) {
  const int blk0 = (PP_NB * static_cast<int>(blockIdx.y)) / static_cast<int>(gridDim.y);
  const int blk1 = (PP_NB * static_cast<int>(blockIdx.y + 1)) / static_cast<int>(gridDim.y);  // both even (gridDim.y in {1, 2, 4, 8})
  extern __shared__ __attribute__((aligned(16))) __bf16 pp_lds[];  // [2][3][96][64] weights, then [128][16] frames (fp32)
  float* Rt = reinterpret_cast<float*>(pp_lds + 2 * PP_STAGE_ELEMS);
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, g = lane >> 4, rw = wv & 3, cw = wv >> 2;
  const int m0 = blockIdx.x * PP_ROWS;

  // Weight staging by LDS-DMA (global_load_lds_dwordx4: 1 KiB per wave instruction, straight into LDS): wave w copies pieces
  // w, w + 8, .. of a stage.  The instruction is written in assembly: hipcc then neither waits for it nor drains it at a barrier,
  // and its completion is counted by hand (s_waitcnt vmcnt(N) in front of the barrier that publishes the stage, N = the global
  // stores this wave has issued behind it - vmcnt retires in order).  No staging registers, no ds_write pass.
  const unsigned lds0 = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(reinterpret_cast<uintptr_t>(pp_lds)));
  const int NSTAGE = 2 * blk1;
  auto dma_stage = [&](int stg, int buf) {
    stg = stg < NSTAGE ? stg : NSTAGE - 1;
    const char* src = reinterpret_cast<const char*>(Wc) + static_cast<size_t>(stg) * PP_STAGE_BYTES + lane * 16;
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      const int pc = wv + 8 * i;
      if (pc < PP_PIECES) {
        const char* gsrc = src + pc * 1024;
        const unsigned dst = lds0 + buf * PP_STAGE_BYTES + pc * 1024;
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep)
                     : "v"(gsrc), "s"(dst)
                     : "memory");
      }
    }
  };
  dma_stage(2 * blk0, 0);
  // x fragments: split(x[m0 + 32 rw + 16 mt + l15][32 q + 8 g .. + 7]); the hi plane stays in registers (a[mt][q]), the mid and lo
  // planes are parked in LDS in fragment order (one copy per row wave, written by its cw = 0 wave; lane-linear 16-byte accesses)
  // and re-read per k-step: 48 VGPRs that the operand epilogues need
  __bf16* xf = reinterpret_cast<__bf16*>(Rt + PP_ROWS * PP_RT) + rw * (2 * PP_XF);  // [plane mid, lo][mt][q][lane][8]
  bf16x8 a[2][4];
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
      bf16x8 am, al;
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        __bf16 hh, mm, ll;
        split3(c < 4 ? v0[c & 3] : v1[c & 3], hh, mm, ll);
        a[mt][q][c] = hh; am[c] = mm; al[c] = ll;
      }
      if (cw == 0) {
        *reinterpret_cast<bf16x8*>(xf + ((mt * 4 + q) * 64 + lane) * 8) = am;
        *reinterpret_cast<bf16x8*>(xf + PP_XF + ((mt * 4 + q) * 64 + lane) * 8) = al;
      }
    }
  }
  for (int idx = tid; idx < PP_ROWS * PP_RT; idx += 512) {
    const int row = idx / PP_RT, cc = idx % PP_RT, gr = m0 + row;
    float v = 0.0f;
    if (FULL || gr < M) {
      if (cc < 9) v = R[static_cast<int64_t>(gr) * 9 + cc];
      else if (cc < 12) v = t[static_cast<int64_t>(gr) * 3 + (cc - 9)];
      else if (cc < 15) v = t[static_cast<int64_t>(gr) * 3 + (cc - 12)] - cent[(gr / K) * 4 + (cc - 12)];
    }
    Rt[idx] = v;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // stage 0 has landed (and the x loads above)
  __syncthreads();

  constexpr int TA[6] = {1, 2, 0, 1, 0, 0}, TB[6] = {1, 0, 2, 0, 1, 0};  // (mid,mid) (lo,hi) (hi,lo) (mid,hi) (hi,mid) (hi,hi)
  const int ntile = K >> 4;

  // destination of the operand piece of lane (l15, g): residue tile (mt), head side hs, k-step ks, plane 0 (planes 512 elements apart)
  auto piece_ptr = [&](int hs, int ks, int mt) -> __bf16* {
    const int row0 = m0 + 32 * rw + 16 * mt;  // first residue of this 16-row tile: one patch, one tile of it (K % 16 == 0)
    const int gt = row0 >> 4, bb = gt / ntile, tile = gt - bb * ntile, side = hs >> 3, h = hs & 7;
    return qk + side * side_stride + ((static_cast<int64_t>(bb * 8 + h) * ntile + tile) * 6 + ks * 3) * 512 + lane * 8;
  };
  // ---- scalar slots: 4 (a single tile) or 8 (a pair) of them, scaled, split into three planes, stored in fragment order
  auto emit_scalar4 = [&](const f32x4 v4, int hs, int half, int mt) {
    const float sc = hs < 8 ? 0.17677669529663687f : 1.0f;  // ds^-1/2 (diffab_pytorch.py:353) folded into the query side
    typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
    bf16x4 p0, p1, p2;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      __bf16 hh, mm, ll;
      split3(v4[i] * sc, hh, mm, ll);
      p0[i] = hh; p1[i] = mm; p2[i] = ll;
    }
    if (FULL || m0 + 32 * rw + 16 * mt < M) {
      __bf16* dst = piece_ptr(hs, 0, mt) + 4 * half;
      *reinterpret_cast<bf16x4*>(dst) = p0;
      *reinterpret_cast<bf16x4*>(dst + 512) = p1;
      *reinterpret_cast<bf16x4*>(dst + 1024) = p2;
    }
  };
  auto store_piece = [&](const float (&v)[8], int hs, int ks, int mt) {
    bf16x8 p0, p1, p2;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      __bf16 hh, mm, ll;
      split3(v[i], hh, mm, ll);
      p0[i] = hh; p1[i] = mm; p2[i] = ll;
    }
    if (FULL || m0 + 32 * rw + 16 * mt < M) {
      __bf16* dst = piece_ptr(hs, ks, mt);
      *reinterpret_cast<bf16x8*>(dst) = p0;
      *reinterpret_cast<bf16x8*>(dst + 512) = p1;
      *reinterpret_cast<bf16x8*>(dst + 1024) = p2;
    }
  };
  // ---- geometry slots of one residue and head side: lo4 = point g (x, y, z, -), hi4 = point 4 + g
  auto emit_geometry = [&](const f32x4 lo4, const f32x4 hi4, int hs, int mt) {
    const int lrow = 32 * rw + 16 * mt + l15;
    const float* F = Rt + lrow * PP_RT;  // R row-major 0..8, t 9..11, t' 12..14
    // local -> global ORIENTATION (p R, row-vector convention of diffab_pytorch.py:324; the translation lives in the t' slots)
    const float r0 = F[0], r1 = F[1], r2 = F[2], r3 = F[3], r4 = F[4], r5 = F[5], r6 = F[6], r7 = F[7], r8 = F[8];
    const float ax = lo4[0] * r0 + lo4[1] * r3 + lo4[2] * r6;
    const float ay = lo4[0] * r1 + lo4[1] * r4 + lo4[2] * r7;
    const float az = lo4[0] * r2 + lo4[1] * r5 + lo4[2] * r8;
    const float bx = hi4[0] * r0 + hi4[1] * r3 + hi4[2] * r6;
    const float by = hi4[0] * r1 + hi4[1] * r4 + hi4[2] * r7;
    const float bz = hi4[0] * r2 + hi4[1] * r5 + hi4[2] * r8;
    float sx = ax + bx, sy = ay + by, sz = az + bz;
    float nb = (ax * ax + ay * ay + az * az) + (bx * bx + by * by + bz * bz);
    // the four lane quarters of a residue hold its eight points: all-reduce over g
    sx += __shfl_xor(sx, 16); sy += __shfl_xor(sy, 16); sz += __shfl_xor(sz, 16); nb += __shfl_xor(nb, 16);
    sx += __shfl_xor(sx, 32); sy += __shfl_xor(sy, 32); sz += __shfl_xor(sz, 32); nb += __shfl_xor(nb, 32);
    const float tcx = F[12], tcy = F[13], tcz = F[14];
    const float gm = gamma[hs & 7];
    const float sg = g == 0 ? sx : (g == 1 ? sy : sz), tg = g == 0 ? tcx : (g == 1 ? tcy : tcz);
    float v[8];
    if (hs < 8) {
      const float c2 = gm * 0.16666666666666666f;  // -2 coef, coef = -1/2 (4.5 x 8)^-1/2 gamma_h  (diffab_pytorch.py:372, :431-436)
      v[0] = c2 * ax; v[1] = c2 * ay; v[2] = c2 * az; v[3] = g == 3 ? 1.0f : c2 * sg;
      v[4] = c2 * bx; v[5] = c2 * by; v[6] = c2 * bz; v[7] = g == 3 ? 0.0f : c2 * tg;
    } else {
      const float coef = -0.5f * 0.16666666666666666f * gm;
      const float ck = coef * (nb + 2.0f * (tcx * sx + tcy * sy + tcz * sz));
      v[0] = ax; v[1] = ay; v[2] = az; v[3] = g == 3 ? ck : tg;
      v[4] = bx; v[5] = by; v[6] = bz; v[7] = g == 3 ? 0.0f : sg;
    }
    store_piece(v, hs, 1, mt);
  };
  // ---- one (mt, r) slice of a finished value-side block: three consecutive virtual columns of one row per lane
  auto value_piece = [&](const f32x4 (&acc)[2][3], int blk, int piece) {
    const int mt = piece >> 2, r = piece & 3;
    const int lrow = 32 * rw + 16 * mt + 4 * g + r;
    const int vc = PP_B * (blk - PP_QKB) + 48 * cw + 3 * l15;
    float vx = acc[mt][0][r], vy = acc[mt][1][r], vz = acc[mt][2][r];
    if (!(FULL || m0 + lrow < M)) return;
    float* yrow = Y + static_cast<int64_t>(m0 + lrow) * PP_NP;
    if (vc >= PP_VPAD1) {  // a value point: local -> global frame (x R + t)
      const f32x4* F = reinterpret_cast<const f32x4*>(Rt + lrow * PP_RT);
      const f32x4 f0 = F[0], f1 = F[1], f2 = F[2];
      const float ox = (vx * f0[0] + vy * f0[3] + vz * f1[2]) + f2[1];
      const float oy = (vx * f0[1] + vy * f1[0] + vz * f1[3]) + f2[2];
      const float oz = (vx * f0[2] + vy * f1[1] + vz * f2[0]) + f2[3];
      *reinterpret_cast<pp_f3*>(yrow + PP_GV + (vc - PP_VPAD1)) = pp_f3{ox, oy, oz};
    } else if (vc + 2 < PP_VPAD0) {
      *reinterpret_cast<pp_f3*>(yrow + PP_VS + vc) = pp_f3{vx, vy, vz};
    } else {  // the triple that straddles the end of v_s (virtual columns 255 | 256 257) and the padding columns
      if (vc < PP_VPAD0) yrow[PP_VS + vc] = vx;
      if (vc + 1 < PP_VPAD0) yrow[PP_VS + vc + 1] = vy;
    }
  };
  // The epilogue of block pb (accumulators `prev`) in four parts, issued between the MFMA groups of the next block
  auto epilogue_part = [&](const f32x4 (&prev)[2][3], int pb, int part) {
    if (pb >= PP_QKB) {
      value_piece(prev, pb, 2 * part);
      value_piece(prev, pb, 2 * part + 1);
      return;
    }
    const int mt = part >> 1;
    if ((part & 1) == 0) {
      const PPTile ti = pp_tile_of(pb, cw, 0);
      if (ti.ks == 1) {
        emit_geometry(prev[mt][0], prev[mt][1], ti.hs, mt);
      } else {
        const float sc = ti.hs < 8 ? 0.17677669529663687f : 1.0f;
        float v[8];
#pragma unroll
        for (int i = 0; i < 4; ++i) { v[i] = prev[mt][0][i] * sc; v[4 + i] = prev[mt][1][i] * sc; }
        store_piece(v, ti.hs, 0, mt);
      }
    } else {
      const PPTile ti = pp_tile_of(pb, cw, 2);
      if (ti.hs >= 0) emit_scalar4(prev[mt][2], ti.hs, ti.half, mt);
    }
  };

  auto run_block = [&](f32x4 (&cur)[2][3], const f32x4 (&prev)[2][3], int blk, auto swapped_tag) {
    constexpr bool SWAPPED = decltype(swapped_tag)::value;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int tt = 0; tt < 3; ++tt) cur[mt][tt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kh = 0; kh < 2; ++kh) {
      // stage 2 blk + kh is in buffer kh; the next one goes to buffer kh ^ 1 (read during the previous stage, barrier since) - requested
      // before this stage issues any global store
      dma_stage(2 * blk + kh + 1, kh ^ 1);
      const __bf16* wl = pp_lds + kh * PP_STAGE_ELEMS + (48 * cw + l15) * 64;
      const int fz = l15 >> 1;  // = ((48 cw + 16 tt + l15) >> 1) & 7: the row's chunk swizzle
      // six (k-step, column tile) groups of 12 MFMAs; the weight fragments of the next group are read while this one runs
      bf16x8 bw[2][3];
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) bw[0][pl] = *reinterpret_cast<const bf16x8*>(wl + (pl * PP_B) * 64 + 8 * (g ^ fz));
      bf16x8 ax[2][3];  // the x fragments of this k-step: [mt][plane]
#pragma unroll
      for (int grp = 0; grp < 6; ++grp) {
        const int ks = grp / 3, tt = grp % 3;
        if (tt == 0) {
#pragma unroll
          for (int mt = 0; mt < 2; ++mt) {
            ax[mt][0] = a[mt][2 * kh + ks];
            ax[mt][1] = *reinterpret_cast<const bf16x8*>(xf + ((mt * 4 + 2 * kh + ks) * 64 + lane) * 8);
            ax[mt][2] = *reinterpret_cast<const bf16x8*>(xf + PP_XF + ((mt * 4 + 2 * kh + ks) * 64 + lane) * 8);
          }
        }
        if (grp + 1 < 6) {
          const int ks1 = (grp + 1) / 3, tt1 = (grp + 1) % 3;
#pragma unroll
          for (int pl = 0; pl < 3; ++pl)
            bw[(grp + 1) & 1][pl] = *reinterpret_cast<const bf16x8*>(wl + (pl * PP_B + 16 * tt1) * 64 + 8 * ((4 * ks1 + g) ^ fz));
        }
#pragma unroll
        for (int term = 0; term < 6; ++term)
#pragma unroll
          for (int mt = 0; mt < 2; ++mt) {
            if (SWAPPED)  // D[weight row 4 g + r][residue l15]: four consecutive slots of one residue per lane
              cur[mt][tt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bw[grp & 1][TB[term]], ax[mt][TA[term]], cur[mt][tt], 0, 0, 0);
            else          // D[residue 4 g + r][column l15]
              cur[mt][tt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ax[mt][TA[term]], bw[grp & 1][TB[term]], cur[mt][tt], 0, 0, 0);
          }
        if (kh == 0 && blk > blk0 && (grp == 0 || grp == 1 || grp == 3 || grp == 4)) {
          epilogue_part(prev, blk - 1, grp < 3 ? grp : grp - 1);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      // the next stage's DMA pieces of this wave have landed once at most the stores issued behind them are outstanding (a lower
      // bound of their number is safe: it only waits longer): 12 per slot block (6 for block 10), >= 8 per value block
      if (kh == 0 && FULL && blk > blk0) {
        const int pb = blk - 1;
        if (pb < 10) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        else if (pb == 10) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __syncthreads();
    }
  };
  f32x4 accA[2][3], accB[2][3];
  // blk0, blk1 are even: blocks run in (even -> accA, odd -> accB) pairs
  for (int blk = blk0; blk < blk1; blk += 2) {
    if (blk < PP_QKB - 1) {            // both blocks of the pair are slot blocks (0..9)
      run_block(accA, accB, blk, std::true_type{});
      run_block(accB, accA, blk + 1, std::true_type{});
    } else if (blk == PP_QKB - 1) {    // 10 (slots) | 11 (values)
      run_block(accA, accB, blk, std::true_type{});
      run_block(accB, accA, blk + 1, std::false_type{});
    } else {
      run_block(accA, accB, blk, std::false_type{});
      run_block(accB, accA, blk + 1, std::false_type{});
    }
  }
#pragma unroll
  for (int part = 0; part < 4; ++part) epilogue_part(accB, blk1 - 1, part);
}

size_t proj_planes_scratch_bytes() { return static_cast<size_t>(2 * PP_NB) * PP_STAGE_ELEMS * sizeof(__bf16); }
size_t proj_planes_operand_floats(int64_t rows) { return static_cast<size_t>(rows) * 2 * 8 * 64 * 3 / 2; }  // both sides, bf16 pairs per float

// W6 = {wq_s, wk_s, wv_s, wq_p, wk_p, wv_p} -> stage-ordered split planes of the 16 blocks (proj_planes_scratch_bytes() bytes)
int launch_ppsplit(const float* const* W6, void* planes, hipStream_t st) {
  DIFFAB_REQUIRE(planes && (reinterpret_cast<uintptr_t>(planes) & 15) == 0, DIFFAB_ERR_ARG, "ppsplit: bad operands");
  hipLaunchKernelGGL(ppsplit_kernel, dim3((PP_NB * PP_B * 128 + 255) / 256), dim3(256), 0, st, W6[0], W6[1], W6[2], W6[3], W6[4], W6[5],
                     static_cast<__bf16*>(planes));
  DIFFAB_LAUNCH_CHECK();
  return DIFFAB_OK;
}

int launch_patch_centroids(const float* t, int B, int K, float* cent, hipStream_t st) {
  hipLaunchKernelGGL(patch_centroid_kernel, dim3(B), dim3(64), 0, st, t, K, cent);
  DIFFAB_LAUNCH_CHECK();
  return DIFFAB_OK;
}

// the six projections of one IPA layer: query / key operand planes into qk (proj_planes_operand_floats(rows) floats, 16-byte
// aligned), v_s and the global value points into the fp32 projection buffer proj[rows x 1344]
int launch_proj_planes_b6(const float* x, const void* planes, const float* R, const float* t, const float* cent, const float* gamma,
                          void* qk, float* proj, int rows, int K, hipStream_t st) {
  DIFFAB_REQUIRE(planes && qk && (reinterpret_cast<uintptr_t>(planes) & 15) == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0 &&
                     (reinterpret_cast<uintptr_t>(qk) & 15) == 0 && rows >= 16 && K % 16 == 0 && rows % K == 0,
                 DIFFAB_ERR_ARG, "proj_planes_b6: unsupported operands");
  const int ntiles = (rows + PP_ROWS - 1) / PP_ROWS;
  int nsplit = 1;  // half the chip or less: the blocks of a row tile shared by 2, 4 or 8 work-groups (even shares)
  while (nsplit < 8 && ntiles * nsplit * 2 <= 256) nsplit *= 2;
  const dim3 grid(ntiles, nsplit);
  const int64_t side_stride = static_cast<int64_t>(rows) * 8 * 64 * 3;
  const __bf16* Wc = static_cast<const __bf16*>(planes);
#define PP_LAUNCH(FULL_)                                                                                                          \
  do {                                                                                                                            \
    DIFFAB_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(proj_planes_b6_kernel<FULL_>),                             \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, PP_LDS_BYTES));                              \
    hipLaunchKernelGGL((proj_planes_b6_kernel<FULL_>), grid, dim3(512), PP_LDS_BYTES, st, x, Wc, R, t, cent, gamma,               \
                       static_cast<__bf16*>(qk), side_stride, proj, rows, K);                                                     \
  } while (0)
  if (rows % PP_ROWS == 0) PP_LAUNCH(true);
  else PP_LAUNCH(false);
#undef PP_LAUNCH
  DIFFAB_LAUNCH_CHECK();
  return DIFFAB_OK;
}

}  // namespace diffab
