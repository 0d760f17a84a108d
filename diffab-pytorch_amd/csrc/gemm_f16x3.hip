// gemm_f16x3.hip - the two big dense products of an IPA layer (six projections + frames; to_out) on the f16 matrix cores as THREE-term
// split products (rowgemm_h3_tile.h, proj_frames_h3_tile.h): half the matrix-pipe work of the six-term bf16 form of gemm_bf16x6.hip
// at the same fp32 accuracy, bought with power-of-two scales (one per weight row, written with the planes; one per x row or per
// (x row, 32-k chunk), found while the operand is staged).  Reference: nn.Linear calls at diffab_pytorch.py:375-379, :391-413, :459-464.
#include "common.h"
#include "denoiser_internal.h"
#include "proj_frames_h3_tile.h"
#include "rowgemm_h3_tile.h"

namespace diffab {

using h3tile::BK;
using h3tile::h3_scale;
using h3tile::split2;

// W[128 x Kd] fp32 (rows ldw floats apart; rows n >= nrows are zero) -> two fp16 planes of W[n][:] s_n, chunk-major:
// out[((chunk * 2 + piece) * 128 + n) * 32 + kk], and wis[n] = 1 / s_n.  One work-group per row (its maximum first).
__global__ __launch_bounds__(256) void wsplit128_h3_kernel(const float* __restrict__ W, int ldw, int Kd, _Float16* __restrict__ out,
                                                          float* __restrict__ wis, int nrows) {
  const int n = blockIdx.x, tid = threadIdx.x;
  const float* row = W + static_cast<int64_t>(n) * ldw;
  float m = 0.0f;
  if (n < nrows)
    for (int k = tid; k < Kd; k += 256) m = fmaxf(m, fabsf(row[k]));
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) m = fmaxf(m, __shfl_xor(m, o));
  __shared__ float wm[4];
  if ((tid & 63) == 0) wm[tid >> 6] = m;
  __syncthreads();
  m = fmaxf(fmaxf(wm[0], wm[1]), fmaxf(wm[2], wm[3]));
  float s, is;
  h3_scale(m, s, is);
  if (tid == 0) wis[n] = is;
  for (int k = tid; k < Kd; k += 256) {
    _Float16 h1, h2;
    split2(n < nrows ? row[k] * s : 0.0f, h1, h2);
    const int chunk = k / BK, kk = k % BK;
    const size_t base = (static_cast<size_t>(chunk) * 2 * 128 + n) * BK + kk;
    out[base] = h1;
    out[base + 128 * BK] = h2;
  }
}

template <bool RELU, int ROWS>
__global__ __launch_bounds__(ROWS * 4) void rowgemm128_h3_kernel(const float* __restrict__ X, int ldx, const _Float16* __restrict__ Wc,
                                                                 const float* __restrict__ wis, const float* __restrict__ bias,
                                                                 const int64_t* __restrict__ bias_idx, int bias_div, float* __restrict__ Y,
                                                                 int ldy, int M, int Kd) {
  extern __shared__ __attribute__((aligned(16))) _Float16 h3_lds[];
  h3tile::rowgemm128_h3_tile<RELU, ROWS>(h3_lds, threadIdx.x, blockIdx.x, X, ldx, Wc, wis, bias, bias_idx, bias_div, Y, ldy, M, Kd);
}
// one work-group per (row tile, 64-k part); a second launch adds the parts in the tile's own order (bitwise the plain kernel)
template <int ROWS>
__global__ __launch_bounds__(ROWS * 4) void rowgemm128_h3_parts_kernel(const float* __restrict__ X, int ldx, const _Float16* __restrict__ Wc, int M,
                                                                       int Kd, float* __restrict__ parts) {
  extern __shared__ __attribute__((aligned(16))) _Float16 h3_lds[];
  float* mine = parts + (static_cast<size_t>(blockIdx.x) * gridDim.y + blockIdx.y) * (ROWS * 128);
  h3tile::rowgemm128_h3_tile<false, ROWS, true>(h3_lds, threadIdx.x, blockIdx.x, X, ldx, Wc, nullptr, nullptr, nullptr, 0, nullptr, 0, M, Kd,
                                                blockIdx.y * h3tile::PART_CHUNKS, mine);
}
template <bool RELU, int ROWS>
__global__ __launch_bounds__(256) void rowgemm128_h3_parts_sum_kernel(const float* __restrict__ parts, int nparts, const float* __restrict__ wis,
                                                                      const float* __restrict__ bias, const int64_t* __restrict__ bias_idx,
                                                                      int bias_div, float* __restrict__ Y, int ldy, int M) {
  const int gid = blockIdx.x * 256 + threadIdx.x;  // (row, 4 columns)
  const int row = gid >> 5, col = (gid & 31) * 4;
  if (row >= M) return;
  const float* src = parts + (static_cast<size_t>(row / ROWS) * nparts * ROWS + row % ROWS) * 128 + col;
  h3tile::f32x4 tot = {0.f, 0.f, 0.f, 0.f};
  for (int p = 0; p < nparts; ++p) {
    const h3tile::f32x4 v = *reinterpret_cast<const h3tile::f32x4*>(src + static_cast<size_t>(p) * (ROWS * 128));
#pragma unroll
    for (int c = 0; c < 4; ++c) tot[c] += v[c];
  }
  const bool table = bias_idx != nullptr || bias_div > 0;
  const float* brow = table ? bias + (bias_idx ? bias_idx[row] : row / bias_div) * 128 : bias;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    float o = __builtin_fmaf(tot[c], wis[col + c], brow ? brow[col + c] : 0.f);
    if (RELU) o = fmaxf(o, 0.f);
    Y[static_cast<int64_t>(row) * ldy + col + c] = o;
  }
}

size_t rowgemm128_h3_planes_bytes(int Kd) { return static_cast<size_t>(2) * 128 * Kd * sizeof(_Float16); }
bool rowgemm128_h3_ok(const float* X, int ldx, const float* Y, int ldy, int M, int Kd) {
  return Kd % BK == 0 && Kd >= BK && ldx % 4 == 0 && ldy % 4 == 0 && (reinterpret_cast<uintptr_t>(X) & 15) == 0 &&
         (reinterpret_cast<uintptr_t>(Y) & 15) == 0 && M >= 1;
}
// planes: rowgemm128_h3_planes_bytes(Kd) bytes, 16-byte aligned; wis: 128 floats
int launch_wsplit128_h3(const float* W, int ldw, int Kd, void* planes, float* wis, hipStream_t st, int nrows) {
  DIFFAB_REQUIRE(W && planes && wis && (reinterpret_cast<uintptr_t>(planes) & 15) == 0 && Kd % BK == 0 && nrows >= 1 && nrows <= 128, DIFFAB_ERR_ARG,
                 "wsplit128_h3: bad operands");
  hipLaunchKernelGGL(wsplit128_h3_kernel, dim3(128), dim3(256), 0, st, W, ldw, Kd, static_cast<_Float16*>(planes), wis, nrows);
  DIFFAB_LAUNCH_CHECK();
  return DIFFAB_OK;
}
// scratch of the (row tile, k part) form: 0 when M is past the few row tiles that form is for
size_t rowgemm128_h3_parts_floats(int M, int Kd) {
  const size_t tiles64 = static_cast<size_t>((M + 63) / 64);
  return tiles64 <= 64 ? tiles64 * ((Kd / BK + h3tile::PART_CHUNKS - 1) / h3tile::PART_CHUNKS) * 64 * 128 : 0;
}
// Y[M x 128] = act(X[:, 0:Kd] W[:, 0:Kd]^T + bias row), W as launch_wsplit128_h3 planes + scales; parts (optional): scratch of
// rowgemm128_h3_parts_floats(M, Kd) floats for the (row tile, k part) form of few row tiles
int launch_rowgemm128_h3p(const float* X, int ldx, const void* planes, const float* wis, const float* bias, const int64_t* bias_idx, int bias_div,
                          float* Y, int ldy, int M, int Kd, bool relu, hipStream_t st, float* parts) {
  DIFFAB_REQUIRE(rowgemm128_h3_ok(X, ldx, Y, ldy, M, Kd) && planes && wis && (reinterpret_cast<uintptr_t>(planes) & 15) == 0, DIFFAB_ERR_ARG,
                 "rowgemm128_h3: unsupported operands");
  const _Float16* Wc = static_cast<const _Float16*>(planes);
  const int tiles64 = (M + 63) / 64, nparts = (Kd / BK + h3tile::PART_CHUNKS - 1) / h3tile::PART_CHUNKS;
  if (parts && nparts > 1 && tiles64 <= 64 && (reinterpret_cast<uintptr_t>(parts) & 15) == 0) {  // up to 32 patches of 128 residues
    DIFFAB_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(rowgemm128_h3_parts_kernel<64>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                         h3tile::lds_bytes<64>()));
    hipLaunchKernelGGL((rowgemm128_h3_parts_kernel<64>), dim3(tiles64, nparts), dim3(256), h3tile::lds_bytes<64>(), st, X, ldx, Wc, M, Kd, parts);
    const dim3 sgrid((M * 32 + 255) / 256);
    if (relu) hipLaunchKernelGGL((rowgemm128_h3_parts_sum_kernel<true, 64>), sgrid, dim3(256), 0, st, parts, nparts, wis, bias, bias_idx, bias_div, Y, ldy, M);
    else hipLaunchKernelGGL((rowgemm128_h3_parts_sum_kernel<false, 64>), sgrid, dim3(256), 0, st, parts, nparts, wis, bias, bias_idx, bias_div, Y, ldy, M);
    DIFFAB_LAUNCH_CHECK();
    return DIFFAB_OK;
  }
  const int rows_wg = (M + 127) / 128 >= 256 ? 128 : 64;  // 128-row groups when they fill the chip, 64-row groups below that
#define H3_LAUNCH(RELU_, ROWS_)                                                                                                          \
  do {                                                                                                                                   \
    DIFFAB_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(rowgemm128_h3_kernel<RELU_, ROWS_>),                               \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, h3tile::lds_bytes<ROWS_>()));                      \
    hipLaunchKernelGGL((rowgemm128_h3_kernel<RELU_, ROWS_>), dim3((M + ROWS_ - 1) / ROWS_), dim3(ROWS_ * 4), h3tile::lds_bytes<ROWS_>(), st, \
                       X, ldx, Wc, wis, bias, bias_idx, bias_div, Y, ldy, M, Kd);                                                        \
  } while (0)
  if (rows_wg == 128) {
    if (relu) H3_LAUNCH(true, 128);
    else H3_LAUNCH(false, 128);
  } else {
    if (relu) H3_LAUNCH(true, 64);
    else H3_LAUNCH(false, 64);
  }
#undef H3_LAUNCH
  DIFFAB_LAUNCH_CHECK();
  return DIFFAB_OK;
}

// ================================================================== six IPA projections + local->global frames
// stage-ordered planes: out[((blk * 2 + kh) * 2 + piece) * 96 + l][kk], l = 48 cw + 16 tt + j <-> output column 96 blk + 48 cw + 3 j + tt,
// k = 64 kh + kk; wis[output column] = 1 / scale.  One wave per output column.
__global__ __launch_bounds__(64) void pjsplit_h3_kernel(const float* __restrict__ W0, const float* __restrict__ W1, const float* __restrict__ W2,
                                                        const float* __restrict__ W3, const float* __restrict__ W4, const float* __restrict__ W5,
                                                        _Float16* __restrict__ out, float* __restrict__ wis) {
  using namespace pjh3;
  const int gc = blockIdx.x, lane = threadIdx.x;
  const float* Wp;
  int row;
  if (gc < PJ_GQ) {
    Wp = gc < 256 ? W0 : (gc < 512 ? W1 : W2);
    row = gc & 255;
  } else {
    Wp = gc < 960 ? W3 : (gc < 1152 ? W4 : W5);
    row = gc - (gc < 960 ? PJ_GQ : (gc < 1152 ? 960 : 1152));
  }
  const float v0 = Wp[row * 128 + lane], v1 = Wp[row * 128 + 64 + lane];
  float m = fmaxf(fabsf(v0), fabsf(v1));
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) m = fmaxf(m, __shfl_xor(m, o));
  float s, is;
  h3_scale(m, s, is);
  if (lane == 0) wis[gc] = is;
  const int blk = gc / PJ_B, rem = gc % PJ_B, cwl = rem / 48, r48 = rem % 48, j = r48 / 3, tt = r48 % 3;
  const int lrow = 48 * cwl + 16 * tt + j;
#pragma unroll
  for (int kh = 0; kh < 2; ++kh) {
    _Float16 h1, h2;
    split2((kh == 0 ? v0 : v1) * s, h1, h2);
    const size_t base = (static_cast<size_t>(blk * 2 + kh) * 2 * PJ_B + lrow) * 64 + lane;
    out[base] = h1;
    out[base + PJ_B * 64] = h2;
  }
}

// d feat = d y W (x-stationary, any N): the projection tile without frames (proj_frames_h3_tile.h, PROJ = false)
template <bool FULL, bool SPLIT>
__global__ __launch_bounds__(512) void xstat_h3_kernel(const float* __restrict__ X, const _Float16* __restrict__ Wc, const float* __restrict__ wis,
                                                       float* __restrict__ Y, int M, int NB, int N, int ldy) {
  extern __shared__ __attribute__((aligned(16))) _Float16 pj_lds[];
  pjh3::proj_frames_h3_tile<FULL, SPLIT, false>(pj_lds, threadIdx.x, blockIdx.x, blockIdx.y, gridDim.y, X, Wc, wis, nullptr, nullptr, Y, M, NB, N, ldy);
}
// planes of a strided weight for xstat_h3_kernel: output column gc = W[gc sn + k sk], k < 128; columns N .. ncols - 1: zeros, wis 1.  The
// layout of pjsplit_h3_kernel; one wave per column.
__global__ __launch_bounds__(64) void xsplit_h3_kernel(const float* __restrict__ W, int64_t sn, int64_t sk, int N, _Float16* __restrict__ out,
                                                       float* __restrict__ wis) {
  using namespace pjh3;
  const int gc = blockIdx.x, lane = threadIdx.x;
  float v0 = 0.0f, v1 = 0.0f;
  if (gc < N) {
    v0 = W[gc * sn + lane * sk];
    v1 = W[gc * sn + (64 + lane) * sk];
  }
  float m = fmaxf(fabsf(v0), fabsf(v1));
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) m = fmaxf(m, __shfl_xor(m, o));
  float s, is;
  h3_scale(m, s, is);
  if (lane == 0) wis[gc] = is;
  const int blk = gc / PJ_B, rem = gc % PJ_B, cwl = rem / 48, r48 = rem % 48, j = r48 / 3, tt = r48 % 3;
  const int lrow = 48 * cwl + 16 * tt + j;
#pragma unroll
  for (int kh = 0; kh < 2; ++kh) {
    _Float16 h1, h2;
    split2((kh == 0 ? v0 : v1) * s, h1, h2);
    const size_t base = (static_cast<size_t>(blk * 2 + kh) * 2 * PJ_B + lrow) * 64 + lane;
    out[base] = h1;
    out[base + PJ_B * 64] = h2;
  }
}

template <bool FULL, bool SPLIT>
__global__ __launch_bounds__(512) void proj_frames_h3_kernel(const float* __restrict__ X, const _Float16* __restrict__ Wc, const float* __restrict__ wis,
                                                             const float* __restrict__ R, const float* __restrict__ t, float* __restrict__ Y, int M) {
  extern __shared__ __attribute__((aligned(16))) _Float16 pj_lds[];
  pjh3::proj_frames_h3_tile<FULL, SPLIT>(pj_lds, threadIdx.x, blockIdx.x, blockIdx.y, gridDim.y, X, Wc, wis, R, t, Y, M);
}

// operand planes of the attention tile (attn_planes_tile.h): floats of workspace for `rows` projection rows
size_t proj_value_planes_floats(int64_t rows) { return static_cast<size_t>(rows) * 512; }  // V: 8 heads x 4 tiles x 16 columns x 2 planes x 2 bytes per row
size_t proj_value_scales_floats(int64_t rows) { return (static_cast<size_t>(rows / 32 + 1) * 64 + 63) & ~static_cast<size_t>(63); }
size_t proj_frames_h3_planes_bytes() { return static_cast<size_t>(2 * pjh3::PJ_NB) * pjh3::PJ_STAGE_ELEMS * sizeof(_Float16); }
// W6 = {wq_s, wk_s, wv_s, wq_p, wk_p, wv_p} -> stage-ordered planes (proj_frames_h3_planes_bytes(), 16-byte aligned) + wis[1344]
int launch_pjsplit_h3(const float* const* W6, void* planes, float* wis, hipStream_t st) {
  DIFFAB_REQUIRE(planes && wis && (reinterpret_cast<uintptr_t>(planes) & 15) == 0, DIFFAB_ERR_ARG, "pjsplit_h3: bad operands");
  hipLaunchKernelGGL(pjsplit_h3_kernel, dim3(pjh3::PJ_NP), dim3(64), 0, st, W6[0], W6[1], W6[2], W6[3], W6[4], W6[5],
                     static_cast<_Float16*>(planes), wis);
  DIFFAB_LAUNCH_CHECK();
  return DIFFAB_OK;
}
// the six projections of one IPA layer (D = 128) into proj[rows x 1344]
int launch_proj_frames_h3p(const float* x, const void* planes, const float* wis, const float* R, const float* t, float* proj, int rows,
                           hipStream_t st) {
  using namespace pjh3;
  DIFFAB_REQUIRE(planes && wis && R && t && (reinterpret_cast<uintptr_t>(planes) & 15) == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0 &&
                     (reinterpret_cast<uintptr_t>(proj) & 3) == 0 && rows >= 1,
                 DIFFAB_ERR_ARG, "proj_frames_h3: unsupported operands");
  const _Float16* Wc = static_cast<const _Float16*>(planes);
  const int ntiles = (rows + PJ_ROWS - 1) / PJ_ROWS;
  int nsplit = 256 / ntiles;  // half the chip or less: several groups per row tile, each with its share of the column blocks
  nsplit = nsplit < 1 ? 1 : (nsplit > PJ_NB ? PJ_NB : nsplit);
  const dim3 grid(ntiles, nsplit);
#define PJH3_LAUNCH(FULL_, SPLIT_)                                                                                                    \
  do {                                                                                                                                \
    DIFFAB_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(proj_frames_h3_kernel<FULL_, SPLIT_>),                          \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, PJ_LDS_BYTES));                                   \
    hipLaunchKernelGGL((proj_frames_h3_kernel<FULL_, SPLIT_>), grid, dim3(512), PJ_LDS_BYTES, st, x, Wc, wis, R, t, proj, rows);        \
  } while (0)
  if (rows % PJ_ROWS == 0) {
    if (nsplit > 1) PJH3_LAUNCH(true, true);
    else PJH3_LAUNCH(true, false);
  } else {
    if (nsplit > 1) PJH3_LAUNCH(false, true);
    else PJH3_LAUNCH(false, false);
  }
#undef PJH3_LAUNCH
  DIFFAB_LAUNCH_CHECK();
  return DIFFAB_OK;
}

// ================================================================== weight-gradient products: C[N1 x N2] += A^T B  (fp16 x 3)
// The three-term form of gemm_tn_b6_kernel (gemm_bf16x6.hip: same tiling - 128 x 128 output tile per work-group, 32-row slabs of both
// operands staged in their natural [row][column] orientation, fragments through ds_read_b64_tr_b16, M split over work-groups, fp32
// atomics): two fp16 planes per operand instead of three bf16 ones, 24 MFMAs per wave and slab instead of 48.  fp16 has no range to
// spare, gradients are small and uneven: every (slab, operand) gets ONE power-of-two scale from the largest magnitude of the work-group's
// 32 x 128 block (the contraction runs over the slab's rows, so the scale may not vary inside it).  The maxima travel ahead of the
// data: a slab sits in the register ring for three steps before it is staged - each wave posts its maximum of slab s + 2 to LDS during
// step s, the step's barrier publishes it, step s + 1 reads the eight wave maxima when it splits that slab.  A slab's products are
// accumulated on their own and added to the running sums with 1 / (s_A s_B).
namespace tnh3 {
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
using h3tile::f16x8;
using h3tile::f32x4;
constexpr int TN_PLANE = 32 * 128;  // fp16 elements of one plane of one slab
constexpr int TN_LDS_BYTES = 2 * 4 * TN_PLANE * 2 + 4 * 16 * 4;  // [2 buffers][A | B][2 planes] | wave maxima [4 ring slots][A | B][8 waves]
__device__ __forceinline__ int tn_off(int row, int chunk) {  // element offset of 16-byte chunk `chunk` of row `row` (gemm_bf16x6.hip tn_off)
  return row * 128 + 8 * (chunk ^ (((row & 3) << 2) | ((row >> 2) & 3)));
}
// maximum of a non-negative value over the wave (non-negative floats order like their bit patterns): the two row-swap instructions of
// gfx950 for lanes 16 / 32 apart, rotations inside the rows of 16 - vector ALU only
__device__ __forceinline__ float wave_max_nonneg(float m) {
  unsigned u = __float_as_uint(m);
  const auto s16 = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  u = s16[0] > s16[1] ? s16[0] : s16[1];
  const auto s32 = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  u = s32[0] > s32[1] ? s32[0] : s32[1];
#define TNH3_ROR(N_)                                                                                      \
  {                                                                                                       \
    const unsigned o = static_cast<unsigned>(__builtin_amdgcn_update_dpp(0, static_cast<int>(u), 0x120 + N_, 0xf, 0xf, false)); \
    u = o > u ? o : u;                                                                                    \
  }
  TNH3_ROR(8) TNH3_ROR(4) TNH3_ROR(2) TNH3_ROR(1)
#undef TNH3_ROR
  return __uint_as_float(u);
}
}  // namespace tnh3

struct TnSegsH3 {
  float* p[6];
  int n_end[6];
  int nseg;  // 0: one plain matrix C
};

// A_VEC / B_VEC, db, segs: as gemm_tn_b6_kernel
template <bool A_VEC, bool B_VEC>
__global__ __launch_bounds__(512) void gemm_tn_h3_kernel(const float* __restrict__ A, int lda, const float* __restrict__ Bm, int ldb,
                                                         float* __restrict__ C, int ldc, int M, int m_chunk, int N1, int N2, TnSegsH3 segs,
                                                         float* __restrict__ db) {
  using namespace tnh3;
  extern __shared__ __attribute__((aligned(16))) _Float16 tnh_lds[];
  float* smax = reinterpret_cast<float*>(tnh_lds + 2 * 4 * TN_PLANE);  // [slot][op][wave]
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, l15 = lane & 15, g = lane >> 4;
  const int aw = wv & 3, bw = wv >> 2;  // wave tile: 32 rows of C (A columns) x 64 columns (B columns)
  const int a0 = blockIdx.y * 128, b0 = blockIdx.x * 128;
  const int m_lo = blockIdx.z * m_chunk, m_hi = min(M, m_lo + m_chunk);
  const int nstep = (m_hi - m_lo + 31) / 32;
  const int s_row = tid >> 5, s_f4 = tid & 31;  // staging: rows tid / 32 and 16 + tid / 32, float4 tid % 32 of a 32 x 128 slab
  const int acol = a0 + 4 * s_f4, bcol = b0 + 4 * s_f4;
  const bool a_ok = acol < N1, b_ok = bcol < N2;
  const bool want_db = db != nullptr && blockIdx.x == 0;
  f32x4 csum = {0.f, 0.f, 0.f, 0.f};
  f32x4 ra[4][2], rb[4][2];  // ring of four slabs in registers
  auto load_slab = [&](int slot, int step) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int m = m_lo + 32 * step + s_row + 16 * j;
      const bool ok = m < m_hi;
      const int mc = ok ? m : m_lo;  // clamped address, zeroed value
      f32x4 va, vb;
      if (A_VEC) {
        va = *reinterpret_cast<const f32x4*>(A + static_cast<int64_t>(mc) * lda + (a_ok ? acol : 0));
        if (!ok || !a_ok) va = f32x4{0.f, 0.f, 0.f, 0.f};
      } else {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const bool in = acol + c < N1;
          const float v = A[static_cast<int64_t>(mc) * lda + (in ? acol + c : 0)];
          va[c] = (ok && in) ? v : 0.0f;
        }
      }
      if (B_VEC) {
        vb = *reinterpret_cast<const f32x4*>(Bm + static_cast<int64_t>(mc) * ldb + (b_ok ? bcol : 0));
        if (!ok || !b_ok) vb = f32x4{0.f, 0.f, 0.f, 0.f};
      } else {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const bool in = bcol + c < N2;
          const float v = Bm[static_cast<int64_t>(mc) * ldb + (in ? bcol + c : 0)];
          vb[c] = (ok && in) ? v : 0.0f;
        }
      }
      if (want_db) csum += va;
      ra[slot][j] = va;
      rb[slot][j] = vb;
    }
  };
  auto post_max = [&](int slot) {  // this wave's largest magnitudes of the slab in ring slot `slot`
    float ma = 0.0f, mb = 0.0f;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        ma = fmaxf(ma, fabsf(ra[slot][j][c]));
        mb = fmaxf(mb, fabsf(rb[slot][j][c]));
      }
    ma = wave_max_nonneg(ma);
    mb = wave_max_nonneg(mb);
    if (lane == 0) {
      smax[slot * 16 + wv] = ma;
      smax[slot * 16 + 8 + wv] = mb;
    }
  };
  auto slab_scales = [&](int slot, float& sa, float& sb, float& inv) {  // (published by the barrier of the step before)
    const f32x4* p = reinterpret_cast<const f32x4*>(smax + slot * 16);
    const f32x4 a0v = p[0], a1v = p[1], b0v = p[2], b1v = p[3];
    const float ma = fmaxf(fmaxf(fmaxf(a0v[0], a0v[1]), fmaxf(a0v[2], a0v[3])), fmaxf(fmaxf(a1v[0], a1v[1]), fmaxf(a1v[2], a1v[3])));
    const float mb = fmaxf(fmaxf(fmaxf(b0v[0], b0v[1]), fmaxf(b0v[2], b0v[3])), fmaxf(fmaxf(b1v[0], b1v[1]), fmaxf(b1v[2], b1v[3])));
    float isa, isb;
    h3_scale(ma, sa, isa);
    h3_scale(mb, sb, isb);
    inv = isa * isb;
  };
  auto store_slab = [&](int slot, int buf, float sa, float sb) {
    _Float16* base = tnh_lds + buf * (4 * TN_PLANE);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int off = tn_off(s_row + 16 * j, s_f4 >> 1) + 4 * (s_f4 & 1);
#pragma unroll
      for (int op = 0; op < 2; ++op) {
        const f32x4 v = op == 0 ? ra[slot][j] : rb[slot][j];
        const float sc = op == 0 ? sa : sb;
        f16x4 h1, h2;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const _Float16 a = static_cast<_Float16>(v[c] * sc);  // (a power of two: exact)
          h1[c] = a;
          h2[c] = static_cast<_Float16>(__builtin_fmaf(v[c], sc, -static_cast<float>(a)));
        }
        _Float16* dst = base + op * (2 * TN_PLANE) + off;
        *reinterpret_cast<f16x4*>(dst) = h1;
        *reinterpret_cast<f16x4*>(dst + TN_PLANE) = h2;
      }
    }
  };
  const int q = l15 >> 2, pp = l15 & 3;
  int offA[2][2], offB[4][2];
#pragma unroll
  for (int rd = 0; rd < 2; ++rd) {
    const int row = 8 * g + 4 * rd + q;
#pragma unroll
    for (int at = 0; at < 2; ++at) offA[at][rd] = tn_off(row, 4 * aw + 2 * at + (pp >> 1)) + 4 * (pp & 1);
#pragma unroll
    for (int bt = 0; bt < 4; ++bt) offB[bt][rd] = 2 * TN_PLANE + tn_off(row, 8 * bw + 2 * bt + (pp >> 1)) + 4 * (pp & 1);
  }
  auto frag = [&](const _Float16* base, int off0, int off1) -> f16x8 {
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(base + off0));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(base + off1));
    const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(f16x8, v);
  };
  f32x4 tot[2][4];
#pragma unroll
  for (int at = 0; at < 2; ++at)
#pragma unroll
    for (int bt = 0; bt < 4; ++bt) tot[at][bt] = f32x4{0.f, 0.f, 0.f, 0.f};

#pragma unroll
  for (int i = 0; i < 4; ++i) load_slab(i, i);
  post_max(0);
  post_max(1);
  asm volatile("" ::: "memory");
  __syncthreads();
  float inv_cur;
  {
    float sa, sb;
    slab_scales(0, sa, sb, inv_cur);
    store_slab(0, 0, sa, sb);
  }
  load_slab(0, 4);  // slot k holds the slab with index == k (mod 4)
  asm volatile("" ::: "memory");
  __syncthreads();
  for (int st0 = 0; st0 < nstep; st0 += 4) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int step = st0 + u;
      if (step >= nstep) break;
      const _Float16* base = tnh_lds + (u & 1) * (4 * TN_PLANE);
      // stage slab step + 1 (requested four steps ago) into the other buffer, request slab step + 5 into the freed slot, post the maxima
      // of slab step + 2
      float sa, sb, inv_next;
      slab_scales((u + 1) & 3, sa, sb, inv_next);
      store_slab((u + 1) & 3, (u & 1) ^ 1, sa, sb);
      load_slab((u + 1) & 3, step + 5);
      post_max((u + 2) & 3);
      asm volatile("" ::: "memory");
      f16x8 fa[2][2], fb[4][2];
#pragma unroll
      for (int p = 0; p < 2; ++p) {
#pragma unroll
        for (int at = 0; at < 2; ++at) fa[at][p] = frag(base + p * TN_PLANE, offA[at][0], offA[at][1]);
#pragma unroll
        for (int bt = 0; bt < 4; ++bt) fb[bt][p] = frag(base + p * TN_PLANE, offB[bt][0], offB[bt][1]);
      }
      f32x4 acc[2][4];
#pragma unroll
      for (int at = 0; at < 2; ++at)
#pragma unroll
        for (int bt = 0; bt < 4; ++bt) acc[at][bt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int term = 0; term < 3; ++term) {  // (a2 b1) (a1 b2) (a1 b1): smallest first
        const int pa = term == 0 ? 1 : 0, pb = term == 1 ? 1 : 0;
#pragma unroll
        for (int at = 0; at < 2; ++at)
#pragma unroll
          for (int bt = 0; bt < 4; ++bt)
            acc[at][bt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[at][pa], fb[bt][pb], acc[at][bt], 0, 0, 0);
      }
#pragma unroll
      for (int at = 0; at < 2; ++at)
#pragma unroll
        for (int bt = 0; bt < 4; ++bt)
#pragma unroll
          for (int r = 0; r < 4; ++r) tot[at][bt][r] = __builtin_fmaf(acc[at][bt][r], inv_cur, tot[at][bt][r]);
      inv_cur = inv_next;
      __syncthreads();
    }
  }
  // D: column l15 <-> C column b0 + 64 bw + 16 bt + l15; row 4 g + r <-> C row a0 + 32 aw + 16 at + 4 g + r
#pragma unroll
  for (int at = 0; at < 2; ++at) {
    const int rbase = a0 + 32 * aw + 16 * at;  // 16-row group: inside one segment (segment ends are multiples of 64)
    if (rbase >= N1) continue;
    float* cbase = C;
    int rloc = rbase;
    if (segs.nseg > 0) {
      int s_ = 0, beg = 0;
      while (s_ + 1 < segs.nseg && rbase >= segs.n_end[s_]) { beg = segs.n_end[s_]; ++s_; }
      cbase = segs.p[s_];
      rloc = rbase - beg;
    }
#pragma unroll
    for (int bt = 0; bt < 4; ++bt) {
      const int col = b0 + 64 * bw + 16 * bt + l15;
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (col < N2 && rbase + 4 * g + r < N1) atomicAdd(cbase + static_cast<int64_t>(rloc + 4 * g + r) * ldc + col, tot[at][bt][r]);
    }
  }
  if (db != nullptr) {  // uniform per launch.  The slabs requested past the end loaded zeros, so csum holds exactly this chunk's rows.
    __syncthreads();  // every wave is done with the staging buffers
    float* red = reinterpret_cast<float*>(tnh_lds);  // [8 waves][32 column groups][4]
#pragma unroll
    for (int c = 0; c < 4; ++c) csum[c] += __shfl_xor(csum[c], 32);  // lanes l and l + 32 hold the same columns (rows 2 wv, 2 wv + 1)
    if (lane < 32) *reinterpret_cast<f32x4*>(red + (wv * 32 + lane) * 4) = csum;
    __syncthreads();
    if (want_db && tid < 128) {
      const int col = a0 + tid;  // column group tid / 4, component tid % 4
      float v = 0.f;
#pragma unroll
      for (int w = 0; w < 8; ++w) v += red[(w * 32 + (tid >> 2)) * 4 + (tid & 3)];
      if (col < N1) atomicAdd(db + col, v);
    }
  }
}

// C[N1 x N2] += A[M x N1]^T B[M x N2] (launch_gemm_tn_b6's contract: db, segmented C rows)
int launch_gemm_tn_h3(const float* A, int lda, const float* B, int ldb, float* C, int ldc, int M, int N1, int N2, float* db,
                      float* const* seg_ptrs, const int* seg_ends, int nseg, hipStream_t st) {
  const bool a_vec = lda % 4 == 0 && N1 % 4 == 0 && (reinterpret_cast<uintptr_t>(A) & 15) == 0;
  const bool b_vec = ldb % 4 == 0 && N2 % 4 == 0 && (reinterpret_cast<uintptr_t>(B) & 15) == 0;
  DIFFAB_REQUIRE(A && B && N1 >= 1 && N2 >= 1 && M >= 1 && (reinterpret_cast<uintptr_t>(A) & 3) == 0 && (reinterpret_cast<uintptr_t>(B) & 3) == 0 &&
                     nseg >= 0 && nseg <= 6 && (nseg > 0 || C) && (nseg == 0 || (a_vec && N1 % 16 == 0)),
                 DIFFAB_ERR_ARG, "gemm_tn_h3: unsupported operands");
  TnSegsH3 sg{};
  sg.nseg = nseg;
  for (int i = 0; i < nseg; ++i) { sg.p[i] = seg_ptrs[i]; sg.n_end[i] = seg_ends[i]; }
  const int t1 = (N1 + 127) / 128, t2 = (N2 + 127) / 128, tiles = t1 * t2;
  int splits = 256 / tiles;  // at most one (tile, M chunk) work-group per CU
  if (splits < 1) splits = 1;
  int m_chunk = (M + splits - 1) / splits;
  m_chunk = ((m_chunk < 256 ? 256 : m_chunk) + 31) / 32 * 32;
  const int nchunks = (M + m_chunk - 1) / m_chunk;
  const dim3 grid(t2, t1, nchunks);
#define TNH3_LAUNCH(AV_, BV_)                                                                                                         \
  do {                                                                                                                                \
    DIFFAB_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_tn_h3_kernel<AV_, BV_>),                                   \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, tnh3::TN_LDS_BYTES));                            \
    hipLaunchKernelGGL((gemm_tn_h3_kernel<AV_, BV_>), grid, dim3(512), tnh3::TN_LDS_BYTES, st, A, lda, B, ldb, C, ldc, M, m_chunk, N1, N2, \
                       sg, db);                                                                                                       \
  } while (0)
  if (a_vec && b_vec) TNH3_LAUNCH(true, true);
  else if (a_vec) TNH3_LAUNCH(true, false);
  else if (b_vec) TNH3_LAUNCH(false, true);
  else TNH3_LAUNCH(false, false);
#undef TNH3_LAUNCH
  DIFFAB_LAUNCH_CHECK();
  return DIFFAB_OK;
}

// Y[rows x N] (leading dimension ldy) = X[rows x 128] W, W[k][n] = W[n sn + k sk] (the backward's d feat = d y W_out: sn = 1, sk = F).
// scratch: xstat_h3_scratch_bytes(N), 16-byte aligned, overwritten (planes | 1 / column scale)
static int xstat_h3_blocks(int N) { return (N + pjh3::PJ_B - 1) / pjh3::PJ_B; }
size_t xstat_h3_scratch_bytes(int N) {
  const size_t nb = xstat_h3_blocks(N);
  return 2 * nb * pjh3::PJ_STAGE_ELEMS * sizeof(_Float16) + nb * pjh3::PJ_B * sizeof(float);
}
int launch_xstat_h3(const float* X, const float* W, int64_t sn, int64_t sk, float* Y, int ldy, int rows, int N, void* scratch, hipStream_t st) {
  using namespace pjh3;
  DIFFAB_REQUIRE(X && W && Y && scratch && (reinterpret_cast<uintptr_t>(scratch) & 15) == 0 && (reinterpret_cast<uintptr_t>(X) & 15) == 0 &&
                     N >= 1 && rows >= 1 && ldy >= N,
                 DIFFAB_ERR_ARG, "xstat_h3: unsupported operands");
  const int NB = xstat_h3_blocks(N);
  _Float16* Wc = static_cast<_Float16*>(scratch);
  float* wis = reinterpret_cast<float*>(Wc + static_cast<size_t>(2 * NB) * PJ_STAGE_ELEMS);
  hipLaunchKernelGGL(xsplit_h3_kernel, dim3(NB * PJ_B), dim3(64), 0, st, W, sn, sk, N, Wc, wis);
  DIFFAB_LAUNCH_CHECK();
  const int ntiles = (rows + PJ_ROWS - 1) / PJ_ROWS;
  int nsplit = 256 / ntiles;
  nsplit = nsplit < 1 ? 1 : (nsplit > NB ? NB : nsplit);
  const dim3 grid(ntiles, nsplit);
#define XSH3_LAUNCH(FULL_, SPLIT_)                                                                                              \
  do {                                                                                                                          \
    DIFFAB_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(xstat_h3_kernel<FULL_, SPLIT_>),                          \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, PJ_LDS_BYTES));                             \
    hipLaunchKernelGGL((xstat_h3_kernel<FULL_, SPLIT_>), grid, dim3(512), PJ_LDS_BYTES, st, X, Wc, wis, Y, rows, NB, N, ldy);     \
  } while (0)
  if (rows % PJ_ROWS == 0) {
    if (nsplit > 1) XSH3_LAUNCH(true, true);
    else XSH3_LAUNCH(true, false);
  } else {
    if (nsplit > 1) XSH3_LAUNCH(false, true);
    else XSH3_LAUNCH(false, false);
  }
#undef XSH3_LAUNCH
  DIFFAB_LAUNCH_CHECK();
  return DIFFAB_OK;
}

}  // namespace diffab
