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
