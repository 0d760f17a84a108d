// gemm_bf16x6.hip - fp32-accurate dense projections on the bf16 matrix cores.
//
// Why: on gfx950 the f32-input MFMA (v_mfma_f32_16x16x4_f32) runs at the fp32 VECTOR rate and - measured, tools/mfma4x4_probe.hip -
// does not overlap with VALU work at all (32 cycles per MFMA + ~3.5 per v_fma issued beside it): in fp32, matrix and vector
// arithmetic are ONE pipe, and the dense projections of an IPA layer (six projections 11.3 GFLOP, to_out 8.6 GFLOP at B = 256)
// hold it for 0.19 ms of a 0.55 ms layer.  The bf16 matrix cores are a separate pipe, 16x the rate.  An fp32 number is exactly
// the sum of three bf16 numbers (8 + 8 + 8 mantissa bits: hi = bf16(x), mid = bf16(x - hi), lo = bf16(x - hi - mid)), and
//     a b = a1 b1 + (a1 b2 + a2 b1) + (a1 b3 + a3 b1 + a2 b2) + O(2^-24 |a b|)
// so SIX bf16 MFMAs with fp32 accumulation reproduce the fp32 product to fp32 rounding (each partial product of two 8-bit
// mantissas is exact in fp32).  Measured on random data against float64: 2.7e-7 max-rel for the six-term form, 4.0e-7 for an
// fp32 GEMM of the same operands (three terms only: 4e-6 - not used).  The parity tests hold the layers built on it to the same
// bars as before.  6 MFMA 16x16x32 = 96 cycles per 16x16x32 block against 256 for the 8 f32 16x16x4 it replaces, and the VALU
// (operand splitting, epilogues) runs beside them.
//
// Y[M x 128] = act(X[M x K] W^T + b): the N = 128 layers of the denoiser (to_out with K = 1024, the MLP layers with K = 128).
// Reference: nn.Linear calls at diffab_pytorch.py:375-379, :459-464 (to_out), :515-556 (MLPs).
#include "common.h"
#include "denoiser_internal.h"
#include "rowgemm_b6_tile.h"
#include "proj_frames_b6_tile.h"
#include "mlp_chain_tile.h"

namespace diffab {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define MEM_FENCE() asm volatile("" ::: "memory")

using b6tile::BK;      // k per MFMA (16x16x32) = k per weight chunk
using b6tile::split3;  // fp32 -> (hi, mid, lo) bf16
using b6tile::b6_off;

// W[128 x Kd] fp32 (rows ldw floats apart) -> three bf16 planes, chunk-major:  out[((chunk * 3 + s) * 128 + n) * 32 + kk],
// chunk = k / 32, kk = k % 32.  One chunk = 24 KiB contiguous: the GEMM stages it with full-line loads and no address arithmetic.
__global__ void wsplit128_kernel(const float* __restrict__ W, int ldw, int Kd, __bf16* __restrict__ out, int nrows) {
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;  // (n, k); rows n >= nrows of the 128-row operand are zero (narrow last layers)
  if (gid >= 128 * Kd) return;
  const int n = gid / Kd, k = gid % Kd;
  __bf16 h, m, l;
  split3(n < nrows ? W[static_cast<int64_t>(n) * ldw + k] : 0.0f, h, m, l);
  const int chunk = k / BK, kk = k % BK;
  const size_t base = (static_cast<size_t>(chunk) * 3 * 128 + n) * BK + kk;
  out[base] = h;
  out[base + 128 * BK] = m;
  out[base + 2 * 128 * BK] = l;
}

// the same planes from a strided view: element (n, k) of the 128 x kseg operand is W[n * sn + k * sk]; it lands at k index k0 + k
// (k0 % 32 == 0).  Transposed weights for the input-gradient products (dX = dY W is Y = X W'^T with W' = W^T).
__global__ void wsplit128_strided_kernel(const float* __restrict__ W, int64_t sn, int64_t sk, int kseg, int k0, __bf16* __restrict__ out) {
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;  // (k, n): n fastest, so sn == 1 reads are coalesced
  if (gid >= 128 * kseg) return;
  const int n = gid & 127, k = gid >> 7;
  __bf16 h, m, l;
  split3(W[n * sn + k * sk], h, m, l);
  const int kg = k0 + k, chunk = kg / BK, kk = kg % BK;
  const size_t base = (static_cast<size_t>(chunk) * 3 * 128 + n) * BK + kk;
  out[base] = h;
  out[base + 128 * BK] = m;
  out[base + 2 * 128 * BK] = l;
}

// the transposed planes of up to six matrices stacked along k (the IPA projections' input gradient): segment q holds rows
// [k_end[q-1], k_end[q]) of the stacked operand as W_q[k - k_begin][n], n = 0..127 (one launch instead of six)
struct SplitSegs { const float* p[6]; int k_end[6]; int nseg; };
__global__ void wsplit128_segs_kernel(SplitSegs sg, int Ktot, __bf16* __restrict__ out) {
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;  // (k, n): n fastest - coalesced reads of W_q rows
  if (gid >= 128 * Ktot) return;
  const int n = gid & 127, kg = gid >> 7;
  int q = 0, beg = 0;
  while (q + 1 < sg.nseg && kg >= sg.k_end[q]) { beg = sg.k_end[q]; ++q; }
  __bf16 h, m, l;
  split3(sg.p[q][static_cast<int64_t>(kg - beg) * 128 + n], h, m, l);
  const int chunk = kg / BK, kk = kg % BK;
  const size_t base = (static_cast<size_t>(chunk) * 3 * 128 + n) * BK + kk;
  out[base] = h;
  out[base + 128 * BK] = m;
  out[base + 2 * 128 * BK] = l;
}

// LDS (dynamic): weights Ws[2 buffers][3 planes][128 rows][32 k], then X As[2][3][ROWS][32] - both operands are staged as split
// bf16 planes: every element of X is split ONCE per work-group (its two column waves share the rows), and X is read from HBM in
// full 128-byte lines (the MFMA fragment shape - adjacent lanes on different rows - costs the texture addresser four lines per
// lane quad: with fragment-shaped loads straight from global memory this kernel took 58 us instead of 48).  Rows are 64 bytes,
// unpadded; the 16-byte slot s of row r lives at slot s ^ ((r >> 2) & 3), which makes the 32-row ds_read_b128 fragment reads
// conflict-free (the four 16-lane groups of the instruction each take 16 different bank quads).
// ROWS = 128: 512 threads, waves 4 (rows) x 2 (columns), 96 KiB of LDS, one work-group per CU.
// ROWS = 64: 256 threads, waves 2 x 2, 72 KiB: two work-groups per CU - one group's MFMAs run under the other's staging and
// barrier - and twice the work-groups for the same rows (B <= 128 patches per GPU leave half the CUs idle with 128-row groups).
template <int ROWS>
constexpr int b6_lds_bytes() { return b6tile::lds_bytes<ROWS>(); }

template <bool RELU, int ROWS>
__global__ __launch_bounds__(ROWS * 4) void rowgemm128_b6_kernel(const float* __restrict__ X, int ldx, const __bf16* __restrict__ Wc,
                                                                 const float* __restrict__ bias, const int64_t* __restrict__ bias_idx,
                                                                 int bias_div, float* __restrict__ Y, int ldy, int M, int Kd) {
  extern __shared__ __attribute__((aligned(16))) __bf16 b6_lds[];
  b6tile::rowgemm128_tile<RELU, ROWS>(b6_lds, threadIdx.x, blockIdx.x, X, ldx, Wc, bias, bias_idx, bias_div, Y, ldy, M, Kd);  // rowgemm_b6_tile.h
}

// The same product for a handful of row tiles (one patch: two 64-row tiles on 256 CUs): one work-group per (row tile, k part) leaves its
// part in parts[tile][part][ROWS][128], a second launch adds the parts of an element in the fixed order of rowgemm_b6_tile.h - the result
// is bitwise the one of rowgemm128_b6_kernel.  (One launch with the last work-group of a tile adding up was measured: 72 us against 31
// for the plain kernel at one patch - the agent-scope release / acquire it needs writes back and invalidates the XCD's whole L2.)
template <int ROWS>
__global__ __launch_bounds__(ROWS * 4) void rowgemm128_b6_parts_kernel(const float* __restrict__ X, int ldx, const __bf16* __restrict__ Wc, int M,
                                                                       int Kd, float* __restrict__ parts) {
  extern __shared__ __attribute__((aligned(16))) __bf16 b6_lds[];
  float* mine = parts + (static_cast<size_t>(blockIdx.x) * gridDim.y + blockIdx.y) * (ROWS * 128);
  b6tile::rowgemm128_tile<false, ROWS, true>(b6_lds, threadIdx.x, blockIdx.x, X, ldx, Wc, nullptr, nullptr, 0, nullptr, 0, M, Kd,
                                             blockIdx.y * b6tile::PART_CHUNKS, mine);
}
template <bool RELU, int ROWS>
__global__ __launch_bounds__(256) void rowgemm128_b6_parts_sum_kernel(const float* __restrict__ parts, int nparts, const float* __restrict__ bias,
                                                                      const int64_t* __restrict__ bias_idx, int bias_div,
                                                                      float* __restrict__ Y, int ldy, int M) {
  const int gid = blockIdx.x * 256 + threadIdx.x;  // (row, 4 columns)
  const int row = gid >> 5, col = (gid & 31) * 4;
  if (row >= M) return;
  const float* src = parts + (static_cast<size_t>(row / ROWS) * nparts * ROWS + row % ROWS) * 128 + col;
  b6tile::f32x4 tot = {0.f, 0.f, 0.f, 0.f};
  for (int p = 0; p < nparts; ++p) {
    const b6tile::f32x4 v = *reinterpret_cast<const b6tile::f32x4*>(src + static_cast<size_t>(p) * (ROWS * 128));
#pragma unroll
    for (int c = 0; c < 4; ++c) tot[c] += v[c];
  }
  const bool table = bias_idx != nullptr || bias_div > 0;
  const float* brow = table ? bias + (bias_idx ? bias_idx[row] : row / bias_div) * 128 : bias;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    float o = tot[c] + (brow ? brow[col + c] : 0.f);
    if (RELU) o = fmaxf(o, 0.f);
    Y[static_cast<int64_t>(row) * ldy + col + c] = o;
  }
}

// ================================================================== a chain of two or three 128-wide dense layers in one kernel
// (the tile body, its structs and the description: mlp_chain_tile.h)
__global__ __launch_bounds__(512) void mlp_chain_b6_kernel(const float* __restrict__ X, int ldx, MlpChainSet set, int M) {
  // the chain of this work-group, field by field (uniform selects: indexing the by-value argument would move it to scratch)
  const int cy = blockIdx.y;
#define CHAIN_SEL(f) (cy == 0 ? set.c[0].f : cy == 1 ? set.c[1].f : set.c[2].f)
  extern __shared__ __attribute__((aligned(16))) __bf16 chain_lds[];
  chaintile::mlp_chain_tile(chain_lds, threadIdx.x, blockIdx.x, X, ldx, CHAIN_SEL(planes[0]), CHAIN_SEL(planes[1]), CHAIN_SEL(planes[2]),
                            CHAIN_SEL(bias[0]), CHAIN_SEL(bias[1]), CHAIN_SEL(bias[2]), CHAIN_SEL(bias_idx0), CHAIN_SEL(bias_div0),
                            CHAIN_SEL(nlayers), CHAIN_SEL(n_out), cy == 0 ? set.Y[0] : cy == 1 ? set.Y[1] : set.Y[2],
                            cy == 0 ? set.ldy[0] : cy == 1 ? set.ldy[1] : set.ldy[2], M);
#undef CHAIN_SEL
}

// X[M x 128] through `nchains` (1..3) chains of 2 or 3 dense layers each, one launch (see mlp_chain_b6_kernel); planes from
// launch_wsplit128 (Kd = 128; the last layer's with nrows = n_out); chain c: planes[3 c + layer], bias[3 c + layer], layer-0 bias table
// selected by bias_idx0 / bias_div0 (shared by the chains: each has its own table behind bias[3 c])
int make_mlp_chain_set(MlpChainSet* out, int nchains, const void* const* planes, const float* const* bias, const int64_t* bias_idx0,
                       int bias_div0, int nlayers, const int* n_out, float* const* Y, const int* ldy) {
  DIFFAB_REQUIRE(out && nchains >= 1 && nchains <= 3 && (nlayers == 2 || nlayers == 3), DIFFAB_ERR_ARG, "mlp_chain_b6: unsupported operands");
  MlpChainSet set{};
  for (int c = 0; c < nchains; ++c) {
    DIFFAB_REQUIRE(Y[c] && n_out[c] >= 1 && n_out[c] <= 128, DIFFAB_ERR_ARG, "mlp_chain_b6: unsupported output");
    for (int i = 0; i < nlayers; ++i) {
      const void* pl = planes[3 * c + i];
      DIFFAB_REQUIRE(pl && (reinterpret_cast<uintptr_t>(pl) & 15) == 0, DIFFAB_ERR_ARG, "mlp_chain_b6: null / misaligned planes");
      set.c[c].planes[i] = static_cast<const __bf16*>(pl);
      set.c[c].bias[i] = bias[3 * c + i];
    }
    set.c[c].bias_idx0 = bias_idx0;
    set.c[c].bias_div0 = bias_div0;
    set.c[c].nlayers = nlayers;
    set.c[c].n_out = n_out[c];
    set.Y[c] = Y[c];
    set.ldy[c] = ldy[c];
  }
  *out = set;
  return DIFFAB_OK;
}
int launch_mlp_chains_b6(const float* X, int ldx, int nchains, const void* const* planes, const float* const* bias, const int64_t* bias_idx0,
                         int bias_div0, int nlayers, const int* n_out, float* const* Y, const int* ldy, int M, hipStream_t st) {
  DIFFAB_REQUIRE(X && M >= 1 && ldx % 4 == 0 && (reinterpret_cast<uintptr_t>(X) & 15) == 0, DIFFAB_ERR_ARG, "mlp_chain_b6: unsupported operands");
  MlpChainSet set{};
  if (int rc = make_mlp_chain_set(&set, nchains, planes, bias, bias_idx0, bias_div0, nlayers, n_out, Y, ldy)) return rc;
  DIFFAB_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(mlp_chain_b6_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, kChainLdsBytes));
  hipLaunchKernelGGL(mlp_chain_b6_kernel, dim3((M + 127) / 128, nchains), dim3(512), kChainLdsBytes, st, X, ldx, set, M);
  DIFFAB_LAUNCH_CHECK();
  return DIFFAB_OK;
}
int launch_mlp_chain_b6(const float* X, int ldx, const void* const* planes, const float* const* bias, const int64_t* bias_idx0, int bias_div0,
                        int nlayers, int n_out, float* Y, int ldy, int M, hipStream_t st) {
  const void* pl[3] = {planes[0], planes[1], nlayers == 3 ? planes[2] : nullptr};
  const float* bs[3] = {bias[0], bias[1], nlayers == 3 ? bias[2] : nullptr};
  return launch_mlp_chains_b6(X, ldx, 1, pl, bs, bias_idx0, bias_div0, nlayers, &n_out, &Y, &ldy, M, st);
}

size_t rowgemm128_b6_scratch_bytes(int Kd) { return static_cast<size_t>(3) * 128 * Kd * sizeof(__bf16); }

bool rowgemm128_b6_ok(const float* X, int ldx, const float* Y, int ldy, int M, int Kd) {
  return Kd % BK == 0 && Kd >= BK && ldx % 4 == 0 && ldy % 4 == 0 && (reinterpret_cast<uintptr_t>(X) & 15) == 0 &&
         (reinterpret_cast<uintptr_t>(Y) & 15) == 0 && M >= 1;
}

// W[128 x Kd] (rows ldw floats apart) -> split planes for rowgemm128_b6p (rowgemm128_b6_scratch_bytes(Kd) bytes, 16-byte aligned)
int launch_wsplit128(const float* W, int ldw, int Kd, void* planes, hipStream_t st, int nrows) {
  DIFFAB_REQUIRE(W && planes && (reinterpret_cast<uintptr_t>(planes) & 15) == 0 && Kd % BK == 0 && nrows >= 1 && nrows <= 128, DIFFAB_ERR_ARG,
                 "wsplit128: bad operands");
  hipLaunchKernelGGL(wsplit128_kernel, dim3((128 * Kd + 255) / 256), dim3(256), 0, st, W, ldw, Kd, static_cast<__bf16*>(planes), nrows);
  DIFFAB_LAUNCH_CHECK();
  return DIFFAB_OK;
}

int launch_wsplit128_strided(const float* W, int64_t sn, int64_t sk, int kseg, int k0, void* planes, hipStream_t st) {
  DIFFAB_REQUIRE(W && planes && (reinterpret_cast<uintptr_t>(planes) & 15) == 0 && k0 % BK == 0 && kseg >= 1, DIFFAB_ERR_ARG,
                 "wsplit128_strided: bad operands");
  hipLaunchKernelGGL(wsplit128_strided_kernel, dim3((128 * kseg + 255) / 256), dim3(256), 0, st, W, sn, sk, kseg, k0, static_cast<__bf16*>(planes));
  DIFFAB_LAUNCH_CHECK();
  return DIFFAB_OK;
}

int launch_wsplit128_segs(const float* const* W, const int* k_end, int nseg, void* planes, hipStream_t st) {
  DIFFAB_REQUIRE(W && k_end && nseg >= 1 && nseg <= 6 && planes && (reinterpret_cast<uintptr_t>(planes) & 15) == 0 && k_end[nseg - 1] % BK == 0,
                 DIFFAB_ERR_ARG, "wsplit128_segs: bad operands");
  SplitSegs sg{};
  sg.nseg = nseg;
  for (int i = 0; i < nseg; ++i) { sg.p[i] = W[i]; sg.k_end[i] = k_end[i]; }
  const int Ktot = k_end[nseg - 1];
  hipLaunchKernelGGL(wsplit128_segs_kernel, dim3((128 * Ktot + 255) / 256), dim3(256), 0, st, sg, Ktot, static_cast<__bf16*>(planes));
  DIFFAB_LAUNCH_CHECK();
  return DIFFAB_OK;
}

// Y[M x 128] = act(X[:, 0:Kd] W[:, 0:Kd]^T + bias row) with W given as split planes (launch_wsplit128).
// parts (optional): rowgemm128_b6_parts_floats(M, Kd) floats of scratch - with it a product of few row tiles (latency-bound: one patch is
// two work-groups on 256 CUs) runs one work-group per (row tile, 128-k part) and a second launch adds the parts up.
size_t rowgemm128_b6_parts_floats(int M, int Kd) {
  return static_cast<size_t>((M + 63) / 64) * ((Kd / b6tile::BK + b6tile::PART_CHUNKS - 1) / b6tile::PART_CHUNKS) * 64 * 128;
}
int launch_rowgemm128_b6p(const float* X, int ldx, const void* planes, const float* bias, const int64_t* bias_idx, int bias_div, float* Y,
                          int ldy, int M, int Kd, bool relu, hipStream_t st, float* parts) {
  DIFFAB_REQUIRE(rowgemm128_b6_ok(X, ldx, Y, ldy, M, Kd) && planes && (reinterpret_cast<uintptr_t>(planes) & 15) == 0, DIFFAB_ERR_ARG,
                 "rowgemm128_b6: unsupported operands");
  const __bf16* Wc = static_cast<const __bf16*>(planes);
  const int tiles64 = (M + 63) / 64, nparts = (Kd / b6tile::BK + b6tile::PART_CHUNKS - 1) / b6tile::PART_CHUNKS;
  if (parts && nparts > 1 && tiles64 <= 64 && (reinterpret_cast<uintptr_t>(parts) & 15) == 0) {  // up to 32 patches of 128 residues
    DIFFAB_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(rowgemm128_b6_parts_kernel<64>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, b6_lds_bytes<64>()));
    hipLaunchKernelGGL((rowgemm128_b6_parts_kernel<64>), dim3(tiles64, nparts), dim3(256), b6_lds_bytes<64>(), st, X, ldx, Wc, M, Kd, parts);
    const dim3 sgrid((M * 32 + 255) / 256);
    if (relu) hipLaunchKernelGGL((rowgemm128_b6_parts_sum_kernel<true, 64>), sgrid, dim3(256), 0, st, parts, nparts, bias, bias_idx, bias_div, Y, ldy, M);
    else hipLaunchKernelGGL((rowgemm128_b6_parts_sum_kernel<false, 64>), sgrid, dim3(256), 0, st, parts, nparts, bias, bias_idx, bias_div, Y, ldy, M);
    DIFFAB_LAUNCH_CHECK();
    return DIFFAB_OK;
  }
  // 128-row work-groups when they fill the chip (measured at 256 groups: 47 us against 53 for 64-row groups, K = 1024), 64-row
  // groups below that (B <= 128 patches of 128 residues per GPU: twice the groups)
  const int rows_env = (M + 127) / 128 >= 256 ? 128 : 64;
#define B6_LAUNCH(RELU_, ROWS_)                                                                                                         \
  do {                                                                                                                                  \
    DIFFAB_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(rowgemm128_b6_kernel<RELU_, ROWS_>),                              \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, b6_lds_bytes<ROWS_>()));                           \
    hipLaunchKernelGGL((rowgemm128_b6_kernel<RELU_, ROWS_>), dim3((M + ROWS_ - 1) / ROWS_), dim3(ROWS_ * 4), b6_lds_bytes<ROWS_>(), st, X, \
                       ldx, Wc, bias, bias_idx, bias_div, Y, ldy, M, Kd);                                                               \
  } while (0)
  if (rows_env == 128) {
    if (relu) B6_LAUNCH(true, 128);
    else B6_LAUNCH(false, 128);
  } else {
    if (relu) B6_LAUNCH(true, 64);
    else B6_LAUNCH(false, 64);
  }
#undef B6_LAUNCH
  DIFFAB_LAUNCH_CHECK();
  return DIFFAB_OK;
}

// split + product in one call; `scratch`: rowgemm128_b6_scratch_bytes(Kd) bytes, 16-byte aligned
int launch_rowgemm128_b6(const float* X, int ldx, const float* W, int ldw, const float* bias, const int64_t* bias_idx, int bias_div, float* Y,
                         int ldy, int M, int Kd, bool relu, void* scratch, hipStream_t st) {
  if (int rc = launch_wsplit128(W, ldw, Kd, scratch, st)) return rc;
  return launch_rowgemm128_b6p(X, ldx, scratch, bias, bias_idx, bias_div, Y, ldy, M, Kd, relu, st);
}

// ================================================================== six IPA projections + local->global frames (bf16x6)
// proj[:, 0:1344] = x [Wq_s; Wk_s; Wv_s; Wq_p; Wk_p; Wv_p]^T, the three point blocks mapped to the global frame (x R + t,
// diffab_pytorch.py:324) before they are stored - the bf16x6 form of proj_frames_kernel (denoiser_fast.hip), same decomposition:
// x-stationary (a wave keeps its 32 x 128 slab of x as SPLIT A fragments: 96 VGPRs), 14 blocks of 96 output columns, MFMA n index
// permuted so that a lane ends up with three consecutive output columns (one point) per row.  The weights arrive pre-split
// (pjsplit_kernel) in stage order - stage = (block, k half): 3 planes x 96 LDS rows x 64 k, 36 KiB contiguous - and are staged
// through a ring of two LDS buffers with rows padded to 160 bytes (conflict-free ds_read_b128 for the 16-row B fragment).
// (the tile body and its constants: proj_frames_b6_tile.h)
using namespace pjtile;

// stage-ordered split weights: out[((blk * 2 + kh) * 3 + plane) * 96 + l][kk], l = 48 cw + 16 tt + j <-> output column
// 96 blk + 48 cw + 3 j + tt, k = 64 kh + kk
__global__ void pjsplit_kernel(const float* __restrict__ W0, const float* __restrict__ W1, const float* __restrict__ W2,
                               const float* __restrict__ W3, const float* __restrict__ W4, const float* __restrict__ W5,
                               __bf16* __restrict__ out) {
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;  // (output column gc, k)
  if (gid >= PJ_NP * 128) return;
  const int gc = gid >> 7, k = gid & 127;
  const float* Wp;
  int row;
  if (gc < PJ_GQ) {
    Wp = gc < 256 ? W0 : (gc < 512 ? W1 : W2);
    row = gc & 255;
  } else {
    Wp = gc < PJ_GK ? W3 : (gc < PJ_GV ? W4 : W5);
    row = gc - (gc < PJ_GK ? PJ_GQ : (gc < PJ_GV ? PJ_GK : PJ_GV));
  }
  __bf16 h, m, l;
  split3(Wp[row * 128 + k], h, m, l);
  const int blk = gc / PJ_B, rem = gc % PJ_B, cwl = rem / 48, r48 = rem % 48, j = r48 / 3, tt = r48 % 3;
  const int lrow = 48 * cwl + 16 * tt + j, kh = k >> 6, kk = k & 63;
  const size_t base = (static_cast<size_t>(blk * 2 + kh) * 3 * PJ_B + lrow) * 64 + kk;
  out[base] = h;
  out[base + PJ_B * 64] = m;
  out[base + 2 * PJ_B * 64] = l;
}

template <bool FULL, bool PROJ, bool SPLIT = false>
__global__ __launch_bounds__(512) void proj_frames_b6_kernel(const float* __restrict__ X, const __bf16* __restrict__ Wc,
                                                             const float* __restrict__ R, const float* __restrict__ t,
                                                             float* __restrict__ Y, int M, int N_, int NB_, int ldy_, int frames_from_) {
  extern __shared__ __attribute__((aligned(16))) __bf16 pj_lds[];
  pjtile::proj_frames_b6_tile<FULL, PROJ, SPLIT>(pj_lds, threadIdx.x, blockIdx.x, blockIdx.y, gridDim.y, X, Wc, R, t, Y, M, N_, NB_, ldy_,
                                                 frames_from_);
}

size_t proj_frames_b6_scratch_bytes() { return static_cast<size_t>(2 * PJ_NB) * PJ_STAGE_ELEMS * sizeof(__bf16); }

// W6 = {wq_s, wk_s, wv_s, wq_p, wk_p, wv_p} -> stage-ordered split planes (proj_frames_b6_scratch_bytes() bytes, 16-byte aligned)
int launch_pjsplit(const float* const* W6, void* planes, hipStream_t st) {
  DIFFAB_REQUIRE(planes && (reinterpret_cast<uintptr_t>(planes) & 15) == 0, DIFFAB_ERR_ARG, "pjsplit: bad operands");
  hipLaunchKernelGGL(pjsplit_kernel, dim3((PJ_NP * 128 + 255) / 256), dim3(256), 0, st, W6[0], W6[1], W6[2], W6[3], W6[4], W6[5],
                     static_cast<__bf16*>(planes));
  DIFFAB_LAUNCH_CHECK();
  return DIFFAB_OK;
}

static int launch_xstat(const float* x, const void* planes, const float* R, const float* t, float* Y, int rows, int N, int NB, int ldy,
                        int frames_from, hipStream_t st) {
  DIFFAB_REQUIRE(planes && (reinterpret_cast<uintptr_t>(planes) & 15) == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0 &&
                     (reinterpret_cast<uintptr_t>(Y) & 3) == 0 && rows >= 1 && NB % 2 == 0 && NB * PJ_B >= N,
                 DIFFAB_ERR_ARG, "proj_frames_b6: unsupported operands");
  const __bf16* Wc = static_cast<const __bf16*>(planes);
  const dim3 grid((rows + PJ_ROWS - 1) / PJ_ROWS);
  const bool proj_geom = N == PJ_NP && NB == PJ_NB && ldy == PJ_NP && frames_from == PJ_GQ / PJ_B;
  const int ntiles = (rows + PJ_ROWS - 1) / PJ_ROWS;
  // half the chip or less: several groups per row tile, each with its share of the column blocks (B = 1: one block per group)
  int nsplit = 256 / ntiles;
  nsplit = nsplit < 1 ? 1 : (nsplit > NB ? NB : nsplit);
  const bool split = nsplit > 1;
  const dim3 grid2(ntiles, nsplit);
#define XSTAT_LAUNCH(FULL_, PROJ_, SPLIT_)                                                                                               \
  do {                                                                                                                                   \
    DIFFAB_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(proj_frames_b6_kernel<FULL_, PROJ_, SPLIT_>),                      \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, PJ_LDS_BYTES));                                      \
    hipLaunchKernelGGL((proj_frames_b6_kernel<FULL_, PROJ_, SPLIT_>), grid2, dim3(512), PJ_LDS_BYTES, st, x, Wc, R, t, Y, rows, N, NB,    \
                       ldy, frames_from);                                                                                                \
  } while (0)
#define XSTAT_PICK(FULL_)                                     \
  do {                                                        \
    if (proj_geom && split) XSTAT_LAUNCH(FULL_, true, true);  \
    else if (proj_geom) XSTAT_LAUNCH(FULL_, true, false);     \
    else if (split) XSTAT_LAUNCH(FULL_, false, true);         \
    else XSTAT_LAUNCH(FULL_, false, false);                   \
  } while (0)
  if (rows % PJ_ROWS == 0) XSTAT_PICK(true);
  else XSTAT_PICK(false);
#undef XSTAT_PICK
#undef XSTAT_LAUNCH
  DIFFAB_LAUNCH_CHECK();
  return DIFFAB_OK;
}

// the six projections of one IPA layer (D = 128) into proj[rows x 1344], weights given as split planes (launch_pjsplit)
int launch_proj_frames_b6p(const float* x, const void* planes, const float* R, const float* t, float* proj, int rows, hipStream_t st) {
  return launch_xstat(x, planes, R, t, proj, rows, PJ_NP, PJ_NB, PJ_NP, PJ_GQ / PJ_B, st);
}

// ---- the x-stationary kernel as a plain product: Y[rows x N] = X[rows x 128] W'^T, element (n, k) of W' = W[n sn + k sk]
static int xstat_blocks(int N) { return ((N + PJ_B - 1) / PJ_B + 1) / 2 * 2; }
size_t xstat_b6_scratch_bytes(int N) { return static_cast<size_t>(2 * xstat_blocks(N)) * PJ_STAGE_ELEMS * sizeof(__bf16); }
__global__ void xsplit_kernel(const float* __restrict__ W, int64_t sn, int64_t sk, int N, int ncols, __bf16* __restrict__ out) {
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;  // (k, column): column fastest (sn == 1 reads are coalesced)
  if (gid >= ncols * 128) return;
  const int gc = gid % ncols, k = gid / ncols;
  __bf16 h, m, l;
  split3(gc < N ? W[gc * sn + k * sk] : 0.0f, h, m, l);
  const int blk = gc / PJ_B, rem = gc % PJ_B, cwl = rem / 48, r48 = rem % 48, j = r48 / 3, tt = r48 % 3;
  const int lrow = 48 * cwl + 16 * tt + j, kh = k >> 6, kk = k & 63;
  const size_t base = (static_cast<size_t>(blk * 2 + kh) * 3 * PJ_B + lrow) * 64 + kk;
  out[base] = h;
  out[base + PJ_B * 64] = m;
  out[base + 2 * PJ_B * 64] = l;
}
int launch_xstat_b6(const float* X, const float* W, int64_t sn, int64_t sk, float* Y, int ldy, int rows, int N, void* scratch, hipStream_t st) {
  DIFFAB_REQUIRE(W && scratch && (reinterpret_cast<uintptr_t>(scratch) & 15) == 0 && N >= 1, DIFFAB_ERR_ARG, "xstat_b6: bad operands");
  const int NB = xstat_blocks(N), ncols = NB * PJ_B;
  hipLaunchKernelGGL(xsplit_kernel, dim3((ncols * 128 + 255) / 256), dim3(256), 0, st, W, sn, sk, N, ncols, static_cast<__bf16*>(scratch));
  DIFFAB_LAUNCH_CHECK();
  return launch_xstat(X, scratch, nullptr, nullptr, Y, rows, N, NB, ldy, NB, st);
}

// ================================================================== weight-gradient products: C[N1 x N2] += A^T B  (bf16x6)
// A = dY [M x N1], B = X [M x N2], both fp32 row-major: the contraction runs over the rows, i.e. over the STRIDED index of both
// operands, while a bf16 MFMA fragment wants 8 consecutive k per lane.  A work-group stages 32-row slabs of both operands as
// split bf16 planes in their natural [row][column] orientation (full-line loads, split once) and reads the fragments with
// ds_read_b64_tr_b16, the transposing LDS read of gfx950: per 16-lane group, lane i receives column i of a 4-row block.  Two such
// reads (rows 8 g .. 8 g + 3 and 8 g + 4 .. 8 g + 7 of the slab) are one 16x16x32 fragment; A and B use the same row order, which
// is all the contraction needs.  Image: 256-byte rows, 16-byte chunk c of row r at chunk c ^ (((r & 3) << 2) | ((r >> 2) & 3))
// (conflict-free transposed reads, MI355X guide T10 image (b)).  Output tile 128 x 128 per work-group, split over M in chunks;
// partial tiles are added with fp32 atomics (as the f32 kernel does).  Rows of C may live in up to six matrices (the IPA
// projections: `segs`), selected per 16-row group.
namespace {
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
constexpr int TN_PLANE = 32 * 128;  // bf16 elements of one plane of one slab
__device__ __forceinline__ int tn_off(int row, int chunk) {  // bf16 element offset of 16-byte chunk `chunk` of row `row`
  return row * 128 + 8 * (chunk ^ (((row & 3) << 2) | ((row >> 2) & 3)));
}
}  // namespace

struct TnSegs {
  float* p[6];
  int n_end[6];
  int nseg;  // 0: one plain matrix C
};

// A_VEC / B_VEC: the operand's rows are 16-byte aligned and its width a multiple of 4 (float4 loads); otherwise guarded scalar loads
// (the 131-wide cat[h, tau] input of the head MLPs, the 3- and 20-wide head outputs).  db (nullable): db[n1] += sum_m A[m][n1], the
// bias gradient, from the A values the b-tile-0 work-groups load anyway.
template <bool A_VEC, bool B_VEC>
__global__ __launch_bounds__(512) void gemm_tn_b6_kernel(const float* __restrict__ A, int lda, const float* __restrict__ Bm, int ldb,
                                                         float* __restrict__ C, int ldc, int M, int m_chunk, int N1, int N2, TnSegs segs,
                                                         float* __restrict__ db) {
  extern __shared__ __attribute__((aligned(16))) __bf16 tn_lds[];  // [2 buffers][A | B][3 planes][32][128]
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, l15 = lane & 15, g = lane >> 4;
  const int aw = wv & 3, bw = wv >> 2;  // wave tile: 32 rows of C (A columns) x 64 columns (B columns)
  const int a0 = blockIdx.y * 128, b0 = blockIdx.x * 128;
  const int m_lo = blockIdx.z * m_chunk, m_hi = min(M, m_lo + m_chunk);
  const int nstep = (m_hi - m_lo + 31) / 32;
  // staging: a slab is 32 rows x 128 columns of each operand = 1024 float4 each; thread -> rows tid / 32 and 16 + tid / 32, float4 tid % 32
  const int s_row = tid >> 5, s_f4 = tid & 31;
  const int acol = a0 + 4 * s_f4, bcol = b0 + 4 * s_f4;
  const bool a_ok = acol < N1, b_ok = bcol < N2;  // vector path: the width is a multiple of 4, a float4 is inside or outside
  const bool want_db = db != nullptr && blockIdx.x == 0;
  f32x4 csum = {0.f, 0.f, 0.f, 0.f};
  f32x4 ra[4][2], rb[4][2];  // ring of four slabs in registers
  auto load_slab = [&](int slot, int step) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int m = m_lo + 32 * step + s_row + 16 * j;
      const bool ok = m < m_hi;
      const int mc = ok ? m : m_lo;  // clamped address, zeroed value (unconditional loads: see rowgemm128_b6_kernel)
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
  typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
  auto store_slab = [&](int slot, int buf) {
    __bf16* base = tn_lds + buf * (6 * TN_PLANE);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int off = tn_off(s_row + 16 * j, s_f4 >> 1) + 4 * (s_f4 & 1);
#pragma unroll
      for (int op = 0; op < 2; ++op) {
        const f32x4 v = op == 0 ? ra[slot][j] : rb[slot][j];
        bf16x4 h, m, l;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          __bf16 hh, mm, ll;
          split3(v[c], hh, mm, ll);
          h[c] = hh; m[c] = mm; l[c] = ll;
        }
        __bf16* dst = base + op * (3 * TN_PLANE) + off;
        *reinterpret_cast<bf16x4*>(dst) = h;
        *reinterpret_cast<bf16x4*>(dst + TN_PLANE) = m;
        *reinterpret_cast<bf16x4*>(dst + 2 * TN_PLANE) = l;
      }
    }
  };
  // transposed fragment reads: lane 4 q + p of a 16-lane group supplies row q, columns 4 p .. 4 p + 3 of the 4 x 16 block; group g takes
  // slab rows 8 g + 4 rd + q.  Offsets per (tile, read) differ by a constant only within a row pair, so all are precomputed.
  const int q = l15 >> 2, pp = l15 & 3;
  int offA[2][2], offB[4][2];
#pragma unroll
  for (int rd = 0; rd < 2; ++rd) {
    const int row = 8 * g + 4 * rd + q;
#pragma unroll
    for (int at = 0; at < 2; ++at) offA[at][rd] = tn_off(row, 4 * aw + 2 * at + (pp >> 1)) + 4 * (pp & 1);
#pragma unroll
    for (int bt = 0; bt < 4; ++bt) offB[bt][rd] = 3 * TN_PLANE + tn_off(row, 8 * bw + 2 * bt + (pp >> 1)) + 4 * (pp & 1);
  }
  auto frag = [&](const __bf16* base, int off0, int off1) -> bf16x8 {
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(base + off0));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(base + off1));
    const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
  };
  f32x4 acc[2][4];
#pragma unroll
  for (int at = 0; at < 2; ++at)
#pragma unroll
    for (int bt = 0; bt < 4; ++bt) acc[at][bt] = f32x4{0.f, 0.f, 0.f, 0.f};

#pragma unroll
  for (int i = 0; i < 4; ++i) load_slab(i, i);
  MEM_FENCE();
  store_slab(0, 0);
  load_slab(0, 4);  // slot k holds the slab with index == k (mod 4)
  MEM_FENCE();
  __syncthreads();
  constexpr int TA[6] = {1, 2, 0, 1, 0, 0}, TB[6] = {1, 0, 2, 0, 1, 0};  // (mid,mid) (lo,hi) (hi,lo) (mid,hi) (hi,mid) (hi,hi)
  for (int st0 = 0; st0 < nstep; st0 += 4) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int step = st0 + u;
      if (step >= nstep) break;
      const __bf16* base = tn_lds + (u & 1) * (6 * TN_PLANE);
      // stage slab step + 1 (requested four steps ago) into the other buffer, request slab step + 5 into the freed slot
      store_slab((u + 1) & 3, (u & 1) ^ 1);
      load_slab((u + 1) & 3, step + 5);
      MEM_FENCE();
      bf16x8 fa[2][3], fb[4][3];
#pragma unroll
      for (int p = 0; p < 3; ++p) {
#pragma unroll
        for (int at = 0; at < 2; ++at) fa[at][p] = frag(base + p * TN_PLANE, offA[at][0], offA[at][1]);
#pragma unroll
        for (int bt = 0; bt < 4; ++bt) fb[bt][p] = frag(base + p * TN_PLANE, offB[bt][0], offB[bt][1]);
      }
#pragma unroll
      for (int term = 0; term < 6; ++term)
#pragma unroll
        for (int at = 0; at < 2; ++at)
#pragma unroll
          for (int bt = 0; bt < 4; ++bt)
            acc[at][bt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[at][TA[term]], fb[bt][TB[term]], acc[at][bt], 0, 0, 0);
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
        if (col < N2 && rbase + 4 * g + r < N1) atomicAdd(cbase + static_cast<int64_t>(rloc + 4 * g + r) * ldc + col, acc[at][bt][r]);
    }
  }
  if (db != nullptr) {  // uniform per launch.  The slabs requested past the end loaded zeros, so csum holds exactly this chunk's rows.
    __syncthreads();  // every wave is done with the staging buffers
    float* red = reinterpret_cast<float*>(tn_lds);  // [8 waves][32 column groups][4]
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

bool gemm_tn_b6_ok(const float* A, int lda, const float* B, int ldb, int M, int N1, int N2) {
  return N1 >= 1 && N2 >= 1 && M >= 1 && (reinterpret_cast<uintptr_t>(A) & 3) == 0 && (reinterpret_cast<uintptr_t>(B) & 3) == 0;
}

// C[N1 x N2] += A[M x N1]^T B[M x N2]; db (nullable): db[N1] += column sums of A; seg_ptrs / seg_ends (nseg <= 6, ends multiples of
// 64, vector-aligned A only): rows of C spread over several matrices
int launch_gemm_tn_b6(const float* A, int lda, const float* B, int ldb, float* C, int ldc, int M, int N1, int N2, float* db,
                      float* const* seg_ptrs, const int* seg_ends, int nseg, hipStream_t st) {
  const bool a_vec = lda % 4 == 0 && N1 % 4 == 0 && (reinterpret_cast<uintptr_t>(A) & 15) == 0;
  const bool b_vec = ldb % 4 == 0 && N2 % 4 == 0 && (reinterpret_cast<uintptr_t>(B) & 15) == 0;
  DIFFAB_REQUIRE(gemm_tn_b6_ok(A, lda, B, ldb, M, N1, N2) && nseg >= 0 && nseg <= 6 && (nseg > 0 || C) && (nseg == 0 || (a_vec && N1 % 16 == 0)),
                 DIFFAB_ERR_ARG, "gemm_tn_b6: unsupported operands");
  TnSegs sg{};
  sg.nseg = nseg;
  for (int i = 0; i < nseg; ++i) { sg.p[i] = seg_ptrs[i]; sg.n_end[i] = seg_ends[i]; }
  const int t1 = (N1 + 127) / 128, t2 = (N2 + 127) / 128, tiles = t1 * t2;
  int splits = 256 / tiles;  // at most one (tile, M chunk) work-group per CU (one group per CU fits; 264 groups would run in two rounds)
  if (splits < 1) splits = 1;
  int m_chunk = (M + splits - 1) / splits;
  m_chunk = ((m_chunk < 256 ? 256 : m_chunk) + 31) / 32 * 32;
  const int nchunks = (M + m_chunk - 1) / m_chunk;
  constexpr int lds = 2 * 6 * TN_PLANE * 2;  // 98 304 bytes
  const dim3 grid(t2, t1, nchunks);
#define TN_LAUNCH(AV_, BV_)                                                                                                           \
  do {                                                                                                                                \
    DIFFAB_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_tn_b6_kernel<AV_, BV_>),                                   \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, lds));                                           \
    hipLaunchKernelGGL((gemm_tn_b6_kernel<AV_, BV_>), grid, dim3(512), lds, st, A, lda, B, ldb, C, ldc, M, m_chunk, N1, N2, sg, db);   \
  } while (0)
  if (a_vec && b_vec) TN_LAUNCH(true, true);
  else if (a_vec) TN_LAUNCH(true, false);
  else if (b_vec) TN_LAUNCH(false, true);
  else TN_LAUNCH(false, false);
#undef TN_LAUNCH
  DIFFAB_LAUNCH_CHECK();
  return DIFFAB_OK;
}

}  // namespace diffab
