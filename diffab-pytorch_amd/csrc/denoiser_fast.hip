// denoiser_fast.hip - MFMA (fp32-in / fp32-accumulate, exact fp32) kernels for the benchmark geometry
// D=128, C=64, H=8, DS=32, PQ=PV=8 (reference train.py:62-70), any K that is a multiple of 16.
//
//  * linear_mfma_kernel: Y = act(X W^T + b) on v_mfma_f32_32x32x2_f32, LDS-staged 128 x {64,128} x 32 tiles.  Both
//    operands are K-contiguous (X rows and nn.Linear weight rows), so one 16-byte LDS read per lane feeds four
//    MFMA k-steps (k order inside a step is permuted identically for A and B, which a dot product does not see).
//  * ipa_attn_fast_kernel: one work-group (8 waves) per (patch, 16 query residues):
//      phase 1 (wave = head):   scalar logits q.k on MFMA 16x16x4, point logits as direct differences on the VALU
//                               (the |q|^2+|k|^2-2qk form loses ~1e-5) -> S in LDS
//      phase 2 (wave = 2 rows): pair bias e.Wb on MFMA with e streamed global->VGPR (each e element read exactly once
//                               from HBM), softmax over the 128 keys in registers, attn-weighted pair sum o_e on MFMA
//                               with e re-read through L2 in the transposed fragment order -> P in LDS
//      phase 3 (wave = head):   attn-weighted scalar / point sums on MFMA (P from LDS, V side from L2), global->local
//                               frames and norms -> feature row (1024) for the output projection.
//    K/V-side operands (448 KiB per patch) are produced once per layer by the projection GEMM and re-read by the 8
//    row tiles of a patch, which the blockIdx map places on one XCD so the re-reads are L2 hits.
//
// Reference: InvariantPointAttentionLayer.forward, diffab_pytorch.py:389-465.
#include <cstdlib>
#include <type_traits>

#include "common.h"
#include "denoiser_internal.h"
#include "rowgemm_b6_tile.h"
#include "ipa_attn_tile.h"
#include "attn_planes_tile.h"

namespace diffab {

// (f32x4 / f32x16 / f32x2 and MEM_FENCE come with ipa_attn_tile.h)

// ================================================================== Y = act(X W^T + b) on MFMA 32x32x2
constexpr int LBM = 128, LBK = 32, LLD = LBK + 4;  // LDS row stride 36 floats: ds_read_b128 conflict-free

struct LinearSegs {  // up to 6 weight matrices sharing X, written side by side into Y (the IPA projections)
  const float* W[6];
  int n_end[6];  // exclusive end column of each segment in Y
  int nseg;
};

template <int BN, bool RELU, bool VEC>
__global__ __launch_bounds__(256) void linear_mfma_kernel(const float* __restrict__ X, int ldx, LinearSegs segs,
                                                          const float* __restrict__ bias, float* __restrict__ Y, int ldy, int M, int N,
                                                          int Kd) {
  __shared__ __attribute__((aligned(16))) float As[LBM * LLD];
  __shared__ __attribute__((aligned(16))) float Bs[BN * LLD];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int m0 = blockIdx.y * LBM, n0 = blockIdx.x * BN;
  // which weight matrix this column block belongs to (segments are multiples of BN wide, or there is one segment)
  int seg = 0;
  while (seg + 1 < segs.nseg && n0 >= segs.n_end[seg]) ++seg;
  const int seg_begin = seg == 0 ? 0 : segs.n_end[seg - 1];
  const float* __restrict__ W = segs.W[seg];
  const int seg_rows = segs.n_end[seg] - seg_begin;  // rows of this weight matrix
  const int wrow0 = n0 - seg_begin;

  constexpr int NT = BN / 32;  // 32-column accumulator tiles per wave
  f32x16 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;

  constexpr int A_F4 = LBM * LBK / 4 / 256;  // float4 per thread for the A tile (4)
  constexpr int B_F4 = BN * LBK / 4 / 256;   // (2 or 4)
  f32x4 ra[A_F4], rb[B_F4];

  auto load_tile = [&](int k0) {
#pragma unroll
    for (int r = 0; r < A_F4; ++r) {
      const int idx = tid + r * 256, row = idx >> 3, c4 = (idx & 7) * 4;
      const int gm = m0 + row, gk = k0 + c4;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (gm < M) {
        const float* p = X + static_cast<int64_t>(gm) * ldx + gk;
        if (VEC) {
          if (gk < Kd) v = *reinterpret_cast<const f32x4*>(p);
        } else {
#pragma unroll
          for (int q = 0; q < 4; ++q)
            if (gk + q < Kd) v[q] = p[q];
        }
      }
      ra[r] = v;
    }
#pragma unroll
    for (int r = 0; r < B_F4; ++r) {
      const int idx = tid + r * 256, row = idx >> 3, c4 = (idx & 7) * 4;
      const int wr = wrow0 + row, gk = k0 + c4;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (wr < seg_rows) {
        const float* p = W + static_cast<int64_t>(wr) * Kd + gk;
        if (VEC) {
          if (gk < Kd) v = *reinterpret_cast<const f32x4*>(p);
        } else {
#pragma unroll
          for (int q = 0; q < 4; ++q)
            if (gk + q < Kd) v[q] = p[q];
        }
      }
      rb[r] = v;
    }
  };
  auto store_tile = [&]() {
#pragma unroll
    for (int r = 0; r < A_F4; ++r) {
      const int idx = tid + r * 256, row = idx >> 3, c4 = (idx & 7) * 4;
      *reinterpret_cast<f32x4*>(&As[row * LLD + c4]) = ra[r];
    }
#pragma unroll
    for (int r = 0; r < B_F4; ++r) {
      const int idx = tid + r * 256, row = idx >> 3, c4 = (idx & 7) * 4;
      *reinterpret_cast<f32x4*>(&Bs[row * LLD + c4]) = rb[r];
    }
  };

  load_tile(0);
  for (int k0 = 0; k0 < Kd; k0 += LBK) {
    __syncthreads();  // previous tile's reads are done
    store_tile();
    __syncthreads();
    if (k0 + LBK < Kd) load_tile(k0 + LBK);  // next tile's global loads fly under the MFMAs
    const int l31 = lane & 31, hh = lane >> 5;
#pragma unroll
    for (int t4 = 0; t4 < LBK / 8; ++t4) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(&As[(wv * 32 + l31) * LLD + t4 * 8 + hh * 4]);
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const f32x4 b = *reinterpret_cast<const f32x4*>(&Bs[(t * 32 + l31) * LLD + t4 * 8 + hh * 4]);
#pragma unroll
        for (int s = 0; s < 4; ++s) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], b[s], acc[t], 0, 0, 0);
      }
    }
  }
  // D layout 32x32: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int col = n0 + t * 32 + (lane & 31);
    if (col >= N) continue;
    const float bv = bias ? bias[col] : 0.0f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = m0 + wv * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      if (row >= M) continue;
      float v = acc[t][r] + bv;
      if (RELU) v = fmaxf(v, 0.0f);
      Y[static_cast<int64_t>(row) * ldy + col] = v;
    }
  }
}

template <int BN>
static int launch_linear_bn(const float* X, int ldx, const LinearSegs& segs, const float* bias, float* Y, int ldy, int M, int N, int Kd,
                            bool relu, bool vec, hipStream_t st) {
  dim3 grid((N + BN - 1) / BN, (M + LBM - 1) / LBM);
#define LAUNCH(R, V) hipLaunchKernelGGL((linear_mfma_kernel<BN, R, V>), grid, dim3(256), 0, st, X, ldx, segs, bias, Y, ldy, M, N, Kd)
  if (relu) { if (vec) LAUNCH(true, true); else LAUNCH(true, false); }
  else      { if (vec) LAUNCH(false, true); else LAUNCH(false, false); }
#undef LAUNCH
  DIFFAB_LAUNCH_CHECK();
  return DIFFAB_OK;
}

static bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// ================================================================== Y[M x 128] = act(X[M x K] W^T + b), K % 128 == 0
// The to_out projection (K = 1024) and the 128-wide MLP layers.  With N = 128 the tiled kernel above has one 4-wave work-group
// per CU (a 1 x M/128 grid) and nothing to hide latency with.  Here a work-group of 8 waves owns 128 rows and ALL 128 columns:
// wave (rw, cw) accumulates a 32 x 64 tile in registers, takes its A fragments straight from global memory (a row slab is not
// shared between waves, so staging it through LDS buys nothing), double-buffered one 64-wide K chunk ahead, and only the weight
// chunk [128][64] goes through LDS (double-buffered, one barrier per chunk).  The MFMA n index is permuted (tile tt, lane column j
// <-> output column 4 j + tt of the wave's 64) so a lane ends with 4 consecutive columns: float4 stores, 256 contiguous bytes per
// 16 lanes.
constexpr int RG_KC = 64, RG_LD = RG_KC + 4;
constexpr int RG_MT = 2;  // 16-row MFMA tiles per wave: 2 -> 128 rows per work-group (1 -> 64 rows, two work-groups per CU: measured slower)
constexpr int RG_ROWS = 64 * RG_MT;
constexpr bool RG_DEEP_A = false;  // true: A register sets span two weight chunks (deeper prefetch) - measured SLOWER (101 vs 88 us at K = 1024)

// AW: weight chunks (64 k) per A register set.  The A fragments of set n+1 are requested while set n is consumed, so AW = 2
// doubles the prefetch distance (2 x 3.5 us of MFMA work at K = 1024): with AW = 1 every CU asks for its next 64 KiB at the same
// instant after each barrier and the burst (16 MiB chip-wide) does not drain within one chunk of compute.
template <bool RELU, int AW>
__global__ __launch_bounds__(512, RG_MT == 1 ? 2 : 1) void rowgemm128_kernel(const float* __restrict__ X, int ldx,
                                                                            const float* __restrict__ W, int ldw,
                                                                            const float* __restrict__ bias,
                                                                            const int64_t* __restrict__ bias_idx, int bias_div,
                                                                            float* __restrict__ Y, int ldy, int M, int Kd) {
  // bias: one vector (bias_idx == nullptr, bias_div == 0), or a table of 128-wide rows indexed by bias_idx[row] or row / bias_div
  // (the folded concatenations of the denoiser: a per-residue-type or per-patch affine term, see fold_tables in api.hip).
  // W rows are ldw floats apart and only 4-byte aligned (the 131-wide head weights are read in place).
  constexpr int MT = RG_MT;
  __shared__ __attribute__((aligned(16))) float Ws[2 * 128 * RG_LD];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int l15 = lane & 15, g = lane >> 4, rw = wv & 3, cw = wv >> 2;
  const int m0 = blockIdx.x * RG_ROWS;
  const int nchunk = Kd / RG_KC;  // a multiple of 2 AW (launcher)

  // weight staging: thread -> (LDS row l = 32 r + tid / 16, float4 column tid % 16); LDS row l = 64 cw' + 16 tt + j <-> W row 64 cw' + 4 j + tt
  f32x4 wreg[4];
  const float* wsrc[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int l = 32 * r + (tid >> 4), c4 = tid & 15;
    const int n = (l & 64) + 4 * (l & 15) + ((l >> 4) & 3);
    wsrc[r] = W + static_cast<int64_t>(n) * ldw + 4 * c4;
  }
  typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
  auto load_w = [&](int ch) {
    ch = ch < nchunk ? ch : nchunk - 1;  // unconditional prefetch: the last trips re-read the final chunk
#pragma unroll
    for (int r = 0; r < 4; ++r) wreg[r] = *reinterpret_cast<const f32x4u*>(wsrc[r] + ch * RG_KC);
  };
  auto store_w = [&](int buf) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int l = 32 * r + (tid >> 4), c4 = tid & 15;
      *reinterpret_cast<f32x4*>(&Ws[(buf * 128 + l) * RG_LD + 4 * c4]) = wreg[r];
    }
  };
  // A fragments: a[mt][kq][s] = X[m0 + 16 MT rw + 16 mt + l15][64 AW set + 16 kq + 4 g + s], kq < 4 AW; rows past M are clamped
  // (their results are never stored)
  const float* asrc[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    int row = m0 + 16 * MT * rw + 16 * mt + l15;
    row = row < M ? row : M - 1;
    asrc[mt] = X + static_cast<int64_t>(row) * ldx + 4 * g;
  }
  const int nset = nchunk / AW;
  f32x4 aA[MT][4 * AW], aB[MT][4 * AW];
  auto load_a = [&](f32x4 (&a)[MT][4 * AW], int set) {
    set = set < nset ? set : nset - 1;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int kq = 0; kq < 4 * AW; ++kq) a[mt][kq] = *reinterpret_cast<const f32x4*>(asrc[mt] + set * (RG_KC * AW) + 16 * kq);
  };
  f32x4 acc[MT][4];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int tt = 0; tt < 4; ++tt) acc[mt][tt] = f32x4{0.f, 0.f, 0.f, 0.f};

  // one weight chunk: `part` selects which 64 k of the A set it pairs with
  auto compute = [&](const f32x4 (&a)[MT][4 * AW], int part, int buf) {
    const float* Wl = Ws + (buf * 128 + 64 * cw + l15) * RG_LD + 4 * g;
    f32x4 b[2][4];
#pragma unroll
    for (int tt = 0; tt < 4; ++tt) b[0][tt] = *reinterpret_cast<const f32x4*>(Wl + 16 * tt * RG_LD);
#pragma unroll
    for (int kq = 0; kq < 4; ++kq) {
      if (kq + 1 < 4) {
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) b[(kq + 1) & 1][tt] = *reinterpret_cast<const f32x4*>(Wl + 16 * tt * RG_LD + 16 * (kq + 1));
      }
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int tt = 0; tt < 4; ++tt)
            acc[mt][tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt][4 * part + kq][s], b[kq & 1][tt][s], acc[mt][tt], 0, 0, 0);
    }
  };

  load_w(0);
  load_a(aA, 0);
  store_w(0);
  __syncthreads();
  // 2 AW weight chunks per trip so the A register sets and the LDS buffers alternate statically.  All prefetches are
  // UNCONDITIONAL (indices are clamped): a branch around them makes the compiler's vmcnt bookkeeping fall back to the no-prefetch
  // path's count, i.e. every chunk waits for the loads just issued.
  for (int ch = 0; ch < nchunk; ch += 2 * AW) {
    load_a(aB, ch / AW + 1);
#pragma unroll
    for (int p = 0; p < AW; ++p) {
      load_w(ch + p + 1);
      MEM_FENCE();
      compute(aA, p, p & 1);
      store_w((p + 1) & 1);
      __syncthreads();
    }
    load_a(aA, ch / AW + 2);
#pragma unroll
    for (int p = 0; p < AW; ++p) {
      load_w(ch + AW + p + 1);
      MEM_FENCE();
      compute(aB, p, (AW + p) & 1);
      store_w((AW + p + 1) & 1);
      __syncthreads();
    }
  }
  // lane holds columns 64 cw + 4 l15 + (0..3) of rows 16 MT rw + 16 mt + 4 g + r
  const int col = 64 * cw + 4 * l15;
  const bool table = bias_idx != nullptr || bias_div > 0;
  f32x4 bv = {0.f, 0.f, 0.f, 0.f};
  if (bias && !table) bv = *reinterpret_cast<const f32x4*>(bias + col);
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = m0 + 16 * MT * rw + 16 * mt + 4 * g + r;
      if (row >= M) continue;
      if (table) {
        const int64_t bi = bias_idx ? bias_idx[row] : row / bias_div;
        bv = *reinterpret_cast<const f32x4*>(bias + bi * 128 + col);
      }
      f32x4 o = {acc[mt][0][r] + bv[0], acc[mt][1][r] + bv[1], acc[mt][2][r] + bv[2], acc[mt][3][r] + bv[3]};
      if (RELU) { o[0] = fmaxf(o[0], 0.f); o[1] = fmaxf(o[1], 0.f); o[2] = fmaxf(o[2], 0.f); o[3] = fmaxf(o[3], 0.f); }
      *reinterpret_cast<f32x4*>(Y + static_cast<int64_t>(row) * ldy + col) = o;
    }
}

bool rowgemm128_ok(const float* X, int ldx, const float* Y, int ldy, int M, int Kd) {
  return Kd % (2 * RG_KC) == 0 && ldx % 4 == 0 && ldy % 4 == 0 && aligned16(X) && aligned16(Y) && M >= RG_ROWS;
}
// Y[M x 128] = act(X[:, 0:Kd] W[:, 0:Kd]^T + table row); W rows ldw apart; see rowgemm128_kernel
int launch_rowgemm128(const float* X, int ldx, const float* W, int ldw, const float* bias, const int64_t* bias_idx, int bias_div, float* Y,
                      int ldy, int M, int Kd, bool relu, hipStream_t st) {
  DIFFAB_REQUIRE(rowgemm128_ok(X, ldx, Y, ldy, M, Kd) && (!bias || aligned16(bias)), DIFFAB_ERR_ARG, "rowgemm128: unsupported operands");
  const dim3 grid((M + RG_ROWS - 1) / RG_ROWS);
#define RG_LAUNCH(R, AW_) \
  hipLaunchKernelGGL((rowgemm128_kernel<R, AW_>), grid, dim3(512), 0, st, X, ldx, W, ldw, bias, bias_idx, bias_div, Y, ldy, M, Kd)
  if (RG_DEEP_A && Kd % (4 * RG_KC) == 0) { if (relu) RG_LAUNCH(true, 2); else RG_LAUNCH(false, 2); }
  else                                    { if (relu) RG_LAUNCH(true, 1); else RG_LAUNCH(false, 1); }
#undef RG_LAUNCH
  DIFFAB_LAUNCH_CHECK();
  return DIFFAB_OK;
}

int launch_linear(const float* X, int ldx, const float* W, const float* bias, float* Y, int ldy, int M, int N, int Kd, bool relu,
                  hipStream_t st) {
  LinearSegs segs{};
  segs.W[0] = W;
  segs.n_end[0] = N;
  segs.nseg = 1;
  const bool vec = (ldx % 4 == 0) && (Kd % 4 == 0) && aligned16(X) && aligned16(W);
  if (N == 128 && vec && rowgemm128_ok(X, ldx, Y, ldy, M, Kd) && (!bias || aligned16(bias)))
    return launch_rowgemm128(X, ldx, W, Kd, bias, nullptr, 0, Y, ldy, M, Kd, relu, st);
  if (N > 64) return launch_linear_bn<128>(X, ldx, segs, bias, Y, ldy, M, N, Kd, relu, vec, st);
  return launch_linear_bn<64>(X, ldx, segs, bias, Y, ldy, M, N, Kd, relu, vec, st);
}

// ================================================================== fused IPA attention (benchmark geometry)
// The body is ipa_attn_tile.h (a device function, shared with the patch-resident module kernel of ipa_persistent.hip); here one
// work-group per (patch, 16 query residues), grid = B K / 16.
template <int NT, bool MULTI, bool PLANES = false, bool TAPE = false, int NW = 8, bool VPL = false>
__global__ __launch_bounds__(64 * NW, NW == 4 ? 2 : 1) void ipa_attn_fast_kernel(const float* __restrict__ proj, const float* __restrict__ e,
                                                            const float* __restrict__ R, const float* __restrict__ t,
                                                            const float* __restrict__ Wb, const float* __restrict__ gamma,
                                                            float* __restrict__ feat, int B, int NC_arg,
                                                            unsigned long long* __restrict__ stamps, const float* __restrict__ esc = nullptr,
                                                            float* __restrict__ tape_p = nullptr, float* __restrict__ tape_d2 = nullptr,
                                                            const unsigned char* __restrict__ tile_needed = nullptr,
                                                            const f32x4* __restrict__ vpl = nullptr, const float* __restrict__ vsc = nullptr) {
  extern __shared__ __attribute__((aligned(16))) float S[];
  const int ntile = (MULTI ? NC_arg : 1) * NT;  // K / TI
  // XCD-aware map: blocks b and b+8 share an XCD (round-robin dispatch), so give all row tiles of one patch to one XCD.
  int b, tile;
  const unsigned bid = blockIdx.x;
  if ((B & 7) == 0) {
    const int xcd = bid & 7, slot = bid >> 3;
    b = (slot / ntile) * 8 + xcd;
    tile = slot % ntile;
  } else {
    b = bid / ntile;
    tile = bid % ntile;
  }
  // tile_needed (reverse sampler, last layer, DIFFAB_FLAG_SKIP_UNUSED_ROWS): the outputs of this layer are read for generated residues
  // only - a row tile without one leaves at once (uniform; its feature rows keep the previous layer's values, which nothing reads)
  if (tile_needed != nullptr && !tile_needed[b * ntile + tile]) return;
  ipa_attn_tile<NT, MULTI, PLANES, TAPE, NW, VPL>(S, b, tile, bid, proj, e, R, t, Wb, gamma, feat, NC_arg, stamps, esc, tape_p, tape_d2, vpl, vsc);
}

static unsigned long long* g_attn_stamps = nullptr;  // diagnostics only (diffab_debug_set_attn_stamps)
static bool g_pair_chain_off = false;                // diffab_debug_set_attn_variant bit 6 (64)
static bool g_tn_b6 = false;                         // diffab_debug_set_attn_variant bit 5 (32)
static int g_attn_variant = 0;                       // diagnostics only (diffab_debug_set_attn_variant): 1 = four-wave work-groups
static bool g_value_planes = false;                  // diffab_debug_set_attn_variant bit 4 (16): the value side of P x V as fp16 planes (measured, not the default:
                                                     // profiles/r06_attention.md)
bool value_planes_enabled() { return g_value_planes; }
void set_pair_embed_fused(bool on);  // pair_embed_fused.hip
void set_attn_variant(int v) {       // A/B switches for tests and tools (include/diffab_hip.h)
  g_attn_variant = v & 9;  // 1: four-wave attention work-groups; 8: the two big dense products of a layer as six-term bf16 products
  g_pair_chain_off = (v & 64) != 0;  // 64: the PairEmbedding backward's 64-wide tail as its separate launches (A/B of pair_chain_bwd_kernel)
  g_tn_b6 = (v & 32) != 0;  // 32: the weight-gradient products of the training backward in the six-term bf16 form (A/B of gemm_tn_h3_kernel)
  set_pair_embed_fused(!(v & 4));
  g_value_planes = (v & 16) != 0;  // 16: value planes - phase 3 of the attention tile on the f16 matrix cores (proj_frames_h3_tile.h "Value planes")
}
bool dense_h3_enabled() { return g_attn_variant != 8; }
bool pair_chain_bwd_enabled() { return !g_pair_chain_off; }
bool tn_h3_enabled() { return g_attn_variant != 8 && !g_tn_b6; }
void set_attn_stamps(void* p) {
  g_attn_stamps = static_cast<unsigned long long*>(p);
}

// in-place local -> global for the three point blocks of the projection buffer (row-vector convention, :324)
__global__ void points_to_global_fast_kernel(float* __restrict__ proj, const float* __restrict__ R, const float* __restrict__ t, int rows) {
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;  // (row, point 0..191)
  if (gid >= rows * 192) return;
  const int r = gid / 192, p = gid % 192;
  float* qv = proj + static_cast<int64_t>(r) * ANP + OFF_GQ + p * 3;
  const float* Rr = R + static_cast<int64_t>(r) * 9;
  const float x = qv[0], y = qv[1], z = qv[2];
  qv[0] = (x * Rr[0] + y * Rr[3] + z * Rr[6]) + t[r * 3 + 0];
  qv[1] = (x * Rr[1] + y * Rr[4] + z * Rr[7]) + t[r * 3 + 1];
  qv[2] = (x * Rr[2] + y * Rr[5] + z * Rr[8]) + t[r * 3 + 2];
}

// ================================================================== folded concatenations of the denoiser (D = 128)
// cat[res_ctx, E[s]] W0^T + b0 = res_ctx W0[:, :D]^T + (E[s] W0[:, D:]^T + b0): the second term depends only on the residue type s
// (25 rows), so it becomes a bias table indexed by seq_t and the 2D-wide concatenation is never materialised
// (reference diffab_pytorch.py:572-574).  Likewise cat[h, (beta, sin beta, cos beta)] W^T + b = h W[:, :D]^T + per-patch row
// (diffab_pytorch.py:584-588) for each of the three heads.
__global__ __launch_bounds__(64) void fold_embed_table_kernel(const float* __restrict__ emb, const float* __restrict__ W0,
                                                              const float* __restrict__ b0, int D, int n_types, float* __restrict__ tab) {
  // one wave per (residue type s, 8 outputs n): lanes along k so the weight rows are read coalesced, the 8 rows' loads in flight
  // together (a serial loop over n is a chain of exposed memory latencies: 20 us for a 25 x 128 table)
  const int s_ = blockIdx.x, n0 = blockIdx.y * 8, lane = threadIdx.x;
  if (s_ >= n_types) return;
  float acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    acc[i] = 0.f;
    const int n = n0 + i;
    if (n < D)
      for (int k = lane; k < D; k += 64) acc[i] += emb[s_ * D + k] * W0[n * 2 * D + D + k];
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    for (int o = 32; o > 0; o >>= 1) acc[i] += __shfl_xor(acc[i], o);
    if (lane == 0 && n0 + i < D) tab[s_ * D + n0 + i] = acc[i] + b0[n0 + i];
  }
}
__global__ void fold_beta_table_kernel(const float* __restrict__ beta, const float* __restrict__ Wa, const float* __restrict__ ba,
                                       const float* __restrict__ Wb_, const float* __restrict__ bb, const float* __restrict__ Wc,
                                       const float* __restrict__ bc, int D, int B, float* __restrict__ tab,
                                       const float* __restrict__ sched_beta, int t, const int* __restrict__ t_dev) {
  const int b = blockIdx.x, hd = blockIdx.y, n = threadIdx.x;  // tab[hd][b][n]
  if (b >= B || n >= D) return;
  const float* W = hd == 0 ? Wa : (hd == 1 ? Wb_ : Wc);
  const float* bias = hd == 0 ? ba : (hd == 1 ? bb : bc);
  // beta per patch, or (reverse sampler: every patch is at the same step) the schedule's entry t, t from the device under graph replay
  const float be = sched_beta ? sched_beta[t_dev ? *t_dev : t] : beta[b];
  const float* wr = W + n * (D + 3) + D;
  tab[(static_cast<int64_t>(hd) * B + b) * D + n] = ((be * wr[0] + sinf(be) * wr[1]) + cosf(be) * wr[2]) + bias[n];
}
int launch_fold_tables(const diffab_dims* d, const diffab_denoiser_weights* w, const float* beta, float* emb_tab, float* beta_tab,
                       hipStream_t st, bool emb_tab_ready, const float* sched_beta, int t, const int* t_dev) {
  if (!emb_tab_ready) {  // weights only: the reverse sampler builds it once per trajectory, not once per step
    hipLaunchKernelGGL(fold_embed_table_kernel, dim3(25, (d->D + 7) / 8), dim3(64), 0, st, w->seq_emb, w->res_w0, w->res_b0, d->D, 25, emb_tab);
    DIFFAB_LAUNCH_CHECK();
  }
  if (beta == nullptr && sched_beta == nullptr) return DIFFAB_OK;
  hipLaunchKernelGGL(fold_beta_table_kernel, dim3(d->B, 3), dim3(d->D), 0, st, beta, w->coord.w0, w->coord.b0, w->orient.w0, w->orient.b0,
                     w->seq.w0, w->seq.b0, d->D, d->B, beta_tab, sched_beta, t, t_dev);
  DIFFAB_LAUNCH_CHECK();
  return DIFFAB_OK;
}

bool fast_path_supported(const diffab_dims* d) {
  return d->D == 128 && d->C == AC && d->H == AH && d->DS == ADS && d->PQ == AP && d->PV == AP && d->K % 64 == 0 && d->K >= 64 &&
         d->K <= 1024;  // any multiple of 64: keys are processed in chunks of 128 (or 64) with an online softmax
}

// ================================================================== fp16 planes of the pair embedding (PLANES attention kernel)
// e s_i = h1 + h2 with h1 = fp16(e s_i), h2 = fp16(e s_i - h1), s_i = the power of two that puts the maximum |e| of PAIR ROW (b, i)
// - the K x 64 values one query residue sees - into [128, 256): two fp16 planes hold e to 2^-23 of its row's maximum in the same
// bytes as fp32.  The scale is per row, not per tensor: an outlier anywhere in the (B, K, K, 64) tensor costs precision only in the
// softmax row that contains it (where it dominates the fp32 sums of the reference just the same); with one scale per tensor a
// single 1e4 x element would leave every other row ~10 good bits.  Layout per (patch, query row i): [key tile jt][plane][k-step
// ks][lane = key % 16 + 16 g][8 channels 32 ks + 8 g ..] - each 1 KiB block is one MFMA A fragment of the bias product, so the
// attention kernel loads fragments with linear 1 KiB wave loads.  Built once per trajectory (the pair embedding does not change
// between reverse steps) or once per call.  Row scales: rs[2 row] = s_i, rs[2 row + 1] = 1 / s_i.
__global__ __launch_bounds__(256) void pair_rowscale_kernel(const float* __restrict__ e, int row_f4, float* __restrict__ rs) {
  const int64_t row = blockIdx.x;
  const f32x4* p = reinterpret_cast<const f32x4*>(e) + row * row_f4;
  float m = 0.f;
  for (int i = threadIdx.x; i < row_f4; i += 256) {
    const f32x4 v = p[i];
    m = fmaxf(fmaxf(m, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
  }
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) m = fmaxf(m, __shfl_xor(m, o));
  __shared__ float wm[4];
  if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned b = __float_as_uint(fmaxf(fmaxf(wm[0], wm[1]), fmaxf(wm[2], wm[3])));  // NaN rows: fmaxf drops NaNs -> scale of the finite part
    const int ex = static_cast<int>((b >> 23) & 255u);
    const bool ok = ex >= 8 && ex < 231 && b < 0x7f800000u;  // zero / below 2^-119 / huge / inf maximum: no scaling (inf / nan propagate as in fp32)
    rs[2 * row] = ok ? __uint_as_float(static_cast<unsigned>(127 + 7 + 127 - ex) << 23) : 1.0f;
    rs[2 * row + 1] = ok ? __uint_as_float(static_cast<unsigned>(ex - 7) << 23) : 1.0f;
  }
}
__global__ void pair_split_kernel(const float* __restrict__ e, const float* __restrict__ rs, int K, int64_t n_groups,
                                  _Float16* __restrict__ out) {
  const int64_t gid = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x;  // (row (b, i), key j, 8-channel group cg)
  if (gid >= n_groups) return;
  const int cg = static_cast<int>(gid & 7);
  const int64_t rj = gid >> 3;
  const int j = static_cast<int>(rj % K);
  const int64_t row = rj / K;
  const float s = rs[2 * row];
  const f32x4 v0 = reinterpret_cast<const f32x4*>(e)[gid * 2], v1 = reinterpret_cast<const f32x4*>(e)[gid * 2 + 1];
  f16x8 h1, h2;
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    const float x = (c < 4 ? v0[c & 3] : v1[c & 3]) * s;
    const _Float16 a = static_cast<_Float16>(x);
    h1[c] = a;
    h2[c] = static_cast<_Float16>(x - static_cast<float>(a));
  }
  const int jt = j >> 4, l15 = j & 15, ks = cg >> 2, g = cg & 3;
  _Float16* base = out + (row * K) * 128 + static_cast<int64_t>(jt) * 2048 + ks * 512 + (l15 + 16 * g) * 8;  // 2048 fp16 per key tile
  *reinterpret_cast<f16x8*>(base) = h1;          // plane 0: blocks (0, ks)
  *reinterpret_cast<f16x8*>(base + 1024) = h2;   // plane 1: blocks (1, ks)
}
bool pair_planes_supported(const diffab_dims* d) { return fast_path_supported(d); }  // any K the fused kernel takes (K % 64 == 0)
size_t pair_planes_floats(const diffab_dims* d) {  // 64 (alignment) | planes | row scales {s, 1 / s} per pair row
  return pair_planes_supported(d) ? static_cast<size_t>(d->B) * d->K * d->K * AC + 64 + 2 * static_cast<size_t>(d->B) * d->K + 64 : 0;
}
// the row scales inside a launch_pair_split() buffer (the planes themselves start at planes + 64)
const float* pair_row_scales(const diffab_dims* d, const float* planes) { return planes + 64 + static_cast<size_t>(d->B) * d->K * d->K * AC; }
// planes: pair_planes_floats(d) floats, 256-byte aligned
int launch_pair_split(const diffab_dims* d, const float* e, float* planes, hipStream_t st) {
  DIFFAB_REQUIRE(pair_planes_supported(d) && e && planes && (reinterpret_cast<uintptr_t>(planes) & 15) == 0 &&
                     (reinterpret_cast<uintptr_t>(e) & 15) == 0,
                 DIFFAB_ERR_ARG, "pair_split: unsupported operands");
  const int64_t n = static_cast<int64_t>(d->B) * d->K * d->K * AC;
  float* rs = const_cast<float*>(pair_row_scales(d, planes));
  hipLaunchKernelGGL(pair_rowscale_kernel, dim3(d->B * d->K), dim3(256), 0, st, e, d->K * AC / 4, rs);
  hipLaunchKernelGGL(pair_split_kernel, dim3(static_cast<unsigned>((n / 8 + 255) / 256)), dim3(256), 0, st, e, rs, d->K, n / 8,
                     reinterpret_cast<_Float16*>(planes + 64));
  DIFFAB_LAUNCH_CHECK();
  return DIFFAB_OK;
}


// DIFFAB_FLAG_FP32_GEMM: the dense projections on the f32-input MFMA kernels of this file instead of the bf16x6 kernels
// (gemm_bf16x6.hip; same results to fp32 rounding) - the plain-fp32 reference path
bool use_b6_gemm(uint32_t flags) { return !(flags & DIFFAB_FLAG_FP32_GEMM); }
static size_t round256(size_t b) { return (b + 255) & ~static_cast<size_t>(255); }
// one layer's prepared weights: [projection planes | to_out planes | small: w_bias 8 x 64, gamma 8 (padded to 64), b_out 128 (fp32)]
// - the small vectors ride along so that the patch-resident module kernel (ipa_persistent.hip) finds everything of layer l at
// base + l * ipa_layer_planes_bytes()
constexpr size_t kLayerSmallFloats = AH * AC + 64 + 128;
size_t ipa_layer_out_planes_offset() { return round256(proj_frames_b6_scratch_bytes()); }
size_t ipa_layer_small_offset() { return ipa_layer_out_planes_offset() + round256(rowgemm128_b6_scratch_bytes(AF)); }
// behind the small vectors: the same two weight sets as two-piece fp16 planes + 1 / scale per output column (gemm_f16x3.hip) - what the
// forward paths use; the bf16 planes in front stay for the A/B switch and for the callers that still hand them to the bf16x6 tiles
size_t ipa_layer_h3_pj_offset() { return ipa_layer_small_offset() + round256(kLayerSmallFloats * sizeof(float)); }
size_t ipa_layer_h3_out_offset() { return ipa_layer_h3_pj_offset() + round256(proj_frames_h3_planes_bytes()); }
size_t ipa_layer_h3_wis_offset() { return ipa_layer_h3_out_offset() + round256(rowgemm128_h3_planes_bytes(AF)); }  // [1344 projections | 128 to_out]
size_t ipa_layer_planes_bytes() { return ipa_layer_h3_wis_offset() + round256((ANP + 128 + 64) * sizeof(float)); }  // + 28 group maxima (value planes)
__global__ void layer_small_copy_kernel(const float* __restrict__ w_bias, const float* __restrict__ gamma, const float* __restrict__ b_out,
                                        float* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < AH * AC) out[i] = w_bias[i];
  else if (i < AH * AC + 64) out[i] = i - AH * AC < AH ? gamma[i - AH * AC] : 0.0f;
  else if (i < static_cast<int>(kLayerSmallFloats)) out[i] = b_out[i - AH * AC - 64];
}
int ipa_layer_split_weights(const diffab_ipa_layer_weights* w, void* planes, hipStream_t st) {
  DIFFAB_REQUIRE(w && w->wq_s && w->wk_s && w->wv_s && w->wq_p && w->wk_p && w->wv_p && w->w_out, DIFFAB_ERR_ARG, "ipa layer: null weight pointer");
  const float* W6[6] = {w->wq_s, w->wk_s, w->wv_s, w->wq_p, w->wk_p, w->wv_p};
  if (int rc = launch_pjsplit(W6, planes, st)) return rc;
  if (w->w_bias && w->gamma && w->b_out) {
    hipLaunchKernelGGL(layer_small_copy_kernel, dim3((kLayerSmallFloats + 255) / 256), dim3(256), 0, st, w->w_bias, w->gamma, w->b_out,
                       reinterpret_cast<float*>(static_cast<char*>(planes) + ipa_layer_small_offset()));
    DIFFAB_LAUNCH_CHECK();
  }
  char* pl = static_cast<char*>(planes);
  float* wis = reinterpret_cast<float*>(pl + ipa_layer_h3_wis_offset());
  if (int rc = launch_pjsplit_h3(W6, pl + ipa_layer_h3_pj_offset(), wis, st)) return rc;
  if (int rc = launch_wsplit128_h3(w->w_out, AF, AF, pl + ipa_layer_h3_out_offset(), wis + ANP, st)) return rc;
  return launch_wsplit128(w->w_out, AF, AF, pl + ipa_layer_out_planes_offset(), st);
}
static size_t b6_scratch_floats() { return (ipa_layer_planes_bytes() + 256) / sizeof(float); }

// workspace of one layer: proj | feat | 128 | three-launch attention's logits (K = 64 / 128) | per-call weight planes | k parts of
// to_out (small batches: launch_rowgemm128_h3p) | operand planes of the logits product (proj_planes.hip) | patch centroids
static size_t ipa_ws_parts_offset(const diffab_dims* d) {
  const size_t rows = static_cast<size_t>(d->B) * d->K;
  const size_t o = rows * (ANP + AF) + 128 + (attention_split_supported(d) ? attention_split_workspace_floats(d) : 0) + b6_scratch_floats();
  return (o + 63) & ~static_cast<size_t>(63);
}
static size_t ipa_ws_operands_offset(const diffab_dims* d) {
  const size_t o = ipa_ws_parts_offset(d) + rowgemm128_h3_parts_floats(d->B * d->K, AF);
  return (o + 63) & ~static_cast<size_t>(63);
}
// ... | value planes + their scales (the P x V operands of the plane attention kernels, written by the h3 projection tile)
static size_t ipa_ws_vpl_offset(const diffab_dims* d) { return ipa_ws_operands_offset(d) + 64; }
size_t ipa_fast_workspace_floats(const diffab_dims* d) {
  const int64_t rows = static_cast<int64_t>(d->B) * d->K;
  return ipa_ws_vpl_offset(d) + proj_value_planes_floats(rows) + proj_value_scales_floats(rows) + 64;
}
// the value planes of a layer workspace: 256-byte aligned planes, the scales behind them
void ipa_ws_value_planes(const diffab_dims* d, float* ws, float** vpl, float** vsc) {
  float* base = ws + ipa_ws_vpl_offset(d);
  *vpl = reinterpret_cast<float*>((reinterpret_cast<uintptr_t>(base) + 255) & ~static_cast<uintptr_t>(255));
  *vsc = *vpl + proj_value_planes_floats(static_cast<int64_t>(d->B) * d->K);
}

// the value side of every patch as fp16 planes (attn_planes_tile.h): one work-group per patch
__global__ __launch_bounds__(512) void attn_value_planes_kernel(const float* __restrict__ proj, const float* __restrict__ t, int K,
                                                                _Float16* __restrict__ vpl, float* __restrict__ osc) {
  extern __shared__ __attribute__((aligned(16))) float ap_lds[];
  aplanes::attn_value_planes_tile(ap_lds, threadIdx.x, blockIdx.x, proj, t, K, vpl, osc);
}
int launch_attn_value_planes(const diffab_dims* d, const float* proj, const float* t, float* vpl, float* vsc, hipStream_t st) {
  DIFFAB_REQUIRE(proj && t && vpl && vsc && d->K % 32 == 0 && fast_path_supported(d), DIFFAB_ERR_ARG, "attn_value_planes: unsupported operands");
  hipLaunchKernelGGL(attn_value_planes_kernel, dim3(d->B), dim3(512), aplanes::LDS_BYTES, st, proj, t, d->K, reinterpret_cast<_Float16*>(vpl), vsc);
  DIFFAB_LAUNCH_CHECK();
  return DIFFAB_OK;
}

int ipa_layer_fast(const diffab_dims* d, const diffab_ipa_layer_weights* w, const float* x, const float* e, const float* R, const float* t,
                   float* y, float* ws, hipStream_t st, float* sp_keep, float* d2_keep, const void* planes, const float* pair_planes,
                   bool fp32_gemm, bool taped, const unsigned char* tile_needed) {
  const int rows = d->B * d->K, D = d->D;
  float* proj = ws;
  float* feat = ws + static_cast<size_t>(rows) * ANP;
  // Dense projections on the bf16 matrix cores (gemm_bf16x6.hip) from split weight planes: the caller's (reverse sampler: split once
  // per trajectory) or, per call, the tail of the workspace.  sp_keep != nullptr (training tape): that workspace slot has no tail,
  // fp32 kernels there.
  const bool b6 = use_b6_gemm(fp32_gemm ? DIFFAB_FLAG_FP32_GEMM : 0u) && (planes != nullptr || sp_keep == nullptr);
  if (b6 && planes == nullptr) {
    float* tail = ws + static_cast<size_t>(rows) * (ANP + AF) + 128 + (attention_split_supported(d) ? attention_split_workspace_floats(d) : 0);
    void* own = reinterpret_cast<void*>((reinterpret_cast<uintptr_t>(tail) + 255) & ~static_cast<uintptr_t>(255));
    if (int rc = ipa_layer_split_weights(w, own, st)) return rc;
    planes = own;
  }
  const void* out_planes = b6 ? static_cast<const char*>(planes) + ipa_layer_out_planes_offset() : nullptr;
  // the three-term fp16 form of the two products (gemm_f16x3.hip) unless the A/B switch asks for the six-term bf16 form
  const bool h3 = b6 && dense_h3_enabled();
  const char* plc = static_cast<const char*>(planes);
  const float* wis = b6 ? reinterpret_cast<const float*>(plc + ipa_layer_h3_wis_offset()) : nullptr;
  // to_out (diffab_pytorch.py:459-464): feat (rows x 1024) Wo^T + b
  auto to_out = [&]() -> int {
    if (h3 && rowgemm128_b6_ok(feat, AF, y, D, rows, AF))
      return launch_rowgemm128_h3p(feat, AF, plc + ipa_layer_h3_out_offset(), wis + ANP, w->b_out, nullptr, 0, y, D, rows, AF, false, st,
                                   taped ? nullptr : ws + ipa_ws_parts_offset(d));
    if (b6 && rowgemm128_b6_ok(feat, AF, y, D, rows, AF))
      // (inference: the projections are dead once the attention has run, their rows take the k parts of a small batch; on the
      // training tape they are kept for the backward)
      return launch_rowgemm128_b6p(feat, AF, out_planes, w->b_out, nullptr, 0, y, D, rows, AF, false, st, taped ? nullptr : proj);
    return launch_linear(feat, AF, w->w_out, w->b_out, y, D, rows, D, AF, false, st);
  };
  // one GEMM for the six projections: Y[:, 0:1344] = x [Wq_s; Wk_s; Wv_s; Wq_p; Wk_p; Wv_p]^T
  LinearSegs segs{};
  segs.W[0] = w->wq_s; segs.W[1] = w->wk_s; segs.W[2] = w->wv_s; segs.W[3] = w->wq_p; segs.W[4] = w->wk_p; segs.W[5] = w->wv_p;
  segs.n_end[0] = 256; segs.n_end[1] = 512; segs.n_end[2] = 768; segs.n_end[3] = 960; segs.n_end[4] = 1152; segs.n_end[5] = 1344;
  segs.nseg = 6;
  bool vec = aligned16(x);
  for (int s = 0; s < 6; ++s) vec = vec && aligned16(segs.W[s]);
  DIFFAB_REQUIRE(vec, DIFFAB_ERR_ARG, "ipa_layer_fast: x and the projection weights must be 16-byte aligned");
  // Value planes (round 6, diffab_debug_set_attn_variant(16)): with the pair planes (inference) a small kernel cuts the value side of the
  // fresh projection rows into fp16 planes (attn_planes_tile.h) and the attention tile's P x V product runs on the f16 matrix cores.
  const bool vpl_on = h3 && !taped && sp_keep == nullptr && pair_planes != nullptr && pair_planes_supported(d) && g_attn_variant != 1 &&
                      d->K % 32 == 0 && g_value_planes;
  float* vpl = nullptr;
  float* vsc = nullptr;
  if (vpl_on) ipa_ws_value_planes(d, ws, &vpl, &vsc);
  if (h3) {
    if (int rc = launch_proj_frames_h3p(x, plc + ipa_layer_h3_pj_offset(), wis, R, t, proj, rows, st)) return rc;
    if (vpl_on)
      if (int rc = launch_attn_value_planes(d, proj, t, vpl, vsc, st)) return rc;
  } else if (b6) {
    if (int rc = launch_proj_frames_b6p(x, planes, R, t, proj, rows, st)) return rc;
  } else {
    if (int rc = launch_proj_frames_f32(x, segs.W, R, t, proj, rows, st)) return rc;  // noslp_kernels.hip
  }
  // Training tape (sp_keep != nullptr, K = 64 / 128): the attention as three launches that leave the probabilities and the squared point
  // distances on the tape for the backward (attention_split.hip); ws has no tail there.  Everything else: the fused kernel.
  if (sp_keep != nullptr && attention_split_supported(d) && !(d->K == 128 && d2_keep != nullptr)) {
    if (int rc = launch_attention_split(d, proj, e, R, t, w->w_bias, w->gamma, feat, sp_keep, st, d2_keep)) return rc;
    return to_out();
  }
  const bool tape = sp_keep != nullptr && attention_split_supported(d);  // K = 128: the fused kernel writes the tape itself
  const int nt = (d->K % 128 == 0) ? 8 : 4;  // key tiles per chunk
  const int nc = d->K / (16 * nt);            // key chunks (online softmax across them)
  const size_t lds = ipa_attn_lds_bytes(nt);
  const dim3 grid(d->B * (d->K / TI));
  // pair_planes (launch_pair_split): the pair-tile products on the f16 matrix cores, the pair stream read as two fp16 planes
  const bool use_planes = pair_planes != nullptr && pair_planes_supported(d);
  const float* e_arg = use_planes ? pair_planes + 64 : e;
  const float* esc = use_planes ? pair_row_scales(d, pair_planes) : nullptr;
#define ATTN_LAUNCH(NT_, MULTI_, PLANES_, VPL_)                                                                                       \
  do {                                                                                                                                \
    DIFFAB_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(ipa_attn_fast_kernel<NT_, MULTI_, PLANES_, false, 8, VPL_>),   \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));                         \
    timer_begin(st);                                                                                                                  \
    hipLaunchKernelGGL((ipa_attn_fast_kernel<NT_, MULTI_, PLANES_, false, 8, VPL_>), grid, dim3(512), lds, st, proj, e_arg, R, t,     \
                       w->w_bias, w->gamma, feat, d->B, nc, g_attn_stamps, esc, nullptr, nullptr, tile_needed,                        \
                       reinterpret_cast<const f32x4*>(vpl), vsc);                                                                     \
    timer_end(st);                                                                                                                    \
  } while (0)
  if (tape) {
    DIFFAB_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(ipa_attn_fast_kernel<8, false, false, true>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
    hipLaunchKernelGGL((ipa_attn_fast_kernel<8, false, false, true>), grid, dim3(512), lds, st, proj, e, R, t, w->w_bias, w->gamma, feat,
                       d->B, nc, nullptr, nullptr, sp_keep, d2_keep);
  } else if (use_planes && g_attn_variant == 1) {
    // four-wave work-groups, two per CU, 64-key chunks (ipa_attn_tile.h, NW = 4)
    const int nc4 = d->K / 64;
    const size_t lds4 = ipa_attn_lds_bytes(4, 4, true);
    DIFFAB_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(ipa_attn_fast_kernel<4, true, true, false, 4>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds4)));
    timer_begin(st);
    hipLaunchKernelGGL((ipa_attn_fast_kernel<4, true, true, false, 4>), grid, dim3(256), lds4, st, proj, e_arg, R, t, w->w_bias, w->gamma, feat,
                       d->B, nc4, g_attn_stamps, esc, nullptr, nullptr, tile_needed);
    timer_end(st);
  } else if (vpl_on) {
    if (nt == 8 && nc == 1) ATTN_LAUNCH(8, false, true, true);
    else if (nt == 8) ATTN_LAUNCH(8, true, true, true);
    else if (nc == 1) ATTN_LAUNCH(4, false, true, true);
    else ATTN_LAUNCH(4, true, true, true);
  } else if (use_planes) {
    if (nt == 8 && nc == 1) ATTN_LAUNCH(8, false, true, false);
    else if (nt == 8) ATTN_LAUNCH(8, true, true, false);
    else if (nc == 1) ATTN_LAUNCH(4, false, true, false);
    else ATTN_LAUNCH(4, true, true, false);
  } else if (nt == 8 && nc == 1) ATTN_LAUNCH(8, false, false, false);
  else if (nt == 8) ATTN_LAUNCH(8, true, false, false);
  else if (nc == 1) ATTN_LAUNCH(4, false, false, false);
  else ATTN_LAUNCH(4, true, false, false);
#undef ATTN_LAUNCH
  DIFFAB_LAUNCH_CHECK();
  return to_out();
}

}  // namespace diffab
