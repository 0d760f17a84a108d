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

namespace diffab {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// Compile-time fence for memory operations: keeps the hand-placed prefetch loads where they are written (hipcc otherwise
// sinks each load next to its first use, leaving one or two in flight and exposing every HBM / L2 round trip).
#define MEM_FENCE() asm volatile("" ::: "memory")

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
#ifndef RG_MT
#define RG_MT 2  // 16-row MFMA tiles per wave: 2 -> 128 rows per work-group (1 -> 64 rows, two work-groups per CU: measured slower)
#endif
constexpr int RG_ROWS = 64 * RG_MT;
#ifndef RG_DEEP_A
#define RG_DEEP_A 0  // 1: A register sets span two weight chunks (deeper prefetch) - measured SLOWER (101 vs 88 us at K = 1024)
#endif

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
constexpr int AH = 8, ADS = 32, AP = 8, AC = 64;
constexpr int ANP = 3 * AH * ADS + 3 * AH * AP * 3;               // 1344 projection columns
constexpr int AF = AH * ADS + AH * AC + AH * AP * 3 + AH * AP;    // 1024 feature columns
constexpr int OFF_QS = 0, OFF_KS = 256, OFF_VS = 512, OFF_GQ = 768, OFF_GK = 960, OFF_GV = 1152;
constexpr int FOFF_OS = 0, FOFF_OE = 256, FOFF_OL = 768, FOFF_ON = 960;
constexpr int TI = 16;  // query residues per work-group

// LDS strides of the logits/probabilities image: head stride K + 8 (== 8 mod 64 for K % 64 == 0) and row stride
// 8 (K + 8) + 8 keep both the (head, quarter)-lane and the (row, quarter)-lane ds_read_b128 patterns conflict-free.

// softmax exponentials: v_exp_f32 path (2^(x log2 e)); arguments are <= 0 and the relative error (<~ |x| 1e-7) is far inside the
// 1e-4 parity bar (measured ~2e-6 on the outputs).  -DDIFFAB_ACCURATE_EXP restores the libm expf expansion (~12 VALU ops each).
#ifdef DIFFAB_ACCURATE_EXP
#define FAST_EXP(x) expf(x)
#else
#define FAST_EXP(x) __expf(x)
#endif

// NT: key tiles (16 keys each) per chunk: 8 when K % 128 == 0, else 4; compile-time so per-lane arrays stay in VGPRs.
// MULTI: more than one chunk.  The single-chunk instantiation (K = 64, 128) has NC == 1 at compile time: the chunk loop and every
// rescale branch fold away and it is the same straight-line kernel as before the chunk loop existed (the loop costs 13 % at K=128).
// PLANES: `e` is not the fp32 pair embedding but its two-plane fp16 image written by pair_split_kernel (same
// bytes: e s = h1 + h2 to 2^-23 of the tensor maximum, fragment order of the bias product), `esc` = {s, 1 / s}; the two products on
// the pair tile then run on the f16 matrix cores as three exact partial products each (h1 w1, h1 w2, h2 w1 with fp32 accumulation)
// instead of f32 MFMAs: 96 instead of 512 matrix-pipe cycles per key tile for the bias, 768 instead of 4096 per row for o_e.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4_t __attribute__((ext_vector_type(4)));
typedef short s16x8_t __attribute__((ext_vector_type(8)));
// TAPE (single-chunk, fp32 pair stream: the forward of a training step): the normalised probabilities and the squared point distances
// are left in tape_p / tape_d2 ([b][h][i][j], what ipa_logits_kernel<true> + ipa_pair_stream_kernel leave) for the backward.
template <int NT, bool MULTI, bool PLANES = false, bool TAPE = false>
__global__ __launch_bounds__(512) void ipa_attn_fast_kernel(const float* __restrict__ proj, const float* __restrict__ e,
                                                            const float* __restrict__ R, const float* __restrict__ t,
                                                            const float* __restrict__ Wb, const float* __restrict__ gamma,
                                                            float* __restrict__ feat, int B, int NC_arg,
                                                            unsigned long long* __restrict__ stamps, const float* __restrict__ esc = nullptr,
                                                            float* __restrict__ tape_p = nullptr, float* __restrict__ tape_d2 = nullptr,
                                                            const unsigned char* __restrict__ tile_needed = nullptr) {
  static_assert(!PLANES || NT % 2 == 0, "the o_e product takes key tiles in pairs");
  static_assert(!TAPE || (!MULTI && !PLANES), "the tape form is the single-chunk fp32-pair kernel");
  const int NC = MULTI ? NC_arg : 1;
  // Keys are processed in NC chunks of KC = 16 NT with an online softmax: the LDS image holds the logits / (unnormalised)
  // probabilities of ONE chunk, each (row, head) keeps a running maximum M and sum L, and the partial outputs of earlier chunks
  // are rescaled by exp(M_old - M_new) through the feature rows in global memory.  K = 64 and 128 are the single-chunk case;
  // K = 192, 256, ... reuse the same 16-row structure instead of needing a K-proportional LDS image.
  extern __shared__ __attribute__((aligned(16))) float S[];  // [TI][AH][KC+8] (+8 per i)
  constexpr int KC = NT * 16;
  constexpr int NS = NT * 4;  // (jt, r) key steps of 4 keys each
  const int K = NC * KC;
  const int ntile = K / TI;
  // diagnostic stamps (stamps == nullptr in every production launch: nothing below executes; diffab_debug_set_attn_stamps,
  // tools/attn_phase_profile.py).  -DAT_STAMP_REALTIME: the chip-wide 100 MHz counter instead of the per-CU cycle counter.
  auto stamp = [&](int k) {
    if (stamps != nullptr) {
      __builtin_amdgcn_sched_barrier(0);
#ifdef AT_STAMP_REALTIME
      const unsigned long long tnow = __builtin_amdgcn_s_memrealtime();
#else
      const unsigned long long tnow = __builtin_amdgcn_s_memtime();
#endif
      if ((threadIdx.x & 63) == 0) stamps[(static_cast<size_t>(blockIdx.x) * 8 + (threadIdx.x >> 6)) * 8 + k] = tnow;
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  stamp(0);
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
  const int i0 = tile * TI;
  const int tid = threadIdx.x, lane0 = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  // (the wave index as a scalar: the addresses built from it stay in SGPRs, which takes the chunked instantiations from 13 spilled VGPRs
  // + 56 B of scratch to none)
  constexpr int HS = KC + 8, IS = AH * (KC + 8) + 8;  // == 8 (mod 64): both ds_read_b128 patterns on the image are conflict-free
  const int64_t prow0 = static_cast<int64_t>(b) * K;  // first projection row of this patch
  const float scale_t = 0.57735026918962576f;         // 3^-1/2   (diffab_pytorch.py:387, :439)

  // Per-wave LDS scratch behind the logits image.  Every global load below is issued in full 128-byte lines (consecutive
  // lanes on consecutive 16-byte chunks): the texture addresser serves one line per lane quad, whereas MFMA-fragment-shaped
  // loads (adjacent lanes on different rows) cost four lines per quad and made this kernel addresser-bound (4x the issue
  // time, measured).  Data is re-oriented into fragments through this scratch; a wave reads only what it wrote, and LDS
  // operations of one wave complete in order, so no barrier is involved.
  constexpr int KLD = 40, GLD = 28;             // key-tile strides (floats): ds_read_b128 conflict-free
  constexpr int P1_TILE = 16 * KLD + 16 * GLD;  // 1088 floats per staged key tile
  constexpr int ELD = 72;                       // pair-tile stride (floats)
  constexpr int SCR_FLOATS = 2 * 16 * ELD;      // 2304 floats per wave (>= 2 * P1_TILE = 2176)
#ifndef DIFFAB_E_EARLY
#define DIFFAB_E_EARLY 1
#endif
#ifndef DIFFAB_E_LAG
#define DIFFAB_E_LAG 2
#endif
  constexpr int E_LAG = MULTI ? DIFFAB_E_LAG : 0;    // the next row's tile loads trail the retiring tiles by this many (VGPRs)
#ifndef DIFFAB_E_EARLY_SINGLE
#define DIFFAB_E_EARLY_SINGLE 2
#endif
  constexpr int E_EARLY = MULTI ? DIFFAB_E_EARLY : DIFFAB_E_EARLY_SINGLE;  // pair tiles of phase 2's first row started under the tail of phase 1
  float* scr = S + TI * IS + wv * SCR_FLOATS;
  float* st_fac = S + TI * IS + 8 * SCR_FLOATS;  // [TI][AH] exp(M_old - M_new) of the current chunk
  float* st_inv = st_fac + TI * AH;              // [TI][AH] 1 / L after the last chunk (1 before)
  float* wb_lds = st_inv + TI * AH;              // [4 sg][64 lanes][4]: B fragments of the bias product (same for every wave)
  if (wv == 0 && !PLANES) {  // Wb[h][16 sg + 4 q + s] for lane (h = l15 < 8, q), zero in the padding columns; first read is behind a barrier
    const int l15_ = lane0 & 15, q_ = lane0 >> 4;
#pragma unroll
    for (int sg = 0; sg < 4; ++sg) {
      f32x4 v = *reinterpret_cast<const f32x4*>(Wb + (l15_ & 7) * AC + 16 * sg + 4 * q_);
#pragma unroll
      for (int s_ = 0; s_ < 4; ++s_) v[s_] = l15_ < 8 ? v[s_] : 0.0f;
      *reinterpret_cast<f32x4*>(wb_lds + (sg * 64 + lane0) * 4) = v;
    }
  }

  // e[b, i0 + 2 wv + ii, :, :]: the two pair-embedding rows this wave owns in phase 2.  Streamed once (non-temporal: it
  // must not evict the K/V-side operands, re-read by the other row tiles of the patch, from L2), in the orientation of
  // the o_e product: lane (l15, q) holds e[i][j = 16 jt + 4 q + r][c = 4 l15 .. 4 l15 + 3] - 1 KiB contiguous per load.
  // A chunk of a row (16 NT VGPRs) stays in registers from the bias product to the o_e product.
  const float* erow[2];
  erow[0] = e + ((prow0 + i0 + 2 * wv) * K) * AC;
  erow[1] = erow[0] + static_cast<int64_t>(K) * AC;
  float Mrun[2] = {-INFINITY, -INFINITY}, Lrun[2] = {0.f, 0.f};  // online-softmax state of (row 2 wv + ii, head l15 & 7)
  // PLANES: 1 / s_i of the wave's two pair rows, fetched here through the scalar cache (wave-uniform address).  As a vector load at
  // the top of each row its s_waitcnt - vmcnt retires in order - drained every pair tile in flight, twice per wave and phase 2.
  float inv_s2[2] = {1.0f, 1.0f};
  if constexpr (PLANES) {
    const float* ep = esc + 2 * (prow0 + i0 + 2 * __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6)));
    inv_s2[0] = ep[1];
    inv_s2[1] = ep[3];
  }

#pragma unroll 1
  for (int c = 0; c < NC; ++c) {
    const bool last = c == NC - 1;
    const int64_t krow0 = prow0 + c * KC;  // first key row of this chunk
    // Re-derive the lane coordinates from an opaque copy each iteration: otherwise every lane-constant address of the three
    // phases is hoisted out of this loop and stays live through phase 2, which spills (hipcc, ROCm 7.2).
    int lane = lane0;
    asm volatile("" : "+v"(lane));
    const int l15 = lane & 15, q = lane >> 4;
    f32x4 ev[2][NT][4];
    f32x4 wv4[2][2];  // PLANES: the bias weights of lane (head, channel group), requested in phase 1's tail AHEAD of the first pair tiles:
                      // vmcnt retires in order, so loaded behind them they would cost every wave a pair-tile latency in front of the barrier
    auto load_e_tile = [&](int ii, int cc_, int jt) {
      if constexpr (PLANES) {  // four 1 KiB blocks per key tile, lane order: ev[ii][jt][2 p + ks] = fragment (plane p, k-step ks)
        const f32x4* ep = reinterpret_cast<const f32x4*>(erow[ii]) + (cc_ * NT + jt) * 256 + lane;
#pragma unroll
        for (int r = 0; r < 4; ++r) ev[ii][jt][r] = __builtin_nontemporal_load(ep + r * 64);
      } else {
        const f32x4* ep = reinterpret_cast<const f32x4*>(erow[ii] + (cc_ * KC + jt * 16 + 4 * q) * AC + 4 * l15);
#pragma unroll
        for (int r = 0; r < 4; ++r) ev[ii][jt][r] = __builtin_nontemporal_load(ep + r * (AC / 4));
      }
    };
    // ---------------------------------------------------------------- phase 1: wave = head
    {
      const int h = wv;
      const float scale_s = 0.17677669529663687f;                    // 32^-1/2  (:353)
      const float coef_p = -0.5f * 0.16666666666666666f * gamma[h];  // -1/2 (4.5*8)^-1/2 gamma_h  (:372, :431-436)
      // line-shaped loads of one key tile (16 keys): k_s 16 x 128 B (8 lanes per key), gk 16 x 96 B (6 lanes per key)
      const float* ks_src = proj + (krow0 + (lane >> 3)) * ANP + OFF_KS + h * ADS + 4 * (lane & 7);  // + 8 t keys, + 16 jt keys
      const int g0 = lane, g1 = lane + 64;  // gk chunk ids (0..95): key = id / 6, chunk = id % 6
      const float* gk_src0 = proj + (krow0 + g0 / 6) * ANP + OFF_GK + h * 24 + 4 * (g0 % 6);
      const float* gk_src1 = proj + (krow0 + g1 / 6) * ANP + OFF_GK + h * 24 + 4 * (g1 % 6);
      const int ks_dst = (lane >> 3) * KLD + 4 * (lane & 7);
      const int gk_dst0 = 16 * KLD + (g0 / 6) * GLD + 4 * (g0 % 6), gk_dst1 = 16 * KLD + (g1 / 6) * GLD + 4 * (g1 % 6);
      constexpr int SD = MULTI ? 3 : 4;  // register staging depth: SD - 1 key tiles of lookahead
      static_assert(E_EARLY <= SD, "the early pair tiles are requested in the last E_EARLY iterations, which must not request key tiles any more");
      f32x4 st[SD][4];
      auto load_keys = [&](int sb, int jt) {
        const int64_t o = static_cast<int64_t>(jt) * 16 * ANP;
        st[sb][0] = *reinterpret_cast<const f32x4*>(ks_src + o);
        st[sb][1] = *reinterpret_cast<const f32x4*>(ks_src + o + 8 * ANP);
        st[sb][2] = *reinterpret_cast<const f32x4*>(gk_src0 + o);
        if (g1 < 96) st[sb][3] = *reinterpret_cast<const f32x4*>(gk_src1 + o);
      };
      auto stage_keys = [&](int sb, int lb) {
        float* t_ = scr + lb * P1_TILE;
        *reinterpret_cast<f32x4*>(t_ + ks_dst) = st[sb][0];
        *reinterpret_cast<f32x4*>(t_ + ks_dst + 8 * KLD) = st[sb][1];
        *reinterpret_cast<f32x4*>(t_ + gk_dst0) = st[sb][2];
        if (g1 < 96) *reinterpret_cast<f32x4*>(t_ + gk_dst1) = st[sb][3];
      };
#pragma unroll
      for (int jt = 0; jt < SD && jt < NT; ++jt) load_keys(jt, jt);
      // A operand: q_s rows i0 + l15, k = 16 sg + 4 q + s
      f32x4 qa[2];
      const float* qrow = proj + (prow0 + i0 + l15) * ANP + OFF_QS + h * ADS + 4 * q;
      qa[0] = *reinterpret_cast<const f32x4*>(qrow);
      qa[1] = *reinterpret_cast<const f32x4*>(qrow + 16);
      // query points of the 4 rows this lane accumulates (rows i0 + 4q + r); the 16 lanes of a quarter share each address
      f32x4 gq[4][6];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float* p = proj + (prow0 + i0 + 4 * q + r) * ANP + OFF_GQ + h * 24;
#pragma unroll
        for (int cc = 0; cc < 6; ++cc) gq[r][cc] = *reinterpret_cast<const f32x4*>(p + 4 * cc);
      }
      MEM_FENCE();
      // Software pipeline over the key tiles (fully unrolled, so the fragment sets are renamed, not copied):
      //   iteration jt:  registers -> LDS for tile jt+2 | LDS -> fragments of tile jt+1 | global -> registers for tile jt+1+SD-1
      //                  | MFMA + VALU on the fragments of tile jt (read during iteration jt-1).
      // Every LDS round trip and every L2 round trip is a full iteration (or SD-2 of them) old when its data is needed; before
      // this the fragment reads of tile jt sat right behind the writes of tile jt+1 and were waited for at once.
      struct KeyFrag { f32x4 kb0, kb1, gk[6]; };
      auto read_frags = [&](int jt) {
        KeyFrag f;
        const float* t_ = scr + (jt & 1) * P1_TILE;
        f.kb0 = *reinterpret_cast<const f32x4*>(t_ + l15 * KLD + 4 * q);  // k_s[16 jt + l15][16 sg + 4 q + s]
        f.kb1 = *reinterpret_cast<const f32x4*>(t_ + l15 * KLD + 16 + 4 * q);
#pragma unroll
        for (int cc = 0; cc < 6; ++cc) f.gk[cc] = *reinterpret_cast<const f32x4*>(t_ + 16 * KLD + l15 * GLD + 4 * cc);
        return f;
      };
      stage_keys(0, 0);
      KeyFrag cur = read_frags(0);
      if (NT > 1) stage_keys(1 % SD, 1);
#pragma unroll
      for (int jt = 0; jt < NT; ++jt) {
        if (jt + 2 < NT) stage_keys((jt + 2) % SD, jt & 1);  // tile jt+2 -> the buffer tile jt was read from (LDS ops retire in order)
        KeyFrag nxt = cur;
        if (jt + 1 < NT) nxt = read_frags(jt + 1);
        if (jt + SD < NT) {
          load_keys(jt % SD, jt + SD);  // slot of tile jt (staged two iterations ago)
        } else if (jt + E_EARLY >= NT) {
          if constexpr (PLANES) {
            if (jt + E_EARLY == NT) {
              const int hh = lane & 7, qq = lane >> 4;
#pragma unroll
              for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int hf = 0; hf < 2; ++hf) wv4[ks][hf] = *reinterpret_cast<const f32x4*>(Wb + hh * AC + 32 * ks + 8 * qq + 4 * hf);
            }
          }
          load_e_tile(0, c, jt + E_EARLY - NT);  // key stream done: start phase 2's pair-embedding stream under this tile
        }
        MEM_FENCE();
        const f32x4 kb0 = cur.kb0, kb1 = cur.kb1;
        f32x4 gk[6];
#pragma unroll
        for (int cc = 0; cc < 6; ++cc) gk[cc] = cur.gk[cc];
        cur = nxt;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(qa[0][s], kb0[s], acc, 0, 0, 0);
#pragma unroll
        for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(qa[1][s], kb1[s], acc, 0, 0, 0);
        // acc[r] = q_s[i0+4q+r] . k_s[key 16jt+l15]
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          // packed fp32 (v_pk_add_f32 / v_pk_fma_f32): two coordinates per instruction, two partial sums added at the end
          f32x2 d2v = {0.f, 0.f};
#pragma unroll
          for (int cc = 0; cc < 6; ++cc) {
            // packed subtract spelled in assembly: the compiler splits a vector fsub (and fma(b, -1, a)) into two v_sub_f32
            f32x2 dlo, dhi;
            asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]"
                : "=v"(dlo)
                : "v"(__builtin_shufflevector(gq[r][cc], gq[r][cc], 0, 1)), "v"(__builtin_shufflevector(gk[cc], gk[cc], 0, 1)));
            asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]"
                : "=v"(dhi)
                : "v"(__builtin_shufflevector(gq[r][cc], gq[r][cc], 2, 3)), "v"(__builtin_shufflevector(gk[cc], gk[cc], 2, 3)));
            d2v = __builtin_elementwise_fma(dlo, dlo, d2v);
            d2v = __builtin_elementwise_fma(dhi, dhi, d2v);
          }
          const float d2 = d2v[0] + d2v[1];
          S[(4 * q + r) * IS + h * HS + jt * 16 + l15] = scale_t * (acc[r] * scale_s + coef_p * d2);
          if constexpr (TAPE) tape_d2[((static_cast<int64_t>(b) * AH + h) * K + i0 + 4 * q + r) * K + jt * 16 + l15] = d2;
        }
        if (c == 0 && jt == 0) stamp(6);
        if (c == 0 && jt == 3) stamp(7);
      }
    }
    if (c == 0) stamp(1);
    // ---------------------------------------------------------------- phase 2: wave = 2 query rows, lanes = (head, key quarter)
    {
      const int h = l15 & 7;  // lanes with l15 >= 8 shadow head l15-8 (their MFMA columns are padding)
#ifndef DIFFAB_E_DEPTH0
#define DIFFAB_E_DEPTH0 3  // tiles of the first row in flight before its bias loop starts; the rest follow one per consumed tile (all 8 at once:
                           // 256 KiB per CU requested in one burst, +3.5 % kernel time: the queue it builds delays every other CU's loads)
#endif
      // PLANES: RT = pair tiles of the wave's 2 NT-tile stream held in registers (requested RT tiles ahead of their use): half a row.
      // (A whole row spills in the chunked kernel, and in the single-chunk one its 24 loads per wave in front of the barrier take
      // 4.6 k cycles to issue on the waves that finish phase 1 last: 0.326 ms against 0.321 with half a row.)
      constexpr int RT = NT / 2;
      constexpr int E_DEPTH0 = PLANES ? RT : (DIFFAB_E_DEPTH0 < NT ? (DIFFAB_E_DEPTH0 > E_EARLY ? DIFFAB_E_DEPTH0 : E_EARLY) : NT);
      if constexpr (!PLANES) {
#pragma unroll
        for (int jt = E_EARLY; jt < E_DEPTH0; ++jt) load_e_tile(0, c, jt);  // the first E_EARLY tiles were started under phase 1's tail
      }
      f32x4 wb[4];  // single-chunk kernel: bias B fragments in registers; multi-chunk: read from LDS per tile (VGPR pressure)
      f16x8 wp[2][2];  // PLANES: bias B fragments as two fp16 planes, wp[plane][ks]: lane (head l15, channels 32 ks + 8 q ..), scaled by sw
      float bscale = scale_t, oscale = 1.0f;
      if constexpr (PLANES) {
        float wmax = 0.f;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
          for (int hf = 0; hf < 2; ++hf)
#pragma unroll
            for (int s = 0; s < 4; ++s) wmax = fmaxf(wmax, fabsf(wv4[ks][hf][s]));
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) wmax = fmaxf(wmax, __shfl_xor(wmax, o));
        // sw = 2^(7 - exponent(wmax)): the largest weight lands in [128, 256), far from fp16's subnormals and its overflow
        const int ew = static_cast<int>((__float_as_uint(wmax) >> 23) & 255u);
        const float sw = (ew == 0 || ew > 230) ? 1.0f : __uint_as_float(static_cast<unsigned>(127 + 7 + 127 - ew) << 23);
        const float isw = (ew == 0 || ew > 230) ? 1.0f : __uint_as_float(static_cast<unsigned>(ew - 7) << 23);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
          for (int c8 = 0; c8 < 8; ++c8) {
            const float x = l15 < 8 ? wv4[ks][c8 >> 2][c8 & 3] * sw : 0.0f;
            const _Float16 h1 = static_cast<_Float16>(x);
            wp[0][ks][c8] = h1;
            wp[1][ks][c8] = static_cast<_Float16>(x - static_cast<float>(h1));
          }
        bscale = scale_t * isw;                   // logits: bias = (sum e s_i w sw) / (s_i sw), x 1 / s_i per row below
        oscale = 1.0f / 256.0f;                   // o_e: probabilities enter scaled by 256
        // the rest of the first row, requested AFTER the weight loads above have been consumed: vmcnt retires in order, a wait for a
        // load issued behind these tiles would wait for all of them (measured: 9 k cycles in front of the barrier)
        asm volatile("" ::"v"(wp[0][0]), "v"(wp[1][0]), "v"(wp[0][1]), "v"(wp[1][1]));
#pragma unroll
        for (int jt = E_EARLY; jt < E_DEPTH0; ++jt) load_e_tile(0, c, jt);
      } else if constexpr (!MULTI) {
#pragma unroll
        for (int sg = 0; sg < 4; ++sg) {
          wb[sg] = *reinterpret_cast<const f32x4*>(Wb + h * AC + 16 * sg + 4 * q);
#pragma unroll
          for (int s = 0; s < 4; ++s) wb[sg][s] = l15 < 8 ? wb[sg][s] : 0.0f;
        }
      }
      MEM_FENCE();
      __syncthreads();  // phase-1 logits of all heads are in LDS (and every wave is done with its key-tile scratch)
      if (c == 0) stamp(2);
      // tile re-orientation for the bias product: write [key 4 q + r][channel chunk l15], read [key l15][channels 16 sg + 4 q ..]
      auto stage_e = [&](int ii, int jt) {
        float* t_ = scr + (jt & 1) * (16 * ELD) + 4 * q * ELD + 4 * l15;
#pragma unroll
        for (int r = 0; r < 4; ++r) *reinterpret_cast<f32x4*>(t_ + r * ELD) = ev[ii][jt][r];
      };

      if constexpr (PLANES) {
        // Key tiles are consumed in pairs (32 keys), each pair completely - bias, softmax bookkeeping, o_e - as soon as it is in
        // registers (an online softmax inside the row), so a tile's registers are free after ONE use and the next row's tiles are
        // requested eight tiles ahead of their use from the first step on: the pair stream of the wave's two rows is one continuous
        // pipeline.  (Two passes per row - all bias products, softmax, all o_e products - left the second row's loads exposed once
        // the products had moved to the f16 matrix cores: phase 2 had become a wait for HBM latency, 38 k of its 20 k cycles.)
        // The probabilities go to LDS relative to the running maximum of their step and are rescaled to the row maximum after the row.
        char* trt = reinterpret_cast<char*>(scr);                                    // [2 tiles][2 planes][16 keys][128 bytes] = 8 KiB
        const int wr_off = l15 * 128 + 8 * ((2 * q) ^ (4 * ((l15 >> 1) & 3)));      // ^ 64 ks: 8-byte unit 8 ks + 2 q of row l15
        const int rrow = 4 * q + (l15 >> 2);
        const int rd_off = rrow * 128 + 8 * ((l15 & 3) ^ (4 * ((rrow >> 1) & 3)));  // ^ 32 ct: unit 4 ct + (l15 & 3) of row rrow
#pragma unroll
        for (int ii = 0; ii < 2; ++ii) {
          const int il = 2 * wv + ii;  // local row
          float* Srow = S + il * IS + h * HS;
          const float inv_s = inv_s2[ii];  // 1 / s_i: the power-of-two scale of this pair row's planes
          const float bscale_r = bscale * inv_s, oscale_r = oscale * inv_s;
          float m_run = -INFINITY, l_run = 0.f, m_hist[NT / 2];
          f32x4 oe[4];
#pragma unroll
          for (int ct = 0; ct < 4; ++ct) oe[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int T = 0; T < NT / 2; ++T) {
            // ---- bias of the two tiles: A fragments straight from the loaded registers (lane = key l15, channels 32 ks + 8 q ..)
            f32x4 acc[2][2];
#pragma unroll
            for (int tl = 0; tl < 2; ++tl)
#pragma unroll
              for (int ks = 0; ks < 2; ++ks) {
                const f16x8 a1 = __builtin_bit_cast(f16x8, ev[ii][2 * T + tl][ks]), a2 = __builtin_bit_cast(f16x8, ev[ii][2 * T + tl][2 + ks]);
                f32x4 a_ = {0.f, 0.f, 0.f, 0.f};
                a_ = __builtin_amdgcn_mfma_f32_16x16x32_f16(a2, wp[0][ks], a_, 0, 0, 0);
                a_ = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, wp[1][ks], a_, 0, 0, 0);
                a_ = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, wp[0][ks], a_, 0, 0, 0);
                acc[tl][ks] = a_;
              }
            float v[8], smax = -INFINITY;
#pragma unroll
            for (int tl = 0; tl < 2; ++tl) {
              const f32x4 sv = *reinterpret_cast<const f32x4*>(Srow + (2 * T + tl) * 16 + 4 * q);
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                v[4 * tl + r] = sv[r] + bscale_r * (acc[tl][0][r] + acc[tl][1][r]);
                smax = fmaxf(smax, v[4 * tl + r]);
              }
            }
            smax = fmaxf(smax, __shfl_xor(smax, 16));
            smax = fmaxf(smax, __shfl_xor(smax, 32));
            const float m_new = fmaxf(m_run, smax);
            const float alpha = T == 0 ? 0.0f : FAST_EXP(m_run - m_new);
            m_run = m_new;
            m_hist[T] = m_new;
            float psum = 0.f;
#pragma unroll
            for (int tt = 0; tt < 8; ++tt) {
              v[tt] = FAST_EXP(v[tt] - m_new);
              psum += v[tt];
            }
            l_run = l_run * alpha + psum;  // lane-partial; the key quarters are added after the row (alpha is the same in all four)
            if (T > 0) {
#pragma unroll
              for (int ct = 0; ct < 4; ++ct)
#pragma unroll
                for (int r = 0; r < 4; ++r) oe[ct][r] *= alpha;
            }
            // ---- probabilities: to LDS for phase 3 (relative to m_hist[T]), and as two fp16 planes (x 256) into the o_e product
            f16x8 p1, p2;
#pragma unroll
            for (int tt = 0; tt < 8; ++tt) {
              const float x = 256.0f * v[tt];
              const _Float16 hh = static_cast<_Float16>(x);
              p1[tt] = hh;
              p2[tt] = static_cast<_Float16>(x - static_cast<float>(hh));
            }
#pragma unroll
            for (int tl = 0; tl < 2; ++tl) {
              if (l15 < 8) *reinterpret_cast<f32x4*>(Srow + (2 * T + tl) * 16 + 4 * q) = f32x4{v[4 * tl], v[4 * tl + 1], v[4 * tl + 2], v[4 * tl + 3]};
#pragma unroll
              for (int pl = 0; pl < 2; ++pl)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
                  *reinterpret_cast<f32x4*>(trt + (tl * 2 + pl) * 2048 + (wr_off ^ (64 * ks))) = ev[ii][2 * T + tl][2 * pl + ks];
            }
            {  // the two tiles are in LDS: request the tiles RT ahead in the wave's stream (rest of this row, then the next row)
              constexpr int dummy_ = 0;
              (void)dummy_;
              const int nx = ii * NT + 2 * T + RT;  // compile-time after unrolling
              if (nx < 2 * NT) {
                load_e_tile(nx / NT, c, nx % NT);
                load_e_tile((nx + 1) / NT, c, (nx + 1) % NT);
                MEM_FENCE();
              }
            }
            // ---- o_e[channel][head] += e^T P: the A operand (8 keys per lane for one channel) through the transposing LDS read
            f16x8 a[2][4];
#pragma unroll
            for (int pl = 0; pl < 2; ++pl)
#pragma unroll
              for (int ct = 0; ct < 4; ++ct) {
                const int ro = rd_off ^ (32 * ct);
                const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4_t __attribute__((address_space(3)))*)(trt + (0 * 2 + pl) * 2048 + ro));
                const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4_t __attribute__((address_space(3)))*)(trt + (1 * 2 + pl) * 2048 + ro));
                const s16x8_t v8 = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                a[pl][ct] = __builtin_bit_cast(f16x8, v8);
              }
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) oe[ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[1][ct], p1, oe[ct], 0, 0, 0);
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) oe[ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[0][ct], p2, oe[ct], 0, 0, 0);
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) oe[ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[0][ct], p1, oe[ct], 0, 0, 0);
          }
          l_run += __shfl_xor(l_run, 16);
          l_run += __shfl_xor(l_run, 32);
          // this chunk's (m_run, l_run) joins the row's running (M, L) over the chunks: everything of this chunk is scaled by cf,
          // everything accumulated before it by fac (single chunk: cf = 1, fac = 0)
          const float Mnew = MULTI ? fmaxf(Mrun[ii], m_run) : m_run;
          const float fac = (!MULTI || c == 0) ? 0.0f : expf(Mrun[ii] - Mnew);
          const float cf = MULTI ? expf(m_run - Mnew) : 1.0f;
          Mrun[ii] = Mnew;
          Lrun[ii] = Lrun[ii] * fac + l_run * cf;
          const float inv = last ? 1.0f / Lrun[ii] : 1.0f;
          // the probabilities of a step are relative to the running maximum of that step: rescale to the row maximum so far
#pragma unroll
          for (int T = 0; T < NT / 2 - (MULTI ? 0 : 1); ++T) {
            const float f = FAST_EXP(m_hist[T] - Mnew);
            if (l15 < 8) {
#pragma unroll
              for (int tl = 0; tl < 2; ++tl) {
                f32x4* sp = reinterpret_cast<f32x4*>(Srow + (2 * T + tl) * 16 + 4 * q);
                f32x4 pv = *sp;
#pragma unroll
                for (int r = 0; r < 4; ++r) pv[r] *= f;
                *sp = pv;
              }
            }
          }
          // D: column = head l15, row 4 q + r <-> channel 16 ct + 4 q + r
          if (l15 < 8) {
            float* fo = feat + (prow0 + i0 + il) * AF + FOFF_OE + h * AC + 4 * q;
            const float sc = cf * oscale_r;
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) {
              f32x4 o = oe[ct];
#pragma unroll
              for (int r = 0; r < 4; ++r) o[r] *= sc;
              if (MULTI && c > 0) {
                const f32x4 old = *reinterpret_cast<const f32x4*>(fo + 16 * ct);
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] += old[r] * fac;
              }
#pragma unroll
              for (int r = 0; r < 4; ++r) o[r] *= inv;
              *reinterpret_cast<f32x4*>(fo + 16 * ct) = o;
            }
            if (q == 0) {
              st_fac[il * AH + h] = fac;
              st_inv[il * AH + h] = inv;
            }
          }
        }
      } else {
#pragma unroll
      for (int ii = 0; ii < 2; ++ii) {
        const int il = 2 * wv + ii;  // local row
        float* Srow = S + il * IS + h * HS;
        float lg[NT][4];  // logits, then exp(logit - M), of keys j = 16 jt + 4 q + r of this chunk for head h
        float mx = -INFINITY;
        stage_e(ii, 0);
#pragma unroll
        for (int jt = 0; jt < NT; ++jt) {
          f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acc2 = {0.f, 0.f, 0.f, 0.f};  // two chains: half the dependent-MFMA latency
          {
            if (jt + 1 < NT) stage_e(ii, jt + 1);
            const float* t_ = scr + (jt & 1) * (16 * ELD) + l15 * ELD + 4 * q;
#pragma unroll
            for (int sg = 0; sg < 4; ++sg) {
              const f32x4 ea = *reinterpret_cast<const f32x4*>(t_ + 16 * sg);  // e[i][16 jt + l15][16 sg + 4 q + s]
              const f32x4 wbf = MULTI ? *reinterpret_cast<const f32x4*>(wb_lds + (sg * 64 + lane) * 4) : wb[sg];
#pragma unroll
              for (int s = 0; s < 4; ++s) {
                if (sg & 1) acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(ea[s], wbf[s], acc2, 0, 0, 0);
                else acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ea[s], wbf[s], acc, 0, 0, 0);
              }
            }
          }
          if (ii == 0 && jt + E_DEPTH0 < NT) {
            load_e_tile(0, c, jt + E_DEPTH0);
            MEM_FENCE();
          }
          const f32x4 sv = *reinterpret_cast<const f32x4*>(Srow + jt * 16 + 4 * q);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float v = sv[r] + bscale * (acc[r] + acc2[r]);
            lg[jt][r] = v;
            mx = fmaxf(mx, v);
          }
        }
        mx = fmaxf(mx, __shfl_xor(mx, 16));
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const float Mnew = fmaxf(Mrun[ii], mx);
        const float fac = c == 0 ? 0.0f : expf(Mrun[ii] - Mnew);  // rescale of everything accumulated before this chunk
        float sum = 0.f;
#pragma unroll
        for (int jt = 0; jt < NT; ++jt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float p = FAST_EXP(lg[jt][r] - Mnew);
            lg[jt][r] = p;
            sum += p;
          }
        sum += __shfl_xor(sum, 16);
        sum += __shfl_xor(sum, 32);
        Mrun[ii] = Mnew;
        Lrun[ii] = Lrun[ii] * fac + sum;
        const float inv = last ? 1.0f / Lrun[ii] : 1.0f;
        // ---- exp(logit - M): to LDS for phase 3, and straight into the o_e product as its B operand
        f32x4 oe[4];
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) oe[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int jt = 0; jt < NT; ++jt) {
          const f32x4 pv = {lg[jt][0], lg[jt][1], lg[jt][2], lg[jt][3]};
          if (l15 < 8) *reinterpret_cast<f32x4*>(Srow + jt * 16 + 4 * q) = pv;
#pragma unroll
          for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int ct = 0; ct < 4; ++ct)  // A: e[i][j = 16 jt + 4 q + r][c = 4 l15 + ct]
              oe[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(ev[ii][jt][r][ct], pv[r], oe[ct], 0, 0, 0);
          if (ii == 0 && jt >= E_LAG) {  // retired tiles free their registers: start the next row, E_LAG tiles behind
            load_e_tile(1, c, jt - E_LAG);
            MEM_FENCE();
          }
        }
        if (ii == 0) {  // the last E_LAG tiles of the next row are needed last by its bias loop
#pragma unroll
          for (int jt = NT - E_LAG; jt < NT; ++jt) load_e_tile(1, c, jt);
          MEM_FENCE();
        }
        // D: column h = l15, row m = 4 q + r' <-> channel 4 m + ct = 16 q + 4 r' + ct
        if (l15 < 8) {
          float* fo = feat + (prow0 + i0 + il) * AF + FOFF_OE + h * AC + 16 * q;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            f32x4 v = {oe[0][r], oe[1][r], oe[2][r], oe[3][r]};
            if (c > 0) {
              const f32x4 old = *reinterpret_cast<const f32x4*>(fo + 4 * r);
#pragma unroll
              for (int s = 0; s < 4; ++s) v[s] += old[s] * fac;
            }
#pragma unroll
            for (int s = 0; s < 4; ++s) v[s] *= inv;
            *reinterpret_cast<f32x4*>(fo + 4 * r) = v;
          }
          if (q == 0) {
            st_fac[il * AH + h] = fac;
            st_inv[il * AH + h] = inv;
          }
        }
      }
      }  // !PLANES
    }
    if (c == 0) stamp(3);

    // ---------------------------------------------------------------- phase 3: wave = head
    {
      constexpr int PFV = 16;  // value prefetch distance, key steps (16 x 160 MFMA cycles ~ the loaded L2 latency)
      const int h = wv;
      const int pp = l15 & 7;
      const float* vbase = proj + (krow0 + 4 * q) * ANP + OFF_VS + h * ADS + 2 * l15;  // + (16 jt + r) rows; d = 2 l15 + dt
      const float* gbase = proj + (krow0 + 4 * q) * ANP + OFF_GV + h * 24 + 3 * pp;    // point pp, coords 0..2
      // Point sums: columns 0..7 of ONE MFMA tile hold x of the 8 points, columns 8..15 y (z in a second tile, its upper half
      // duplicates): 4 f32 MFMAs per key step instead of 5 - this phase is bound by exactly those (32 cycles each).
      const int xy = l15 >> 3;  // 0: this lane's column is x of point pp, 1: y
      float2 vs[NS];
      float gxy[NS], gz[NS];
      auto load_vals = [&](int stp) {
        const int64_t o = static_cast<int64_t>((stp >> 2) * 16 + (stp & 3)) * ANP;
        vs[stp] = *reinterpret_cast<const float2*>(vbase + o);
        gxy[stp] = gbase[o + xy];
        gz[stp] = gbase[o + 2];
      };
#pragma unroll
      for (int stp = 0; stp < PFV && stp < NS; ++stp) load_vals(stp);
      MEM_FENCE();
      __syncthreads();  // exp(logit - M) of all rows and the rescale factors are in LDS
      if (c == 0) stamp(4);
      f32x4 os[2], og[2];  // og[0]: x | y of the points, og[1]: z
#pragma unroll
      for (int d = 0; d < 2; ++d) os[d] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int cc = 0; cc < 2; ++cc) og[cc] = f32x4{0.f, 0.f, 0.f, 0.f};
      const float* Prow = S + l15 * IS + h * HS + 4 * q;  // A operand: P[i = l15][j = 16 jt + 4 q + r]
#pragma unroll
      for (int jt = 0; jt < NT; ++jt) {
        const f32x4 pa = *reinterpret_cast<const f32x4*>(Prow + jt * 16);
        if constexpr (TAPE) {  // P[i = l15][16 jt + 4 q ..]: the image holds exp(logit - M), the row's 1 / L is in st_inv
          const float pinv = st_inv[l15 * AH + h];
          *reinterpret_cast<f32x4*>(tape_p + ((static_cast<int64_t>(b) * AH + h) * K + i0 + l15) * K + jt * 16 + 4 * q) = pa * pinv;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int stp = jt * 4 + r;
          if (stp + PFV < NS) {
            load_vals(stp + PFV);
            MEM_FENCE();
          }
          os[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(pa[r], vs[stp].x, os[0], 0, 0, 0);
          os[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(pa[r], vs[stp].y, os[1], 0, 0, 0);
          og[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(pa[r], gxy[stp], og[0], 0, 0, 0);
          og[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(pa[r], gz[stp], og[1], 0, 0, 0);
        }
      }
      // D rows i = 4 q + r, column n = l15; earlier chunks' sums are rescaled through the feature row
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int il = 4 * q + r;
        const int64_t row = prow0 + i0 + il;
        const float fac = st_fac[il * AH + h], inv = st_inv[il * AH + h];
        float* fr = feat + row * AF;
        float2 o2 = make_float2(os[0][r], os[1][r]);
        float2* po = reinterpret_cast<float2*>(fr + FOFF_OS + h * ADS + 2 * l15);
        if (c > 0) {
          const float2 old = *po;
          o2.x += old.x * fac;
          o2.y += old.y * fac;
        }
        o2.x *= inv;
        o2.y *= inv;
        *po = o2;
        const float gy_lane = __shfl_xor(og[0][r], 8);  // lanes 0..7 hold x of point l15, lanes 8..15 y of point l15 - 8
        if (l15 < 8) {
          float* fo = fr + FOFF_OL + h * 24 + 3 * l15;
          float g0_ = og[0][r], g1_ = gy_lane, g2_ = og[1][r];
          if (c > 0) {  // running (unnormalised, global-frame) sums are parked in the o_l slot between chunks
            g0_ += fo[0] * fac;
            g1_ += fo[1] * fac;
            g2_ += fo[2] * fac;
          }
          if (last) {
            const float* Rr = R + row * 9;
            const float* tr = t + row * 3;
            const float dx = g0_ * inv - tr[0], dy = g1_ * inv - tr[1], dz = g2_ * inv - tr[2];
            const float lx = dx * Rr[0] + dy * Rr[1] + dz * Rr[2];  // (p - t) R^T   (diffab_pytorch.py:336)
            const float ly = dx * Rr[3] + dy * Rr[4] + dz * Rr[5];
            const float lz = dx * Rr[6] + dy * Rr[7] + dz * Rr[8];
            fo[0] = lx; fo[1] = ly; fo[2] = lz;
            fr[FOFF_ON + h * AP + l15] = sqrtf(lx * lx + ly * ly + lz * lz);
          } else {
            fo[0] = g0_; fo[1] = g1_; fo[2] = g2_;
          }
        }
      }
    }
    if (!last) __syncthreads();  // the next chunk's phase 1 overwrites the image
  }
  stamp(5);
}

static unsigned long long* g_attn_stamps = nullptr;  // diagnostics only (diffab_debug_set_attn_stamps)
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

// ================================================================== six projections + local->global frames in one kernel
// proj[:, 0:1344] = x [Wq_s; Wk_s; Wv_s; Wq_p; Wk_p; Wv_p]^T with the three point blocks mapped to the global frame
// (x R + t, row-vector convention of diffab_pytorch.py:324) before they are stored.
//
// x-stationary: a work-group owns 128 rows of x for the whole kernel and every wave keeps its 32 x 128 slab as MFMA A fragments
// in 64 VGPRs, so x is read from HBM exactly once and the LDS only double-buffers 96-column blocks of the weights (14 blocks).
// Inside a block the MFMA n index is permuted: tile tt (0..2), lane column j holds output column 3 j + tt of the wave's 48, so a
// lane ends up with three CONSECUTIVE output columns per row - a whole (x, y, z) point in the point blocks, and a 12-byte
// store (16 lanes = 192 contiguous bytes) everywhere.  The previous block's epilogue is issued between the MFMAs of the
// current one (two accumulator sets), which keeps the matrix pipe fed across the one barrier per block.
constexpr int PJB = 96, PJLD = 132, PJROWS = 128, PJNB = ANP / PJB;  // 14 blocks
static_assert(ANP % PJB == 0 && OFF_GQ % PJB == 0, "projection blocks must tile the scalar and point column ranges");
struct __attribute__((packed, aligned(4))) pj_f3 { float x, y, z; };

struct PjW {  // the six weight matrices, by value (kept in SGPRs)
  const float *w0, *w1, *w2, *w3, *w4, *w5;
};
struct PjCtx {  // per-thread constants of proj_frames_kernel
  float* PW;
  float* Rt;
  float* ybase;
  int tid, l15, g, rw, cw, m0, M;
};

// weight staging: thread -> (LDS row l = 16 r + tid / 32, float4 column tid % 32); LDS row l = 48 cw' + 16 tt + j holds output
// column 48 cw' + 3 j + tt of the block.  The six weight pointers stay in SGPRs (selects, no indexed kernarg loads).
__device__ __forceinline__ void pj_load_w(const PjCtx& c, const PjW w, int blk, f32x4 (&wreg)[6]) {
#pragma unroll
  for (int r = 0; r < 6; ++r) {
    const int l = 16 * r + (c.tid >> 5), c4 = c.tid & 31;
    const int cwl = l / 48, rem = l % 48, tt = rem >> 4, j = rem & 15;
    const int gc = PJB * blk + 48 * cwl + 3 * j + tt;
    const float* Wp;
    int row;
    if (gc < OFF_GQ) {
      Wp = gc < OFF_KS ? w.w0 : (gc < OFF_VS ? w.w1 : w.w2);
      row = gc & 255;
    } else {
      Wp = gc < OFF_GK ? w.w3 : (gc < OFF_GV ? w.w4 : w.w5);
      row = gc - (gc < OFF_GK ? OFF_GQ : (gc < OFF_GV ? OFF_GK : OFF_GV));
    }
    wreg[r] = *reinterpret_cast<const f32x4*>(Wp + row * 128 + 4 * c4);
  }
}
__device__ __forceinline__ void pj_store_w(const PjCtx& c, int buf, const f32x4 (&wreg)[6]) {
#pragma unroll
  for (int r = 0; r < 6; ++r) {
    const int l = 16 * r + (c.tid >> 5), c4 = c.tid & 31;
    *reinterpret_cast<f32x4*>(&c.PW[(buf * PJB + l) * PJLD + 4 * c4]) = wreg[r];
  }
}
// one (mt, r) slice of a finished block: 3 consecutive columns of one row per lane
template <bool FULL, bool FRAMES>
__device__ __forceinline__ void pj_epilogue_piece(const PjCtx& c, const f32x4 (&acc)[2][3], int blk, int piece) {
  const int mt = piece >> 2, r = piece & 3;
  const int lrow = 32 * c.rw + 16 * mt + 4 * c.g + r;
  float vx = acc[mt][0][r], vy = acc[mt][1][r], vz = acc[mt][2][r];
  if (FRAMES) {
    const f32x4* F = reinterpret_cast<const f32x4*>(c.Rt + lrow * 12);
    const f32x4 f0 = F[0], f1 = F[1], f2 = F[2];  // R row-major 0..8, t 9..11
    const float ox = (vx * f0[0] + vy * f0[3] + vz * f1[2]) + f2[1];
    const float oy = (vx * f0[1] + vy * f1[0] + vz * f1[3]) + f2[2];
    const float oz = (vx * f0[2] + vy * f1[1] + vz * f2[0]) + f2[3];
    vx = ox; vy = oy; vz = oz;
  }
  if (FULL || c.m0 + lrow < c.M) {
    pj_f3 o{vx, vy, vz};
    *reinterpret_cast<pj_f3*>(c.ybase + (16 * mt + r) * ANP + PJB * blk) = o;
  }
}
template <bool FULL, bool HAVE_PREV, bool PREV_FRAMES>
__device__ __forceinline__ void pj_run_block(const PjCtx& c, const PjW w, const f32x4 (&a)[2][8], f32x4 (&wreg)[6], f32x4 (&cur)[2][3],
                                             const f32x4 (&prev)[2][3], int blk) {
  if (blk + 1 < PJNB) pj_load_w(c, w, blk + 1, wreg);
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int tt = 0; tt < 3; ++tt) cur[mt][tt] = f32x4{0.f, 0.f, 0.f, 0.f};
  const float* Wl = c.PW + ((blk & 1) * PJB + 48 * c.cw + c.l15) * PJLD + 4 * c.g;
  f32x4 b[2][3];
#pragma unroll
  for (int tt = 0; tt < 3; ++tt) b[0][tt] = *reinterpret_cast<const f32x4*>(Wl + 16 * tt * PJLD);
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    if (q + 1 < 8) {
#pragma unroll
      for (int tt = 0; tt < 3; ++tt) b[(q + 1) & 1][tt] = *reinterpret_cast<const f32x4*>(Wl + 16 * tt * PJLD + 16 * (q + 1));
    }
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int tt = 0; tt < 3; ++tt)
          cur[mt][tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt][q][s], b[q & 1][tt][s], cur[mt][tt], 0, 0, 0);
    // The next block's weights go to LDS (q = 1) BEFORE this block issues any global store (q = 2..7): on gfx9 a wait for loads
    // with stores in flight degenerates to vmcnt(0), i.e. to waiting for the L2 acknowledgement of the newest store (measured:
    // 8 % of the kernel when the wait sat right behind the last store of the block).
    if (q == 1 && blk + 1 < PJNB) pj_store_w(c, (blk + 1) & 1, wreg);
    if (HAVE_PREV) {  // 8 epilogue slices of the previous block spread over q = 2..7
      if (q == 2) { pj_epilogue_piece<FULL, PREV_FRAMES>(c, prev, blk - 1, 0); pj_epilogue_piece<FULL, PREV_FRAMES>(c, prev, blk - 1, 1); }
      if (q == 3) { pj_epilogue_piece<FULL, PREV_FRAMES>(c, prev, blk - 1, 2); pj_epilogue_piece<FULL, PREV_FRAMES>(c, prev, blk - 1, 3); }
      if (q >= 4) pj_epilogue_piece<FULL, PREV_FRAMES>(c, prev, blk - 1, q);
    }
    __builtin_amdgcn_sched_barrier(0);  // keep each slice's stores where they are (the scheduler otherwise sinks all 8 to the barrier)
  }
  __syncthreads();
}

template <bool FULL>  // FULL: M is a multiple of 128, no row guards
__global__ __launch_bounds__(512) void proj_frames_kernel(const float* __restrict__ X, const float* __restrict__ W0,
                                                          const float* __restrict__ W1, const float* __restrict__ W2,
                                                          const float* __restrict__ W3, const float* __restrict__ W4,
                                                          const float* __restrict__ W5, const float* __restrict__ R,
                                                          const float* __restrict__ t, float* __restrict__ Y, int M) {
  extern __shared__ __attribute__((aligned(16))) float PW[];  // [2][PJB][PJLD] weights, then [PJROWS][12] frames
  PjCtx c;
  PjW w;
  {  // pin the six weight pointers in SGPRs: without this the selects in pj_load_w become per-lane indexed loads of the
     // pointer itself (a dependent memory round trip in front of every weight load)
    const float *w0 = W0, *w1 = W1, *w2 = W2, *w3 = W3, *w4 = W4, *w5 = W5;
    asm volatile("" : "+s"(w0), "+s"(w1), "+s"(w2), "+s"(w3), "+s"(w4), "+s"(w5));
    w = PjW{w0, w1, w2, w3, w4, w5};
  }
  c.PW = PW;
  c.Rt = PW + 2 * PJB * PJLD;
  c.tid = threadIdx.x;
  const int lane = c.tid & 63, wv = c.tid >> 6;
  c.l15 = lane & 15; c.g = lane >> 4; c.rw = wv & 3; c.cw = wv >> 2;
  c.m0 = blockIdx.x * PJROWS;
  c.M = M;
  c.ybase = Y + static_cast<int64_t>(c.m0 + 32 * c.rw + 4 * c.g) * ANP + 48 * c.cw + 3 * c.l15;

  f32x4 wreg[6];
  pj_load_w(c, w, 0, wreg);
  // A fragments: a[mt][q][s] = x[m0 + 32 rw + 16 mt + l15][16 q + 4 g + s]
  f32x4 a[2][8];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
    const int row = c.m0 + 32 * c.rw + 16 * mt + c.l15;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (FULL || row < M) v = *reinterpret_cast<const f32x4*>(X + static_cast<int64_t>(row) * 128 + 16 * q + 4 * c.g);
      a[mt][q] = v;
    }
  }
  for (int idx = c.tid; idx < PJROWS * 12; idx += 512) {
    const int row = idx / 12, cc = idx % 12, gr = c.m0 + row;
    float v = 0.0f;
    if (FULL || gr < M) v = cc < 9 ? R[static_cast<int64_t>(gr) * 9 + cc] : t[static_cast<int64_t>(gr) * 3 + (cc - 9)];
    c.Rt[idx] = v;
  }
  pj_store_w(c, 0, wreg);
  __syncthreads();

  f32x4 accA[2][3], accB[2][3];
  constexpr int FIRST_PT = OFF_GQ / PJB;  // 8: blocks 0..7 are the scalar q/k/v columns, 8..13 the point columns
  static_assert(FIRST_PT % 2 == 0 && PJNB % 2 == 0, "block schedule below assumes even counts");
  pj_run_block<FULL, false, false>(c, w, a, wreg, accA, accB, 0);
  for (int blk = 1; blk < FIRST_PT; blk += 2) {  // previous block is a scalar block
    pj_run_block<FULL, true, false>(c, w, a, wreg, accB, accA, blk);
    pj_run_block<FULL, true, false>(c, w, a, wreg, accA, accB, blk + 1);
  }
  for (int blk = FIRST_PT + 1; blk + 1 < PJNB; blk += 2) {  // previous block is a point block
    pj_run_block<FULL, true, true>(c, w, a, wreg, accB, accA, blk);
    pj_run_block<FULL, true, true>(c, w, a, wreg, accA, accB, blk + 1);
  }
  pj_run_block<FULL, true, true>(c, w, a, wreg, accB, accA, PJNB - 1);
#pragma unroll
  for (int q = 0; q < 8; ++q) pj_epilogue_piece<FULL, true>(c, accB, PJNB - 1, q);
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
    const bool ok = ex > 0 && ex < 231 && b < 0x7f800000u;  // zero / subnormal / huge / inf maximum: no scaling (inf / nan propagate as in fp32)
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
size_t ipa_layer_planes_bytes() { return round256(proj_frames_b6_scratch_bytes()) + round256(rowgemm128_b6_scratch_bytes(AF)); }
int ipa_layer_split_weights(const diffab_ipa_layer_weights* w, void* planes, hipStream_t st) {
  DIFFAB_REQUIRE(w && w->wq_s && w->wk_s && w->wv_s && w->wq_p && w->wk_p && w->wv_p && w->w_out, DIFFAB_ERR_ARG, "ipa layer: null weight pointer");
  const float* W6[6] = {w->wq_s, w->wk_s, w->wv_s, w->wq_p, w->wk_p, w->wv_p};
  if (int rc = launch_pjsplit(W6, planes, st)) return rc;
  return launch_wsplit128(w->w_out, AF, AF, static_cast<char*>(planes) + round256(proj_frames_b6_scratch_bytes()), st);
}
static size_t b6_scratch_floats() { return (ipa_layer_planes_bytes() + 256) / sizeof(float); }

// workspace of one layer: proj | feat | 128 | three-launch attention's logits (K = 64 / 128) | per-call weight planes | operand planes
// of the logits product (proj_planes.hip) | patch centroids
static size_t ipa_ws_operands_offset(const diffab_dims* d) {
  const size_t rows = static_cast<size_t>(d->B) * d->K;
  const size_t o = rows * (ANP + AF) + 128 + (attention_split_supported(d) ? attention_split_workspace_floats(d) : 0) + b6_scratch_floats();
  return (o + 63) & ~static_cast<size_t>(63);
}
size_t ipa_fast_workspace_floats(const diffab_dims* d) { return ipa_ws_operands_offset(d); }

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
  const void* out_planes = b6 ? static_cast<const char*>(planes) + round256(proj_frames_b6_scratch_bytes()) : nullptr;
  // to_out (diffab_pytorch.py:459-464): feat (rows x 1024) Wo^T + b
  auto to_out = [&]() -> int {
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
  if (b6) {
    if (int rc = launch_proj_frames_b6p(x, planes, R, t, proj, rows, st)) return rc;
  } else {
    const size_t pj_lds = (2 * PJB * PJLD + PJROWS * 12) * sizeof(float);
#define PROJ_LAUNCH(FULL_)                                                                                                        \
  do {                                                                                                                            \
    DIFFAB_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(proj_frames_kernel<FULL_>),                                \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(pj_lds)));                  \
    hipLaunchKernelGGL((proj_frames_kernel<FULL_>), dim3((rows + PJROWS - 1) / PJROWS), dim3(512), pj_lds, st, x, segs.W[0],      \
                       segs.W[1], segs.W[2], segs.W[3], segs.W[4], segs.W[5], R, t, proj, rows);                                  \
  } while (0)
    if (rows % PJROWS == 0) PROJ_LAUNCH(true);
    else PROJ_LAUNCH(false);
#undef PROJ_LAUNCH
    DIFFAB_LAUNCH_CHECK();
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
  const size_t lds = (static_cast<size_t>(TI) * (AH * (16 * nt + 8) + 8) + 8 * 2 * 16 * 72 + 2 * TI * AH + 4 * 64 * 4) * sizeof(float);
  const dim3 grid(d->B * (d->K / TI));
  // pair_planes (launch_pair_split): the pair-tile products on the f16 matrix cores, the pair stream read as two fp16 planes
  const bool use_planes = pair_planes != nullptr && pair_planes_supported(d);
  const float* e_arg = use_planes ? pair_planes + 64 : e;
  const float* esc = use_planes ? pair_row_scales(d, pair_planes) : nullptr;
#define ATTN_LAUNCH(NT_, MULTI_, PLANES_)                                                                                             \
  do {                                                                                                                                \
    DIFFAB_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(ipa_attn_fast_kernel<NT_, MULTI_, PLANES_>),                   \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));                         \
    timer_begin(st);                                                                                                                  \
    hipLaunchKernelGGL((ipa_attn_fast_kernel<NT_, MULTI_, PLANES_>), grid, dim3(512), lds, st, proj, e_arg, R, t, w->w_bias,          \
                       w->gamma, feat, d->B, nc, g_attn_stamps, esc, nullptr, nullptr, tile_needed);                                  \
    timer_end(st);                                                                                                                    \
  } while (0)
  if (tape) {
    DIFFAB_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(ipa_attn_fast_kernel<8, false, false, true>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
    hipLaunchKernelGGL((ipa_attn_fast_kernel<8, false, false, true>), grid, dim3(512), lds, st, proj, e, R, t, w->w_bias, w->gamma, feat,
                       d->B, nc, nullptr, nullptr, sp_keep, d2_keep);
  } else if (use_planes) {
    if (nt == 8 && nc == 1) ATTN_LAUNCH(8, false, true);
    else if (nt == 8) ATTN_LAUNCH(8, true, true);
    else if (nc == 1) ATTN_LAUNCH(4, false, true);
    else ATTN_LAUNCH(4, true, true);
  } else if (nt == 8 && nc == 1) ATTN_LAUNCH(8, false, false);
  else if (nt == 8) ATTN_LAUNCH(8, true, false);
  else if (nc == 1) ATTN_LAUNCH(4, false, false);
  else ATTN_LAUNCH(4, true, false);
#undef ATTN_LAUNCH
  DIFFAB_LAUNCH_CHECK();
  return to_out();
}

}  // namespace diffab
