// pair_embed_fused.hip - PairEmbedding forward (reference diffab_pytorch.py:186-312) as ONE kernel for the benchmark geometry
// (C = 64, K % 128 == 0, A <= 16): a 512-thread work-group owns 128 pair rows (b, i, j0 .. j0 + 127) from the atom-atom distance
// features to the masked output row, so neither the 225-wide distance features (14.7 MB per K = 128 patch) nor the 210-wide
// concatenation (13.8 MB) nor any hidden layer ever exists in HBM:
//   features (exp(-softplus(coef) d^2) mask, generated per 32-k chunk straight into split bf16 planes in LDS)
//     -> 225 -> 64 ReLU -> 64 ReLU                                   distance_embedding (:212-217)
//     -> [aa_pair_emb | relpos_emb * same-chain | . | dihedral enc] -> 64 ReLU -> 64 ReLU -> 64, x CA mask   mlp (:219-226, :303-312)
// The dense layers run on the bf16 matrix cores as six-term split products (gemm_bf16x6.hip: fp32-accurate); the two embedding-table
// segments of the concatenation are folded: cat[:, 0:C] W_a^T = (aa_pair_emb W_a^T)[s_i 21 + s_j] is a 441-row table T1 built once
// per call, likewise T2 = relpos_emb W_b^T, so those 128 of the 210 input columns cost one gathered row each instead of 128 k of GEMM.
// TAPE: the four hidden activations (after their ReLUs) are also written out, [rows][64] each - what the backward needs
// (context_kernels.hip: diffab_pair_embedding_bwd), instead of re-running six unfused kernels per chunk.
#include "common.h"
#include "denoiser_internal.h"
#include "rowgemm_b6_tile.h"

namespace diffab {

namespace {
using b6tile::b6_off;
using b6tile::BK;
using b6tile::split3;
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr int kAAf = 21, kUNKf = 20, kCAf = 1;
constexpr int PC = 64;     // d_feat of the pair embedding (compile-time: wave tiles, table rows)
constexpr int PROWS = 128; // pair rows per work-group
constexpr int W_CHUNK = 3 * PC * BK;        // bf16 elements of one staged weight chunk (3 planes x 64 rows x 32 k) = 12 KiB
constexpr int A_CHUNK = 3 * PROWS * BK;     // bf16 elements of one activation / feature chunk (3 planes x 128 rows x 32 k) = 24 KiB
// LDS: weights [2][W_CHUNK] | features [2][A_CHUNK] | activations [2][A_CHUNK] | xyz_j [128][16 x 3] f32 | amask_j [128][16] f32 |
// xyz_i [16 x 3], amask_i [16] | row info.  The distance features are indexed k = 16 a1 + a2 (A padded to 16: a thread's eight
// features share a1 and take eight consecutive atoms a2 of the j side), the planes of distance_embedding[0].weight likewise.
constexpr int kMaxA = 16;
struct RowInfo { int idx; int rel; float same; float mask; };
constexpr size_t kFusedLdsBytes = static_cast<size_t>(2 * W_CHUNK + 4 * A_CHUNK) * 2 + static_cast<size_t>(PROWS + 1) * kMaxA * 4 * sizeof(float) +
                                  PROWS * sizeof(RowInfo);

struct FusedArgs {
  const int64_t* seq; const uint8_t* seq_m; const float* distmat; const float* xyz; const float* amask; const float* pdih;
  const int64_t* resid; int resid_bstride; const int64_t* chain;
  const float* coef_sp;           // [441][A A] softplus(pair2distcoef)
  const __bf16* pl_dw0;           // [nch0][3][64][32] planes of distance_embedding[0].weight (K = A A padded to 32 nch0)
  const __bf16* pl_dw2;           // [2][3][64][32]
  const __bf16* pl_m0;            // [3][3][64][32]: mlp[0].weight[:, 2C:3C] (2 chunks) | [:, 3C:3C+18] padded to 32 (1 chunk)
  const __bf16* pl_m2;            // [2]...
  const __bf16* pl_m4;
  const float* db0; const float* db2; const float* mb0; const float* mb2; const float* mb4;
  const float* T1;                // [441][64] aa_pair_emb mlp[0].weight[:, 0:C]^T
  const float* T2;                // [2 max_dist + 1][64]
  float* out;                     // [rows][64]
  float* tape_h1; float* tape_df; float* tape_m1; float* tape_m2;  // TAPE: [rows][64] each
  int K, A, max_dist, nch0;
  int64_t row0, nrows;            // this launch covers pair rows [row0, row0 + nrows), nrows % 128 == 0
};

// W[64 x ncols] (row stride ldw, first column col0; columns >= ncols are zero) -> chunk-major split planes [nch][3][64][32].
// atoms > 0: k = 16 a1 + a2 takes column a1 atoms + a2 (zero where a1 or a2 >= atoms): the padded index of the distance features
__global__ void wsplit64_kernel(const float* __restrict__ W, int ldw, int col0, int ncols, int nch, int atoms, __bf16* __restrict__ out) {
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;  // (n, k)
  if (gid >= PC * nch * BK) return;
  const int n = gid / (nch * BK), k = gid % (nch * BK);
  int col = k;
  bool ok = k < ncols;
  if (atoms > 0) {
    const int a1 = k >> 4, a2 = k & 15;
    ok = a1 < atoms && a2 < atoms;
    col = a1 * atoms + a2;
  }
  __bf16 h, m, l;
  split3(ok ? W[static_cast<int64_t>(n) * ldw + col0 + col] : 0.0f, h, m, l);
  const int chunk = k / BK, kk = k % BK;
  const size_t base = (static_cast<size_t>(chunk) * 3 * PC + n) * BK + kk;
  out[base] = h;
  out[base + PC * BK] = m;
  out[base + 2 * PC * BK] = l;
}
// T[r][n] = sum_c E[r][c] W[n][col0 + c]   (r < nrows_e; C = 64): the folded embedding-table segments
__global__ void fold_table64_kernel(const float* __restrict__ E, int nrows_e, const float* __restrict__ W, int ldw, int col0, float* __restrict__ T) {
  const int r = blockIdx.x, n = threadIdx.x;
  if (r >= nrows_e || n >= PC) return;
  float acc = 0.f;
  for (int c = 0; c < PC; ++c) acc += E[r * PC + c] * W[static_cast<int64_t>(n) * ldw + col0 + c];
  T[r * PC + n] = acc;
}
}  // namespace

template <bool TAPE>
__global__ __launch_bounds__(512) void pair_embed_fused_kernel(const FusedArgs a) {
  extern __shared__ __attribute__((aligned(16))) __bf16 lds[];
  __bf16* Ws = lds;                   // [2][W_CHUNK]
  __bf16* Fs = lds + 2 * W_CHUNK;     // [2][A_CHUNK]: generated feature chunks (then the dihedral-encoding chunk)
  __bf16* Act = Fs + 2 * A_CHUNK;     // [2][A_CHUNK]: a 64-wide activation as two 32-k chunks of split planes
  float* xj = reinterpret_cast<float*>(Act + 2 * A_CHUNK);  // [128][A 3]
  float* mj = xj + PROWS * kMaxA * 3;                       // [128][16]
  float* xi = mj + PROWS * kMaxA;                           // [16 x 3]
  float* mi = xi + kMaxA * 3;                               // [16]
  RowInfo* info = reinterpret_cast<RowInfo*>(mi + kMaxA);
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int l31 = lane & 31, hk = lane >> 5, rw = wv & 3, cw = wv >> 2;  // wave tile: rows 32 rw .., columns 32 cw ..
  const int A = a.A, AA2 = A * A, K = a.K;
  const int64_t tile_row0 = a.row0 + static_cast<int64_t>(blockIdx.x) * PROWS;  // (b, i, j0): K % 128 == 0, so i is the same for the tile
  const int64_t bi = tile_row0 / K;                                              // b K + i
  const int j0 = static_cast<int>(tile_row0 % K);
  const int64_t b = bi / K;
  const int i = static_cast<int>(bi % K);
  const int64_t ri = bi;
  const int si = static_cast<int>((a.seq_m && !a.seq_m[ri]) ? kUNKf : a.seq[ri]);
  // ---- per-row constants, the j side of the coordinates
  if (tid < PROWS) {
    const int64_t rj = b * K + j0 + tid;
    const int sj = static_cast<int>((a.seq_m && !a.seq_m[rj]) ? kUNKf : a.seq[rj]);
    int64_t rel = a.resid[b * a.resid_bstride + i] - a.resid[b * a.resid_bstride + j0 + tid];
    rel = rel < -a.max_dist ? -a.max_dist : (rel > a.max_dist ? a.max_dist : rel);
    RowInfo ri_;
    ri_.idx = si * kAAf + sj;
    ri_.rel = static_cast<int>(rel) + a.max_dist;
    ri_.same = static_cast<float>(a.chain[ri] * a.chain[rj]);  // a product, not an equality test (:279)
    ri_.mask = a.amask[ri * A + kCAf] * a.amask[rj * A + kCAf];
    info[tid] = ri_;
  }
  for (int idx = tid; idx < (PROWS + 1) * kMaxA; idx += 512) {  // rows 0..127: the j side; row 128: residue i (atoms >= A: mask 0, origin)
    const int r = idx / kMaxA, at = idx % kMaxA;
    const int64_t rr = r < PROWS ? b * K + j0 + r : ri;
    float m_ = 0.0f, x0 = 0.0f, x1 = 0.0f, x2 = 0.0f;
    if (at < A) {
      m_ = a.amask[rr * A + at];
      if (a.xyz) {
        const float* p = a.xyz + (rr * A + at) * 3;
        x0 = p[0]; x1 = p[1]; x2 = p[2];
      }
    }
    float* xd = r < PROWS ? xj + (r * kMaxA + at) * 3 : xi + at * 3;
    (r < PROWS ? mj[r * kMaxA + at] : mi[at]) = m_;
    xd[0] = x0; xd[1] = x1; xd[2] = x2;
  }
  // Weight staging.  The planes of the five layers are ONE sequence of nch0 + 9 chunks in memory (launch_pair_embed_fused), so the stream
  // is a plain ring: chunk g is used from LDS buffer g % 2 while chunk g + 1 sits in registers (requested a chunk earlier: its L2 round
  // trip is hidden behind a chunk of MFMAs) and chunk g + 2 is being requested.  A chunk is 768 16-byte pieces (3 planes x 64 rows x 4
  // slots): thread tid takes piece tid, threads < 256 also piece 512 + tid.
  const int n_wchunks = a.nch0 + 9;
  f32x4 wreg[2];
  auto load_w = [&](int g_) {
    g_ = g_ < n_wchunks ? g_ : n_wchunks - 1;  // (unconditional: a branch around a prefetch makes the compiler wait for it at once)
    const __bf16* src = a.pl_dw0 + static_cast<size_t>(g_) * W_CHUNK;
    wreg[0] = *reinterpret_cast<const f32x4*>(src + static_cast<size_t>(tid) * 8);
    wreg[1] = *reinterpret_cast<const f32x4*>(src + static_cast<size_t>(512 + (tid & 255)) * 8);
  };
  auto store_w = [&](int buf) {
    {
      const int piece = tid, p = piece / 256, row = (piece % 256) >> 2, part = piece & 3;
      *reinterpret_cast<f32x4*>(Ws + buf * W_CHUNK + p * PC * BK + b6_off(row, part)) = wreg[0];
    }
    if (tid < 256) {
      const int piece = 512 + tid, p = piece / 256, row = (piece % 256) >> 2, part = piece & 3;
      *reinterpret_cast<f32x4*>(Ws + buf * W_CHUNK + p * PC * BK + b6_off(row, part)) = wreg[1];
    }
  };
  // feature generation: thread -> (row tid / 4, features k = 32 chunk + 8 (tid % 4) .. + 7 = atom a1 of residue i against atoms a2 .. a2 + 7
  // of residue j); exp(-softplus(coef) d^2) mask_i mask_j (:288-295), the arithmetic of pair_dist_kernel
  const int frow = tid >> 2, fslot = tid & 3;
  auto gen_features = [&](int chunk, int buf) {
    const RowInfo inf = info[frow];
    const int k0 = chunk * BK + 8 * fslot;
    const int a1 = k0 >> 4, a2 = k0 & 15;  // a2 = 0 or 8
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = 0.0f;
    if (a1 < A) {
      // (eight coefficients of row a1 from a2 on: the two past the row's end at a2 = 8 belong to the next row or the pad behind the table)
      const float* crow_ = a.coef_sp + static_cast<int64_t>(inf.idx) * AA2 + a1 * A + a2;
      float crow[8];
      *reinterpret_cast<f32x4*>(crow) = *reinterpret_cast<const f32x4u*>(crow_);
      *reinterpret_cast<f32x4*>(crow + 4) = *reinterpret_cast<const f32x4u*>(crow_ + 4);
      const float* drow = a.distmat ? a.distmat + (tile_row0 + frow) * AA2 + a1 * A : nullptr;
      const float xi0 = xi[3 * a1], xi1 = xi[3 * a1 + 1], xi2 = xi[3 * a1 + 2], m1_ = mi[a1];
      float xb[24], mb[8];
#pragma unroll
      for (int q4 = 0; q4 < 6; ++q4) *reinterpret_cast<f32x4*>(xb + 4 * q4) = *reinterpret_cast<const f32x4*>(xj + (frow * kMaxA + a2) * 3 + 4 * q4);
#pragma unroll
      for (int q4 = 0; q4 < 2; ++q4) *reinterpret_cast<f32x4*>(mb + 4 * q4) = *reinterpret_cast<const f32x4*>(mj + frow * kMaxA + a2 + 4 * q4);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        if (a2 + e < A) {
          float d;
          if (drow) {
            d = drow[a2 + e];
          } else {
            const float dx = xi0 - xb[3 * e], dy = xi1 - xb[3 * e + 1], dz = xi2 - xb[3 * e + 2];
            d = __builtin_amdgcn_sqrtf((dx * dx + dy * dy) + dz * dz);  // (v_sqrt_f32, 1 ulp: d only enters as d d)
          }
          v[e] = __expf(-1.0f * crow[e] * (d * d)) * (m1_ * mb[e]);
        }
      }
    }
    bf16x8 h, m, l;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      __bf16 hh, mm, ll;
      split3(v[e], hh, mm, ll);
      h[e] = hh; m[e] = mm; l[e] = ll;
    }
    __bf16* dst = Fs + buf * A_CHUNK + b6_off(frow, fslot);
    *reinterpret_cast<bf16x8*>(dst) = h;
    *reinterpret_cast<bf16x8*>(dst + PROWS * BK) = m;
    *reinterpret_cast<bf16x8*>(dst + 2 * PROWS * BK) = l;
  };
  // one 32-k chunk of a product: the wave's 32 x 32 tile, six split terms per 16-k step, smallest first
  const int fx = (l31 >> 2) & 3;
  const int a_off = (32 * rw + l31) * BK, w_off = (32 * cw + l31) * BK;
  constexpr int TA[6] = {1, 2, 0, 1, 0, 0}, TB[6] = {1, 0, 2, 0, 1, 0};
  auto mma_chunk = [&](f32x16& acc, const __bf16* __restrict__ Achunk, const __bf16* __restrict__ Wchunk) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int so = 8 * ((2 * ks + hk) ^ fx);
      bf16x8 af[3], bf[3];
#pragma unroll
      for (int p = 0; p < 3; ++p) {
        af[p] = *reinterpret_cast<const bf16x8*>(Achunk + p * PROWS * BK + a_off + so);
        bf[p] = *reinterpret_cast<const bf16x8*>(Wchunk + p * PC * BK + w_off + so);
      }
#pragma unroll
      for (int term = 0; term < 6; ++term) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[TA[term]], bf[TB[term]], acc, 0, 0, 0);
    }
  };
  float* T1s = reinterpret_cast<float*>(Fs + A_CHUNK);  // feature buffer 1, free once distance_embedding[0] is done: [21][64] | [2 max_dist + 1][64]
  float* T2s = T1s + kAAf * PC;
  // epilogue of a hidden layer: v = relu(acc + bias (+ gathered table rows)) -> split planes into Act (the next layer's A operand), tape
  // D 32x32: column = l31 (+ 32 cw), row = (r & 3) + 8 (r >> 2) + 4 hk (+ 32 rw)
  auto store_act = [&](const f32x16& acc, const float* __restrict__ bias, bool tables, float* __restrict__ tape) {
    const int col = 32 * cw + l31;
    const float bv = bias[col];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int lrow = 32 * rw + (r & 3) + 8 * (r >> 2) + 4 * hk;
      float o = acc[r] + bv;
      if (tables) {  // the folded embedding-table segments of the concatenation (staged in LDS: 21 rows of T1 for this i, all of T2)
        const RowInfo inf = info[lrow];
        o += T1s[(inf.idx - si * kAAf) * PC + col] + inf.same * T2s[inf.rel * PC + col];
      }
      o = fmaxf(o, 0.0f);
      if (TAPE && tape != nullptr) tape[(tile_row0 - a.row0 + lrow) * PC + col] = o;
      __bf16 hh, mm, ll;
      split3(o, hh, mm, ll);
      __bf16* dst = Act + (col >> 5) * A_CHUNK + b6_off(lrow, (col & 31) >> 3) + (col & 7);
      dst[0] = hh;
      dst[PROWS * BK] = mm;
      dst[2 * PROWS * BK] = ll;
    }
  };
  auto zero = [](f32x16& acc) {
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  };
  __syncthreads();  // row info, xj, mj are in LDS

  // One chunk step of the layer pipeline: chunk g of the weight stream is in LDS buffer g % 2, `work` is what has to be produced for the
  // NEXT step (feature chunk c + 1) - it and the weight hand-over run before the MFMAs on waves 0-3 and after them on waves 4-7, so the
  // two waves of a SIMD spend the step in different pipes (vector ALU | matrix cores) instead of in lockstep.
  int g = 0;
  f32x16 acc;
  auto chunk_step = [&](const __bf16* __restrict__ Achunk, auto&& work) {
    if (wv < 4) {
      work();
      store_w((g + 1) & 1);
      load_w(g + 2);
      mma_chunk(acc, Achunk, Ws + (g & 1) * W_CHUNK);
    } else {
      mma_chunk(acc, Achunk, Ws + (g & 1) * W_CHUNK);
      store_w((g + 1) & 1);
      load_w(g + 2);
      work();
    }
    ++g;
    __syncthreads();
  };
  auto nothing = [] {};
  // ================= distance_embedding[0]: features (16 A -> nch0 chunks) x dw0^T
  zero(acc);
  load_w(0);
  gen_features(0, 0);
  store_w(0);
  load_w(1);
  __syncthreads();
  for (int c = 0; c < a.nch0; ++c) {
    if (c + 1 < a.nch0) chunk_step(Fs + (c & 1) * A_CHUNK, [&] { gen_features(c + 1, (c + 1) & 1); });
    else chunk_step(Fs + (c & 1) * A_CHUNK, nothing);
  }
  store_act(acc, a.db0, false, a.tape_h1);
  // ================= 64 -> 64 layers: A = Act (2 chunks, + the dihedral-encoding chunk in Fs for mlp[0])
  auto layer64 = [&](int nchunks) {
    zero(acc);
    __syncthreads();  // the A operand (written by the previous epilogue) is complete
    for (int c = 0; c < nchunks; ++c) chunk_step(c < 2 ? Act + c * A_CHUNK : Fs, nothing);
  };
  layer64(2);
  store_act(acc, a.db2, false, a.tape_df);
  for (int idx = tid; idx < kAAf * PC; idx += 512) T1s[idx] = a.T1[si * kAAf * PC + idx];
  for (int idx = tid; idx < (2 * a.max_dist + 1) * PC; idx += 512) T2s[idx] = a.T2[idx];
  // the dihedral-encoding chunk of mlp[0]'s input: AngularEncoding(2) of the two pairwise dihedrals (:20-54, :299-301), 18 features
  {
    float enc[18];
#pragma unroll
    for (int t_ = 0; t_ < 2; ++t_) {
      const float x = a.pdih[(tile_row0 + frow) * 2 + t_];
      enc[9 * t_] = x;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float f = k < 2 ? static_cast<float>(k + 1) : 1.0f / static_cast<float>(k - 1);
        enc[9 * t_ + 1 + k] = sinf(f * x);
        enc[9 * t_ + 5 + k] = cosf(f * x);
      }
    }
    bf16x8 h, m, l;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int k = 8 * fslot + e;
      float val = 0.0f;
#pragma unroll
      for (int kk = 0; kk < 18; ++kk) val = k == kk ? enc[kk] : val;
      __bf16 hh, mm, ll;
      split3(val, hh, mm, ll);
      h[e] = hh; m[e] = mm; l[e] = ll;
    }
    __bf16* dst = Fs + b6_off(frow, fslot);
    *reinterpret_cast<bf16x8*>(dst) = h;
    *reinterpret_cast<bf16x8*>(dst + PROWS * BK) = m;
    *reinterpret_cast<bf16x8*>(dst + 2 * PROWS * BK) = l;
  }
  layer64(3);
  store_act(acc, a.mb0, true, a.tape_m1);
  layer64(2);
  store_act(acc, a.mb2, false, a.tape_m2);
  if (a.out == nullptr) return;  // (the backward's recompute: the tape is complete)
  layer64(2);
  {  // output row: (acc + b) x CA mask of the pair (:269-271, :312)
    const int col = 32 * cw + l31;
    const float bv = a.mb4[col];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int lrow = 32 * rw + (r & 3) + 8 * (r >> 2) + 4 * hk;
      a.out[(tile_row0 + lrow) * PC + col] = (acc[r] + bv) * info[lrow].mask;
    }
  }
}

static bool g_pair_fused = true;  // diagnostics (diffab_debug_set_attn_variant bit 2): off = the unfused launches everywhere
void set_pair_embed_fused(bool on) { g_pair_fused = on; }
bool pair_embed_fused_supported(const diffab_ctx_dims* d) {
  return g_pair_fused && d->C == PC && d->K % PROWS == 0 && d->A <= kMaxA && d->A > kCAf &&
         static_cast<size_t>(kAAf + 2 * d->max_dist + 1) * PC * sizeof(float) <= static_cast<size_t>(A_CHUNK) * 2;  // the tables fit a feature buffer
}

// floats of the prepared operands behind the caller's workspace pointer: planes (bf16) + tables
size_t pair_embed_fused_prep_floats(const diffab_ctx_dims* d) {
  const int nch0 = (d->A * 16 + BK - 1) / BK;
  const size_t plane_bf16 = static_cast<size_t>(nch0 + 2 + 3 + 2 + 2) * W_CHUNK;
  return plane_bf16 / 2 + 64 + static_cast<size_t>(kAAf * kAAf + 2 * d->max_dist + 1) * PC + static_cast<size_t>(kAAf) * kAAf * d->A * d->A + 64;
}

// prep: pair_embed_fused_prep_floats(d) floats, 256-byte aligned.  tapes (nullable, all or none): [B K K][64] floats each.
// coef_sp_out (nullable): receives the pointer to the softplus table inside prep (the backward needs it too)
int launch_pair_embed_fused(const diffab_ctx_dims* d, const diffab_pair_emb_weights* w, const int64_t* seq_idx, const float* distmat,
                            const float* xyz, const float* pairwise_dihedrals, const int64_t* residue_idx, int32_t residue_idx_batch_stride,
                            const int64_t* chain_idx, const float* atom_mask, const uint8_t* sequence_context_mask, float* out, float* prep,
                            float* tape_h1, float* tape_df, float* tape_m1, float* tape_m2, int64_t row0, int64_t nrows, hipStream_t st,
                            const float** coef_sp_out) {
  DIFFAB_REQUIRE(pair_embed_fused_supported(d) && prep && (reinterpret_cast<uintptr_t>(prep) & 15) == 0 && nrows % PROWS == 0, DIFFAB_ERR_ARG,
                 "pair_embed_fused: unsupported operands");
  const int C = PC, AA2 = d->A * d->A, W = 3 * C + 18;
  const int nch0 = (d->A * 16 + BK - 1) / BK;
  __bf16* pl = reinterpret_cast<__bf16*>(prep);
  __bf16* pl_dw0 = pl;
  __bf16* pl_dw2 = pl_dw0 + static_cast<size_t>(nch0) * W_CHUNK;
  __bf16* pl_m0 = pl_dw2 + 2 * W_CHUNK;
  __bf16* pl_m2 = pl_m0 + 3 * W_CHUNK;
  __bf16* pl_m4 = pl_m2 + 2 * W_CHUNK;
  float* T1 = prep + (static_cast<size_t>(nch0 + 9) * W_CHUNK) / 2 + 64;
  float* T2 = T1 + kAAf * kAAf * C;
  float* coef_sp = T2 + (2 * d->max_dist + 1) * C;
  auto split = [&](const float* Wm, int ldw, int col0, int ncols, int nch, int atoms, __bf16* dst) {
    hipLaunchKernelGGL(wsplit64_kernel, dim3((PC * nch * BK + 255) / 256), dim3(256), 0, st, Wm, ldw, col0, ncols, nch, atoms, dst);
  };
  split(w->dw0, AA2, 0, AA2, nch0, d->A, pl_dw0);
  split(w->dw2, C, 0, C, 2, 0, pl_dw2);
  split(w->mw0, W, 2 * C, C, 2, 0, pl_m0);
  split(w->mw0, W, 3 * C, 18, 1, 0, pl_m0 + 2 * W_CHUNK);
  split(w->mw2, C, 0, C, 2, 0, pl_m2);
  split(w->mw4, C, 0, C, 2, 0, pl_m4);
  hipLaunchKernelGGL(fold_table64_kernel, dim3(kAAf * kAAf), dim3(64), 0, st, w->aa_pair_emb, kAAf * kAAf, w->mw0, W, 0, T1);
  hipLaunchKernelGGL(fold_table64_kernel, dim3(2 * d->max_dist + 1), dim3(64), 0, st, w->relpos_emb, 2 * d->max_dist + 1, w->mw0, W, C, T2);
  launch_softplus_table(w->pair2distcoef, kAAf * kAAf * AA2, coef_sp, st);
  DIFFAB_LAUNCH_CHECK();
  if (coef_sp_out) *coef_sp_out = coef_sp;
  FusedArgs a{};
  a.seq = seq_idx; a.seq_m = sequence_context_mask; a.distmat = distmat; a.xyz = xyz; a.amask = atom_mask; a.pdih = pairwise_dihedrals;
  a.resid = residue_idx; a.resid_bstride = residue_idx_batch_stride; a.chain = chain_idx;
  a.coef_sp = coef_sp; a.pl_dw0 = pl_dw0; a.pl_dw2 = pl_dw2; a.pl_m0 = pl_m0; a.pl_m2 = pl_m2; a.pl_m4 = pl_m4;
  a.db0 = w->db0; a.db2 = w->db2; a.mb0 = w->mb0; a.mb2 = w->mb2; a.mb4 = w->mb4;
  a.T1 = T1; a.T2 = T2; a.out = out;
  a.tape_h1 = tape_h1; a.tape_df = tape_df; a.tape_m1 = tape_m1; a.tape_m2 = tape_m2;
  a.K = d->K; a.A = d->A; a.max_dist = d->max_dist; a.nch0 = nch0;
  a.row0 = row0; a.nrows = nrows;
  const dim3 grid(static_cast<unsigned>(nrows / PROWS));
  if (tape_h1 != nullptr) {
    DIFFAB_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(pair_embed_fused_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                         static_cast<int>(kFusedLdsBytes)));
    hipLaunchKernelGGL(pair_embed_fused_kernel<true>, grid, dim3(512), kFusedLdsBytes, st, a);
  } else {
    DIFFAB_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(pair_embed_fused_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                         static_cast<int>(kFusedLdsBytes)));
    hipLaunchKernelGGL(pair_embed_fused_kernel<false>, grid, dim3(512), kFusedLdsBytes, st, a);
  }
  DIFFAB_LAUNCH_CHECK();
  return DIFFAB_OK;
}

}  // namespace diffab
