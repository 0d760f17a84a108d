// pair_chain_bwd.hip - the PairEmbedding backward on the matrix cores (round 6): four kernels + a slab reduce, one set per chunk of pair rows.
//
// Reference: autograd of PairEmbedding.forward, /root/reference/diffab_pytorch/diffab_pytorch.py:186-312 (distance_embedding :212-217,
// the three Linear layers of `mlp` :219-229, the embedding tables and pair2distcoef :190-210, the atom-mask product at the end of
// forward).  With the hidden activations of a chunk available (recomputed by pair_embed_fused_kernel<true>, or read from the forward's
// tape: diffab_pair_embedding_fwd_taped) the backward was a chain of small launches bound by row-buffer traffic and by LDS atomics.
//
//   pair_chain_bwd_kernel        the 64-wide tail: atom mask, four masked d x products, their weight and bias gradients (below)
//   pair_table_mfma_kernel       G1[s_i 21 + s_j] += g, G2[rel] += same g as one-hot products (was: LDS atomics)
//   pair_dist_bwd_mfma_kernel    d softplus(pair2distcoef) class sums from a materialised d din (the unfused backward's form)
//   pair_dist_bwd_fused64_kernel ... with d din = d h1 W formed in the kernel (the fused backward's form)
//   parts_reduce_kernel          per-work-group partial sums -> their destinations
//
// ---- pair_chain_bwd_kernel.  The backward of the four 64 x 64 layers was a mask kernel, four masked d x products, and a four-product
// weight-gradient launch: 11 passes over 126 MB row buffers per chunk of 30 patches (2.7 GB, 875 us).
// Here a work-group walks 128-row tiles and keeps a tile on the CU through the whole chain:
//
//   d A   = d out * atom_mask_i[CA] * atom_mask_j[CA]                                   (pair_mask_bwd_kernel)
//   layer 0: d mlp[4].W += d A^T  m2,   d B   = (d A   mlp[4].W)          * [m2 > 0]
//   layer 1: d mlp[2].W += d B^T  m1,   d C   = (d B   mlp[2].W)          * [m1 > 0]     -> global (the table sums and the dihedral
//   layer 2: d mlp[0].W[:, 2C:3C] += d C^T df,  d df = (d C mlp[0].W[:, 2C:3C]) * [df > 0]                  columns read it)
//   layer 3: d dist[2].W += d df^T h1,  d h1  = (d df  dist[2].W)         * [h1 > 0]     -> global (distance_embedding[0]'s backward)
//   and the four bias gradients = column sums of d A, d B, d C, d df.
//
// It reads d out and the four activations once (5 x 32 KiB per tile) and writes d C and d h1 (2 x 32 KiB): 880 MB per chunk, 254 us.
// Arithmetic: bf16 x 6 split products (three exact bf16 pieces per operand, six partial products, fp32 accumulation - gemm_bf16x6.hip):
// fp32 accuracy without any scale bookkeeping for gradients that span decades.  Per layer a tile's d y and x sit in LDS as split
// planes [128 rows][64 columns] in ONE orientation that serves three reads: the d x product takes d y rows as the MFMA's B operand
// (ds_read_b128; the weights are the A operand, so a lane ends with four consecutive columns of one row: the next layer's d y goes back
// into the planes as 8-byte pieces, mlp_chain_tile.h's arrangement), the weight gradient contracts over the ROWS and takes both
// operands through the transposing read ds_read_b64_tr_b16 (gemm_tn_b6_kernel's arrangement), the ReLU mask is the sign of x's high
// plane.  Weight fragments come pre-split and fragment-ordered from a 96 KiB prep buffer (L2-resident), one layer at a time.  The
// weight-gradient accumulators of all tiles of a work-group live in LDS (64 KiB; as registers the kernel spilled), the bias sums in
// registers; both leave as one slab per work-group for parts_reduce_kernel.  LDS: 96 KiB of planes + 64 KiB = all 160 KiB of a CU.
#include "common.h"
#include "denoiser_internal.h"
#include "rowgemm_b6_tile.h"

namespace diffab {
namespace {
using b6tile::split3;
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
#define PC_FENCE() asm volatile("" ::: "memory")

constexpr int PC_C = 64, PC_ROWS = 128;
constexpr int PC_PLANE = PC_ROWS * PC_C;                 // bf16 elements of one plane of one operand (16 KiB)
constexpr int PC_LDS_BYTES = 2 * 3 * PC_PLANE * 2 + 4 * 64 * 64 * 4;  // d y planes | x planes | the four weight-gradient accumulators: 160 KiB
constexpr int PC_PART = 4 * 64 * 64 + 4 * 64;            // floats of one work-group's partial sums: four 64 x 64 weight gradients | four biases
constexpr int PC_WFRAG = 4 * 2 * 3 * 4 * 64 * 8;         // bf16 elements of the prep buffer: [layer][k step][plane][column tile][lane][8]
// bf16 element offset of the 16-byte chunk `chunk` (8 columns) of row `row`: rows are 128 bytes = all 32 banks, so the chunk index is
// swizzled with the row - four consecutive rows (a transposing read's 16-lane group) land on four different chunk PAIRS, eight
// consecutive rows (a ds_read_b128's eight lanes per cycle) on eight different chunks
__device__ __forceinline__ int pc_off(int row, int chunk) { return row * PC_C + 8 * (chunk ^ (((row & 3) << 1) | ((row >> 2) & 1))); }

struct PairChainArgs {
  const float* d_out;   // [B K K][64], the whole tensor
  const float* amask;   // [B K][A]
  const float* X[4];    // taped activations of the chunk: m2, m1, df, h1, [nrows][64]
  const __bf16* wfrag;  // prep buffer (pair_chain_prep_kernel)
  float* dC;            // [nrows][64]
  float* dh1;           // [nrows][64]
  float* part;          // [work-groups][PC_PART]: this launch's weight- and bias-gradient sums of every work-group
  int64_t row0, nrows;  // the chunk's rows inside d_out; nrows a multiple of 128
  int K, A, ca;
};

// W_L[n][i] (rows ldw floats apart) -> A fragments of W_L^T for v_mfma_f32_16x16x32_bf16: fragment (L, ks, plane, mi), lane (l15, g):
// the eight values W_L[32 ks + 8 g + c][16 mi + l15], c = 0..7
// (columns i >= nc_L of layer L: zeros - the distance kernel's last 64-column block of a 228-wide weight)
__global__ void pair_chain_prep_kernel(const float* __restrict__ W0, int ld0, const float* __restrict__ W1, int ld1, const float* __restrict__ W2,
                                       int ld2, const float* __restrict__ W3, int ld3, int nc0, int nc1, int nc2, int nc3,
                                       __bf16* __restrict__ out) {
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;  // (L, ks, mi, lane, c)
  if (gid >= 4 * 2 * 4 * 64 * 8) return;
  const int c = gid & 7, lane = (gid >> 3) & 63, mi = (gid >> 9) & 3, ks = (gid >> 11) & 1, L = gid >> 12;
  const float* W = L == 0 ? W0 : L == 1 ? W1 : L == 2 ? W2 : W3;
  const int ld = L == 0 ? ld0 : L == 1 ? ld1 : L == 2 ? ld2 : ld3;
  const int nc = L == 0 ? nc0 : L == 1 ? nc1 : L == 2 ? nc2 : nc3;
  const int n = 32 * ks + 8 * (lane >> 4) + c, i = 16 * mi + (lane & 15);
  __bf16 h, m, l;
  split3(i < nc ? W[n * ld + i] : 0.0f, h, m, l);
  const size_t base = ((static_cast<size_t>(L * 2 + ks) * 3) * 4 + mi) * 512 + lane * 8 + c;
  out[base] = h;
  out[base + 4 * 512] = m;
  out[base + 8 * 512] = l;
}

__global__ __launch_bounds__(512) void pair_chain_bwd_kernel(PairChainArgs a) {
  extern __shared__ __attribute__((aligned(16))) __bf16 pc_lds[];
  __bf16* Yp = pc_lds;                 // d y planes [3][128][64]
  __bf16* Xp = pc_lds + 3 * PC_PLANE;  // x planes
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, l15 = lane & 15, g = lane >> 4;
  const int ntiles = static_cast<int>(a.nrows / PC_ROWS);
  // ---- staging: thread -> rows (tid >> 4) + 32 j, float4 column tid & 15 of a 128 x 64 tile
  const int s_row = tid >> 4, s_c4 = tid & 15;
  auto load_tile = [&](f32x4 (&r)[4], const float* src, int tile) {
    const float* p = src + (static_cast<int64_t>(tile) * PC_ROWS + s_row) * PC_C + 4 * s_c4;
#pragma unroll
    for (int j = 0; j < 4; ++j) r[j] = *reinterpret_cast<const f32x4*>(p + 32 * j * PC_C);
  };
  f32x4 csumA = {0.f, 0.f, 0.f, 0.f};  // bias gradient of layer 0: column sums of d A over this thread's rows
  auto load_dA = [&](f32x4 (&r)[4], int tile) {
    // a tile is 128 consecutive j of ONE (patch, i) (K is a multiple of 128): the divisions are per tile and wave-uniform
    const int64_t t0 = a.row0 + static_cast<int64_t>(tile) * PC_ROWS;
    const int64_t bi = t0 / a.K;  // b K + i
    const int j0 = static_cast<int>(t0 - bi * a.K);
    const int64_t bK = (bi / a.K) * a.K;
    const float mi_ = a.amask[bi * a.A + a.ca];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int rl = s_row + 32 * j;
      const float mk = mi_ * a.amask[(bK + j0 + rl) * a.A + a.ca];
      r[j] = *reinterpret_cast<const f32x4*>(a.d_out + (t0 + rl) * PC_C + 4 * s_c4) * mk;
    }
  };
  auto stage = [&](__bf16* planes, const f32x4 (&r)[4]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      bf16x4 h, m, l;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        __bf16 hh, mm, ll;
        split3(r[j][c], hh, mm, ll);
        h[c] = hh; m[c] = mm; l[c] = ll;
      }
      __bf16* dst = planes + pc_off(s_row + 32 * j, s_c4 >> 1) + 4 * (s_c4 & 1);
      *reinterpret_cast<bf16x4*>(dst) = h;
      *reinterpret_cast<bf16x4*>(dst + PC_PLANE) = m;
      *reinterpret_cast<bf16x4*>(dst + 2 * PC_PLANE) = l;
    }
  };
  // ---- d x product: wave = (column tile mi of the OUTPUT, row half ch): four 16 x 16 tiles D^T[i = 16 mi + 4 g + e][row 64 ch + 16 ct + l15]
  const int mi = wv & 3, ch = wv >> 2;
  // ---- weight gradient: wave = (n tile nt, i tiles 2 (wv >> 2) + t): D[n = 16 nt + 4 g + e][i = 16 it + l15], contraction over the tile's rows
  const int nt = wv & 3, it0 = 2 * (wv >> 2);
  const int q = l15 >> 2, pp = l15 & 3;
  auto frag_tr = [&](const __bf16* plane, int kk, int cb) -> bf16x8 {  // rows 32 kk + 8 g + 4 rd + q, the 16-column block cb
    const int r0 = 32 * kk + 8 * g + q;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (s16x4 __attribute__((address_space(3)))*)(plane + pc_off(r0, 2 * cb + (pp >> 1)) + 4 * (pp & 1)));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (s16x4 __attribute__((address_space(3)))*)(plane + pc_off(r0 + 4, 2 * cb + (pp >> 1)) + 4 * (pp & 1)));
    const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
  };
  constexpr int TA[6] = {1, 2, 0, 1, 0, 0}, TB[6] = {1, 0, 2, 0, 1, 0};  // (mid,mid) (lo,hi) (hi,lo) (mid,hi) (hi,mid) (hi,hi)
  // weight-gradient accumulators over all tiles of this work-group: in LDS (32 registers otherwise - the kernel spilled), each wave its
  // own 16-byte slots [layer][wave][tile t][lane]
  f32x4* gws = reinterpret_cast<f32x4*>(pc_lds + 6 * PC_PLANE);
  auto gw_slot = [&](int L, int t) -> f32x4* { return gws + ((L * 8 + wv) * 2 + t) * 64 + lane; };
#pragma unroll
  for (int L = 0; L < 4; ++L)
#pragma unroll
    for (int t = 0; t < 2; ++t) *gw_slot(L, t) = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 gbv[3];    // bias gradients of layers 1..3: this lane's four columns 16 mi + 4 g + e over its rows
#pragma unroll
  for (int L = 0; L < 3; ++L) gbv[L] = f32x4{0.f, 0.f, 0.f, 0.f};

  int tile = blockIdx.x;
  f32x4 RA[4], RX[4];
  if (tile < ntiles) {
    load_dA(RA, tile);
    load_tile(RX, a.X[0], tile);
  }
  for (; tile < ntiles; tile += gridDim.x) {
#pragma unroll
    for (int j = 0; j < 4; ++j) csumA += RA[j];
    stage(Yp, RA);
    stage(Xp, RX);
    PC_FENCE();
    __syncthreads();
#pragma unroll
    for (int L = 0; L < 4; ++L) {
      // ---- requests of this phase: the weight fragments of the layer, the next layer's x (or the next tile's first two operands)
      // (the fragment addresses do not depend on the tile: behind an opaque zero, or the compiler hoists all four layers' fragments - 96
      // registers - out of the tile loop and spills)
      int zoff = 0;
      asm volatile("" : "+v"(zoff));
      bf16x8 wf[2][3];
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int p = 0; p < 3; ++p)
          wf[ks][p] = *reinterpret_cast<const bf16x8*>(a.wfrag + ((static_cast<size_t>(L * 2 + ks) * 3 + p) * 4 + mi) * 512 + lane * 8 + zoff);
      const int tnext = tile + static_cast<int>(gridDim.x);
      if (L < 3) load_tile(RX, a.X[L + 1], tile);
      else if (tnext < ntiles) {
        load_dA(RA, tnext);
        load_tile(RX, a.X[0], tnext);
      }
      PC_FENCE();
      // ---- weight gradient of the layer: d y^T x over the tile's 128 rows
      f32x4 gw[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) gw[t] = *gw_slot(L, t);
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        bf16x8 fa[3], fb[2][3];
#pragma unroll
        for (int p = 0; p < 3; ++p) {
          fa[p] = frag_tr(Yp + p * PC_PLANE, kk, nt);
#pragma unroll
          for (int t = 0; t < 2; ++t) fb[t][p] = frag_tr(Xp + p * PC_PLANE, kk, it0 + t);
        }
#pragma unroll
        for (int term = 0; term < 6; ++term)
#pragma unroll
          for (int t = 0; t < 2; ++t)
            gw[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[TA[term]], fb[t][TB[term]], gw[t], 0, 0, 0);
      }
#pragma unroll
      for (int t = 0; t < 2; ++t) *gw_slot(L, t) = gw[t];
      // ---- d x = d y W, masked by the ReLU below it
      f32x4 o[4];
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) {
        const int r = 64 * ch + 16 * ct + l15;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          bf16x8 fy[3];
#pragma unroll
          for (int p = 0; p < 3; ++p) fy[p] = *reinterpret_cast<const bf16x8*>(Yp + p * PC_PLANE + pc_off(r, g + 4 * ks));
#pragma unroll
          for (int term = 0; term < 6; ++term) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ks][TA[term]], fy[TB[term]], acc, 0, 0, 0);
        }
        // columns 16 mi + 4 g .. + 3 of row r: x > 0 <=> its high piece > 0 (bf16 has fp32's exponent range)
        const bf16x4 xh = *reinterpret_cast<const bf16x4*>(Xp + pc_off(r, 2 * mi + (g >> 1)) + 4 * (g & 1));
#pragma unroll
        for (int e = 0; e < 4; ++e) o[ct][e] = static_cast<float>(xh[e]) > 0.0f ? acc[e] : 0.0f;
      }
      if (L < 3) {
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) gbv[L] += o[ct];
      }
      if (L == 1 || L == 3) {
        float* out = (L == 1 ? a.dC : a.dh1) + (static_cast<int64_t>(tile) * PC_ROWS + 64 * ch + l15) * PC_C + 16 * mi + 4 * g;
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) *reinterpret_cast<f32x4*>(out + 16 * ct * PC_C) = o[ct];
      }
      PC_FENCE();
      __syncthreads();  // every wave has read this layer's planes
      if (L < 3) {
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) {
          const int r = 64 * ch + 16 * ct + l15;
          bf16x4 h, m, l;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            __bf16 hh, mm, ll;
            split3(o[ct][e], hh, mm, ll);
            h[e] = hh; m[e] = mm; l[e] = ll;
          }
          __bf16* dst = Yp + pc_off(r, 2 * mi + (g >> 1)) + 4 * (g & 1);
          *reinterpret_cast<bf16x4*>(dst) = h;
          *reinterpret_cast<bf16x4*>(dst + PC_PLANE) = m;
          *reinterpret_cast<bf16x4*>(dst + 2 * PC_PLANE) = l;
        }
        stage(Xp, RX);
        PC_FENCE();
        __syncthreads();
      }
      // (L == 3: the loop head stages the next tile and holds the barrier)
    }
  }
  // ---- the gradients of this work-group -> its own slab of `part` (256 work-groups adding 16 640 values each to the same addresses
  // cost more atomics time than the kernel's tiles; parts_reduce_kernel sums the slabs); bias sums through LDS first (all planes are
  // dead behind the last barrier)
  float* part = a.part + static_cast<size_t>(blockIdx.x) * PC_PART;
#pragma unroll
  for (int L = 0; L < 4; ++L)
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const f32x4 gw = *gw_slot(L, t);
#pragma unroll
      for (int e = 0; e < 4; ++e) part[(L * 64 + 16 * nt + 4 * g + e) * 64 + 16 * (it0 + t) + l15] = gw[e];
    }
  float* red = reinterpret_cast<float*>(pc_lds);  // [4 layers][64 columns]
  if (tid < 256) red[tid] = 0.0f;
  __syncthreads();
#pragma unroll
  for (int c = 0; c < 4; ++c) atomicAdd(red + 4 * s_c4 + c, csumA[c]);
#pragma unroll
  for (int L = 0; L < 3; ++L)
#pragma unroll
    for (int e = 0; e < 4; ++e) atomicAdd(red + 64 * (L + 1) + 16 * mi + 4 * g + e, gbv[L][e]);
  __syncthreads();
  if (tid < 256) part[4 * 64 * 64 + tid] = red[tid];
}

// ================================================================== the embedding-table segments of mlp[0]'s backward as matrix products
// G1[s_i 21 + s_j][c] += g[row][c] and G2[rel][c] += same g[row][c] (pair_table_scatter_kernel's sums) are products with one-hot
// operands: G = OneHot^T g, contraction over the rows.  As LDS atomics they were this backward's slowest small kernel (290 us per chunk:
// a 64-lane ds_add_f32 takes about 130 cycles; 48 us with the atomics taken out).  Here the matrix cores do the class reduction: the
// g tile is staged as three bf16 planes (exact fp32 split, the chain kernel's layout and transposing reads), the one-hot A fragments
// are built in registers from the rows' class ids (1.0 and small integers are exact in bf16: three partial products, no rounding
// beyond the fp32 accumulation), and the 33 x 4 output tiles (441 -> 448 pair classes, 65 -> 80 relative positions, 64 channels) stay
// in the accumulators of the work-group's eight waves over all its tiles.  No table in LDS, no atomics; the work-group's sums leave
// as one slab for launch_parts_reduce.
constexpr int PT_MT = 33;  // 16-class output tiles: 28 for the pair classes, 5 for the relative positions
struct PairTableArgs {
  const float* g;  // [nrows][64] (chunk-local)
  const int64_t* seq; const uint8_t* seq_m; const int64_t* resid; const int64_t* chain;
  float* part;  // [work-groups][(n_pair_rows + n_rel_rows) * 64]
  int64_t row0, nrows;
  int K, resid_bstride, max_dist, n_aa, unk;
};
__global__ __launch_bounds__(512) void pair_table_mfma_kernel(PairTableArgs a) {
  extern __shared__ __attribute__((aligned(16))) __bf16 pt_lds[];
  __bf16* Gp = pt_lds;                                            // g planes [3][128][64]
  int* cls = reinterpret_cast<int*>(pt_lds + 3 * PC_PLANE);       // [128] pair class of row j
  int* rel = cls + PC_ROWS;                                       // [128] relative-position row
  float* same = reinterpret_cast<float*>(rel + PC_ROWS);          // [128] chain_i chain_j
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, l15 = lane & 15, g = lane >> 4;
  const int ntiles = static_cast<int>(a.nrows / PC_ROWS);
  const int n_pair = a.n_aa * a.n_aa, n_rel = 2 * a.max_dist + 1;
  const int s_row = tid >> 4, s_c4 = tid & 15;  // staging: rows (tid >> 4) + 32 j, float4 column tid & 15
  const int q = l15 >> 2, pp = l15 & 3;
  auto frag_tr = [&](const __bf16* plane, int kk, int cb) -> bf16x8 {
    const int r0 = 32 * kk + 8 * g + q;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (s16x4 __attribute__((address_space(3)))*)(plane + pc_off(r0, 2 * cb + (pp >> 1)) + 4 * (pp & 1)));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (s16x4 __attribute__((address_space(3)))*)(plane + pc_off(r0 + 4, 2 * cb + (pp >> 1)) + 4 * (pp & 1)));
    const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
  };
  // this wave's class tiles: wv + 8 m (< 33); four channel tiles each
  constexpr int PT_M = 5;
  f32x4 acc[PT_M][4];
#pragma unroll
  for (int m = 0; m < PT_M; ++m)
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) acc[m][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 R[4];
  auto load_g = [&](int tile) {
    const float* p = a.g + (static_cast<int64_t>(tile) * PC_ROWS + s_row) * PC_C + 4 * s_c4;
#pragma unroll
    for (int j = 0; j < 4; ++j) R[j] = *reinterpret_cast<const f32x4*>(p + 32 * j * PC_C);
  };
  int tile = blockIdx.x;
  if (tile < ntiles) load_g(tile);
  for (; tile < ntiles; tile += gridDim.x) {
    // ---- the tile's 128 rows are 128 consecutive j of one (patch, i): class ids of the rows, then the planes of g
    if (tid < PC_ROWS) {
      const int64_t t0 = a.row0 + static_cast<int64_t>(tile) * PC_ROWS;
      const int64_t bi = t0 / a.K, b = bi / a.K;
      const int i = static_cast<int>(bi - b * a.K), j = static_cast<int>(t0 - bi * a.K) + tid;
      const int64_t ri = bi, rj = b * a.K + j;
      const int64_t si = (a.seq_m && !a.seq_m[ri]) ? a.unk : a.seq[ri], sj = (a.seq_m && !a.seq_m[rj]) ? a.unk : a.seq[rj];
      int64_t rl = a.resid[b * a.resid_bstride + i] - a.resid[b * a.resid_bstride + j];
      rl = rl < -a.max_dist ? -a.max_dist : (rl > a.max_dist ? a.max_dist : rl);
      cls[tid] = static_cast<int>(si * a.n_aa + sj);
      rel[tid] = static_cast<int>(rl) + a.max_dist;
      same[tid] = static_cast<float>(a.chain[ri] * a.chain[rj]);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      bf16x4 h, m, l;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        __bf16 hh, mm, ll;
        split3(R[j][c], hh, mm, ll);
        h[c] = hh; m[c] = mm; l[c] = ll;
      }
      __bf16* dst = Gp + pc_off(s_row + 32 * j, s_c4 >> 1) + 4 * (s_c4 & 1);
      *reinterpret_cast<bf16x4*>(dst) = h;
      *reinterpret_cast<bf16x4*>(dst + PC_PLANE) = m;
      *reinterpret_cast<bf16x4*>(dst + 2 * PC_PLANE) = l;
    }
    PC_FENCE();
    __syncthreads();
    if (tile + static_cast<int>(gridDim.x) < ntiles) load_g(tile + static_cast<int>(gridDim.x));
    PC_FENCE();
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      // one-hot A fragments: lane (class 16 mt + l15, rows 32 kk + 8 g + c): the element order of the transposing reads of the B side
      const int rb = 32 * kk + 8 * g;
      int ci[8], ri_[8];
      float sm[8];
#pragma unroll
      for (int h4 = 0; h4 < 2; ++h4) {
        const auto cv = *reinterpret_cast<const int __attribute__((ext_vector_type(4)))*>(cls + rb + 4 * h4);
        const auto rv = *reinterpret_cast<const int __attribute__((ext_vector_type(4)))*>(rel + rb + 4 * h4);
        const f32x4 sv = *reinterpret_cast<const f32x4*>(same + rb + 4 * h4);
#pragma unroll
        for (int c = 0; c < 4; ++c) { ci[4 * h4 + c] = cv[c]; ri_[4 * h4 + c] = rv[c]; sm[4 * h4 + c] = sv[c]; }
      }
      bf16x8 fa[PT_M];
#pragma unroll
      for (int m = 0; m < PT_M; ++m) {
        const int mt = wv + 8 * m;
        const int mine = (mt < 28 ? 16 * mt : 16 * (mt - 28)) + l15;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
          const float v = mt < 28 ? (ci[c] == mine ? 1.0f : 0.0f) : (ri_[c] == mine ? sm[c] : 0.0f);
          fa[m][c] = static_cast<__bf16>(v);
        }
      }
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        bf16x8 fb[3];
#pragma unroll
        for (int p = 0; p < 3; ++p) fb[p] = frag_tr(Gp + p * PC_PLANE, kk, nt);
#pragma unroll
        for (int m = 0; m < PT_M; ++m) {
          if (wv + 8 * m >= PT_MT) continue;  // (wave-uniform)
#pragma unroll
          for (int p = 2; p >= 0; --p) acc[m][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[m], fb[p], acc[m][nt], 0, 0, 0);
        }
      }
    }
    PC_FENCE();
    __syncthreads();  // every wave has read the planes and the class ids
  }
  // ---- the work-group's sums: D row 4 g + e = class, column l15 = channel
  float* part = a.part + static_cast<size_t>(blockIdx.x) * (static_cast<size_t>(n_pair + n_rel) * PC_C);
#pragma unroll
  for (int m = 0; m < PT_M; ++m) {
    const int mt = wv + 8 * m;
    if (mt >= PT_MT) continue;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int c_ = (mt < 28 ? 16 * mt : 16 * (mt - 28)) + 4 * g + e;
      const bool ok = mt < 28 ? c_ < n_pair : c_ < n_rel;
      if (!ok) continue;
      float* dst = part + static_cast<size_t>(mt < 28 ? c_ : n_pair + c_) * PC_C + l15;
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) dst[16 * nt] = acc[m][nt][e];
    }
  }
}

// ================================================================== d softplus(pair2distcoef): the class sums on the matrix cores
// g_sp[s_i 21 + s_j][p] += sum over the rows of a (patch, i) group of E[row][p], E = d din (-d^2 din) (pair_dist_bwd_group_kernel's sum:
// din = exp(-c d^2) mask, d din / d c = -d^2 din).  The group's rows share s_i; as LDS atomics on a [21][A A] table the class sum cost
// 512 ds_add_f32 wave instructions per group (578 us per chunk).  Here it is OneHot(s_j)^T E on the matrix cores, 64 atom-pair columns
// at a time: E is formed while the block is staged as three bf16 planes (the chain kernel's layout), the one-hot A fragments are built
// from the rows' s_j, each of the eight waves owns one 16-class x 16-column tile, and a block's sums go straight to the table rows of
// s_i (21 x 64 global atomics per block, as before).  One work-group per 128 consecutive j of a (patch, i): K a multiple of 128.
struct PairDistBwdArgs {
  const int64_t* seq; const uint8_t* seq_m; const float* distmat; const float* xyz; const float* din; const float* ddin;
  float* g_sp;
  int64_t row0;
  int K, A, ld, n_aa, unk;
};
__global__ __launch_bounds__(512) void pair_dist_bwd_mfma_kernel(PairDistBwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) __bf16 pd_lds[];
  __bf16* Ep = pd_lds;                                       // E planes [3][128][64]
  int* sjs = reinterpret_cast<int*>(pd_lds + 3 * PC_PLANE);  // [128] s_j of the rows
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, l15 = lane & 15, g = lane >> 4;
  const int AA2 = a.A * a.A;
  const int64_t lrow0 = static_cast<int64_t>(blockIdx.x) * PC_ROWS;  // first row of the tile inside this launch: 128 consecutive j of one (b, i)
  const int64_t grow0 = a.row0 + lrow0;                               // global pair row (b, i, j0)
  const int64_t ri = grow0 / a.K;                                     // b K + i
  const int j0 = static_cast<int>(grow0 - ri * a.K);                  // (K a multiple of 128)
  const int64_t b = ri / a.K;
  const int64_t si = (a.seq_m && !a.seq_m[ri]) ? a.unk : a.seq[ri];
  if (tid < PC_ROWS) {
    const int64_t rj = b * a.K + j0 + tid;
    sjs[tid] = static_cast<int>((a.seq_m && !a.seq_m[rj]) ? a.unk : a.seq[rj]);
  }
  const int s_row = tid >> 4, s_c4 = tid & 15;  // staging: rows (tid >> 4) + 32 j, columns 4 (tid & 15) .. + 3 of the block
  const int q = l15 >> 2, pp = l15 & 3;
  auto frag_tr = [&](const __bf16* plane, int kk, int cb) -> bf16x8 {
    const int r0 = 32 * kk + 8 * g + q;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (s16x4 __attribute__((address_space(3)))*)(plane + pc_off(r0, 2 * cb + (pp >> 1)) + 4 * (pp & 1)));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (s16x4 __attribute__((address_space(3)))*)(plane + pc_off(r0 + 4, 2 * cb + (pp >> 1)) + 4 * (pp & 1)));
    const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
  };
  const int mt = wv & 1, nt = wv >> 1;  // this wave's tile: classes 16 mt + 4 g + e, columns 16 nt + l15 of the block
  const int nblk = (AA2 + 63) / 64;
  for (int cb = 0; cb < nblk; ++cb) {
    // ---- E of the block: rows s_row + 32 j, atom pairs p = 64 cb + 4 s_c4 + e
    const int pbase = 64 * cb + 4 * s_c4;
    float xa[4][3];
    int a2s[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int p = pbase + e < AA2 ? pbase + e : 0;
      const int a1 = p / a.A;
      a2s[e] = p - a1 * a.A;
      xa[e][0] = xa[e][1] = xa[e][2] = 0.0f;
      if (!a.distmat) {
        const float* pa = a.xyz + (ri * a.A + a1) * 3;
        xa[e][0] = pa[0]; xa[e][1] = pa[1]; xa[e][2] = pa[2];
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int r = s_row + 32 * j;
      const int64_t lr = lrow0 + r, rj = b * a.K + j0 + r;
      f32x4 v = {0.f, 0.f, 0.f, 0.f}, gd = {0.f, 0.f, 0.f, 0.f};
      if (pbase + 3 < a.ld) {  // (ld is a multiple of 4: a float4 is inside the row or outside)
        v = *reinterpret_cast<const f32x4*>(a.din + lr * a.ld + pbase);
        gd = *reinterpret_cast<const f32x4*>(a.ddin + lr * a.ld + pbase);
      }
      bf16x4 h, m, l;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float E = 0.0f;
        if (pbase + e < AA2 && v[e] != 0.0f) {  // v == 0: masked atom pair (or underflow), no gradient
          float d;
          if (a.distmat) {
            d = a.distmat[(grow0 + r) * AA2 + pbase + e];
          } else {
            const float* pb = a.xyz + (rj * a.A + a2s[e]) * 3;
            const float dx = xa[e][0] - pb[0], dy = xa[e][1] - pb[1], dz = xa[e][2] - pb[2];
            d = sqrtf((dx * dx + dy * dy) + dz * dz);
          }
          E = gd[e] * (-(d * d) * v[e]);
        }
        __bf16 hh, mm, ll;
        split3(E, hh, mm, ll);
        h[e] = hh; m[e] = mm; l[e] = ll;
      }
      __bf16* dst = Ep + pc_off(r, s_c4 >> 1) + 4 * (s_c4 & 1);
      *reinterpret_cast<bf16x4*>(dst) = h;
      *reinterpret_cast<bf16x4*>(dst + PC_PLANE) = m;
      *reinterpret_cast<bf16x4*>(dst + 2 * PC_PLANE) = l;
    }
    PC_FENCE();
    __syncthreads();
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const int rb = 32 * kk + 8 * g;
      bf16x8 fa;
#pragma unroll
      for (int h4 = 0; h4 < 2; ++h4) {
        const auto cv = *reinterpret_cast<const int __attribute__((ext_vector_type(4)))*>(sjs + rb + 4 * h4);
#pragma unroll
        for (int c = 0; c < 4; ++c) fa[4 * h4 + c] = static_cast<__bf16>(cv[c] == 16 * mt + l15 ? 1.0f : 0.0f);
      }
#pragma unroll
      for (int p = 2; p >= 0; --p) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, frag_tr(Ep + p * PC_PLANE, kk, nt), acc, 0, 0, 0);
    }
    const int pcol = 64 * cb + 16 * nt + l15;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int sj = 16 * mt + 4 * g + e;
      if (sj < a.n_aa && pcol < AA2 && acc[e] != 0.0f) atomicAdd(a.g_sp + ((si * a.n_aa + sj) * AA2 + pcol), acc[e]);
    }
    PC_FENCE();
    __syncthreads();  // every wave has read the planes
  }
}

// The same sums with d din formed HERE: d din = d h1 . distance_embedding[0].W (64 -> A A columns) was a GEMM launch of its own that wrote
// 448 MB per chunk for pair_dist_bwd_mfma_kernel to read back.  The group's d h1 tile is staged once as split planes; per 64-column
// block the product runs with the weights as the MFMA's A operand (the chain kernel's d x product: fragments from a prep buffer, a lane
// ends with four consecutive atom pairs of one row), E = d din (-d^2 din) is formed in the accumulators' layout and written into the E
// planes as 8-byte pieces, and the one-hot product follows as above.
struct PairDistFusedArgs {
  const int64_t* seq; const uint8_t* seq_m; const float* distmat; const float* xyz; const float* din; const float* dh1;
  const __bf16* wfrag;  // pair_chain_prep_kernel's fragments of W[:, 64 cb .. 64 cb + 63], cb = 0..3
  float* g_sp;
  int64_t row0;
  int K, A, ld, n_aa, unk;
};
// 64-row tiles in 4-wave work-groups (49 KiB of LDS, 144 VGPRs: three groups per CU).  The first form - 128 rows, eight waves, the patch's
// atoms staged in LDS: 120 KiB - ran ONE group per CU, whose load, vector and matrix phases follow each other in lockstep between
// barriers: 389-395 us per chunk against 341 for this one (independent groups overlap their phases; the atoms come through L1).
__global__ __launch_bounds__(256) void pair_dist_bwd_fused64_kernel(PairDistFusedArgs a) {
  extern __shared__ __attribute__((aligned(16))) __bf16 pd_lds[];
  constexpr int TR = 64, PL = TR * PC_C;                // rows of a tile, bf16 elements of one plane
  __bf16* Yp = pd_lds;                                  // d h1 planes [3][64][64]
  __bf16* Ep = pd_lds + 3 * PL;                         // E planes
  int* sjs = reinterpret_cast<int*>(pd_lds + 6 * PL);   // [64] s_j of the rows
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, l15 = lane & 15, g = lane >> 4;
  const int AA2 = a.A * a.A;
  const int64_t lrow0 = static_cast<int64_t>(blockIdx.x) * TR;  // the tile: 64 consecutive j of one (b, i) (K a multiple of 128)
  const int64_t grow0 = a.row0 + lrow0;
  const int64_t ri = grow0 / a.K;
  const int j0 = static_cast<int>(grow0 - ri * a.K);
  const int64_t b = ri / a.K;
  const int64_t si = (a.seq_m && !a.seq_m[ri]) ? a.unk : a.seq[ri];
  if (tid < TR) {
    const int64_t rj = b * a.K + j0 + tid;
    sjs[tid] = static_cast<int>((a.seq_m && !a.seq_m[rj]) ? a.unk : a.seq[rj]);
  }
  {  // d h1 tile -> planes
    const int s_row = tid >> 4, s_c4 = tid & 15;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int r = s_row + 16 * j;
      const f32x4 v = *reinterpret_cast<const f32x4*>(a.dh1 + (lrow0 + r) * PC_C + 4 * s_c4);
      bf16x4 h, m, l;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        __bf16 hh, mm, ll;
        split3(v[c], hh, mm, ll);
        h[c] = hh; m[c] = mm; l[c] = ll;
      }
      __bf16* dst = Yp + pc_off(r, s_c4 >> 1) + 4 * (s_c4 & 1);
      *reinterpret_cast<bf16x4*>(dst) = h;
      *reinterpret_cast<bf16x4*>(dst + PL) = m;
      *reinterpret_cast<bf16x4*>(dst + 2 * PL) = l;
    }
  }
  PC_FENCE();
  __syncthreads();
  const int q = l15 >> 2, pp = l15 & 3;
  auto frag_tr = [&](const __bf16* plane, int kk, int cb) -> bf16x8 {
    const int r0 = 32 * kk + 8 * g + q;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (s16x4 __attribute__((address_space(3)))*)(plane + pc_off(r0, 2 * cb + (pp >> 1)) + 4 * (pp & 1)));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (s16x4 __attribute__((address_space(3)))*)(plane + pc_off(r0 + 4, 2 * cb + (pp >> 1)) + 4 * (pp & 1)));
    const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
  };
  constexpr int TA[6] = {1, 2, 0, 1, 0, 0}, TB[6] = {1, 0, 2, 0, 1, 0};
  const int mi = wv;  // d din product: columns 16 mi + 4 g + e of the block, rows 16 ct + l15
  const int nt = wv;  // one-hot product: classes 16 mt + 4 g + e (mt = 0, 1), columns 16 nt + l15 of the block
  const int nblk = (AA2 + 63) / 64;
  for (int cb = 0; cb < nblk; ++cb) {
    bf16x8 wf[2][3];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int p = 0; p < 3; ++p)
        wf[ks][p] = *reinterpret_cast<const bf16x8*>(a.wfrag + ((static_cast<size_t>(cb * 2 + ks) * 3 + p) * 4 + mi) * 512 + lane * 8);
    const int p0 = 64 * cb + 16 * mi + 4 * g;  // this lane's four atom pairs
    f32x4 dn[4];
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
      const int r = 16 * ct + l15;
      dn[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (p0 + 3 < a.ld) dn[ct] = *reinterpret_cast<const f32x4*>(a.din + (lrow0 + r) * a.ld + p0);
    }
    float xa[4][3];
    int a2s[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int p = p0 + e < AA2 ? p0 + e : 0;
      const int a1 = p / a.A;
      a2s[e] = p - a1 * a.A;
      xa[e][0] = xa[e][1] = xa[e][2] = 0.0f;
      if (!a.distmat) {
        const float* pa = a.xyz + (ri * a.A + a1) * 3;
        xa[e][0] = pa[0]; xa[e][1] = pa[1]; xa[e][2] = pa[2];
      }
    }
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
      const int r = 16 * ct + l15;
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        bf16x8 fy[3];
#pragma unroll
        for (int p = 0; p < 3; ++p) fy[p] = *reinterpret_cast<const bf16x8*>(Yp + p * PL + pc_off(r, g + 4 * ks));
#pragma unroll
        for (int term = 0; term < 6; ++term) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ks][TA[term]], fy[TB[term]], acc, 0, 0, 0);
      }
      bf16x4 h, m, l;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float E = 0.0f;
        const float v = dn[ct][e];
        if (p0 + e < AA2 && v != 0.0f) {  // v == 0: masked atom pair (or underflow), no gradient
          float d;
          if (a.distmat) {
            d = a.distmat[(grow0 + r) * AA2 + p0 + e];
          } else {
            const float* pb = a.xyz + ((b * a.K + j0 + r) * a.A + a2s[e]) * 3;
            const float dx = xa[e][0] - pb[0], dy = xa[e][1] - pb[1], dz = xa[e][2] - pb[2];
            d = sqrtf((dx * dx + dy * dy) + dz * dz);
          }
          E = acc[e] * (-(d * d) * v);
        }
        __bf16 hh, mm, ll;
        split3(E, hh, mm, ll);
        h[e] = hh; m[e] = mm; l[e] = ll;
      }
      __bf16* dst = Ep + pc_off(r, 2 * mi + (g >> 1)) + 4 * (g & 1);
      *reinterpret_cast<bf16x4*>(dst) = h;
      *reinterpret_cast<bf16x4*>(dst + PL) = m;
      *reinterpret_cast<bf16x4*>(dst + 2 * PL) = l;
    }
    PC_FENCE();
    __syncthreads();
    f32x4 acc1[2];
    acc1[0] = acc1[1] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const int rb = 32 * kk + 8 * g;
      bf16x8 fa[2];
#pragma unroll
      for (int h4 = 0; h4 < 2; ++h4) {
        const auto cv = *reinterpret_cast<const int __attribute__((ext_vector_type(4)))*>(sjs + rb + 4 * h4);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          fa[0][4 * h4 + c] = static_cast<__bf16>(cv[c] == l15 ? 1.0f : 0.0f);
          fa[1][4 * h4 + c] = static_cast<__bf16>(cv[c] == 16 + l15 ? 1.0f : 0.0f);
        }
      }
#pragma unroll
      for (int p = 2; p >= 0; --p) {
        const bf16x8 fb = frag_tr(Ep + p * PL, kk, nt);
        acc1[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[0], fb, acc1[0], 0, 0, 0);
        acc1[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[1], fb, acc1[1], 0, 0, 0);
      }
    }
    const int pcol = 64 * cb + 16 * nt + l15;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int sj = 16 * mt + 4 * g + e;
        if (sj < a.n_aa && pcol < AA2 && acc1[mt][e] != 0.0f) atomicAdd(a.g_sp + ((si * a.n_aa + sj) * AA2 + pcol), acc1[mt][e]);
      }
    PC_FENCE();
    __syncthreads();  // every wave has read the E planes
  }
}

// out_seg[row][col] += sum over the slabs of part[slab][off_seg + row cols_seg + col]: thread = element (slabs read coalesced),
// blockIdx.y = a group of 16 slabs whose loads are all in flight at once; one atomic per (element, group)
constexpr int PR_GROUP = 16;
__global__ void parts_reduce_kernel(const float* __restrict__ parts, int nparts, int64_t stride, PartsSegs sg) {
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  int q = 0;
  while (q < sg.nseg && gid >= sg.off[q] + sg.n[q]) ++q;
  if (q >= sg.nseg || gid < sg.off[q] || sg.out[q] == nullptr) return;
  const int p0 = blockIdx.y * PR_GROUP;
  float x[PR_GROUP];
#pragma unroll
  for (int i = 0; i < PR_GROUP; ++i) x[i] = p0 + i < nparts ? parts[(p0 + i) * stride + gid] : 0.0f;
  float v = 0.0f;
#pragma unroll
  for (int i = 0; i < PR_GROUP; ++i) v += x[i];
  const int idx = gid - sg.off[q];
  atomicAdd(sg.out[q] + static_cast<int64_t>(idx / sg.cols[q]) * sg.ld[q] + idx % sg.cols[q], v);
}
}  // namespace

int launch_parts_reduce(const float* parts, int nparts, int64_t stride, const PartsSegs& sg, hipStream_t st) {
  int total = 0;
  for (int q = 0; q < sg.nseg; ++q) {
    DIFFAB_REQUIRE(sg.off[q] >= total && sg.n[q] >= 1 && sg.cols[q] >= 1, DIFFAB_ERR_ARG, "parts_reduce: segments must ascend");
    total = sg.off[q] + sg.n[q];
  }
  hipLaunchKernelGGL(parts_reduce_kernel, dim3((total + 255) / 256, (nparts + PR_GROUP - 1) / PR_GROUP), dim3(256), 0, st, parts, nparts, stride,
                     sg);
  DIFFAB_LAUNCH_CHECK();
  return DIFFAB_OK;
}

// G1[n_aa^2][64] | G2[2 max_dist + 1][64] (adjacent) += the one-hot products of g (pair_table_mfma_kernel); part: 256 slabs of the same size
bool pair_table_mfma_supported(int C, int K, int64_t nrows, int n_aa, int max_dist) {
  return C == PC_C && K % PC_ROWS == 0 && nrows % PC_ROWS == 0 && nrows >= PC_ROWS && n_aa * n_aa <= 28 * 16 && 2 * max_dist + 1 <= 5 * 16;
}
int launch_pair_table_mfma(const float* g, const int64_t* seq, const uint8_t* seq_m, const int64_t* resid, int resid_bstride, const int64_t* chain,
                           int K, int max_dist, int n_aa, int unk, int64_t row0, int64_t nrows, float* G1, float* part, hipStream_t st) {
  DIFFAB_REQUIRE(g && seq && resid && chain && G1 && part && pair_table_mfma_supported(PC_C, K, nrows, n_aa, max_dist) &&
                     (reinterpret_cast<uintptr_t>(g) & 15) == 0,
                 DIFFAB_ERR_ARG, "pair_table_mfma: unsupported operands");
  PairTableArgs a{};
  a.g = g; a.seq = seq; a.seq_m = seq_m; a.resid = resid; a.chain = chain; a.part = part;
  a.row0 = row0; a.nrows = nrows; a.K = K; a.resid_bstride = resid_bstride; a.max_dist = max_dist; a.n_aa = n_aa; a.unk = unk;
  int dev = 0, ncu = 256;
  if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
  const int64_t ntiles = nrows / PC_ROWS;
  int grid = static_cast<int>(ntiles < ncu ? ntiles : ncu);
  grid = grid > 256 ? 256 : grid;
  const int lds = 3 * PC_PLANE * 2 + 3 * PC_ROWS * 4;
  hipLaunchKernelGGL(pair_table_mfma_kernel, dim3(grid), dim3(512), lds, st, a);
  DIFFAB_LAUNCH_CHECK();
  const int n_tab = (n_aa * n_aa + 2 * max_dist + 1) * PC_C;
  PartsSegs sg{};
  sg.nseg = 1; sg.off[0] = 0; sg.n[0] = n_tab; sg.cols[0] = n_tab; sg.ld[0] = n_tab; sg.out[0] = G1;
  return launch_parts_reduce(part, grid, n_tab, sg, st);
}

// pair_dist_bwd_group_kernel's sums for whole (patch, i) groups of K = 128 rows (pair_dist_bwd_mfma_kernel); ld = leading dimension of din / ddin
bool pair_dist_bwd_mfma_supported(int K, int A, int64_t row0, int64_t nrows, int ld, int n_aa) {
  return K % PC_ROWS == 0 && nrows % K == 0 && row0 % K == 0 && ld % 4 == 0 && ld >= A * A && n_aa <= 32;
}
int launch_pair_dist_bwd_mfma(const int64_t* seq, const uint8_t* seq_m, const float* distmat, const float* xyz, const float* din, const float* ddin,
                              int K, int A, int n_aa, int unk, int64_t row0, int64_t nrows, int ld, float* g_sp, hipStream_t st) {
  DIFFAB_REQUIRE(seq && din && ddin && g_sp && (distmat || xyz) && pair_dist_bwd_mfma_supported(K, A, row0, nrows, ld, n_aa) &&
                     (reinterpret_cast<uintptr_t>(din) & 15) == 0 && (reinterpret_cast<uintptr_t>(ddin) & 15) == 0,
                 DIFFAB_ERR_ARG, "pair_dist_bwd_mfma: unsupported operands");
  PairDistBwdArgs a{};
  a.seq = seq; a.seq_m = seq_m; a.distmat = distmat; a.xyz = xyz; a.din = din; a.ddin = ddin; a.g_sp = g_sp;
  a.row0 = row0; a.K = K; a.A = A; a.ld = ld; a.n_aa = n_aa; a.unk = unk;
  const int lds = 3 * PC_PLANE * 2 + PC_ROWS * 4;
  hipLaunchKernelGGL(pair_dist_bwd_mfma_kernel, dim3(static_cast<unsigned>(nrows / PC_ROWS)), dim3(512), lds, st, a);
  DIFFAB_LAUNCH_CHECK();
  return DIFFAB_OK;
}

// ... with d din = d h1 W formed in the kernel (W: distance_embedding[0].weight padded to ld_w columns, [64][ld_w]); prep as for the chain
int launch_pair_dist_bwd_fused(const int64_t* seq, const uint8_t* seq_m, const float* distmat, const float* xyz, const float* din, const float* dh1,
                               const float* W, int ld_w, int K, int A, int n_aa, int unk, int64_t row0, int64_t nrows, int ld, float* g_sp,
                               float* prep, hipStream_t st) {
  DIFFAB_REQUIRE(seq && din && dh1 && W && g_sp && prep && (distmat || xyz) && pair_dist_bwd_mfma_supported(K, A, row0, nrows, ld, n_aa) &&
                     A * A <= 256 && ld_w >= A * A && (reinterpret_cast<uintptr_t>(din) & 15) == 0 && (reinterpret_cast<uintptr_t>(dh1) & 15) == 0 &&
                     (reinterpret_cast<uintptr_t>(prep) & 15) == 0,
                 DIFFAB_ERR_ARG, "pair_dist_bwd_fused: unsupported operands");
  auto nc = [&](int cb) { const int n = ld_w - 64 * cb; return n < 0 ? 0 : (n > 64 ? 64 : n); };
  auto wp = [&](int cb) { return nc(cb) > 0 ? W + 64 * cb : W; };
  hipLaunchKernelGGL(pair_chain_prep_kernel, dim3((4 * 2 * 4 * 64 * 8 + 255) / 256), dim3(256), 0, st, wp(0), ld_w, wp(1), ld_w, wp(2), ld_w, wp(3),
                     ld_w, nc(0), nc(1), nc(2), nc(3), reinterpret_cast<__bf16*>(prep));
  DIFFAB_LAUNCH_CHECK();
  PairDistFusedArgs a{};
  a.seq = seq; a.seq_m = seq_m; a.distmat = distmat; a.xyz = xyz; a.din = din; a.dh1 = dh1; a.wfrag = reinterpret_cast<const __bf16*>(prep);
  a.g_sp = g_sp; a.row0 = row0; a.K = K; a.A = A; a.ld = ld; a.n_aa = n_aa; a.unk = unk;
  const int lds = 6 * (64 * PC_C) * 2 + 64 * 4;  // 64-row tiles: 49 KiB, three 4-wave work-groups per CU
  hipLaunchKernelGGL(pair_dist_bwd_fused64_kernel, dim3(static_cast<unsigned>(nrows / 64)), dim3(256), lds, st, a);
  DIFFAB_LAUNCH_CHECK();
  return DIFFAB_OK;
}

size_t pair_chain_bwd_prep_floats() { return PC_WFRAG / 2 + 64; }
size_t pair_chain_bwd_part_floats() { return static_cast<size_t>(256) * PC_PART; }
bool pair_chain_bwd_supported(int C, int K, int64_t nrows) { return C == PC_C && K % PC_ROWS == 0 && nrows % PC_ROWS == 0 && nrows >= PC_ROWS; }

// W: {mlp[4].W, mlp[2].W, mlp[0].W[:, 2C:3C], distance_embedding[2].W} with their leading dimensions; X: {m2, m1, df, h1}; gW / ldg / gb:
// the matching gradients (+=).  prep: pair_chain_bwd_prep_floats() floats, 16-byte aligned, overwritten; part: pair_chain_bwd_part_floats().
int launch_pair_chain_bwd(const float* d_out, const float* amask, int K, int A, int ca, int64_t row0, int64_t nrows, const float* const* X,
                          const float* const* W, const int* ldw, float* dC, float* dh1, float* const* gW, const int* ldg, float* const* gb,
                          float* prep, float* part, hipStream_t st) {
  DIFFAB_REQUIRE(d_out && amask && X && W && dC && dh1 && gW && gb && prep && part && (reinterpret_cast<uintptr_t>(prep) & 15) == 0 &&
                     pair_chain_bwd_supported(PC_C, K, nrows) && (reinterpret_cast<uintptr_t>(d_out) & 15) == 0 &&
                     (reinterpret_cast<uintptr_t>(dC) & 15) == 0 && (reinterpret_cast<uintptr_t>(dh1) & 15) == 0,
                 DIFFAB_ERR_ARG, "pair_chain_bwd: unsupported operands");
  PairChainArgs a{};
  a.d_out = d_out;
  a.amask = amask;
  for (int i = 0; i < 4; ++i) {
    DIFFAB_REQUIRE(X[i] && W[i] && gW[i] && (reinterpret_cast<uintptr_t>(X[i]) & 15) == 0, DIFFAB_ERR_ARG, "pair_chain_bwd: bad operand %d", i);
    a.X[i] = X[i];
  }
  a.part = part;
  a.wfrag = reinterpret_cast<const __bf16*>(prep);
  a.dC = dC;
  a.dh1 = dh1;
  a.row0 = row0;
  a.nrows = nrows;
  a.K = K;
  a.A = A;
  a.ca = ca;
  hipLaunchKernelGGL(pair_chain_prep_kernel, dim3((4 * 2 * 4 * 64 * 8 + 255) / 256), dim3(256), 0, st, W[0], ldw[0], W[1], ldw[1], W[2], ldw[2],
                     W[3], ldw[3], 64, 64, 64, 64, reinterpret_cast<__bf16*>(prep));
  DIFFAB_LAUNCH_CHECK();
  int dev = 0, ncu = 256;
  if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
  const int64_t ntiles = nrows / PC_ROWS;
  int grid = static_cast<int>(ntiles < ncu ? ntiles : ncu);  // one work-group per CU (160 KiB of LDS), each walks its tiles
  grid = grid > 256 ? 256 : grid;                              // (the partial-sum slabs are sized for 256)
  DIFFAB_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(pair_chain_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       PC_LDS_BYTES));
  hipLaunchKernelGGL(pair_chain_bwd_kernel, dim3(grid), dim3(512), PC_LDS_BYTES, st, a);
  DIFFAB_LAUNCH_CHECK();
  PartsSegs sg{};
  sg.nseg = 8;
  for (int i = 0; i < 4; ++i) {
    sg.off[i] = i * 64 * 64; sg.n[i] = 64 * 64; sg.cols[i] = 64; sg.ld[i] = ldg[i]; sg.out[i] = gW[i];
    sg.off[4 + i] = 4 * 64 * 64 + 64 * i; sg.n[4 + i] = 64; sg.cols[4 + i] = 64; sg.ld[4 + i] = 64; sg.out[4 + i] = gb[i];
  }
  return launch_parts_reduce(part, grid, PC_PART, sg, st);
}

}  // namespace diffab
