// attention_split.hip - the IPA attention of the benchmark geometry as three kernels, each tiled for what bounds it
// (single key chunk: K = 64 or 128; reference InvariantPointAttentionLayer.forward, diffab_pytorch.py:389-465).
//
// The fused kernel in denoiser_fast.hip runs its three phases back to back inside one work-group, so the pair stream is in
// flight for only half of a work-group's life and the key-side operands are re-staged by each of the 8 row tiles of a patch.
// Here the phases are separate launches that exchange the (B, 8, K, K) logits / probabilities through HBM (+25 % bytes on top
// of the pair stream, which is not the bound):
//   A  ipa_logits_kernel       work-group = (patch, head, 128 or 64 query rows): k_s / k_pts of the head staged ONCE in LDS for the
//                              whole work-group, scalar logits on the MFMA, point logits as direct differences (packed fp32)
//                              -> S[b][h][i][j] = 3^-1/2 (q.k / sqrt(ds) + coef_h |q_pts - k_pts|^2)
//   B  ipa_pair_stream_kernel  wave = query rows, no barrier, no logits image in LDS: streams the pair row e[i] once
//                              (non-temporal, 128 VGPRs), adds the pair bias (MFMA), softmax in registers, o_e on the MFMA,
//                              writes the NORMALISED probabilities back in place of the logits and o_e to the feature row;
//                              the next row's tiles are requested while the current row's are retired
//   C  ipa_pv_kernel           work-group = (patch, head, 128 or 64 query rows): v_s / v_pts staged once in LDS, o_s and o_pts on
//                              the MFMA with P as the A operand, global->local frames and norms -> feature row.
// Users: the training tape (the forward of a training step runs A, B, C and keeps P
// and the squared point distances for the backward); the attention backward (B' = ipa_pair_stream_bwd_kernel below, and the
// probability recompute when a tape has no slot for them).
#include <type_traits>

#include "common.h"
#include "denoiser_internal.h"

namespace diffab {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define MEM_FENCE() asm volatile("" ::: "memory")
#define FAST_EXP(x) __expf(x)

namespace {
constexpr int AH = 8, ADS = 32, AP = 8, AC = 64;
constexpr int ANP = 3 * AH * ADS + 3 * AH * AP * 3;             // 1344 projection columns
constexpr int AF = AH * ADS + AH * AC + AH * AP * 3 + AH * AP;  // 1024 feature columns
constexpr int OFF_QS = 0, OFF_KS = 256, OFF_VS = 512, OFF_GQ = 768, OFF_GK = 960, OFF_GV = 1152;
constexpr int FOFF_OS = 0, FOFF_OE = 256, FOFF_OL = 768, FOFF_ON = 960;
// query rows per work-group in kernels A and C: 4 waves x RB / 64 16-row MFMA tiles; 128 when K allows, else 64
inline int rows_per_wg(int K) { return K % 128 == 0 ? 128 : 64; }
}  // namespace

// ================================================================== A: logits
constexpr int KLD = 40, GLD = 28;  // LDS row strides (floats) of the staged k_s / k_pts rows: ds_read_b128 conflict-free

template <bool WD2>  // WD2: also write the squared point distances (the training backward needs them for d gamma)
__global__ __launch_bounds__(256) void ipa_logits_kernel(const float* __restrict__ proj, const float* __restrict__ gamma,
                                                         float* __restrict__ SP, float* __restrict__ D2, int K, int RB) {
  extern __shared__ __attribute__((aligned(16))) float lds[];  // [K][KLD] k_s rows, then [K][GLD] k_pts rows of this head
  float* ks_l = lds;
  float* gk_l = lds + K * KLD;
  const int nrb = K / RB;
  const int rb = blockIdx.x % nrb, h = (blockIdx.x / nrb) % AH, b = blockIdx.x / (nrb * AH);
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, l15 = lane & 15, q = lane >> 4;
  const int64_t prow0 = static_cast<int64_t>(b) * K;
  // stage the key side of head h: full 128-byte lines (8 lanes per k_s row, 6 per k_pts row)
  for (int idx = tid; idx < K * 8; idx += 256) {
    const int j = idx >> 3, c4 = idx & 7;
    *reinterpret_cast<f32x4*>(ks_l + j * KLD + 4 * c4) = *reinterpret_cast<const f32x4*>(proj + (prow0 + j) * ANP + OFF_KS + h * ADS + 4 * c4);
  }
  for (int idx = tid; idx < K * 6; idx += 256) {
    const int j = idx / 6, c4 = idx % 6;
    *reinterpret_cast<f32x4*>(gk_l + j * GLD + 4 * c4) = *reinterpret_cast<const f32x4*>(proj + (prow0 + j) * ANP + OFF_GK + h * 24 + 4 * c4);
  }
  // Waves are launched at a limited rate (about one per 90 cycles per XCD, measured: with one 16-row tile per wave this kernel
  // was bound by that, CUs idle a third of the time), so a wave walks RT row tiles of the same (patch, head) instead of one.
  const int RT = RB / 64;  // row tiles per wave
  const float scale_t = 0.57735026918962576f;                     // 3^-1/2   (diffab_pytorch.py:387, :439)
  const float scale_s = 0.17677669529663687f;                     // 32^-1/2  (:353)
  const float coef_p = -0.5f * 0.16666666666666666f * gamma[h];   // -1/2 (4.5*8)^-1/2 gamma_h  (:372, :431-436)
  __syncthreads();
  for (int rt = 0; rt < RT; ++rt) {
  const int i0 = rb * RB + 16 * (wv + 4 * rt);  // this pass's 16 query rows
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
  // Results leave through a per-wave LDS tile [16 rows][32 keys] (two key tiles): the MFMA layout gives a lane one key of four rows,
  // i.e. 4-byte stores in 64-byte runs; re-read row-major, a lane stores 16 bytes and 8 lanes cover a full 128-byte line of a row.
  constexpr int OLD = 36;  // tile row stride (floats)
  float* otile = gk_l + K * GLD + wv * (16 * OLD);
  float* dtile = gk_l + K * GLD + 4 * (16 * OLD) + wv * (16 * OLD);  // second tile: squared distances (WD2 only)
  float* sbase = SP + ((static_cast<int64_t>(b) * AH + h) * K + i0) * K;
  float* dbase = WD2 ? D2 + ((static_cast<int64_t>(b) * AH + h) * K + i0) * K : nullptr;
  const int ntile = K / 16;
  for (int jt = 0; jt < ntile; ++jt) {
    const float* kt = ks_l + (jt * 16 + l15) * KLD + 4 * q;
    const f32x4 kb0 = *reinterpret_cast<const f32x4*>(kt);  // k_s[16 jt + l15][16 sg + 4 q + s]
    const f32x4 kb1 = *reinterpret_cast<const f32x4*>(kt + 16);
    f32x4 gk[6];
#pragma unroll
    for (int cc = 0; cc < 6; ++cc) gk[cc] = *reinterpret_cast<const f32x4*>(gk_l + (jt * 16 + l15) * GLD + 4 * cc);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(qa[0][s], kb0[s], acc, 0, 0, 0);
#pragma unroll
    for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(qa[1][s], kb1[s], acc, 0, 0, 0);
    // acc[r] = q_s[i0 + 4q + r] . k_s[key 16 jt + l15]
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      f32x2 d2v = {0.f, 0.f};  // packed fp32: two coordinates per instruction, two partial sums added at the end
#pragma unroll
      for (int cc = 0; cc < 6; ++cc) {
        f32x2 dlo, dhi;  // packed subtract spelled in assembly: the compiler splits a vector fsub into two v_sub_f32
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
      otile[(4 * q + r) * OLD + (jt & 1) * 16 + l15] = scale_t * (acc[r] * scale_s + coef_p * d2);
      if (WD2) dtile[(4 * q + r) * OLD + (jt & 1) * 16 + l15] = d2;
    }
    if (jt & 1) {  // two key tiles complete: rows (lane >> 3) and 8 + (lane >> 3), keys 4 (lane & 7) .. + 3 of the 32
      const int orow = lane >> 3, oc = 4 * (lane & 7);
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(otile + (orow + 8 * half) * OLD + oc);
        *reinterpret_cast<f32x4*>(sbase + static_cast<int64_t>(orow + 8 * half) * K + (jt - 1) * 16 + oc) = v;
        if (WD2)
          *reinterpret_cast<f32x4*>(dbase + static_cast<int64_t>(orow + 8 * half) * K + (jt - 1) * 16 + oc) =
              *reinterpret_cast<const f32x4*>(dtile + (orow + 8 * half) * OLD + oc);
      }
    }
  }
  }
}

// ================================================================== C: probabilities x values
constexpr int VLD = 36, PLD = 28;  // LDS row strides of the staged v_s (32) / v_pts (24) rows

__global__ __launch_bounds__(256) void ipa_pv_kernel(const float* __restrict__ proj, const float* __restrict__ SP,
                                                     const float* __restrict__ R, const float* __restrict__ t, float* __restrict__ feat,
                                                     int K, int RB) {
  extern __shared__ __attribute__((aligned(16))) float lds[];  // [K][VLD] v_s rows, then [K][PLD] v_pts rows of this head
  float* vs_l = lds;
  float* gv_l = lds + K * VLD;
  const int nrb = K / RB;
  const int rb = blockIdx.x % nrb, h = (blockIdx.x / nrb) % AH, b = blockIdx.x / (nrb * AH);
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, l15 = lane & 15, q = lane >> 4;
  const int64_t prow0 = static_cast<int64_t>(b) * K;
  for (int idx = tid; idx < K * 8; idx += 256) {
    const int j = idx >> 3, c4 = idx & 7;
    *reinterpret_cast<f32x4*>(vs_l + j * VLD + 4 * c4) = *reinterpret_cast<const f32x4*>(proj + (prow0 + j) * ANP + OFF_VS + h * ADS + 4 * c4);
  }
  for (int idx = tid; idx < K * 6; idx += 256) {
    const int j = idx / 6, c4 = idx % 6;
    *reinterpret_cast<f32x4*>(gv_l + j * PLD + 4 * c4) = *reinterpret_cast<const f32x4*>(proj + (prow0 + j) * ANP + OFF_GV + h * 24 + 4 * c4);
  }
  const int RT = RB / 64;  // row tiles per wave (see ipa_logits_kernel)
  const int ntile = K / 16;
  // A operand: P[i = i0 + l15][j = 16 jt + 4 q + r]; all tiles of the row requested up front (K / 16 <= 8 float4 per lane), the
  // next row tile's while the current one is multiplied
  auto prow_of = [&](int rt) {
    return SP + ((static_cast<int64_t>(b) * AH + h) * K + rb * RB + 16 * (wv + 4 * rt) + l15) * K + 4 * q;
  };
  f32x4 pa[8], pn[8];
#pragma unroll
  for (int jt = 0; jt < 8; ++jt)
    if (jt < ntile) pa[jt] = *reinterpret_cast<const f32x4*>(prow_of(0) + jt * 16);
  __syncthreads();
  for (int rt = 0; rt < RT; ++rt) {
  const int i0 = rb * RB + 16 * (wv + 4 * rt);
  if (rt + 1 < RT) {
#pragma unroll
    for (int jt = 0; jt < 8; ++jt)
      if (jt < ntile) pn[jt] = *reinterpret_cast<const f32x4*>(prow_of(rt + 1) + jt * 16);
  }
  f32x4 os[2], og[3];
#pragma unroll
  for (int d = 0; d < 2; ++d) os[d] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int cc = 0; cc < 3; ++cc) og[cc] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int pp = l15 & 7;
#pragma unroll
  for (int jt = 0; jt < 8; ++jt) {
    if (jt < ntile) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = jt * 16 + 4 * q + r;  // B operand row k = q (keys 4 q + r of the step), column n = l15
        const f32x2 vv = *reinterpret_cast<const f32x2*>(vs_l + key * VLD + 2 * l15);  // d = 2 l15 + dt
        const float* g = gv_l + key * PLD + 3 * pp;                                      // point pp, coords 0..2
        const float gx = g[0], gy = g[1], gz = g[2];
        os[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(pa[jt][r], vv[0], os[0], 0, 0, 0);
        os[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(pa[jt][r], vv[1], os[1], 0, 0, 0);
        og[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(pa[jt][r], gx, og[0], 0, 0, 0);
        og[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(pa[jt][r], gy, og[1], 0, 0, 0);
        og[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(pa[jt][r], gz, og[2], 0, 0, 0);
      }
    }
  }
  // D rows i = 4 q + r, column n = l15 (probabilities arrive normalised: no 1/L here)
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int64_t row = prow0 + i0 + 4 * q + r;
    float* fr = feat + row * AF;
    *reinterpret_cast<float2*>(fr + FOFF_OS + h * ADS + 2 * l15) = make_float2(os[0][r], os[1][r]);
    if (l15 < 8) {
      const float* Rr = R + row * 9;
      const float* tr = t + row * 3;
      const float dx = og[0][r] - tr[0], dy = og[1][r] - tr[1], dz = og[2][r] - tr[2];
      const float lx = dx * Rr[0] + dy * Rr[1] + dz * Rr[2];  // (p - t) R^T   (diffab_pytorch.py:336)
      const float ly = dx * Rr[3] + dy * Rr[4] + dz * Rr[5];
      const float lz = dx * Rr[6] + dy * Rr[7] + dz * Rr[8];
      float* fo = fr + FOFF_OL + h * 24 + 3 * l15;
      fo[0] = lx; fo[1] = ly; fo[2] = lz;
      fr[FOFF_ON + h * AP + l15] = sqrtf(lx * lx + ly * ly + lz * lz);
    }
  }
#pragma unroll
  for (int jt = 0; jt < 8; ++jt) pa[jt] = pn[jt];
  }
}

// ================================================================== B: pair stream
// NT key tiles of 16 (K = 16 NT).  A work-group is 8 independent waves; wave w owns RPW consecutive query rows of one patch.
constexpr int ELD = 72;  // pair-tile stride (floats) of the per-wave re-orientation scratch

// OE = false (training backward: only the probabilities are wanted): no o_e product, no feature-row store
template <int NT, int RPW, bool OE = true>
__global__ __launch_bounds__(512) void ipa_pair_stream_kernel(const float* __restrict__ e, const float* __restrict__ Wb,
                                                              float* __restrict__ SP, float* __restrict__ feat, int rows_total) {
  extern __shared__ __attribute__((aligned(16))) float lds[];  // per wave: 2 tiles [16][ELD]
  constexpr int K = 16 * NT;
  const int tid = threadIdx.x, lane0 = tid & 63, wv = tid >> 6;
  float* scr = lds + wv * (2 * 16 * ELD);
  const int64_t row_first = (static_cast<int64_t>(blockIdx.x) * 8 + wv) * RPW;  // global row (b K + i) of this wave's first row
  if (row_first >= rows_total) return;
  const int l15 = lane0 & 15, q = lane0 >> 4;
  const int h = l15 & 7;  // lanes with l15 >= 8 shadow head l15 - 8 (their MFMA columns are padding)
  const float scale_t = 0.57735026918962576f;
  // bias B fragments: Wb[h][16 sg + 4 q + s] for lane (h = l15 < 8, q), zero in the padding columns
  f32x4 wb[4];
#pragma unroll
  for (int sg = 0; sg < 4; ++sg) {
    wb[sg] = *reinterpret_cast<const f32x4*>(Wb + h * AC + 16 * sg + 4 * q);
#pragma unroll
    for (int s = 0; s < 4; ++s) wb[sg][s] = l15 < 8 ? wb[sg][s] : 0.0f;
  }
  // e[row, :, :] streamed once in the orientation of the o_e product: lane (l15, q) holds e[row][j = 16 jt + 4 q + r][c = 4 l15 .. + 3]
  f32x4 ev[2][NT][4];
  auto load_e_tile = [&](int slot, int64_t row, int jt) {
    const f32x4* ep = reinterpret_cast<const f32x4*>(e + (row * K + jt * 16 + 4 * q) * AC + 4 * l15);
#pragma unroll
    for (int r = 0; r < 4; ++r) ev[slot][jt][r] = __builtin_nontemporal_load(ep + r * (AC / 4));
  };
  // logits of (row, head h, keys 16 jt + 4 q + r): S[b][h][i][j] with row = b K + i
  auto sp_of = [&](int64_t row) { return SP + ((row / K) * AH * K + static_cast<int64_t>(h) * K + row % K) * K + 4 * q; };
  auto stage_e = [&](int slot, int jt) {  // tile re-orientation for the bias product: write [key 4q+r][chunk l15], read [key l15][16 sg + 4 q ..]
    float* t_ = scr + (jt & 1) * (16 * ELD) + 4 * q * ELD + 4 * l15;
#pragma unroll
    for (int r = 0; r < 4; ++r) *reinterpret_cast<f32x4*>(t_ + r * ELD) = ev[slot][jt][r];
  };
#pragma unroll
  for (int jt = 0; jt < NT; ++jt) load_e_tile(0, row_first, jt);
  MEM_FENCE();

  auto do_row = [&](auto slot_c, int64_t row, auto has_next_c) {
    constexpr int slot = decltype(slot_c)::value;
    constexpr bool has_next = decltype(has_next_c)::value;
    float* sp = sp_of(row);
    f32x4 lgv[NT];  // logits, then exp(logit - M), of keys 16 jt + 4 q + r for head h
#pragma unroll
    for (int jt = 0; jt < NT; ++jt) lgv[jt] = *reinterpret_cast<const f32x4*>(sp + jt * 16);
    MEM_FENCE();
    float mx = -INFINITY;
    stage_e(slot, 0);
#pragma unroll
    for (int jt = 0; jt < NT; ++jt) {
      if (jt + 1 < NT) stage_e(slot, jt + 1);
      const float* t_ = scr + (jt & 1) * (16 * ELD) + l15 * ELD + 4 * q;
      f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acc2 = {0.f, 0.f, 0.f, 0.f};  // two chains of 8: half the dependent-MFMA latency
#pragma unroll
      for (int sg = 0; sg < 4; ++sg) {
        const f32x4 ea = *reinterpret_cast<const f32x4*>(t_ + 16 * sg);  // e[row][16 jt + l15][16 sg + 4 q + s]
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          if (sg & 1) acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(ea[s], wb[sg][s], acc2, 0, 0, 0);
          else acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ea[s], wb[sg][s], acc, 0, 0, 0);
        }
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float v = lgv[jt][r] + scale_t * (acc[r] + acc2[r]);
        lgv[jt][r] = v;
        mx = fmaxf(mx, v);
      }
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16));
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    float sum = 0.f;
#pragma unroll
    for (int jt = 0; jt < NT; ++jt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float p = FAST_EXP(lgv[jt][r] - mx);
        lgv[jt][r] = p;
        sum += p;
      }
    sum += __shfl_xor(sum, 16);
    sum += __shfl_xor(sum, 32);
    const float inv = 1.0f / sum;
    // normalised probabilities back in place of the logits, BEFORE any load of the next row is issued: a later wait for those
    // loads then finds these stores long acknowledged (a wait for loads with younger stores in flight waits for the stores too)
    if (l15 < 8) {
#pragma unroll
      for (int jt = 0; jt < NT; ++jt) *reinterpret_cast<f32x4*>(sp + jt * 16) = lgv[jt] * inv;
    }
    if constexpr (!OE) {
      if (has_next) {
#pragma unroll
        for (int jt = 0; jt < NT; ++jt) load_e_tile(1 - slot, row + 1, jt);
        MEM_FENCE();
      }
      return;
    } else {
    // o_e[h][c] = sum_j P[h][j] e[row][j][c]: P (unnormalised) is the B operand straight from registers
    f32x4 oe[4];
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) oe[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int jt = 0; jt < NT; ++jt) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int ct = 0; ct < 4; ++ct)  // A: e[row][j = 16 jt + 4 q + r][c = 4 l15 + ct]
          oe[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(ev[slot][jt][r][ct], lgv[jt][r], oe[ct], 0, 0, 0);
      if (has_next) {  // the retired tile frees its registers' twin in the other slot: request the next row's tile
        load_e_tile(1 - slot, row + 1, jt);
        MEM_FENCE();
      }
    }
    // D: column h = l15, row m = 4 q + r' <-> channel 4 m + ct = 16 q + 4 r' + ct
    if (l15 < 8) {
      float* fo = feat + row * AF + FOFF_OE + h * AC + 16 * q;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        f32x4 v = {oe[0][r], oe[1][r], oe[2][r], oe[3][r]};
        *reinterpret_cast<f32x4*>(fo + 4 * r) = v * inv;
      }
    }
    }
  };
  // RPW is small and even: the row loop is spelled out so that "is there a next row" is a compile-time fact (a run-time flag
  // puts a branch around every prefetch, and the compiler's vmcnt bookkeeping then waits for loads it has just issued)
  static_assert(RPW == 2 || RPW == 4 || RPW == 8, "rows per wave");
  constexpr std::integral_constant<int, 0> s0{};
  constexpr std::integral_constant<int, 1> s1{};
  constexpr std::true_type more{};
  constexpr std::false_type done{};
  do_row(s0, row_first + 0, more);
  if constexpr (RPW == 2) {
    do_row(s1, row_first + 1, done);
  } else {
    do_row(s1, row_first + 1, more);
    do_row(s0, row_first + 2, more);
    if constexpr (RPW == 4) {
      do_row(s1, row_first + 3, done);
    } else {
      do_row(s1, row_first + 3, more);
      do_row(s0, row_first + 4, more);
      do_row(s1, row_first + 5, more);
      do_row(s0, row_first + 6, more);
      do_row(s1, row_first + 7, done);
    }
  }
}

// ================================================================== B': pair stream of the training backward
// Same streaming structure as ipa_pair_stream_kernel (wave = RPW query rows, pair row in registers, no barrier), for the terms of the
// attention backward that need the pair row:
//   dA[h][j]  = dA_kv[h][j] (from ipa_attn_bwd_dakv_mfma_kernel) + do_e[h] . e[i][j]          (the bias-shaped MFMA product)
//   g[h][j]   = 3^-1/2 P (dA - sum_j P dA)                     -> written over dA_kv ([b][h][i][j])
//   d gamma_h partial = sum_j g scale_p d2,   d w_bias[h][c] partial = sum_j g e[i][j][c]       (the o_e-shaped MFMA product)
// Partials: each wave sums its RPW rows in registers, the work-group's eight waves are added through LDS in a fixed order, and ONE row
// wb_part[work-group][8 * 64 + 8] is written per 32 query rows; one column sum over the work-groups finishes them (per-row partials
// were 34 MB written and re-read per layer at B = 128, and 15 us per column sum).
template <int NT, int RPW>
__global__ __launch_bounds__(512) void ipa_pair_stream_bwd_kernel(const float* __restrict__ e, const float* __restrict__ P,
                                                                  float* __restrict__ G /* in: dA_kv, out: g */,
                                                                  const float* __restrict__ D2, const float* __restrict__ dfeat,
                                                                  float* __restrict__ wb_part, int rows_total,
                                                                  const float* __restrict__ Wb, float* __restrict__ de) {
  // de != nullptr:  de[i][j][c] += sum_h (P[h][j] do_e[h][c] + g[h][j] w_bias[h][c])  as an MFMA product over the 16 "heads"
  // (8 x P, 8 x g): D[c][j] with A[c][h'] = [do_e ; w_bias] and B[h'][j] = [P ; g], the B tile re-oriented through the wave's LDS.
  extern __shared__ __attribute__((aligned(16))) float lds[];  // per wave: 2 tiles [16][ELD]
  constexpr int K = 16 * NT;
  const int tid = threadIdx.x, lane0 = tid & 63, wv = tid >> 6;
  float* scr = lds + wv * (2 * 16 * ELD);
  const int64_t row_first = (static_cast<int64_t>(blockIdx.x) * 8 + wv) * RPW;  // rows_total % (8 RPW) == 0 (launcher): every wave has rows
  const int l15 = lane0 & 15, q = lane0 >> 4;
  const int h = l15 & 7;
  const float scale_t = 0.57735026918962576f, scale_p = -0.5f * 0.16666666666666666f;
  f32x4 ev[2][NT][4];
  auto load_e_tile = [&](int slot, int64_t row, int jt) {
    const f32x4* ep = reinterpret_cast<const f32x4*>(e + (row * K + jt * 16 + 4 * q) * AC + 4 * l15);
#pragma unroll
    for (int r = 0; r < 4; ++r) ev[slot][jt][r] = __builtin_nontemporal_load(ep + r * (AC / 4));
  };
  auto off_of = [&](int64_t row) { return ((row / K) * AH * K + static_cast<int64_t>(h) * K + row % K) * K + 4 * q; };
  auto stage_e = [&](int slot, int jt) {
    float* t_ = scr + (jt & 1) * (16 * ELD) + 4 * q * ELD + 4 * l15;
#pragma unroll
    for (int r = 0; r < 4; ++r) *reinterpret_cast<f32x4*>(t_ + r * ELD) = ev[slot][jt][r];
  };
  // A fragments of the d e product: lane (l15, q) holds M[h' = 4 q + s][c = 16 ct + l15]; h' < 8: do_e of the row (reloaded per row
  // by the lanes q < 2), h' >= 8: w_bias (lanes q >= 2, loaded once)
  float af[4][4];
  if (de != nullptr && q >= 2) {
#pragma unroll
    for (int ct = 0; ct < 4; ++ct)
#pragma unroll
      for (int s = 0; s < 4; ++s) af[ct][s] = Wb[(4 * (q - 2) + s) * AC + 16 * ct + l15];
  }
#pragma unroll
  for (int jt = 0; jt < NT; ++jt) load_e_tile(0, row_first, jt);
  MEM_FENCE();

  f32x4 oe_sum[4];   // d w_bias partial of this wave's rows: D column h = l15, row 4 q + r' <-> channel 16 q + 4 r' + ct
  float dg_sum = 0.f;  // d gamma partial (head h)
#pragma unroll
  for (int ct = 0; ct < 4; ++ct) oe_sum[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
  auto do_row = [&](auto slot_c, int64_t row, auto has_next_c) {
    constexpr int slot = decltype(slot_c)::value;
    constexpr bool has_next = decltype(has_next_c)::value;
    const int64_t off = off_of(row);
    f32x4 dv[NT];  // dA, then g, of keys 16 jt + 4 q + r for head h (P is read per tile, twice, from L2: 32 VGPRs short otherwise)
    f32x4 wb[4];   // B fragments of the dA_e product: do_e[h][16 sg + 4 q + s] of THIS row, zero in the padding columns
#pragma unroll
    for (int sg = 0; sg < 4; ++sg) {
      wb[sg] = *reinterpret_cast<const f32x4*>(dfeat + row * AF + FOFF_OE + h * AC + 16 * sg + 4 * q);
#pragma unroll
      for (int s = 0; s < 4; ++s) wb[sg][s] = l15 < 8 ? wb[sg][s] : 0.0f;
    }
    if (de != nullptr && q < 2) {
#pragma unroll
      for (int ct = 0; ct < 4; ++ct)
#pragma unroll
        for (int s = 0; s < 4; ++s) af[ct][s] = dfeat[row * AF + FOFF_OE + (4 * q + s) * AC + 16 * ct + l15];
    }
    MEM_FENCE();
    float red = 0.f;
    stage_e(slot, 0);
#pragma unroll
    for (int jt = 0; jt < NT; ++jt) {
      if (jt + 1 < NT) stage_e(slot, jt + 1);
      const f32x4 pj = *reinterpret_cast<const f32x4*>(P + off + jt * 16);
      const f32x4 dj = *reinterpret_cast<const f32x4*>(G + off + jt * 16);
      MEM_FENCE();
      const float* t_ = scr + (jt & 1) * (16 * ELD) + l15 * ELD + 4 * q;
      f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acc2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int sg = 0; sg < 4; ++sg) {
        const f32x4 ea = *reinterpret_cast<const f32x4*>(t_ + 16 * sg);  // e[row][16 jt + l15][16 sg + 4 q + s]
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          if (sg & 1) acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(ea[s], wb[sg][s], acc2, 0, 0, 0);
          else acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ea[s], wb[sg][s], acc, 0, 0, 0);
        }
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float v = dj[r] + (acc[r] + acc2[r]);
        dv[jt][r] = v;
        red += pj[r] * v;
      }
    }
    red += __shfl_xor(red, 16);
    red += __shfl_xor(red, 32);
#pragma unroll
    for (int jt = 0; jt < NT; ++jt) {  // g out, before any load of the next row is issued (see ipa_pair_stream_kernel)
      const f32x4 pj = *reinterpret_cast<const f32x4*>(P + off + jt * 16);
#pragma unroll
      for (int r = 0; r < 4; ++r) dv[jt][r] = scale_t * pj[r] * (dv[jt][r] - red);
      if (l15 < 8) *reinterpret_cast<f32x4*>(G + off + jt * 16) = dv[jt];
    }
    // d gamma partial
    float dg = 0.f;
#pragma unroll
    for (int jt = 0; jt < NT; ++jt) {
      const f32x4 d2 = *reinterpret_cast<const f32x4*>(D2 + off + jt * 16);
#pragma unroll
      for (int r = 0; r < 4; ++r) dg += dv[jt][r] * d2[r];
    }
    dg += __shfl_xor(dg, 16);
    dg += __shfl_xor(dg, 32);
    dg_sum += dg * scale_p;
    // d w_bias partial: sum_j g[h][j] e[row][j][c], g as the B operand straight from registers
    f32x4 oe[4];
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) oe[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
    float* pg = scr;  // [16 h'][PGLD] tile of [P ; g] for 16 keys (the pair-tile scratch is idle after the dA_e product)
    constexpr int PGLD = 20;
    float* derow = de != nullptr ? de + (row * K + l15) * AC + 4 * q : nullptr;  // + 16 jt keys, + 16 ct channels
#pragma unroll
    for (int jt = 0; jt < NT; ++jt) {
      f32x4 dold[4];
      if (de != nullptr) {
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) dold[ct] = *reinterpret_cast<const f32x4*>(derow + static_cast<int64_t>(jt) * 16 * AC + 16 * ct);
        if (l15 < 8) {
          *reinterpret_cast<f32x4*>(pg + h * PGLD + 4 * q) = *reinterpret_cast<const f32x4*>(P + off + jt * 16);
          *reinterpret_cast<f32x4*>(pg + (8 + h) * PGLD + 4 * q) = dv[jt];
        }
      }
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) oe[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(ev[slot][jt][r][ct], dv[jt][r], oe[ct], 0, 0, 0);
      if (de != nullptr) {
        float bf[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) bf[s] = pg[(4 * q + s) * PGLD + l15];  // B[h' = 4 q + s][key 16 jt + l15]
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) {
          f32x4 dacc = dold[ct];  // D[c = 16 ct + 4 q' + r'][key l15]: accumulate straight onto the old values
#pragma unroll
          for (int s = 0; s < 4; ++s) dacc = __builtin_amdgcn_mfma_f32_16x16x4f32(af[ct][s], bf[s], dacc, 0, 0, 0);
          *reinterpret_cast<f32x4*>(derow + static_cast<int64_t>(jt) * 16 * AC + 16 * ct) = dacc;
        }
      }
      if (has_next) {
        load_e_tile(1 - slot, row + 1, jt);
        MEM_FENCE();
      }
    }
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) oe_sum[ct] += oe[ct];
  };
  static_assert(RPW == 4, "rows per wave");
  constexpr std::integral_constant<int, 0> s0{};
  constexpr std::integral_constant<int, 1> s1{};
  constexpr std::true_type more{};
  constexpr std::false_type done{};
  do_row(s0, row_first + 0, more);
  do_row(s1, row_first + 1, more);
  do_row(s0, row_first + 2, more);
  do_row(s1, row_first + 3, done);
  // the wave's sums -> its own (now idle) scratch, then the work-group's eight waves in a fixed order -> one row of partials
  if (l15 < 8) {
#pragma unroll
    for (int r = 0; r < 4; ++r) *reinterpret_cast<f32x4*>(scr + h * AC + 16 * q + 4 * r) = f32x4{oe_sum[0][r], oe_sum[1][r], oe_sum[2][r], oe_sum[3][r]};
    if (q == 0) scr[AH * AC + h] = dg_sum;
  }
  __syncthreads();
  for (int c = tid; c < AH * AC + AH; c += 512) {
    float sum = 0.f;
#pragma unroll
    for (int w = 0; w < 8; ++w) sum += lds[w * (2 * 16 * ELD) + c];
    wb_part[static_cast<int64_t>(blockIdx.x) * (AH * AC + AH) + c] = sum;
  }
}

int launch_pair_stream_bwd(const diffab_dims* d, const float* e, const float* P, float* G, const float* D2, const float* dfeat,
                           float* wb_part, const float* Wb, float* de, hipStream_t st) {
  const int K = d->K, rows = d->B * K;
  DIFFAB_REQUIRE(attention_split_supported(d), DIFFAB_ERR_UNSUPPORTED, "pair_stream_bwd: K must be 64 or 128");
  constexpr int RPW = 4;
  const size_t lds_b = static_cast<size_t>(8) * 2 * 16 * ELD * sizeof(float);
  DIFFAB_REQUIRE(rows % (8 * RPW) == 0, DIFFAB_ERR_UNSUPPORTED, "pair_stream_bwd: B K must be a multiple of %d", 8 * RPW);
  const dim3 grid_b(rows / (8 * RPW));
  if (K == 128) {
    DIFFAB_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(ipa_pair_stream_bwd_kernel<8, RPW>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds_b)));
    hipLaunchKernelGGL((ipa_pair_stream_bwd_kernel<8, RPW>), grid_b, dim3(512), lds_b, st, e, P, G, D2, dfeat, wb_part, rows, Wb, de);
  } else {
    DIFFAB_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(ipa_pair_stream_bwd_kernel<4, RPW>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds_b)));
    hipLaunchKernelGGL((ipa_pair_stream_bwd_kernel<4, RPW>), grid_b, dim3(512), lds_b, st, e, P, G, D2, dfeat, wb_part, rows, Wb, de);
  }
  DIFFAB_LAUNCH_CHECK();
  return DIFFAB_OK;
}

// ================================================================== d pair_ctx of ALL layers in one pass (round 6)
// de[i][j][c] += sum_l sum_h (P_l[h][i][j] do_e_l[i][h][c] + g_l[h][i][j] w_bias_l[h][c]).  Inside ipa_pair_stream_bwd_kernel the term costs
// a read-modify-write of the whole (B, K, K, C) gradient PER LAYER (250 of that kernel's 450 us at B = 128: 1.5 ms of an 8.3 ms step).
// With g_l and do_e_l of every layer kept (workspace) and P_l on the tape, one kernel at the end of the backward takes the same MFMA
// product over all layers - K = 16 "heads" (8 x P, 8 x g) per layer, up to 96 - and touches d pair_ctx once.  Same wave / lane layout
// as B' above: wave = RPW query rows, lane (l15, q); the A fragments of every layer (do_e of the row | w_bias) stay in registers over
// the row's key tiles, the [P ; g] B fragments come straight from the [b][h][i][j] images (sixteen consecutive keys per 16-lane group).
constexpr int DE_NL = 6;  // layers per launch (NL = 6: one launch)
struct DeLayers {
  const float* P[DE_NL]; const float* G[DE_NL]; const float* doe[DE_NL]; const float* Wb[DE_NL];
  int nl;
};
template <int NT, int RPW>
__global__ __launch_bounds__(256) void ipa_pair_de_layers_kernel(DeLayers L, float* __restrict__ de, int rows_total) {
  constexpr int K = 16 * NT;
  const int tid = threadIdx.x, lane0 = tid & 63, wv = tid >> 6;
  const int l15 = lane0 & 15, q = lane0 >> 4;
  const int64_t row_first = (static_cast<int64_t>(blockIdx.x) * 4 + wv) * RPW;  // four-wave work-groups: three groups per CU
#pragma unroll 1
  for (int rr = 0; rr < RPW; ++rr) {
    const int64_t row = row_first + rr;
    if (row >= rows_total) break;
    // B fragments straight from the [b][h][i][j] images: lane (key l15, q) takes B[h' = 4 q + s][key] = P (h' < 8: lanes q < 2) or g
    // (lanes q >= 2) of head (4 q + s) & 7 - sixteen consecutive keys per 16-lane group, no re-orientation through LDS
    const int64_t boff = ((row / K) * AH * K + static_cast<int64_t>((4 * q) & 7) * K + row % K) * K + l15;  // + s K K heads, + 16 jt keys
    // A fragments: lane (l15, q) holds M_l[h' = 4 q + s][c = 16 ct + l15]; h' < 8: do_e_l of the row, h' >= 8: w_bias_l
    float af[DE_NL][4][4];
    const float* bsrc[DE_NL];
#pragma unroll
    for (int l = 0; l < DE_NL; ++l) {
      // (ONE load per element through a per-lane base pointer: as `q < 2 ? doe[..] : Wb[..]` both loads were issued - 192 registers in flight)
      const int lc = l < L.nl ? l : 0;
      const float* bp = q < 2 ? L.doe[lc] + row * (AH * AC) + 4 * q * AC : L.Wb[lc] + 4 * (q - 2) * AC;
      bsrc[l] = (q < 2 ? L.P[lc] : L.G[lc]) + boff;
#pragma unroll
      for (int ct = 0; ct < 4; ++ct)
#pragma unroll
        for (int s = 0; s < 4; ++s) af[l][ct][s] = bp[s * AC + 16 * ct + l15];
    }
    float* derow = de + (row * K + l15) * AC + 4 * q;  // + 16 jt keys, + 16 ct channels
#pragma unroll 1
    for (int jt = 0; jt < NT; ++jt) {
      float bf[DE_NL][4];
#pragma unroll
      for (int l = 0; l < DE_NL; ++l)
#pragma unroll
        for (int s = 0; s < 4; ++s) bf[l][s] = bsrc[l][static_cast<int64_t>(s) * K * K + jt * 16];
      f32x4 dacc[4];
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) dacc[ct] = *reinterpret_cast<const f32x4*>(derow + static_cast<int64_t>(jt) * 16 * AC + 16 * ct);
#pragma unroll
      for (int l = 0; l < DE_NL; ++l) {
        if (l >= L.nl) break;
#pragma unroll
        for (int ct = 0; ct < 4; ++ct)
#pragma unroll
          for (int s = 0; s < 4; ++s) dacc[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[l][ct][s], bf[l][s], dacc[ct], 0, 0, 0);
      }
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) *reinterpret_cast<f32x4*>(derow + static_cast<int64_t>(jt) * 16 * AC + 16 * ct) = dacc[ct];
    }
  }
}

// P / G: [B][8][K][K] per layer; doe: [B K][8 * 64] per layer (the o_e columns of that layer's d feat); Wb: [8][64] per layer
int launch_pair_de_layers(const diffab_dims* d, int nl, const float* const* P, const float* const* G, const float* const* doe,
                          const float* const* Wb, float* de, hipStream_t st) {
  const int K = d->K, rows = d->B * K;
  DIFFAB_REQUIRE(attention_split_supported(d) && nl >= 1 && nl <= DE_NL && de != nullptr, DIFFAB_ERR_UNSUPPORTED,
                 "pair_de_layers: K must be 64 or 128, 1 <= layers <= %d", DE_NL);
  DeLayers L{};
  L.nl = nl;
  for (int l = 0; l < nl; ++l) { L.P[l] = P[l]; L.G[l] = G[l]; L.doe[l] = doe[l]; L.Wb[l] = Wb[l]; }
  constexpr int RPW = 2;
  const size_t lds_b = 0;
  const dim3 grid((rows + 4 * RPW - 1) / (4 * RPW));
  if (K == 128) hipLaunchKernelGGL((ipa_pair_de_layers_kernel<8, RPW>), grid, dim3(256), lds_b, st, L, de, rows);
  else hipLaunchKernelGGL((ipa_pair_de_layers_kernel<4, RPW>), grid, dim3(256), lds_b, st, L, de, rows);
  DIFFAB_LAUNCH_CHECK();
  return DIFFAB_OK;
}

// ------------------------------------------------------------------ host side
bool attention_split_supported(const diffab_dims* d) { return d->K == 64 || d->K == 128; }

size_t attention_split_workspace_floats(const diffab_dims* d) { return static_cast<size_t>(d->B) * AH * d->K * d->K; }

int launch_ipa_logits(const diffab_dims* d, const float* proj, const float* gamma, float* SP, hipStream_t st) {
  const int K = d->K;
  DIFFAB_REQUIRE(attention_split_supported(d), DIFFAB_ERR_UNSUPPORTED, "ipa_logits: K must be 64 or 128");
  const size_t lds_a = (static_cast<size_t>(K) * (KLD + GLD) + 4 * 16 * 36) * sizeof(float);
  const int RB = rows_per_wg(K);
  hipLaunchKernelGGL(ipa_logits_kernel<false>, dim3(d->B * AH * (K / RB)), dim3(256), lds_a, st, proj, gamma, SP, nullptr, K, RB);
  DIFFAB_LAUNCH_CHECK();
  return DIFFAB_OK;
}

// training backward: P[b][h][i][j] (normalised probabilities) and D2[b][h][i][j] (squared point distances), nothing else
int launch_attention_probs(const diffab_dims* d, const float* proj, const float* e, const float* Wb, const float* gamma, float* P,
                           float* D2, hipStream_t st) {
  const int K = d->K, rows = d->B * K;
  DIFFAB_REQUIRE(attention_split_supported(d), DIFFAB_ERR_UNSUPPORTED, "attention_probs: K must be 64 or 128");
  const size_t lds_a = (static_cast<size_t>(K) * (KLD + GLD) + 8 * 16 * 36) * sizeof(float);
  const int RB = rows_per_wg(K);
  hipLaunchKernelGGL(ipa_logits_kernel<true>, dim3(d->B * AH * (K / RB)), dim3(256), lds_a, st, proj, gamma, P, D2, K, RB);
  DIFFAB_LAUNCH_CHECK();
  constexpr int RPW = 4;
  const size_t lds_b = static_cast<size_t>(8) * 2 * 16 * ELD * sizeof(float);
  const dim3 grid_b((rows + 8 * RPW - 1) / (8 * RPW));
  if (K == 128) {
    DIFFAB_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(ipa_pair_stream_kernel<8, RPW, false>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds_b)));
    hipLaunchKernelGGL((ipa_pair_stream_kernel<8, RPW, false>), grid_b, dim3(512), lds_b, st, e, Wb, P, nullptr, rows);
  } else {
    DIFFAB_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(ipa_pair_stream_kernel<4, RPW, false>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds_b)));
    hipLaunchKernelGGL((ipa_pair_stream_kernel<4, RPW, false>), grid_b, dim3(512), lds_b, st, e, Wb, P, nullptr, rows);
  }
  DIFFAB_LAUNCH_CHECK();
  return DIFFAB_OK;
}

int launch_attention_split(const diffab_dims* d, const float* proj, const float* e, const float* R, const float* t, const float* Wb,
                           const float* gamma, float* feat, float* SP, hipStream_t st, float* D2) {
  const int K = d->K, rows = d->B * K;
  DIFFAB_REQUIRE(attention_split_supported(d), DIFFAB_ERR_UNSUPPORTED, "attention_split: K must be 64 or 128");
  const int RB = rows_per_wg(K);
  const dim3 grid_ac(d->B * AH * (K / RB));
  const size_t lds_a = (static_cast<size_t>(K) * (KLD + GLD) + 8 * 16 * 36) * sizeof(float), lds_c = static_cast<size_t>(K) * (VLD + PLD) * sizeof(float);
  if (D2) hipLaunchKernelGGL(ipa_logits_kernel<true>, grid_ac, dim3(256), lds_a, st, proj, gamma, SP, D2, K, RB);
  else hipLaunchKernelGGL(ipa_logits_kernel<false>, grid_ac, dim3(256), lds_a, st, proj, gamma, SP, nullptr, K, RB);
  DIFFAB_LAUNCH_CHECK();
  constexpr int RPW = 4;  // rows per wave: 8 RPW per work-group
  const size_t lds_b = static_cast<size_t>(8) * 2 * 16 * ELD * sizeof(float);
  const dim3 grid_b((rows + 8 * RPW - 1) / (8 * RPW));
  timer_begin(st);
  if (K == 128) {
    DIFFAB_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(ipa_pair_stream_kernel<8, RPW>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds_b)));
    hipLaunchKernelGGL((ipa_pair_stream_kernel<8, RPW>), grid_b, dim3(512), lds_b, st, e, Wb, SP, feat, rows);
  } else {
    DIFFAB_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(ipa_pair_stream_kernel<4, RPW>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds_b)));
    hipLaunchKernelGGL((ipa_pair_stream_kernel<4, RPW>), grid_b, dim3(512), lds_b, st, e, Wb, SP, feat, rows);
  }
  timer_end(st);
  DIFFAB_LAUNCH_CHECK();
  hipLaunchKernelGGL(ipa_pv_kernel, grid_ac, dim3(256), lds_c, st, proj, SP, R, t, feat, K, RB);
  DIFFAB_LAUNCH_CHECK();
  return DIFFAB_OK;
}

}  // namespace diffab
