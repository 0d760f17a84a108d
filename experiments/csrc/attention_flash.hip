// attention_flash.hip - the IPA attention of the benchmark geometry (D=128, C=64, H=8, DS=32, PQ=PV=8) as ONE kernel whose pair
// stream never stops: a software pipeline over key tiles with an online softmax.
// Reference: InvariantPointAttentionLayer.forward, diffab_pytorch.py:389-465 (logits :416-439, softmax :443, sums :445-457).
//
// Why (round-2 measurements, tools/attn_phase_profile.py on the three-phase kernel of denoiser_fast.hip): a work-group there lives
// 98.6 k cycles = logits 24 % (VALU-bound, matrix pipe idle) | barrier 10 % | pair stream 41 % (matrix-pipe-bound, the ONLY phase
// with HBM loads in flight) | barrier 9 % | P x V 16 %.  The matrix pipe needs 47 k of those cycles and the HBM share of a CU
// 47 k: the phases serialise what could overlap, because a wave holds a whole 32 KiB pair row in registers until the row's maximum
// is known, and the (16 rows x 8 heads x K) logits / probabilities image fills the LDS.
//
// Here a work-group (8 waves, one per CU) still owns (patch, 16 query rows), but the work is cut per KEY TILE (16 keys) and every
// wave does a slice of all three kinds, meeting the others once per tile:
//   A(k)  wave = head h:      logits of key tile k for 16 rows (scalar part on the MFMA, point part as direct differences on the
//                             VALU - the expanded form loses ~1e-5)                                      -> S ring in LDS
//   B(k)  wave = rows 2w,2w+1: per row the pair tile e[i][16 keys][64] (4 KiB, non-temporal, register ring, requested from the first
//                             instruction of the kernel), pair bias on the MFMA, + S_k, running max / sum, P~_k = exp(s - m)
//                             -> P ring in LDS, o_e = alpha . o_e + P~_k e on the MFMA
//   C(k)  wave = head h:      acc = alpha_k . acc + P~_k x [v_s | v_pts]_k  (MFMA; rescaled like flash attention)
//   step k:  C(k-2), B(k-1), A(k), barrier          (k = 0 .. NT+1, NT = K / 16)
// No wave ever holds more than a 4 KiB pair tile per row, LDS holds two key tiles of S and P~ (41 KiB) instead of the whole image,
// and the two waves of a SIMD run the same mix of matrix, vector and memory work out of phase, so each one's dependency stalls
// are the other's issue slots.  (A first version with fixed roles - four stream waves, four head waves - was parity-green but 8 %
// SLOWER than the three-phase kernel: timing-only ablations showed the two roles' times nearly adding up, the stream wave's
// bias -> softmax -> o_e chain being serial and its SIMD partner not having enough matrix work to fill it.)
// Matrix-pipe work: 5 632 MFMA 16x16x4 per work-group (5 888 before: the x and y sums of the value points now share a tile).
//
// Numerics: online softmax = the reference's softmax up to fp32 rounding (running maximum instead of the row maximum; every
// partial sum rescaled by exp(m_old - m_new)); deterministic (no atomics), independent of the batch composition.
#include <cstdlib>
#include <type_traits>

#include "common.h"
#include "denoiser_internal.h"

namespace diffab {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
#define MEM_FENCE() asm volatile("" ::: "memory")
#ifdef DIFFAB_ACCURATE_EXP
#define FAST_EXP(x) expf(x)
#else
#define FAST_EXP(x) __expf(x)
#endif

namespace {
constexpr int AH = 8, ADS = 32, AP = 8, AC = 64;
constexpr int ANP = 3 * AH * ADS + 3 * AH * AP * 3;             // 1344 projection columns
constexpr int AF = AH * ADS + AH * AC + AH * AP * 3 + AH * AP;  // 1024 feature columns
constexpr int OFF_QS = 0, OFF_KS = 256, OFF_VS = 512, OFF_GQ = 768, OFF_GK = 960, OFF_GV = 1152;
constexpr int FOFF_OS = 0, FOFF_OE = 256, FOFF_OL = 768, FOFF_ON = 960;
constexpr int TI = 16;  // query rows per work-group

// ---- LDS map (floats)
constexpr int HS = 20;                 // head stride inside a ring row: 16 keys + 4 (16-byte aligned, spreads banks)
constexpr int RS = AH * HS + 4;        // ring row stride 164: 4 RS = 16 (mod 32) -> the 4-byte S stores are 2-way at worst (free)
constexpr int RING = TI * RS;          // one key tile of S or P~: 2624 floats
constexpr int KLD = 40, GLD = 28;      // staged key-tile strides (k_s 32 + 8, k_pts 24 + 4): ds_read_b128 conflict-free
constexpr int KT = 16 * KLD + 16 * GLD;  // 1088 floats per staged (head, key tile)
constexpr int ELD = 72;                // pair-tile stride for the bias re-orientation
constexpr int L_S = 0;                        // S ring       [2][TI][RS]
constexpr int L_P = L_S + 2 * RING;           // P~ ring      [2][TI][RS]
constexpr int L_AL = L_P + 2 * RING;          // alpha ring   [2][AH][TI]
constexpr int L_LI = L_AL + 2 * AH * TI;      // 1 / l        [AH][TI]
constexpr int L_GQ = L_LI + AH * TI;          // query points [AH][TI][24]
constexpr int L_HK = L_GQ + AH * TI * 24;     // key-tile staging of each wave's head   [8 waves][KT]
constexpr int L_SE = L_HK + 8 * KT;           // pair-tile re-orientation of each wave  [8 waves][16][ELD]
constexpr int L_WB = L_SE + 8 * 16 * ELD;     // bias-product B fragments      [4][64 lanes][4]
constexpr int L_END = L_WB + 4 * 64 * 4;
constexpr size_t kFlashLdsBytes = static_cast<size_t>(L_END) * sizeof(float);  // 127 488 B
#ifndef DIFFAB_FLASH_ERING
#define DIFFAB_FLASH_ERING 3
#endif
constexpr int ERING = DIFFAB_FLASH_ERING;  // pair row-tiles per wave in registers (ERING - 1 in flight beside the one consumed)
}  // namespace

template <int NT>
__global__ __launch_bounds__(512) void ipa_attn_flash_kernel(const float* __restrict__ proj, const float* __restrict__ e,
                                                             const float* __restrict__ R, const float* __restrict__ t,
                                                             const float* __restrict__ Wb, const float* __restrict__ gamma,
                                                             float* __restrict__ feat, int B, unsigned long long* __restrict__ stamps) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int K = NT * 16;
  constexpr int ntile = K / TI;
  constexpr int NRS = 2 * NT;  // row-steps of a wave: (key tile jt, row ii) = rs = 2 jt + ii
  // XCD-aware map (blocks b and b + 8 share an XCD): all row tiles of a patch on one XCD, so the K/V side is an L2 hit for 7 of 8
  int b, tile;
  if ((B & 7) == 0) {
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    b = (slot / ntile) * 8 + xcd;
    tile = slot % ntile;
  } else {
    b = blockIdx.x / ntile;
    tile = blockIdx.x % ntile;
  }
  const int i0 = tile * TI;
  const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, q = lane >> 4;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int64_t prow0 = static_cast<int64_t>(b) * K;
  const float scale_t = 0.57735026918962576f;  // 3^-1/2   (diffab_pytorch.py:387, :439)
  const float scale_s = 0.17677669529663687f;  // 32^-1/2  (:353)
  auto stamp = [&](int k) {  // diagnostics (stamps == nullptr in production)
    if (stamps != nullptr) {
      __builtin_amdgcn_sched_barrier(0);
      const unsigned long long tnow = __builtin_amdgcn_s_memtime();
      if (lane == 0) stamps[(static_cast<size_t>(blockIdx.x) * 8 + wv) * 8 + k] = tnow;
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  stamp(0);

  // ======================================================================================= B: pair stream of rows 2 wv, 2 wv + 1
  const int hl = l15 & 7;  // lanes l15 >= 8 duplicate head l15 - 8 (their MFMA columns are the padding half of the 16-wide tile)
  // addresses as (wave-uniform pointer, 32-bit lane offset): one VGPR per stream instead of a 64-bit pointer per load
  const float* ebase = e + ((prow0 + i0 + 2 * wv) * K) * AC;  // + ii K AC + jt 16 AC (uniform)
  const int eoff = (4 * q) * AC + 4 * l15;                    // + r AC
  f32x4 ev[ERING][4];
  auto load_rs = [&](int rs) {  // -> ring slot rs % ERING
    const float* eu = ebase + (rs & 1) * (K * AC) + (rs >> 1) * (16 * AC);
#pragma unroll
    for (int r = 0; r < 4; ++r) ev[rs % ERING][r] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(eu + eoff + r * AC));
  };
#pragma unroll
  for (int rs = 0; rs < ERING && rs < NRS; ++rs) load_rs(rs);  // the stream starts with the kernel
  MEM_FENCE();
  // B fragments of the bias product, Wb[h][16 sg + 4 q + s], per lane: kept in LDS [sg][lane] (the same for every wave; 16 VGPRs
  // that the pipeline needs more), read back beside the pair tile in front()
  if (wv == 0) {
#pragma unroll
    for (int sg = 0; sg < 4; ++sg)
      *reinterpret_cast<f32x4*>(lds + L_WB + (sg * 64 + lane) * 4) = *reinterpret_cast<const f32x4*>(Wb + hl * AC + 16 * sg + 4 * q);
  }
  float* escr = lds + L_SE + wv * (16 * ELD);
  float m_run[2] = {-INFINITY, -INFINITY}, l_run[2] = {0.f, 0.f};
  f32x4 oe[2][4];
#pragma unroll
  for (int ii = 0; ii < 2; ++ii)
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) oe[ii][ct] = f32x4{0.f, 0.f, 0.f, 0.f};

  // ======================================================================================= A / C: head h = wv
  const int h = wv;
  float* kscr = lds + L_HK + wv * KT;
  // line-shaped loads of one key tile (16 keys) of head h: k_s 16 x 128 B (8 lanes per key), k_pts 16 x 96 B (6 lanes per key)
  const int g0 = lane, g1 = lane < 32 ? lane + 64 : 95;  // k_pts chunk ids (0..95): key = id / 6, chunk = id % 6; lanes >= 32 repeat
                                                         // chunk 95 (same data to the same LDS address: no divergent branch)
  const float* krow = proj + prow0 * ANP;
  const int ks_off = (lane >> 3) * ANP + OFF_KS + h * ADS + 4 * (lane & 7);  // + 8 ANP for keys 8..15, + 16 jt ANP
  const int gk_off0 = (g0 / 6) * ANP + OFF_GK + h * 24 + 4 * (g0 % 6), gk_off1 = (g1 / 6) * ANP + OFF_GK + h * 24 + 4 * (g1 % 6);
  const int ks_dst = (lane >> 3) * KLD + 4 * (lane & 7);
  const int gk_dst0 = 16 * KLD + (g0 / 6) * GLD + 4 * (g0 % 6), gk_dst1 = 16 * KLD + (g1 / 6) * GLD + 4 * (g1 % 6);
  f32x4 st[4];
  auto load_keys = [&](int jt) {
    const float* base = krow + jt * (16 * ANP);  // wave-uniform
    st[0] = *reinterpret_cast<const f32x4*>(base + ks_off);
    st[1] = *reinterpret_cast<const f32x4*>(base + ks_off + 8 * ANP);
    st[2] = *reinterpret_cast<const f32x4*>(base + gk_off0);
    st[3] = *reinterpret_cast<const f32x4*>(base + gk_off1);
  };
  auto stage_keys = [&]() {
    *reinterpret_cast<f32x4*>(kscr + ks_dst) = st[0];
    *reinterpret_cast<f32x4*>(kscr + ks_dst + 8 * KLD) = st[1];
    *reinterpret_cast<f32x4*>(kscr + gk_dst0) = st[2];
    *reinterpret_cast<f32x4*>(kscr + gk_dst1) = st[3];
  };
  // value side of one key tile, B operands of the P~ x V products: lane (n = l15, k = q), key 16 jt + 4 q + s.
  // Point sums: columns 0..7 of one tile hold x of the 8 points, columns 8..15 y (z in a second tile), so 16 MFMAs per tile, not 20.
  float2 vs[4];
  float gxy[4], gz[4];
  const int vs_off = (4 * q) * ANP + OFF_VS + h * ADS + 2 * l15;
  const int gv_off = (4 * q) * ANP + OFF_GV + h * 24 + 3 * hl;  // + (16 jt + s) ANP
  auto load_vals = [&](int jt) {
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const float* pu = krow + (jt * 16 + s) * ANP;  // wave-uniform
      vs[s] = *reinterpret_cast<const float2*>(pu + vs_off);
      gxy[s] = pu[gv_off + (l15 >> 3)];
      gz[s] = pu[gv_off + 2];
    }
  };
  load_keys(0);
  const float coef_p = -0.5f * 0.16666666666666666f * gamma[h];  // -1/2 (4.5*8)^-1/2 gamma_h  (:372, :431-436)
  f32x4 qa[2];
  {
    const float* qrow = proj + (prow0 + i0 + l15) * ANP + OFF_QS + h * ADS + 4 * q;  // A operand: q_s rows i0 + l15, k = 16 sg + 4 q + s
    qa[0] = *reinterpret_cast<const f32x4*>(qrow);
    qa[1] = *reinterpret_cast<const f32x4*>(qrow + 16);
    // query points of the 16 rows of this head -> LDS [h][row][24] (96 chunks of 16 bytes)
    float* gdst = lds + L_GQ + h * (TI * 24);
    const float* gsrc = proj + (prow0 + i0) * ANP + OFF_GQ + h * 24;
    *reinterpret_cast<f32x4*>(gdst + (g0 / 6) * 24 + 4 * (g0 % 6)) = *reinterpret_cast<const f32x4*>(gsrc + (g0 / 6) * ANP + 4 * (g0 % 6));
    *reinterpret_cast<f32x4*>(gdst + (g1 / 6) * 24 + 4 * (g1 % 6)) = *reinterpret_cast<const f32x4*>(gsrc + (g1 / 6) * ANP + 4 * (g1 % 6));
  }
  f32x4 os[2], ogxy = {0.f, 0.f, 0.f, 0.f}, ogz = {0.f, 0.f, 0.f, 0.f};
  os[0] = os[1] = f32x4{0.f, 0.f, 0.f, 0.f};

  // A row-step in two halves, software-pipelined by hand (an in-order wave overlaps nothing by itself):
  //   front(rs): pair tile -> bias orientation through LDS, 16 bias MFMAs in four independent chains   (needs only the pair tile:
  //                                                                              runs a row-step early, across the step barrier too)
  //   back(rs):  + S, running max / sum, P~ -> LDS, rescale, 16 o_e MFMAs, refill  (needs S of the tile: after the step barrier)
  // No lane-divergent branch inside (they end a scheduling region): lanes l15 >= 8 and the four quarters store duplicates.
  f32x4 acc_n;
  auto front = [&](int rs) {
    const int sl = rs % ERING;
    float* t_ = escr + 4 * q * ELD + 4 * l15;  // write [key 4 q + r][channel chunk l15], read [key l15][channels 16 sg + 4 q ..]
#pragma unroll
    for (int r = 0; r < 4; ++r) *reinterpret_cast<f32x4*>(t_ + r * ELD) = ev[sl][r];
    f32x4 a4[4];
#pragma unroll
    for (int sg = 0; sg < 4; ++sg) a4[sg] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 ea[4];
    f32x4 wb[4];
#pragma unroll
    for (int sg = 0; sg < 4; ++sg) {
      ea[sg] = *reinterpret_cast<const f32x4*>(escr + l15 * ELD + 16 * sg + 4 * q);  // e[i][key l15][16 sg + 4 q + s]
      wb[sg] = *reinterpret_cast<const f32x4*>(lds + L_WB + (sg * 64 + lane) * 4);
    }
#ifdef FLASH_ABL_NOBIAS
    for (int sg = 0; sg < 4; ++sg) a4[sg] = ea[sg];
#else
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int sg = 0; sg < 4; ++sg) a4[sg] = __builtin_amdgcn_mfma_f32_16x16x4f32(ea[sg][s], wb[sg][s], a4[sg], 0, 0, 0);
#endif
    acc_n = (a4[0] + a4[1]) + (a4[2] + a4[3]);  // D: column = head l15, rows = keys 4 q + r of this tile
  };
  // Reductions over the four quarters (lanes l, l^16, l^32, l^48) with the gfx950 row swaps instead of LDS round trips:
  // v_permlane32_swap exchanges lanes 32-63 of its first register with lanes 0-31 of its second, v_permlane16_swap the odd rows
  // of the first with the even rows of the second; fed two COPIES of x they leave (x of the lower half / even row, x of the upper
  // half / odd row) in every lane.  Written in assembly: the builtin, given the same value twice, is folded to "both results = its
  // first" by hipcc (ROCm 7.2; tools/permlane_probe.hip shows it), which silently drops the other half.  s_nop 1 = the two wait
  // states a VALU write of an operand needs before the swap reads it (the compiler pads nothing inside an asm statement).
  // Not volatile: a pure function of its operands, free to be scheduled.
  struct F2 { float a, b; };
  auto swap32 = [](float x) {
    F2 r{x, x};
    asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(r.a), "+v"(r.b));
    return r;
  };
  auto swap16 = [](float x) {
    F2 r{x, x};
    asm("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(r.a), "+v"(r.b));
    return r;
  };
  auto xq_max = [&](float x) {
    F2 r = swap32(x);
    r = swap16(fmaxf(r.a, r.b));
    return fmaxf(r.a, r.b);
  };
  auto xq_sum = [&](float x) {
    F2 r = swap32(x);
    r = swap16(r.a + r.b);
    return r.a + r.b;
  };
  auto back = [&](int rs, const f32x4 acc) {
    const int jt = rs >> 1, ii = rs & 1, sl = rs % ERING, il = 2 * wv + ii;
    const float* Sk = lds + L_S + (jt & 1) * RING;
    float* Pk = lds + L_P + (jt & 1) * RING;
    float* Ak = lds + L_AL + (jt & 1) * (AH * TI);
    const f32x4 sv = *reinterpret_cast<const f32x4*>(Sk + il * RS + hl * HS + 4 * q);
    float s_[4], mt = -INFINITY;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      s_[r] = sv[r] + scale_t * acc[r];
      mt = fmaxf(mt, s_[r]);
    }
#ifdef FLASH_ABL_NOSOFTMAX
    const float m_new = mt, alpha = 0.5f;
    f32x4 pv = {s_[0], s_[1], s_[2], s_[3]};
    float ls = s_[0];
#else
    mt = xq_max(mt);
    const float m_new = fmaxf(m_run[ii], mt);
    const float alpha = FAST_EXP(m_run[ii] - m_new);  // 0 on the first tile (m_run = -inf)
    f32x4 pv;
    float ls = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      pv[r] = FAST_EXP(s_[r] - m_new);
      ls += pv[r];
    }
#endif
    *reinterpret_cast<f32x4*>(Pk + il * RS + hl * HS + 4 * q) = pv;  // (lanes l15 and l15 + 8 store the same values)
    Ak[hl * TI + il] = alpha;                                         // (so do the four quarters)
    // o_e[head n = l15][channel 4 m + ct] += sum_keys e[key][4 m + ct] P~[n][key]:  A = e (m = l15, k = q), B = P~ (n = l15, k = q)
#pragma unroll
    for (int ct = 0; ct < 4; ++ct)
#pragma unroll
      for (int r = 0; r < 4; ++r) oe[ii][ct][r] *= alpha;
#ifdef FLASH_ABL_NOOE
    for (int ct = 0; ct < 4; ++ct) oe[ii][ct] += ev[sl][ct] * pv[ct];
#else
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) oe[ii][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(ev[sl][r][ct], pv[r], oe[ii][ct], 0, 0, 0);
#endif
#ifndef FLASH_ABL_NOSOFTMAX
    ls = xq_sum(ls);
#endif
    l_run[ii] = l_run[ii] * alpha + ls;
    m_run[ii] = m_new;
    asm("" : "+v"(l_run[ii]));  // keep the running sum a running sum: the scheduler otherwise defers the whole chain to the end of
                                // the kernel and spills every tile's (alpha, partial sum) pair until then
#ifdef FLASH_ABL_NOLOAD
    if (false) {
#else
    if (rs + ERING < NRS) {  // the slot is free: request the row-tile ERING row-steps ahead
#endif
      load_rs(rs + ERING);
      MEM_FENCE();
    }
    if (jt == NT - 1) {  // last key tile of this row: normalise and store o_e, publish 1 / l for the head products
      const float inv = 1.0f / l_run[ii];
      int le = lane;  // opaque copy: otherwise the store addresses are computed at the top of the kernel and spilled
      asm volatile("" : "+v"(le));
      const int he = le & 7, qe = le >> 4;
      lds[L_LI + he * TI + il] = inv;
      if ((le & 15) < 8) {
        float* fo = feat + (prow0 + i0 + il) * AF + FOFF_OE + he * AC + 16 * qe;  // D rows m = 4 q + r' <-> channels 16 q + 4 r' + ct
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const f32x4 v = {oe[ii][0][r] * inv, oe[ii][1][r] * inv, oe[ii][2][r] * inv, oe[ii][3][r] * inv};
          *reinterpret_cast<f32x4*>(fo + 4 * r) = v;
        }
      }
    }
  };
  MEM_FENCE();
  __syncthreads();  // prologue barrier (the staged query points are read by their own wave only; it also lines the waves up)
  stamp(1);
  front(0);
#pragma unroll
  for (int k = 0; k < NT + 2; ++k) {
    if (k < NT) {  // ---- keys of tile k: registers -> LDS (fragments are read back in A(k) below), then request tile k + 1
      stage_keys();
      if (k + 1 < NT) load_keys(k + 1);
      MEM_FENCE();
    }
#ifndef FLASH_ABL_NOC  // (FLASH_ABL_*: timing-only ablation builds, results are garbage)
    if (k >= 2) {  // ---- C(k - 2): acc = alpha . acc + P~ x V
      const int jt = k - 2;
      const float* Pk = lds + L_P + (jt & 1) * RING;
      const float* Ak = lds + L_AL + (jt & 1) * (AH * TI);
      const f32x4 pa = *reinterpret_cast<const f32x4*>(Pk + l15 * RS + h * HS + 4 * q);  // A: P~[row l15][key 4 q + s]
      const f32x4 al = *reinterpret_cast<const f32x4*>(Ak + h * TI + 4 * q);             // D rows 4 q + r
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        os[0][r] *= al[r];
        os[1][r] *= al[r];
        ogxy[r] *= al[r];
        ogz[r] *= al[r];
      }
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        os[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(pa[s], vs[s].x, os[0], 0, 0, 0);
        os[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(pa[s], vs[s].y, os[1], 0, 0, 0);
        ogxy = __builtin_amdgcn_mfma_f32_16x16x4f32(pa[s], gxy[s], ogxy, 0, 0, 0);
        ogz = __builtin_amdgcn_mfma_f32_16x16x4f32(pa[s], gz[s], ogz, 0, 0, 0);
      }
    }
    if (k >= 1 && k <= NT) {  // ---- values of tile k - 1, consumed by C(k - 1) in the next step
      load_vals(k - 1);
      MEM_FENCE();
    }
#endif
    if (k >= 1 && k <= NT) {  // ---- B(k - 1): this wave's two rows
#pragma unroll
      for (int ii = 0; ii < 2; ++ii) {
        const int rs = 2 * (k - 1) + ii;
        const f32x4 a0 = acc_n;
#ifndef FLASH_ABL_NOB
        if (rs + 1 < NRS) front(rs + 1);
        back(rs, a0);
#endif
      }
    }
#ifndef FLASH_ABL_NOA
    if (k < NT) {  // ---- A(k): logits of key tile k for head h
      float* Sk = lds + L_S + (k & 1) * RING;
      const f32x4 kb0 = *reinterpret_cast<const f32x4*>(kscr + l15 * KLD + 4 * q);  // k_s[16 k + l15][16 sg + 4 q + s]
      const f32x4 kb1 = *reinterpret_cast<const f32x4*>(kscr + l15 * KLD + 16 + 4 * q);
      f32x4 gk[6];
#pragma unroll
      for (int cc = 0; cc < 6; ++cc) gk[cc] = *reinterpret_cast<const f32x4*>(kscr + 16 * KLD + l15 * GLD + 4 * cc);
      f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acc2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(qa[0][s], kb0[s], acc, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(qa[1][s], kb1[s], acc2, 0, 0, 0);
      }
      // acc[r] + acc2[r] = q_s[i0 + 4 q + r] . k_s[key 16 k + l15]
      const float* gql = lds + L_GQ + h * (TI * 24) + (4 * q) * 24;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        f32x2 d2v = {0.f, 0.f};  // packed fp32: two coordinates per instruction, two partial sums added at the end
#ifdef FLASH_ABL_NOPTS
        for (int cc = 0; cc < 0; ++cc) {
#else
#pragma unroll
        for (int cc = 0; cc < 6; ++cc) {
#endif
          const f32x4 gq = *reinterpret_cast<const f32x4*>(gql + r * 24 + 4 * cc);  // the 16 lanes of a quarter share the address
          f32x2 dlo, dhi;  // packed subtract spelled in assembly: the compiler splits a vector fsub into two v_sub_f32
          asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]"
              : "=v"(dlo)
              : "v"(__builtin_shufflevector(gq, gq, 0, 1)), "v"(__builtin_shufflevector(gk[cc], gk[cc], 0, 1)));
          asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]"
              : "=v"(dhi)
              : "v"(__builtin_shufflevector(gq, gq, 2, 3)), "v"(__builtin_shufflevector(gk[cc], gk[cc], 2, 3)));
          d2v = __builtin_elementwise_fma(dlo, dlo, d2v);
          d2v = __builtin_elementwise_fma(dhi, dhi, d2v);
        }
        const float d2 = d2v[0] + d2v[1];
        Sk[(4 * q + r) * RS + h * HS + l15] = scale_t * ((acc[r] + acc2[r]) * scale_s + coef_p * d2);
      }
    }
#endif
    __syncthreads();
    if (k == 0) stamp(2);
    if (k == 1) stamp(3);
    if (k == 4) stamp(4);
    if (k == NT) stamp(6);
    if (k == NT + 1) stamp(7);
  }
  // (Tried: waves 4-7 taking the stages of a step in another order - logits first - so that one wave's VALU-heavy stage runs beside
  //  its SIMD partner's MFMA-heavy one.  Expressed as a per-step branch it spills 267 registers, as two copies of the pipeline 74.)
  // ---- epilogue of head h: D rows i = 4 q + r, column n = l15; normalise, global -> local frames, norms -> feature row
  {
    const f32x4 inv4 = *reinterpret_cast<const f32x4*>(lds + L_LI + h * TI + 4 * q);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int il = 4 * q + r;
      const int64_t row = prow0 + i0 + il;
      const float inv = inv4[r];
      float* fr = feat + row * AF;
      *reinterpret_cast<float2*>(fr + FOFF_OS + h * ADS + 2 * l15) = make_float2(os[0][r] * inv, os[1][r] * inv);
      const float gy_ = __shfl_xor(ogxy[r], 8);  // lanes 0..7 hold x of point l15, lanes 8..15 y of point l15 - 8
      if (l15 < 8) {
        const float* Rr = R + row * 9;
        const float* tr = t + row * 3;
        const float dx = ogxy[r] * inv - tr[0], dy = gy_ * inv - tr[1], dz = ogz[r] * inv - tr[2];
        const float lx = dx * Rr[0] + dy * Rr[1] + dz * Rr[2];  // (p - t) R^T   (diffab_pytorch.py:336)
        const float ly = dx * Rr[3] + dy * Rr[4] + dz * Rr[5];
        const float lz = dx * Rr[6] + dy * Rr[7] + dz * Rr[8];
        float* fo = fr + FOFF_OL + h * 24 + 3 * l15;
        fo[0] = lx; fo[1] = ly; fo[2] = lz;
        fr[FOFF_ON + h * AP + l15] = sqrtf(lx * lx + ly * ly + lz * lz);
      }
    }
  }
  stamp(5);
}

// Sixteen-wave form of the same pipeline: waves 0-7 run B only (two rows each), waves 8-15 run A and C only (one head each), 128
// registers per wave, four waves per SIMD - twice as many instruction streams for the scheduler to interleave as in the eight-wave
// form above, where matrix, vector and LDS work of the two co-resident waves was measured to add up rather than overlap.
// PLANES: `e` is the two-plane fp16 image of the pair embedding (pair_split_kernel, denoiser_fast.hip: same bytes, e s = h1 + h2 to
// 2^-23 of the tensor maximum, each 1 KiB block one A fragment of the bias product), esc = {s, 1 / s}.  The producers' two products
// then run on the f16 matrix cores as three exact partial products each: the bias straight from the loaded registers (6 MFMAs
// 16x16x32 instead of 16 f32 MFMAs and an LDS round trip), o_e with the tile staged once in a per-wave [key][channel] image and read
// back through ds_read_b64_tr_b16 (12 MFMAs 16x16x16 instead of 16 f32 MFMAs).
template <int NT, bool PLANES = false>
__global__ __launch_bounds__(1024) void ipa_attn_flash16_kernel(const float* __restrict__ proj, const float* __restrict__ e,
                                                                const float* __restrict__ R, const float* __restrict__ t,
                                                                const float* __restrict__ Wb, const float* __restrict__ gamma,
                                                                float* __restrict__ feat, int B, unsigned long long* __restrict__ stamps,
                                                                const float* __restrict__ esc = nullptr) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int K = NT * 16;
  constexpr int ntile = K / TI;
  constexpr int NRS = 2 * NT;  // row-steps of a wave: (key tile jt, row ii) = rs = 2 jt + ii
  // XCD-aware map (blocks b and b + 8 share an XCD): all row tiles of a patch on one XCD, so the K/V side is an L2 hit for 7 of 8
  int b, tile;
  if ((B & 7) == 0) {
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    b = (slot / ntile) * 8 + xcd;
    tile = slot % ntile;
  } else {
    b = blockIdx.x / ntile;
    tile = blockIdx.x % ntile;
  }
  const int i0 = tile * TI;
  const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, q = lane >> 4;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int64_t prow0 = static_cast<int64_t>(b) * K;
  const float scale_t = 0.57735026918962576f;  // 3^-1/2   (diffab_pytorch.py:387, :439)
  const float scale_s = 0.17677669529663687f;  // 32^-1/2  (:353)
  auto stamp = [&](int k) {  // diagnostics (stamps == nullptr in production)
    if (stamps != nullptr) {
      __builtin_amdgcn_sched_barrier(0);
      const unsigned long long tnow = __builtin_amdgcn_s_memtime();
      if (lane == 0 && (wv < 4 || wv >= 12)) stamps[(static_cast<size_t>(blockIdx.x) * 8 + (wv & 7)) * 8 + k] = tnow;
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  stamp(0);

  const int wv_role = wv;
  if (wv_role < 8) {
    const int wv = wv_role;  // row pair of this B wave
    // ======================================================================================= B: pair stream of rows 2 wv, 2 wv + 1
    const int hl = l15 & 7;  // lanes l15 >= 8 duplicate head l15 - 8 (their MFMA columns are the padding half of the 16-wide tile)
    // addresses as (wave-uniform pointer, 32-bit lane offset): one VGPR per stream instead of a 64-bit pointer per load
    const float* ebase = e + ((prow0 + i0 + 2 * wv) * K) * AC;  // + ii K AC + jt 16 AC (uniform)
    const int eoff = (4 * q) * AC + 4 * l15;                    // + r AC
    f32x4 ev[ERING][4];
    auto load_rs = [&](int rs) {  // -> ring slot rs % ERING
      const float* eu = ebase + (rs & 1) * (K * AC) + (rs >> 1) * (16 * AC);
      if constexpr (PLANES) {  // four 1 KiB blocks (plane p, k-step ks) of the tile, lane order: ev[slot][2 p + ks]
#pragma unroll
        for (int r = 0; r < 4; ++r) ev[rs % ERING][r] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(eu + r * 256 + lane * 4));
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) ev[rs % ERING][r] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(eu + eoff + r * AC));
      }
    };
#pragma unroll
    for (int rs = 0; rs < ERING && rs < NRS; ++rs) load_rs(rs);  // the stream starts with the kernel
    MEM_FENCE();
    // B fragments of the bias product, Wb[h][16 sg + 4 q + s], per lane: kept in LDS [sg][lane] (the same for every wave; 16 VGPRs
    // that the pipeline needs more), read back beside the pair tile in front()
    // PLANES: bias B fragments as two fp16 planes [plane][ks]: lane (head l15, channels 32 ks + 8 q ..), scaled by sw into [128, 256);
    // kept in LDS (L_WB: [2 p + ks][lane] x 16 bytes, written by wave 0, the same for every wave) - the producers have 128 registers
    float bscale = scale_t, oscale = 1.0f;
    if constexpr (PLANES) {
      f32x4 wv4[2][2];
      float wmax = 0.f;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
          wv4[ks][hf] = *reinterpret_cast<const f32x4*>(Wb + hl * AC + 32 * ks + 8 * q + 4 * hf);
#pragma unroll
          for (int s_ = 0; s_ < 4; ++s_) wmax = fmaxf(wmax, fabsf(wv4[ks][hf][s_]));
        }
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) wmax = fmaxf(wmax, __shfl_xor(wmax, o));
      const int ew = static_cast<int>((__float_as_uint(wmax) >> 23) & 255u);
      const bool okw = ew > 0 && ew < 231;
      const float sw = okw ? __uint_as_float(static_cast<unsigned>(127 + 7 + 127 - ew) << 23) : 1.0f;
      const float isw = okw ? __uint_as_float(static_cast<unsigned>(ew - 7) << 23) : 1.0f;
      if (wv == 0) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          f16x8 w1, w2;
#pragma unroll
          for (int c8 = 0; c8 < 8; ++c8) {
            const float x = wv4[ks][c8 >> 2][c8 & 3] * sw;  // lanes l15 >= 8 DUPLICATE head l15 - 8 (they store the same P~ / alpha values)
            const _Float16 h1 = static_cast<_Float16>(x);
            w1[c8] = h1;
            w2[c8] = static_cast<_Float16>(x - static_cast<float>(h1));
          }
          *reinterpret_cast<f16x8*>(lds + L_WB + ((0 * 2 + ks) * 64 + lane) * 4) = w1;
          *reinterpret_cast<f16x8*>(lds + L_WB + ((1 * 2 + ks) * 64 + lane) * 4) = w2;
        }
      }
      bscale = scale_t * isw;          // logits: bias = (sum e s_i w sw) / (s_i sw), x 1 / s_i per row in the row-step
      oscale = 1.0f / 256.0f;          // o_e: probabilities enter scaled by 256
    } else if (wv == 0) {
#pragma unroll
      for (int sg = 0; sg < 4; ++sg)
        *reinterpret_cast<f32x4*>(lds + L_WB + (sg * 64 + lane) * 4) = *reinterpret_cast<const f32x4*>(Wb + hl * AC + 16 * sg + 4 * q);
    }
    float inv_s2[2] = {1.0f, 1.0f};  // PLANES: 1 / s_i of this wave's two pair rows (per-row power-of-two scales of the planes)
    if constexpr (PLANES) {
      inv_s2[0] = esc[2 * (prow0 + i0 + 2 * wv) + 1];
      inv_s2[1] = esc[2 * (prow0 + i0 + 2 * wv + 1) + 1];
    }
    float* escr = lds + L_SE + wv * (16 * ELD);
    // PLANES: per-wave [2 planes][16 keys][128 bytes] image of the current tile (4 KiB of the 4.5 KiB slot); 8-byte unit u of row r at
    // u ^ (4 ((r >> 1) & 3)): the transposed reads of a 32-lane half touch 32 distinct bank pairs
    char* trt = reinterpret_cast<char*>(escr);
    const int wr_off = l15 * 128 + 8 * ((2 * q) ^ (4 * ((l15 >> 1) & 3)));      // ^ 64 ks: unit 8 ks + 2 q of row (key) l15
    const int rrow = 4 * q + (l15 >> 2);
    const int rd_off = rrow * 128 + 8 * ((l15 & 3) ^ (4 * ((rrow >> 1) & 3)));  // ^ 32 ct: unit 4 ct + (l15 & 3) of row rrow
    float m_run[2] = {-INFINITY, -INFINITY}, l_run[2] = {0.f, 0.f};
    f32x4 oe[2][4];
#pragma unroll
    for (int ii = 0; ii < 2; ++ii)
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) oe[ii][ct] = f32x4{0.f, 0.f, 0.f, 0.f};

    // A row-step in two halves, software-pipelined by hand (an in-order wave overlaps nothing by itself):
    //   front(rs): pair tile -> bias orientation through LDS, 16 bias MFMAs in four independent chains   (needs only the pair tile:
    //                                                                              runs a row-step early, across the step barrier too)
    //   back(rs):  + S, running max / sum, P~ -> LDS, rescale, 16 o_e MFMAs, refill  (needs S of the tile: after the step barrier)
    // No lane-divergent branch inside (they end a scheduling region): lanes l15 >= 8 and the four quarters store duplicates.
    f32x4 acc_n;
    auto front = [&](int rs) {
      const int sl = rs % ERING;
      if constexpr (PLANES) return;  // the bias product is six MFMAs straight from the loaded registers: done in back(), no hand pipelining
      float* t_ = escr + 4 * q * ELD + 4 * l15;  // write [key 4 q + r][channel chunk l15], read [key l15][channels 16 sg + 4 q ..]
#pragma unroll
      for (int r = 0; r < 4; ++r) *reinterpret_cast<f32x4*>(t_ + r * ELD) = ev[sl][r];
      // (128-register budget: one 16-channel group of the tile and of Wb at a time, two accumulator chains)
      f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acc2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int sg = 0; sg < 4; ++sg) {
        const f32x4 ea = *reinterpret_cast<const f32x4*>(escr + l15 * ELD + 16 * sg + 4 * q);  // e[i][key l15][16 sg + 4 q + s]
        const f32x4 wbv = *reinterpret_cast<const f32x4*>(lds + L_WB + (sg * 64 + lane) * 4);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          if (sg & 1) acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(ea[s], wbv[s], acc2, 0, 0, 0);
          else acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ea[s], wbv[s], acc, 0, 0, 0);
        }
      }
      acc_n = acc + acc2;  // D: column = head l15, rows = keys 4 q + r of this tile
    };
    // Reductions over the four quarters (lanes l, l^16, l^32, l^48) with the gfx950 row swaps instead of LDS round trips:
    // v_permlane32_swap exchanges lanes 32-63 of its first register with lanes 0-31 of its second, v_permlane16_swap the odd rows
    // of the first with the even rows of the second; fed two COPIES of x they leave (x of the lower half / even row, x of the upper
    // half / odd row) in every lane.  Written in assembly: the builtin, given the same value twice, is folded to "both results = its
    // first" by hipcc (ROCm 7.2; tools/permlane_probe.hip shows it), which silently drops the other half.  s_nop 1 = the two wait
    // states a VALU write of an operand needs before the swap reads it (the compiler pads nothing inside an asm statement).
    // Not volatile: a pure function of its operands, free to be scheduled.
    struct F2 { float a, b; };
    auto swap32 = [](float x) {
      F2 r{x, x};
      asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(r.a), "+v"(r.b));
      return r;
    };
    auto swap16 = [](float x) {
      F2 r{x, x};
      asm("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(r.a), "+v"(r.b));
      return r;
    };
    auto xq_max = [&](float x) {
      F2 r = swap32(x);
      r = swap16(fmaxf(r.a, r.b));
      return fmaxf(r.a, r.b);
    };
    auto xq_sum = [&](float x) {
      F2 r = swap32(x);
      r = swap16(r.a + r.b);
      return r.a + r.b;
    };
    auto back = [&](int rs, const f32x4 acc) {
      const int jt = rs >> 1, ii = rs & 1, sl = rs % ERING, il = 2 * wv + ii;
      const float* Sk = lds + L_S + (jt & 1) * RING;
      float* Pk = lds + L_P + (jt & 1) * RING;
      float* Ak = lds + L_AL + (jt & 1) * (AH * TI);
      f32x4 accb = acc;
      if constexpr (PLANES) {
        // bias[key 4 q + r][head l15] = sum_c e w on the f16 matrix cores: A fragments are the loaded registers themselves (lane = key
        // l15, channels 32 ks + 8 q ..), B fragments from LDS one plane at a time (128-register budget); h2 w1 + h1 w1 + h1 w2
        f32x4 b0 = {0.f, 0.f, 0.f, 0.f}, b1 = {0.f, 0.f, 0.f, 0.f};
        const f16x8 a10 = __builtin_bit_cast(f16x8, ev[sl][0]), a11 = __builtin_bit_cast(f16x8, ev[sl][1]);
        {
          const f16x8 w10 = *reinterpret_cast<const f16x8*>(lds + L_WB + (0 * 64 + lane) * 4), w11 = *reinterpret_cast<const f16x8*>(lds + L_WB + (1 * 64 + lane) * 4);
          b0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, ev[sl][2]), w10, b0, 0, 0, 0);
          b1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, ev[sl][3]), w11, b1, 0, 0, 0);
          b0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a10, w10, b0, 0, 0, 0);
          b1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a11, w11, b1, 0, 0, 0);
        }
        {
          const f16x8 w20 = *reinterpret_cast<const f16x8*>(lds + L_WB + (2 * 64 + lane) * 4), w21 = *reinterpret_cast<const f16x8*>(lds + L_WB + (3 * 64 + lane) * 4);
          b0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a10, w20, b0, 0, 0, 0);
          b1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a11, w21, b1, 0, 0, 0);
        }
        accb = b0 + b1;
      }
      const f32x4 sv = *reinterpret_cast<const f32x4*>(Sk + il * RS + hl * HS + 4 * q);
      float s_[4], mt = -INFINITY;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        s_[r] = sv[r] + (bscale * inv_s2[ii]) * accb[r];
        mt = fmaxf(mt, s_[r]);
      }
#ifdef FLASH_ABL_NOSOFTMAX
      const float m_new = mt, alpha = 0.5f;
      f32x4 pv = {s_[0], s_[1], s_[2], s_[3]};
      float ls = s_[0];
#else
      mt = xq_max(mt);
      const float m_new = fmaxf(m_run[ii], mt);
      const float alpha = FAST_EXP(m_run[ii] - m_new);  // 0 on the first tile (m_run = -inf)
      f32x4 pv;
      float ls = 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        pv[r] = FAST_EXP(s_[r] - m_new);
        ls += pv[r];
      }
#endif
      *reinterpret_cast<f32x4*>(Pk + il * RS + hl * HS + 4 * q) = pv;  // (lanes l15 and l15 + 8 store the same values)
      Ak[hl * TI + il] = alpha;                                         // (so do the four quarters)
      // o_e[head n = l15][channel 4 m + ct] += sum_keys e[key][4 m + ct] P~[n][key]:  A = e (m = l15, k = q), B = P~ (n = l15, k = q)
#pragma unroll
      for (int ct = 0; ct < 4; ++ct)
#pragma unroll
        for (int r = 0; r < 4; ++r) oe[ii][ct][r] *= alpha;
      if constexpr (PLANES) {
        // o_e[channel 16 ct + 4 q + r][head l15] += e^T P~ on the f16 matrix cores (16x16x16: 4 keys per lane).  A: the tile goes as
        // loaded (lane = key, 16 bytes of channels) into the per-wave image and comes back transposed (lane i of a 16-lane group =
        // channel i of the 4-key block 4 q ..); B: this lane's own four probabilities, x 256, as two fp16 planes
        f16x4 p1, p2;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float x = 256.0f * pv[r];
          const _Float16 hh = static_cast<_Float16>(x);
          p1[r] = hh;
          p2[r] = static_cast<_Float16>(x - static_cast<float>(hh));
        }
#pragma unroll
        for (int pl = 0; pl < 2; ++pl)
#pragma unroll
          for (int ks = 0; ks < 2; ++ks) *reinterpret_cast<f32x4*>(trt + pl * 2048 + (wr_off ^ (64 * ks))) = ev[sl][2 * pl + ks];
        MEM_FENCE();  // the image is complete before the transposed reads (LDS operations of a wave complete in issue order)
        auto tr4 = [&](int pl, int ct) {
          const s16x4 v4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(trt + pl * 2048 + (rd_off ^ (32 * ct))));
          return __builtin_bit_cast(f16x4, v4);
        };
        {  // plane 2 (the small one) first: h2 p1
          f16x4 a2[4];
#pragma unroll
          for (int ct = 0; ct < 4; ++ct) a2[ct] = tr4(1, ct);
#pragma unroll
          for (int ct = 0; ct < 4; ++ct) oe[ii][ct] = __builtin_amdgcn_mfma_f32_16x16x16f16(a2[ct], p1, oe[ii][ct], 0, 0, 0);
        }
        {  // h1 p2, h1 p1
          f16x4 a1[4];
#pragma unroll
          for (int ct = 0; ct < 4; ++ct) a1[ct] = tr4(0, ct);
#pragma unroll
          for (int ct = 0; ct < 4; ++ct) oe[ii][ct] = __builtin_amdgcn_mfma_f32_16x16x16f16(a1[ct], p2, oe[ii][ct], 0, 0, 0);
#pragma unroll
          for (int ct = 0; ct < 4; ++ct) oe[ii][ct] = __builtin_amdgcn_mfma_f32_16x16x16f16(a1[ct], p1, oe[ii][ct], 0, 0, 0);
        }
        MEM_FENCE();  // ... and read before the next row-step overwrites it
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int ct = 0; ct < 4; ++ct) oe[ii][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(ev[sl][r][ct], pv[r], oe[ii][ct], 0, 0, 0);
      }
#ifndef FLASH_ABL_NOSOFTMAX
      ls = xq_sum(ls);
#endif
      l_run[ii] = l_run[ii] * alpha + ls;
      m_run[ii] = m_new;
      asm("" : "+v"(l_run[ii]));  // keep the running sum a running sum: the scheduler otherwise defers the whole chain to the end of
                                  // the kernel and spills every tile's (alpha, partial sum) pair until then
#ifdef FLASH_ABL_NOLOAD
      if (false) {
#else
      if (rs + ERING < NRS) {  // the slot is free: request the row-tile ERING row-steps ahead
#endif
        load_rs(rs + ERING);
        MEM_FENCE();
      }
      if (jt == NT - 1) {  // last key tile of this row: normalise and store o_e, publish 1 / l for the head products
        const float inv = 1.0f / l_run[ii];
        int le = lane;  // opaque copy: otherwise the store addresses are computed at the top of the kernel and spilled
        asm volatile("" : "+v"(le));
        const int he = le & 7, qe = le >> 4;
        lds[L_LI + he * TI + il] = inv;
        if ((le & 15) < 8) {
          if constexpr (PLANES) {  // D rows 4 q + r <-> channels 16 ct + 4 q + r
            float* fo = feat + (prow0 + i0 + il) * AF + FOFF_OE + he * AC + 4 * qe;
            const float sc = inv * oscale * inv_s2[ii];
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) {
              const f32x4 v = {oe[ii][ct][0] * sc, oe[ii][ct][1] * sc, oe[ii][ct][2] * sc, oe[ii][ct][3] * sc};
              *reinterpret_cast<f32x4*>(fo + 16 * ct) = v;
            }
          } else {
            float* fo = feat + (prow0 + i0 + il) * AF + FOFF_OE + he * AC + 16 * qe;  // D rows m = 4 q + r' <-> channels 16 q + 4 r' + ct
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const f32x4 v = {oe[ii][0][r] * inv, oe[ii][1][r] * inv, oe[ii][2][r] * inv, oe[ii][3][r] * inv};
              *reinterpret_cast<f32x4*>(fo + 4 * r) = v;
            }
          }
        }
      }
    };
    MEM_FENCE();
    __syncthreads();  // prologue barrier
    stamp(1);
    front(0);
#pragma unroll
    for (int k = 0; k < NT + 2; ++k) {
      if (k >= 1 && k <= NT) {  // ---- B(k - 1): this wave's two rows
#pragma unroll
        for (int ii = 0; ii < 2; ++ii) {
          const int rs = 2 * (k - 1) + ii;
          const f32x4 a0 = acc_n;
#ifndef FLASH_ABL_NOB
          if (rs + 1 < NRS) front(rs + 1);
          back(rs, a0);
          if constexpr (PLANES) __builtin_amdgcn_sched_barrier(0);  // one row-step's transients at a time (128 registers)
#endif
        }
      }
      __syncthreads();
      if (k == 0) stamp(2);
      if (k == 1) stamp(3);
      if (k == 4) stamp(4);
      if (k == NT) stamp(6);
      if (k == NT + 1) stamp(7);
    }
  } else {
    const int wv = wv_role - 8;  // head of this A / C wave
    const int hl = l15 & 7;
    // ======================================================================================= A / C: head h = wv
    const int h = wv;
    float* kscr = lds + L_HK + wv * KT;
    // line-shaped loads of one key tile (16 keys) of head h: k_s 16 x 128 B (8 lanes per key), k_pts 16 x 96 B (6 lanes per key)
    const int g0 = lane, g1 = lane < 32 ? lane + 64 : 95;  // k_pts chunk ids (0..95): key = id / 6, chunk = id % 6; lanes >= 32 repeat
                                                           // chunk 95 (same data to the same LDS address: no divergent branch)
    const float* krow = proj + prow0 * ANP;
    const int ks_off = (lane >> 3) * ANP + OFF_KS + h * ADS + 4 * (lane & 7);  // + 8 ANP for keys 8..15, + 16 jt ANP
    const int gk_off0 = (g0 / 6) * ANP + OFF_GK + h * 24 + 4 * (g0 % 6), gk_off1 = (g1 / 6) * ANP + OFF_GK + h * 24 + 4 * (g1 % 6);
    const int ks_dst = (lane >> 3) * KLD + 4 * (lane & 7);
    const int gk_dst0 = 16 * KLD + (g0 / 6) * GLD + 4 * (g0 % 6), gk_dst1 = 16 * KLD + (g1 / 6) * GLD + 4 * (g1 % 6);
    f32x4 st[4];
    auto load_keys = [&](int jt) {
      const float* base = krow + jt * (16 * ANP);  // wave-uniform
      st[0] = *reinterpret_cast<const f32x4*>(base + ks_off);
      st[1] = *reinterpret_cast<const f32x4*>(base + ks_off + 8 * ANP);
      st[2] = *reinterpret_cast<const f32x4*>(base + gk_off0);
      st[3] = *reinterpret_cast<const f32x4*>(base + gk_off1);
    };
    auto stage_keys = [&]() {
      *reinterpret_cast<f32x4*>(kscr + ks_dst) = st[0];
      *reinterpret_cast<f32x4*>(kscr + ks_dst + 8 * KLD) = st[1];
      *reinterpret_cast<f32x4*>(kscr + gk_dst0) = st[2];
      *reinterpret_cast<f32x4*>(kscr + gk_dst1) = st[3];
    };
    // value side of one key tile, B operands of the P~ x V products: lane (n = l15, k = q), key 16 jt + 4 q + s.
    // Point sums: columns 0..7 of one tile hold x of the 8 points, columns 8..15 y (z in a second tile), so 16 MFMAs per tile, not 20.
    float2 vs[4];
    float gxy[4], gz[4];
    const int vs_off = (4 * q) * ANP + OFF_VS + h * ADS + 2 * l15;
    const int gv_off = (4 * q) * ANP + OFF_GV + h * 24 + 3 * hl;  // + (16 jt + s) ANP
    auto load_vals = [&](int jt) {
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const float* pu = krow + (jt * 16 + s) * ANP;  // wave-uniform
        vs[s] = *reinterpret_cast<const float2*>(pu + vs_off);
        gxy[s] = pu[gv_off + (l15 >> 3)];
        gz[s] = pu[gv_off + 2];
      }
    };
    load_keys(0);
    const float coef_p = -0.5f * 0.16666666666666666f * gamma[h];  // -1/2 (4.5*8)^-1/2 gamma_h  (:372, :431-436)
    f32x4 qa[2];
    {
      const float* qrow = proj + (prow0 + i0 + l15) * ANP + OFF_QS + h * ADS + 4 * q;  // A operand: q_s rows i0 + l15, k = 16 sg + 4 q + s
      qa[0] = *reinterpret_cast<const f32x4*>(qrow);
      qa[1] = *reinterpret_cast<const f32x4*>(qrow + 16);
      // query points of the 16 rows of this head -> LDS [h][row][24] (96 chunks of 16 bytes)
      float* gdst = lds + L_GQ + h * (TI * 24);
      const float* gsrc = proj + (prow0 + i0) * ANP + OFF_GQ + h * 24;
      *reinterpret_cast<f32x4*>(gdst + (g0 / 6) * 24 + 4 * (g0 % 6)) = *reinterpret_cast<const f32x4*>(gsrc + (g0 / 6) * ANP + 4 * (g0 % 6));
      *reinterpret_cast<f32x4*>(gdst + (g1 / 6) * 24 + 4 * (g1 % 6)) = *reinterpret_cast<const f32x4*>(gsrc + (g1 / 6) * ANP + 4 * (g1 % 6));
    }
    f32x4 os[2], ogxy = {0.f, 0.f, 0.f, 0.f}, ogz = {0.f, 0.f, 0.f, 0.f};
    os[0] = os[1] = f32x4{0.f, 0.f, 0.f, 0.f};

    MEM_FENCE();
    __syncthreads();  // prologue barrier
    stamp(1);
#pragma unroll
    for (int k = 0; k < NT + 2; ++k) {
      if (k < NT) {  // ---- keys of tile k: registers -> LDS (fragments are read back in A(k) below), then request tile k + 1
        stage_keys();
        if (k + 1 < NT) load_keys(k + 1);
        MEM_FENCE();
      }
#ifndef FLASH_ABL_NOC  // (FLASH_ABL_*: timing-only ablation builds, results are garbage)
      if (k >= 2) {  // ---- C(k - 2): acc = alpha . acc + P~ x V
        const int jt = k - 2;
        const float* Pk = lds + L_P + (jt & 1) * RING;
        const float* Ak = lds + L_AL + (jt & 1) * (AH * TI);
        const f32x4 pa = *reinterpret_cast<const f32x4*>(Pk + l15 * RS + h * HS + 4 * q);  // A: P~[row l15][key 4 q + s]
        const f32x4 al = *reinterpret_cast<const f32x4*>(Ak + h * TI + 4 * q);             // D rows 4 q + r
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          os[0][r] *= al[r];
          os[1][r] *= al[r];
          ogxy[r] *= al[r];
          ogz[r] *= al[r];
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          os[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(pa[s], vs[s].x, os[0], 0, 0, 0);
          os[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(pa[s], vs[s].y, os[1], 0, 0, 0);
          ogxy = __builtin_amdgcn_mfma_f32_16x16x4f32(pa[s], gxy[s], ogxy, 0, 0, 0);
          ogz = __builtin_amdgcn_mfma_f32_16x16x4f32(pa[s], gz[s], ogz, 0, 0, 0);
        }
      }
      if (k >= 1 && k <= NT) {  // ---- values of tile k - 1, consumed by C(k - 1) in the next step
        load_vals(k - 1);
        MEM_FENCE();
      }
#endif
#ifndef FLASH_ABL_NOA
      if (k < NT) {  // ---- A(k): logits of key tile k for head h
        float* Sk = lds + L_S + (k & 1) * RING;
        const f32x4 kb0 = *reinterpret_cast<const f32x4*>(kscr + l15 * KLD + 4 * q);  // k_s[16 k + l15][16 sg + 4 q + s]
        const f32x4 kb1 = *reinterpret_cast<const f32x4*>(kscr + l15 * KLD + 16 + 4 * q);
        f32x4 gk[6];
#pragma unroll
        for (int cc = 0; cc < 6; ++cc) gk[cc] = *reinterpret_cast<const f32x4*>(kscr + 16 * KLD + l15 * GLD + 4 * cc);
        f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acc2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(qa[0][s], kb0[s], acc, 0, 0, 0);
          acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(qa[1][s], kb1[s], acc2, 0, 0, 0);
        }
        // acc[r] + acc2[r] = q_s[i0 + 4 q + r] . k_s[key 16 k + l15]
        const float* gql = lds + L_GQ + h * (TI * 24) + (4 * q) * 24;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          f32x2 d2v = {0.f, 0.f};  // packed fp32: two coordinates per instruction, two partial sums added at the end
#ifdef FLASH_ABL_NOPTS
          for (int cc = 0; cc < 0; ++cc) {
#else
#pragma unroll
          for (int cc = 0; cc < 6; ++cc) {
#endif
            const f32x4 gq = *reinterpret_cast<const f32x4*>(gql + r * 24 + 4 * cc);  // the 16 lanes of a quarter share the address
            f32x2 dlo, dhi;  // packed subtract spelled in assembly: the compiler splits a vector fsub into two v_sub_f32
            asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]"
                : "=v"(dlo)
                : "v"(__builtin_shufflevector(gq, gq, 0, 1)), "v"(__builtin_shufflevector(gk[cc], gk[cc], 0, 1)));
            asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]"
                : "=v"(dhi)
                : "v"(__builtin_shufflevector(gq, gq, 2, 3)), "v"(__builtin_shufflevector(gk[cc], gk[cc], 2, 3)));
            d2v = __builtin_elementwise_fma(dlo, dlo, d2v);
            d2v = __builtin_elementwise_fma(dhi, dhi, d2v);
          }
          const float d2 = d2v[0] + d2v[1];
          Sk[(4 * q + r) * RS + h * HS + l15] = scale_t * ((acc[r] + acc2[r]) * scale_s + coef_p * d2);
        }
      }
#endif
      __syncthreads();
      if (k == 0) stamp(2);
      if (k == 1) stamp(3);
      if (k == 4) stamp(4);
      if (k == NT) stamp(6);
      if (k == NT + 1) stamp(7);
    }
    // ---- epilogue of head h: D rows i = 4 q + r, column n = l15; normalise, global -> local frames, norms -> feature row
    {
      const f32x4 inv4 = *reinterpret_cast<const f32x4*>(lds + L_LI + h * TI + 4 * q);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int il = 4 * q + r;
        const int64_t row = prow0 + i0 + il;
        const float inv = inv4[r];
        float* fr = feat + row * AF;
        *reinterpret_cast<float2*>(fr + FOFF_OS + h * ADS + 2 * l15) = make_float2(os[0][r] * inv, os[1][r] * inv);
        const float gy_ = __shfl_xor(ogxy[r], 8);  // lanes 0..7 hold x of point l15, lanes 8..15 y of point l15 - 8
        if (l15 < 8) {
          const float* Rr = R + row * 9;
          const float* tr = t + row * 3;
          const float dx = ogxy[r] * inv - tr[0], dy = gy_ * inv - tr[1], dz = ogz[r] * inv - tr[2];
          const float lx = dx * Rr[0] + dy * Rr[1] + dz * Rr[2];  // (p - t) R^T   (diffab_pytorch.py:336)
          const float ly = dx * Rr[3] + dy * Rr[4] + dz * Rr[5];
          const float lz = dx * Rr[6] + dy * Rr[7] + dz * Rr[8];
          float* fo = fr + FOFF_OL + h * 24 + 3 * l15;
          fo[0] = lx; fo[1] = ly; fo[2] = lz;
          fr[FOFF_ON + h * AP + l15] = sqrtf(lx * lx + ly * ly + lz * lz);
        }
      }
    }
  }
  stamp(5);
}

bool attention_flash_supported(const diffab_dims* d) {
  return d->D == 128 && d->C == AC && d->H == AH && d->DS == ADS && d->PQ == AP && d->PV == AP && (d->K == 128 || d->K == 64);
}

int launch_attention_flash(const diffab_dims* d, const float* proj, const float* e, const float* R, const float* t, const float* Wb,
                           const float* gamma, float* feat, unsigned long long* stamps, hipStream_t st, const float* pair_planes) {
  DIFFAB_REQUIRE(attention_flash_supported(d), DIFFAB_ERR_UNSUPPORTED, "attention_flash: K must be 64 or 128 at the benchmark geometry");
  const dim3 grid(d->B * (d->K / TI));
#define FLASH_LAUNCH(NT_)                                                                                                         \
  do {                                                                                                                            \
    DIFFAB_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(ipa_attn_flash_kernel<NT_>),                               \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(kFlashLdsBytes)));          \
    timer_begin(st);                                                                                                              \
    hipLaunchKernelGGL((ipa_attn_flash_kernel<NT_>), grid, dim3(512), kFlashLdsBytes, st, proj, e, R, t, Wb, gamma, feat, d->B,   \
                       stamps);                                                                                                   \
    timer_end(st);                                                                                                                \
  } while (0)
  static const int waves = env_int("DIFFAB_FLASH_WAVES", 16);
#define FLASH16_LAUNCH(NT_)                                                                                                       \
  do {                                                                                                                            \
    DIFFAB_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(ipa_attn_flash16_kernel<NT_>),                             \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(kFlashLdsBytes)));          \
    timer_begin(st);                                                                                                              \
    hipLaunchKernelGGL((ipa_attn_flash16_kernel<NT_>), grid, dim3(1024), kFlashLdsBytes, st, proj, e, R, t, Wb, gamma, feat, d->B, \
                       stamps);                                                                                                   \
    timer_end(st);                                                                                                                \
  } while (0)
#define FLASH16P_LAUNCH(NT_)                                                                                                      \
  do {                                                                                                                            \
    DIFFAB_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(ipa_attn_flash16_kernel<NT_, true>),                       \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(kFlashLdsBytes)));          \
    timer_begin(st);                                                                                                              \
    hipLaunchKernelGGL((ipa_attn_flash16_kernel<NT_, true>), grid, dim3(1024), kFlashLdsBytes, st, proj, pair_planes + 64, R, t, Wb, \
                       gamma, feat, d->B, stamps, pair_row_scales(d, pair_planes));                                                               \
    timer_end(st);                                                                                                                \
  } while (0)
  if (pair_planes != nullptr) {  // launch_pair_split() output: producers on the f16 matrix cores (16-wave form only)
    if (d->K == 128) FLASH16P_LAUNCH(8);
    else FLASH16P_LAUNCH(4);
  } else if (waves == 16) {
    if (d->K == 128) FLASH16_LAUNCH(8);
    else FLASH16_LAUNCH(4);
  } else {
    if (d->K == 128) FLASH_LAUNCH(8);
    else FLASH_LAUNCH(4);
  }
#undef FLASH16P_LAUNCH
#undef FLASH16_LAUNCH
#undef FLASH_LAUNCH
  DIFFAB_LAUNCH_CHECK();
  return DIFFAB_OK;
}

}  // namespace diffab
