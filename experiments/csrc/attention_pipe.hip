// attention_pipe.hip - the fused IPA attention as a key-tile pipeline of sixteen waves whose three kinds of work all run on the
// low-precision matrix cores' side of the SIMD (K = 64 / 128, benchmark geometry).
// Reference: InvariantPointAttentionLayer.forward, diffab_pytorch.py:416-457 (logits :416-439, softmax :443, sums :445-457).
//
// Structure (the sixteen-wave form of attention_flash.hip, round 2): a work-group owns (patch, 16 query rows); waves 0-7 are
// PRODUCERS (two query rows each: the pair stream - non-temporal, a register ring requested from the first instruction - pair bias,
// online softmax, attention-weighted pair sum), waves 8-15 are CONSUMERS (one head each: logits of key tile k, P x V of key tile
// k - 2); they meet once per key tile:   step k:  C(k-2) | B(k-1) | A(k) | barrier.   The pair stream never stops, no wave holds
// more than a few 4 KiB pair tiles, LDS holds two key tiles of logits and probabilities instead of the whole image.
//
// What changed against round 2 (where this structure measured 0.333 ms against 0.325 for the three-phase kernel, both bound by the
// fp32 pipe: f32 MFMA and VALU share one issue resource on gfx950): the consumers' logits no longer touch that pipe.  A(k) is ONE
// 64-slot split-precision dot product per (head, query, key) on the bf16 matrix cores - operand planes written by
// proj_planes_b6_kernel (proj_planes.hip: scalar q.k and the point-distance bilinear form in the same slots, fragment order, linear
// 1 KiB loads, no LDS staging) - plus 8 coef |t_i - t_j|^2 from a 16 x K table of direct differences built once per work-group:
// 12 MFMAs of 16 cycles and ~10 VALU instructions per key tile instead of 8 f32 MFMAs of 32 cycles, 96 packed VALU instructions and
// an LDS round trip.  The producers were already on the f16 matrix cores (two-plane fp16 image of the pair embedding, round 2).
// P x V stays on f32 MFMA here.
#include <cstdlib>

#include "common.h"
#include "denoiser_internal.h"

namespace diffab {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
#define MEM_FENCE() asm volatile("" ::: "memory")

namespace {
constexpr int AH = 8, ADS = 32, AP = 8, AC = 64;
constexpr int ANP = 1344, AF = 1024;  // columns of the fp32 projection buffer (value side used here) / of the feature rows
constexpr int OFF_VS = 512, OFF_GV = 1152;
constexpr int FOFF_OS = 0, FOFF_OE = 256, FOFF_OL = 768, FOFF_ON = 960;
constexpr int TI = 16;  // query rows per work-group

// ---- LDS map (floats)
constexpr int HS = 20;                 // head stride inside a ring row: 16 keys + 4 (16-byte aligned, spreads banks)
constexpr int RS = AH * HS + 4;        // ring row stride 164: the 4-byte S stores are 2-way at worst (free)
constexpr int RING = TI * RS;          // one key tile of S or P~: 2624 floats
constexpr int D2LD = 20;               // distance-table stride per key (16 rows + 4): conflict-free b128
constexpr int L_S = 0;                        // S ring       [2][TI][RS]
constexpr int L_P = L_S + 2 * RING;           // P~ ring      [2][TI][RS]
constexpr int L_AL = L_P + 2 * RING;          // alpha ring   [2][AH][TI]
constexpr int L_LI = L_AL + 2 * AH * TI;      // 1 / l        [AH][TI]
constexpr int L_D2 = L_LI + AH * TI;          // |t_i - t_j|^2 [128 keys][D2LD]
constexpr int L_SE = L_D2 + 128 * D2LD;       // per-producer [2 planes][16 keys][128 bytes] transposition image (4 KiB of a 4.5 KiB slot)
constexpr int L_WB = L_SE + 8 * 1152;         // bias-product B fragments, fp16 planes [2 p + ks][64 lanes] x 16 bytes
constexpr int L_END = L_WB + 4 * 64 * 4;
constexpr size_t kPipeLdsBytes = static_cast<size_t>(L_END) * sizeof(float);  // 94 720 B
#ifndef DIFFAB_PIPE_ERING
#define DIFFAB_PIPE_ERING 3
#endif
constexpr int ERING = DIFFAB_PIPE_ERING;  // pair row-tiles per producer in registers (ERING - 1 in flight beside the one consumed)
}  // namespace

template <int NT>
__global__ __launch_bounds__(1024) void ipa_attn_pipe_kernel(const float* __restrict__ proj, const float* __restrict__ e,
                                                             const float* __restrict__ R, const float* __restrict__ t,
                                                             const float* __restrict__ Wb, const float* __restrict__ gamma,
                                                             float* __restrict__ feat, int B, unsigned long long* __restrict__ stamps,
                                                             const float* __restrict__ esc, const f32x4* __restrict__ qkp,
                                                             int64_t kside_off) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int K = NT * 16;
  constexpr int ntile = K / TI;
  constexpr int NRS = 2 * NT;  // row-steps of a producer: (key tile jt, row ii) = rs = 2 jt + ii
  // XCD-aware map (blocks b and b + 8 share an XCD): all row tiles of a patch on one XCD, so the key / value side is an L2 hit for 7 of 8
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
  const int wv_role = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int64_t prow0 = static_cast<int64_t>(b) * K;
  const float scale_t = 0.57735026918962576f;  // 3^-1/2   (diffab_pytorch.py:387, :439)
  auto stamp = [&](int k) {  // diagnostics (stamps == nullptr in production)
    if (stamps != nullptr) {
      __builtin_amdgcn_sched_barrier(0);
      const unsigned long long tnow = __builtin_amdgcn_s_memtime();
      if (lane == 0 && (wv_role < 4 || wv_role >= 12)) stamps[(static_cast<size_t>(blockIdx.x) * 8 + (wv_role & 7)) * 8 + k] = tnow;
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  stamp(0);

  if (wv_role < 8) {
    const int wv = wv_role;  // row pair of this producer
    // ======================================================================================= B: pair stream of rows 2 wv, 2 wv + 1
    const int hl = l15 & 7;  // lanes l15 >= 8 duplicate head l15 - 8 (their MFMA columns are the padding half of the 16-wide tile)
    const float* ebase = e + ((prow0 + i0 + 2 * wv) * K) * AC;  // + ii K AC + jt 16 AC (uniform)
    f32x4 ev[ERING][4];
    auto load_rs = [&](int rs) {  // -> ring slot rs % ERING: four 1 KiB blocks (plane p, k-step ks) of the tile, lane order: ev[slot][2 p + ks]
      const float* eu = ebase + (rs & 1) * (K * AC) + (rs >> 1) * (16 * AC);
#pragma unroll
      for (int r = 0; r < 4; ++r) ev[rs % ERING][r] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(eu + r * 256 + lane * 4));
    };
#pragma unroll
    for (int rs = 0; rs < ERING && rs < NRS; ++rs) load_rs(rs);  // the stream starts with the kernel
    MEM_FENCE();
    // bias B fragments as two fp16 planes [plane][ks]: lane (head l15, channels 32 ks + 8 q ..), scaled by sw into [128, 256); kept in
    // LDS (written by wave 0, the same for every wave) - the producers have 128 registers
    float bscale, oscale;
    {
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
    }
    const float inv_s2[2] = {esc[2 * (prow0 + i0 + 2 * wv) + 1], esc[2 * (prow0 + i0 + 2 * wv + 1) + 1]};  // 1 / s_i of the two pair rows
    // per-wave [2 planes][16 keys][128 bytes] image of the current tile; 8-byte unit u of row r at u ^ (4 ((r >> 1) & 3)): the
    // transposed reads of a 32-lane half touch 32 distinct bank pairs
    char* trt = reinterpret_cast<char*>(lds + L_SE + wv * 1152);
    const int wr_off = l15 * 128 + 8 * ((2 * q) ^ (4 * ((l15 >> 1) & 3)));      // ^ 64 ks: unit 8 ks + 2 q of row (key) l15
    const int rrow = 4 * q + (l15 >> 2);
    const int rd_off = rrow * 128 + 8 * ((l15 & 3) ^ (4 * ((rrow >> 1) & 3)));  // ^ 32 ct: unit 4 ct + (l15 & 3) of row rrow
    float m_run[2] = {-INFINITY, -INFINITY}, l_run[2] = {0.f, 0.f};
    f32x4 oe[2][4];
#pragma unroll
    for (int ii = 0; ii < 2; ++ii)
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) oe[ii][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
    // Reductions over the four quarters (lanes l, l^16, l^32, l^48) with the gfx950 row swaps instead of LDS round trips (written in
    // assembly: the builtin, given the same value twice, is folded by hipcc - tools/permlane_probe.hip; s_nop 1 = the wait states a
    // VALU write needs before the swap reads it)
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
    auto row_step = [&](int rs) {
      const int jt = rs >> 1, ii = rs & 1, sl = rs % ERING, il = 2 * wv + ii;
      const float* Sk = lds + L_S + (jt & 1) * RING;
      float* Pk = lds + L_P + (jt & 1) * RING;
      float* Ak = lds + L_AL + (jt & 1) * (AH * TI);
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
      const f32x4 accb = b0 + b1;
      const f32x4 sv = *reinterpret_cast<const f32x4*>(Sk + il * RS + hl * HS + 4 * q);
      float s_[4], mt = -INFINITY;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        s_[r] = sv[r] + (bscale * inv_s2[ii]) * accb[r];
        mt = fmaxf(mt, s_[r]);
      }
      mt = xq_max(mt);
      const float m_new = fmaxf(m_run[ii], mt);
      const float alpha = __expf(m_run[ii] - m_new);  // 0 on the first tile (m_run = -inf)
      f32x4 pv;
      float ls = 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        pv[r] = __expf(s_[r] - m_new);
        ls += pv[r];
      }
      *reinterpret_cast<f32x4*>(Pk + il * RS + hl * HS + 4 * q) = pv;  // (lanes l15 and l15 + 8 store the same values)
      Ak[hl * TI + il] = alpha;                                         // (so do the four quarters)
#pragma unroll
      for (int ct = 0; ct < 4; ++ct)
#pragma unroll
        for (int r = 0; r < 4; ++r) oe[ii][ct][r] *= alpha;
      // o_e[channel 16 ct + 4 q + r][head l15] += e^T P~ on the f16 matrix cores (16x16x16: 4 keys per lane).  A: the tile goes as
      // loaded (lane = key, 16 bytes of channels) into the per-wave image and comes back transposed; B: this lane's own four
      // probabilities, x 256, as two fp16 planes
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
      ls = xq_sum(ls);
      l_run[ii] = l_run[ii] * alpha + ls;
      m_run[ii] = m_new;
      asm("" : "+v"(l_run[ii]));  // keep the running sum a running sum (the scheduler otherwise defers the chain and spills its inputs)
      if (rs + ERING < NRS) {  // the slot is free: request the row-tile ERING row-steps ahead
        load_rs(rs + ERING);
        MEM_FENCE();
      }
      if (jt == NT - 1) {  // last key tile of this row: normalise and store o_e, publish 1 / l for the head products
        const float inv = 1.0f / l_run[ii];
        int le = lane;  // opaque copy: otherwise the store addresses are computed at the top of the kernel and spilled
        asm volatile("" : "+v"(le));
        const int he = le & 7, qe = le >> 4;
        lds[L_LI + he * TI + il] = inv;
        if ((le & 15) < 8) {  // D rows 4 q + r <-> channels 16 ct + 4 q + r
          float* fo = feat + (prow0 + i0 + il) * AF + FOFF_OE + he * AC + 4 * qe;
          const float sc = inv * oscale * inv_s2[ii];
#pragma unroll
          for (int ct = 0; ct < 4; ++ct) {
            const f32x4 v = {oe[ii][ct][0] * sc, oe[ii][ct][1] * sc, oe[ii][ct][2] * sc, oe[ii][ct][3] * sc};
            *reinterpret_cast<f32x4*>(fo + 16 * ct) = v;
          }
        }
      }
    };
    MEM_FENCE();
    __syncthreads();  // prologue barrier
    stamp(1);
#pragma unroll
    for (int k = 0; k < NT + 2; ++k) {
      if (k >= 1 && k <= NT) {  // ---- B(k - 1): this wave's two rows
#pragma unroll
        for (int ii = 0; ii < 2; ++ii) {
          row_step(2 * (k - 1) + ii);
          __builtin_amdgcn_sched_barrier(0);  // one row-step's transients at a time (128 registers)
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
    // ======================================================================================= A / C: head h
    const int h = wv_role - 8;
    const int hl = l15 & 7;
    const float coef8 = -0.5f * 0.16666666666666666f * gamma[h] * 8.0f;  // 8 coef of head h: the |t_i - t_j|^2 term of all eight points
    // ---- |t_i - t_j|^2 of the 16 rows x K keys from direct differences: consumer h < NT writes key tile h of the table
    if (h < NT) {
      const float* tj = t + (prow0 + 16 * h + l15) * 3;
      const float tjx = tj[0], tjy = tj[1], tjz = tj[2];
      f32x4 dd;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float* ti = t + (prow0 + i0 + 4 * q + r) * 3;
        const float dx = ti[0] - tjx, dy = ti[1] - tjy, dz = ti[2] - tjz;
        dd[r] = (dx * dx + dy * dy) + dz * dz;
      }
      *reinterpret_cast<f32x4*>(lds + L_D2 + (16 * h + l15) * D2LD + 4 * q) = dd;
    }
    // ---- operand planes of the logits product: A = query side (rows i0 .. i0 + 15), B = key side, [k-step][plane], 1 KiB each
    const f32x4* qsrc = qkp + ((static_cast<int64_t>(b) * AH + h) * ntile + tile) * (6 * 64) + lane;
    const f32x4* ksrc = qkp + kside_off + ((static_cast<int64_t>(b) * AH + h) * ntile) * (6 * 64) + lane;
    f32x4 qa[6], kb[2][6];
#pragma unroll
    for (int u = 0; u < 6; ++u) qa[u] = qsrc[u * 64];
#pragma unroll
    for (int jt = 0; jt < 2 && jt < NT; ++jt)
#pragma unroll
      for (int u = 0; u < 6; ++u) kb[jt][u] = ksrc[(jt * 6 + u) * 64];
    // value side of one key tile, B operands of the P~ x V products: lane (n = l15, k = q), key 16 jt + 4 q + s.
    // Point sums: columns 0..7 of one tile hold x of the 8 points, columns 8..15 y (z in a second tile): 16 MFMAs per tile.
    const float* krow = proj + prow0 * ANP;
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
    f32x4 os[2], ogxy = {0.f, 0.f, 0.f, 0.f}, ogz = {0.f, 0.f, 0.f, 0.f};
    os[0] = os[1] = f32x4{0.f, 0.f, 0.f, 0.f};
    constexpr int TA[6] = {1, 2, 0, 1, 0, 0}, TB[6] = {1, 0, 2, 0, 1, 0};  // (mid,mid) (lo,hi) (hi,lo) (mid,hi) (hi,mid) (hi,hi)

    MEM_FENCE();
    __syncthreads();  // prologue barrier (the distance table and the bias planes are complete)
    stamp(1);
#pragma unroll
    for (int k = 0; k < NT + 2; ++k) {
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
      if (k < NT) {  // ---- A(k): logits of key tile k for head h
        float* Sk = lds + L_S + (k & 1) * RING;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
          for (int term = 0; term < 6; ++term)
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, qa[3 * ks + TA[term]]),
                                                          __builtin_bit_cast(bf16x8, kb[k & 1][3 * ks + TB[term]]), acc, 0, 0, 0);
        if (k + 2 < NT) {
#pragma unroll
          for (int u = 0; u < 6; ++u) kb[k & 1][u] = ksrc[((k + 2) * 6 + u) * 64];
          MEM_FENCE();
        }
        const f32x4 d2v = *reinterpret_cast<const f32x4*>(lds + L_D2 + (16 * k + l15) * D2LD + 4 * q);
        // acc[r] = ds^-1/2 q_s.k_s + coef (sum_p |gq_p - gk_p|^2 - 8 |t_i - t_j|^2) [- row terms], row i0 + 4 q + r, key 16 k + l15
#pragma unroll
        for (int r = 0; r < 4; ++r) Sk[(4 * q + r) * RS + h * HS + l15] = scale_t * (acc[r] + coef8 * d2v[r]);
      }
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

static unsigned long long* g_pipe_stamps = nullptr;
void set_attn_pipe_stamps(void* p) { g_pipe_stamps = static_cast<unsigned long long*>(p); }

bool attention_pipe_supported(const diffab_dims* d) {
  return d->D == 128 && d->C == AC && d->H == AH && d->DS == ADS && d->PQ == AP && d->PV == AP && (d->K == 128 || d->K == 64);
}

// proj: fp32 projection buffer (value side), pair_planes: launch_pair_split() output, qk_ops: proj_planes_b6 operand planes
int launch_attention_pipe(const diffab_dims* d, const float* proj, const float* pair_planes, const float* R, const float* t, const float* Wb,
                          const float* gamma, float* feat, const float* qk_ops, hipStream_t st) {
  DIFFAB_REQUIRE(attention_pipe_supported(d) && proj && pair_planes && qk_ops, DIFFAB_ERR_ARG, "attention_pipe: unsupported operands");
  const dim3 grid(d->B * (d->K / TI));
  const int64_t kside = static_cast<int64_t>(d->B) * d->K * (8 * 64 * 3 * 2 / 16);
#define PIPE_LAUNCH(NT_)                                                                                                              \
  do {                                                                                                                                \
    DIFFAB_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(ipa_attn_pipe_kernel<NT_>),                                    \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(kPipeLdsBytes)));               \
    timer_begin(st);                                                                                                                  \
    hipLaunchKernelGGL((ipa_attn_pipe_kernel<NT_>), grid, dim3(1024), kPipeLdsBytes, st, proj, pair_planes + 64, R, t, Wb, gamma,     \
                       feat, d->B, g_pipe_stamps, pair_row_scales(d, pair_planes), reinterpret_cast<const f32x4*>(qk_ops), kside);                    \
    timer_end(st);                                                                                                                    \
  } while (0)
  if (d->K == 128) PIPE_LAUNCH(8);
  else PIPE_LAUNCH(4);
#undef PIPE_LAUNCH
  DIFFAB_LAUNCH_CHECK();
  return DIFFAB_OK;
}

}  // namespace diffab
