// attention_b6.hip - the fused IPA attention kernel of the reverse sampler: persistent work-groups, logits on the bf16 matrix cores.
//
// Same three phases and the same arithmetic as ipa_attn_fast_kernel<NT, false, false, PLANES, B6L> (denoiser_fast.hip) - phase 1
// (wave = head): scalar + point-distance logits as ONE 64-slot split-precision dot product of the operand planes written by
// proj_planes_b6_kernel, plus 8 coef |t_i - t_j|^2 from a table of direct differences; phase 2 (wave = 2 query rows): pair bias,
// online softmax over 32-key steps and the attention-weighted pair sum on the f16 matrix cores from the two-plane fp16 image of
// the pair embedding; phase 3 (wave = head): P x V on f32 MFMA, global -> local frames, norms - but a work-group walks a LIST of
// 16-row tiles (all tiles of one patch at the benchmark size: grid = 256) instead of one.
//
// Why: the stamps of the one-tile kernel show every work-group waiting ~10 k cycles (of ~85 k) for its first operands - 192 KiB
// per CU of query / key planes that miss L2 (the query side always, the key side for the first of a patch's tiles), fetched at the
// ~20 B/clk a CU gets for misses - with nothing to do meanwhile; and tools/ablate.sh (nok / noe / nov) shows each phase bound by
// the bytes its CU can pull, not by instructions.  A persistent work-group requests the NEXT tile's phase-1 operands (query planes,
// the first three key tiles, the translations of the distance table) in the second half of phase 3, when its value loads have all
// been issued (vmcnt retires in order: nothing later in phase 3 waits behind them): they travel under phase 3's MFMAs and the
// barrier, and phase 1 starts with its operands in registers.  The key side of a patch stays hot in the XCD's L2 across its tiles.
// K = 64 / 128 (single key chunk); K = 192, 256, .. keep the one-tile kernel with its chunk loop.
// Reference math: InvariantPointAttentionLayer.forward, diffab_pytorch.py:416-457.
#include "common.h"
#include "denoiser_internal.h"

namespace diffab {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4_t __attribute__((ext_vector_type(4)));
typedef short s16x8_t __attribute__((ext_vector_type(8)));
#define MEM_FENCE() asm volatile("" ::: "memory")

namespace {
constexpr int PH = 8, PDS = 32, PC = 64;           // heads, scalar dims per head, pair channels
constexpr int PNP = 1344, PF = 1024;               // columns of the fp32 projection buffer / of the feature rows
constexpr int P_VS = 512, P_GV = 1152;             // v_s and global value points in the projection buffer
constexpr int PF_OS = 0, PF_OE = 256, PF_OL = 768, PF_ON = 960;
constexpr int PTI = 16;                            // query residues per tile
constexpr int P_SCR = 2304;                        // floats of LDS scratch per wave (8 KiB transposition image of phase 2; the distance table
                                                   // of phase 1 lies over the first waves' scratch)
}  // namespace

size_t attention_b6p_lds_bytes(int nt) {  // image | per-wave scratch | normalisers | bias-weight planes
  return (static_cast<size_t>(PTI) * (PH * (16 * nt + 8) + 8) + 8 * P_SCR + PTI * PH + 64 * 16) * sizeof(float);
}

template <int NT>
__global__ __launch_bounds__(512) void ipa_attn_b6p_kernel(const float* __restrict__ proj, const float* __restrict__ e,
                                                           const float* __restrict__ R, const float* __restrict__ t,
                                                           const float* __restrict__ Wb, const float* __restrict__ gamma,
                                                           float* __restrict__ feat, int B, int tiles_per_wg,
                                                           unsigned long long* __restrict__ stamps, const float* __restrict__ esc,
                                                           const f32x4* __restrict__ qkp, int64_t kside_off) {
  extern __shared__ __attribute__((aligned(16))) float S[];  // [16 rows][8 heads][K + 8] (+ 8 per row) logits -> probabilities
  constexpr int K = 16 * NT, NS = 4 * NT, HS = K + 8, IS = PH * HS + 8, D2LD = 20, RT = NT / 2, E_EARLY = 2, SD = 3;
  const int tid = threadIdx.x, lane0 = tid & 63, wv = tid >> 6;
  const int total = B * NT;  // 16-row tiles (K / 16 = NT per patch)
  const int first = blockIdx.x * tiles_per_wg;
  const int count = total - first < tiles_per_wg ? total - first : tiles_per_wg;
  float* const scr = S + PTI * IS + wv * P_SCR;
  float* const d2t = S + PTI * IS;                   // [K keys][20]: |t_i - t_j|^2 of the 16 rows (stride 20: conflict-free b128)
  float* const st_inv = S + PTI * IS + 8 * P_SCR;    // [16][8] 1 / softmax denominator
  const float scale_t = 0.57735026918962576f;        // 3^-1/2   (diffab_pytorch.py:387, :439)
  const int stamp_it = count > 1 ? 1 : 0;            // the iteration whose phase boundaries are stamped (steady state when there is one)
  auto stamp = [&](int it, int k) {
    if (stamps != nullptr && it == stamp_it) {
      __builtin_amdgcn_sched_barrier(0);
      const unsigned long long tnow = __builtin_amdgcn_s_memtime();
      if ((threadIdx.x & 63) == 0) stamps[(static_cast<size_t>(blockIdx.x) * 8 + (threadIdx.x >> 6)) * 8 + k] = tnow;
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  // ---- once per work-group: the bias weights of lane (head l15 & 7, channel group) as two fp16 planes scaled by a power of two,
  // parked in LDS (the same for every wave; 16 VGPRs that the loop-carried operands need) and re-read per tile in phase 2
  f32x4* const wp_lds = reinterpret_cast<f32x4*>(st_inv + PTI * PH);  // [plane][k-step][64 lanes] x 16 bytes
  float bscale, oscale;
  {
    f16x8 wp[2][2];
    const int l15 = lane0 & 15, qq = lane0 >> 4;
    f32x4 wv4[2][2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) wv4[ks][hf] = *reinterpret_cast<const f32x4*>(Wb + (l15 & 7) * PC + 32 * ks + 8 * qq + 4 * hf);
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
    bscale = scale_t * esc[1] * isw;        // logits: bias = (sum e s w sw) / (s sw)
    oscale = esc[1] * (1.0f / 256.0f);      // o_e: probabilities enter scaled by 256
    if (wv == 0) {
#pragma unroll
      for (int pl = 0; pl < 2; ++pl)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) wp_lds[(pl * 2 + ks) * 64 + lane0] = __builtin_bit_cast(f32x4, wp[pl][ks]);
    }
  }
  const float coef8 = -0.5f * 0.16666666666666666f * gamma[wv] * 8.0f;  // 8 coef of head wv: the |t_i - t_j|^2 term of all eight points

  // ---- phase-1 operands of a tile, loop-carried: requested for tile `it + 1` in the second half of tile `it`'s phase 3
  f32x4 qa[6], kb[SD][6];  // [k-step][plane] operand pieces: query side of the tile, key tiles 0 .. 2 of its patch, head wv
  float tj[3], ti[4][3];   // translations for the distance table: key 16 wv + l15, rows 4 q + r of the tile
  auto request_tile = [&](int gt) {
    const int lane = lane0, l15 = lane & 15, q = lane >> 4;
    const int b = gt / NT, tile = gt - b * NT;
    const int64_t prow0 = static_cast<int64_t>(b) * K;
    {  // unconditional (waves >= NT read key tile NT - 1 and discard it): a conditional load would keep the previous tile's values live
      const float* pj = t + (prow0 + 16 * (wv < NT ? wv : NT - 1) + l15) * 3;
      tj[0] = pj[0]; tj[1] = pj[1]; tj[2] = pj[2];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float* pi = t + (prow0 + tile * PTI + 4 * q + r) * 3;
        ti[r][0] = pi[0]; ti[r][1] = pi[1]; ti[r][2] = pi[2];
      }
    }
    const f32x4* qsrc = qkp + ((static_cast<int64_t>(b) * PH + wv) * NT + tile) * (6 * 64) + lane;
    const f32x4* ksrc = qkp + kside_off + ((static_cast<int64_t>(b) * PH + wv) * NT) * (6 * 64) + lane;
#pragma unroll
    for (int u = 0; u < 6; ++u) qa[u] = qsrc[u * 64];
#pragma unroll
    for (int jt = 0; jt < SD && jt < NT; ++jt)
#pragma unroll
      for (int u = 0; u < 6; ++u) kb[jt][u] = ksrc[(jt * 6 + u) * 64];
  };
  // |t_i - t_j|^2 of the 16 rows x K keys from direct differences: wave w < NT writes key tile w of the table
  auto write_d2 = [&]() {
    if (wv < NT) {
      const int l15 = lane0 & 15, q = lane0 >> 4;
      f32x4 dd;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float dx = ti[r][0] - tj[0], dy = ti[r][1] - tj[1], dz = ti[r][2] - tj[2];
        dd[r] = (dx * dx + dy * dy) + dz * dz;
      }
      *reinterpret_cast<f32x4*>(d2t + (16 * wv + l15) * D2LD + 4 * q) = dd;
    }
  };
  if (count <= 0) return;
  request_tile(first);
  MEM_FENCE();
  write_d2();
  __syncthreads();

#pragma unroll 1
  for (int it = 0; it < count; ++it) {
    const int gt = first + it;
    const int b = gt / NT, tile = gt - b * NT, i0 = tile * PTI;
    const int64_t prow0 = static_cast<int64_t>(b) * K;
    // Re-derive the lane coordinates from an opaque copy each iteration: otherwise every lane-constant address of the three phases is
    // hoisted out of the loop and stays live through phase 2
    int lane = lane0;
    asm volatile("" : "+v"(lane));
    const int l15 = lane & 15, q = lane >> 4;
    stamp(it, 0);
    // the two pair rows of this wave (phase 2): fp16 planes in the fragment order of the bias product, four 1 KiB blocks per key tile
    const f32x4* erow[2];
    erow[0] = reinterpret_cast<const f32x4*>(e + ((prow0 + i0 + 2 * wv) * K) * PC) + lane;
    erow[1] = erow[0] + K * PC / 4;
    f32x4 ev[2][NT][4];
    auto load_e_tile = [&](int ii, int jt) {
#pragma unroll
      for (int r = 0; r < 4; ++r) ev[ii][jt][r] = __builtin_nontemporal_load(erow[ii] + jt * 256 + r * 64);
    };
    // ---------------------------------------------------------------- phase 1: wave = head
    {
      const int h = wv;
      const f32x4* ksrc = qkp + kside_off + ((static_cast<int64_t>(b) * PH + h) * NT) * (6 * 64) + lane;
      constexpr int TA[6] = {1, 2, 0, 1, 0, 0}, TB[6] = {1, 0, 2, 0, 1, 0};  // (mid,mid) (lo,hi) (hi,lo) (mid,hi) (hi,mid) (hi,hi)
#pragma unroll
      for (int jt = 0; jt < NT; ++jt) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
          for (int term = 0; term < 6; ++term)
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, qa[3 * ks + TA[term]]),
                                                          __builtin_bit_cast(bf16x8, kb[jt % SD][3 * ks + TB[term]]), acc, 0, 0, 0);
        if (jt + SD < NT) {
#pragma unroll
          for (int u = 0; u < 6; ++u) kb[jt % SD][u] = ksrc[((jt + SD) * 6 + u) * 64];
        } else if (jt + E_EARLY >= NT) {
          load_e_tile(0, jt + E_EARLY - NT);  // key stream done: start phase 2's pair stream under this tile
        }
        MEM_FENCE();
        const f32x4 d2v = *reinterpret_cast<const f32x4*>(d2t + (16 * jt + l15) * D2LD + 4 * q);
        // acc[r] = ds^-1/2 q_s.k_s + coef (sum_p |gq_p - gk_p|^2 - 8 |t_i - t_j|^2) [- row terms], row i0 + 4 q + r, key 16 jt + l15
#pragma unroll
        for (int r = 0; r < 4; ++r) S[(4 * q + r) * IS + h * HS + jt * 16 + l15] = scale_t * (acc[r] + coef8 * d2v[r]);
      }
    }
    stamp(it, 1);
    // ---------------------------------------------------------------- phase 2: wave = 2 query rows, lanes = (head, key quarter)
    {
      const int h = l15 & 7;  // lanes with l15 >= 8 shadow head l15 - 8 (their MFMA columns are padding)
#pragma unroll
      for (int jt = E_EARLY; jt < RT; ++jt) load_e_tile(0, jt);
      MEM_FENCE();
      f16x8 wp[2][2];
#pragma unroll
      for (int pl = 0; pl < 2; ++pl)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) wp[pl][ks] = __builtin_bit_cast(f16x8, wp_lds[(pl * 2 + ks) * 64 + lane]);
      __syncthreads();  // the logits of all heads are in LDS, and every wave is done with the distance table (it lies over the scratch)
      stamp(it, 2);
      // Key tiles are consumed in pairs (32 keys), each pair completely - bias, softmax bookkeeping, o_e - as soon as it is in
      // registers (an online softmax inside the row); the probabilities go to LDS relative to the running maximum of their step
      // and are rescaled to the row maximum after the row.
      char* trt = reinterpret_cast<char*>(scr);                                    // [2 tiles][2 planes][16 keys][128 bytes] = 8 KiB
      const int wr_off = l15 * 128 + 8 * ((2 * q) ^ (4 * ((l15 >> 1) & 3)));      // ^ 64 ks: 8-byte unit 8 ks + 2 q of row l15
      const int rrow = 4 * q + (l15 >> 2);
      const int rd_off = rrow * 128 + 8 * ((l15 & 3) ^ (4 * ((rrow >> 1) & 3)));  // ^ 32 ct: unit 4 ct + (l15 & 3) of row rrow
#pragma unroll
      for (int ii = 0; ii < 2; ++ii) {
        const int il = 2 * wv + ii;  // local row
        float* Srow = S + il * IS + h * HS;
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
              v[4 * tl + r] = sv[r] + bscale * (acc[tl][0][r] + acc[tl][1][r]);
              smax = fmaxf(smax, v[4 * tl + r]);
            }
          }
          smax = fmaxf(smax, __shfl_xor(smax, 16));
          smax = fmaxf(smax, __shfl_xor(smax, 32));
          const float m_new = fmaxf(m_run, smax);
          const float alpha = T == 0 ? 0.0f : __expf(m_run - m_new);
          m_run = m_new;
          m_hist[T] = m_new;
          float psum = 0.f;
#pragma unroll
          for (int tt = 0; tt < 8; ++tt) {
            v[tt] = __expf(v[tt] - m_new);
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
            const int nx = ii * NT + 2 * T + RT;  // compile-time after unrolling
            if (nx < 2 * NT) {
              load_e_tile(nx / NT, nx % NT);
              load_e_tile((nx + 1) / NT, (nx + 1) % NT);
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
        const float inv = 1.0f / l_run;
        // the probabilities of a step are relative to the running maximum of that step: rescale to the row maximum
#pragma unroll
        for (int T = 0; T < NT / 2 - 1; ++T) {
          const float f = __expf(m_hist[T] - m_run);
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
          float* fo = feat + (prow0 + i0 + il) * PF + PF_OE + h * PC + 4 * q;
          const float sc = oscale * inv;
#pragma unroll
          for (int ct = 0; ct < 4; ++ct) {
            f32x4 o = oe[ct];
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] *= sc;
            *reinterpret_cast<f32x4*>(fo + 16 * ct) = o;
          }
          if (q == 0) st_inv[il * PH + h] = inv;
        }
      }
    }
    stamp(it, 3);
    // ---------------------------------------------------------------- phase 3: wave = head
    {
      constexpr int PFV = 16;  // value prefetch distance, key steps
      const int h = wv;
      const int pp = l15 & 7;
      const float* vbase = proj + (prow0 + 4 * q) * PNP + P_VS + h * PDS + 2 * l15;  // + (16 jt + r) rows; d = 2 l15 + dt
      const float* gbase = proj + (prow0 + 4 * q) * PNP + P_GV + h * 24 + 3 * pp;    // point pp, coords 0..2
      float2 vs[NS];
      float gx[NS], gy[NS], gz[NS];
      auto load_vals = [&](int stp) {
        const int64_t o = static_cast<int64_t>((stp >> 2) * 16 + (stp & 3)) * PNP;
        vs[stp] = *reinterpret_cast<const float2*>(vbase + o);
        gx[stp] = gbase[o];
        gy[stp] = gbase[o + 1];
        gz[stp] = gbase[o + 2];
      };
#pragma unroll
      for (int stp = 0; stp < PFV && stp < NS; ++stp) load_vals(stp);
      MEM_FENCE();
      __syncthreads();  // exp(logit - M) of all rows and the normalisers are in LDS
      stamp(it, 4);
      f32x4 os[2], og[3];
#pragma unroll
      for (int d = 0; d < 2; ++d) os[d] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int cc = 0; cc < 3; ++cc) og[cc] = f32x4{0.f, 0.f, 0.f, 0.f};
      const float* Prow = S + l15 * IS + h * HS + 4 * q;  // A operand: P[i = l15][j = 16 jt + 4 q + r]
      const bool more = it + 1 < count;
#pragma unroll
      for (int jt = 0; jt < NT; ++jt) {
        const f32x4 pa = *reinterpret_cast<const f32x4*>(Prow + jt * 16);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int stp = jt * 4 + r;
          if (stp + PFV < NS) {
            load_vals(stp + PFV);
            MEM_FENCE();
          }
          if (stp + PFV == NS - 1 || (NS <= PFV && stp == 0)) {
            // every value load of this tile has been issued: the next tile's phase-1 operands travel under the rest of phase 3
            // (unconditional - the last tile requests itself again: a conditional request would keep this tile's operands live
            // through phase 2 on the path that skips it)
            request_tile(more ? gt + 1 : gt);
            MEM_FENCE();
          }
          os[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(pa[r], vs[stp].x, os[0], 0, 0, 0);
          os[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(pa[r], vs[stp].y, os[1], 0, 0, 0);
          og[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(pa[r], gx[stp], og[0], 0, 0, 0);
          og[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(pa[r], gy[stp], og[1], 0, 0, 0);
          og[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(pa[r], gz[stp], og[2], 0, 0, 0);
        }
      }
      // D rows i = 4 q + r, column n = l15
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int il = 4 * q + r;
        const int64_t row = prow0 + i0 + il;
        const float inv = st_inv[il * PH + h];
        float* fr = feat + row * PF;
        *reinterpret_cast<float2*>(fr + PF_OS + h * PDS + 2 * l15) = make_float2(os[0][r] * inv, os[1][r] * inv);
        if (l15 < 8) {
          float* fo = fr + PF_OL + h * 24 + 3 * l15;
          const float* Rr = R + row * 9;
          const float* tr = t + row * 3;
          const float dx = og[0][r] * inv - tr[0], dy = og[1][r] * inv - tr[1], dz = og[2][r] * inv - tr[2];
          const float lx = dx * Rr[0] + dy * Rr[1] + dz * Rr[2];  // (p - t) R^T   (diffab_pytorch.py:336)
          const float ly = dx * Rr[3] + dy * Rr[4] + dz * Rr[5];
          const float lz = dx * Rr[6] + dy * Rr[7] + dz * Rr[8];
          fo[0] = lx; fo[1] = ly; fo[2] = lz;
          fr[PF_ON + h * 8 + l15] = sqrtf(lx * lx + ly * ly + lz * lz);
        }
      }
      if (more) write_d2();  // the next tile's distance table (the scratch it lies over is free since phase 2)
    }
    stamp(it, 5);
    __syncthreads();  // the image and the distance table belong to the next tile
  }
}

static unsigned long long* g_b6p_stamps = nullptr;
void set_attn_b6p_stamps(void* p) { g_b6p_stamps = static_cast<unsigned long long*>(p); }

bool attention_b6p_supported(const diffab_dims* d) {
  return d->D == 128 && d->C == PC && d->H == PH && d->DS == PDS && d->PQ == 8 && d->PV == 8 && (d->K == 64 || d->K == 128);
}
int attention_b6p_grid(const diffab_dims* d, int* tiles_per_wg) {
  const int total = d->B * (d->K / PTI);
  const int per = (total + 255) / 256;
  *tiles_per_wg = per;
  return (total + per - 1) / per;
}

// proj: fp32 projection buffer (value side), pair_planes: launch_pair_split() output, qk_ops: proj_planes_b6 operand planes
int launch_attention_b6p(const diffab_dims* d, const float* proj, const float* pair_planes, const float* R, const float* t, const float* Wb,
                         const float* gamma, float* feat, const float* qk_ops, hipStream_t st) {
  DIFFAB_REQUIRE(attention_b6p_supported(d) && proj && pair_planes && qk_ops, DIFFAB_ERR_ARG, "attention_b6p: unsupported operands");
  const int nt = d->K / PTI;
  const int lds = static_cast<int>(attention_b6p_lds_bytes(nt));
  int per = 1;
  const int grid = attention_b6p_grid(d, &per);
  const int64_t kside = static_cast<int64_t>(d->B) * d->K * (8 * 64 * 3 * 2 / 16);
#define B6P_LAUNCH(NT_)                                                                                                               \
  do {                                                                                                                                \
    DIFFAB_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(ipa_attn_b6p_kernel<NT_>),                                     \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, lds));                                           \
    timer_begin(st);                                                                                                                  \
    hipLaunchKernelGGL((ipa_attn_b6p_kernel<NT_>), dim3(grid), dim3(512), lds, st, proj, pair_planes + 64, R, t, Wb, gamma, feat,     \
                       d->B, per, g_b6p_stamps, pair_planes + 1, reinterpret_cast<const f32x4*>(qk_ops), kside);                      \
    timer_end(st);                                                                                                                    \
  } while (0)
  if (nt == 8) B6P_LAUNCH(8);
  else B6P_LAUNCH(4);
#undef B6P_LAUNCH
  DIFFAB_LAUNCH_CHECK();
  return DIFFAB_OK;
}

}  // namespace diffab
