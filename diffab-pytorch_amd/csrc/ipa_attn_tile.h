// ipa_attn_tile.h - the fused IPA attention of one (patch, 16 query rows) item as a device function (see denoiser_fast.hip for
// the structure: phase 1 logits, phase 2 pair stream + softmax + o_e, phase 3 P V).  Reference: diffab_pytorch.py:389-465.
#pragma once
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>

namespace diffab {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// Compile-time fence for memory operations: keeps the hand-placed prefetch loads where they are written (hipcc otherwise
// sinks each load next to its first use, leaving one or two in flight and exposing every HBM / L2 round trip).
#ifndef MEM_FENCE
#define MEM_FENCE() asm volatile("" ::: "memory")
#endif

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
// 1e-4 parity bar (measured ~2e-6 on the outputs); libm's expf expansion is ~12 VALU ops each.
#define FAST_EXP(x) __expf(x)

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
// Every vector-memory instruction of this kernel costs the CU's memory pipe ~20 cycles whatever its width or how its lanes coalesce
// (profiles/r05_persistent.md: 261 per wave and item = 19.9 us of the item's 40), so operands are fetched as FULL 1 KiB wave
// instructions in memory order ("line-shaped": consecutive lanes on consecutive 16-byte chunks) and re-oriented through LDS.
//
// Phase 1's first operands of an item: the first SD key tiles of the wave's head (k_s 16 x 128 B + gk 16 x 96 B per tile), the q_s
// fragments and the query points of the item's 16 rows (16 x 96 B, line-shaped like a gk tile).  (Requested inside the PREVIOUS item's
// phase 3 by the patch-resident kernel - measured: phase 1 -3 us, phase 3 +2.2, phase 2 +1: the memory pipe is busy either way -
// and dropped; profiles/r05_persistent.md.)
template <int SD>
struct AttnP1Pre {
  f32x4 st[SD][4];
  f32x4 qa[2];
  f32x4 gqr[2];
};
// one operand tile of a head in memory order: 16 rows x (32 scalars at off_s + 32 h | 24 point coordinates at off_p + 24 h) = 3.5 KiB as
// four wave instructions.  Lanes 32..63 of the fourth have no chunk (ids 96..127) and re-read chunk 95: unconditional, so that every
// register of the destination is rewritten and nothing of it stays live across the phases that do not use it.  Addresses are a
// wave-uniform base (SGPRs) + one of three per-lane 32-bit offsets that serve every tile of the item: as 64-bit per-lane addresses the
// compiler builds all of them up front and parks them in scratch.
struct AttnLaneOff { int s, p0, p1; };  // float offsets of the lane's chunks inside a tile
__device__ __forceinline__ AttnLaneOff attn_lane_off(const int lane) {
  const int g0 = lane, g1 = lane + 64 < 96 ? lane + 64 : 95;  // point chunk ids (0..95): row = id / 6, chunk = id % 6
  return AttnLaneOff{(lane >> 3) * ANP + 4 * (lane & 7), (g0 / 6) * ANP + 4 * (g0 % 6), (g1 / 6) * ANP + 4 * (g1 % 6)};
}
__device__ __forceinline__ void attn_tile_request(f32x4 (&dst)[4], const float* __restrict__ proj, const int64_t row0, const int off_s,
                                                  const int off_p, const AttnLaneOff lo, const int h) {
  const float* us = proj + row0 * ANP + (off_s + h * ADS);  // wave-uniform
  const float* up = proj + row0 * ANP + (off_p + h * 24);
  dst[0] = *reinterpret_cast<const f32x4*>(us + lo.s);
  dst[1] = *reinterpret_cast<const f32x4*>(us + 8 * ANP + lo.s);
  dst[2] = *reinterpret_cast<const f32x4*>(up + lo.p0);
  dst[3] = *reinterpret_cast<const f32x4*>(up + lo.p1);
}
// piece 0 .. SD-1: key tile `piece`; piece SD: q_s fragments + the 16 rows' query points
template <int SD>
__device__ __forceinline__ void attn_p1_request(AttnP1Pre<SD>& pre, const int piece, const float* __restrict__ proj, const int64_t prow0,
                                                const int64_t krow0, const int i0, const int lane, const AttnLaneOff lo, const int h) {
  if (piece < SD) {
    attn_tile_request(pre.st[piece], proj, krow0 + 16 * piece, OFF_KS, OFF_GK, lo, h);
    return;
  }
  const int l15 = lane & 15, q = lane >> 4;
  const float* uq = proj + (prow0 + i0) * ANP + h * ADS;  // wave-uniform
  pre.qa[0] = *reinterpret_cast<const f32x4*>(uq + OFF_QS + l15 * ANP + 4 * q);
  pre.qa[1] = *reinterpret_cast<const f32x4*>(uq + OFF_QS + 16 + l15 * ANP + 4 * q);
  const float* ug = proj + (prow0 + i0) * ANP + (OFF_GQ + h * 24);
  pre.gqr[0] = *reinterpret_cast<const f32x4*>(ug + lo.p0);
  pre.gqr[1] = *reinterpret_cast<const f32x4*>(ug + lo.p1);
}

// One (patch b, 16-row tile) item of the fused attention: the body of ipa_attn_fast_kernel (denoiser_fast.hip: one item per work-group)
// and of the patch-resident module kernel (ipa_persistent.hip: a work-group walks the eight row tiles of ITS patch, layer after layer).
// 512 threads; S: the dynamic LDS (ipa_attn_lds_bytes(NT)); stamp_id: the slot of this item in the diagnostic stamp buffer.
// NW: waves of the work-group (8: wave = head in phases 1 and 3, two rows in phase 2; 4: two heads / four rows per wave - the form that
// fits two work-groups on a CU: with the 64-key chunk image its LDS is 79.5 KiB, and the two groups' phases interleave on the CU's pipes).
// VPL (with PLANES, NW = 8; round 6): phase 3 (P x V) runs on the f16 matrix cores as well.  The value side arrives as two fp16 planes in
// fragment order, cut once per (patch, layer) by attn_planes_tile.h (per (patch, head, 32-key step) the four 16-column tiles [v_s 0..15 |
// v_s 16..31 | x, y of the 8 points | z of the 8 points + a ones column], one power-of-two scale per (head, step) and kind in vsc, point
// coordinates relative to the patch's first translation); the probabilities are split into two fp16 planes
// on the fly (2^15 P = p1 + p2): three exact partial products per tile and 32 keys - 12 MFMAs of 16 cycles per (head, 32 keys) against 32
// f32 MFMAs of 32 cycles (which also block the vector ALU), 8 linear 1 KiB loads per 32 keys straight into B fragments, no LDS staging.
template <int NT, bool MULTI, bool PLANES = false, bool TAPE = false, int NW = 8, bool VPL = false>
__device__ __forceinline__ void ipa_attn_tile(float* __restrict__ S, const int b, const int tile, const unsigned stamp_id,
                                              const float* __restrict__ proj, const float* __restrict__ e,
                                              const float* __restrict__ R, const float* __restrict__ t,
                                              const float* __restrict__ Wb, const float* __restrict__ gamma,
                                              float* __restrict__ feat, int NC_arg,
                                              unsigned long long* __restrict__ stamps, const float* __restrict__ esc = nullptr,
                                              float* __restrict__ tape_p = nullptr, float* __restrict__ tape_d2 = nullptr,
                                              const f32x4* __restrict__ vpl = nullptr, const float* __restrict__ vsc = nullptr) {
  static_assert(NW == 8 || (NW == 4 && PLANES && !TAPE), "the four-wave form exists for the plane kernels");
  static_assert(!VPL || (PLANES && NW == 8 && NT % 2 == 0), "value planes: the eight-wave plane kernels");
  constexpr int HPW = AH / NW;  // heads per wave (phases 1 and 3)
  constexpr int RPW = TI / NW;  // query rows per wave (phase 2)
  static_assert(!PLANES || NT % 2 == 0, "the o_e product takes key tiles in pairs");
  static_assert(!TAPE || (!MULTI && !PLANES), "the tape form is the single-chunk fp32-pair kernel");
  const int NC = MULTI ? NC_arg : 1;
  // Keys are processed in NC chunks of KC = 16 NT with an online softmax: the LDS image holds the logits / (unnormalised)
  // probabilities of ONE chunk, each (row, head) keeps a running maximum M and sum L, and the partial outputs of earlier chunks
  // are rescaled by exp(M_old - M_new) through the feature rows in global memory.  K = 64 and 128 are the single-chunk case;
  // K = 192, 256, ... reuse the same 16-row structure instead of needing a K-proportional LDS image.
  // S: [TI][AH][KC+8] (+8 per i), then the per-wave scratch and the softmax state
  constexpr int KC = NT * 16;
  const int K = NC * KC;
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
      if ((threadIdx.x & 63) == 0) stamps[(static_cast<size_t>(stamp_id) * 8 + (threadIdx.x >> 6)) * 8 + k] = tnow;
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  stamp(0);
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
  constexpr int E_LAG = MULTI ? 2 : 0;    // the next row's tile loads trail the retiring tiles by this many (VGPRs)
  constexpr int E_EARLY = MULTI ? 1 : 2;  // pair tiles of phase 2's first row started under the tail of phase 1 (swept in rounds 2-3)
  float* scr = S + TI * IS + wv * SCR_FLOATS;
  float* st_fac = S + TI * IS + NW * SCR_FLOATS;  // [TI][AH] exp(M_old - M_new) of the current chunk
  float* st_inv = st_fac + TI * AH;              // [TI][AH] 1 / L after the last chunk (1 before)
  float* wb_lds = st_inv + TI * AH;              // [4 sg][64 lanes][4]: B fragments of the bias product (same for every wave)
  float* gq_lds = wb_lds + (PLANES ? 0 : 4 * 64 * 4) + wv * (TI * 24);  // per wave: the item's query points of its head, [16 rows][24] as loaded
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
  const float* erow[RPW];
  float Mrun[RPW], Lrun[RPW];  // online-softmax state of (row RPW wv + ii, head l15 & 7)
#pragma unroll
  for (int ii = 0; ii < RPW; ++ii) {
    erow[ii] = e + ((prow0 + i0 + RPW * wv + ii) * K) * AC;
    Mrun[ii] = -INFINITY;
    Lrun[ii] = 0.f;
  }
  // PLANES: 1 / s_i of the wave's two pair rows, fetched here through the scalar cache (wave-uniform address).  As a vector load at
  // the top of each row its s_waitcnt - vmcnt retires in order - drained every pair tile in flight, twice per wave and phase 2.
  float inv_s2[RPW];
#pragma unroll
  for (int ii = 0; ii < RPW; ++ii) inv_s2[ii] = 1.0f;
  if constexpr (PLANES) {
    const float* ep = esc + 2 * (prow0 + i0 + RPW * __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6)));
#pragma unroll
    for (int ii = 0; ii < RPW; ++ii) inv_s2[ii] = ep[2 * ii + 1];
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
    f32x4 ev[RPW][NT][4];
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
    // ---------------------------------------------------------------- phase 1: wave = head (NW = 4: two heads, one after the other)
#pragma unroll
    for (int hw = 0; hw < HPW; ++hw) {
      const int h = HPW * wv + hw;
      const float scale_s = 0.17677669529663687f;                    // 32^-1/2  (:353)
      const float coef_p = -0.5f * 0.16666666666666666f * gamma[h];  // -1/2 (4.5*8)^-1/2 gamma_h  (:372, :431-436)
      // line-shaped loads of one key tile (16 keys): k_s 16 x 128 B (8 lanes per key), gk 16 x 96 B (6 lanes per key)
      const int ks_dst = (lane >> 3) * KLD + 4 * (lane & 7);
      const int g0 = lane, g1 = lane + 64;  // gk chunk ids (0..95): key = id / 6, chunk = id % 6
      const int gk_dst0 = 16 * KLD + (g0 / 6) * GLD + 4 * (g0 % 6), gk_dst1 = 16 * KLD + (g1 / 6) * GLD + 4 * (g1 % 6);
      constexpr int SD = MULTI ? 3 : 4;  // register staging depth: SD - 1 key tiles of lookahead
      static_assert(E_EARLY <= SD, "the early pair tiles are requested in the last E_EARLY iterations, which must not request key tiles any more");
      AttnP1Pre<SD> pre;
      f32x4 (&st)[SD][4] = pre.st;
      const AttnLaneOff lo = attn_lane_off(lane);
      auto load_keys = [&](int sb, int jt) { attn_tile_request(st[sb], proj, krow0 + 16 * jt, OFF_KS, OFF_GK, lo, h); };
      auto stage_keys = [&](int sb, int lb) {
        float* t_ = scr + lb * P1_TILE;
        *reinterpret_cast<f32x4*>(t_ + ks_dst) = st[sb][0];
        *reinterpret_cast<f32x4*>(t_ + ks_dst + 8 * KLD) = st[sb][1];
        *reinterpret_cast<f32x4*>(t_ + gk_dst0) = st[sb][2];
        if (g1 < 96) *reinterpret_cast<f32x4*>(t_ + gk_dst1) = st[sb][3];
      };
      // the first SD key tiles; A operand: q_s rows i0 + l15, k = 16 sg + 4 q + s; the 16 rows' query points
      // (the query side first: its LDS round trip runs while the key tiles arrive)
      attn_p1_request<SD>(pre, SD, proj, prow0, krow0, i0, lane, lo, h);
#pragma unroll
      for (int piece = 0; piece < SD && piece < NT; ++piece) attn_p1_request<SD>(pre, piece, proj, prow0, krow0, i0, lane, lo, h);
      f32x4 (&qa)[2] = pre.qa;
      MEM_FENCE();
      // query points of the 4 rows this lane accumulates (rows i0 + 4q + r): through the wave's LDS tile - 2 vector-memory instructions
      // instead of 24 (the 16 lanes of a quarter share each address, but a broadcast load costs the memory pipe what any other does)
      *reinterpret_cast<f32x4*>(gq_lds + 4 * g0) = pre.gqr[0];
      if (g1 < 96) *reinterpret_cast<f32x4*>(gq_lds + 4 * g1) = pre.gqr[1];
      f32x4 gq[4][6];
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int cc = 0; cc < 6; ++cc) gq[r][cc] = *reinterpret_cast<const f32x4*>(gq_lds + (4 * q + r) * 24 + 4 * cc);
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
        } else if (jt + E_EARLY >= NT && hw == HPW - 1) {
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
        if (c == 0 && jt == 0 && hw == 0) stamp(6);
        if (c == 0 && jt == 3 && hw == 0) stamp(7);
      }
    }
    if (c == 0) stamp(1);
    // ---------------------------------------------------------------- phase 2: wave = 2 query rows, lanes = (head, key quarter)
    {
      const int h = l15 & 7;  // lanes with l15 >= 8 shadow head l15-8 (their MFMA columns are padding)
      constexpr int E_DEPTH_FP32 = 3;  // fp32 pair stream: tiles of the first row in flight before its bias loop starts; the rest follow one per
                                       // consumed tile (all 8 at once: 256 KiB per CU requested in one burst, +3.5 % kernel time: the queue it
                                       // builds delays every other CU's loads)
      // PLANES: RT = pair tiles of the wave's 2 NT-tile stream held in registers (requested RT tiles ahead of their use): half a row.
      // (A whole row spills in the chunked kernel, and in the single-chunk one its 24 loads per wave in front of the barrier take
      // 4.6 k cycles to issue on the waves that finish phase 1 last: 0.326 ms against 0.321 with half a row.)
      constexpr int RT = NT / 2;
      constexpr int E_DEPTH0 = PLANES ? RT : (E_DEPTH_FP32 < NT ? (E_DEPTH_FP32 > E_EARLY ? E_DEPTH_FP32 : E_EARLY) : NT);
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
        // (round 6: the row's low bit enters the swizzle too - without it rows 2 m and 2 m + 1 of an 8-lane store group shared their four
        // banks: every ds_write_b128 of the pair tile was a 2-way conflict, half of the kernel's 26 % conflict cycles, SQ_LDS_BANK_CONFLICT)
        const int wr_off = l15 * 128 + 8 * ((2 * q) ^ (4 * ((l15 >> 1) & 3)) ^ (2 * (l15 & 1)));  // ^ 64 ks: 8-byte unit 8 ks + 2 q of row l15
        const int rrow = 4 * q + (l15 >> 2);
        const int rd_off = rrow * 128 + 8 * ((l15 & 3) ^ (4 * ((rrow >> 1) & 3)) ^ (2 * (rrow & 1)));  // ^ 32 ct: unit 4 ct + (l15 & 3) of row rrow
#pragma unroll
        for (int ii = 0; ii < RPW; ++ii) {
          const int il = RPW * wv + ii;  // local row
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
              if (nx < RPW * NT) {
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

    // ---------------------------------------------------------------- phase 3: wave = head (NW = 4: two heads, one after the other)
    if constexpr (VPL) {
      const int h = wv;
      constexpr int TPC = NT / 2;  // 32-key steps per chunk
      const int Tsteps = K >> 5;   // per patch
      // all value fragments of this chunk (TPC x 8 KiB per head), requested before the barrier: [step][tile 0..3][plane h1, h2]
      int lane3 = lane0;
      asm volatile("" : "+v"(lane3));
      const f32x4* vsrc = vpl + ((static_cast<int64_t>(b) * AH + h) * Tsteps + c * TPC) * 512 + lane3;
      f32x4 vf[TPC][8];
#pragma unroll
      for (int T = 0; T < TPC; ++T)
#pragma unroll
        for (int u = 0; u < 8; ++u) vf[T][u] = vsrc[(T * 8 + u) * 64];
      // 1 / s of the head's v_s and point planes, per step (wave-uniform: scalar loads); x 2^-15 for the scaled P
      const float* scp = vsc + ((static_cast<int64_t>(b) * AH + h) * Tsteps + c * TPC) * 2;
      float isc[TPC][3];
#pragma unroll
      for (int T = 0; T < TPC; ++T) {
        isc[T][0] = isc[T][1] = scp[2 * T] * (1.0f / 32768.0f);
        isc[T][2] = scp[2 * T + 1] * (1.0f / 32768.0f);
      }
      MEM_FENCE();
      __syncthreads();  // exp(logit - M) of all rows and the rescale factors are in LDS (the value fragments arrive during the wait)
      if (c == 0) stamp(4);
      // tot: o_s dims 0..15 | o_s dims 16..31 | x, y of the points | z of the points.  Column 8 of the z tile is the planes' ones column:
      // mass = sum_j (p1 + p2)_j, the probability mass AS THE SPLIT PLANES SEE IT - the local points are (sum_j P~_j (gv_j - c) -
      // (t_i - c) mass) / L = sum_j P~_j (gv_j - t_i) / L, so the 2^-22 the planes cut off P multiplies the distance of key j from the
      // QUERY (small where the mass is), not from the patch's reference point c.
      f32x4 tot[4], mass = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int u = 0; u < 4; ++u) tot[u] = f32x4{0.f, 0.f, 0.f, 0.f};
      const float* Prow = S + l15 * IS + h * HS + 4 * q;  // A operand: P[i = l15][j = 32 T + 16 tl + 4 q + r] <-> k slot (q, e = 4 tl + r)
#pragma unroll
      for (int T = 0; T < TPC; ++T) {
        const f32x4 pa0 = *reinterpret_cast<const f32x4*>(Prow + (2 * T) * 16), pa1 = *reinterpret_cast<const f32x4*>(Prow + (2 * T + 1) * 16);
        f16x8 p1, p2;  // 2^15 P = p1 + p2 (P <= 1: as high in the fp16 range as overflow allows, so small P keep their second plane)
#pragma unroll
        for (int e8 = 0; e8 < 8; ++e8) {
          const float x = 32768.0f * (e8 < 4 ? pa0[e8 & 3] : pa1[e8 & 3]);
          const _Float16 hh = static_cast<_Float16>(x);
          p1[e8] = hh;
          p2[e8] = static_cast<_Float16>(x - static_cast<float>(hh));
        }
        f32x4 acc[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) acc[u] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < 4; ++u) acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(p2, __builtin_bit_cast(f16x8, vf[T][2 * u]), acc[u], 0, 0, 0);
#pragma unroll
        for (int u = 0; u < 4; ++u) acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(p1, __builtin_bit_cast(f16x8, vf[T][2 * u + 1]), acc[u], 0, 0, 0);
#pragma unroll
        for (int u = 0; u < 4; ++u) acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(p1, __builtin_bit_cast(f16x8, vf[T][2 * u]), acc[u], 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          tot[0][r] = __builtin_fmaf(isc[T][0], acc[0][r], tot[0][r]);
          tot[1][r] = __builtin_fmaf(isc[T][1], acc[1][r], tot[1][r]);
          tot[2][r] = __builtin_fmaf(isc[T][2], acc[2][r], tot[2][r]);
          tot[3][r] = __builtin_fmaf(isc[T][2], acc[3][r], tot[3][r]);
          mass[r] += acc[3][r];  // (lanes l15 == 8: the ones column, unscaled)
        }
      }
      // D rows i = 4 q + r, column n = l15; earlier chunks' sums are rescaled through the feature row
      const float* tc = t + prow0 * 3;  // the reference point of the value planes' coordinates: the patch's first translation
      const float c0 = tc[0], c1 = tc[1], c2 = tc[2];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int il = 4 * q + r;
        const int64_t row = prow0 + i0 + il;
        const float fac = st_fac[il * AH + h], inv = st_inv[il * AH + h];
        float* fr = feat + row * AF;
        float o0 = tot[0][r], o1 = tot[1][r];
        float* po = fr + FOFF_OS + h * ADS + l15;
        if (MULTI && c > 0) {
          o0 += po[0] * fac;
          o1 += po[16] * fac;
        }
        po[0] = o0 * inv;
        po[16] = o1 * inv;
        const float gy_lane = __shfl_xor(tot[2][r], 8);  // lanes 0..7 hold x of point l15, lanes 8..15 y of point l15 - 8
        float m_ = __shfl(mass[r], (lane & 48) | 8) * (1.0f / 32768.0f);  // the mass of row 4 q + r, from the ones column's lane
        if (l15 < 8) {
          float* fo = fr + FOFF_OL + h * 24 + 3 * l15;
          float* fm = fr + FOFF_ON + h * AP + l15;  // the norm's slot parks the running mass between chunks
          float g0_ = tot[2][r], g1_ = gy_lane, g2_ = tot[3][r];
          if (MULTI && c > 0) {  // running (unnormalised, centred) sums are parked in the o_l slot between chunks
            g0_ += fo[0] * fac;
            g1_ += fo[1] * fac;
            g2_ += fo[2] * fac;
            m_ += fm[0] * fac;
          }
          if (last) {
            const float* Rr = R + row * 9;
            const float* tr = t + row * 3;
            const float dx = (g0_ - m_ * (tr[0] - c0)) * inv, dy = (g1_ - m_ * (tr[1] - c1)) * inv, dz = (g2_ - m_ * (tr[2] - c2)) * inv;
            const float lx = dx * Rr[0] + dy * Rr[1] + dz * Rr[2];  // (p - t) R^T   (diffab_pytorch.py:336)
            const float ly = dx * Rr[3] + dy * Rr[4] + dz * Rr[5];
            const float lz = dx * Rr[6] + dy * Rr[7] + dz * Rr[8];
            fo[0] = lx; fo[1] = ly; fo[2] = lz;
            fm[0] = sqrtf(lx * lx + ly * ly + lz * lz);
          } else {
            fo[0] = g0_; fo[1] = g1_; fo[2] = g2_;
            fm[0] = m_;
          }
        }
      }
    } else {
#pragma unroll
    for (int hw = 0; hw < HPW; ++hw) {
      const int h = HPW * wv + hw;
      const int pp = l15 & 7;
      // Point sums: columns 0..7 of ONE MFMA tile hold x of the 8 points, columns 8..15 y (z in a second tile, its upper half
      // duplicates): 4 f32 MFMAs per key step instead of 5 - this phase is bound by exactly those (32 cycles each).
      const int xy = l15 >> 3;  // 0: this lane's column is x of point pp, 1: y
      // Value tiles (v_s 16 x 128 B + gv 16 x 96 B per 16 keys) take the route of phase 1's key tiles: four line-shaped wave loads per
      // tile, SD3 tiles ahead, staged through the wave's scratch (free once the wave has left phase 2: a wave reads only what it wrote)
      // and read back as B fragments - 32 vector-memory instructions per wave instead of the 96 narrow ones (8 / 4 / 4 bytes per lane) the
      // fragment-shaped loads took.
      constexpr int SD3 = NT < 4 ? NT : 4;
      f32x4 sv[SD3][4];
      int lane3 = lane0;  // (an opaque copy: the offsets below must be rebuilt here, not kept alive from phase 1 through phase 2)
      asm volatile("" : "+v"(lane3));
      const AttnLaneOff lo3 = attn_lane_off(lane3);
      auto load_vals = [&](int sb, int jt) { attn_tile_request(sv[sb], proj, krow0 + 16 * jt, OFF_VS, OFF_GV, lo3, h); };
      const int vs_dst = (lane >> 3) * KLD + 4 * (lane & 7);
      const int g0v = lane, g1v = lane + 64;
      const int gv_dst0 = 16 * KLD + (g0v / 6) * GLD + 4 * (g0v % 6), gv_dst1 = 16 * KLD + (g1v / 6) * GLD + 4 * (g1v % 6);
      auto stage_vals = [&](int sb, int lb) {
        float* t_ = scr + lb * P1_TILE;
        *reinterpret_cast<f32x4*>(t_ + vs_dst) = sv[sb][0];
        *reinterpret_cast<f32x4*>(t_ + vs_dst + 8 * KLD) = sv[sb][1];
        *reinterpret_cast<f32x4*>(t_ + gv_dst0) = sv[sb][2];
        if (g1v < 96) *reinterpret_cast<f32x4*>(t_ + gv_dst1) = sv[sb][3];
      };
      // fragments of one key tile: step r = key 16 jt + 4 q + r: d = 2 l15, 2 l15 + 1 of v_s | x or y of point pp | z of point pp
      struct ValFrag { float2 vs[4]; float gxy[4], gz[4]; };
      auto read_vfrags = [&](int jt) {
        ValFrag f;
        const float* t_ = scr + (jt & 1) * P1_TILE;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          f.vs[r] = *reinterpret_cast<const float2*>(t_ + (4 * q + r) * KLD + 2 * l15);
          f.gxy[r] = t_[16 * KLD + (4 * q + r) * GLD + 3 * pp + xy];
          f.gz[r] = t_[16 * KLD + (4 * q + r) * GLD + 3 * pp + 2];
        }
        return f;
      };
#pragma unroll
      for (int jt = 0; jt < SD3; ++jt) load_vals(jt, jt);
      MEM_FENCE();
      if (hw == 0) {
        __syncthreads();  // exp(logit - M) of all rows and the rescale factors are in LDS (the first value tiles arrive during the wait)
        if (c == 0) stamp(4);
      }
      stage_vals(0, 0);
      ValFrag vcur = read_vfrags(0);
      if (NT > 1) stage_vals(1 % SD3, 1);
      f32x4 os[2], og[2];  // og[0]: x | y of the points, og[1]: z
#pragma unroll
      for (int d = 0; d < 2; ++d) os[d] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int cc = 0; cc < 2; ++cc) og[cc] = f32x4{0.f, 0.f, 0.f, 0.f};
      const float* Prow = S + l15 * IS + h * HS + 4 * q;  // A operand: P[i = l15][j = 16 jt + 4 q + r]
#pragma unroll
      for (int jt = 0; jt < NT; ++jt) {
        if (jt + 2 < NT) stage_vals((jt + 2) % SD3, jt & 1);  // tile jt+2 -> the buffer tile jt was read from (LDS ops retire in order)
        ValFrag vnxt = vcur;
        if (jt + 1 < NT) vnxt = read_vfrags(jt + 1);
        if (jt + SD3 < NT) {
          load_vals(jt % SD3, jt + SD3);
          MEM_FENCE();
        }
        const f32x4 pa = *reinterpret_cast<const f32x4*>(Prow + jt * 16);
        if constexpr (TAPE) {  // P[i = l15][16 jt + 4 q ..]: the image holds exp(logit - M), the row's 1 / L is in st_inv
          const float pinv = st_inv[l15 * AH + h];
          *reinterpret_cast<f32x4*>(tape_p + ((static_cast<int64_t>(b) * AH + h) * K + i0 + l15) * K + jt * 16 + 4 * q) = pa * pinv;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          os[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(pa[r], vcur.vs[r].x, os[0], 0, 0, 0);
          os[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(pa[r], vcur.vs[r].y, os[1], 0, 0, 0);
          og[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(pa[r], vcur.gxy[r], og[0], 0, 0, 0);
          og[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(pa[r], vcur.gz[r], og[1], 0, 0, 0);
        }
        vcur = vnxt;
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
    }  // !VPL
    if (!last) __syncthreads();  // the next chunk's phase 1 overwrites the image
  }
  stamp(5);
}

// dynamic LDS of one item: logits image + per-wave scratch + softmax state + bias fragments + per-wave query-point tiles
constexpr size_t ipa_attn_lds_bytes(int nt, int nw = 8, bool planes = false) {
  return (static_cast<size_t>(TI) * (AH * (16 * nt + 8) + 8) + nw * 2 * 16 * 72 + 2 * TI * AH + (planes ? 0 : 4 * 64 * 4) + nw * TI * 24) * sizeof(float);
}

}  // namespace diffab
