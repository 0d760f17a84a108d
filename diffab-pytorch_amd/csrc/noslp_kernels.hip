// noslp_kernels.hip - the kernels of the library that hipcc's SLP vectoriser turned into the packed-fp32 form gfx950 miscomputes:
// v_pk_{mul,add,fma}_f32 op_sel:[0,1...] returns a wrong low result in lanes 48-63 while f16 / bf16 MFMAs are in flight on the SIMD - the
// kernel's own or, for these VALU / fp32-MFMA kernels, those of ANOTHER kernel's waves sharing the SIMD (a second stream):
// profiles/r05_pk_opsel_hazard.md, profiles/r06_lanes_48_63.md.  They were collected here to be compiled with the SLP vectoriser off on
// the device side (-Xarch_device -fno-slp-vectorize); since the value-plane work of round 6 EVERY translation unit is (csrc/Makefile
// NOSLP: measured neutral on the hot path): no packed fp32 arithmetic is formed from scalar code anywhere, the results are the same
// IEEE operations.  (A per-kernel
// __attribute__((target("no-packed-fp32-ops"))) was tried first: it blocks the inlining of every callee, __syncthreads included.)
// tools/isa_hazard_lint.py + tests/test_isa_lint.py keep the form out of every kernel of the built library.
#include "common.h"
#include "denoiser_internal.h"
#include "so3_math.h"
#include "ipa_attn_tile.h"  // geometry constants (ANP, OFF_*), f32x4

namespace diffab {

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

int launch_proj_frames_f32(const float* x, const float* const* W6, const float* R, const float* t, float* proj, int rows, hipStream_t st) {
  const size_t pj_lds = (2 * PJB * PJLD + PJROWS * 12) * sizeof(float);
#define PROJ_LAUNCH(FULL_)                                                                                                        \
  do {                                                                                                                            \
    DIFFAB_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(proj_frames_kernel<FULL_>),                                \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(pj_lds)));                  \
    hipLaunchKernelGGL((proj_frames_kernel<FULL_>), dim3((rows + PJROWS - 1) / PJROWS), dim3(512), pj_lds, st, x, W6[0], W6[1],   \
                       W6[2], W6[3], W6[4], W6[5], R, t, proj, rows);                                                             \
  } while (0)
  if (rows % PJROWS == 0) PROJ_LAUNCH(true);
  else PROJ_LAUNCH(false);
#undef PROJ_LAUNCH
  DIFFAB_LAUNCH_CHECK();
  return DIFFAB_OK;
}

// ------------------------------------------------------------------ heads epilogue
// O0 = O_t @ exp(hat(v))  (diffab_pytorch.py:594-596);  posterior = softmax(logits)  (:555)
__global__ void heads_finish_kernel(const float* __restrict__ v, const float* __restrict__ O_t, const float* __restrict__ logits, int V,
                                    int64_t rows, float* __restrict__ O0, float* __restrict__ post) {
  const int64_t r = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x;
  if (r >= rows) return;
  float ex[9], o[9], res[9];
  so3_rotvec_to_matrix(v[r * 3], v[r * 3 + 1], v[r * 3 + 2], ex);
#pragma unroll
  for (int k = 0; k < 9; ++k) o[k] = O_t[r * 9 + k];
  mat3_mul(o, ex, res);
#pragma unroll
  for (int k = 0; k < 9; ++k) O0[r * 9 + k] = res[k];
  float m = -INFINITY;
  for (int c = 0; c < V; ++c) m = fmaxf(m, logits[r * V + c]);
  float s = 0.f;
  for (int c = 0; c < V; ++c) s += expf(logits[r * V + c] - m);
  const float inv = 1.0f / s;
  for (int c = 0; c < V; ++c) post[r * V + c] = expf(logits[r * V + c] - m) * inv;
}

int launch_heads_finish(const float* v, const float* O_t, const float* logits, int V, int64_t rows, float* O0, float* post, hipStream_t st) {
  hipLaunchKernelGGL(heads_finish_kernel, dim3(static_cast<unsigned>((rows + 127) / 128)), dim3(128), 0, st, v, O_t, logits, V, rows, O0, post);
  DIFFAB_LAUNCH_CHECK();
  return DIFFAB_OK;
}

// ================================================================== backward pieces (denoiser_backward.hip launches them)
// d L / d v from G0 = d L / d O0 for O0 = O_t E, E = exp(hat(v)) = I + a S + b S^2 (diffab_pytorch.py:594-596, so3.py:219-237)
__device__ __forceinline__ void rotvec_head_bwd(const float (&G0)[9], const float* __restrict__ Ot, const float* __restrict__ v3,
                                                float* __restrict__ dv3, float* __restrict__ dOt = nullptr) {
  if (dOt != nullptr) {  // d L / d O_t = G0 E^T (the caller asked for frame gradients)
    float E[9];
    so3_rotvec_to_matrix(v3[0], v3[1], v3[2], E);
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int c = 0; c < 3; ++c) dOt[r * 3 + c] = (G0[r * 3 + 0] * E[c * 3 + 0] + G0[r * 3 + 1] * E[c * 3 + 1]) + G0[r * 3 + 2] * E[c * 3 + 2];
  }
  // G = dL/dE = O_t^T G0
  float G[9];
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int b = 0; b < 3; ++b) {
      float s = 0.f;
#pragma unroll
      for (int r = 0; r < 3; ++r) s += Ot[r * 3 + a] * G0[r * 3 + b];
      G[a * 3 + b] = s;
    }
  // S = hat(v), n = |v|, a = sin n / n, b = (1 - cos n) / n^2
  const float vx = v3[0], vy = v3[1], vz = v3[2];
  const float n = sqrtf(vx * vx + vy * vy + vz * vz);
  float sn, cn;
  sincosf(n, &sn, &cn);
  const float a = sn / n, b = (1.0f - cn) / (n * n);
  const float da = (cn - a) / n, db = (a - 2.0f * b) / n;  // derivatives with respect to n
  float S[9], S2[9];
  so3_hat(vx, vy, vz, S);
  mat3_mul(S, S, S2);
  float gS = 0.f, gS2 = 0.f;
#pragma unroll
  for (int k = 0; k < 9; ++k) { gS += G[k] * S[k]; gS2 += G[k] * S2[k]; }
  // <G, dS S + S dS> = <G S^T + S^T G, dS>;   dS = hat(dv)
  float H[9], ST[9], T1[9], T2[9];
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) ST[r * 3 + c] = S[c * 3 + r];
  mat3_mul(G, ST, T1);
  mat3_mul(ST, G, T2);
#pragma unroll
  for (int k = 0; k < 9; ++k) H[k] = a * G[k] + b * (T1[k] + T2[k]);
  const float vv[3] = {vx, vy, vz};
  // <H, hat(e_x)> = H21 - H12, <H, hat(e_y)> = H02 - H20, <H, hat(e_z)> = H10 - H01
  const float hk[3] = {H[7] - H[5], H[2] - H[6], H[3] - H[1]};
#pragma unroll
  for (int k = 0; k < 3; ++k) dv3[k] = hk[k] + (da * gS + db * gS2) * vv[k] / n;
}

__global__ void losses_bwd_kernel(const float* __restrict__ post, const float* __restrict__ tpost, const float* __restrict__ eps,
                                  const float* __restrict__ teps, const float* __restrict__ O0, const float* __restrict__ tO,
                                  const float* __restrict__ O_t, const float* __restrict__ v, const uint8_t* __restrict__ gm,
                                  const uint8_t* __restrict__ rm, const float* __restrict__ count, const float* __restrict__ up, int V,
                                  int64_t rows, float* __restrict__ d_logits, float* __restrict__ d_eps, float* __restrict__ d_v) {
  const int64_t i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x;
  if (i >= rows) return;
  const float g_seq = up[0], g_x = up[1], g_o = up[2];  // upstream gradients of the three losses
  const bool m = gm[i] && rm[i];
  const float invN = 1.0f / count[0];
  // KL(q || p) with p = softmax(logits): dL/dlogit_v = (p_v sum_u q_u - q_v) / N      (kl_div(log p, q), :857-859)
  float qs = 0.f;
  for (int c = 0; c < V; ++c) qs += tpost[i * V + c] > 0.0f ? tpost[i * V + c] : 0.0f;
  for (int c = 0; c < V; ++c) {
    const float q = tpost[i * V + c] > 0.0f ? tpost[i * V + c] : 0.0f;
    d_logits[i * V + c] = m ? g_seq * invN * (post[i * V + c] * qs - q) : 0.0f;
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) d_eps[i * 3 + c] = m ? g_x * invN * 2.0f * (eps[i * 3 + c] - teps[i * 3 + c]) : 0.0f;  // MSE (:860-862)
  if (!m) {
    d_v[i * 3] = d_v[i * 3 + 1] = d_v[i * 3 + 2] = 0.0f;
    return;
  }
  // orientation loss sum_jk (sum_r O0[r][j] tO[r][k] - delta_jk)^2   (:620-625)  -> G0 = dL/dO0
  float D[9], G0[9];
#pragma unroll
  for (int j = 0; j < 3; ++j)
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      float d = 0.f;
#pragma unroll
      for (int r = 0; r < 3; ++r) d += O0[i * 9 + r * 3 + j] * tO[i * 9 + r * 3 + k];
      D[j * 3 + k] = d - (j == k ? 1.0f : 0.0f);
    }
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      float s = 0.f;
#pragma unroll
      for (int k = 0; k < 3; ++k) s += 2.0f * D[j * 3 + k] * tO[i * 9 + r * 3 + k];
      G0[r * 3 + j] = g_o * invN * s;
    }
  rotvec_head_bwd(G0, O_t + i * 9, v + i * 3, d_v + i * 3);
}

// Denoiser outputs -> head pre-activations for ARBITRARY upstream cotangents (diffab_denoise_step_bwd): d eps-hat passes through,
// d posterior goes through the softmax of sequence_denoising (:555, :599), d O0-hat through O0 = O_t exp(hat(v)) (:594-596).
// Null cotangent pointers stand for zeros.
__global__ void heads_cotangent_kernel(const float* __restrict__ post, const float* __restrict__ c_post, const float* __restrict__ c_eps,
                                       const float* __restrict__ c_O0, const float* __restrict__ O_t, const float* __restrict__ v, int V,
                                       int64_t rows, float* __restrict__ d_logits, float* __restrict__ d_eps, float* __restrict__ d_v,
                                       float* __restrict__ d_Ot) {
  const int64_t i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x;
  if (i >= rows) return;
  float dot = 0.f;
  if (c_post)
    for (int c = 0; c < V; ++c) dot += post[i * V + c] * c_post[i * V + c];
  for (int c = 0; c < V; ++c) d_logits[i * V + c] = c_post ? post[i * V + c] * (c_post[i * V + c] - dot) : 0.0f;
#pragma unroll
  for (int c = 0; c < 3; ++c) d_eps[i * 3 + c] = c_eps ? c_eps[i * 3 + c] : 0.0f;
  if (c_O0 == nullptr) {
    d_v[i * 3] = d_v[i * 3 + 1] = d_v[i * 3 + 2] = 0.0f;
    if (d_Ot)
      for (int k = 0; k < 9; ++k) d_Ot[i * 9 + k] = 0.0f;
    return;
  }
  float G0[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) G0[k] = c_O0[i * 9 + k];
  rotvec_head_bwd(G0, O_t + i * 9, v + i * 3, d_v + i * 3, d_Ot ? d_Ot + i * 9 : nullptr);  // d O_t is WRITTEN here, the layers add to it
}

int launch_losses_bwd(const float* post, const float* tpost, const float* eps, const float* teps, const float* O0, const float* tO,
                      const float* O_t, const float* v, const uint8_t* gm, const uint8_t* rm, const float* count, const float* up, int V,
                      int64_t rows, float* d_logits, float* d_eps, float* d_v, hipStream_t st) {
  hipLaunchKernelGGL(losses_bwd_kernel, dim3(static_cast<unsigned>((rows + 127) / 128)), dim3(128), 0, st, post, tpost, eps, teps, O0, tO, O_t,
                     v, gm, rm, count, up, V, rows, d_logits, d_eps, d_v);
  DIFFAB_LAUNCH_CHECK();
  return DIFFAB_OK;
}
int launch_heads_cotangent(const float* post, const float* c_post, const float* c_eps, const float* c_O0, const float* O_t, const float* v,
                           int V, int64_t rows, float* d_logits, float* d_eps, float* d_v, float* d_Ot, hipStream_t st) {
  hipLaunchKernelGGL(heads_cotangent_kernel, dim3(static_cast<unsigned>((rows + 127) / 128)), dim3(128), 0, st, post, c_post, c_eps, c_O0,
                     O_t, v, V, rows, d_logits, d_eps, d_v, d_Ot);
  DIFFAB_LAUNCH_CHECK();
  return DIFFAB_OK;
}

// Gradients with respect to the FRAME (R_i, t_i) of residue i through one IPA layer (reference: euclidean_transform /
// inverse_euclidean_transform are plain differentiable torch code, diffab_pytorch.py:315-336, used at :410-413 and :453).  One
// work-group per residue row.  Two contributions, both accumulated (+=) into dR[row][9], dt[row][3]:
//  (a) local -> global of the q / k / v points, g = p R + t (:324): with dg = d loss / d g (dproj, BEFORE points_bwd_kernel rewrites it),
//      d t += sum_points dg,  d R[k][c] += sum_points p[k] dg[c],  p = (g - t) R^T recomputed from the stored global points;
//  (b) global -> local of the attention-weighted value points, o_l = (o_g - t) R^T (:336) and o_n = |o_l| (:454): with
//      dl = d o_l + d o_n o_l / o_n,  d t -= dl R,  d R[c][k] += dl[c] (o_g - t)[k],  (o_g - t) = o_l R.
// p and (o_g - t) are recovered through R^T = R^-1: the frames on this path are rotations (to fp32 rounding), as in every caller.
__global__ __launch_bounds__(256) void ipa_frames_bwd_kernel(const float* __restrict__ proj, const float* __restrict__ dproj, int NP, int pt_col0,
                                                             int n_pts, const float* __restrict__ feat, const float* __restrict__ dfeat, int F,
                                                             int ol_col0, int on_col0, int n_vpts, const float* __restrict__ R,
                                                             const float* __restrict__ t, float* __restrict__ dR, float* __restrict__ dt) {
  const int64_t row = blockIdx.x;
  const float* Rr = R + row * 9;
  const float r0 = Rr[0], r1 = Rr[1], r2 = Rr[2], r3 = Rr[3], r4 = Rr[4], r5 = Rr[5], r6 = Rr[6], r7 = Rr[7], r8 = Rr[8];
  const float tx = t[row * 3], ty = t[row * 3 + 1], tz = t[row * 3 + 2];
  float acc[12];  // dR row-major, then dt
#pragma unroll
  for (int k = 0; k < 12; ++k) acc[k] = 0.f;
  for (int pt = threadIdx.x; pt < n_pts; pt += blockDim.x) {
    const float* g = proj + row * NP + pt_col0 + pt * 3;
    const float* dg = dproj + row * NP + pt_col0 + pt * 3;
    const float gx = g[0] - tx, gy = g[1] - ty, gz = g[2] - tz;
    const float p[3] = {gx * r0 + gy * r1 + gz * r2, gx * r3 + gy * r4 + gz * r5, gx * r6 + gy * r7 + gz * r8};  // (g - t) R^T
    const float d[3] = {dg[0], dg[1], dg[2]};
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
      for (int c = 0; c < 3; ++c) acc[k * 3 + c] += p[k] * d[c];
    acc[9] += d[0]; acc[10] += d[1]; acc[11] += d[2];
  }
  for (int vp = threadIdx.x; vp < n_vpts; vp += blockDim.x) {
    const float* ol = feat + row * F + ol_col0 + vp * 3;
    const float* dol = dfeat + row * F + ol_col0 + vp * 3;
    const float on = feat[row * F + on_col0 + vp], don = dfeat[row * F + on_col0 + vp];
    float dl[3], og[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) dl[c] = dol[c] + (on > 0.0f ? don * ol[c] / on : 0.0f);
    og[0] = ol[0] * r0 + ol[1] * r3 + ol[2] * r6;  // (o_g - t) = o_l R
    og[1] = ol[0] * r1 + ol[1] * r4 + ol[2] * r7;
    og[2] = ol[0] * r2 + ol[1] * r5 + ol[2] * r8;
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
      for (int k = 0; k < 3; ++k) acc[c * 3 + k] += dl[c] * og[k];
    acc[9] -= dl[0] * r0 + dl[1] * r3 + dl[2] * r6;  // d t -= dl R
    acc[10] -= dl[0] * r1 + dl[1] * r4 + dl[2] * r7;
    acc[11] -= dl[0] * r2 + dl[1] * r5 + dl[2] * r8;
  }
  __shared__ float red[4][12];
#pragma unroll
  for (int k = 0; k < 12; ++k) {
    float v = acc[k];
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][k] = v;
  }
  __syncthreads();
  if (threadIdx.x < 12) {
    const int k = threadIdx.x;
    const float v = (red[0][k] + red[1][k]) + (red[2][k] + red[3][k]);
    if (k < 9) { if (dR) dR[row * 9 + k] += v; }
    else if (dt) dt[row * 3 + (k - 9)] += v;
  }
}

int launch_ipa_frames_bwd(const float* proj, const float* dproj, int NP, int pt_col0, int n_pts, const float* feat, const float* dfeat, int F,
                          int ol_col0, int on_col0, int n_vpts, const float* R, const float* t, float* dR, float* dt, int rows,
                          hipStream_t st) {
  hipLaunchKernelGGL(ipa_frames_bwd_kernel, dim3(rows), dim3(256), 0, st, proj, dproj, NP, pt_col0, n_pts, feat, dfeat, F, ol_col0, on_col0,
                     n_vpts, R, t, dR, dt);
  DIFFAB_LAUNCH_CHECK();
  return DIFFAB_OK;
}

}  // namespace diffab
