// denoiser_generic.hip - the denoise step for ARBITRARY model dims (any D, C, H, DS, PQ, PV, K).
// Correct-by-construction VALU kernels: an LDS-tiled fp32 GEMM for every nn.Linear and one work-group
// per (patch, query residue) for the attention.  The benchmark geometry (D=128, C=64, H=8, DS=32, P=8)
// is served by the MFMA kernels in denoiser_fast.hip; this file is what every other geometry runs on
// and the in-GPU cross-check for the fast path.
//
// Reference: InvariantPointAttentionLayer.forward diffab_pytorch.py:389-465, Denoiser.forward :558-607.
#include "common.h"
#include "denoiser_internal.h"
#include "so3_math.h"

namespace diffab {

// ------------------------------------------------------------------ Y = act(X W^T + b)
// X (M, Kd) row stride ldx; W (N, Kd) contiguous (nn.Linear layout); Y (M, N) row stride ldy.
constexpr int GB_M = 64, GB_N = 64, GB_K = 16;

template <bool RELU>
__global__ __launch_bounds__(256) void linear_generic_kernel(const float* __restrict__ X, int ldx, const float* __restrict__ W,
                                                             const float* __restrict__ bias, float* __restrict__ Y, int ldy, int M, int N,
                                                             int Kd) {
  __shared__ float As[GB_K][GB_M + 4];
  __shared__ float Bs[GB_K][GB_N + 4];
  const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
  const int m0 = blockIdx.y * GB_M, n0 = blockIdx.x * GB_N;
  float acc[4][4] = {};
  for (int k0 = 0; k0 < Kd; k0 += GB_K) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int idx = tid + r * 256;          // 0..1023
      const int row = idx >> 4, kk = idx & 15;  // 64 rows x 16 k
      const int gm = m0 + row, gn = n0 + row, gk = k0 + kk;
      As[kk][row] = (gm < M && gk < Kd) ? X[static_cast<int64_t>(gm) * ldx + gk] : 0.0f;
      Bs[kk][row] = (gn < N && gk < Kd) ? W[static_cast<int64_t>(gn) * Kd + gk] : 0.0f;
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < GB_K; ++kk) {
      float a[4], b[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) { a[i] = As[kk][ty * 4 + i]; b[i] = Bs[kk][tx * 4 + i]; }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] += a[i] * b[j];
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int gm = m0 + ty * 4 + i;
    if (gm >= M) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int gn = n0 + tx * 4 + j;
      if (gn >= N) continue;
      float v = acc[i][j] + (bias ? bias[gn] : 0.0f);
      if (RELU) v = fmaxf(v, 0.0f);
      Y[static_cast<int64_t>(gm) * ldy + gn] = v;
    }
  }
}

int launch_linear_generic(const float* X, int ldx, const float* W, const float* bias, float* Y, int ldy, int M, int N, int Kd, bool relu,
                          hipStream_t st) {
  dim3 grid((N + GB_N - 1) / GB_N, (M + GB_M - 1) / GB_M);
  if (relu)
    hipLaunchKernelGGL(linear_generic_kernel<true>, grid, dim3(256), 0, st, X, ldx, W, bias, Y, ldy, M, N, Kd);
  else
    hipLaunchKernelGGL(linear_generic_kernel<false>, grid, dim3(256), 0, st, X, ldx, W, bias, Y, ldy, M, N, Kd);
  DIFFAB_LAUNCH_CHECK();
  return DIFFAB_OK;
}

// ------------------------------------------------------------------ small glue kernels
// cat[res_ctx, E[seq]]   (diffab_pytorch.py:572-573)
__global__ void embed_concat_kernel(const float* __restrict__ res_ctx, const float* __restrict__ emb, const int64_t* __restrict__ seq, int D,
                                    int64_t rows, float* __restrict__ out) {
  const int64_t r = blockIdx.x;
  if (r >= rows) return;
  const int64_t s = seq[r];
  for (int d = threadIdx.x; d < D; d += blockDim.x) {
    out[r * 2 * D + d] = res_ctx[r * D + d];
    out[r * 2 * D + D + d] = emb[s * D + d];
  }
}

// cat[h, (beta, sin beta, cos beta)]   (diffab_pytorch.py:584-588)
__global__ void beta_concat_kernel(const float* __restrict__ h, const float* __restrict__ beta, int D, int K, int64_t rows,
                                   float* __restrict__ out) {
  const int64_t r = blockIdx.x;
  if (r >= rows) return;
  for (int d = threadIdx.x; d < D; d += blockDim.x) out[r * (D + 3) + d] = h[r * D + d];
  if (threadIdx.x == 0) {
    const float b = beta[r / K];
    out[r * (D + 3) + D + 0] = b;
    out[r * (D + 3) + D + 1] = sinf(b);
    out[r * (D + 3) + D + 2] = cosf(b);
  }
}

// local -> global points in place: g = p R + t (row-vector convention, diffab_pytorch.py:324).
// buf: (rows, ld) with the point block at column `col0`, n_pts points per residue.
__global__ void points_to_global_kernel(float* __restrict__ buf, int ld, int col0, int n_pts, const float* __restrict__ R,
                                        const float* __restrict__ t, int64_t rows) {
  const int64_t gid = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x;
  if (gid >= rows * n_pts) return;
  const int64_t r = gid / n_pts;
  const int p = static_cast<int>(gid % n_pts);
  float* q = buf + r * ld + col0 + p * 3;
  const float* Rr = R + r * 9;
  const float x = q[0], y = q[1], z = q[2];
  q[0] = (x * Rr[0] + y * Rr[3] + z * Rr[6]) + t[r * 3 + 0];
  q[1] = (x * Rr[1] + y * Rr[4] + z * Rr[7]) + t[r * 3 + 1];
  q[2] = (x * Rr[2] + y * Rr[5] + z * Rr[8]) + t[r * 3 + 2];
}

// ------------------------------------------------------------------ attention, one work-group per (patch, query residue)
// proj row layout (ld = NP): [q_s H*DS | k_s H*DS | v_s H*DS | gq H*PQ*3 | gk H*PQ*3 | gv H*PV*3], points already global.
// feat row layout (F): [o_s H*DS | o_e H*C | o_l H*PV*3 | o_n H*PV]   (diffab_pytorch.py:460)
__global__ __launch_bounds__(256) void ipa_attn_generic_kernel(const float* __restrict__ proj, const float* __restrict__ e,
                                                               const float* __restrict__ R, const float* __restrict__ t,
                                                               const float* __restrict__ Wb, const float* __restrict__ gamma,
                                                               float* __restrict__ feat, int K, int C, int H, int DS, int PQ, int PV) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int b = blockIdx.x / K, i = blockIdx.x % K;
  const int NP = 3 * H * DS + 2 * H * PQ * 3 + H * PV * 3;
  const int F = H * DS + H * C + H * PV * 3 + H * PV;
  const int off_ks = H * DS, off_vs = 2 * H * DS, off_gq = 3 * H * DS, off_gk = off_gq + H * PQ * 3, off_gv = off_gk + H * PQ * 3;
  float* attn = smem;                  // H*K
  float* qrow = attn + H * K;          // H*DS + H*PQ*3
  float* og = qrow + H * DS + H * PQ * 3;  // H*PV*3
  const int64_t row_i = static_cast<int64_t>(b) * K + i;
  const float* prow = proj + row_i * NP;
  for (int d = threadIdx.x; d < H * DS; d += blockDim.x) qrow[d] = prow[d];
  for (int d = threadIdx.x; d < H * PQ * 3; d += blockDim.x) qrow[H * DS + d] = prow[off_gq + d];
  __syncthreads();
  const float* erow = e + row_i * K * C;  // e[b, i, :, :]
  const float scale_s = 1.0f / sqrtf(static_cast<float>(DS));
  const float scale_p = -0.5f / sqrtf(4.5f * PQ);
  const float scale_t = 1.0f / sqrtf(C > 0 ? 3.0f : 2.0f);  // num_independent_logits^-1/2: 3 with the pair bias, 2 without (C == 0, :385-387)
  for (int idx = threadIdx.x; idx < H * K; idx += blockDim.x) {
    const int h = idx / K, j = idx % K;
    const float* krow = proj + (static_cast<int64_t>(b) * K + j) * NP;
    float ls = 0.f, lb = 0.f, lp = 0.f;
    for (int d = 0; d < DS; ++d) ls += qrow[h * DS + d] * krow[off_ks + h * DS + d];
    for (int c = 0; c < C; ++c) lb += erow[static_cast<int64_t>(j) * C + c] * Wb[h * C + c];
    for (int p = 0; p < PQ * 3; ++p) {
      const float dd = qrow[H * DS + h * PQ * 3 + p] - krow[off_gk + h * PQ * 3 + p];
      lp += dd * dd;
    }
    attn[idx] = scale_t * ((ls * scale_s + lb) + (scale_p * gamma[h]) * lp);
  }
  __syncthreads();
  // softmax over j, one wave per head (round-robin)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwave = blockDim.x >> 6;
  for (int h = wave; h < H; h += nwave) {
    float m = -INFINITY;
    for (int j = lane; j < K; j += 64) m = fmaxf(m, attn[h * K + j]);
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    float s = 0.f;
    for (int j = lane; j < K; j += 64) {
      const float ex = expf(attn[h * K + j] - m);
      attn[h * K + j] = ex;
      s += ex;
    }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    const float inv = 1.0f / s;
    for (int j = lane; j < K; j += 64) attn[h * K + j] *= inv;
  }
  __syncthreads();
  float* frow = feat + row_i * F;
  const int n_os = H * DS, n_oe = H * C, n_og = H * PV * 3;
  for (int o = threadIdx.x; o < n_os + n_oe + n_og; o += blockDim.x) {
    float acc = 0.f;
    if (o < n_os) {
      const int h = o / DS;
      for (int j = 0; j < K; ++j) acc += attn[h * K + j] * proj[(static_cast<int64_t>(b) * K + j) * NP + off_vs + o];
      frow[o] = acc;
    } else if (o < n_os + n_oe) {
      const int oo = o - n_os, h = oo / C, c = oo % C;
      for (int j = 0; j < K; ++j) acc += attn[h * K + j] * erow[static_cast<int64_t>(j) * C + c];
      frow[o] = acc;
    } else {
      const int oo = o - n_os - n_oe, h = oo / (PV * 3);
      for (int j = 0; j < K; ++j) acc += attn[h * K + j] * proj[(static_cast<int64_t>(b) * K + j) * NP + off_gv + oo];
      og[oo] = acc;
    }
  }
  __syncthreads();
  // global -> local: (p - t) R^T, then norms (diffab_pytorch.py:336, :454)
  const float* Rr = R + row_i * 9;
  const float* tr = t + row_i * 3;
  for (int hp = threadIdx.x; hp < H * PV; hp += blockDim.x) {
    const float dx = og[hp * 3 + 0] - tr[0], dy = og[hp * 3 + 1] - tr[1], dz = og[hp * 3 + 2] - tr[2];
    const float lx = dx * Rr[0] + dy * Rr[1] + dz * Rr[2];
    const float ly = dx * Rr[3] + dy * Rr[4] + dz * Rr[5];
    const float lz = dx * Rr[6] + dy * Rr[7] + dz * Rr[8];
    frow[n_os + n_oe + hp * 3 + 0] = lx;
    frow[n_os + n_oe + hp * 3 + 1] = ly;
    frow[n_os + n_oe + hp * 3 + 2] = lz;
    frow[n_os + n_oe + n_og + hp] = sqrtf(lx * lx + ly * ly + lz * lz);
  }
}

size_t ipa_generic_workspace_floats(const diffab_dims* d) {
  const size_t rows = static_cast<size_t>(d->B) * d->K;
  const size_t NP = 3 * d->H * d->DS + 2 * d->H * d->PQ * 3 + d->H * d->PV * 3;
  const size_t F = d->H * d->DS + d->H * d->C + d->H * d->PV * 3 + d->H * d->PV;
  return rows * (NP + F) + 128;
}

int ipa_layer_generic(const diffab_dims* d, const diffab_ipa_layer_weights* w, const float* x, const float* e, const float* R, const float* t,
                      float* y, float* ws, hipStream_t st) {
  const int rows = d->B * d->K;
  const int H = d->H, DS = d->DS, PQ = d->PQ, PV = d->PV, D = d->D, C = d->C;
  const int NP = 3 * H * DS + 2 * H * PQ * 3 + H * PV * 3;
  const int F = H * DS + H * C + H * PV * 3 + H * PV;
  float* proj = ws;
  float* feat = ws + static_cast<size_t>(rows) * NP;
  int col = 0;
  const float* Ws[6] = {w->wq_s, w->wk_s, w->wv_s, w->wq_p, w->wk_p, w->wv_p};
  const int Ns[6] = {H * DS, H * DS, H * DS, H * PQ * 3, H * PQ * 3, H * PV * 3};
  for (int q = 0; q < 6; ++q) {
    if (int rc = launch_linear_generic(x, D, Ws[q], nullptr, proj + col, NP, rows, Ns[q], D, false, st)) return rc;
    if (q >= 3) {
      const int npts = Ns[q] / 3;
      const int64_t n = static_cast<int64_t>(rows) * npts;
      hipLaunchKernelGGL(points_to_global_kernel, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, st, proj, NP, col, npts, R, t,
                         static_cast<int64_t>(rows));
      DIFFAB_LAUNCH_CHECK();
    }
    col += Ns[q];
  }
  const size_t lds = (static_cast<size_t>(H) * d->K + H * DS + H * PQ * 3 + H * PV * 3) * sizeof(float);
  DIFFAB_REQUIRE(lds <= 160 * 1024, DIFFAB_ERR_UNSUPPORTED, "generic attention: H*K too large for LDS (%zu bytes)", lds);
  if (lds > 64 * 1024)
    DIFFAB_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(ipa_attn_generic_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                         static_cast<int>(lds)));
  timer_begin(st);
  hipLaunchKernelGGL(ipa_attn_generic_kernel, dim3(rows), dim3(256), lds, st, proj, e, R, t, w->w_bias, w->gamma, feat, d->K, C, H, DS, PQ, PV);
  timer_end(st);
  DIFFAB_LAUNCH_CHECK();
  return launch_linear_generic(feat, F, w->w_out, w->b_out, y, D, rows, D, F, false, st);
}

int launch_embed_concat(const float* res_ctx, const float* emb, const int64_t* seq, int D, int64_t rows, float* out, hipStream_t st) {
  hipLaunchKernelGGL(embed_concat_kernel, dim3(static_cast<unsigned>(rows)), dim3(128), 0, st, res_ctx, emb, seq, D, rows, out);
  DIFFAB_LAUNCH_CHECK();
  return DIFFAB_OK;
}
int launch_beta_concat(const float* h, const float* beta, int D, int K, int64_t rows, float* out, hipStream_t st) {
  hipLaunchKernelGGL(beta_concat_kernel, dim3(static_cast<unsigned>(rows)), dim3(128), 0, st, h, beta, D, K, rows, out);
  DIFFAB_LAUNCH_CHECK();
  return DIFFAB_OK;
}

}  // namespace diffab
