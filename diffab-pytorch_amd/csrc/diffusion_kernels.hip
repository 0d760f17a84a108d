// diffusion_kernels.hip - forward/reverse diffusion, SO(3)/IGSO3 and loss kernels + their C-ABI entries.
// All of these are per-residue elementwise work (KBs per patch): one thread per residue (or per bin),
// coalesced loads, no LDS except the loss and CDF reductions.  Compiled with -ffp-contract=off so that
// a*x + b*y rounds like the reference's separate ATen ops.
#include "common.h"
#include "denoiser_internal.h"
#include "philox.h"
#include "so3_math.h"

namespace diffab {

constexpr int kThreads = 256;
static inline int blocks_for(int64_t n) { return static_cast<int>((n + kThreads - 1) / kThreads); }
constexpr float kPiF = 3.14159265358979323846f;
constexpr double kPiD = 3.14159265358979323846;

// Python-style negative indexing of the schedule (reference indexes sched[...][t - 1] with t = 0 -> last entry).
__device__ inline int sched_index(int64_t t, int T) { return static_cast<int>(t < 0 ? t + (T + 1) : t); }

// ------------------------------------------------------------------ SO(3) maps
__global__ void so3_log_kernel(const float* __restrict__ R, float* __restrict__ S, int64_t n) {
  const int64_t i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x;
  if (i >= n) return;
  float r[9], s[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) r[k] = R[i * 9 + k];
  so3_log(r, s);
#pragma unroll
  for (int k = 0; k < 9; ++k) S[i * 9 + k] = s[k];
}

__global__ void so3_exp_kernel(const float* __restrict__ S, float* __restrict__ R, int64_t n) {
  const int64_t i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x;
  if (i >= n) return;
  float s[9], r[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) s[k] = S[i * 9 + k];
  so3_exp(s, r);
#pragma unroll
  for (int k = 0; k < 9; ++k) R[i * 9 + k] = r[k];
}

__global__ void so3_matrix_to_rotvec_kernel(const float* __restrict__ R, float* __restrict__ v, int64_t n) {
  const int64_t i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x;
  if (i >= n) return;
  float r[9], s[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) r[k] = R[i * 9 + k];
  so3_log(r, s);
  v[i * 3 + 0] = s[7];
  v[i * 3 + 1] = s[2];
  v[i * 3 + 2] = s[3];
}

__global__ void so3_rotvec_to_matrix_kernel(const float* __restrict__ v, float* __restrict__ R, int64_t n) {
  const int64_t i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x;
  if (i >= n) return;
  float r[9];
  so3_rotvec_to_matrix(v[i * 3 + 0], v[i * 3 + 1], v[i * 3 + 2], r);
#pragma unroll
  for (int k = 0; k < 9; ++k) R[i * 9 + k] = r[k];
}

__global__ void so3_scale_rot_kernel(const float* __restrict__ R, const float* __restrict__ kk, float* __restrict__ out, int64_t n,
                                     int64_t per_k) {
  const int64_t i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x;
  if (i >= n) return;
  float r[9], o[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) r[k] = R[i * 9 + k];
  so3_scale(r, kk[i / per_k], o);
#pragma unroll
  for (int k = 0; k < 9; ++k) out[i * 9 + k] = o[k];
}

// ------------------------------------------------------------------ IGSO3 tables
// pdf(theta; sigma) = (1 - cos theta)/pi * sum_l (2l+1) exp(-l(l+1) sigma^2) sin((l+1/2) theta)/sin(theta/2)
// One thread per (sigma, bin).
//
// FAITHFUL (the table DiffAb uses): the reference's table is NOT the exact density - it is the density plus the rounding noise of
// its fp32 evaluation (so3.py:65-72), e.g. a = (1 - cos theta)/pi loses 2e-5 relative at the peak of row 1 to the rounding of
// cos theta, and the clamp turns the +-1e-3 noise of the cancelling tail into ~1e-4 of spurious probability mass on the small-sigma
// rows.  That noise is part of the distribution the reference samples from, so it is reproduced, operation by operation:
//   * every arithmetic step rounded to fp32 in the reference's order: theta, a, b = (2l+1) * exp(fl(-(l(l+1))) * sigma^2),
//     c = sin((l + 0.5) * theta) / sin(theta / 2), term = (a * b) * c;
//   * cos / sin / exp correctly rounded (float64 evaluation, rounded once): torch's SLEEF kernels are correctly rounded on ~95 %
//     of these arguments, so ~95 % of the evaluations are bit-identical and the rest differ by one ulp;
//   * the sum over l in the order of torch's fp32 cascade sum for a reduction over the outer dimension (ATen SumKernel.cpp,
//     multi_row_sum: 4 levels, level step max(16, 2^(ceil_log2(n)/4)) terms).
// ACCURATE (opt-in): the whole series in float64, rounded once - the true density to fp32 accuracy.
template <bool FAITHFUL>
__global__ void igso3_pdf_kernel(const float* __restrict__ sigmas, int n_sigmas, int n_bins, int num_iters, float* __restrict__ pdf) {
  const int64_t gid = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x;
  if (gid >= static_cast<int64_t>(n_sigmas) * n_bins) return;
  const int s = static_cast<int>(gid / n_bins), m = static_cast<int>(gid % n_bins);
  const double width = kPiD / n_bins;
  const float theta_f = static_cast<float>(m * width) + static_cast<float>(width / 2.0);  // arange(0, pi, w) + w / 2 (so3.py:57)
  float v;
  if (FAITHFUL) {
    auto cr_cos = [](float x) { return static_cast<float>(cos(static_cast<double>(x))); };
    auto cr_sin = [](float x) { return static_cast<float>(sin(static_cast<double>(x))); };
    auto cr_exp = [](float x) { return static_cast<float>(exp(static_cast<double>(x))); };
    const float sg2 = sigmas[s] * sigmas[s];
    const float a = (1.0f - cr_cos(theta_f)) / kPiF;
    const float sh = cr_sin(theta_f / 2.0f);
    auto term = [&](int l) {
      const float b = static_cast<float>(2 * l + 1) * cr_exp(static_cast<float>(-static_cast<int64_t>(l) * (l + 1)) * sg2);
      const float c = cr_sin((static_cast<float>(l) + 0.5f) * theta_f) / sh;
      return (a * b) * c;
    };
    int lg = 0;
    while ((1 << lg) < num_iters) ++lg;  // ceil_log2
    const int level_power = lg / 4 > 4 ? lg / 4 : 4, level_step = 1 << level_power, level_mask = level_step - 1;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    int i = 0;
    while (i + level_step <= num_iters) {
      for (int j = 0; j < level_step; ++j, ++i) acc[0] += term(i);
      for (int j = 1; j < 4; ++j) {
        acc[j] += acc[j - 1];
        acc[j - 1] = 0.f;
        if ((i & (level_mask << (j * level_power))) != 0) break;
      }
    }
    for (; i < num_iters; ++i) acc[0] += term(i);
    for (int j = 1; j < 4; ++j) acc[0] += acc[j];
    v = acc[0];
  } else {
    const double theta = static_cast<double>(theta_f);
    const double sg = static_cast<double>(sigmas[s]);
    const double a = (1.0 - cos(theta)) / kPiD;
    const double inv_sh = 1.0 / sin(theta / 2.0);
    double acc = 0.0;
    for (int l = 0; l < num_iters; ++l) {
      const double ld = static_cast<double>(l);
      acc += (2.0 * ld + 1.0) * exp(-ld * (ld + 1.0) * sg * sg) * sin((ld + 0.5) * theta) * inv_sh;
    }
    v = static_cast<float>(a * acc);
  }
  if (!(v == v)) v = 0.0f;           // nan_to_num (so3.py:61)
  if (isinf(v)) v = v > 0 ? 3.4028234663852886e38f : -3.4028234663852886e38f;
  pdf[gid] = v < 0.0f ? 0.0f : v;    // clamp_min(0)
}

// One block per row: float64 inclusive prefix sum, normalised, last entry = 1.
__global__ void igso3_cdf_kernel(const float* __restrict__ pdf, int n_bins, float* __restrict__ cdf) {
  __shared__ double part[kThreads];
  const int row = blockIdx.x, tid = threadIdx.x;
  const float* p = pdf + static_cast<int64_t>(row) * n_bins;
  float* c = cdf + static_cast<int64_t>(row) * n_bins;
  const int per = (n_bins + kThreads - 1) / kThreads;
  const int lo = tid * per, hi = min(lo + per, n_bins);
  double s = 0.0;
  for (int i = lo; i < hi; ++i) s += static_cast<double>(p[i]);
  part[tid] = s;
  __syncthreads();
  if (tid == 0) {
    double run = 0.0;
    for (int i = 0; i < kThreads; ++i) {
      const double v = part[i];
      part[i] = run;
      run += v;
    }
  }
  __syncthreads();
  __shared__ double total;
  if (tid == kThreads - 1) total = part[tid] + s;
  __syncthreads();
  double run = part[tid];
  const double inv = 1.0 / total;
  for (int i = lo; i < hi; ++i) {
    run += static_cast<double>(p[i]);
    c[i] = (i == n_bins - 1) ? 1.0f : static_cast<float>(run * inv);
  }
}

// first m with cdf[m] > u  (searchsorted right=True), clamped to n_bins-1
__device__ inline int cdf_search(const float* __restrict__ row, int n_bins, float u) {
  int lo = 0, hi = n_bins;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (row[mid] > u) hi = mid; else lo = mid + 1;
  }
  return min(lo, n_bins - 1);
}

__device__ inline float floor_mod_pi(float x) {
  float r = fmodf(x, kPiF);
  if (r != 0.0f && r < 0.0f) r += kPiF;
  return r;
}

// theta for one draw (so3.py:74-96, 118-125)
__device__ inline float igso3_theta(const float* __restrict__ cdf, int n_bins, float sigma, float thr, int row, float u_bin, float u_in,
                                    float z) {
  if (sigma < thr) {
    const int m = cdf_search(cdf + static_cast<int64_t>(row) * n_bins, n_bins, u_bin);
    const double width = kPiD / n_bins;
    return static_cast<float>(m * width) + static_cast<float>(width) * u_in;
  }
  return floor_mod_pi(sigma * 2.0f + sigma * z);
}

__device__ inline void normalize3(float& x, float& y, float& z) {
  const float n = fmaxf(sqrtf(x * x + y * y + z * z), 1e-12f);  // F.normalize eps (so3.py:114)
  x /= n; y /= n; z /= n;
}

__global__ void igso3_sample_kernel(const float* __restrict__ sigmas, const float* __restrict__ cdf, int n_bins, float thr,
                                    const int64_t* __restrict__ sigma_idx, int B, int K, const float* __restrict__ axis_raw,
                                    const float* __restrict__ u_bin, const float* __restrict__ u_in, const float* __restrict__ z,
                                    float* __restrict__ rotvec) {
  const int64_t i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x;
  if (i >= static_cast<int64_t>(B) * K) return;
  const int b = static_cast<int>(i / K);
  const int row = static_cast<int>(sigma_idx[b]);
  const float theta = igso3_theta(cdf, n_bins, sigmas[row], thr, row, u_bin[i], u_in[i], z[i]);
  float x = axis_raw[i * 3 + 0], y = axis_raw[i * 3 + 1], zz = axis_raw[i * 3 + 2];
  normalize3(x, y, zz);
  rotvec[i * 3 + 0] = x * theta;
  rotvec[i * 3 + 1] = y * theta;
  rotvec[i * 3 + 2] = zz * theta;
}

// so3.py:78: `torch.multinomial(probs, num_samples)` draws the K bins of ONE patch WITHOUT replacement (its default).  The exponential
// race is that distribution exactly (and is how torch itself draws it on a GPU): with E_b ~ Exp(1) independent, the bins ordered by
// key_b = p_b / E_b, largest first, are a sample without replacement in draw order.  One work-group per patch: the n_bins keys of the
// patch's sigma row go through a bitonic sort in LDS (order: larger key first, equal keys by lower bin index - a total order, so the
// result does not depend on the network), the first K bin indices are the draws.  Bins of zero mass have key 0 and can only be drawn
// when the row has fewer than K bins of positive mass (the reference raises there).
__global__ __launch_bounds__(1024) void igso3_race_kernel(const float* __restrict__ pdf, int n_bins, int n_pad,
                                                          const int64_t* __restrict__ sigma_idx, int K, const float* __restrict__ race,
                                                          int32_t* __restrict__ bins, const float* __restrict__ sigmas, float sigma_thr) {
  extern __shared__ float race_lds[];
  // rows whose sigma is not below the threshold take the Gaussian angle (so3.py:122-125): their bins are never read - no sort (with
  // the T = 100 schedule that is every row with t > 5: 95 % of a training batch); uniform exit, before any barrier
  if (sigmas != nullptr && !(sigmas[sigma_idx[blockIdx.x]] < sigma_thr)) {
    for (int k = threadIdx.x; k < K; k += blockDim.x) bins[static_cast<int64_t>(blockIdx.x) * K + k] = 0;
    return;
  }
  float* key = race_lds;                                    // [n_pad]
  int* idx = reinterpret_cast<int*>(race_lds + n_pad);      // [n_pad]
  const int b = blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
  const float* prow = pdf + static_cast<int64_t>(sigma_idx[b]) * n_bins;
  const float* erow = race + static_cast<int64_t>(b) * n_bins;
  for (int i = tid; i < n_pad; i += nt) {
    // IEEE division (no fast-math): the host restatement gets the same bits.  NaN-free for any caller-supplied race: a zero-mass bin has
    // key 0 whatever its draw, and a draw of exactly 0 (u = 1; the library's own uniforms exclude it, philox.h) counts as the smallest
    // positive float - a NaN key would break the total order the bitonic network relies on.
    const float p_ = i < n_bins ? prow[i] : 0.0f;
    key[i] = i < n_bins ? (p_ > 0.0f ? p_ / fmaxf(erow[i], 1.17549435e-38f) : 0.0f) : -1.0f;
    idx[i] = i;
  }
  __syncthreads();
  for (int k = 2; k <= n_pad; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = tid; i < n_pad; i += nt) {
        const int l = i ^ j;
        if (l > i) {
          const float ka = key[i], kb = key[l];
          const int ia = idx[i], ib = idx[l];
          const bool a_first = ka > kb || (ka == kb && ia < ib);  // a sorts in front of b
          const bool asc = (i & k) == 0;                          // this segment puts its "first" elements at the low indices
          if (a_first != asc) {
            key[i] = kb; key[l] = ka;
            idx[i] = ib; idx[l] = ia;
          }
        }
      }
      __syncthreads();
    }
  for (int r = tid; r < K; r += nt) bins[static_cast<int64_t>(b) * K + r] = idx[r];
}

// igso3_sample_kernel with the histogram bin given (drawn by igso3_race_kernel) instead of found by inverse CDF
__global__ void igso3_sample_bins_kernel(const float* __restrict__ sigmas, int n_bins, float thr, const int64_t* __restrict__ sigma_idx,
                                         int B, int K, const float* __restrict__ axis_raw, const int32_t* __restrict__ bins,
                                         const float* __restrict__ u_in, const float* __restrict__ z, float* __restrict__ rotvec) {
  const int64_t i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x;
  if (i >= static_cast<int64_t>(B) * K) return;
  const float sigma = sigmas[sigma_idx[i / K]];
  float theta;
  if (sigma < thr) {
    const double width = kPiD / n_bins;
    theta = static_cast<float>(bins[i] * width) + static_cast<float>(width) * u_in[i];
  } else {
    theta = floor_mod_pi(sigma * 2.0f + sigma * z[i]);
  }
  float x = axis_raw[i * 3 + 0], y = axis_raw[i * 3 + 1], zz = axis_raw[i * 3 + 2];
  normalize3(x, y, zz);
  rotvec[i * 3 + 0] = x * theta;
  rotvec[i * 3 + 1] = y * theta;
  rotvec[i * 3 + 2] = zz * theta;
}

// ------------------------------------------------------------------ sequence diffusion
constexpr int kV = 21;  // diffusion.py:47

// centre weight w on the one-hot, (1-w)/21 uniform; exact one-hot when not generated.
__device__ inline float seq_prob(int v, int64_t centre, float w_keep, float w_noise, bool gen) {
  const float oh = (v == centre) ? 1.0f : 0.0f;
  return gen ? (w_keep * oh + w_noise * (1.0f / 21.0f)) : oh;
}

// w1[b] p1 + w2[b] p2 over the trailing (K, V) elements of patch b (diffusion.py:38-41; separate mul / add roundings, as ATen)
__global__ void weighted_multinomial_kernel(const float* __restrict__ p1, const float* __restrict__ p2, const float* __restrict__ w1,
                                            const float* __restrict__ w2, int64_t n, int64_t per_b, float* __restrict__ out) {
  const int64_t i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x;
  if (i >= n) return;
  const int64_t b = i / per_b;
  out[i] = w1[b] * p1[i] + w2[b] * p2[i];
}

__global__ void seq_forward_prob_kernel(const float* __restrict__ beta, const float* __restrict__ alpha_bar, int T, int mode,
                                        const int64_t* __restrict__ seq, const int64_t* __restrict__ t, const uint8_t* __restrict__ mask,
                                        int B, int K, float* __restrict__ prob) {
  const int64_t i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x;
  if (i >= static_cast<int64_t>(B) * K) return;
  const int ti = sched_index(t[i / K], T);
  float wk, wn;
  if (mode == 0) { wn = beta[ti]; wk = 1.0f - wn; } else { wk = alpha_bar[ti]; wn = 1.0f - wk; }
  const int64_t c = seq[i];
  const bool g = mask[i] != 0;
  for (int v = 0; v < kV; ++v) prob[i * kV + v] = seq_prob(v, c, wk, wn, g);
}

__global__ void seq_posterior_kernel(const float* __restrict__ beta, const float* __restrict__ alpha_bar, int T,
                                     const int64_t* __restrict__ seq_t, const int64_t* __restrict__ seq_0, const int64_t* __restrict__ t,
                                     const uint8_t* __restrict__ mask, int B, int K, float* __restrict__ post) {
  const int64_t i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x;
  if (i >= static_cast<int64_t>(B) * K) return;
  const int64_t tt = t[i / K];
  const float b = beta[sched_index(tt, T)];
  const float ab = alpha_bar[sched_index(tt - 1, T)];
  const int64_t ct = seq_t[i], c0 = seq_0[i];
  const bool g = mask[i] != 0;
  float p[kV];
  float s = 0.0f;
  for (int v = 0; v < kV; ++v) {
    p[v] = seq_prob(v, ct, 1.0f - b, b, g) * seq_prob(v, c0, ab, 1.0f - ab, g);
    s += p[v];
  }
  for (int v = 0; v < kV; ++v) post[i * kV + v] = p[v] / s;
}

__device__ inline int categorical_draw(const float* __restrict__ p, int V, float u) {
  float tot = 0.0f;
  for (int v = 0; v < V; ++v) tot += p[v];
  const float thr = u * tot;
  float acc = 0.0f;
  for (int v = 0; v < V; ++v) {
    acc += p[v];
    if (acc > thr) return v;
  }
  return V - 1;
}

__global__ void categorical_sample_kernel(const float* __restrict__ prob, const float* __restrict__ u, int64_t n, int V,
                                          int64_t* __restrict__ out) {
  const int64_t i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x;
  if (i >= n) return;
  out[i] = categorical_draw(prob + i * V, V, u[i]);
}

// ------------------------------------------------------------------ translation / orientation forward
__global__ void coord_forward_kernel(const float* __restrict__ abs_, const float* __restrict__ omabs, int T, const float* __restrict__ x0,
                                     const int64_t* __restrict__ t, const uint8_t* __restrict__ mask, const float* __restrict__ eps, int B,
                                     int K, float* __restrict__ xt) {
  const int64_t i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x;
  if (i >= static_cast<int64_t>(B) * K) return;
  const int ti = sched_index(t[i / K], T);
  const float a = abs_[ti], b = omabs[ti];
  const bool g = mask[i] != 0;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float x = x0[i * 3 + c];
    xt[i * 3 + c] = g ? (a * x + b * eps[i * 3 + c]) : x;
  }
}

__global__ void orient_forward_kernel(const float* __restrict__ abs_, int T, const float* __restrict__ O0, const uint8_t* __restrict__ mask,
                                      const int64_t* __restrict__ t, const float* __restrict__ rotvec, int B, int K, float* __restrict__ Ot) {
  const int64_t i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x;
  if (i >= static_cast<int64_t>(B) * K) return;
  float r[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) r[k] = O0[i * 9 + k];
  if (mask[i] != 0) {
    float mean[9], noise[9], o[9];
    so3_scale(r, abs_[sched_index(t[i / K], T)], mean);
    so3_rotvec_to_matrix(rotvec[i * 3 + 0], rotvec[i * 3 + 1], rotvec[i * 3 + 2], noise);
    mat3_mul(mean, noise, o);
#pragma unroll
    for (int k = 0; k < 9; ++k) r[k] = o[k];
  }
#pragma unroll
  for (int k = 0; k < 9; ++k) Ot[i * 9 + k] = r[k];
}

// ------------------------------------------------------------------ Philox fill (tests / explicit-noise callers)
__global__ void philox_fill_kernel(uint64_t seed, int64_t first_patch, int B, int K, int step, int stream_id, int kind,
                                   float* __restrict__ out) {
  const int64_t i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x;
  if (i >= static_cast<int64_t>(B) * K) return;
  const uint32_t patch = static_cast<uint32_t>(first_patch + i / K), res = static_cast<uint32_t>(i % K);
  const f32x4 u = philox_uniform4(seed, patch, res, static_cast<uint32_t>(step), static_cast<uint32_t>(stream_id));
  const f32x4 r = kind == 0 ? normals_from_uniforms(u) : u;
  reinterpret_cast<float4*>(out)[i] = make_float4(r.x, r.y, r.z, r.w);
}

// ------------------------------------------------------------------ losses
// One block; deterministic tree reduction (diffab_pytorch.py:856-880).
__global__ void losses_kernel(const float* __restrict__ pp, const float* __restrict__ tp, const float* __restrict__ pe,
                              const float* __restrict__ te, const float* __restrict__ pO, const float* __restrict__ tO,
                              const uint8_t* __restrict__ gm, const uint8_t* __restrict__ rm, int64_t n, int V, float* __restrict__ out3) {
  __shared__ float red[4][1024];
  float a_kl = 0.f, a_mse = 0.f, a_o = 0.f, a_n = 0.f;
  for (int64_t i = threadIdx.x; i < n; i += blockDim.x) {
    if (!(gm[i] && rm[i])) continue;
    a_n += 1.0f;
    for (int v = 0; v < V; ++v) {
      const float q = tp[i * V + v];
      if (q > 0.0f) a_kl += q * logf(q) - q * logf(pp[i * V + v]);  // kl_div(log p, q): xlogy(q,q) - q log p
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float d = pe[i * 3 + c] - te[i * 3 + c];
      a_mse += d * d;
    }
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        float d = 0.f;
#pragma unroll
        for (int r = 0; r < 3; ++r) d += pO[i * 9 + r * 3 + j] * tO[i * 9 + r * 3 + k];  // pred^T target (:620-622)
        d -= (j == k) ? 1.0f : 0.0f;
        a_o += d * d;
      }
  }
  red[0][threadIdx.x] = a_kl; red[1][threadIdx.x] = a_mse; red[2][threadIdx.x] = a_o; red[3][threadIdx.x] = a_n;
  __syncthreads();
  for (int s = blockDim.x / 2; s > 0; s >>= 1) {
    if (static_cast<int>(threadIdx.x) < s)
      for (int q = 0; q < 4; ++q) red[q][threadIdx.x] += red[q][threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x < 3) out3[threadIdx.x] = red[threadIdx.x][0] / red[3][0];
}

// Two-stage form for the training step (B K = 16 k rows, ~21 logf per row: one work-group serialises 0.2 ms of it): kLossBlocks
// work-groups reduce contiguous row chunks to partial[block][4] (fixed tree order inside a block), one work-group adds the partials
// in block order - deterministic, and independent of how rows are spread over compute units.
constexpr int kLossBlocks = 256;
__global__ __launch_bounds__(256) void losses_partial_kernel(const float* __restrict__ pp, const float* __restrict__ tp,
                                                             const float* __restrict__ pe, const float* __restrict__ te,
                                                             const float* __restrict__ pO, const float* __restrict__ tO,
                                                             const uint8_t* __restrict__ gm, const uint8_t* __restrict__ rm, int64_t n, int V,
                                                             float* __restrict__ partial) {
  __shared__ float red[4][256];
  const int64_t chunk = (n + gridDim.x - 1) / gridDim.x;
  const int64_t lo = blockIdx.x * chunk, hi = lo + chunk < n ? lo + chunk : n;
  float a_kl = 0.f, a_mse = 0.f, a_o = 0.f, a_n = 0.f;
  for (int64_t i = lo + threadIdx.x; i < hi; i += blockDim.x) {
    if (!(gm[i] && rm[i])) continue;
    a_n += 1.0f;
    for (int v = 0; v < V; ++v) {
      const float q = tp[i * V + v];
      if (q > 0.0f) a_kl += q * logf(q) - q * logf(pp[i * V + v]);
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float d = pe[i * 3 + c] - te[i * 3 + c];
      a_mse += d * d;
    }
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        float d = 0.f;
#pragma unroll
        for (int r = 0; r < 3; ++r) d += pO[i * 9 + r * 3 + j] * tO[i * 9 + r * 3 + k];
        d -= (j == k) ? 1.0f : 0.0f;
        a_o += d * d;
      }
  }
  red[0][threadIdx.x] = a_kl; red[1][threadIdx.x] = a_mse; red[2][threadIdx.x] = a_o; red[3][threadIdx.x] = a_n;
  __syncthreads();
  for (int s = blockDim.x / 2; s > 0; s >>= 1) {
    if (static_cast<int>(threadIdx.x) < s)
      for (int q = 0; q < 4; ++q) red[q][threadIdx.x] += red[q][threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x < 4) partial[blockIdx.x * 4 + threadIdx.x] = red[threadIdx.x][0];
}
__global__ __launch_bounds__(256) void losses_final_kernel(const float* __restrict__ partial, int nblocks, float* __restrict__ out3) {
  __shared__ float red[4][256];
  for (int q = 0; q < 4; ++q) red[q][threadIdx.x] = static_cast<int>(threadIdx.x) < nblocks ? partial[threadIdx.x * 4 + q] : 0.0f;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (static_cast<int>(threadIdx.x) < s)
      for (int q = 0; q < 4; ++q) red[q][threadIdx.x] += red[q][threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x < 3) out3[threadIdx.x] = red[threadIdx.x][0] / red[3][0];
}

// (pred^T target - I)^2 element-wise (+ optional total), diffab_pytorch.py:610-625.  One block, fixed-order reduction.
__global__ void orientation_loss_kernel(const float* __restrict__ pO, const float* __restrict__ tO, int64_t n, float* __restrict__ elems,
                                        float* __restrict__ sum1) {
  __shared__ float red[1024];
  float acc = 0.f;
  for (int64_t i = threadIdx.x; i < n; i += blockDim.x) {
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        float d = 0.f;
#pragma unroll
        for (int r = 0; r < 3; ++r) d += pO[i * 9 + r * 3 + j] * tO[i * 9 + r * 3 + k];
        d -= (j == k) ? 1.0f : 0.0f;
        if (elems) elems[i * 9 + j * 3 + k] = d * d;
        acc += d * d;
      }
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int s = blockDim.x / 2; s > 0; s >>= 1) {
    if (static_cast<int>(threadIdx.x) < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0 && sum1) sum1[0] = red[0];
}

// d pred / d target of the same loss: D = pred^T target - I, L_jk = D_jk^2 with cotangent c_jk (g_elems, or the scalar *g_total for
// every element); d pred[i][j] = sum_k 2 c_jk D_jk target[i][k], d target[i][k] = sum_j 2 c_jk D_jk pred[i][j]
__global__ void orientation_loss_bwd_kernel(const float* __restrict__ pO, const float* __restrict__ tO, int64_t n,
                                            const float* __restrict__ g_elems, const float* __restrict__ g_total,
                                            float* __restrict__ d_pred, float* __restrict__ d_target) {
  const int64_t i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x;
  if (i >= n) return;
  const float gt = g_total ? g_total[0] : 0.0f;
  float P[9], T[9], G[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) { P[k] = pO[i * 9 + k]; T[k] = tO[i * 9 + k]; }
#pragma unroll
  for (int j = 0; j < 3; ++j)
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      float d = 0.f;
#pragma unroll
      for (int r = 0; r < 3; ++r) d += P[r * 3 + j] * T[r * 3 + k];
      d -= (j == k) ? 1.0f : 0.0f;
      G[j * 3 + k] = 2.0f * d * (g_elems ? g_elems[i * 9 + j * 3 + k] : gt);
    }
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      if (d_pred) {  // d pred[r][c] = sum_k G[c][k] T[r][k]
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 3; ++k) s += G[c * 3 + k] * T[r * 3 + k];
        d_pred[i * 9 + r * 3 + c] = s;
      }
      if (d_target) {  // d target[r][c] = sum_j G[j][c] P[r][j]
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < 3; ++j) s += G[j * 3 + c] * P[r * 3 + j];
        d_target[i * 9 + r * 3 + c] = s;
      }
    }
}

// ------------------------------------------------------------------ rigid frames applied to points (diffab_pytorch.py:315-336)
// x (B, N, L, P, 3) points of N heads, R (B, L, 3, 3), t (B, L, 3) broadcast over the heads; row-vector convention.
//   INVERT = false: out = x R + t        (euclidean_transform, :315-324)
//   INVERT = true:  out = (x - t) R^T    (inverse_euclidean_transform, :327-336)
// t == nullptr: the rotation alone (the x-gradient of the other direction).
template <bool INVERT>
__global__ void frames_kernel(const float* __restrict__ x, const float* __restrict__ R, const float* __restrict__ t, float* __restrict__ out,
                              int N, int L, int P, int64_t n_points) {
  const int64_t i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x;  // point (b, n, l, p)
  if (i >= n_points) return;
  const int64_t bnl = i / P;
  const int64_t l = bnl % L, b = bnl / (static_cast<int64_t>(L) * N);
  const float* Rr = R + (b * L + l) * 9;
  float tx = 0.f, ty = 0.f, tz = 0.f;
  if (t) {
    const float* tr = t + (b * L + l) * 3;
    tx = tr[0]; ty = tr[1]; tz = tr[2];
  }
  const float px = x[i * 3], py = x[i * 3 + 1], pz = x[i * 3 + 2];
  if (INVERT) {
    const float dx = px - tx, dy = py - ty, dz = pz - tz;
    out[i * 3 + 0] = (dx * Rr[0] + dy * Rr[1]) + dz * Rr[2];
    out[i * 3 + 1] = (dx * Rr[3] + dy * Rr[4]) + dz * Rr[5];
    out[i * 3 + 2] = (dx * Rr[6] + dy * Rr[7]) + dz * Rr[8];
  } else {
    out[i * 3 + 0] = ((px * Rr[0] + py * Rr[3]) + pz * Rr[6]) + tx;
    out[i * 3 + 1] = ((px * Rr[1] + py * Rr[4]) + pz * Rr[7]) + ty;
    out[i * 3 + 2] = ((px * Rr[2] + py * Rr[5]) + pz * Rr[8]) + tz;
  }
}

// d R, d t of the two frame maps from the cotangent g of their output (the reference's einsums are differentiable in r and t,
// diffab_pytorch.py:315-336); one work-group per (b, l), reduction over the heads and points that share the frame:
//   apply  (out = x R + t):      d R[k][c] = sum x[k] g[c],        d t = sum g
//   invert (out = (x - t) R^T):  d R[c][k] = sum g[c] (x - t)[k],  d t[k] = - sum_c g[c] R[c][k]
template <bool INVERT>
__global__ __launch_bounds__(256) void frames_bwd_kernel(const float* __restrict__ x, const float* __restrict__ R, const float* __restrict__ t,
                                                         const float* __restrict__ g, int N, int L, int P, float* __restrict__ dR,
                                                         float* __restrict__ dt) {
  const int64_t bl = blockIdx.x;
  const int64_t b = bl / L, l = bl % L;
  const float* Rr = R + bl * 9;
  const float tx = t[bl * 3], ty = t[bl * 3 + 1], tz = t[bl * 3 + 2];
  float acc[12];
#pragma unroll
  for (int k = 0; k < 12; ++k) acc[k] = 0.f;
  for (int idx = threadIdx.x; idx < N * P; idx += blockDim.x) {
    const int n = idx / P, p = idx % P;
    const int64_t i = ((b * N + n) * L + l) * P + p;
    const float gx = g[i * 3], gy = g[i * 3 + 1], gz = g[i * 3 + 2];
    const float gg[3] = {gx, gy, gz};
    if (INVERT) {
      const float d[3] = {x[i * 3] - tx, x[i * 3 + 1] - ty, x[i * 3 + 2] - tz};
#pragma unroll
      for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int k = 0; k < 3; ++k) acc[c * 3 + k] += gg[c] * d[k];
#pragma unroll
      for (int k = 0; k < 3; ++k) acc[9 + k] -= (gx * Rr[0 * 3 + k] + gy * Rr[1 * 3 + k]) + gz * Rr[2 * 3 + k];
    } else {
      const float xx[3] = {x[i * 3], x[i * 3 + 1], x[i * 3 + 2]};
#pragma unroll
      for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int c = 0; c < 3; ++c) acc[k * 3 + c] += xx[k] * gg[c];
      acc[9] += gx; acc[10] += gy; acc[11] += gz;
    }
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
    if (k < 9) { if (dR) dR[bl * 9 + k] = v; }
    else if (dt) dt[bl * 3 + (k - 9)] = v;
  }
}

// AngularEncoding (diffab_pytorch.py:20-54): per input element x -> [x, sin(f_i x) (2 nf values), cos(f_i x) (2 nf values)],
// f = [1, .., nf, 1/1, .., 1/nf] (the reference builds the band table in fp32: 1.0 / (i + 1.0) rounded once)
__global__ void angular_encoding_kernel(const float* __restrict__ x, int64_t n, int nf, float* __restrict__ out) {
  const int64_t i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x;
  if (i >= n) return;
  const float v = x[i];
  const int w = 4 * nf + 1;
  float* o = out + i * w;
  o[0] = v;
  for (int k = 0; k < 2 * nf; ++k) {
    const float f = k < nf ? static_cast<float>(k + 1) : 1.0f / static_cast<float>(k - nf + 1);
    const float a = f * v;
    o[1 + k] = sinf(a);
    o[1 + 2 * nf + k] = cosf(a);
  }
}

// ------------------------------------------------------------------ reverse update
__device__ inline void reverse_update_one(int64_t i, int t, float beta, float alpha, float omabs, int64_t* seq, float* x, float* O,
                                          const float* eps_hat, const float* O0_hat, const float* post, int V, float zx, float zy,
                                          float zz, float rx, float ry, float rz, float u_seq) {
  const float c = beta / omabs;
  const float sa = sqrtf(alpha), sb = sqrtf(beta);
  const float zn[3] = {zx, zy, zz};
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    float v = (x[i * 3 + k] - c * eps_hat[i * 3 + k]) / sa;
    if (t > 1) v = v + sb * zn[k];
    x[i * 3 + k] = v;
  }
  float o[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) o[k] = O0_hat[i * 9 + k];
  if (t > 1) {
    float nz[9], r[9];
    so3_rotvec_to_matrix(rx, ry, rz, nz);
    mat3_mul(o, nz, r);
#pragma unroll
    for (int k = 0; k < 9; ++k) o[k] = r[k];
  }
#pragma unroll
  for (int k = 0; k < 9; ++k) O[i * 9 + k] = o[k];
  seq[i] = categorical_draw(post + i * V, V, u_seq);
}

__global__ void reverse_update_kernel(const float* __restrict__ beta, const float* __restrict__ alpha, const float* __restrict__ omabs, int t,
                                      int64_t* __restrict__ seq, float* __restrict__ x, float* __restrict__ O,
                                      const float* __restrict__ eps_hat, const float* __restrict__ O0_hat, const float* __restrict__ post,
                                      const uint8_t* __restrict__ gm, const float* __restrict__ z, const float* __restrict__ rotvec,
                                      const float* __restrict__ u_seq, int B, int K, int V) {
  const int64_t i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x;
  if (i >= static_cast<int64_t>(B) * K || !gm[i]) return;
  reverse_update_one(i, t, beta[t], alpha[t], omabs[t], seq, x, O, eps_hat, O0_hat, post, V, z[i * 3], z[i * 3 + 1], z[i * 3 + 2],
                     rotvec[i * 3], rotvec[i * 3 + 1], rotvec[i * 3 + 2], u_seq[i]);
}

// Same update with the noise drawn in-kernel from Philox (the production sampler).
// head_v / head_logits != nullptr (the folded sampler path): the heads' epilogue of the row - O0 = O_t exp(hat(v)) (diffab_pytorch.py:594-596)
// and posterior = softmax(logits) (:555), what heads_finish_kernel computes for every row - is done here, for the generated rows only,
// into O0_hat / post (read back by the same thread below; no __restrict__ on the two for that reason): one launch less per step.
__global__ void reverse_update_philox_kernel(const float* __restrict__ beta, const float* __restrict__ alpha, const float* __restrict__ omabs,
                                             int t, const float* __restrict__ rev_sigmas, const float* __restrict__ rev_cdf, int n_bins,
                                             float thr, int64_t* __restrict__ seq, float* __restrict__ x, float* __restrict__ O,
                                             const float* __restrict__ eps_hat, float* O0_hat, float* post, const uint8_t* __restrict__ gm,
                                             uint64_t seed, int64_t first_patch, int B, int K, int V, const int* __restrict__ t_dev,
                                             const float* __restrict__ head_v, const float* __restrict__ head_logits) {
  const int64_t i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x;
  if (i >= static_cast<int64_t>(B) * K || !gm[i]) return;
  if (t_dev != nullptr) t = *t_dev;  // graph replay: the timestep lives in device memory (one captured step serves every t)
  if (head_v != nullptr) {
    float ex[9], o[9], res[9];
    so3_rotvec_to_matrix(head_v[i * 3], head_v[i * 3 + 1], head_v[i * 3 + 2], ex);
#pragma unroll
    for (int k = 0; k < 9; ++k) o[k] = O[i * 9 + k];
    mat3_mul(o, ex, res);
#pragma unroll
    for (int k = 0; k < 9; ++k) O0_hat[i * 9 + k] = res[k];
    float m = -INFINITY;
    for (int c = 0; c < V; ++c) m = fmaxf(m, head_logits[i * V + c]);
    float s = 0.f;
    for (int c = 0; c < V; ++c) s += expf(head_logits[i * V + c] - m);
    const float inv = 1.0f / s;
    for (int c = 0; c < V; ++c) post[i * V + c] = expf(head_logits[i * V + c] - m) * inv;
  }
  const uint32_t patch = static_cast<uint32_t>(first_patch + i / K), res = static_cast<uint32_t>(i % K), st = static_cast<uint32_t>(t);
  const f32x4 zt = philox_normal4(seed, patch, res, st, STREAM_TRANS);
  f32x4 ax = philox_normal4(seed, patch, res, st, STREAM_AXIS);
  const f32x4 ua = philox_uniform4(seed, patch, res, st, STREAM_ANGLE);
  const f32x4 na = normals_from_uniforms(ua);  // .z is the Box-Muller normal of (u2,u3)
  const f32x4 us = philox_uniform4(seed, patch, res, st, STREAM_SEQ);
  const float theta = igso3_theta(rev_cdf, n_bins, rev_sigmas[t], thr, t, ua.x, ua.y, na.z);
  normalize3(ax.x, ax.y, ax.z);
  reverse_update_one(i, t, beta[t], alpha[t], omabs[t], seq, x, O, eps_hat, O0_hat, post, V, zt.x, zt.y, zt.z, ax.x * theta,
                     ax.y * theta, ax.z * theta, us.x);
}

__global__ void sample_init_kernel(int64_t* __restrict__ seq, float* __restrict__ x, float* __restrict__ O, const uint8_t* __restrict__ gm,
                                   uint64_t seed, int64_t first_patch, int B, int K, int T) {
  const int64_t i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x;
  if (i >= static_cast<int64_t>(B) * K || !gm[i]) return;
  const uint32_t patch = static_cast<uint32_t>(first_patch + i / K), res = static_cast<uint32_t>(i % K), st = static_cast<uint32_t>(T + 1);
  const f32x4 nx = philox_normal4(seed, patch, res, st, STREAM_INIT_X);
  x[i * 3 + 0] = nx.x; x[i * 3 + 1] = nx.y; x[i * 3 + 2] = nx.z;
  f32x4 q = philox_normal4(seed, patch, res, st, STREAM_INIT_O);
  const float qn = fmaxf(sqrtf(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w), 1e-12f);
  const float w = q.x / qn, a = q.y / qn, b = q.z / qn, c = q.w / qn;  // (w, x, y, z)
  float* o = O + i * 9;
  o[0] = 1 - 2 * (b * b + c * c); o[1] = 2 * (a * b - c * w);     o[2] = 2 * (a * c + b * w);
  o[3] = 2 * (a * b + c * w);     o[4] = 1 - 2 * (a * a + c * c); o[5] = 2 * (b * c - a * w);
  o[6] = 2 * (a * c - b * w);     o[7] = 2 * (b * c + a * w);     o[8] = 1 - 2 * (a * a + b * b);
  const f32x4 us = philox_uniform4(seed, patch, res, st, STREAM_INIT_S);
  seq[i] = min(static_cast<int>(us.x * 20.0f), 19);
}

__global__ void fill_beta_kernel(const float* __restrict__ beta, int t, int B, float* __restrict__ out, const int* __restrict__ t_dev) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (t_dev != nullptr) t = *t_dev;
  if (i < B) out[i] = beta[t];
}
__global__ void set_int_kernel(int* __restrict__ p, int v) { *p = v; }
__global__ void dec_int_kernel(int* __restrict__ p) { *p -= 1; }

// launchers used by api.hip (sample loop)
int launch_reverse_update_philox(const diffab_sched* s, const diffab_igso3* tab, int t, int64_t* seq, float* x, float* O,
                                 const float* eps_hat, float* O0_hat, float* post, const uint8_t* gm, uint64_t seed,
                                 int64_t first_patch, int B, int K, int V, hipStream_t st, const int* t_dev, const float* head_v,
                                 const float* head_logits) {
  const int64_t n = static_cast<int64_t>(B) * K;
  hipLaunchKernelGGL(reverse_update_philox_kernel, dim3(blocks_for(n)), dim3(kThreads), 0, st, s->beta, s->alpha,
                     s->one_minus_alpha_bar_sqrt, t, tab->sigmas, tab->cdf, tab->n_bins, tab->sigma_threshold, seq, x, O, eps_hat, O0_hat,
                     post, gm, seed, first_patch, B, K, V, t_dev, head_v, head_logits);
  DIFFAB_LAUNCH_CHECK();
  return DIFFAB_OK;
}

int launch_losses_fwd(const float* pp, const float* tp, const float* pe, const float* te, const float* pO, const float* tO, const uint8_t* gm,
                      const uint8_t* rm, int B, int K, int V, float* out3, hipStream_t st, float* scratch) {
  const int64_t n = static_cast<int64_t>(B) * K;
  if (scratch != nullptr && n >= 4096) {  // scratch: 4 kLossBlocks floats (the training tape has them)
    hipLaunchKernelGGL(losses_partial_kernel, dim3(kLossBlocks), dim3(256), 0, st, pp, tp, pe, te, pO, tO, gm, rm, n, V, scratch);
    hipLaunchKernelGGL(losses_final_kernel, dim3(1), dim3(256), 0, st, scratch, kLossBlocks, out3);
    DIFFAB_LAUNCH_CHECK();
    return DIFFAB_OK;
  }
  hipLaunchKernelGGL(losses_kernel, dim3(1), dim3(1024), 0, st, pp, tp, pe, te, pO, tO, gm, rm, static_cast<int64_t>(B) * K, V, out3);
  DIFFAB_LAUNCH_CHECK();
  return DIFFAB_OK;
}

int launch_fill_beta(const diffab_sched* s, int t, int B, float* out, hipStream_t st, const int* t_dev) {
  hipLaunchKernelGGL(fill_beta_kernel, dim3((B + 255) / 256), dim3(256), 0, st, s->beta, t, B, out, t_dev);
  DIFFAB_LAUNCH_CHECK();
  return DIFFAB_OK;
}
__global__ void tiles_needed_kernel(const uint8_t* __restrict__ gm, int64_t ntiles, unsigned char* __restrict__ out) {
  const int64_t i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x;  // (patch, 16-row tile): K % 16 == 0
  if (i >= ntiles) return;
  unsigned char any = 0;
  for (int r = 0; r < 16; ++r) any |= gm[i * 16 + r];
  out[i] = any != 0;
}
int launch_tiles_needed(const uint8_t* gm, int B, int K, unsigned char* out, hipStream_t st) {
  const int64_t n = static_cast<int64_t>(B) * (K / 16);
  hipLaunchKernelGGL(tiles_needed_kernel, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, st, gm, n, out);
  DIFFAB_LAUNCH_CHECK();
  return DIFFAB_OK;
}
int launch_set_int(int* p, int v, hipStream_t st) {
  hipLaunchKernelGGL(set_int_kernel, dim3(1), dim3(1), 0, st, p, v);
  DIFFAB_LAUNCH_CHECK();
  return DIFFAB_OK;
}
int launch_dec_int(int* p, hipStream_t st) {
  hipLaunchKernelGGL(dec_int_kernel, dim3(1), dim3(1), 0, st, p);
  DIFFAB_LAUNCH_CHECK();
  return DIFFAB_OK;
}

}  // namespace diffab

using namespace diffab;

static int check_sched(const diffab_sched* s) {
  DIFFAB_REQUIRE(s && s->T > 0 && s->alpha && s->alpha_bar && s->alpha_bar_sqrt && s->one_minus_alpha_bar_sqrt && s->beta, DIFFAB_ERR_ARG,
                 "schedule: null table or T <= 0");
  return DIFFAB_OK;
}

#define ELEMENTWISE_ENTRY(NAME, KERNEL, NPTR_OK, N, ...)                                              \
  DIFFAB_REQUIRE((N) >= 0, DIFFAB_ERR_ARG, NAME ": negative count");                                  \
  if ((N) == 0) return DIFFAB_OK; /* empty tensors carry null pointers */                             \
  DIFFAB_REQUIRE((NPTR_OK), DIFFAB_ERR_ARG, NAME ": null pointer");                                   \
  hipLaunchKernelGGL(KERNEL, dim3(blocks_for(N)), dim3(kThreads), 0, as_stream(stream), __VA_ARGS__); \
  DIFFAB_LAUNCH_CHECK();                                                                              \
  return DIFFAB_OK;

extern "C" {

int diffab_so3_log(const float* R, float* S, int64_t n, void* stream) {
  StreamOrder order_(stream);
  ELEMENTWISE_ENTRY("so3_log", so3_log_kernel, R && S, n, R, S, n)
}
int diffab_so3_exp(const float* S, float* R, int64_t n, void* stream) {
  StreamOrder order_(stream);
  ELEMENTWISE_ENTRY("so3_exp", so3_exp_kernel, R && S, n, S, R, n)
}
int diffab_so3_matrix_to_rotvec(const float* R, float* v, int64_t n, void* stream) {
  StreamOrder order_(stream);
  ELEMENTWISE_ENTRY("so3_matrix_to_rotvec", so3_matrix_to_rotvec_kernel, R && v, n, R, v, n)
}
int diffab_so3_rotvec_to_matrix(const float* v, float* R, int64_t n, void* stream) {
  StreamOrder order_(stream);
  ELEMENTWISE_ENTRY("so3_rotvec_to_matrix", so3_rotvec_to_matrix_kernel, R && v, n, v, R, n)
}
int diffab_so3_scale_rot(const float* R, const float* k, float* out, int64_t n, int64_t per_k, void* stream) {
  StreamOrder order_(stream);
  DIFFAB_REQUIRE(per_k > 0, DIFFAB_ERR_ARG, "so3_scale_rot: per_k must be positive");
  ELEMENTWISE_ENTRY("so3_scale_rot", so3_scale_rot_kernel, R && k && out, n, R, k, out, n, per_k)
}

static int igso3_table(const float* sigmas, int32_t n_sigmas, int32_t n_bins, int32_t num_iters, float* pdf, bool faithful, void* stream) {
  DIFFAB_REQUIRE(sigmas && pdf && n_sigmas > 0 && n_bins > 0 && num_iters > 0 && num_iters <= 4096, DIFFAB_ERR_ARG,
                 "igso3_table_build: bad argument");
  const int64_t n = static_cast<int64_t>(n_sigmas) * n_bins;
  if (faithful)
    hipLaunchKernelGGL(igso3_pdf_kernel<true>, dim3(blocks_for(n)), dim3(kThreads), 0, as_stream(stream), sigmas, n_sigmas, n_bins, num_iters, pdf);
  else
    hipLaunchKernelGGL(igso3_pdf_kernel<false>, dim3(blocks_for(n)), dim3(kThreads), 0, as_stream(stream), sigmas, n_sigmas, n_bins, num_iters, pdf);
  DIFFAB_LAUNCH_CHECK();
  return DIFFAB_OK;
}
int diffab_igso3_table_build(const float* sigmas, int32_t n_sigmas, int32_t n_bins, int32_t num_iters, float* pdf, void* stream) {
  StreamOrder order_(stream);
  return igso3_table(sigmas, n_sigmas, n_bins, num_iters, pdf, true, stream);
}
int diffab_igso3_table_build_accurate(const float* sigmas, int32_t n_sigmas, int32_t n_bins, int32_t num_iters, float* pdf, void* stream) {
  StreamOrder order_(stream);
  return igso3_table(sigmas, n_sigmas, n_bins, num_iters, pdf, false, stream);
}

int diffab_igso3_cdf_build(const float* pdf, int32_t n_sigmas, int32_t n_bins, float* cdf, void* stream) {
  StreamOrder order_(stream);
  DIFFAB_REQUIRE(pdf && cdf && n_sigmas > 0 && n_bins > 0, DIFFAB_ERR_ARG, "igso3_cdf_build: bad argument");
  hipLaunchKernelGGL(igso3_cdf_kernel, dim3(n_sigmas), dim3(kThreads), 0, as_stream(stream), pdf, n_bins, cdf);
  DIFFAB_LAUNCH_CHECK();
  return DIFFAB_OK;
}

int diffab_igso3_sample(const diffab_igso3* tab, const int64_t* sigma_idx, int32_t B, int32_t K, const float* axis_raw, const float* u_bin,
                        const float* u_in, const float* z, float* rotvec, void* stream) {
  StreamOrder order_(stream);
  DIFFAB_REQUIRE(tab && tab->sigmas && tab->cdf && tab->n_bins > 0 && B >= 0 && K >= 0, DIFFAB_ERR_ARG, "igso3_sample: bad argument");
  const int64_t n = static_cast<int64_t>(B) * K;
  if (n == 0) return DIFFAB_OK;  // empty tensors carry null pointers
  DIFFAB_REQUIRE(sigma_idx && axis_raw && u_bin && u_in && z && rotvec, DIFFAB_ERR_ARG, "igso3_sample: null pointer");
  hipLaunchKernelGGL(igso3_sample_kernel, dim3(blocks_for(n)), dim3(kThreads), 0, as_stream(stream), tab->sigmas, tab->cdf, tab->n_bins,
                     tab->sigma_threshold, sigma_idx, B, K, axis_raw, u_bin, u_in, z, rotvec);
  DIFFAB_LAUNCH_CHECK();
  return DIFFAB_OK;
}

int diffab_igso3_bins_without_replacement(const float* pdf, int32_t n_sigmas, int32_t n_bins, const int64_t* sigma_idx, int32_t B, int32_t K,
                                          const float* race, int32_t* bins, const float* sigmas, float sigma_threshold, void* stream) {
  StreamOrder order_(stream);
  DIFFAB_REQUIRE(n_sigmas > 0 && n_bins > 0 && n_bins <= 16384 && B >= 0 && K >= 0 && K <= n_bins, DIFFAB_ERR_ARG,
                 "igso3_bins_without_replacement: bad argument (n_bins <= 16384, K <= n_bins)");
  if (B == 0 || K == 0) return DIFFAB_OK;  // empty tensors carry null pointers
  DIFFAB_REQUIRE(pdf && sigma_idx && race && bins, DIFFAB_ERR_ARG, "igso3_bins_without_replacement: null pointer");
  int n_pad = 2;
  while (n_pad < n_bins) n_pad <<= 1;
  const int lds = n_pad * 8;
  DIFFAB_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(igso3_race_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  hipLaunchKernelGGL(igso3_race_kernel, dim3(B), dim3(1024), lds, as_stream(stream), pdf, n_bins, n_pad, sigma_idx, K, race, bins, sigmas,
                     sigma_threshold);
  DIFFAB_LAUNCH_CHECK();
  return DIFFAB_OK;
}

int diffab_igso3_sample_bins(const diffab_igso3* tab, const int64_t* sigma_idx, int32_t B, int32_t K, const float* axis_raw,
                             const int32_t* bins, const float* u_in, const float* z, float* rotvec, void* stream) {
  StreamOrder order_(stream);
  DIFFAB_REQUIRE(tab && tab->sigmas && tab->n_bins > 0 && B >= 0 && K >= 0, DIFFAB_ERR_ARG, "igso3_sample_bins: bad argument");
  const int64_t n = static_cast<int64_t>(B) * K;
  if (n == 0) return DIFFAB_OK;
  DIFFAB_REQUIRE(sigma_idx && axis_raw && bins && u_in && z && rotvec, DIFFAB_ERR_ARG, "igso3_sample_bins: null pointer");
  hipLaunchKernelGGL(igso3_sample_bins_kernel, dim3(blocks_for(n)), dim3(kThreads), 0, as_stream(stream), tab->sigmas, tab->n_bins,
                     tab->sigma_threshold, sigma_idx, B, K, axis_raw, bins, u_in, z, rotvec);
  DIFFAB_LAUNCH_CHECK();
  return DIFFAB_OK;
}

int diffab_weighted_multinomial(const float* p1, const float* p2, const float* w1, const float* w2, int64_t n, int64_t per_b, float* out,
                                void* stream) {
  StreamOrder order_(stream);
  DIFFAB_REQUIRE(per_b > 0, DIFFAB_ERR_ARG, "weighted_multinomial: per_b must be positive");
  ELEMENTWISE_ENTRY("weighted_multinomial", weighted_multinomial_kernel, p1 && p2 && w1 && w2 && out, n, p1, p2, w1, w2, n, per_b, out)
}

int diffab_seq_forward_prob(const diffab_sched* s, int mode, const int64_t* seq, const int64_t* t, const uint8_t* mask, int32_t B,
                            int32_t K, float* prob, void* stream) {
  StreamOrder order_(stream);
  if (int rc = check_sched(s)) return rc;
  DIFFAB_REQUIRE(B >= 0 && K >= 0 && (mode == 0 || mode == 1), DIFFAB_ERR_ARG, "seq_forward_prob: bad argument");
  const int64_t n = static_cast<int64_t>(B) * K;
  if (n == 0) return DIFFAB_OK;  // empty tensors carry null pointers
  DIFFAB_REQUIRE(seq && t && mask && prob, DIFFAB_ERR_ARG, "seq_forward_prob: null pointer");
  hipLaunchKernelGGL(seq_forward_prob_kernel, dim3(blocks_for(n)), dim3(kThreads), 0, as_stream(stream), s->beta, s->alpha_bar, s->T, mode,
                     seq, t, mask, B, K, prob);
  DIFFAB_LAUNCH_CHECK();
  return DIFFAB_OK;
}

int diffab_seq_posterior(const diffab_sched* s, const int64_t* seq_t, const int64_t* seq_0, const int64_t* t, const uint8_t* mask,
                         int32_t B, int32_t K, float* post, void* stream) {
  StreamOrder order_(stream);
  if (int rc = check_sched(s)) return rc;
  DIFFAB_REQUIRE(B >= 0 && K >= 0, DIFFAB_ERR_ARG, "seq_posterior: bad argument");
  const int64_t n = static_cast<int64_t>(B) * K;
  if (n == 0) return DIFFAB_OK;
  DIFFAB_REQUIRE(seq_t && seq_0 && t && mask && post, DIFFAB_ERR_ARG, "seq_posterior: null pointer");
  hipLaunchKernelGGL(seq_posterior_kernel, dim3(blocks_for(n)), dim3(kThreads), 0, as_stream(stream), s->beta, s->alpha_bar, s->T, seq_t,
                     seq_0, t, mask, B, K, post);
  DIFFAB_LAUNCH_CHECK();
  return DIFFAB_OK;
}

int diffab_categorical_sample(const float* prob, const float* u, int64_t n_rows, int32_t V, int64_t* out, void* stream) {
  StreamOrder order_(stream);
  DIFFAB_REQUIRE(V > 0, DIFFAB_ERR_ARG, "categorical_sample: V must be positive");
  ELEMENTWISE_ENTRY("categorical_sample", categorical_sample_kernel, prob && u && out, n_rows, prob, u, n_rows, V, out)
}

int diffab_coord_forward(const diffab_sched* s, const float* x0, const int64_t* t, const uint8_t* mask, const float* eps, int32_t B,
                         int32_t K, float* xt, void* stream) {
  StreamOrder order_(stream);
  if (int rc = check_sched(s)) return rc;
  DIFFAB_REQUIRE(B >= 0 && K >= 0, DIFFAB_ERR_ARG, "coord_forward: bad argument");
  const int64_t n = static_cast<int64_t>(B) * K;
  if (n == 0) return DIFFAB_OK;
  DIFFAB_REQUIRE(x0 && t && mask && eps && xt, DIFFAB_ERR_ARG, "coord_forward: null pointer");
  hipLaunchKernelGGL(coord_forward_kernel, dim3(blocks_for(n)), dim3(kThreads), 0, as_stream(stream), s->alpha_bar_sqrt,
                     s->one_minus_alpha_bar_sqrt, s->T, x0, t, mask, eps, B, K, xt);
  DIFFAB_LAUNCH_CHECK();
  return DIFFAB_OK;
}

int diffab_orient_forward(const diffab_sched* s, const float* O0, const uint8_t* mask, const int64_t* t, const float* rotvec, int32_t B,
                          int32_t K, float* Ot, void* stream) {
  StreamOrder order_(stream);
  if (int rc = check_sched(s)) return rc;
  DIFFAB_REQUIRE(B >= 0 && K >= 0, DIFFAB_ERR_ARG, "orient_forward: bad argument");
  const int64_t n = static_cast<int64_t>(B) * K;
  if (n == 0) return DIFFAB_OK;
  DIFFAB_REQUIRE(O0 && t && mask && rotvec && Ot, DIFFAB_ERR_ARG, "orient_forward: null pointer");
  hipLaunchKernelGGL(orient_forward_kernel, dim3(blocks_for(n)), dim3(kThreads), 0, as_stream(stream), s->alpha_bar_sqrt, s->T, O0, mask, t,
                     rotvec, B, K, Ot);
  DIFFAB_LAUNCH_CHECK();
  return DIFFAB_OK;
}

int diffab_philox_fill(uint64_t seed, int64_t first_patch, int32_t B, int32_t K, int32_t step, int32_t stream_id, int kind, float* out,
                       void* stream) {
  StreamOrder order_(stream);
  DIFFAB_REQUIRE(B >= 0 && K >= 0 && (kind == 0 || kind == 1), DIFFAB_ERR_ARG, "philox_fill: bad argument");
  const int64_t n = static_cast<int64_t>(B) * K;
  if (n == 0) return DIFFAB_OK;
  DIFFAB_REQUIRE(out, DIFFAB_ERR_ARG, "philox_fill: null pointer");
  hipLaunchKernelGGL(philox_fill_kernel, dim3(blocks_for(n)), dim3(kThreads), 0, as_stream(stream), seed, first_patch, B, K, step, stream_id,
                     kind, out);
  DIFFAB_LAUNCH_CHECK();
  return DIFFAB_OK;
}

int diffab_losses_fwd(const float* pred_post, const float* true_post, const float* pred_eps, const float* true_eps, const float* pred_O0,
                      const float* true_O0, const uint8_t* gen_mask, const uint8_t* res_mask, int32_t B, int32_t K, int32_t V,
                      float* losses3, void* stream) {
  StreamOrder order_(stream);
  DIFFAB_REQUIRE(pred_post && true_post && pred_eps && true_eps && pred_O0 && true_O0 && gen_mask && res_mask && losses3 && B > 0 && K > 0 &&
                     V > 0,
                 DIFFAB_ERR_ARG, "losses_fwd: bad argument");
  hipLaunchKernelGGL(losses_kernel, dim3(1), dim3(1024), 0, as_stream(stream), pred_post, true_post, pred_eps, true_eps, pred_O0, true_O0,
                     gen_mask, res_mask, static_cast<int64_t>(B) * K, V, losses3);
  DIFFAB_LAUNCH_CHECK();
  return DIFFAB_OK;
}

int diffab_orientation_loss(const float* pred, const float* target, int64_t n, float* elems, float* sum1, void* stream) {
  StreamOrder order_(stream);
  DIFFAB_REQUIRE(pred && target && n > 0 && (elems || sum1), DIFFAB_ERR_ARG, "orientation_loss: bad argument");
  hipLaunchKernelGGL(orientation_loss_kernel, dim3(1), dim3(1024), 0, as_stream(stream), pred, target, n, elems, sum1);
  DIFFAB_LAUNCH_CHECK();
  return DIFFAB_OK;
}

int diffab_orientation_loss_bwd(const float* pred, const float* target, int64_t n, const float* g_elems, const float* g_total,
                                float* d_pred, float* d_target, void* stream) {
  StreamOrder order_(stream);
  DIFFAB_REQUIRE(pred && target && n > 0 && (g_elems || g_total) && (d_pred || d_target), DIFFAB_ERR_ARG, "orientation_loss_bwd: bad argument");
  hipLaunchKernelGGL(orientation_loss_bwd_kernel, dim3(blocks_for(n)), dim3(kThreads), 0, as_stream(stream), pred, target, n, g_elems, g_total,
                     d_pred, d_target);
  DIFFAB_LAUNCH_CHECK();
  return DIFFAB_OK;
}

int diffab_frames_apply(const float* x, const float* R, const float* t, float* out, int32_t B, int32_t N, int32_t L, int32_t P, void* stream) {
  StreamOrder order_(stream);
  DIFFAB_REQUIRE(B >= 0 && N >= 0 && L >= 0 && P >= 0, DIFFAB_ERR_ARG, "frames_apply: negative extent");
  const int64_t n = static_cast<int64_t>(B) * N * L * P;
  if (n == 0) return DIFFAB_OK;  // empty tensors carry null pointers
  DIFFAB_REQUIRE(x && R && out, DIFFAB_ERR_ARG, "frames_apply: null pointer");
  hipLaunchKernelGGL(frames_kernel<false>, dim3(blocks_for(n)), dim3(kThreads), 0, as_stream(stream), x, R, t, out, N, L, P, n);
  DIFFAB_LAUNCH_CHECK();
  return DIFFAB_OK;
}

int diffab_frames_bwd(const float* x, const float* R, const float* t, const float* g_out, int32_t invert, float* dR, float* dt, int32_t B,
                      int32_t N, int32_t L, int32_t P, void* stream) {
  StreamOrder order_(stream);
  DIFFAB_REQUIRE(B >= 0 && N >= 0 && L >= 0 && P >= 0, DIFFAB_ERR_ARG, "frames_bwd: negative extent");
  if (static_cast<int64_t>(B) * L == 0) return DIFFAB_OK;
  DIFFAB_REQUIRE(x && R && t && g_out, DIFFAB_ERR_ARG, "frames_bwd: null pointer");
  if (invert) hipLaunchKernelGGL(frames_bwd_kernel<true>, dim3(B * L), dim3(256), 0, as_stream(stream), x, R, t, g_out, N, L, P, dR, dt);
  else hipLaunchKernelGGL(frames_bwd_kernel<false>, dim3(B * L), dim3(256), 0, as_stream(stream), x, R, t, g_out, N, L, P, dR, dt);
  DIFFAB_LAUNCH_CHECK();
  return DIFFAB_OK;
}

int diffab_frames_invert(const float* x, const float* R, const float* t, float* out, int32_t B, int32_t N, int32_t L, int32_t P, void* stream) {
  StreamOrder order_(stream);
  DIFFAB_REQUIRE(B >= 0 && N >= 0 && L >= 0 && P >= 0, DIFFAB_ERR_ARG, "frames_invert: negative extent");
  const int64_t n = static_cast<int64_t>(B) * N * L * P;
  if (n == 0) return DIFFAB_OK;  // empty tensors carry null pointers
  DIFFAB_REQUIRE(x && R && out, DIFFAB_ERR_ARG, "frames_invert: null pointer");
  hipLaunchKernelGGL(frames_kernel<true>, dim3(blocks_for(n)), dim3(kThreads), 0, as_stream(stream), x, R, t, out, N, L, P, n);
  DIFFAB_LAUNCH_CHECK();
  return DIFFAB_OK;
}

// d x = g[0] + sum_k f_k (cos(f_k x) g_sin[k] - sin(f_k x) g_cos[k]); sin / cos are taken from the saved forward output
__global__ void angular_encoding_bwd_kernel(const float* __restrict__ enc, const float* __restrict__ g, int64_t n, int nf, float* __restrict__ dx) {
  const int64_t i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x;
  if (i >= n) return;
  const int w = 4 * nf + 1;
  const float* o = enc + i * w;
  const float* gi = g + i * w;
  float acc = gi[0];
  for (int k = 0; k < 2 * nf; ++k) {
    const float f = k < nf ? static_cast<float>(k + 1) : 1.0f / static_cast<float>(k - nf + 1);
    acc += f * (o[1 + 2 * nf + k] * gi[1 + k] - o[1 + k] * gi[1 + 2 * nf + k]);
  }
  dx[i] = acc;
}

int diffab_angular_encoding_bwd(const float* enc, const float* g_out, int64_t n, int32_t num_funcs, float* dx, void* stream) {
  StreamOrder order_(stream);
  DIFFAB_REQUIRE(n >= 0 && num_funcs > 0, DIFFAB_ERR_ARG, "angular_encoding_bwd: bad argument");
  if (n == 0) return DIFFAB_OK;
  DIFFAB_REQUIRE(enc && g_out && dx, DIFFAB_ERR_ARG, "angular_encoding_bwd: null pointer");
  hipLaunchKernelGGL(angular_encoding_bwd_kernel, dim3(blocks_for(n)), dim3(kThreads), 0, as_stream(stream), enc, g_out, n, num_funcs, dx);
  DIFFAB_LAUNCH_CHECK();
  return DIFFAB_OK;
}

int diffab_angular_encoding(const float* x, int64_t n, int32_t num_funcs, float* out, void* stream) {
  StreamOrder order_(stream);
  DIFFAB_REQUIRE(n >= 0 && num_funcs > 0, DIFFAB_ERR_ARG, "angular_encoding: bad argument");
  if (n == 0) return DIFFAB_OK;  // empty tensors carry null pointers
  DIFFAB_REQUIRE(x && out, DIFFAB_ERR_ARG, "angular_encoding: null pointer");
  hipLaunchKernelGGL(angular_encoding_kernel, dim3(blocks_for(n)), dim3(kThreads), 0, as_stream(stream), x, n, num_funcs, out);
  DIFFAB_LAUNCH_CHECK();
  return DIFFAB_OK;
}

int diffab_reverse_update(const diffab_sched* s, int32_t t, int64_t* seq, float* x, float* O, const float* eps_hat, const float* O0_hat,
                          const float* posterior, const uint8_t* gen_mask, const float* z, const float* rotvec, const float* u_seq,
                          int32_t B, int32_t K, int32_t V, void* stream) {
  StreamOrder order_(stream);
  if (int rc = check_sched(s)) return rc;
  DIFFAB_REQUIRE(seq && x && O && eps_hat && O0_hat && posterior && gen_mask && z && rotvec && u_seq && B >= 0 && K > 0 && V > 0 && t >= 1 &&
                     t <= s->T,
                 DIFFAB_ERR_ARG, "reverse_update: bad argument (t must be in [1, T])");
  const int64_t n = static_cast<int64_t>(B) * K;
  if (n == 0) return DIFFAB_OK;
  hipLaunchKernelGGL(reverse_update_kernel, dim3(blocks_for(n)), dim3(kThreads), 0, as_stream(stream), s->beta, s->alpha,
                     s->one_minus_alpha_bar_sqrt, t, seq, x, O, eps_hat, O0_hat, posterior, gen_mask, z, rotvec, u_seq, B, K, V);
  DIFFAB_LAUNCH_CHECK();
  return DIFFAB_OK;
}

int diffab_sample_init(int64_t* seq, float* x, float* O, const uint8_t* gen_mask, uint64_t seed, int64_t first_patch, int32_t B, int32_t K,
                       int32_t T, void* stream) {
  StreamOrder order_(stream);
  DIFFAB_REQUIRE(seq && x && O && gen_mask && B >= 0 && K > 0 && T > 0, DIFFAB_ERR_ARG, "sample_init: bad argument");
  const int64_t n = static_cast<int64_t>(B) * K;
  if (n == 0) return DIFFAB_OK;
  hipLaunchKernelGGL(sample_init_kernel, dim3(blocks_for(n)), dim3(kThreads), 0, as_stream(stream), seq, x, O, gen_mask, seed, first_patch, B,
                     K, T);
  DIFFAB_LAUNCH_CHECK();
  return DIFFAB_OK;
}

}  // extern "C"
