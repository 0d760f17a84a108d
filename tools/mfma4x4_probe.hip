// Diagnostic: v_mfma_f32_4x4x1_16B_f32 on gfx950 - operand / result lane maps (exact integer data) and sustained rate.
// 16 independent 4x4 outer products (K = 1) per instruction: no N padding for the 8-head products of the attention kernel.
//   hipcc -O3 --offload-arch=gfx950 tools/mfma4x4_probe.hip -o tools/bin/mfma4x4_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void layout_kernel(const float* a, const float* b, float* d) {
  const int l = threadIdx.x;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a[l], b[l], acc, 0, 0, 0);
  for (int r = 0; r < 4; ++r) d[l * 4 + r] = acc[r];
}

template <int NACC>
__global__ __launch_bounds__(512) void rate_kernel(float* out, int iters) {
  const float a = threadIdx.x * 1e-3f, b = 1.0f + blockIdx.x * 1e-6f;
  f32x4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int it = 0; it < iters; ++it)
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[i], 0, 0, 0);
  float r = 0.f;
  for (int i = 0; i < NACC; ++i) r += acc[i][0] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

// mixed: per 4x4x1 MFMA one independent v_fma (does VALU co-issue beside the short MFMAs?)
template <int NACC>
__global__ __launch_bounds__(512) void rate_mixed_kernel(float* out, int iters) {
  const float a = threadIdx.x * 1e-3f, b = 1.0f + blockIdx.x * 1e-6f;
  f32x4 acc[NACC];
  float v[NACC];
  for (int i = 0; i < NACC; ++i) { acc[i] = f32x4{0.f, 0.f, 0.f, 0.f}; v[i] = a + i; }
  for (int it = 0; it < iters; ++it)
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int i = 0; i < NACC; ++i) {
        acc[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[i], 0, 0, 0);
        v[i] = __builtin_fmaf(v[i], b, a);
      }
  float r = 0.f;
  for (int i = 0; i < NACC; ++i) r += acc[i][0] + acc[i][3] + v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

// does independent VALU work hide in the shadow of a 32-cycle v_mfma_f32_16x16x4_f32?  NF v_fma per MFMA, 4 accumulators
template <int NF>
__global__ __launch_bounds__(512) void rate16_mixed_kernel(float* out, int iters) {
  const float a = threadIdx.x * 1e-3f, b = 1.0f + blockIdx.x * 1e-6f;
  f32x4 acc[4];
  float v[8];
  for (int i = 0; i < 4; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int i = 0; i < 8; ++i) v[i] = a + i;
  for (int it = 0; it < iters; ++it)
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
#pragma unroll
        for (int f = 0; f < NF; ++f) v[f & 7] = __builtin_fmaf(v[f & 7], b, a);
      }
  float r = 0.f;
  for (int i = 0; i < 4; ++i) r += acc[i][0] + acc[i][3];
  for (int i = 0; i < 8; ++i) r += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

// the same question for the bf16 matrix core: NF v_fma per v_mfma_f32_16x16x32_bf16 (16 cycles alone)
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
template <int NF>
__global__ __launch_bounds__(512) void rate_bf16_mixed_kernel(float* out, int iters) {
  const float a = threadIdx.x * 1e-3f, b = 1.0f + blockIdx.x * 1e-6f;
  bf16x8 fa, fb;
  for (int i = 0; i < 8; ++i) { fa[i] = static_cast<__bf16>(a + i); fb[i] = static_cast<__bf16>(b - i); }
  f32x4 acc[8];
  float v[8];
  for (int i = 0; i < 8; ++i) { acc[i] = f32x4{0.f, 0.f, 0.f, 0.f}; v[i] = a + i; }
  for (int it = 0; it < iters; ++it)
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, acc[i], 0, 0, 0);
#pragma unroll
        for (int f = 0; f < NF; ++f) v[f & 7] = __builtin_fmaf(v[f & 7], b, a);
      }
  float r = 0.f;
  for (int i = 0; i < 8; ++i) r += acc[i][0] + acc[i][3] + v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

int main() {
  // ---- layout: a[l] = 1 + l, b[l] = 100 + l  ->  which (la, lb) product lands in d[l][r]?
  std::vector<float> ha(64), hb(64), hd(256);
  for (int l = 0; l < 64; ++l) { ha[l] = float(1 + l); hb[l] = float(101 + l); }
  float *a, *b, *d;
  hipMalloc(&a, 256); hipMalloc(&b, 256); hipMalloc(&d, 1024);
  hipMemcpy(a, ha.data(), 256, hipMemcpyHostToDevice);
  hipMemcpy(b, hb.data(), 256, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(layout_kernel, dim3(1), dim3(64), 0, 0, a, b, d);
  hipMemcpy(hd.data(), d, 1024, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; ++l)
    for (int r = 0; r < 4; ++r) {
      // hypothesis: block = l >> 2, column j = l & 3 (B lane 4 blk + j), row i = r (A lane 4 blk + r)
      const int blk = l >> 2, j = l & 3;
      const float want = ha[4 * blk + r] * hb[4 * blk + j];
      if (hd[l * 4 + r] != want) {
        if (bad < 8) {
          // find the actual pair
          int fa = -1, fb = -1;
          for (int x = 0; x < 64 && fa < 0; ++x)
            for (int y = 0; y < 64; ++y)
              if (ha[x] * hb[y] == hd[l * 4 + r]) { fa = x; fb = y; break; }
          printf("lane %d reg %d: got %.0f = a[%d] * b[%d], hypothesis a[%d] * b[%d]\n", l, r, hd[l * 4 + r], fa, fb, 4 * blk + r, 4 * blk + j);
        }
        ++bad;
      }
    }
  printf("layout hypothesis D[lane l][reg r] = A[lane 4 (l>>2) + r] * B[lane l]: %s (%d mismatches)\n", bad ? "WRONG" : "confirmed", bad);

  // ---- rate
  float* out;
  hipMalloc(&out, 4096 * 512 * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  auto run = [&](const char* name, auto kern, int nacc, int threads) {
    const int iters = 40000, grid = 256;
    float ms = 0.f;
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), 0, 0, out, iters);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      hipEventElapsedTime(&ms, e0, e1);
    }
    const double n_mfma = double(iters) * 4 * nacc;                  // per wave
    const double flop = n_mfma * 512.0 * (threads / 64) * grid;
    const double cyc_at_2p3 = ms * 1e-3 * 2.3e9 / (n_mfma * (threads / 256.0));  // cycles per MFMA per SIMD at 2.3 GHz
    printf("%-28s nacc %d, %4d threads/WG: %.3f ms  %.1f TFLOP/s  ~%.1f cycles per MFMA per SIMD (at 2.3 GHz)\n", name, nacc, threads, ms,
           flop / ms * 1e-9, cyc_at_2p3);
  };
  for (int threads : {256, 512}) {
    run("4x4x1 chain", rate_kernel<1>, 1, threads);
    run("4x4x1", rate_kernel<2>, 2, threads);
    run("4x4x1", rate_kernel<4>, 4, threads);
    run("4x4x1", rate_kernel<8>, 8, threads);
    run("4x4x1 + 1 v_fma each", rate_mixed_kernel<8>, 8, threads);
  }
  auto run16 = [&](const char* name, auto kern, int nf, int threads) {
    const int iters = 20000, grid = 256;
    float ms = 0.f;
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), 0, 0, out, iters);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      hipEventElapsedTime(&ms, e0, e1);
    }
    const double n_mfma = double(iters) * 16;
    printf("16x16x4 + %d v_fma per MFMA, %4d threads/WG: %.3f ms  ~%.1f cycles per MFMA per SIMD at 2.3 GHz (32 = the MFMA alone; +4 per v_fma if nothing overlaps)\n",
           nf, threads, ms, ms * 1e-3 * 2.3e9 / (n_mfma * (threads / 256.0)));
  };
  for (int threads : {256, 512}) {
    run16("", rate16_mixed_kernel<0>, 0, threads);
    run16("", rate16_mixed_kernel<2>, 2, threads);
    run16("", rate16_mixed_kernel<4>, 4, threads);
    run16("", rate16_mixed_kernel<6>, 6, threads);
    run16("", rate16_mixed_kernel<8>, 8, threads);
    run16("", rate16_mixed_kernel<12>, 12, threads);
  }
  auto runb = [&](auto kern, int nf, int threads) {
    const int iters = 20000, grid = 256;
    float ms = 0.f;
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), 0, 0, out, iters);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      hipEventElapsedTime(&ms, e0, e1);
    }
    const double n_mfma = double(iters) * 16;
    printf("bf16 16x16x32 + %d v_fma per MFMA, %4d threads/WG: %.3f ms  ~%.1f cycles per MFMA per SIMD at 2.3 GHz (16 = the MFMA alone)\n", nf, threads,
           ms, ms * 1e-3 * 2.3e9 / (n_mfma * (threads / 256.0)));
  };
  for (int threads : {256, 512}) {
    runb(rate_bf16_mixed_kernel<0>, 0, threads);
    runb(rate_bf16_mixed_kernel<1>, 1, threads);
    runb(rate_bf16_mixed_kernel<2>, 2, threads);
    runb(rate_bf16_mixed_kernel<3>, 3, threads);
    runb(rate_bf16_mixed_kernel<4>, 4, threads);
    runb(rate_bf16_mixed_kernel<6>, 6, threads);
  }
  return 0;
}