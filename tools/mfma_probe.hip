// Diagnostic: what fp32 MFMA rate does the chip sustain with no memory traffic at all?  (The 157 TFLOP/s datasheet number
// assumes 2.4 GHz; this measures the rate and the implied clock under a matrix-pipe-bound load.)
//   hipcc -O3 --offload-arch=gfx950 tools/mfma_probe.hip -o tools/bin/mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int KIND>
__global__ __launch_bounds__(512) void probe(float* out, int iters, unsigned long long* clk) {
  const float a = threadIdx.x * 1e-3f, b = 1.0f + blockIdx.x * 1e-6f;
  unsigned long long t0 = __builtin_readcyclecounter();
  unsigned long long m0 = __builtin_amdgcn_s_memtime();
  float r = 0.f;
  if (KIND == 0) {
    f32x4 acc[6];
    for (int i = 0; i < 6; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int i = 0; i < 6; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    for (int i = 0; i < 6; ++i) r += acc[i][0] + acc[i][3];
  } else {
    f32x16 acc[3];
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int i = 0; i < 3; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    for (int i = 0; i < 3; ++i) r += acc[i][0] + acc[i][7];
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  unsigned long long m1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
  if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = t1 - t0; clk[1] = m1 - m0; }
}

int main() {
  float* out;
  unsigned long long* clk;
  hipMalloc(&out, 4096 * 512 * 4);
  hipMalloc(&clk, 16);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int kind = 0; kind < 2; ++kind)
    for (int threads : {256, 512, 1024}) {
      const int iters = 20000, grid = 256;
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        if (kind == 0) hipLaunchKernelGGL(probe<0>, dim3(grid), dim3(threads), 0, 0, out, iters, clk);
        else hipLaunchKernelGGL(probe<1>, dim3(grid), dim3(threads), 0, 0, out, iters, clk);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
      }
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      unsigned long long h[2];
      hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
      const double mfma_per_wave = double(iters) * (kind == 0 ? 24 : 12);
      const double flop = mfma_per_wave * (kind == 0 ? 2048.0 : 4096.0) * (threads / 64) * grid;
      // pipe cycles per SIMD: waves per SIMD x MFMAs x (32 | 64) cycles
      const double pipe_cycles = mfma_per_wave * (kind == 0 ? 32 : 64) * (threads / 256.0);
      printf("%s  %4d threads/WG (1 WG/CU): %.3f ms  %.1f TFLOP/s  implied clock %.2f GHz if the pipe never idles; cyclecounter %llu memtime %llu\n",
             kind == 0 ? "16x16x4 " : "32x32x2 ", threads, ms, flop / ms * 1e-9, pipe_cycles / ms * 1e-6, h[0], h[1]);
    }
  return 0;
}
