// Hardware question (gfx950): may the destination of v_mfma_f32_16x16x32_f16 overlap its SrcC PARTIALLY (vDst = v[118:121], SrcC = v[116:119])?
// hipcc emits that form (the 128-bit destination carries no early-clobber constraint).  Two dependent MFMAs, the second either with the
// shifted destination or in place; results compared over many launches with the matrix pipe shared by two waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(512) void k(const f16x8* __restrict__ A, const f16x8* __restrict__ B, f32x4* __restrict__ out, int iters) {
  const int lane = threadIdx.x & 63;
  f16x8 a = A[lane], b = B[(threadIdx.x >> 6) * 64 + lane];
  f32x4 sum = {0.f, 0.f, 0.f, 0.f};
  for (int it = 0; it < iters; ++it) {
    float r0, r1, r2, r3;
    if (MODE == 1) {
      asm volatile(
          "v_mov_b32 v116, 0\n v_mov_b32 v117, 0\n v_mov_b32 v118, 0\n v_mov_b32 v119, 0\n s_nop 7\n"
          "v_mfma_f32_16x16x32_f16 v[116:119], %4, %5, v[116:119]\n"
          "v_mfma_f32_16x16x32_f16 v[118:121], %4, %5, v[116:119]\n"
          "s_nop 15\n s_nop 15\n"
          "v_mov_b32 %0, v118\n v_mov_b32 %1, v119\n v_mov_b32 %2, v120\n v_mov_b32 %3, v121\n"
          : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3)
          : "v"(a), "v"(b)
          : "v116", "v117", "v118", "v119", "v120", "v121");
    } else {
      asm volatile(
          "v_mov_b32 v116, 0\n v_mov_b32 v117, 0\n v_mov_b32 v118, 0\n v_mov_b32 v119, 0\n s_nop 7\n"
          "v_mfma_f32_16x16x32_f16 v[116:119], %4, %5, v[116:119]\n"
          "v_mfma_f32_16x16x32_f16 v[116:119], %4, %5, v[116:119]\n"
          "s_nop 15\n s_nop 15\n"
          "v_mov_b32 %0, v116\n v_mov_b32 %1, v117\n v_mov_b32 %2, v118\n v_mov_b32 %3, v119\n"
          : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3)
          : "v"(a), "v"(b)
          : "v116", "v117", "v118", "v119", "v120", "v121");
    }
    sum[0] += r0; sum[1] += r1; sum[2] += r2; sum[3] += r3;
    asm volatile("" : "+v"(a), "+v"(b));
  }
  out[static_cast<size_t>(blockIdx.x) * 512 + threadIdx.x] = sum;
}

int main(int argc, char** argv) {
  const int launches = argc > 1 ? atoi(argv[1]) : 50, iters = 200, grid = 512;
  std::vector<_Float16> hA(64 * 8), hB(8 * 64 * 8);
  srand(1);
  for (auto& v : hA) v = static_cast<_Float16>((rand() % 2001 - 1000) / 8.0f);
  for (auto& v : hB) v = static_cast<_Float16>((rand() % 2001 - 1000) / 8.0f);
  f16x8 *dA, *dB;
  f32x4 *o0, *o1;
  hipMalloc(&dA, hA.size() * 2); hipMalloc(&dB, hB.size() * 2);
  hipMalloc(&o0, grid * 512 * 16); hipMalloc(&o1, grid * 512 * 16);
  hipMemcpy(dA, hA.data(), hA.size() * 2, hipMemcpyHostToDevice);
  hipMemcpy(dB, hB.data(), hB.size() * 2, hipMemcpyHostToDevice);
  std::vector<float> h0(grid * 512 * 4), h1(grid * 512 * 4);
  long bad_total = 0;
  for (int l = 0; l < launches; ++l) {
    hipLaunchKernelGGL(k<0>, dim3(grid), dim3(512), 0, 0, dA, dB, o0, iters);
    hipLaunchKernelGGL(k<1>, dim3(grid), dim3(512), 0, 0, dA, dB, o1, iters);
    if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 2; }
    hipMemcpy(h0.data(), o0, h0.size() * 4, hipMemcpyDeviceToHost);
    hipMemcpy(h1.data(), o1, h1.size() * 4, hipMemcpyDeviceToHost);
    long bad = 0;
    for (size_t i = 0; i < h0.size(); ++i)
      if (h0[i] != h1[i]) {
        if (bad_total + bad < 8) printf("  launch %d elem %zu (thread %zu, r %zu): in-place %g shifted %g\n", l, i, (i / 4) % 512, i % 4, h0[i], h1[i]);
        ++bad;
      }
    bad_total += bad;
  }
  printf("mfma partial overlap: %ld mismatching words over %d launches (%d x 512 threads x %d iterations each)\n", bad_total, launches, grid, iters);
  return 0;
}
