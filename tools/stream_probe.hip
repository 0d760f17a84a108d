// Microbenchmark: what HBM read rate can the attention kernel's pair-embedding access pattern reach?
//   copy-like  : grid-stride float4 reads, many waves per CU
//   rows       : 512-thread work-groups, ONE per CU (forced by a large LDS request), wave w reads rows 2w, 2w+1 of a 16-row tile,
//                32 KiB per row as 32 loads of 1 KiB (float4 per lane), optional non-temporal hint, DEPTH loads in flight.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void copy_like(const f32x4* __restrict__ p, size_t n, float* out) {
  f32x4 acc = {0, 0, 0, 0};
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    f32x4 v = __builtin_nontemporal_load(p + i);
    acc += v;
  }
  if (acc.x + acc.y + acc.z + acc.w == 123.456f) out[0] = 1;
}

template <int DEPTH, bool NT_HINT>
__global__ __launch_bounds__(512) void rows_kernel(const f32x4* __restrict__ p, float* out, int lds_bytes_unused) {
  extern __shared__ float smem[];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const size_t tile = blockIdx.x;  // 16 rows x 32 KiB = 512 KiB per tile
  f32x4 acc = {0, 0, 0, 0};
  for (int ii = 0; ii < 2; ++ii) {
    const f32x4* row = p + (tile * 16 + 2 * wv + ii) * (32768 / 16) + lane;
    f32x4 buf[DEPTH];
#pragma unroll
    for (int c0 = 0; c0 < 32; c0 += DEPTH) {
#pragma unroll
      for (int d = 0; d < DEPTH; ++d) buf[d] = NT_HINT ? __builtin_nontemporal_load(row + (c0 + d) * 64) : row[(c0 + d) * 64];
#pragma unroll
      for (int d = 0; d < DEPTH; ++d) acc += buf[d];
    }
  }
  if (acc.x + acc.y + acc.z + acc.w == 123.456f) { out[0] = 1; smem[threadIdx.x] = acc.x; }
}

// the same pattern from a limited number of CUs: gridDim.x work-groups (one per CU) walk all 2048 tiles - what can ONE CU pull when
// only some of the CUs stream at a time (the attention kernel: ~40 % of the CUs are in their streaming phase at any moment)?
template <int DEPTH>
__global__ __launch_bounds__(512) void rows_some_cus_kernel(const f32x4* __restrict__ p, float* out, int ntiles) {
  extern __shared__ float smem[];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  f32x4 acc = {0, 0, 0, 0};
  for (size_t tile = blockIdx.x; tile < (size_t)ntiles; tile += gridDim.x)
    for (int ii = 0; ii < 2; ++ii) {
      const f32x4* row = p + (tile * 16 + 2 * wv + ii) * (32768 / 16) + lane;
      f32x4 buf[DEPTH];
#pragma unroll
      for (int c0 = 0; c0 < 32; c0 += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) buf[d] = __builtin_nontemporal_load(row + (c0 + d) * 64);
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) acc += buf[d];
      }
    }
  if (acc.x + acc.y + acc.z + acc.w == 123.456f) { out[0] = 1; smem[threadIdx.x] = acc.x; }
}

template <typename F>
float time_ms(F f, int iters = 20) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  f(); hipDeviceSynchronize();
  hipEventRecord(a);
  for (int i = 0; i < iters; ++i) f();
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  return ms / iters;
}

int main() {
  const size_t bytes = 1ull << 30;  // 1 GiB: the pair embedding of 256 K=128 patches
  f32x4* p; float* out;
  hipMalloc(&p, 2 * bytes); hipMalloc(&out, 64);
  hipMemset(p, 1, 2 * bytes);
  const size_t n = bytes / 16;
  float ms = time_ms([&] { hipLaunchKernelGGL(copy_like, dim3(256 * 8), dim3(256), 0, 0, p, n, out); });
  printf("copy-like read, 2048x256 threads        : %.3f ms  %.0f GB/s\n", ms, bytes / ms / 1e6);
  const int lds = 140 * 1024;
#define RUN(D, NT)                                                                                              \
  do {                                                                                                          \
    hipFuncSetAttribute((const void*)rows_kernel<D, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);     \
    float t = time_ms([&] { hipLaunchKernelGGL((rows_kernel<D, NT>), dim3(2048), dim3(512), lds, 0, p, out, 0); }); \
    printf("rows pattern depth %2d nt=%d, 1 WG/CU       : %.3f ms  %.0f GB/s\n", D, (int)NT, t, bytes / t / 1e6);   \
  } while (0)
  RUN(4, true); RUN(8, true); RUN(16, true); RUN(32, true); RUN(32, false); RUN(16, false);
  // two work-groups per CU (smaller LDS request)
  const int lds2 = 70 * 1024;
  hipFuncSetAttribute((const void*)rows_kernel<32, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds2);
  float t2 = time_ms([&] { hipLaunchKernelGGL((rows_kernel<32, true>), dim3(2048), dim3(512), lds2, 0, p, out, 0); });
  printf("rows pattern depth 32 nt=1, 2 WG/CU       : %.3f ms  %.0f GB/s\n", t2, bytes / t2 / 1e6);
  for (int ncu : {16, 32, 64, 102, 128, 192, 256}) {
    hipFuncSetAttribute((const void*)rows_some_cus_kernel<16>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    float t = time_ms([&] { hipLaunchKernelGGL((rows_some_cus_kernel<16>), dim3(ncu), dim3(512), lds, 0, p, out, 2048); }, 5);
    printf("rows pattern depth 16, %3d CUs streaming   : %.3f ms  %.0f GB/s total  %.1f GB/s per CU\n", ncu, t, bytes / t / 1e6, bytes / t / 1e6 / ncu);
  }
  return 0;
}
