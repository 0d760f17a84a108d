// Microbenchmark: does a CU's L2-hit traffic share its throughput limit with its HBM stream?
// One 512-thread work-group per CU (forced by a large LDS request).  Per wave and iteration: NH loads of 1 KiB from a 1 GiB HBM-resident
// stream (non-temporal, each byte once) and NL loads of 1 KiB from a small table (2 MiB, re-read by every CU: L2-resident after the
// first pass).  Reports the time of the HBM-only, L2-only and mixed loops; if mixed ~ max(HBM, L2) the two paths are parallel, if
// mixed ~ HBM + L2 they share one window.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NH, int NL>
__global__ __launch_bounds__(512) void mixed(const f32x4* __restrict__ hbm, const f32x4* __restrict__ tab, float* out, int iters) {
  extern __shared__ float smem[];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  f32x4 acc = {0, 0, 0, 0};
  // per CU: iters * 8 waves * NH KiB of the stream, contiguous per wave
  const f32x4* hp = hbm + (static_cast<size_t>(blockIdx.x) * 8 + wv) * static_cast<size_t>(iters) * NH * 64 + lane;
  const f32x4* tp = tab + lane;
  for (int it = 0; it < iters; ++it) {
    f32x4 hb[NH > 0 ? NH : 1], lb[NL > 0 ? NL : 1];
#pragma unroll
    for (int d = 0; d < NH; ++d) hb[d] = __builtin_nontemporal_load(hp + (static_cast<size_t>(it) * NH + d) * 64);
#pragma unroll
    for (int d = 0; d < NL; ++d) lb[d] = tp[(((it * NL + d) * 8 + wv) * 37 % 2048) * 64];  // 2048 KiB-blocks = 2 MiB table
#pragma unroll
    for (int d = 0; d < NH; ++d) acc += hb[d];
#pragma unroll
    for (int d = 0; d < NL; ++d) acc += lb[d];
  }
  if (acc.x + acc.y + acc.z + acc.w == 123.456f) { out[0] = 1; smem[threadIdx.x] = acc.x; }
}

// the two streams on DIFFERENT waves of the work-group (no in-order coupling inside a wave): waves < WH stream HBM (16 KiB per
// iteration), the other waves read the table (16 KiB per iteration); MODE 1 = HBM waves only, 2 = table waves only, 3 = both
template <int WH, int MODE>
__global__ __launch_bounds__(512) void split(const f32x4* __restrict__ hbm, const f32x4* __restrict__ tab, float* out, int iters) {
  extern __shared__ float smem[];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  f32x4 acc = {0, 0, 0, 0};
  if (wv < WH) {
    if (MODE & 1) {
      const f32x4* hp = hbm + (static_cast<size_t>(blockIdx.x) * 8 + wv) * static_cast<size_t>(iters) * 16 * 64 + lane;
      for (int it = 0; it < iters; ++it) {
        f32x4 hb[16];
#pragma unroll
        for (int d = 0; d < 16; ++d) hb[d] = __builtin_nontemporal_load(hp + (static_cast<size_t>(it) * 16 + d) * 64);
#pragma unroll
        for (int d = 0; d < 16; ++d) acc += hb[d];
      }
    }
  } else if (MODE & 2) {
    const f32x4* tp = tab + lane;
    for (int it = 0; it < 4 * iters; ++it) {  // the table waves run 4x the iterations (they are ~4x faster per iteration)
      f32x4 lb[16];
#pragma unroll
      for (int d = 0; d < 16; ++d) lb[d] = tp[(((it * 16 + d) * 8 + wv) * 37 % 2048) * 64];
#pragma unroll
      for (int d = 0; d < 16; ++d) acc += lb[d];
    }
  }
  if (acc.x + acc.y + acc.z + acc.w == 123.456f) { out[0] = 1; smem[threadIdx.x] = acc.x; }
}

// roles by work-group: blocks < NS stream HBM (8 waves x 16 KiB in flight, `iters` x 128 KiB each), the others read the L2-resident
// table in a loop of `titers` iterations (16 KiB per wave and iteration).  Each streaming block reports its own duration in 100 MHz
// ticks (s_memrealtime): the HBM rate of a streaming CU while the REST of the chip does L2-hit traffic (the attention kernel's mix).
__global__ __launch_bounds__(512) void roles(const f32x4* __restrict__ hbm, const f32x4* __restrict__ tab, float* out,
                                             unsigned long long* __restrict__ ticks, int NS, int iters, int titers) {
  extern __shared__ float smem[];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  f32x4 acc = {0, 0, 0, 0};
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  if (static_cast<int>(blockIdx.x) < NS) {
    const f32x4* hp = hbm + (static_cast<size_t>(blockIdx.x) * 8 + wv) * static_cast<size_t>(iters) * 16 * 64 + lane;
    for (int it = 0; it < iters; ++it) {
      f32x4 hb[16];
#pragma unroll
      for (int d = 0; d < 16; ++d) hb[d] = __builtin_nontemporal_load(hp + (static_cast<size_t>(it) * 16 + d) * 64);
#pragma unroll
      for (int d = 0; d < 16; ++d) acc += hb[d];
    }
  } else {
    const f32x4* tp = tab + lane;
    for (int it = 0; it < titers; ++it) {
      f32x4 lb[16];
#pragma unroll
      for (int d = 0; d < 16; ++d) lb[d] = tp[(((it * 16 + d) * 8 + wv) * 37 % 2048) * 64];
#pragma unroll
      for (int d = 0; d < 16; ++d) acc += lb[d];
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
  if (acc.x + acc.y + acc.z + acc.w == 123.456f) { out[0] = 1; smem[threadIdx.x] = acc.x; }
}

// the phase-2 loop shape of the attention kernel: a wave streams 8 KiB steps with two steps of lookahead (8..16 loads in flight);
// per step NW ds_write_b128 of the loaded registers and NR ds_read_b128 from the wave's own LDS region, NM f16 MFMAs.
// Does the LDS traffic of a CU slow its HBM stream?
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
template <int NW, int NR, int NM>
__global__ __launch_bounds__(512) void stream_lds(const f32x4* __restrict__ hbm, float* out, unsigned long long* __restrict__ ticks,
                                                  int steps) {
  extern __shared__ float smem[];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  f32x4* my = reinterpret_cast<f32x4*>(smem) + wv * (16 * 64) + lane;  // 16 KiB per wave
  f32x4 acc = {0, 0, 0, 0}, macc = {0, 0, 0, 0};
  const f32x4* hp = hbm + (static_cast<size_t>(blockIdx.x) * 8 + wv) * static_cast<size_t>(steps) * 8 * 64 + lane;
  f32x4 ev[2][8];
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int d = 0; d < 8; ++d) ev[s][d] = __builtin_nontemporal_load(hp + (static_cast<size_t>(s) * 8 + d) * 64);
  for (int st = 0; st < steps; st += 2) {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
#pragma unroll
      for (int m = 0; m < NM; ++m)
        macc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, ev[s][m & 7]), __builtin_bit_cast(f16x8, ev[s][(m + 1) & 7]), macc, 0, 0, 0);
#pragma unroll
      for (int d = 0; d < NW; ++d) my[(d & 15) * 64] = ev[s][d & 7];
      if (NW == 0) {
#pragma unroll
        for (int d = 0; d < 8; ++d) acc += ev[s][d];
      }
      asm volatile("" ::: "memory");
      if (st + s + 2 < steps) {
#pragma unroll
        for (int d = 0; d < 8; ++d) ev[s][d] = __builtin_nontemporal_load(hp + (static_cast<size_t>(st + s + 2) * 8 + d) * 64);
      }
      asm volatile("" ::: "memory");
#pragma unroll
      for (int d = 0; d < NR; ++d) acc += my[((d + 3) & 15) * 64];
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
  acc += macc;
  if (acc.x + acc.y + acc.z + acc.w == 123.456f) out[0] = 1;
}

template <typename F>
float time_ms(F f, int reps = 10) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  f(); hipDeviceSynchronize();
  hipEventRecord(a);
  for (int i = 0; i < reps; ++i) f();
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  return ms / reps;
}

int main() {
  const size_t bytes = 1ull << 30;
  f32x4 *hbm, *tab; float* out;
  hipMalloc(&hbm, bytes + (64 << 20)); hipMalloc(&tab, 2 << 20); hipMalloc(&out, 64);
  hipMemset(hbm, 1, bytes); hipMemset(tab, 1, 2 << 20);
  const int lds = 140 * 1024;
#define RUN(NH, NL, NCU)                                                                                               \
  do {                                                                                                                 \
    const int iters = 16;  /* per wave: 16 x NH KiB of stream, 16 x NL KiB of table */                                 \
    hipFuncSetAttribute((const void*)mixed<NH, NL>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);                  \
    float t = time_ms([&] { hipLaunchKernelGGL((mixed<NH, NL>), dim3(NCU), dim3(512), lds, 0, hbm, tab, out, iters); }); \
    const double hb = double(NCU) * 8 * iters * NH * 1024, lb = double(NCU) * 8 * iters * NL * 1024;                   \
    printf("NH=%2d NL=%2d CUs=%3d: %.4f ms  stream %.0f GB/s (%.1f per CU)  table %.0f GB/s (%.1f per CU)\n", NH, NL, NCU, t, \
           hb / t / 1e6, hb / t / 1e6 / NCU, lb / t / 1e6, lb / t / 1e6 / NCU);                                        \
  } while (0)
  for (int ncu : {64, 128, 256}) {
    if (ncu == 64) { RUN(16, 0, 64); RUN(0, 16, 64); RUN(16, 16, 64); RUN(8, 8, 64); RUN(16, 8, 64); }
    if (ncu == 128) { RUN(16, 0, 128); RUN(0, 16, 128); RUN(16, 16, 128); RUN(8, 8, 128); RUN(16, 8, 128); }
    if (ncu == 256) { RUN(16, 0, 256); RUN(0, 16, 256); RUN(16, 16, 256); RUN(8, 8, 256); RUN(16, 8, 256); }
  }
#define RUNS(WH, MODE, NCU)                                                                                            \
  do {                                                                                                                 \
    const int iters = 16;                                                                                              \
    hipFuncSetAttribute((const void*)split<WH, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);                \
    float t = time_ms([&] { hipLaunchKernelGGL((split<WH, MODE>), dim3(NCU), dim3(512), lds, 0, hbm, tab, out, iters); }); \
    const double hb = (MODE & 1) ? double(NCU) * WH * iters * 16 * 1024 : 0, lb = (MODE & 2) ? double(NCU) * (8 - WH) * 4 * iters * 16 * 1024 : 0; \
    printf("split WH=%d mode=%d CUs=%3d: %.4f ms  stream %.0f GB/s (%.1f per CU)  table %.0f GB/s (%.1f per CU)\n", WH, MODE, NCU, t, \
           hb / t / 1e6, hb / t / 1e6 / NCU, lb / t / 1e6, lb / t / 1e6 / NCU);                                        \
  } while (0)
  {
    unsigned long long* ticks; hipMalloc(&ticks, 256 * 8);
    hipFuncSetAttribute((const void*)roles, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    for (int ns : {64, 105, 128}) {
      for (int titers : {0, 200, 400}) {
        const int iters = 32;  // 4 MiB per streaming block
        for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(roles, dim3(256), dim3(512), lds, 0, hbm, tab, out, ticks, ns, iters, titers);
        hipDeviceSynchronize();
        unsigned long long h[256]; hipMemcpy(h, ticks, 256 * 8, hipMemcpyDeviceToHost);
        double ts = 0, tt = 0; for (int i = 0; i < ns; ++i) ts += h[i]; for (int i = ns; i < 256; ++i) tt += h[i];
        ts /= ns; tt /= (256 - ns);
        const double sb = double(iters) * 8 * 16 * 1024, tb = double(titers) * 8 * 16 * 1024;
        printf("roles: %3d CUs stream HBM, %3d CUs read the L2 table (%d iters): stream %.1f GB/s per CU (%.1f us), table %.1f GB/s per CU (%.1f us)\n",
               ns, 256 - ns, titers, sb / (ts * 10.0), ts / 100.0, titers ? tb / (tt * 10.0) : 0.0, tt / 100.0);
      }
    }
  }
#define RUNL(NW, NR, NM, NCU)                                                                                          \
  do {                                                                                                                 \
    unsigned long long* ticks; hipMalloc(&ticks, 256 * 8);                                                             \
    const int steps = 64; /* 512 KiB per wave, 4 MiB per work-group */                                                 \
    hipFuncSetAttribute((const void*)stream_lds<NW, NR, NM>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);         \
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL((stream_lds<NW, NR, NM>), dim3(NCU), dim3(512), lds, 0, hbm, out, ticks, steps); \
    hipDeviceSynchronize();                                                                                            \
    unsigned long long h[256]; hipMemcpy(h, ticks, NCU * 8, hipMemcpyDeviceToHost);                                    \
    double ts = 0; for (int i = 0; i < NCU; ++i) ts += h[i]; ts /= NCU;                                                \
    printf("stream_lds: %3d CUs, per 8 KiB step %2d ds_write_b128 + %2d ds_read_b128 + %2d MFMA: %.1f GB/s per CU (%.1f us)\n", NCU, NW, NR, NM, \
           double(steps) * 8 * 8192 / (ts * 10.0), ts / 100.0);                                                        \
    hipFree(ticks);                                                                                                    \
  } while (0)
  RUNL(0, 0, 0, 105); RUNL(8, 0, 0, 105); RUNL(0, 8, 0, 105); RUNL(8, 8, 0, 105); RUNL(10, 12, 0, 105); RUNL(10, 12, 24, 105); RUNL(0, 0, 24, 105);
  RUNL(16, 16, 0, 105); RUNL(0, 0, 0, 64); RUNL(10, 12, 24, 64); RUNL(0, 0, 0, 256); RUNL(10, 12, 24, 256);
  RUNS(4, 1, 128); RUNS(4, 2, 128); RUNS(4, 3, 128);
  RUNS(4, 1, 256); RUNS(4, 2, 256); RUNS(4, 3, 256);
  RUNS(6, 1, 128); RUNS(6, 2, 128); RUNS(6, 3, 128);
  return 0;
}
