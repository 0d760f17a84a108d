// Which path limits a streaming CU: the chip, or something its shader engine / shader array shares?
// 256 work-groups of 512 threads with 140 KiB of LDS (one per CU).  Each reads its hardware id (XCC, SE, SH, CU) and streams
// 4 MiB from HBM (8 waves x 16 KiB in flight) if its id is selected by `mode`; every work-group reports id + duration.
//   mode 0: every CU streams          mode 1: CUs with even cu_id          mode 2: CUs of even SEs (all CUs of half the SEs)
//   mode 3: the first half of the cu_ids of every (SE, SH)                 mode 4: one quarter (cu_id & 3) == 0
// build: hipcc -O3 --offload-arch=gfx950 tools/se_probe.hip -o tools/bin/se_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>
#include <algorithm>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(512) void probe(const f32x4* __restrict__ hbm, float* out, unsigned* __restrict__ ids,
                                             unsigned long long* __restrict__ ticks, int mode, int iters) {
  extern __shared__ float smem[];
  unsigned hw, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  const unsigned cu = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
  bool on = true;
  if (mode == 1) on = (cu & 1) == 0;
  if (mode == 2) on = (se & 1) == 0;
  if (mode == 3) on = cu < 4;
  if (mode == 4) on = (cu & 3) == 0;
  if (mode == 5) on = (xcc & 1) == 0;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  f32x4 acc = {0, 0, 0, 0};
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  if (on) {
    const f32x4* hp = hbm + (static_cast<size_t>(blockIdx.x) * 8 + wv) * static_cast<size_t>(iters) * 16 * 64 + lane;
    for (int it = 0; it < iters; ++it) {
      f32x4 hb[16];
#pragma unroll
      for (int d = 0; d < 16; ++d) hb[d] = __builtin_nontemporal_load(hp + (static_cast<size_t>(it) * 16 + d) * 64);
#pragma unroll
      for (int d = 0; d < 16; ++d) acc += hb[d];
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) {
    ticks[blockIdx.x] = t1 - t0;
    ids[blockIdx.x] = (on ? 1u << 31 : 0u) | ((xcc & 15) << 16) | (se << 8) | (sh << 4) | cu;
  }
  if (acc.x + acc.y + acc.z + acc.w == 123.456f) { out[0] = 1; smem[threadIdx.x] = acc.x; }
}

int main() {
  const size_t bytes = 1ull << 30;
  f32x4* hbm; float* out; unsigned* ids; unsigned long long* ticks;
  hipMalloc(&hbm, bytes + (64 << 20)); hipMalloc(&out, 64); hipMalloc(&ids, 256 * 4); hipMalloc(&ticks, 256 * 8);
  hipMemset(hbm, 1, bytes);
  const int lds = 140 * 1024, iters = 32;
  hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  for (int mode = 0; mode < 6; ++mode) {
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(probe, dim3(256), dim3(512), lds, 0, hbm, out, ids, ticks, mode, iters);
    hipDeviceSynchronize();
    unsigned hi[256]; unsigned long long ht[256];
    hipMemcpy(hi, ids, sizeof hi, hipMemcpyDeviceToHost); hipMemcpy(ht, ticks, sizeof ht, hipMemcpyDeviceToHost);
    double ts = 0; int n = 0;
    for (int i = 0; i < 256; ++i) if (hi[i] >> 31) { ts += ht[i]; ++n; }
    printf("mode %d: %3d CUs stream, %.1f GB/s per CU (mean %.1f us) -> %.2f TB/s\n", mode, n, n ? double(iters) * 8 * 16384 / (ts / n * 10.0) : 0.0,
           n ? ts / n / 100.0 : 0.0, n ? n * double(iters) * 8 * 16384 / (ts / n * 10.0) / 1e3 : 0.0);
    if (mode == 0) {
      std::map<unsigned, int> per_sesh; std::map<unsigned, int> per_xcc;
      for (int i = 0; i < 256; ++i) { per_sesh[(hi[i] >> 4) & 0xfffff]++; per_xcc[(hi[i] >> 16) & 15]++; }
      printf("  distinct (xcc, se, sh): %zu; xcc ids: %zu\n  first 40 blocks (xcc.se.sh.cu):", per_sesh.size(), per_xcc.size());
      for (int i = 0; i < 40; ++i) printf(" %u.%u.%u.%u", (hi[i] >> 16) & 15, (hi[i] >> 8) & 7, (hi[i] >> 4) & 1, hi[i] & 15);
      printf("\n  CUs per (xcc 0, se, sh):");
      for (auto& kv : per_sesh) if ((kv.first >> 12) == 0) printf(" se%u.sh%u:%d", (kv.first >> 4) & 7, kv.first & 1, kv.second);
      printf("\n");
    }
  }
  return 0;
}
