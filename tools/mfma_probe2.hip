// Diagnostic 2: does the fp32 MFMA rate survive (a) distinct A/B registers per instruction, (b) LDS fragment reads in the loop,
// (c) a work-group barrier every 128 MFMAs?   hipcc -O3 --offload-arch=gfx950 tools/mfma_probe2.hip -o tools/bin/mfma_probe2
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(512) void probe(float* out, const float* in, int iters, const float* Wg, const float* Xg) {
  __shared__ __attribute__((aligned(16))) float L[2 * 128 * 68];
  const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, g = lane >> 4, cw = (tid >> 6) >> 2;
  for (int i = tid; i < 2 * 128 * 68; i += 512) L[i] = in[i & 1023];
  f32x4 a[2][4];
  for (int mt = 0; mt < 2; ++mt)
    for (int kq = 0; kq < 4; ++kq) a[mt][kq] = *reinterpret_cast<const f32x4*>(in + 64 * (mt * 4 + kq) + lane);
  __syncthreads();
  f32x4 acc[2][4];
  for (int mt = 0; mt < 2; ++mt)
    for (int tt = 0; tt < 4; ++tt) acc[mt][tt] = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 bfix[4];
  for (int tt = 0; tt < 4; ++tt) bfix[tt] = *reinterpret_cast<const f32x4*>(in + 512 + 4 * lane + tt);
  f32x4 wreg[4];
  for (int r = 0; r < 4; ++r) wreg[r] = bfix[r];
  f32x4 an[2][4];
  const float* xrow0 = Xg + (size_t)(blockIdx.x * 128 + 32 * ((tid >> 6) & 3) + l15) * 1024 + 4 * g;
  for (int it = 0; it < iters; ++it) {
    const float* Wl = L + ((it & 1) * 128 + 64 * cw + l15) * 68 + 4 * g;
    if (MODE >= 4) {
#pragma unroll
      for (int r = 0; r < 4; ++r) wreg[r] = *reinterpret_cast<const f32x4*>(Wg + (size_t)(32 * r + (tid >> 4)) * 1024 + (it & 15) * 64 + 4 * (tid & 15));
    }
    if (MODE == 5) {  // MFMA-fragment-shaped: 16 rows x 64 B per instruction
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int kq = 0; kq < 4; ++kq) an[mt][kq] = *reinterpret_cast<const f32x4*>(xrow0 + (size_t)mt * 16 * 1024 + (it & 15) * 64 + 16 * kq);
    }
    if (MODE == 6) {  // same bytes, line-shaped: 4 rows x 256 B per instruction (not usable as fragments: timing only)
      const float* xl = Xg + (size_t)(blockIdx.x * 128 + 32 * ((tid >> 6) & 3) + (lane >> 4)) * 1024 + 4 * (lane & 15);
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int kq = 0; kq < 4; ++kq) an[mt][kq] = *reinterpret_cast<const f32x4*>(xl + (size_t)(mt * 16 + kq * 4) * 1024 + (it & 15) * 64);
    }
    asm volatile("" ::: "memory");
#pragma unroll
    for (int kq = 0; kq < 4; ++kq) {
      f32x4 b[4];
#pragma unroll
      for (int tt = 0; tt < 4; ++tt) {
        if (MODE >= 1) b[tt] = *reinterpret_cast<const f32x4*>(Wl + 16 * tt * 68 + 16 * kq);
        else b[tt] = bfix[tt];
      }
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
          for (int tt = 0; tt < 4; ++tt) acc[mt][tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt][kq][s], b[tt][s], acc[mt][tt], 0, 0, 0);
    }
    if (MODE >= 3) {
#pragma unroll
      for (int r = 0; r < 4; ++r) *reinterpret_cast<f32x4*>(&L[(((it + 1) & 1) * 128 + 32 * r + (tid >> 4)) * 68 + 4 * (tid & 15)]) = wreg[r];
    }
    if (MODE >= 5) {
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int kq = 0; kq < 4; ++kq) a[mt][kq] = an[mt][kq];
    }
    if (MODE >= 2) __syncthreads();
  }
  float r = 0.f;
  for (int mt = 0; mt < 2; ++mt)
    for (int tt = 0; tt < 4; ++tt) r += acc[mt][tt][0] + acc[mt][tt][3];
  out[blockIdx.x * 512 + tid] = r;
}

int main() {
  float *out, *in, *Wg, *Xg;
  hipMalloc(&out, 256 * 512 * 4);
  hipMalloc(&Wg, 128 * 1024 * 4);
  hipMalloc(&Xg, (size_t)32768 * 1024 * 4);
  hipMemset(Wg, 0, 128 * 1024 * 4);
  hipMemset(Xg, 0, (size_t)32768 * 1024 * 4);
  hipMalloc(&in, 4096 * 4);
  hipMemset(in, 0, 4096 * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const char* names[7] = {"distinct regs, no LDS      ", "+ 16 ds_read_b128 / 128 MFMA", "+ barrier every 128 MFMA   ", "+ 4 ds_write_b128 / chunk   ",
                          "+ W chunk loads (L2)        ", "+ A fragment loads (HBM)    ", "+ A line-shaped loads (HBM) "};
  for (int mode = 0; mode < 7; ++mode) {
    const int iters = 4000;
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      if (mode == 0) hipLaunchKernelGGL(probe<0>, dim3(256), dim3(512), 0, 0, out, in, iters, Wg, Xg);
      if (mode == 1) hipLaunchKernelGGL(probe<1>, dim3(256), dim3(512), 0, 0, out, in, iters, Wg, Xg);
      if (mode == 2) hipLaunchKernelGGL(probe<2>, dim3(256), dim3(512), 0, 0, out, in, iters, Wg, Xg);
      if (mode == 3) hipLaunchKernelGGL(probe<3>, dim3(256), dim3(512), 0, 0, out, in, iters, Wg, Xg);
      if (mode == 4) hipLaunchKernelGGL(probe<4>, dim3(256), dim3(512), 0, 0, out, in, iters, Wg, Xg);
      if (mode == 5) hipLaunchKernelGGL(probe<5>, dim3(256), dim3(512), 0, 0, out, in, iters, Wg, Xg);
      if (mode == 6) hipLaunchKernelGGL(probe<6>, dim3(256), dim3(512), 0, 0, out, in, iters, Wg, Xg);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      hipEventElapsedTime(&ms, e0, e1);
    }
    const double flop = double(iters) * 128 * 2048.0 * 8 * 256;
    printf("%s: %.3f ms  %.1f TFLOP/s\n", names[mode], ms, flop / ms * 1e-9);
  }
  return 0;
}
