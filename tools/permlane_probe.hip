// Diagnostic: v_permlane32_swap / v_permlane16_swap as xor-32 / xor-16 reductions (vs __shfl_xor).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
__global__ void k(const float* in, float* o) {
  const int l = threadIdx.x;
  float x = in[l];
  u32x2 r = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, x), __builtin_bit_cast(unsigned, x), false, false);
  o[l] = __builtin_bit_cast(float, r[0]);
  o[64 + l] = __builtin_bit_cast(float, r[1]);
  u32x2 r2 = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, x), __builtin_bit_cast(unsigned, x), false, false);
  o[128 + l] = __builtin_bit_cast(float, r2[0]);
  o[192 + l] = __builtin_bit_cast(float, r2[1]);
  float m = fmaxf(x, __shfl_xor(x, 16));
  m = fmaxf(m, __shfl_xor(m, 32));
  o[256 + l] = m;
  // the working form: the swap in assembly on two copies (see attention_flash.hip)
  float a = x, b = x;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
  float y = fmaxf(a, b), c = y, d = y;
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(c), "+v"(d));
  o[320 + l] = fmaxf(c, d);
}
int main() {
  float h[64], out[384], *d, *o;
  for (int i = 0; i < 64; ++i) h[i] = float((i * 37) % 64);
  hipMalloc(&d, 256); hipMalloc(&o, 384 * 4);
  hipMemcpy(d, h, 256, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o);
  hipMemcpy(out, o, 384 * 4, hipMemcpyDeviceToHost);
  const char* names[] = {"p32 r0", "p32 r1", "p16 r0", "p16 r1", "shfl max", "swap max"};
  printf("in     :"); for (int i = 0; i < 64; ++i) printf(" %2.0f", h[i]); printf("\n");
  for (int a = 0; a < 6; ++a) { printf("%-7s:", names[a]); for (int i = 0; i < 64; ++i) printf(" %2.0f", out[a * 64 + i]); printf("\n"); }
  int bad = 0; for (int i = 0; i < 64; ++i) bad += out[256 + i] != out[320 + i];
  printf("mismatches swap-max vs shfl-max: %d\n", bad);
  return 0;
}
