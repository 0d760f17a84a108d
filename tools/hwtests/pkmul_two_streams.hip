// Hardware question (gfx950): does a VALU-ONLY kernel that issues v_pk_mul_f32 ... op_sel:[0,1] compute wrong low results when ANOTHER
// kernel's waves run f16 / bf16 MFMAs on the same SIMDs (a second stream)?  Round 5 measured the form wrong in lanes 48-63 with the MFMAs in
// the SAME wave (pkmul_mfma.hip); rounds 4-5 saw heads_finish_kernel / reverse_update_philox_kernel (VALU only, both holding this form)
// wrong in lanes 48-63 while bf16 x 6 GEMM work-groups of a second stream were resident (profiles/r04_two_queue.md).  If this program
// counts wrong results for "victim op_sel:[0,1] + bf16 aggressor" and none for the controls, the two findings are one hazard.
//   victim    : 1 wave per work-group, no MFMA, no LDS: a loop of the packed op under test, results checked in registers
//   aggressor : 256-thread work-groups looping over v_mfma_f32_16x16x32_bf16 (or _f16, or the fp32 16x16x4 form as a control)
// build: hipcc --offload-arch=gfx950 -O2 -o pkmul_two_streams pkmul_two_streams.hip      usage: pkmul_two_streams [rounds]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// FORM 0: op_sel:[0,1] (the hazard form)   1: op_sel:[1,0] (measured clean in-wave)   2: no op_sel   3: v_pk_fma_f32 op_sel:[0,1,0]
template <int FORM>
__global__ __launch_bounds__(64) void victim(unsigned* __restrict__ bad, float* __restrict__ sink, int iters) {
  const int lane = threadIdx.x & 63;
  f32x2 s0 = {1.0f / 2048.0f + lane * 1e-6f, 3.0f}, s1 = {5.0f, 1.0f / 256.0f};
  unsigned nbad = 0;
  float keep = 0.f;
  for (int it = 0; it < iters; ++it) {
    f32x2 d;
    s0[0] += 1e-7f;
    float want_lo, want_hi = s0[1] * s1[1];
    if (FORM == 0) { asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1]\n s_nop 1" : "=&v"(d) : "v"(s0), "v"(s1)); want_lo = s0[0] * s1[1]; }
    if (FORM == 1) { asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0]\n s_nop 1" : "=&v"(d) : "v"(s0), "v"(s1)); want_lo = s0[1] * s1[0]; }
    if (FORM == 2) { asm volatile("v_pk_mul_f32 %0, %1, %2\n s_nop 1" : "=&v"(d) : "v"(s0), "v"(s1)); want_lo = s0[0] * s1[0]; }
    if (FORM == 3) {
      asm volatile("v_pk_fma_f32 %0, %1, %2, %1 op_sel:[0,1,0]\n s_nop 1" : "=&v"(d) : "v"(s0), "v"(s1));
      want_lo = __builtin_fmaf(s0[0], s1[1], s0[0]);
      want_hi = __builtin_fmaf(s0[1], s1[1], s0[1]);
    }
    if (d[0] != want_lo || d[1] != want_hi) ++nbad;
    keep += d[0];
  }
  if (nbad) atomicAdd(bad + (lane >> 4), nbad);
  sink[blockIdx.x * 64 + threadIdx.x] = keep;
}

// KIND 0: bf16 16x16x32   1: f16 16x16x32   2: fp32 16x16x4 (control: did not trigger the in-wave hazard)   3: VALU only (control)
template <int KIND>
__global__ __launch_bounds__(256) void aggressor(const float* __restrict__ src, float* __restrict__ sink, int iters) {
  const int lane = threadIdx.x & 63;
  bf16x8 ab, bb;
  f16x8 ah, bh;
  for (int i = 0; i < 8; ++i) {
    const float v = src[(lane * 8 + i) & 511];
    ab[i] = static_cast<__bf16>(v); bb[i] = static_cast<__bf16>(v * 0.5f);
    ah[i] = static_cast<_Float16>(v); bh[i] = static_cast<_Float16>(v * 0.5f);
  }
  f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0, acc2 = acc0, acc3 = acc0;
  float x = src[lane], y = 1.0f;
  for (int it = 0; it < iters; ++it) {
    if (KIND == 0) {
      acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab, bb, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab, bb, acc1, 0, 0, 0);
      acc2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab, bb, acc2, 0, 0, 0);
      acc3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab, bb, acc3, 0, 0, 0);
    } else if (KIND == 1) {
      acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, acc1, 0, 0, 0);
      acc2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, acc2, 0, 0, 0);
      acc3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, acc3, 0, 0, 0);
    } else if (KIND == 2) {
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, acc1, 0, 0, 0);
      acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, acc2, 0, 0, 0);
      acc3 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, acc3, 0, 0, 0);
    } else {
      y = __builtin_fmaf(y, 1.0000001f, x);
      acc0[0] += y;
    }
  }
  sink[blockIdx.x * 256 + threadIdx.x] = acc0[0] + acc1[1] + acc2[2] + acc3[3] + y;
}

#define CHECK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { printf("%s: %s\n", #e, hipGetErrorString(r_)); return 2; } } while (0)

int main(int argc, char** argv) {
  const int rounds = argc > 1 ? atoi(argv[1]) : 8;
  const int vgrid = 2048, viters = 200000, agrid = 1024, aiters = 100000;  // each launch runs for milliseconds: the two overlap
  unsigned* dbad; float *vsink, *asink, *src;
  constexpr int NCASE = 8;
  CHECK(hipMalloc(&dbad, NCASE * 16)); CHECK(hipMalloc(&vsink, vgrid * 64 * 4)); CHECK(hipMalloc(&asink, agrid * 256 * 4));
  CHECK(hipMalloc(&src, 512 * 4));
  float h[512];
  srand(1);
  for (auto& v : h) v = (rand() % 201 - 100) / 64.0f;
  CHECK(hipMemcpy(src, h, sizeof(h), hipMemcpyHostToDevice));
  CHECK(hipMemset(dbad, 0, NCASE * 16));
  hipStream_t s1, s2;
  CHECK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
  CHECK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  const char* names[NCASE] = {
      "victim op_sel:[0,1]      alone",
      "victim op_sel:[0,1]    + bf16 MFMA stream",
      "victim op_sel:[0,1]    + f16 MFMA stream",
      "victim op_sel:[0,1]    + fp32 MFMA stream",
      "victim op_sel:[0,1]    + VALU-only stream",
      "victim op_sel:[1,0]    + bf16 MFMA stream",
      "victim (no op_sel)     + bf16 MFMA stream",
      "victim fma [0,1,0]     + bf16 MFMA stream",
  };
  for (int r = 0; r < rounds; ++r) {
    for (int c = 0; c < NCASE; ++c) {
      // aggressor first, so its work-groups are resident when the victim's waves arrive and while they run
      if (c == 1 || c >= 5) hipLaunchKernelGGL(aggressor<0>, dim3(agrid), dim3(256), 0, s2, src, asink, aiters);
      if (c == 2) hipLaunchKernelGGL(aggressor<1>, dim3(agrid), dim3(256), 0, s2, src, asink, aiters);
      if (c == 3) hipLaunchKernelGGL(aggressor<2>, dim3(agrid), dim3(256), 0, s2, src, asink, aiters);
      if (c == 4) hipLaunchKernelGGL(aggressor<3>, dim3(agrid), dim3(256), 0, s2, src, asink, aiters);
      if (c <= 4) hipLaunchKernelGGL(victim<0>, dim3(vgrid), dim3(64), 0, s1, dbad + 4 * c, vsink, viters);
      if (c == 5) hipLaunchKernelGGL(victim<1>, dim3(vgrid), dim3(64), 0, s1, dbad + 4 * c, vsink, viters);
      if (c == 6) hipLaunchKernelGGL(victim<2>, dim3(vgrid), dim3(64), 0, s1, dbad + 4 * c, vsink, viters);
      if (c == 7) hipLaunchKernelGGL(victim<3>, dim3(vgrid), dim3(64), 0, s1, dbad + 4 * c, vsink, viters);
      CHECK(hipStreamSynchronize(s1));
      CHECK(hipStreamSynchronize(s2));
    }
  }
  unsigned hb[NCASE * 4];
  CHECK(hipMemcpy(hb, dbad, sizeof(hb), hipMemcpyDeviceToHost));
  printf("wrong packed results in the VALU-only victim kernel by lane quarter, of %.3g per quarter and row\n",
         static_cast<double>(rounds) * vgrid * 16 * viters);
  printf("  %-44s %12s %12s %12s %12s\n", "", "lanes 0-15", "16-31", "32-47", "48-63");
  for (int c = 0; c < NCASE; ++c) printf("  %-44s %12u %12u %12u %12u\n", names[c], hb[4 * c], hb[4 * c + 1], hb[4 * c + 2], hb[4 * c + 3]);
  return 0;
}
