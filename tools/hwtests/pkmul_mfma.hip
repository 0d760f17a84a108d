// Hardware question (gfx950, ROCm 7.2 hipcc): a packed-fp32 VALU op whose LOW result reads the HIGH register of a source pair
// (op_sel bit set) issued right in front of an MFMA - is the packed result always right?
// Found in round 5 (ipa_persistent.hip): v_pk_mul_f32 ... op_sel:[0,1] followed by v_mfma_f32_16x16x32_f16 gave 0 as the low product in
// lanes 48-63, now and then; hipcc inserts no wait states for the pair.  This program measures how often, for which forms, and at what
// distance.  usage: pkmul_mfma [launches]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define MF16 "v_mfma_f32_16x16x32_f16 %7, %5, %6, %7\n"
#define PRE "v_mfma_f32_16x16x32_f16 %1, %5, %6, %1\n v_mfma_f32_16x16x32_f16 %2, %5, %6, %2\n"
#define CASES(X)                                                                                   \
  X(0, "v_pk_mul_f32 %0, %3, %4 op_sel:[0,1]\n" MF16, 0, 1, 0, 0, "pk_mul op_sel:[0,1] | mfma16")                                  \
  X(1, "v_pk_mul_f32 %0, %3, %4 op_sel:[0,1]\n s_nop 0\n" MF16, 0, 1, 0, 0, "pk_mul op_sel:[0,1] | s_nop 0 | mfma16")              \
  X(2, "v_pk_mul_f32 %0, %3, %4 op_sel:[0,1]\n s_nop 1\n" MF16, 0, 1, 0, 0, "pk_mul op_sel:[0,1] | s_nop 1 | mfma16")              \
  X(3, "v_pk_mul_f32 %0, %3, %4 op_sel:[0,1]\n s_nop 2\n" MF16, 0, 1, 0, 0, "pk_mul op_sel:[0,1] | s_nop 2 | mfma16")              \
  X(4, "v_pk_mul_f32 %0, %3, %4 op_sel:[0,1]\n s_nop 3\n" MF16, 0, 1, 0, 0, "pk_mul op_sel:[0,1] | s_nop 3 | mfma16")              \
  X(5, "v_pk_mul_f32 %0, %3, %4 op_sel:[0,1]\n s_nop 5\n" MF16, 0, 1, 0, 0, "pk_mul op_sel:[0,1] | s_nop 5 | mfma16")              \
  X(6, "v_pk_mul_f32 %0, %3, %4 op_sel:[0,1]\n v_nop\n v_nop\n" MF16, 0, 1, 0, 0, "pk_mul op_sel:[0,1] | 2 v_nop | mfma16")        \
  X(7, "v_pk_mul_f32 %0, %3, %4 op_sel:[1,0]\n" MF16, 1, 0, 0, 0, "pk_mul op_sel:[1,0] | mfma16")                                  \
  X(8, "v_pk_mul_f32 %0, %3, %4 op_sel:[1,1]\n" MF16, 1, 1, 0, 0, "pk_mul op_sel:[1,1] | mfma16")                                  \
  X(9, "v_pk_mul_f32 %0, %3, %4 op_sel_hi:[1,0]\n" MF16, 0, 0, 0, 0, "pk_mul op_sel_hi:[1,0] | mfma16")                            \
  X(10, "v_pk_mul_f32 %0, %3, %4\n" MF16, 0, 0, 0, 0, "pk_mul (no op_sel) | mfma16")                                               \
  X(11, "v_pk_add_f32 %0, %3, %4 op_sel:[0,1]\n" MF16, 0, 1, 0, 1, "pk_add op_sel:[0,1] | mfma16")                                 \
  X(12, "v_pk_fma_f32 %0, %3, %4, %3 op_sel:[0,1,0]\n" MF16, 0, 1, 0, 2, "pk_fma op_sel:[0,1,0] | mfma16")                         \
  X(13, "v_pk_mul_f32 %0, %3, %4 op_sel:[0,1]\n v_mfma_f32_32x32x16_f16 %8, %5, %6, %8\n", 0, 1, 0, 0, "pk_mul op_sel:[0,1] | mfma32") \
  X(14, "v_pk_mul_f32 %0, %3, %4 op_sel:[0,1]\n v_mfma_f32_16x16x4_f32 %7, %9, %9, %7\n", 0, 1, 0, 0, "pk_mul op_sel:[0,1] | mfma f32 16x16x4") \
  X(15, "v_pk_mul_f32 %0, %3, %4 op_sel:[0,1]\n v_mul_f32 %9, %9, %9\n", 0, 1, 0, 0, "pk_mul op_sel:[0,1] | v_mul (no mfma behind)") \
  X(16, "v_pk_fma_f32 %0, %3, %4, %3 op_sel:[0,0,1]\n" MF16, 0, 0, 1, 2, "pk_fma op_sel:[0,0,1] | mfma16")                      \
  X(17, "v_pk_fma_f32 %0, %3, %4, %3 op_sel:[1,0,0]\n" MF16, 1, 0, 0, 2, "pk_fma op_sel:[1,0,0] | mfma16")                      \
  X(18, "v_pk_fma_f32 %0, %3, %4, %3 op_sel:[0,1,1]\n" MF16, 0, 1, 1, 2, "pk_fma op_sel:[0,1,1] | mfma16")                      \
  X(19, "v_pk_fma_f32 %0, %3, %4, %3 op_sel:[1,1,0]\n" MF16, 1, 1, 0, 2, "pk_fma op_sel:[1,1,0] | mfma16")                      \
  X(20, "v_pk_fma_f32 %0, %3, %4, %3 op_sel:[1,0,1]\n" MF16, 1, 0, 1, 2, "pk_fma op_sel:[1,0,1] | mfma16")                      \
  X(21, "v_pk_mul_f32 %0, %3, %4 op_sel:[0,1] op_sel_hi:[0,1]\n" MF16, 0, 1, 0, 3, "pk_mul op_sel:[0,1] op_sel_hi:[0,1] | mfma16") \
  X(22, "v_pk_mul_f32 %0, %3, %4 op_sel:[0,1]\n v_mfma_f32_16x16x32_bf16 %7, %5, %6, %7\n", 0, 1, 0, 0, "pk_mul op_sel:[0,1] | mfma bf16 16x16x32") \
  X(23, "v_pk_mov_b32 %0, %3, %4 op_sel:[1,0]\n" MF16, 1, 0, 0, 4, "pk_mov op_sel:[1,0] (the form hipcc emits) | mfma16")           \
  X(24, "v_pk_mov_b32 %0, %3, %4 op_sel:[0,1]\n" MF16, 0, 1, 0, 4, "pk_mov op_sel:[0,1] | mfma16")

template <int MODE>
__global__ __launch_bounds__(512) void k(const f16x8* __restrict__ A, unsigned* __restrict__ bad, float* __restrict__ sink, int iters) {
  const int lane = threadIdx.x & 63;
  f16x8 a = A[lane], b = A[64 + lane];
  f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0, acc2 = acc0;
  f32x16 big;
  for (int i = 0; i < 16; ++i) big[i] = 0.f;
  f32x2 s0 = {1.0f / 2048.0f + lane * 1e-6f, 3.0f}, s1 = {5.0f, 1.0f / 256.0f};
  float one = 1.0f;
  unsigned nbad = 0;
  for (int it = 0; it < iters; ++it) {
    f32x2 d;
    s0[0] += 1e-7f;
    int h0 = 0, h1 = 0, h2 = 0, kind = 0;
    // operand numbering of the texts above: %0 d, %1 acc0, %2 acc1, %3 s0, %4 s1, %5 a, %6 b, %7 acc2, %8 big, %9 one
#define X(M, TXT, H0, H1, H2, KIND, NAME)                                                                                                     \
  if (MODE == M) {                                                                                                                  \
    asm volatile(PRE TXT "s_nop 7\n"                                                                                              \
                 : "=&v"(d), "+v"(acc0), "+v"(acc1)                                                                                 \
                 : "v"(s0), "v"(s1), "v"(a), "v"(b), "v"(acc2), "v"(big), "v"(one));                                                \
    h0 = H0; h1 = H1; h2 = H2; kind = KIND;                                                                                         \
  }
    CASES(X)
#undef X
    // low result: sources selected by op_sel; high result: the high registers (op_sel_hi default), except the listed forms
    const float x0 = s0[h0], x1 = s1[h1], x2 = s0[h2];
    float want_lo = kind == 1 ? x0 + x1 : kind == 2 ? __builtin_fmaf(x0, x1, x2) : x0 * x1;
    const float y0 = (MODE == 21) ? s0[0] : s0[1], y1 = (MODE == 9) ? s1[0] : s1[1];
    float want_hi = kind == 1 ? y0 + y1 : kind == 2 ? __builtin_fmaf(y0, y1, s0[1]) : y0 * y1;
    if (kind == 4) {  // v_pk_mov_b32: D.lo = S0[op_sel[0]], D.hi = S1[op_sel[1]]
      want_lo = s0[h0];
      want_hi = s1[h1];
    }
    if (d[0] != want_lo || d[1] != want_hi) ++nbad;
  }
  if (nbad) atomicAdd(bad + (lane >> 4), nbad);
  sink[blockIdx.x * 512 + threadIdx.x] = acc0[0] + acc1[1] + acc2[2] + big[3] + one;
}

int main(int argc, char** argv) {
  const int launches = argc > 1 ? atoi(argv[1]) : 4, iters = 1000, grid = 512, NC = 25;
  std::vector<_Float16> hA(128 * 8);
  srand(1);
  for (auto& v : hA) v = static_cast<_Float16>((rand() % 201 - 100) / 64.0f);
  f16x8* dA; unsigned* dbad; float* sink;
  hipMalloc(&dA, hA.size() * 2); hipMalloc(&dbad, NC * 16); hipMalloc(&sink, grid * 512 * 4);
  hipMemcpy(dA, hA.data(), hA.size() * 2, hipMemcpyHostToDevice);
  hipMemset(dbad, 0, NC * 16);
  const char* names[NC];
#define X(M, TXT, H0, H1, H2, KIND, NAME) names[M] = NAME;
  CASES(X)
#undef X
  for (int l = 0; l < launches; ++l) {
#define X(M, TXT, H0, H1, H2, KIND, NAME) hipLaunchKernelGGL(k<M>, dim3(grid), dim3(512), 0, 0, dA, dbad + 4 * M, sink, iters);
    CASES(X)
#undef X
  }
  if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed: %s\n", hipGetErrorString(hipGetLastError())); return 2; }
  std::vector<unsigned> h(NC * 4);
  hipMemcpy(h.data(), dbad, NC * 16, hipMemcpyDeviceToHost);
  printf("wrong packed results by lane quarter, of %ld per quarter\n", static_cast<long>(launches) * grid * 8 * 16 * iters);
  for (int m = 0; m < NC; ++m) printf("  %-52s %10u %10u %10u %10u\n", names[m], h[4 * m], h[4 * m + 1], h[4 * m + 2], h[4 * m + 3]);
  return 0;
}
