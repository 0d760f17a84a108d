// Standalone probe for the two-queue wrong result of profiles/r03_lockstep.md section 4 / profiles/r04_two_queue.md.
// Stream 1 repeats [producer: v[r][0..2] = f(r, it) with per-lane dword stores | consumer: out[r] = v[r] loaded in one of several
// ways | checker: counts rows where out != f(r, it), classifies them (== f(r, it - 1): stale; lane histogram)].
// Stream 2 meanwhile keeps a second hardware queue busy (a 256 MiB copy kernel, or many tiny kernels).
//   mode 0: plain C loads (hipcc merges the three floats into one global_load_dwordx3 at 12-byte lane stride, 4-byte aligned)
//   mode 1: three volatile dword loads (global_load_dword sc0 sc1)
//   mode 2: three plain global_load_dword (inline asm, no cache-policy bits)
//   mode 3: one global_load_dwordx3 sc0 sc1 (inline asm)
//   mode 4: one global_load_dwordx3 (inline asm, no bits)
// build: hipcc -O3 --offload-arch=gfx950 tools/two_queue_load_probe.hip -o tools/bin/two_queue_load_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ float val(int r, int c, int it) { return __uint_as_float(0x3f800000u | ((unsigned)(r * 3 + c) * 2654435761u + (unsigned)it * 40503u) >> 9); }

// like the 3-wide epilogue of the MLP chain: lanes (row pair, column) store single dwords
__global__ void producer(float* __restrict__ v, int rows, int it) {
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  const int r = gid / 3, c = gid % 3;
  if (r < rows) v[r * 3 + c] = val(r, c, it);
}

struct f3 { float x, y, z; };
template <int MODE>
__global__ void consumer(const float* __restrict__ v, float* __restrict__ out, int rows) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= rows) return;
  float x, y, z;
  const float* p = v + (size_t)r * 3;
  if (MODE == 0) { x = p[0]; y = p[1]; z = p[2]; }
  else if (MODE == 1) { const volatile float* q = p; x = q[0]; y = q[1]; z = q[2]; }
  else if (MODE == 2) {
    asm volatile("global_load_dword %0, %3, off\n global_load_dword %1, %3, off offset:4\n global_load_dword %2, %3, off offset:8\n s_waitcnt vmcnt(0)"
                 : "=&v"(x), "=&v"(y), "=&v"(z) : "v"(p) : "memory");
  } else if (MODE == 3) {
    typedef float f32x3 __attribute__((ext_vector_type(3)));
    f32x3 t;
    asm volatile("global_load_dwordx3 %0, %1, off sc0 sc1\n s_waitcnt vmcnt(0)" : "=&v"(t) : "v"(p) : "memory");
    x = t.x; y = t.y; z = t.z;
  } else {
    typedef float f32x3 __attribute__((ext_vector_type(3)));
    f32x3 t;
    asm volatile("global_load_dwordx3 %0, %1, off\n s_waitcnt vmcnt(0)" : "=&v"(t) : "v"(p) : "memory");
    x = t.x; y = t.y; z = t.z;
  }
  // some arithmetic with divisions and a sincos, like heads_finish_kernel (keeps the kernel's shape; the result is not checked)
  const float n = sqrtf(x * x + y * y + z * z);
  float sn, cn;
  sincosf(n, &sn, &cn);
  out[(size_t)rows * 3 + r] = sn / n + (1.0f - cn) / (n * n);
  out[r * 3 + 0] = x; out[r * 3 + 1] = y; out[r * 3 + 2] = z;
}

// 9-float rows (36-byte lane stride), constant data: mode 0 plain C (hipcc merges into two 4-byte-aligned dwordx4 + one dword), 1 dword loads
template <int MODE>
__global__ void consumer9(const float* __restrict__ O, float* __restrict__ out, int rows) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= rows) return;
  float o[9];
  if (MODE == 0) {
#pragma unroll
    for (int k = 0; k < 9; ++k) o[k] = O[(size_t)r * 9 + k];
  } else {
#pragma unroll
    for (int k = 0; k < 9; ++k) o[k] = __hip_atomic_load(O + (size_t)r * 9 + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SINGLETHREAD);
  }
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < 9; ++k) s += o[k] * (float)(k + 1);
  out[r] = s;
}
__global__ void fill9(float* O, int rows) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g < rows * 9) O[g] = (float)((g * 2654435761u) >> 20);
}
__global__ void checker9(const float* __restrict__ O, const float* __restrict__ out, int rows, unsigned* __restrict__ cnt) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= rows) return;
  float s = 0.f;
  for (int k = 0; k < 9; ++k) s += __hip_atomic_load(O + (size_t)r * 9 + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SINGLETHREAD) * (float)(k + 1);
  if (s != out[r]) { atomicAdd(cnt + 0, 1u); atomicAdd(cnt + 2 + ((r & 63) >> 4), 1u); }
}

__global__ void checker(const float* __restrict__ out, int rows, int it, unsigned* __restrict__ cnt) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= rows) return;
  bool bad = false, stale = true;
  for (int c = 0; c < 3; ++c) {
    const float o = out[r * 3 + c];
    if (o != val(r, c, it)) bad = true;
    if (o != val(r, c, it - 1)) stale = false;
  }
  if (bad) {
    atomicAdd(cnt + 0, 1u);
    if (stale) atomicAdd(cnt + 1, 1u);
    atomicAdd(cnt + 2 + ((r & 63) >> 4), 1u);  // quarter of the consumer's wave
  }
}

__global__ void noise_copy(const float4* __restrict__ a, float4* __restrict__ b, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}
__global__ void noise_tiny(float* p) { if (threadIdx.x == 0) p[blockIdx.x] += 1.0f; }

template <int MODE>
static void run(const char* name, int rows, int iters, int noise, float* v, float* out, unsigned* cnt, float4* na, float4* nb, size_t nn, float* tiny,
                hipStream_t s1, hipStream_t s2) {
  CK(hipMemset(cnt, 0, 8 * sizeof(unsigned)));
  CK(hipDeviceSynchronize());
  for (int it = 1; it <= iters; ++it) {
    if (noise == 1) hipLaunchKernelGGL(noise_copy, dim3(2048), dim3(256), 0, s2, na, nb, nn);
    if (noise == 2) for (int k = 0; k < 8; ++k) hipLaunchKernelGGL(noise_tiny, dim3(256), dim3(64), 0, s2, tiny);
    hipLaunchKernelGGL(producer, dim3((rows * 3 + 127) / 128), dim3(128), 0, s1, v, rows, it);
    hipLaunchKernelGGL(noise_tiny, dim3(256), dim3(64), 0, s1, tiny + 256);  // an unrelated launch between producer and consumer, as in the step
    hipLaunchKernelGGL(consumer<MODE>, dim3((rows + 127) / 128), dim3(128), 0, s1, v, out, rows);
    hipLaunchKernelGGL(checker, dim3((rows + 127) / 128), dim3(128), 0, s1, out, rows, it, cnt);
    if ((it & 63) == 0) CK(hipDeviceSynchronize());  // bound the queue depth
  }
  CK(hipDeviceSynchronize());
  unsigned h[8];
  CK(hipMemcpy(h, cnt, sizeof(h), hipMemcpyDeviceToHost));
  printf("noise %d  %-34s: %u wrong rows of %lld (stale: %u; by wave quarter: %u %u %u %u)\n", noise, name, h[0], (long long)rows * iters, h[1], h[2], h[3],
         h[4], h[5]);
}

template <int MODE>
static void run9(const char* name, int rows, int iters, int noise, float* O, float* out, unsigned* cnt, float4* na, float4* nb, size_t nn, float* tiny,
                 hipStream_t s1, hipStream_t s2) {
  CK(hipMemset(cnt, 0, 8 * sizeof(unsigned)));
  CK(hipDeviceSynchronize());
  for (int it = 1; it <= iters; ++it) {
    if (noise == 1) hipLaunchKernelGGL(noise_copy, dim3(2048), dim3(256), 0, s2, na, nb, nn);
    if (noise == 2) for (int k = 0; k < 8; ++k) hipLaunchKernelGGL(noise_tiny, dim3(256), dim3(64), 0, s2, tiny);
    hipLaunchKernelGGL(consumer9<MODE>, dim3((rows + 127) / 128), dim3(128), 0, s1, O, out, rows);
    hipLaunchKernelGGL(checker9, dim3((rows + 127) / 128), dim3(128), 0, s1, O, out, rows, cnt);
    if ((it & 63) == 0) CK(hipDeviceSynchronize());
  }
  CK(hipDeviceSynchronize());
  unsigned h[8];
  CK(hipMemcpy(h, cnt, sizeof(h), hipMemcpyDeviceToHost));
  printf("noise %d  %-34s: %u wrong rows of %lld (by wave quarter: %u %u %u %u)\n", noise, name, h[0], (long long)rows * iters, h[2], h[3], h[4], h[5]);
}

int main(int argc, char** argv) {
  const int rows = argc > 1 ? atoi(argv[1]) : 32768, iters = argc > 2 ? atoi(argv[2]) : 3000;
  float *v, *out, *tiny;
  unsigned* cnt;
  float4 *na, *nb;
  const size_t nn = (256u << 20) / 16;
  CK(hipMalloc(&v, (size_t)rows * 3 * 4));
  CK(hipMalloc(&out, (size_t)rows * 4 * 4));
  CK(hipMalloc(&tiny, 4096));
  CK(hipMalloc(&cnt, 64));
  CK(hipMalloc(&na, nn * 16));
  CK(hipMalloc(&nb, nn * 16));
  CK(hipMemset(na, 1, nn * 16));
  CK(hipMemset(tiny, 0, 4096));
  hipStream_t s1, s2;
  CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  for (int noise = 0; noise < 3; ++noise) {
    run<0>("plain C (dwordx3, merged)", rows, iters, noise, v, out, cnt, na, nb, nn, tiny, s1, s2);
    run<1>("volatile dword x 3 (sc0 sc1)", rows, iters, noise, v, out, cnt, na, nb, nn, tiny, s1, s2);
    run<2>("asm dword x 3", rows, iters, noise, v, out, cnt, na, nb, nn, tiny, s1, s2);
    run<3>("asm dwordx3 sc0 sc1", rows, iters, noise, v, out, cnt, na, nb, nn, tiny, s1, s2);
    run<4>("asm dwordx3", rows, iters, noise, v, out, cnt, na, nb, nn, tiny, s1, s2);
  }
  float* O9;
  CK(hipMalloc(&O9, (size_t)rows * 9 * 4));
  hipLaunchKernelGGL(fill9, dim3((rows * 9 + 255) / 256), dim3(256), 0, 0, O9, rows);
  CK(hipDeviceSynchronize());
  for (int noise = 0; noise < 3; ++noise) {
    run9<0>("9-float rows, plain C (dwordx4 x2)", rows, iters, noise, O9, out, cnt, na, nb, nn, tiny, s1, s2);
    run9<1>("9-float rows, dword loads", rows, iters, noise, O9, out, cnt, na, nb, nn, tiny, s1, s2);
  }
  return 0;
}
