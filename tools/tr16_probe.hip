#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void k(const unsigned short* in, unsigned short* out, int stride_elems) {
  __shared__ __attribute__((aligned(16))) unsigned short lds[64 * 64];
  for (int i = threadIdx.x; i < 64 * 64; i += 64) lds[i] = in[i];
  __syncthreads();
  const int l = threadIdx.x, grp = l >> 4, li = l & 15, q = li >> 2, p = li & 3;
  // group grp reads block rows 4*grp .. 4*grp+3, columns 0..15: lane 4q+p supplies the address of row q, columns 4p..4p+3
  const unsigned short* a = lds + (4 * grp + q) * stride_elems + 4 * p;
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)a);
  for (int e = 0; e < 4; ++e) out[l * 4 + e] = (unsigned short)v[e];
}
int main() {
  unsigned short h[4096], o[256];
  for (int r = 0; r < 64; ++r) for (int c = 0; c < 64; ++c) h[r * 64 + c] = r * 100 + c;
  unsigned short *di, *dout;
  hipMalloc(&di, sizeof(h)); hipMalloc(&dout, sizeof(o));
  hipMemcpy(di, h, sizeof(h), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, di, dout, 64);
  hipMemcpy(o, dout, sizeof(o), hipMemcpyDeviceToHost);
  for (int l = 0; l < 64; l += 1) printf("lane %2d: %4d %4d %4d %4d\n", l, o[l*4], o[l*4+1], o[l*4+2], o[l*4+3]);
  return 0;
}
