// so3_math.h - 3x3 rotation helpers (row-major float[9]) in the reference's formulation.
// Operation order mirrors the reference expressions so fp32 results agree to rounding:
//   log  : so3.py:146-162     exp : so3.py:219-237     hat/vee : so3.py:165-204
// Singular inputs (theta = 0, |v| = 0) give NaN exactly as the reference does.
#pragma once
#include <hip/hip_runtime.h>

namespace diffab {

__device__ inline void mat3_mul(const float* a, const float* b, float* c) {
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) c[i * 3 + j] = a[i * 3 + 0] * b[0 * 3 + j] + a[i * 3 + 1] * b[1 * 3 + j] + a[i * 3 + 2] * b[2 * 3 + j];
}

// S = theta / (2 sin theta) * (R - R^T), theta = acos((tr R - 1) / 2)
__device__ inline void so3_log(const float* R, float* S) {
  const float tr = (R[0] + R[4]) + R[8];
  const float theta = acosf((tr - 1.0f) / 2.0f);
  const float f = theta / (2.0f * sinf(theta));
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) S[i * 3 + j] = f * (R[i * 3 + j] - R[j * 3 + i]);
}

// R = I + S sin(n)/n + S^2 (1 - cos n)/n^2,  n = |(S21, S02, S10)|
__device__ inline void so3_exp(const float* S, float* R) {
  const float vx = S[7], vy = S[2], vz = S[3];
  const float n = sqrtf(vx * vx + vy * vy + vz * vz);
  float sn, cn;
  sincosf(n, &sn, &cn);
  float S2[9];
  mat3_mul(S, S, S2);
  const float n2 = n * n;
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    const float eye = (i == 0 || i == 4 || i == 8) ? 1.0f : 0.0f;
    R[i] = (eye + S[i] * sn / n) + S2[i] * (1.0f - cn) / n2;
  }
}

__device__ inline void so3_hat(float x, float y, float z, float* S) {
  S[0] = 0.0f; S[1] = -z;   S[2] = y;
  S[3] = z;    S[4] = 0.0f; S[5] = -x;
  S[6] = -y;   S[7] = x;    S[8] = 0.0f;
}

__device__ inline void so3_rotvec_to_matrix(float x, float y, float z, float* R) {
  float S[9];
  so3_hat(x, y, z, S);
  so3_exp(S, R);
}

// exp(k log R)
__device__ inline void so3_scale(const float* R, float k, float* out) {
  float S[9];
  so3_log(R, S);
#pragma unroll
  for (int i = 0; i < 9; ++i) S[i] *= k;
  so3_exp(S, out);
}

}  // namespace diffab
