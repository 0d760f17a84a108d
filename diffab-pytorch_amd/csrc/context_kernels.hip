// context_kernels.hip - encode_context: ResidueEmbedding (reference diffab_pytorch.py:57-183) and PairEmbedding
// (:186-312) forward.  Runs once per sample (not per denoise step); produces the (B,K,D) / (B,K,K,C) context embeddings the hot
// path consumes.  Feature rows are assembled by small gather kernels and pushed through the MFMA linear kernel; the pair part
// is processed a few patches at a time so its row buffers (K^2 rows x 225 floats per patch) stay bounded.
// Reference quirks reproduced on purpose: same_chain_mask is the PRODUCT of chain ids (:279); the structure-context mask is
// applied to `distmat` only after its use, so neither the distance nor the dihedral pair feature is masked (:292-301);
// non-context residues take amino-acid type UNK = 20 (:115, :273); nn.Embedding padding_idx only affects gradients.
#include "common.h"
#include "denoiser_internal.h"

namespace diffab {

constexpr int kAA = 21, kUNK = 20, kCA = 1;

__device__ inline float softplus_f(float x) { return x > 20.0f ? x : log1pf(expf(x)); }  // F.softplus defaults (beta 1, threshold 20)

// AngularEncoding (:20-54): [x, sin(f x) for f in bands, cos(f x) for f in bands], bands = [1..n, 1/1..1/n]
__device__ inline void angular_encode(float x, int nf, float* out) {
  out[0] = x;
  for (int k = 0; k < 2 * nf; ++k) {
    const float f = k < nf ? static_cast<float>(k + 1) : 1.0f / static_cast<float>(k - nf + 1);
    out[1 + k] = sinf(f * x);
    out[1 + 2 * nf + k] = cosf(f * x);
  }
}

// one block per residue: [aa_emb (D) | coord one-hot-by-type (21*A*3) | dihedral enc (3*13) | chain_emb (D)]
__global__ void residue_feat_kernel(const int64_t* __restrict__ seq, const float* __restrict__ xyz, const float* __restrict__ O,
                                    const float* __restrict__ dih, const int64_t* __restrict__ chain, const float* __restrict__ amask,
                                    const uint8_t* __restrict__ struct_m, const uint8_t* __restrict__ seq_m, const float* __restrict__ aa_emb,
                                    const float* __restrict__ chain_emb, int K, int A, int D, float* __restrict__ out) {
  const int64_t r = blockIdx.x;
  const int l = static_cast<int>(r % K);
  const int64_t b0 = r - l;
  const int Din = D + kAA * A * 3 + 39 + D;
  float* o = out + r * Din;
  int64_t s = seq[r];
  if (seq_m && !seq_m[r]) s = kUNK;
  const bool sm = struct_m ? struct_m[r] != 0 : true;
  const bool dm = struct_m ? (struct_m[r] != 0 && struct_m[b0 + (l + 1) % K] != 0) : true;  // roll(mask,-1) & mask (:160-166)
  for (int d = threadIdx.x; d < D; d += blockDim.x) {
    o[d] = aa_emb[s * D + d];
    o[D + kAA * A * 3 + 39 + d] = chain_emb[chain[r] * D + d];
  }
  for (int idx = threadIdx.x; idx < kAA * A * 3; idx += blockDim.x) {
    const int t = idx / (A * 3), a = (idx / 3) % A, i = idx % 3;
    float v = 0.0f;
    if (t == s && sm) {  // local = O^T (x - x_CA), times the atom mask (:119-127)
      const float* x = xyz + (r * A + a) * 3;
      const float* ca = xyz + (r * A + kCA) * 3;
      const float* Or = O + r * 9;
      v = (Or[0 * 3 + i] * (x[0] - ca[0]) + Or[1 * 3 + i] * (x[1] - ca[1]) + Or[2 * 3 + i] * (x[2] - ca[2])) * amask[r * A + a];
    }
    o[D + idx] = v;
  }
  if (threadIdx.x < 3) {
    float enc[13];
    angular_encode(dih[r * 3 + threadIdx.x], 3, enc);
    for (int k = 0; k < 13; ++k) o[D + kAA * A * 3 + threadIdx.x * 13 + k] = dm ? enc[k] : 0.0f;
  }
}

// one thread per (pair row, atom pair): exp(-softplus(coef[s_i*21+s_j]) d^2) * atom_mask_i * atom_mask_j   (:288-295)
// softplus of the whole pair2distcoef table (441 x A*A) once per call: it depends on the residue-type pair only, not on the
// residue pair, and the two transcendentals per element were most of pair_dist_kernel's time
__global__ void softplus_table_kernel(const float* __restrict__ coefw, int n, float* __restrict__ out) {
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  if (gid < n) out[gid] = softplus_f(coefw[gid]);
}

int launch_softplus_table(const float* coefw, int n, float* out, hipStream_t st) {
  hipLaunchKernelGGL(softplus_table_kernel, dim3((n + 255) / 256), dim3(256), 0, st, coefw, n, out);
  DIFFAB_LAUNCH_CHECK();
  return DIFFAB_OK;
}

// one block per pair row (b, i, j), one thread per atom pair: exp(-softplus(coef[s_i*21+s_j]) d^2) * atom_mask_i * atom_mask_j (:288-295).
// distmat == nullptr: the distance is taken from the coordinates, d = |xyz[b,i,a1] - xyz[b,j,a2]| (what the reference's data layer
// computes with protstruc and then leaves out of its batches, data.py:76 / preprocess_pdb.py:61): the 14.7 MB/patch distance
// tensor is then never materialised.  out rows are ldo = round_up(A*A, 4) floats apart, pad columns zero (vector path of the GEMM).
__global__ void pair_dist_kernel(const int64_t* __restrict__ seq, const uint8_t* __restrict__ seq_m, const float* __restrict__ distmat,
                                 const float* __restrict__ xyz, const float* __restrict__ amask, const float* __restrict__ coef_sp, int K,
                                 int A, int64_t row0, int64_t nrows, float* __restrict__ out, int ldo) {
  const int AA2 = A * A;
  constexpr int RPB = 8;  // pair rows per block: one row per block is bound by the work-group dispatch rate (174 k tiny groups)
  for (int rr = 0; rr < RPB; ++rr) {
    const int64_t lr = static_cast<int64_t>(blockIdx.x) * RPB + rr;
    if (lr >= nrows) return;
    const int64_t row = row0 + lr;  // global pair row (b, i, j)
    const int64_t b = row / (static_cast<int64_t>(K) * K);
    const int i = static_cast<int>((row / K) % K), j = static_cast<int>(row % K);
    const int64_t ri = b * K + i, rj = b * K + j;
    const int64_t si = (seq_m && !seq_m[ri]) ? kUNK : seq[ri], sj = (seq_m && !seq_m[rj]) ? kUNK : seq[rj];
    const float* crow = coef_sp + (si * kAA + sj) * AA2;
    for (int p = threadIdx.x; p < ldo; p += blockDim.x) {
      float v = 0.0f;
      if (p < AA2) {
        const int a1 = p / A, a2 = p % A;
        float d;
        if (distmat) {
          d = distmat[row * AA2 + p];
        } else {
          const float* pa = xyz + (ri * A + a1) * 3;
          const float* pb = xyz + (rj * A + a2) * 3;
          const float dx = pa[0] - pb[0], dy = pa[1] - pb[1], dz = pa[2] - pb[2];
          d = sqrtf((dx * dx + dy * dy) + dz * dz);
        }
        v = expf(-1.0f * crow[p] * (d * d)) * (amask[ri * A + a1] * amask[rj * A + a2]);
      }
      out[lr * ldo + p] = v;
    }
  }
}

// The same rows per (b, i) group of K pair rows (the fused backward's one remaining use of the materialised features): the group's
// coordinates, masks and residue types are staged in LDS once, the (row, p) elements are walked in memory order (coalesced 4-byte
// stores), no 64-bit division, no dependent index load per row - 482 us per 393 k rows for the kernel above was 0.7 TB/s.
__global__ __launch_bounds__(256) void pair_dist_group_kernel(const int64_t* __restrict__ seq, const uint8_t* __restrict__ seq_m,
                                                              const float* __restrict__ distmat, const float* __restrict__ xyz,
                                                              const float* __restrict__ amask, const float* __restrict__ coef_sp, int K, int A,
                                                              int64_t row0, float* __restrict__ out, int ldo) {
  extern __shared__ float gl[];  // xj [K][A 3] | mj [K][A] | xi [A 3] | mi [A] | sj [K] (int)
  const int AA2 = A * A;
  float* xj = gl;
  float* mj = xj + K * A * 3;
  float* xi = mj + K * A;
  float* mi = xi + A * 3;
  int* sjs = reinterpret_cast<int*>(mi + A);
  const int64_t lrow0 = static_cast<int64_t>(blockIdx.x) * K;  // first row of the group inside this launch
  const int64_t grow0 = row0 + lrow0;
  const int64_t bi = grow0 / K, b = bi / K;
  const int64_t ri = bi;
  const int si = static_cast<int>((seq_m && !seq_m[ri]) ? kUNK : seq[ri]);
  for (int idx = threadIdx.x; idx < K * A; idx += blockDim.x) {
    const int64_t rj = b * K + idx / A;
    mj[idx] = amask[rj * A + idx % A];
    if (xyz) {
      const float* p = xyz + (rj * A + idx % A) * 3;
      xj[idx * 3] = p[0]; xj[idx * 3 + 1] = p[1]; xj[idx * 3 + 2] = p[2];
    }
  }
  for (int idx = threadIdx.x; idx < A; idx += blockDim.x) {
    mi[idx] = amask[ri * A + idx];
    if (xyz) {
      const float* p = xyz + (ri * A + idx) * 3;
      xi[idx * 3] = p[0]; xi[idx * 3 + 1] = p[1]; xi[idx * 3 + 2] = p[2];
    }
  }
  for (int j = threadIdx.x; j < K; j += blockDim.x) {
    const int64_t rj = b * K + j;
    sjs[j] = static_cast<int>((seq_m && !seq_m[rj]) ? kUNK : seq[rj]);
  }
  __syncthreads();
  // thread = atom pair p (its a1, a2, the atom of i and its mask in registers), rows j in memory order: coalesced 4-byte stores, no
  // division and no index arithmetic per element (walking (row, p) as one flat index cost two 32-bit divisions per element: the kernel
  // was bound by its ~90 vector instructions per element, 270 us per chunk)
  for (unsigned p = threadIdx.x; p < static_cast<unsigned>(ldo); p += blockDim.x) {
    const bool valid = p < static_cast<unsigned>(AA2);
    const unsigned a1 = valid ? p / static_cast<unsigned>(A) : 0u, a2 = valid ? p - a1 * A : 0u;
    const float pa0 = xi[3 * a1], pa1 = xi[3 * a1 + 1], pa2 = xi[3 * a1 + 2], m1 = mi[a1];
    const float* crow = coef_sp + static_cast<size_t>(si * kAA) * AA2 + p;
#pragma unroll 4
    for (int j = 0; j < K; ++j) {
      float v = 0.0f;
      if (valid) {
        float d;
        if (distmat) {
          d = distmat[(grow0 + j) * AA2 + p];
        } else {
          const float* pb = xj + (j * A + a2) * 3;
          const float dx = pa0 - pb[0], dy = pa1 - pb[1], dz = pa2 - pb[2];
          d = sqrtf((dx * dx + dy * dy) + dz * dz);
        }
        v = expf(-1.0f * crow[static_cast<size_t>(sjs[j]) * AA2] * (d * d)) * (m1 * mj[j * A + a2]);
      }
      out[(lrow0 + j) * ldo + p] = v;
    }
  }
}

// one block per pair row: [aa_pair_emb (C) | relpos_emb * chain_i*chain_j (C) | dist_feat (C) | dihedral enc (2*9)]
__global__ void pair_cat_kernel(const int64_t* __restrict__ seq, const uint8_t* __restrict__ seq_m, const int64_t* __restrict__ resid,
                                int resid_bstride, const int64_t* __restrict__ chain, const float* __restrict__ pdih,
                                const float* __restrict__ dist_feat, const float* __restrict__ pair_emb, const float* __restrict__ rel_emb,
                                int K, int C, int max_dist, int64_t row0, float* __restrict__ out, int ldo) {
  const int64_t lr = blockIdx.x, row = row0 + lr;
  const int64_t b = row / (static_cast<int64_t>(K) * K);
  const int i = static_cast<int>((row / K) % K), j = static_cast<int>(row % K);
  const int64_t ri = b * K + i, rj = b * K + j;
  const int64_t si = (seq_m && !seq_m[ri]) ? kUNK : seq[ri], sj = (seq_m && !seq_m[rj]) ? kUNK : seq[rj];
  int64_t rel = resid[b * resid_bstride + i] - resid[b * resid_bstride + j];
  rel = rel < -max_dist ? -max_dist : (rel > max_dist ? max_dist : rel);
  const float same = static_cast<float>(chain[ri] * chain[rj]);  // a product, not an equality test (:279)
  const int W = 3 * C + 18;
  float* o = out + lr * ldo;  // ldo = round_up(W, 4); pad columns zero
  for (int c = W + threadIdx.x; c < ldo; c += blockDim.x) o[c] = 0.0f;
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    o[c] = pair_emb[(si * kAA + sj) * C + c];
    o[C + c] = rel_emb[(rel + max_dist) * C + c] * same;
    o[2 * C + c] = dist_feat[lr * C + c];
  }
  if (threadIdx.x < 2) {
    float enc[9];
    angular_encode(pdih[row * 2 + threadIdx.x], 2, enc);
    for (int k = 0; k < 9; ++k) o[3 * C + threadIdx.x * 9 + k] = enc[k];
  }
}

// out[row][:] *= atom_mask_i[CA] * atom_mask_j[CA]   (:269-271, :312)
__global__ void pair_mask_kernel(float* __restrict__ out, const float* __restrict__ amask, int K, int A, int C, int64_t rows) {
  const int64_t gid = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x;
  if (gid >= rows * C) return;
  const int64_t row = gid / C;
  const int64_t b = row / (static_cast<int64_t>(K) * K);
  const int i = static_cast<int>((row / K) % K), j = static_cast<int>(row % K);
  out[gid] *= amask[(b * K + i) * A + kCA] * amask[(b * K + j) * A + kCA];
}

// ------------------------------------------------------------------ featurisation from coordinates
// What the reference's data layer computes with protstruc before a batch reaches the model (data.py:75-82, preprocess_pdb.py:60-65):
// backbone orientations, backbone dihedrals (+ mask) and the pairwise phi / psi dihedrals.  protstruc is not in the reference tree, so
// these follow the geometric definitions (stated at each kernel) and parity with protstruc is UNPINNED; the oracle restates the same
// definitions in float64.  Atom slots: N = 0, CA = 1, C = 2 (protstruc.general.ATOM).
__device__ inline void v3_load(const float* p, float* o) { o[0] = p[0]; o[1] = p[1]; o[2] = p[2]; }
__device__ inline void v3_sub(const float* a, const float* b, float* o) { o[0] = a[0] - b[0]; o[1] = a[1] - b[1]; o[2] = a[2] - b[2]; }
__device__ inline float v3_dot(const float* a, const float* b) { return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]; }
__device__ inline void v3_cross(const float* a, const float* b, float* o) {
  o[0] = a[1] * b[2] - a[2] * b[1];
  o[1] = a[2] * b[0] - a[0] * b[2];
  o[2] = a[0] * b[1] - a[1] * b[0];
}
// IUPAC dihedral of four points: angle between the planes (p0,p1,p2) and (p1,p2,p3), atan2(|b1| b0.(b1 x b2), (b0 x b1).(b1 x b2)),
// b_k = p_{k+1} - p_k; in (-pi, pi], 0 for cis, pi for trans
__device__ inline float dihedral4(const float* p0, const float* p1, const float* p2, const float* p3) {
  float b0[3], b1[3], b2[3], n1[3], n2[3];
  v3_sub(p1, p0, b0);
  v3_sub(p2, p1, b1);
  v3_sub(p3, p2, b2);
  v3_cross(b0, b1, n1);
  v3_cross(b1, b2, n2);
  const float y = sqrtf(v3_dot(b1, b1)) * v3_dot(b0, n2);
  const float x = v3_dot(n1, n2);
  return atan2f(y, x);
}

// one thread per residue: frame with origin CA, rows of R = local axes in global coordinates (x along CA->C, y in the N-CA-C plane
// towards N, z = x cross y: the Gram-Schmidt frame of io.frames_from_backbone, `global = local @ R + t` as the hot path uses it);
// backbone dihedrals (phi, psi, omega) with their validity mask: phi_i = (C_{i-1}, N_i, CA_i, C_i), psi_i = (N_i, CA_i, C_i, N_{i+1}),
// omega_i = (CA_i, C_i, N_{i+1}, CA_{i+1}); a dihedral is valid when both residues exist, are in the same chain and are consecutive
// in the patch (angle 0 where invalid)
__global__ void featurize_residue_kernel(const float* __restrict__ xyz, const int64_t* __restrict__ chain, const uint8_t* __restrict__ rmask,
                                         int K, int A, int64_t rows, float* __restrict__ R, float* __restrict__ dih,
                                         uint8_t* __restrict__ dih_mask) {
  const int64_t r = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x;
  if (r >= rows) return;
  const float* at = xyz + r * A * 3;
  float n[3], ca[3], c[3];
  v3_load(at, n);
  v3_load(at + 3, ca);
  v3_load(at + 6, c);
  if (R != nullptr) {
    float e1[3], u[3], e2[3], e3[3];
    v3_sub(c, ca, e1);
    float inv = 1.0f / fmaxf(sqrtf(v3_dot(e1, e1)), 1e-12f);
    for (int k = 0; k < 3; ++k) e1[k] *= inv;
    v3_sub(n, ca, u);
    const float pr = v3_dot(e1, u);
    for (int k = 0; k < 3; ++k) u[k] -= pr * e1[k];
    inv = 1.0f / fmaxf(sqrtf(v3_dot(u, u)), 1e-12f);
    for (int k = 0; k < 3; ++k) e2[k] = u[k] * inv;
    v3_cross(e1, e2, e3);
    float* o = R + r * 9;
    for (int k = 0; k < 3; ++k) { o[k] = e1[k]; o[3 + k] = e2[k]; o[6 + k] = e3[k]; }
  }
  if (dih != nullptr) {
    const int l = static_cast<int>(r % K);
    const bool me = rmask ? rmask[r] != 0 : true;
    const bool prev_ok = me && l > 0 && (rmask ? rmask[r - 1] != 0 : true) && (chain ? chain[r - 1] == chain[r] : true);
    const bool next_ok = me && l + 1 < K && (rmask ? rmask[r + 1] != 0 : true) && (chain ? chain[r + 1] == chain[r] : true);
    float phi = 0.f, psi = 0.f, omg = 0.f;
    if (prev_ok) phi = dihedral4(at - A * 3 + 6, n, ca, c);
    if (next_ok) {
      psi = dihedral4(n, ca, c, at + A * 3);
      omg = dihedral4(ca, c, at + A * 3, at + A * 3 + 3);
    }
    dih[r * 3 + 0] = phi; dih[r * 3 + 1] = psi; dih[r * 3 + 2] = omg;
    if (dih_mask) { dih_mask[r * 3 + 0] = prev_ok; dih_mask[r * 3 + 1] = next_ok; dih_mask[r * 3 + 2] = next_ok; }
  }
}

// one thread per residue pair (i, j): phi_ij = (C_i, N_j, CA_j, C_j), psi_ij = (N_i, CA_i, C_i, N_j)   (data.py:78-80:
// pairwise_dihedrals(atoms_i=["C"], atoms_j=["N","CA","C"]) and (atoms_i=["N","CA","C"], atoms_j=["N"]))
__global__ void featurize_pair_kernel(const float* __restrict__ xyz, int K, int A, int64_t pairs, float* __restrict__ out) {
  const int64_t g = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x;
  if (g >= pairs) return;
  const int64_t b = g / (static_cast<int64_t>(K) * K);
  const int i = static_cast<int>((g / K) % K), j = static_cast<int>(g % K);
  const float* ai = xyz + (b * K + i) * A * 3;
  const float* aj = xyz + (b * K + j) * A * 3;
  float2 o;
  o.x = dihedral4(ai + 6, aj, aj + 3, aj + 6);
  o.y = dihedral4(ai, ai + 3, ai + 6, aj);
  reinterpret_cast<float2*>(out)[g] = o;
}

// ------------------------------------------------------------------ backward (training through encode_context)
// The reference cannot back-propagate through PairEmbedding: `distmat = torch.exp(...)` is multiplied IN PLACE by the structure
// mask after the distance MLP has saved it (diffab_pytorch.py:295-301), which autograd rejects.  The fix-forward is an
// out-of-place product and nothing else (its result is unused, so the forward is unchanged); gradients are pinned against autograd
// of the oracle restatement, and ResidueEmbedding's against the real reference.  Nothing is taped: the backward recomputes the
// forward of its chunk (the row buffers of a K = 128 patch are 27 MB; keeping them for a batch would cost more than the recompute).

// d aa_emb[s] += dfeat[:, 0:D]; d chain_emb[chain] += dfeat_chain (padding_idx = 0 takes no gradient, nn.Embedding(10, D, padding_idx=0) :65)
__global__ void residue_embed_bwd_kernel(const float* __restrict__ d_aa, const float* __restrict__ d_ch, const int64_t* __restrict__ seq,
                                         const uint8_t* __restrict__ seq_m, const int64_t* __restrict__ chain, int D, int64_t rows,
                                         float* __restrict__ g_aa, float* __restrict__ g_chain) {
  const int64_t r = blockIdx.x;
  if (r >= rows) return;
  int64_t s = seq[r];
  if (seq_m && !seq_m[r]) s = kUNK;
  const int64_t c = chain[r];
  for (int d = threadIdx.x; d < D; d += blockDim.x) {
    atomicAdd(g_aa + s * D + d, d_aa[r * D + d]);
    if (c != 0) atomicAdd(g_chain + c * D + d, d_ch[r * D + d]);
  }
}

// dO[row][:] = d_out[row][:] * atom_mask_i[CA] * atom_mask_j[CA]   (backward of pair_mask_kernel)
__global__ void pair_mask_bwd_kernel(const float* __restrict__ d_out, const float* __restrict__ amask, int K, int A, int C, int64_t row0,
                                     int64_t rows, float* __restrict__ dO) {
  const int64_t gid = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x;
  if (gid >= rows * C) return;
  const int64_t row = row0 + gid / C;
  const int64_t b = row / (static_cast<int64_t>(K) * K);
  const int i = static_cast<int>((row / K) % K), j = static_cast<int>(row % K);
  dO[gid] = d_out[row * C + gid % C] * (amask[(b * K + i) * A + kCA] * amask[(b * K + j) * A + kCA]);
}

// backward of pair_cat_kernel: d aa_pair_emb[si*21+sj] += dcat[:, 0:C]; d relpos_emb[rel] += dcat[:, C:2C] * same;
// ddf[row][c] = dcat[:, 2C + c] * (df > 0)   (distance_embedding ends with a ReLU, :212-217); the dihedral columns carry no parameter
__global__ void pair_cat_bwd_kernel(const float* __restrict__ dcat, int ldc, const float* __restrict__ df, const int64_t* __restrict__ seq,
                                    const uint8_t* __restrict__ seq_m, const int64_t* __restrict__ resid, int resid_bstride,
                                    const int64_t* __restrict__ chain, int K, int C, int max_dist, int64_t row0,
                                    float* __restrict__ g_pair, float* __restrict__ g_rel, float* __restrict__ ddf) {
  const int64_t lr = blockIdx.x, row = row0 + lr;
  const int64_t b = row / (static_cast<int64_t>(K) * K);
  const int i = static_cast<int>((row / K) % K), j = static_cast<int>(row % K);
  const int64_t ri = b * K + i, rj = b * K + j;
  const int64_t si = (seq_m && !seq_m[ri]) ? kUNK : seq[ri], sj = (seq_m && !seq_m[rj]) ? kUNK : seq[rj];
  int64_t rel = resid[b * resid_bstride + i] - resid[b * resid_bstride + j];
  rel = rel < -max_dist ? -max_dist : (rel > max_dist ? max_dist : rel);
  const float same = static_cast<float>(chain[ri] * chain[rj]);
  const float* g = dcat + lr * ldc;
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    atomicAdd(g_pair + (si * kAA + sj) * C + c, g[c]);
    if (same != 0.0f) atomicAdd(g_rel + (rel + max_dist) * C + c, g[C + c] * same);
    ddf[lr * C + c] = df[lr * C + c] > 0.0f ? g[2 * C + c] : 0.0f;
  }
}

// The same with the two table gradients summed per work-group in LDS first ((441 + 2 max_dist + 1) x C floats: 130 KiB at C = 64) and
// flushed once: the 506 table rows take 64 x 2 atomics from EVERY pair row, and as global atomics on so few addresses that was the
// single most expensive kernel of the encoder backward (1.3 ms per 11-patch chunk, profiles/r05_encode_context_bwd_kernel_stats_before.csv).
__global__ __launch_bounds__(256) void pair_cat_bwd_lds_kernel(const float* __restrict__ dcat, int ldc, const float* __restrict__ df,
                                                               const int64_t* __restrict__ seq, const uint8_t* __restrict__ seq_m,
                                                               const int64_t* __restrict__ resid, int resid_bstride,
                                                               const int64_t* __restrict__ chain, int K, int C, int max_dist, int64_t row0,
                                                               int64_t nrows, float* __restrict__ g_pair, float* __restrict__ g_rel,
                                                               float* __restrict__ ddf) {
  extern __shared__ float tab[];  // [441][C] pair | [2 max_dist + 1][C] relpos
  const int n_pair = kAA * kAA * C, n_rel = (2 * max_dist + 1) * C;
  for (int i = threadIdx.x; i < n_pair + n_rel; i += blockDim.x) tab[i] = 0.0f;
  __syncthreads();
  const int rpb = blockDim.x / C > 0 ? blockDim.x / C : 1;  // rows in flight per pass (4 at C = 64)
  const int c0 = threadIdx.x % C, rsub = threadIdx.x / C;
  for (int64_t lr = static_cast<int64_t>(blockIdx.x) * rpb + rsub; lr < nrows && rsub < rpb; lr += static_cast<int64_t>(gridDim.x) * rpb) {
    const int64_t row = row0 + lr;
    const int64_t b = row / (static_cast<int64_t>(K) * K);
    const int i = static_cast<int>((row / K) % K), j = static_cast<int>(row % K);
    const int64_t ri = b * K + i, rj = b * K + j;
    const int64_t si = (seq_m && !seq_m[ri]) ? kUNK : seq[ri], sj = (seq_m && !seq_m[rj]) ? kUNK : seq[rj];
    int64_t rel = resid[b * resid_bstride + i] - resid[b * resid_bstride + j];
    rel = rel < -max_dist ? -max_dist : (rel > max_dist ? max_dist : rel);
    const float same = static_cast<float>(chain[ri] * chain[rj]);
    const float* g = dcat + lr * ldc;
    for (int c = c0; c < C; c += blockDim.x) {  // (one trip: blockDim.x >= C)
      atomicAdd(&tab[(si * kAA + sj) * C + c], g[c]);
      if (same != 0.0f) atomicAdd(&tab[n_pair + (rel + max_dist) * C + c], g[C + c] * same);
      ddf[lr * C + c] = df[lr * C + c] > 0.0f ? g[2 * C + c] : 0.0f;
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < n_pair; i += blockDim.x)
    if (tab[i] != 0.0f) atomicAdd(g_pair + i, tab[i]);
  for (int i = threadIdx.x; i < n_rel; i += blockDim.x)
    if (tab[n_pair + i] != 0.0f) atomicAdd(g_rel + i, tab[n_pair + i]);
}

// The folded backward of mlp[0] (pair_embed_fused.hip): with g = d (mlp[0] pre-activation) [rows][C], the two embedding-table segments
// of the concatenation need only G1[s_i 21 + s_j] += g and G2[rel] += same g (summed per work-group in LDS, flushed once); everything
// about aa_pair_emb, relpos_emb and the first 2 C columns of mlp[0].weight follows from G1 / G2 by 441- and 65-row products.
__global__ __launch_bounds__(1024) void pair_table_scatter_kernel(const float* __restrict__ g, const int64_t* __restrict__ seq,
                                                                 const uint8_t* __restrict__ seq_m, const int64_t* __restrict__ resid,
                                                                 int resid_bstride, const int64_t* __restrict__ chain, int K, int C, int max_dist,
                                                                 int64_t row0, int64_t nrows, float* __restrict__ G1, float* __restrict__ G2) {
  extern __shared__ float tab[];  // [441][C] | [2 max_dist + 1][C] | row constants of the current block of 256 rows
  const int n_pair = kAA * kAA * C, n_rel = (2 * max_dist + 1) * C;
  int* r_idx = reinterpret_cast<int*>(tab + n_pair + n_rel);  // [256] table row of aa_pair_emb
  int* r_rel = r_idx + 256;                                    // [256] table row of relpos_emb
  float* r_same = reinterpret_cast<float*>(r_rel + 256);       // [256] chain_i chain_j
  for (int i = threadIdx.x; i < n_pair + n_rel; i += blockDim.x) tab[i] = 0.0f;
  // (row0 is a multiple of K K and nrows < 2^31: the (b, i, j) of a row from 32-bit divisions; as 64-bit divisions by run-time values,
  // behind dependent index loads, one row per wave and pass, the index arithmetic was this kernel's time)
  const int64_t b00 = row0 / (static_cast<int64_t>(K) * K);
  const unsigned KK = static_cast<unsigned>(K) * static_cast<unsigned>(K), nr = static_cast<unsigned>(nrows);
  const int rpb = blockDim.x / C > 0 ? blockDim.x / C : 1;  // rows scattered per pass (16 at C = 64 with 1024 threads: the loop is bound by
                                                            // the latency of its row loads, round 6: 300 -> us per chunk with 256 threads)
  const int c0 = threadIdx.x % C, rsub = threadIdx.x / C;
  for (unsigned blk = blockIdx.x * 256u; blk < nr; blk += gridDim.x * 256u) {
    __syncthreads();  // the previous block's constants are consumed (first trip: the tables are zero)
    if (threadIdx.x < 256) {  // lanes = rows: the first 256 threads take the constants of one row each (coalesced index loads)
      const unsigned lr = blk + threadIdx.x;
      if (lr < nr) {
        const unsigned bl = lr / KK, rem = lr - bl * KK;
        const int i = static_cast<int>(rem / static_cast<unsigned>(K)), j = static_cast<int>(rem - static_cast<unsigned>(i) * K);
        const int64_t b = b00 + bl;
        const int64_t ri = b * K + i, rj = b * K + j;
        const int64_t si = (seq_m && !seq_m[ri]) ? kUNK : seq[ri], sj = (seq_m && !seq_m[rj]) ? kUNK : seq[rj];
        int64_t rel = resid[b * resid_bstride + i] - resid[b * resid_bstride + j];
        rel = rel < -max_dist ? -max_dist : (rel > max_dist ? max_dist : rel);
        r_idx[threadIdx.x] = static_cast<int>(si * kAA + sj);
        r_rel[threadIdx.x] = static_cast<int>(rel) + max_dist;
        r_same[threadIdx.x] = static_cast<float>(chain[ri] * chain[rj]);
      }
    }
    __syncthreads();
    if (rsub < rpb) {  // lanes = channels: rpb rows per pass, four passes' loads in flight
#pragma unroll 16
      for (int r = rsub; r < 256; r += rpb) {
        const unsigned lr = blk + r;
        if (lr < nr) {
          const float v = g[static_cast<size_t>(lr) * C + c0];
          const float same = r_same[r];
          atomicAdd(&tab[r_idx[r] * C + c0], v);
          if (same != 0.0f) atomicAdd(&tab[n_pair + r_rel[r] * C + c0], v * same);
        }
      }
    }
  }
  __syncthreads();
  // the work-group's tables -> its own slab G1[blockIdx.x][n_pair + n_rel] (G2 unused): 256 work-groups adding 32 384 values each to the
  // same addresses were 8 M atomics per launch - most of this kernel's 296 us; launch_parts_reduce sums the slabs
  float* slab = G1 + static_cast<size_t>(blockIdx.x) * (n_pair + n_rel);
  (void)G2;
  for (int i = threadIdx.x; i < n_pair + n_rel; i += blockDim.x) slab[i] = tab[i];
}
// enc[lr][0:18] = AngularEncoding(2) of the row's two pairwise dihedrals (what pair_cat_kernel puts behind the three C-wide segments), [18:20] = 0
__global__ void pair_enc_kernel(const float* __restrict__ pdih, int64_t row0, int64_t nrows, float* __restrict__ enc) {
  const int64_t gid = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x;  // (row, dihedral)
  if (gid >= nrows * 2) return;
  const int64_t lr = gid >> 1;
  const int t = static_cast<int>(gid & 1);
  float e[9];
  angular_encode(pdih[(row0 + lr) * 2 + t], 2, e);
  for (int k = 0; k < 9; ++k) enc[lr * 20 + t * 9 + k] = e[k];
  if (t == 1) { enc[lr * 20 + 18] = 0.0f; enc[lr * 20 + 19] = 0.0f; }
}

// gW[n][e] += sum_r g[r][n] enc[r][e], n < 64, e < NE (the dihedral-encoding columns of mlp[0].weight: 64 x 18 of 64 x 210).  As a tile
// of the weight-gradient GEMM this product used 20 of 128 columns and took 166 us per chunk; here lane = n, the row index is
// wave-uniform, so a row of enc arrives through the scalar cache and the products are NE v_fmac with a scalar operand per 256-byte
// row of g: bound by the 126 MB read of g.
template <int NE>
__global__ __launch_bounds__(256) void pair_enc_tn_kernel(const float* __restrict__ g, const float* __restrict__ enc, int64_t nrows,
                                                          float* __restrict__ gW, int ldg) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(static_cast<int>(blockIdx.x * 4 + (threadIdx.x >> 6)));
  const int nwaves = static_cast<int>(gridDim.x) * 4;
  float acc[NE];
#pragma unroll
  for (int e = 0; e < NE; ++e) acc[e] = 0.0f;
#pragma unroll 8
  for (int64_t r = wave; r < nrows; r += nwaves) {
    const float v = g[r * 64 + lane];
    const float* er = enc + r * NE;
#pragma unroll
    for (int e = 0; e < NE; ++e) acc[e] = __builtin_fmaf(v, er[e], acc[e]);
  }
  __shared__ float red[4][NE][64];
#pragma unroll
  for (int e = 0; e < NE; ++e) red[threadIdx.x >> 6][e][lane] = acc[e];
  __syncthreads();
  (void)ldg;
  for (int i = threadIdx.x; i < NE * 64; i += 256) {  // -> the work-group's slab gW[blockIdx.x][n][e] (launch_parts_reduce sums the slabs)
    const int n = i / NE, e = i % NE;
    gW[static_cast<size_t>(blockIdx.x) * (NE * 64) + i] = (red[0][e][n] + red[1][e][n]) + (red[2][e][n] + red[3][e][n]);
  }
}

// d softplus(coef)[si*21+sj][p] += ddin[row][p] * d din / d c,  din = exp(-c d^2) mask  =>  d din / d c = -d^2 din
__global__ void pair_dist_bwd_kernel(const int64_t* __restrict__ seq, const uint8_t* __restrict__ seq_m, const float* __restrict__ distmat,
                                     const float* __restrict__ xyz, const float* __restrict__ din, const float* __restrict__ ddin, int K,
                                     int A, int64_t row0, int64_t nrows, int ld, float* __restrict__ g_coef_sp) {
  const int AA2 = A * A;
  constexpr int RPB = 8;
  for (int rr = 0; rr < RPB; ++rr) {
    const int64_t lr = static_cast<int64_t>(blockIdx.x) * RPB + rr;
    if (lr >= nrows) return;
    const int64_t row = row0 + lr;
    const int64_t b = row / (static_cast<int64_t>(K) * K);
    const int i = static_cast<int>((row / K) % K), j = static_cast<int>(row % K);
    const int64_t ri = b * K + i, rj = b * K + j;
    const int64_t si = (seq_m && !seq_m[ri]) ? kUNK : seq[ri], sj = (seq_m && !seq_m[rj]) ? kUNK : seq[rj];
    float* grow = g_coef_sp + (si * kAA + sj) * AA2;
    for (int p = threadIdx.x; p < AA2; p += blockDim.x) {
      const float v = din[lr * ld + p];
      if (v == 0.0f) continue;  // masked atom pair (or underflow): no gradient
      const int a1 = p / A, a2 = p % A;
      float d;
      if (distmat) {
        d = distmat[row * AA2 + p];
      } else {
        const float* pa = xyz + (ri * A + a1) * 3;
        const float* pb = xyz + (rj * A + a2) * 3;
        const float dx = pa[0] - pb[0], dy = pa[1] - pb[1], dz = pa[2] - pb[2];
        d = sqrtf((dx * dx + dy * dy) + dz * dz);
      }
      atomicAdd(grow + p, ddin[lr * ld + p] * (-(d * d) * v));
    }
  }
}

// The same per (b, i) group of K pair rows: the group's rows share s_i, so its gradient lands in 21 rows of the table - summed in LDS
// ([21][A A] floats) and flushed once per group: 21 A A global atomics per K rows instead of K A A (the kernel above took 4.4 ms per
// backward at B = 128: 118 M atomics on 99 k addresses).  Four rows in flight per pass (blockDim = 4 x 256).
__global__ __launch_bounds__(1024) void pair_dist_bwd_group_kernel(const int64_t* __restrict__ seq, const uint8_t* __restrict__ seq_m,
                                                                   const float* __restrict__ distmat, const float* __restrict__ xyz,
                                                                   const float* __restrict__ din, const float* __restrict__ ddin, int K, int A,
                                                                   int64_t row0, int ld, float* __restrict__ g_coef_sp) {
  extern __shared__ float tab[];  // [21][A A]
  const int AA2 = A * A;
  for (int i = threadIdx.x; i < kAA * AA2; i += blockDim.x) tab[i] = 0.0f;
  __syncthreads();
  const int64_t lrow0 = static_cast<int64_t>(blockIdx.x) * K;  // first row of the group inside this launch
  const int64_t grow0 = row0 + lrow0;                           // global pair row (b, i, 0)
  const int64_t bi = grow0 / K;                                 // b K + i
  const int64_t b = bi / K;
  const int64_t ri = bi;
  const int64_t si = (seq_m && !seq_m[ri]) ? kUNK : seq[ri];
  const int sub = threadIdx.x >> 8, p0 = threadIdx.x & 255;
  for (int p = p0; p < AA2; p += 256) {  // (one trip for A <= 16)
    const int a1 = p / A, a2 = p % A;
    float xa0 = 0.f, xa1 = 0.f, xa2 = 0.f;
    if (!distmat) {
      const float* pa = xyz + (ri * A + a1) * 3;
      xa0 = pa[0]; xa1 = pa[1]; xa2 = pa[2];
    }
    // four rows per thread in flight (sixteen per work-group): a row costs five dependent-free loads, and one row at a time left the
    // group waiting for an L2 round trip per pass
    for (int j0 = sub; j0 < K; j0 += 16) {
      float v[4], gd[4], dd[4];
      int sj[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int j = j0 + 4 * u;
        v[u] = 0.0f; gd[u] = 0.0f; dd[u] = 0.0f; sj[u] = 0;
        if (j < K) {
          const int64_t rj = b * K + j, lr = lrow0 + j;
          sj[u] = static_cast<int>((seq_m && !seq_m[rj]) ? kUNK : seq[rj]);
          v[u] = din[lr * ld + p];
          gd[u] = ddin[lr * ld + p];
          if (distmat) {
            const float d = distmat[(grow0 + j) * AA2 + p];
            dd[u] = d * d;
          } else {
            const float* pb = xyz + (rj * A + a2) * 3;
            const float dx = xa0 - pb[0], dy = xa1 - pb[1], dz = xa2 - pb[2];
            const float d = sqrtf((dx * dx + dy * dy) + dz * dz);
            dd[u] = d * d;
          }
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (v[u] != 0.0f) atomicAdd(&tab[sj[u] * AA2 + p], gd[u] * (-dd[u] * v[u]));  // v == 0: masked atom pair (or underflow), no gradient
    }
  }
  __syncthreads();
  float* grow = g_coef_sp + si * kAA * AA2;
  for (int i = threadIdx.x; i < kAA * AA2; i += blockDim.x)
    if (tab[i] != 0.0f) atomicAdd(grow + i, tab[i]);
}

// d softplus(coef) of a chunk of pair rows: the per-(b, i)-group kernel where its [21][A A] LDS table fits (A <= 27 within the 64 KiB a
// launch gets by default; up to 160 KiB with the attribute raised: A <= 43) and the chunk is whole groups, the per-row atomic kernel otherwise
static int launch_pair_dist_bwd(const int64_t* seq, const uint8_t* seq_m, const float* distmat, const float* xyz, const float* din,
                                const float* ddin, int K, int A, int64_t row0, int64_t nrows, int ld, float* g_coef_sp, hipStream_t st) {
  const size_t lds = sizeof(float) * kAA * A * A;
  if (pair_chain_bwd_enabled() && pair_dist_bwd_mfma_supported(K, A, row0, nrows, ld, kAA))  // the class sums on the matrix cores
    return launch_pair_dist_bwd_mfma(seq, seq_m, distmat, xyz, din, ddin, K, A, kAA, kUNK, row0, nrows, ld, g_coef_sp, st);
  if (lds <= 160 * 1024 && nrows % K == 0 && row0 % K == 0) {
    if (lds > 64 * 1024)
      DIFFAB_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(pair_dist_bwd_group_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           static_cast<int>(lds)));
    hipLaunchKernelGGL(pair_dist_bwd_group_kernel, dim3(static_cast<unsigned>(nrows / K)), dim3(1024), lds, st, seq, seq_m, distmat, xyz, din, ddin,
                       K, A, row0, ld, g_coef_sp);
  } else {
    hipLaunchKernelGGL(pair_dist_bwd_kernel, dim3(static_cast<unsigned>((nrows + 7) / 8)), dim3(256), 0, st, seq, seq_m, distmat, xyz, din, ddin, K, A,
                       row0, nrows, ld, g_coef_sp);
  }
  DIFFAB_LAUNCH_CHECK();
  return DIFFAB_OK;
}

// g_coef[n] += g_coef_sp[n] * softplus'(coef[n])   (sigmoid; 1 beyond F.softplus's threshold of 20)
__global__ void softplus_bwd_kernel(const float* __restrict__ coefw, const float* __restrict__ g_sp, int n, float* __restrict__ g_coef) {
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= n) return;
  const float x = coefw[gid];
  g_coef[gid] += g_sp[gid] * (x > 20.0f ? 1.0f : 1.0f / (1.0f + expf(-x)));
}

// g[r][0:n] += gpad[r][0:n]   (weight gradients of the padded first layers back to the parameter's own row stride)
__global__ void unpad_add_kernel(const float* __restrict__ gpad, int ld, int n, int rows, float* __restrict__ g) {
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= rows * n) return;
  const int r = gid / n, c = gid % n;
  g[gid] += gpad[r * ld + c];
}

}  // namespace diffab

using namespace diffab;

extern "C" {

static int check_ctx(const diffab_ctx_dims* d, const char* who) {
  DIFFAB_REQUIRE(d && d->B > 0 && d->K > 0 && d->A > kCA && d->D > 0 && d->C > 0 && d->max_dist > 0, DIFFAB_ERR_ARG, "%s: bad dims", who);
  return DIFFAB_OK;
}

size_t diffab_residue_embedding_workspace_bytes(const diffab_ctx_dims* d) {
  if (check_ctx(d, "residue_embedding_workspace_bytes")) return 0;
  const size_t rows = static_cast<size_t>(d->B) * d->K, Din = 2 * d->D + kAA * d->A * 3 + 39;
  return (rows * (Din + 2 * d->D + d->D + d->D) + 64) * sizeof(float);
}

int diffab_residue_embedding_fwd(const diffab_ctx_dims* d, const diffab_residue_emb_weights* w, const int64_t* seq_idx, const float* xyz,
                                 const float* orientations, const float* dihedrals, const int64_t* chain_idx, const float* atom_mask,
                                 const uint8_t* structure_context_mask, const uint8_t* sequence_context_mask, float* out, void* workspace,
                                 size_t workspace_bytes, void* stream) {
  StreamOrder order_(stream);
  if (int rc = check_ctx(d, "residue_embedding_fwd")) return rc;
  DIFFAB_REQUIRE(w && w->aa_emb && w->chain_emb && w->w0 && w->b0 && w->w2 && w->b2 && w->w4 && w->b4 && w->w6 && w->b6, DIFFAB_ERR_ARG,
                 "residue_embedding_fwd: null weight");
  DIFFAB_REQUIRE(seq_idx && xyz && orientations && dihedrals && chain_idx && atom_mask && out && workspace, DIFFAB_ERR_ARG,
                 "residue_embedding_fwd: null pointer");
  DIFFAB_REQUIRE(workspace_bytes >= diffab_residue_embedding_workspace_bytes(d), DIFFAB_ERR_WORKSPACE, "residue_embedding_fwd: workspace");
  hipStream_t st = as_stream(stream);
  const int rows = d->B * d->K, D = d->D, Din = 2 * D + kAA * d->A * 3 + 39;
  float* feat = static_cast<float*>(workspace);
  float* h1 = feat + static_cast<size_t>(rows) * Din;
  float* h2 = h1 + static_cast<size_t>(rows) * 2 * D;
  float* h3 = h2 + static_cast<size_t>(rows) * D;
  hipLaunchKernelGGL(residue_feat_kernel, dim3(rows), dim3(128), 0, st, seq_idx, xyz, orientations, dihedrals, chain_idx, atom_mask,
                     structure_context_mask, sequence_context_mask, w->aa_emb, w->chain_emb, d->K, d->A, D, feat);
  DIFFAB_LAUNCH_CHECK();
  if (int rc = launch_linear(feat, Din, w->w0, w->b0, h1, 2 * D, rows, 2 * D, Din, true, st)) return rc;
  if (int rc = launch_linear(h1, 2 * D, w->w2, w->b2, h2, D, rows, D, 2 * D, true, st)) return rc;
  if (int rc = launch_linear(h2, D, w->w4, w->b4, h3, D, rows, D, D, true, st)) return rc;
  return launch_linear(h3, D, w->w6, w->b6, out, D, rows, D, D, false, st);
}

static int round4(int n) { return (n + 3) & ~3; }

// dst[r][0:ld_dst] = src[r][0:n] followed by zeros (weight rows padded to a multiple of 4 floats: 16-byte aligned rows)
__global__ void pad_rows_kernel(const float* __restrict__ src, int n, int rows, float* __restrict__ dst, int ld_dst) {
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= rows * ld_dst) return;
  const int r = gid / ld_dst, c = gid % ld_dst;
  dst[gid] = c < n ? src[r * n + c] : 0.0f;
}

static int pair_chunk_patches(const diffab_ctx_dims* d) {
  const size_t per_patch = static_cast<size_t>(d->K) * d->K * (round4(d->A * d->A) + 4 * d->C + round4(3 * d->C + 18)) * sizeof(float);
  const size_t budget = static_cast<size_t>(512) << 20;  // ~0.5 GiB of row buffers at a time
  const size_t n = budget / per_patch;
  return static_cast<int>(n < 1 ? 1 : (n > static_cast<size_t>(d->B) ? d->B : n));
}

size_t diffab_pair_embedding_workspace_bytes(const diffab_ctx_dims* d) {
  if (check_ctx(d, "pair_embedding_workspace_bytes")) return 0;
  if (pair_embed_fused_supported(d)) return (pair_embed_fused_prep_floats(d) + 64) * sizeof(float);
  const size_t rows = static_cast<size_t>(pair_chunk_patches(d)) * d->K * d->K;
  const size_t wpad = static_cast<size_t>(d->C) * (round4(d->A * d->A) + round4(3 * d->C + 18)) + static_cast<size_t>(kAA) * kAA * d->A * d->A;
  return (rows * (round4(d->A * d->A) + 4 * d->C + round4(3 * d->C + 18)) + wpad + 64) * sizeof(float);
}

static int pair_embedding_impl(const diffab_ctx_dims* d, const diffab_pair_emb_weights* w, const int64_t* seq_idx, const float* distmat,
                               const float* xyz, const float* pairwise_dihedrals, const int64_t* residue_idx,
                               int32_t residue_idx_batch_stride, const int64_t* chain_idx, const float* atom_mask,
                               const uint8_t* sequence_context_mask, float* out, void* workspace, size_t workspace_bytes, void* stream) {
  if (int rc = check_ctx(d, "pair_embedding_fwd")) return rc;
  DIFFAB_REQUIRE(w && w->aa_pair_emb && w->relpos_emb && w->pair2distcoef && w->dw0 && w->db0 && w->dw2 && w->db2 && w->mw0 && w->mb0 &&
                     w->mw2 && w->mb2 && w->mw4 && w->mb4,
                 DIFFAB_ERR_ARG, "pair_embedding_fwd: null weight");
  DIFFAB_REQUIRE(seq_idx && (distmat || xyz) && pairwise_dihedrals && residue_idx && chain_idx && atom_mask && out && workspace,
                 DIFFAB_ERR_ARG, "pair_embedding_fwd: null pointer");
  DIFFAB_REQUIRE(workspace_bytes >= diffab_pair_embedding_workspace_bytes(d), DIFFAB_ERR_WORKSPACE, "pair_embedding_fwd: workspace");
  hipStream_t st = as_stream(stream);
  if (pair_embed_fused_supported(d))  // C = 64, K % 128 == 0: the whole forward as one kernel (pair_embed_fused.hip)
    return launch_pair_embed_fused(d, w, seq_idx, distmat, xyz, pairwise_dihedrals, residue_idx, residue_idx_batch_stride, chain_idx, atom_mask,
                                   sequence_context_mask, out, static_cast<float*>(workspace), nullptr, nullptr, nullptr, nullptr, 0,
                                   static_cast<int64_t>(d->B) * d->K * d->K, st);
  const int C = d->C, AA2 = d->A * d->A, W = 3 * C + 18, AA2p = round4(AA2), Wp = round4(W);
  const int bc = pair_chunk_patches(d);
  const int64_t per_patch = static_cast<int64_t>(d->K) * d->K;
  float* din = static_cast<float*>(workspace);
  float* h1 = din + static_cast<size_t>(bc) * per_patch * AA2p;
  float* df = h1 + static_cast<size_t>(bc) * per_patch * C;
  float* cat = df + static_cast<size_t>(bc) * per_patch * C;
  float* m1 = cat + static_cast<size_t>(bc) * per_patch * Wp;
  float* m2 = m1 + static_cast<size_t>(bc) * per_patch * C;
  float* dw0p = m2 + static_cast<size_t>(bc) * per_patch * C;  // [C][AA2p]
  float* mw0p = dw0p + static_cast<size_t>(C) * AA2p;          // [C][Wp]
  float* coef_sp = mw0p + static_cast<size_t>(C) * Wp;          // [441][AA2] softplus(pair2distcoef)
  hipLaunchKernelGGL(softplus_table_kernel, dim3((kAA * kAA * AA2 + 255) / 256), dim3(256), 0, st, w->pair2distcoef, kAA * kAA * AA2, coef_sp);
  hipLaunchKernelGGL(pad_rows_kernel, dim3((C * AA2p + 255) / 256), dim3(256), 0, st, w->dw0, AA2, C, dw0p, AA2p);
  hipLaunchKernelGGL(pad_rows_kernel, dim3((C * Wp + 255) / 256), dim3(256), 0, st, w->mw0, W, C, mw0p, Wp);
  DIFFAB_LAUNCH_CHECK();
  for (int b0 = 0; b0 < d->B; b0 += bc) {
    const int nb = (d->B - b0) < bc ? (d->B - b0) : bc;
    const int64_t row0 = b0 * per_patch, nrows = nb * per_patch;
    const int rows = static_cast<int>(nrows);
    hipLaunchKernelGGL(pair_dist_kernel, dim3(static_cast<unsigned>((nrows + 7) / 8)), dim3(256), 0, st, seq_idx, sequence_context_mask, distmat, xyz,
                       atom_mask, coef_sp, d->K, d->A, row0, nrows, din, AA2p);
    DIFFAB_LAUNCH_CHECK();
    if (int rc = launch_linear(din, AA2p, dw0p, w->db0, h1, C, rows, C, AA2p, true, st)) return rc;
    if (int rc = launch_linear(h1, C, w->dw2, w->db2, df, C, rows, C, C, true, st)) return rc;
    hipLaunchKernelGGL(pair_cat_kernel, dim3(static_cast<unsigned>(nrows)), dim3(64), 0, st, seq_idx, sequence_context_mask, residue_idx,
                       residue_idx_batch_stride, chain_idx, pairwise_dihedrals, df, w->aa_pair_emb, w->relpos_emb, d->K, C, d->max_dist,
                       row0, cat, Wp);
    DIFFAB_LAUNCH_CHECK();
    float* o = out + row0 * C;
    if (int rc = launch_linear(cat, Wp, mw0p, w->mb0, m1, C, rows, C, Wp, true, st)) return rc;
    if (int rc = launch_linear(m1, C, w->mw2, w->mb2, m2, C, rows, C, C, true, st)) return rc;
    if (int rc = launch_linear(m2, C, w->mw4, w->mb4, o, C, rows, C, C, false, st)) return rc;
  }
  const int64_t total = static_cast<int64_t>(d->B) * per_patch;
  hipLaunchKernelGGL(pair_mask_kernel, dim3(static_cast<unsigned>((total * C + 255) / 256)), dim3(256), 0, st, out, atom_mask, d->K, d->A, C,
                     total);
  DIFFAB_LAUNCH_CHECK();
  return DIFFAB_OK;
}

int diffab_pair_embedding_fwd(const diffab_ctx_dims* d, const diffab_pair_emb_weights* w, const int64_t* seq_idx, const float* distmat,
                              const float* pairwise_dihedrals, const int64_t* residue_idx, int32_t residue_idx_batch_stride,
                              const int64_t* chain_idx, const float* atom_mask, const uint8_t* sequence_context_mask, float* out,
                              void* workspace, size_t workspace_bytes, void* stream) {
  StreamOrder order_(stream);
  DIFFAB_REQUIRE(distmat != nullptr, DIFFAB_ERR_ARG, "pair_embedding_fwd: distmat is null");
  return pair_embedding_impl(d, w, seq_idx, distmat, nullptr, pairwise_dihedrals, residue_idx, residue_idx_batch_stride, chain_idx, atom_mask,
                             sequence_context_mask, out, workspace, workspace_bytes, stream);
}

// ---- backward of ResidueEmbedding: parameter gradients accumulate (+=) into `g` (same layout as the weights; caller zero-fills)
size_t diffab_residue_embedding_bwd_workspace_bytes(const diffab_ctx_dims* d) {
  if (check_ctx(d, "residue_embedding_bwd_workspace_bytes")) return 0;
  const size_t rows = static_cast<size_t>(d->B) * d->K, Din = 2 * d->D + kAA * d->A * 3 + 39;
  // forward row buffers (feat, h1, h2, h3) + dh3, dh2 (D each), dh1 (2D), d_aa, d_chain (D each)
  return (rows * (Din + 2 * d->D + d->D + d->D) + rows * (2 * d->D + 2 * d->D + 2 * d->D) + 64) * sizeof(float);
}

int diffab_residue_embedding_bwd(const diffab_ctx_dims* d, const diffab_residue_emb_weights* w, const diffab_residue_emb_weights* g,
                                 const int64_t* seq_idx, const float* xyz, const float* orientations, const float* dihedrals,
                                 const int64_t* chain_idx, const float* atom_mask, const uint8_t* structure_context_mask,
                                 const uint8_t* sequence_context_mask, const float* d_out, void* workspace, size_t workspace_bytes,
                                 void* stream) {
  StreamOrder order_(stream);
  if (int rc = check_ctx(d, "residue_embedding_bwd")) return rc;
  DIFFAB_REQUIRE(w && w->aa_emb && w->chain_emb && w->w0 && w->b0 && w->w2 && w->b2 && w->w4 && w->b4 && w->w6 && w->b6, DIFFAB_ERR_ARG,
                 "residue_embedding_bwd: null weight");
  DIFFAB_REQUIRE(g && g->aa_emb && g->chain_emb && g->w0 && g->b0 && g->w2 && g->b2 && g->w4 && g->b4 && g->w6 && g->b6, DIFFAB_ERR_ARG,
                 "residue_embedding_bwd: null gradient buffer");
  DIFFAB_REQUIRE(seq_idx && xyz && orientations && dihedrals && chain_idx && atom_mask && d_out && workspace, DIFFAB_ERR_ARG,
                 "residue_embedding_bwd: null pointer");
  DIFFAB_REQUIRE(workspace_bytes >= diffab_residue_embedding_bwd_workspace_bytes(d), DIFFAB_ERR_WORKSPACE, "residue_embedding_bwd: workspace");
  hipStream_t st = as_stream(stream);
  const int rows = d->B * d->K, D = d->D, Din = 2 * D + kAA * d->A * 3 + 39;
  float* feat = static_cast<float*>(workspace);
  float* h1 = feat + static_cast<size_t>(rows) * Din;
  float* h2 = h1 + static_cast<size_t>(rows) * 2 * D;
  float* h3 = h2 + static_cast<size_t>(rows) * D;
  float* dh3 = h3 + static_cast<size_t>(rows) * D;
  float* dh2 = dh3 + static_cast<size_t>(rows) * D;
  float* dh1 = dh2 + static_cast<size_t>(rows) * D;
  float* d_aa = dh1 + static_cast<size_t>(rows) * 2 * D;
  float* d_ch = d_aa + static_cast<size_t>(rows) * D;
  auto mut = [](const float* p) { return const_cast<float*>(p); };
  // forward recompute (same launches as diffab_residue_embedding_fwd up to the last hidden layer)
  hipLaunchKernelGGL(residue_feat_kernel, dim3(rows), dim3(128), 0, st, seq_idx, xyz, orientations, dihedrals, chain_idx, atom_mask,
                     structure_context_mask, sequence_context_mask, w->aa_emb, w->chain_emb, d->K, d->A, D, feat);
  DIFFAB_LAUNCH_CHECK();
  if (int rc = launch_linear(feat, Din, w->w0, w->b0, h1, 2 * D, rows, 2 * D, Din, true, st)) return rc;
  if (int rc = launch_linear(h1, 2 * D, w->w2, w->b2, h2, D, rows, D, 2 * D, true, st)) return rc;
  if (int rc = launch_linear(h2, D, w->w4, w->b4, h3, D, rows, D, D, true, st)) return rc;
  // backward chain
  if (int rc = bwd_linear(d_out, D, h3, D, w->w6, mut(g->w6), mut(g->b6), dh3, D, rows, D, D, false, st)) return rc;
  if (int rc = bwd_relu_mask(dh3, h3, static_cast<int64_t>(rows) * D, st)) return rc;
  if (int rc = bwd_linear(dh3, D, h2, D, w->w4, mut(g->w4), mut(g->b4), dh2, D, rows, D, D, false, st)) return rc;
  if (int rc = bwd_relu_mask(dh2, h2, static_cast<int64_t>(rows) * D, st)) return rc;
  if (int rc = bwd_linear(dh2, D, h1, 2 * D, w->w2, mut(g->w2), mut(g->b2), dh1, 2 * D, rows, D, 2 * D, false, st)) return rc;
  if (int rc = bwd_relu_mask(dh1, h1, static_cast<int64_t>(rows) * 2 * D, st)) return rc;
  if (int rc = bwd_linear(dh1, 2 * D, feat, Din, w->w0, mut(g->w0), mut(g->b0), nullptr, 0, rows, 2 * D, Din, false, st)) return rc;
  // only the two embedding slices of d feat are needed: columns [0, D) and [Din - D, Din) of dh1 W0
  if (int rc = bwd_gemm_nn(dh1, 2 * D, w->w0, Din, d_aa, D, rows, D, 2 * D, false, st)) return rc;
  if (int rc = bwd_gemm_nn(dh1, 2 * D, w->w0 + (Din - D), Din, d_ch, D, rows, D, 2 * D, false, st)) return rc;
  hipLaunchKernelGGL(residue_embed_bwd_kernel, dim3(rows), dim3(128), 0, st, d_aa, d_ch, seq_idx, sequence_context_mask, chain_idx, D,
                     static_cast<int64_t>(rows), mut(g->aa_emb), mut(g->chain_emb));
  DIFFAB_LAUNCH_CHECK();
  return DIFFAB_OK;
}

// ---- backward of PairEmbedding
static int pair_bwd_chunk_patches(const diffab_ctx_dims* d) {
  const size_t per_patch = static_cast<size_t>(d->K) * d->K * (2 * round4(d->A * d->A) + 9 * d->C + 2 * round4(3 * d->C + 18)) * sizeof(float);
  const size_t budget = static_cast<size_t>(1024) << 20;
  const size_t n = budget / per_patch;
  return static_cast<int>(n < 1 ? 1 : (n > static_cast<size_t>(d->B) ? d->B : n));
}

// ---- the backward where the fused forward applies: per chunk ONE recompute launch that leaves the four hidden activations on a tape,
// then the chain of 64-wide linear backward steps on the taped rows; the 210-wide concatenation and its gradient are never built
// (pair_table_scatter_kernel + four small products), only the 225-wide distance features are, for distance_embedding[0]'s gradients.
static int fused_bwd_chunk_patches(const diffab_ctx_dims* d) {
  const size_t per_patch = static_cast<size_t>(d->K) * d->K * (9 * d->C + 2 * round4(d->A * d->A) + 20) * sizeof(float);
  const size_t n = (static_cast<size_t>(2) << 30) / per_patch;  // ~2 GiB of row buffers at a time (30 K = 128 patches)
  return static_cast<int>(n < 1 ? 1 : (n > static_cast<size_t>(d->B) ? d->B : n));
}
static size_t fused_bwd_parts_floats(const diffab_ctx_dims* d) {
  const size_t a = pair_chain_bwd_part_floats(), b = static_cast<size_t>(256) * (kAA * kAA + 2 * d->max_dist + 1) * d->C, c = 2048 * 20 * 64;
  return (a > b ? (a > c ? a : c) : (b > c ? b : c)) + 64;
}
static size_t fused_bwd_workspace_floats(const diffab_ctx_dims* d) {
  const size_t R = static_cast<size_t>(fused_bwd_chunk_patches(d)) * d->K * d->K;
  const size_t AA2p = round4(d->A * d->A), Wp = round4(3 * d->C + 18);
  return R * (9 * d->C + 2 * AA2p + 20) + 2 * static_cast<size_t>(d->C) * (AA2p + Wp) + static_cast<size_t>(kAA) * kAA * d->A * d->A +
         static_cast<size_t>(kAA * kAA + 2 * d->max_dist + 1) * d->C + pair_embed_fused_prep_floats(d) + 2 * pair_chain_bwd_prep_floats() + fused_bwd_parts_floats(d) + 1024;
}

static int pair_embedding_bwd_fused(const diffab_ctx_dims* d, const diffab_pair_emb_weights* w, const diffab_pair_emb_weights* g,
                                    const int64_t* seq_idx, const float* distmat, const float* xyz, const float* pairwise_dihedrals,
                                    const int64_t* residue_idx, int32_t residue_idx_batch_stride, const int64_t* chain_idx,
                                    const float* atom_mask, const uint8_t* sequence_context_mask, const float* d_out, float* ws, hipStream_t st,
                                    const float* tape = nullptr) {
  // tape (diffab_pair_embedding_fwd_taped): h1 | df | m1 | m2 of ALL rows, [B K K][C] each - no recompute per chunk
  const int C = d->C, AA2 = d->A * d->A, W = 3 * C + 18, AA2p = round4(AA2), Wp = round4(W);
  const int bc = fused_bwd_chunk_patches(d);
  const int64_t per_patch = static_cast<int64_t>(d->K) * d->K;
  const size_t R = static_cast<size_t>(bc) * per_patch;
  Carver cv(ws);
  float* h1w = cv.take<float>(R * C);
  float* dfw = cv.take<float>(R * C);
  float* m1w = cv.take<float>(R * C);
  float* m2w = cv.take<float>(R * C);
  float* dA = cv.take<float>(R * C);
  float* dB = cv.take<float>(R * C);
  float* dC = cv.take<float>(R * C);
  float* ddf = cv.take<float>(R * C);
  float* dh1 = cv.take<float>(R * C);
  float* din = cv.take<float>(R * AA2p);
  float* ddin = cv.take<float>(R * AA2p);
  float* enc = cv.take<float>(R * 20);
  float* dw0p = cv.take<float>(static_cast<size_t>(C) * AA2p);   // padded first-layer weights and their gradients
  float* mw0p = cv.take<float>(static_cast<size_t>(C) * Wp);
  float* gdw0p = cv.take<float>(static_cast<size_t>(C) * (AA2p + Wp));  // gdw0p | gmw0p adjacent
  float* gmw0p = gdw0p + static_cast<size_t>(C) * AA2p;
  float* g_sp = cv.take<float>(static_cast<size_t>(kAA) * kAA * AA2);
  float* G1 = cv.take<float>(static_cast<size_t>(kAA * kAA + 2 * d->max_dist + 1) * C);  // G1 | G2 adjacent
  float* G2 = G1 + static_cast<size_t>(kAA) * kAA * C;
  float* prep = cv.take<float>(pair_embed_fused_prep_floats(d));
  float* chain_prep = cv.take<float>(pair_chain_bwd_prep_floats());
  float* dist_prep = cv.take<float>(pair_chain_bwd_prep_floats());
  float* parts = cv.take<float>(fused_bwd_parts_floats(d));  // per-work-group partial sums of the chain / scatter / narrow-product kernels
  auto mut = [](const float* p) { return const_cast<float*>(p); };
  hipLaunchKernelGGL(pad_rows_kernel, dim3((C * AA2p + 255) / 256), dim3(256), 0, st, w->dw0, AA2, C, dw0p, AA2p);
  hipLaunchKernelGGL(pad_rows_kernel, dim3((C * Wp + 255) / 256), dim3(256), 0, st, w->mw0, W, C, mw0p, Wp);
  DIFFAB_LAUNCH_CHECK();
  DIFFAB_HIP_CHECK(hipMemsetAsync(gdw0p, 0, sizeof(float) * static_cast<size_t>(C) * (AA2p + Wp), st));
  DIFFAB_HIP_CHECK(hipMemsetAsync(g_sp, 0, sizeof(float) * static_cast<size_t>(kAA) * kAA * AA2, st));
  DIFFAB_HIP_CHECK(hipMemsetAsync(G1, 0, sizeof(float) * static_cast<size_t>(kAA * kAA + 2 * d->max_dist + 1) * C, st));
  const size_t tab_bytes = static_cast<size_t>(kAA * kAA + 2 * d->max_dist + 1) * C * sizeof(float) + 3 * 256 * sizeof(float);
  DIFFAB_REQUIRE(tab_bytes <= 150 * 1024, DIFFAB_ERR_UNSUPPORTED, "pair_embedding_bwd: embedding tables too large for the LDS scatter");
  const float* coef_sp = nullptr;
  const size_t total_rows = static_cast<size_t>(d->B) * per_patch;
  if (tape != nullptr) {  // only the softplus table of the recompute's preparation is needed (the 225-wide features below)
    float* csp = prep;
    if (int rc = launch_softplus_table(w->pair2distcoef, kAA * kAA * AA2, csp, st)) return rc;
    coef_sp = csp;
  }
  for (int b0 = 0; b0 < d->B; b0 += bc) {
    const int nb = (d->B - b0) < bc ? (d->B - b0) : bc;
    const int64_t row0 = b0 * per_patch, nrows = nb * per_patch;
    const int rows = static_cast<int>(nrows);
    const float* h1 = h1w; const float* df = dfw; const float* m1 = m1w; const float* m2 = m2w;
    if (tape != nullptr) {
      h1 = tape + row0 * C; df = tape + total_rows * C + row0 * C; m1 = tape + 2 * total_rows * C + row0 * C; m2 = tape + 3 * total_rows * C + row0 * C;
    } else if (int rc = launch_pair_embed_fused(d, w, seq_idx, distmat, xyz, pairwise_dihedrals, residue_idx, residue_idx_batch_stride,
                                                chain_idx, atom_mask, sequence_context_mask, nullptr, prep, h1w, dfw, m1w, m2w, row0, nrows, st,
                                                &coef_sp)) {
      // ---- forward recompute: one launch, the hidden activations (after their ReLUs) on the tape
      return rc;
    }
    if (pair_chain_bwd_enabled() && pair_chain_bwd_supported(C, d->K, nrows)) {
      // ---- the atom-mask product, the four 64 x 64 layers' d x chain and their weight / bias gradients: ONE launch, a 128-row tile stays
      //      on the CU from d out to d h1 (pair_chain_bwd.hip); d C and d h1 leave for the steps below
      const float* Xs[4] = {m2, m1, df, h1};
      const float* Wsrc[4] = {w->mw4, w->mw2, mw0p + 2 * C, w->dw2};
      const int ldws[4] = {C, C, Wp, C};
      float* Gs[4] = {mut(g->mw4), mut(g->mw2), gmw0p + 2 * C, mut(g->dw2)};
      const int ldgs[4] = {C, C, Wp, C};
      float* gbs[4] = {mut(g->mb4), mut(g->mb2), mut(g->mb0), mut(g->db2)};
      if (int rc = launch_pair_chain_bwd(d_out, atom_mask, d->K, d->A, kCA, row0, nrows, Xs, Wsrc, ldws, dC, dh1, Gs, ldgs, gbs, chain_prep, parts, st))
        return rc;
    } else {
      // ---- mlp[4], mlp[2]
      hipLaunchKernelGGL(pair_mask_bwd_kernel, dim3(static_cast<unsigned>((nrows * C + 255) / 256)), dim3(256), 0, st, d_out, atom_mask, d->K,
                         d->A, C, row0, nrows, dA);
      DIFFAB_LAUNCH_CHECK();
      // ---- the dX chain: four 64 x 64 products, each masked by the ReLU below it in the product's epilogue
      //      dA = d out (masked) -> dB = d mlp[2] pre-activation -> dC = d mlp[0] pre-activation -> ddf -> dh1
      if (int rc = bwd_gemm_nn_masked(dA, C, w->mw4, C, dB, C, rows, C, C, m2, st)) return rc;
      if (int rc = bwd_gemm_nn_masked(dB, C, w->mw2, C, dC, C, rows, C, C, m1, st)) return rc;
      if (int rc = bwd_gemm_nn_masked(dC, C, mw0p + 2 * C, Wp, ddf, C, rows, C, C, df, st)) return rc;  // distance_embedding ends with a ReLU (:212-217)
      if (int rc = bwd_gemm_nn_masked(ddf, C, w->dw2, C, dh1, C, rows, C, C, h1, st)) return rc;
      // ---- the four 64 x 64 weight gradients (+ bias gradients) of mlp[4], mlp[2], mlp[0][:, 2C:3C], distance_embedding[2] in one launch
      {
        const float* As[4] = {dA, dB, dC, ddf};
        const float* Bs[4] = {m2, m1, df, h1};
        float* Cs[4] = {mut(g->mw4), mut(g->mw2), gmw0p + 2 * C, mut(g->dw2)};
        const int ldcs[4] = {C, C, Wp, C};
        float* dbs[4] = {mut(g->mb4), mut(g->mb2), mut(g->mb0), mut(g->db2)};
        if (int rc = bwd_tn64_set(4, As, Bs, Cs, ldcs, dbs, nrows, st)) return rc;
      }
    }
    // ---- the rest of mlp[0] without its concatenation: the dihedral-encoding columns [3C, 3C + 18) are a plain product; the table
    // segments go through G1 / G2 after the loop
    hipLaunchKernelGGL(pair_enc_kernel, dim3(static_cast<unsigned>((nrows * 2 + 255) / 256)), dim3(256), 0, st, pairwise_dihedrals, row0, nrows,
                       enc);
    DIFFAB_LAUNCH_CHECK();
    if (C == 64 && Wp - 3 * C == 20) {
      hipLaunchKernelGGL(pair_enc_tn_kernel<20>, dim3(1024), dim3(256), 0, st, dC, enc, nrows, parts, 0);
      DIFFAB_LAUNCH_CHECK();
      PartsSegs sg{};
      sg.nseg = 1; sg.off[0] = 0; sg.n[0] = 20 * 64; sg.cols[0] = 20; sg.ld[0] = Wp; sg.out[0] = gmw0p + 3 * C;
      if (int rc = launch_parts_reduce(parts, 1024, 20 * 64, sg, st)) return rc;
    } else if (int rc = bwd_gemm_tn(dC, C, enc, 20, gmw0p + 3 * C, Wp, rows, C, Wp - 3 * C, nullptr, st)) {
      return rc;
    }
    if (pair_chain_bwd_enabled() && pair_table_mfma_supported(C, d->K, nrows, kAA, d->max_dist)) {
      // the table sums as one-hot products on the matrix cores (pair_chain_bwd.hip): no LDS atomics
      if (int rc = launch_pair_table_mfma(dC, seq_idx, sequence_context_mask, residue_idx, residue_idx_batch_stride, chain_idx, d->K, d->max_dist,
                                          kAA, kUNK, row0, nrows, G1, parts, st))
        return rc;
    } else {
      DIFFAB_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(pair_table_scatter_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           static_cast<int>(tab_bytes)));
      hipLaunchKernelGGL(pair_table_scatter_kernel, dim3(256), dim3(1024), tab_bytes, st, dC, seq_idx, sequence_context_mask, residue_idx,
                         residue_idx_batch_stride, chain_idx, d->K, C, d->max_dist, row0, nrows, parts, nullptr);
      DIFFAB_LAUNCH_CHECK();
      {
        const int n_tab = (kAA * kAA + 2 * d->max_dist + 1) * C;
        PartsSegs sg{};
        sg.nseg = 1; sg.off[0] = 0; sg.n[0] = n_tab; sg.cols[0] = n_tab; sg.ld[0] = n_tab; sg.out[0] = G1;  // (G1 | G2 adjacent)
        if (int rc = launch_parts_reduce(parts, 256, n_tab, sg, st)) return rc;
      }
    }
    // ---- distance_embedding[0]: the only place the 225-wide features are materialised
    {
      const size_t gl_bytes = (static_cast<size_t>(d->K) * d->A * 4 + d->A * 4 + d->K) * sizeof(float);
      DIFFAB_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(pair_dist_group_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           static_cast<int>(gl_bytes)));
      hipLaunchKernelGGL(pair_dist_group_kernel, dim3(static_cast<unsigned>(nrows / d->K)), dim3(256), gl_bytes, st, seq_idx, sequence_context_mask,
                         distmat, xyz, atom_mask, coef_sp, d->K, d->A, row0, din, AA2p);
      DIFFAB_LAUNCH_CHECK();
    }
    if (pair_chain_bwd_enabled() && pair_dist_bwd_mfma_supported(d->K, d->A, row0, nrows, AA2p, kAA) && AA2 <= 256) {
      // weight / bias gradient as before; d din is formed inside the coefficient-gradient kernel and never written (pair_chain_bwd.hip)
      if (int rc = bwd_linear(dh1, C, din, AA2p, dw0p, gdw0p, mut(g->db0), nullptr, AA2p, rows, C, AA2p, false, st)) return rc;
      if (int rc = launch_pair_dist_bwd_fused(seq_idx, sequence_context_mask, distmat, xyz, din, dh1, dw0p, AA2p, d->K, d->A, kAA, kUNK, row0, nrows,
                                              AA2p, g_sp, dist_prep, st))
        return rc;
    } else {
      if (int rc = bwd_linear(dh1, C, din, AA2p, dw0p, gdw0p, mut(g->db0), ddin, AA2p, rows, C, AA2p, false, st)) return rc;
      if (int rc = launch_pair_dist_bwd(seq_idx, sequence_context_mask, distmat, xyz, din, ddin, d->K, d->A, row0, nrows, AA2p, g_sp, st)) return rc;
    }
  }
  // ---- the table segments: d aa_pair_emb += G1 W_a, d W_a += G1^T aa_pair_emb (W_a = mlp[0].weight[:, 0:C]); the same for relpos_emb / G2
  const int n1 = kAA * kAA, n2 = 2 * d->max_dist + 1;
  if (int rc = bwd_gemm_nn(G1, C, mw0p, Wp, mut(g->aa_pair_emb), C, n1, C, C, true, st)) return rc;
  if (int rc = bwd_gemm_tn(G1, C, w->aa_pair_emb, C, gmw0p, Wp, n1, C, C, nullptr, st)) return rc;
  if (int rc = bwd_gemm_nn(G2, C, mw0p + C, Wp, mut(g->relpos_emb), C, n2, C, C, true, st)) return rc;
  if (int rc = bwd_gemm_tn(G2, C, w->relpos_emb, C, gmw0p + C, Wp, n2, C, C, nullptr, st)) return rc;
  hipLaunchKernelGGL(unpad_add_kernel, dim3((C * AA2 + 255) / 256), dim3(256), 0, st, gdw0p, AA2p, AA2, C, mut(g->dw0));
  hipLaunchKernelGGL(unpad_add_kernel, dim3((C * W + 255) / 256), dim3(256), 0, st, gmw0p, Wp, W, C, mut(g->mw0));
  hipLaunchKernelGGL(softplus_bwd_kernel, dim3((kAA * kAA * AA2 + 255) / 256), dim3(256), 0, st, w->pair2distcoef, g_sp, kAA * kAA * AA2,
                     mut(g->pair2distcoef));
  DIFFAB_LAUNCH_CHECK();
  return DIFFAB_OK;
}

size_t diffab_pair_embedding_bwd_workspace_bytes(const diffab_ctx_dims* d) {
  if (check_ctx(d, "pair_embedding_bwd_workspace_bytes")) return 0;
  if (pair_embed_fused_supported(d)) return fused_bwd_workspace_floats(d) * sizeof(float) + 32 * 256;
  const size_t rows = static_cast<size_t>(pair_bwd_chunk_patches(d)) * d->K * d->K;
  const size_t AA2p = round4(d->A * d->A), Wp = round4(3 * d->C + 18);
  const size_t wpad = 2 * static_cast<size_t>(d->C) * (AA2p + Wp) + 2 * static_cast<size_t>(kAA) * kAA * d->A * d->A;
  return (rows * (2 * AA2p + 9 * d->C + 2 * Wp) + wpad + 64) * sizeof(float);
}

int diffab_pair_embedding_bwd(const diffab_ctx_dims* d, const diffab_pair_emb_weights* w, const diffab_pair_emb_weights* g,
                              const int64_t* seq_idx, const float* distmat, const float* xyz, const float* pairwise_dihedrals,
                              const int64_t* residue_idx, int32_t residue_idx_batch_stride, const int64_t* chain_idx,
                              const float* atom_mask, const uint8_t* sequence_context_mask, const float* d_out, void* workspace,
                              size_t workspace_bytes, void* stream) {
  StreamOrder order_(stream);
  if (int rc = check_ctx(d, "pair_embedding_bwd")) return rc;
  DIFFAB_REQUIRE(w && w->aa_pair_emb && w->relpos_emb && w->pair2distcoef && w->dw0 && w->db0 && w->dw2 && w->db2 && w->mw0 && w->mb0 &&
                     w->mw2 && w->mb2 && w->mw4 && w->mb4,
                 DIFFAB_ERR_ARG, "pair_embedding_bwd: null weight");
  DIFFAB_REQUIRE(g && g->aa_pair_emb && g->relpos_emb && g->pair2distcoef && g->dw0 && g->db0 && g->dw2 && g->db2 && g->mw0 && g->mb0 &&
                     g->mw2 && g->mb2 && g->mw4 && g->mb4,
                 DIFFAB_ERR_ARG, "pair_embedding_bwd: null gradient buffer");
  DIFFAB_REQUIRE(seq_idx && (distmat || xyz) && pairwise_dihedrals && residue_idx && chain_idx && atom_mask && d_out && workspace,
                 DIFFAB_ERR_ARG, "pair_embedding_bwd: null pointer");
  DIFFAB_REQUIRE(workspace_bytes >= diffab_pair_embedding_bwd_workspace_bytes(d), DIFFAB_ERR_WORKSPACE, "pair_embedding_bwd: workspace");
  hipStream_t st = as_stream(stream);
  if (pair_embed_fused_supported(d))
    return pair_embedding_bwd_fused(d, w, g, seq_idx, distmat, xyz, pairwise_dihedrals, residue_idx, residue_idx_batch_stride, chain_idx,
                                    atom_mask, sequence_context_mask, d_out, static_cast<float*>(workspace), st);
  const int C = d->C, AA2 = d->A * d->A, W = 3 * C + 18, AA2p = round4(AA2), Wp = round4(W);
  const int bc = pair_bwd_chunk_patches(d);
  const int64_t per_patch = static_cast<int64_t>(d->K) * d->K;
  const size_t R = static_cast<size_t>(bc) * per_patch;
  float* din = static_cast<float*>(workspace);  // forward: [R][AA2p]
  float* h1 = din + R * AA2p;                   // [R][C]
  float* df = h1 + R * C;
  float* cat = df + R * C;                      // [R][Wp]
  float* m1 = cat + R * Wp;
  float* m2 = m1 + R * C;
  float* dA = m2 + R * C;                       // backward ping-pong [R][C] x 2
  float* dB = dA + R * C;
  float* dcat = dB + R * C;                     // [R][Wp]
  float* ddf = dcat + R * Wp;                   // [R][C]
  float* dh1 = ddf + R * C;                     // [R][C]
  float* ddin = dh1 + R * C;                    // [R][AA2p]
  float* dw0p = ddin + R * AA2p;                // padded first-layer weights and their gradients
  float* mw0p = dw0p + static_cast<size_t>(C) * AA2p;
  float* gdw0p = mw0p + static_cast<size_t>(C) * Wp;
  float* gmw0p = gdw0p + static_cast<size_t>(C) * AA2p;
  float* coef_sp = gmw0p + static_cast<size_t>(C) * Wp;  // [441][AA2]
  float* g_sp = coef_sp + static_cast<size_t>(kAA) * kAA * AA2;
  auto mut = [](const float* p) { return const_cast<float*>(p); };
  hipLaunchKernelGGL(softplus_table_kernel, dim3((kAA * kAA * AA2 + 255) / 256), dim3(256), 0, st, w->pair2distcoef, kAA * kAA * AA2, coef_sp);
  hipLaunchKernelGGL(pad_rows_kernel, dim3((C * AA2p + 255) / 256), dim3(256), 0, st, w->dw0, AA2, C, dw0p, AA2p);
  hipLaunchKernelGGL(pad_rows_kernel, dim3((C * Wp + 255) / 256), dim3(256), 0, st, w->mw0, W, C, mw0p, Wp);
  DIFFAB_LAUNCH_CHECK();
  DIFFAB_HIP_CHECK(hipMemsetAsync(gdw0p, 0, sizeof(float) * (static_cast<size_t>(C) * (AA2p + Wp) + 0), st));  // gdw0p and gmw0p are adjacent
  DIFFAB_HIP_CHECK(hipMemsetAsync(g_sp, 0, sizeof(float) * static_cast<size_t>(kAA) * kAA * AA2, st));
  for (int b0 = 0; b0 < d->B; b0 += bc) {
    const int nb = (d->B - b0) < bc ? (d->B - b0) : bc;
    const int64_t row0 = b0 * per_patch, nrows = nb * per_patch;
    const int rows = static_cast<int>(nrows);
    // ---- forward recompute of this chunk (the launches of pair_embedding_impl, last layer excepted)
    hipLaunchKernelGGL(pair_dist_kernel, dim3(static_cast<unsigned>((nrows + 7) / 8)), dim3(256), 0, st, seq_idx, sequence_context_mask, distmat, xyz,
                       atom_mask, coef_sp, d->K, d->A, row0, nrows, din, AA2p);
    DIFFAB_LAUNCH_CHECK();
    if (int rc = launch_linear(din, AA2p, dw0p, w->db0, h1, C, rows, C, AA2p, true, st)) return rc;
    if (int rc = launch_linear(h1, C, w->dw2, w->db2, df, C, rows, C, C, true, st)) return rc;
    hipLaunchKernelGGL(pair_cat_kernel, dim3(static_cast<unsigned>(nrows)), dim3(64), 0, st, seq_idx, sequence_context_mask, residue_idx,
                       residue_idx_batch_stride, chain_idx, pairwise_dihedrals, df, w->aa_pair_emb, w->relpos_emb, d->K, C, d->max_dist,
                       row0, cat, Wp);
    DIFFAB_LAUNCH_CHECK();
    if (int rc = launch_linear(cat, Wp, mw0p, w->mb0, m1, C, rows, C, Wp, true, st)) return rc;
    if (int rc = launch_linear(m1, C, w->mw2, w->mb2, m2, C, rows, C, C, true, st)) return rc;
    // ---- backward
    hipLaunchKernelGGL(pair_mask_bwd_kernel, dim3(static_cast<unsigned>((nrows * C + 255) / 256)), dim3(256), 0, st, d_out, atom_mask, d->K,
                       d->A, C, row0, nrows, dA);
    DIFFAB_LAUNCH_CHECK();
    if (int rc = bwd_linear(dA, C, m2, C, w->mw4, mut(g->mw4), mut(g->mb4), dB, C, rows, C, C, false, st)) return rc;
    if (int rc = bwd_relu_mask(dB, m2, nrows * C, st)) return rc;
    if (int rc = bwd_linear(dB, C, m1, C, w->mw2, mut(g->mw2), mut(g->mb2), dA, C, rows, C, C, false, st)) return rc;
    if (int rc = bwd_relu_mask(dA, m1, nrows * C, st)) return rc;
    if (int rc = bwd_linear(dA, C, cat, Wp, mw0p, gmw0p, mut(g->mb0), dcat, Wp, rows, C, Wp, false, st)) return rc;
    const size_t tab_bytes = static_cast<size_t>(kAA * kAA + 2 * d->max_dist + 1) * C * sizeof(float);
    if (tab_bytes <= 150 * 1024 && C <= 256) {
      DIFFAB_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(pair_cat_bwd_lds_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           static_cast<int>(tab_bytes)));
      hipLaunchKernelGGL(pair_cat_bwd_lds_kernel, dim3(256), dim3(256), tab_bytes, st, dcat, Wp, df, seq_idx, sequence_context_mask, residue_idx,
                         residue_idx_batch_stride, chain_idx, d->K, C, d->max_dist, row0, nrows, mut(g->aa_pair_emb), mut(g->relpos_emb), ddf);
    } else {
      hipLaunchKernelGGL(pair_cat_bwd_kernel, dim3(static_cast<unsigned>(nrows)), dim3(64), 0, st, dcat, Wp, df, seq_idx,
                         sequence_context_mask, residue_idx, residue_idx_batch_stride, chain_idx, d->K, C, d->max_dist, row0,
                         mut(g->aa_pair_emb), mut(g->relpos_emb), ddf);
    }
    DIFFAB_LAUNCH_CHECK();
    if (int rc = bwd_linear(ddf, C, h1, C, w->dw2, mut(g->dw2), mut(g->db2), dh1, C, rows, C, C, false, st)) return rc;
    if (int rc = bwd_relu_mask(dh1, h1, nrows * C, st)) return rc;
    if (int rc = bwd_linear(dh1, C, din, AA2p, dw0p, gdw0p, mut(g->db0), ddin, AA2p, rows, C, AA2p, false, st)) return rc;
    if (int rc = launch_pair_dist_bwd(seq_idx, sequence_context_mask, distmat, xyz, din, ddin, d->K, d->A, row0, nrows, AA2p, g_sp, st)) return rc;
  }
  hipLaunchKernelGGL(unpad_add_kernel, dim3((C * AA2 + 255) / 256), dim3(256), 0, st, gdw0p, AA2p, AA2, C, mut(g->dw0));
  hipLaunchKernelGGL(unpad_add_kernel, dim3((C * W + 255) / 256), dim3(256), 0, st, gmw0p, Wp, W, C, mut(g->mw0));
  hipLaunchKernelGGL(softplus_bwd_kernel, dim3((kAA * kAA * AA2 + 255) / 256), dim3(256), 0, st, w->pair2distcoef, g_sp, kAA * kAA * AA2,
                     mut(g->pair2distcoef));
  DIFFAB_LAUNCH_CHECK();
  return DIFFAB_OK;
}

int diffab_featurize_xyz(const float* xyz, const int64_t* chain_idx, const uint8_t* residue_mask, int32_t B, int32_t K, int32_t A,
                         float* orientations, float* backbone_dihedrals, uint8_t* backbone_dihedrals_mask, float* pairwise_dihedrals,
                         void* stream) {
  StreamOrder order_(stream);
  DIFFAB_REQUIRE(xyz && B > 0 && K > 0 && A >= 3, DIFFAB_ERR_ARG, "featurize_xyz: bad argument (needs the N, CA, C slots: A >= 3)");
  DIFFAB_REQUIRE(orientations || backbone_dihedrals || pairwise_dihedrals, DIFFAB_ERR_ARG, "featurize_xyz: no output requested");
  hipStream_t st = as_stream(stream);
  const int64_t rows = static_cast<int64_t>(B) * K;
  if (orientations || backbone_dihedrals) {
    hipLaunchKernelGGL(featurize_residue_kernel, dim3(static_cast<unsigned>((rows + 255) / 256)), dim3(256), 0, st, xyz, chain_idx,
                       residue_mask, K, A, rows, orientations, backbone_dihedrals, backbone_dihedrals_mask);
    DIFFAB_LAUNCH_CHECK();
  }
  if (pairwise_dihedrals) {
    const int64_t pairs = rows * K;
    hipLaunchKernelGGL(featurize_pair_kernel, dim3(static_cast<unsigned>((pairs + 255) / 256)), dim3(256), 0, st, xyz, K, A, pairs,
                       pairwise_dihedrals);
    DIFFAB_LAUNCH_CHECK();
  }
  return DIFFAB_OK;
}

int diffab_pair_embedding_xyz_fwd(const diffab_ctx_dims* d, const diffab_pair_emb_weights* w, const int64_t* seq_idx, const float* xyz,
                                  const float* pairwise_dihedrals, const int64_t* residue_idx, int32_t residue_idx_batch_stride,
                                  const int64_t* chain_idx, const float* atom_mask, const uint8_t* sequence_context_mask, float* out,
                                  void* workspace, size_t workspace_bytes, void* stream) {
  StreamOrder order_(stream);
  DIFFAB_REQUIRE(xyz != nullptr, DIFFAB_ERR_ARG, "pair_embedding_xyz_fwd: xyz is null");
  return pair_embedding_impl(d, w, seq_idx, nullptr, xyz, pairwise_dihedrals, residue_idx, residue_idx_batch_stride, chain_idx, atom_mask,
                             sequence_context_mask, out, workspace, workspace_bytes, stream);
}

// ---- taped form (round 6): the forward leaves the four hidden activations of every pair row, the backward reads them instead of
// recomputing the forward chunk by chunk.  4 B K K C floats (8.6 GB at B = 128, K = 128, C = 64): what 288 GB of HBM are for.
size_t diffab_pair_embedding_tape_bytes(const diffab_ctx_dims* d) {
  if (check_ctx(d, "pair_embedding_tape_bytes") || !pair_embed_fused_supported(d)) return 0;
  return static_cast<size_t>(4) * d->B * d->K * d->K * d->C * sizeof(float);
}
int diffab_pair_embedding_fwd_taped(const diffab_ctx_dims* d, const diffab_pair_emb_weights* w, const int64_t* seq_idx, const float* distmat,
                                    const float* xyz, const float* pairwise_dihedrals, const int64_t* residue_idx,
                                    int32_t residue_idx_batch_stride, const int64_t* chain_idx, const float* atom_mask,
                                    const uint8_t* sequence_context_mask, float* out, float* tape, size_t tape_bytes, void* workspace,
                                    size_t workspace_bytes, void* stream) {
  StreamOrder order_(stream);
  if (int rc = check_ctx(d, "pair_embedding_fwd_taped")) return rc;
  DIFFAB_REQUIRE(pair_embed_fused_supported(d), DIFFAB_ERR_UNSUPPORTED,
                 "pair_embedding_fwd_taped: no taped form for these dims (diffab_pair_embedding_tape_bytes returns 0): use the plain entry points");
  DIFFAB_REQUIRE(w && w->aa_pair_emb && w->relpos_emb && w->pair2distcoef && w->dw0 && w->db0 && w->dw2 && w->db2 && w->mw0 && w->mb0 &&
                     w->mw2 && w->mb2 && w->mw4 && w->mb4,
                 DIFFAB_ERR_ARG, "pair_embedding_fwd_taped: null weight");
  DIFFAB_REQUIRE(seq_idx && ((distmat != nullptr) != (xyz != nullptr)) && pairwise_dihedrals && residue_idx && chain_idx && atom_mask && out &&
                     tape && workspace && (reinterpret_cast<uintptr_t>(tape) & 15) == 0,
                 DIFFAB_ERR_ARG, "pair_embedding_fwd_taped: null pointer (exactly one of distmat / xyz), or a tape that is not 16-byte aligned");
  DIFFAB_REQUIRE(tape_bytes >= diffab_pair_embedding_tape_bytes(d), DIFFAB_ERR_WORKSPACE, "pair_embedding_fwd_taped: tape");
  DIFFAB_REQUIRE(workspace_bytes >= diffab_pair_embedding_workspace_bytes(d), DIFFAB_ERR_WORKSPACE, "pair_embedding_fwd_taped: workspace");
  const int64_t rows = static_cast<int64_t>(d->B) * d->K * d->K;
  return launch_pair_embed_fused(d, w, seq_idx, distmat, xyz, pairwise_dihedrals, residue_idx, residue_idx_batch_stride, chain_idx, atom_mask,
                                 sequence_context_mask, out, static_cast<float*>(workspace), tape, tape + rows * d->C, tape + 2 * rows * d->C,
                                 tape + 3 * rows * d->C, 0, rows, as_stream(stream), nullptr);
}
int diffab_pair_embedding_bwd_taped(const diffab_ctx_dims* d, const diffab_pair_emb_weights* w, const diffab_pair_emb_weights* g,
                                    const int64_t* seq_idx, const float* distmat, const float* xyz, const float* pairwise_dihedrals,
                                    const int64_t* residue_idx, int32_t residue_idx_batch_stride, const int64_t* chain_idx,
                                    const float* atom_mask, const uint8_t* sequence_context_mask, const float* d_out, const float* tape,
                                    size_t tape_bytes, void* workspace, size_t workspace_bytes, void* stream) {
  StreamOrder order_(stream);
  if (int rc = check_ctx(d, "pair_embedding_bwd_taped")) return rc;
  DIFFAB_REQUIRE(pair_embed_fused_supported(d), DIFFAB_ERR_UNSUPPORTED, "pair_embedding_bwd_taped: no taped form for these dims");
  DIFFAB_REQUIRE(w && w->aa_pair_emb && w->relpos_emb && w->pair2distcoef && w->dw0 && w->db0 && w->dw2 && w->db2 && w->mw0 && w->mb0 &&
                     w->mw2 && w->mb2 && w->mw4 && w->mb4,
                 DIFFAB_ERR_ARG, "pair_embedding_bwd_taped: null weight");
  DIFFAB_REQUIRE(g && g->aa_pair_emb && g->relpos_emb && g->pair2distcoef && g->dw0 && g->db0 && g->dw2 && g->db2 && g->mw0 && g->mb0 &&
                     g->mw2 && g->mb2 && g->mw4 && g->mb4,
                 DIFFAB_ERR_ARG, "pair_embedding_bwd_taped: null gradient buffer");
  DIFFAB_REQUIRE(seq_idx && (distmat || xyz) && pairwise_dihedrals && residue_idx && chain_idx && atom_mask && d_out && tape && workspace &&
                     (reinterpret_cast<uintptr_t>(tape) & 15) == 0,
                 DIFFAB_ERR_ARG, "pair_embedding_bwd_taped: null pointer");
  DIFFAB_REQUIRE(tape_bytes >= diffab_pair_embedding_tape_bytes(d), DIFFAB_ERR_WORKSPACE, "pair_embedding_bwd_taped: tape");
  DIFFAB_REQUIRE(workspace_bytes >= diffab_pair_embedding_bwd_workspace_bytes(d), DIFFAB_ERR_WORKSPACE, "pair_embedding_bwd_taped: workspace");
  return pair_embedding_bwd_fused(d, w, g, seq_idx, distmat, xyz, pairwise_dihedrals, residue_idx, residue_idx_batch_stride, chain_idx, atom_mask,
                                  sequence_context_mask, d_out, static_cast<float*>(workspace), as_stream(stream), tape);
}

}  // extern "C"
