// gemm_bf16x6.hip - fp32-accurate dense projections on the bf16 matrix cores.
//
// Why: on gfx950 the f32-input MFMA (v_mfma_f32_16x16x4_f32) runs at the fp32 VECTOR rate and - measured, tools/mfma4x4_probe.hip -
// does not overlap with VALU work at all (32 cycles per MFMA + ~3.5 per v_fma issued beside it): in fp32, matrix and vector
// arithmetic are ONE pipe, and the dense projections of an IPA layer (six projections 11.3 GFLOP, to_out 8.6 GFLOP at B = 256)
// hold it for 0.19 ms of a 0.55 ms layer.  The bf16 matrix cores are a separate pipe, 16x the rate.  An fp32 number is exactly
// the sum of three bf16 numbers (8 + 8 + 8 mantissa bits: hi = bf16(x), mid = bf16(x - hi), lo = bf16(x - hi - mid)), and
//     a b = a1 b1 + (a1 b2 + a2 b1) + (a1 b3 + a3 b1 + a2 b2) + O(2^-24 |a b|)
// so SIX bf16 MFMAs with fp32 accumulation reproduce the fp32 product to fp32 rounding (each partial product of two 8-bit
// mantissas is exact in fp32).  Measured on random data against float64: 2.7e-7 max-rel for the six-term form, 4.0e-7 for an
// fp32 GEMM of the same operands (three terms only: 4e-6 - not used).  The parity tests hold the layers built on it to the same
// bars as before.  6 MFMA 16x16x32 = 96 cycles per 16x16x32 block against 256 for the 8 f32 16x16x4 it replaces, and the VALU
// (operand splitting, epilogues) runs beside them.
//
// Y[M x 128] = act(X[M x K] W^T + b): the N = 128 layers of the denoiser (to_out with K = 1024, the MLP layers with K = 128).
// Reference: nn.Linear calls at diffab_pytorch.py:375-379, :459-464 (to_out), :515-556 (MLPs).
#include "common.h"
#include "denoiser_internal.h"

namespace diffab {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define MEM_FENCE() asm volatile("" ::: "memory")

namespace {
constexpr int BK = 32;        // k per MFMA (16x16x32) = k per weight chunk

__device__ __forceinline__ void split3(float x, __bf16& h, __bf16& m, __bf16& l) {
  h = static_cast<__bf16>(x);
  const float r = x - static_cast<float>(h);
  m = static_cast<__bf16>(r);
  l = static_cast<__bf16>(r - static_cast<float>(m));
}
}  // namespace

// W[128 x Kd] fp32 (rows ldw floats apart) -> three bf16 planes, chunk-major:  out[((chunk * 3 + s) * 128 + n) * 32 + kk],
// chunk = k / 32, kk = k % 32.  One chunk = 24 KiB contiguous: the GEMM stages it with full-line loads and no address arithmetic.
__global__ void wsplit128_kernel(const float* __restrict__ W, int ldw, int Kd, __bf16* __restrict__ out) {
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;  // (n, k)
  if (gid >= 128 * Kd) return;
  const int n = gid / Kd, k = gid % Kd;
  __bf16 h, m, l;
  split3(W[static_cast<int64_t>(n) * ldw + k], h, m, l);
  const int chunk = k / BK, kk = k % BK;
  const size_t base = (static_cast<size_t>(chunk) * 3 * 128 + n) * BK + kk;
  out[base] = h;
  out[base + 128 * BK] = m;
  out[base + 2 * 128 * BK] = l;
}

// LDS (dynamic): weights Ws[2 buffers][3 planes][128 rows][32 k], then X As[2][3][ROWS][32] - both operands are staged as split
// bf16 planes: every element of X is split ONCE per work-group (its two column waves share the rows), and X is read from HBM in
// full 128-byte lines (the MFMA fragment shape - adjacent lanes on different rows - costs the texture addresser four lines per
// lane quad: with fragment-shaped loads straight from global memory this kernel took 58 us instead of 48).  Rows are 64 bytes,
// unpadded; the 16-byte slot s of row r lives at slot s ^ ((r >> 2) & 3), which makes the 32-row ds_read_b128 fragment reads
// conflict-free (the four 16-lane groups of the instruction each take 16 different bank quads).
// ROWS = 128: 512 threads, waves 4 (rows) x 2 (columns), 96 KiB of LDS, one work-group per CU.
// ROWS = 64: 256 threads, waves 2 x 2, 72 KiB: two work-groups per CU - one group's MFMAs run under the other's staging and
// barrier - and twice the work-groups for the same rows (B <= 128 patches per GPU leave half the CUs idle with 128-row groups).
template <int ROWS>
constexpr int b6_lds_bytes() { return 2 * 3 * (128 + ROWS) * BK * 2; }
__device__ __forceinline__ int b6_off(int row, int slot) { return row * BK + 8 * (slot ^ ((row >> 2) & 3)); }  // bf16 elements

template <bool RELU, int ROWS>
__global__ __launch_bounds__(ROWS * 4) void rowgemm128_b6_kernel(const float* __restrict__ X, int ldx, const __bf16* __restrict__ Wc,
                                                                 const float* __restrict__ bias, const int64_t* __restrict__ bias_idx,
                                                                 int bias_div, float* __restrict__ Y, int ldy, int M, int Kd) {
  // bias: one vector (bias_idx == nullptr, bias_div == 0), or a table of 128-wide rows indexed by bias_idx[row] or row / bias_div
  constexpr int T = ROWS * 4, NRW = ROWS / 32;  // threads; row waves
  extern __shared__ __attribute__((aligned(16))) __bf16 b6_lds[];
  __bf16* Ws = b6_lds;
  __bf16* As = b6_lds + 2 * 3 * 128 * BK;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int l31 = lane & 31, hk = lane >> 5, rw = wv % NRW, cw = wv / NRW;  // v_mfma_f32_32x32x16_bf16: wave tile 32 rows x 64 columns
  const int m0 = blockIdx.x * ROWS;
  const int nchunk = Kd / BK;
  // weight staging: a chunk is 3 planes x 128 rows x 64 bytes = 1536 16-byte pieces, a straight copy of 24 KiB
  constexpr int WP = 1536 / T;  // pieces per thread (3 or 6)
  int w_dst[WP];
#pragma unroll
  for (int i = 0; i < WP; ++i) {
    const int idx = tid + T * i, p = idx / 512, row = (idx % 512) >> 2, part = idx & 3;
    w_dst[i] = (p * 128) * BK + b6_off(row, part);
  }
  f32x4 wreg[2][WP];
  auto load_w = [&](int slot, int ch) {
    ch = ch < nchunk ? ch : nchunk - 1;  // unconditional prefetch (a branch around it makes the compiler wait for it at once)
    const __bf16* src = Wc + static_cast<size_t>(ch) * (3 * 128 * BK) + tid * 8;
#pragma unroll
    for (int i = 0; i < WP; ++i) wreg[slot][i] = *reinterpret_cast<const f32x4*>(src + T * 8 * i);
  };
  auto store_w = [&](int slot, int buf) {
#pragma unroll
    for (int i = 0; i < WP; ++i) *reinterpret_cast<f32x4*>(Ws + buf * (3 * 128 * BK) + w_dst[i]) = wreg[slot][i];
  };
  // X staging: a chunk is ROWS rows x 128 bytes; thread -> rows tid / 8 and ROWS / 2 + tid / 8, 16-byte part tid % 8 (8 lanes = one
  // line); rows past M are clamped (their results are never stored).  Requested four chunks ahead (HBM), ring slots are compile-time.
  const int xa_row = tid >> 3, xa_part = tid & 7;
  const float* xsrc[2];
  int x_dst[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int lrow = xa_row + (ROWS / 2) * j;
    int row = m0 + lrow;
    row = row < M ? row : M - 1;
    xsrc[j] = X + static_cast<int64_t>(row) * ldx + 4 * xa_part;
    x_dst[j] = b6_off(lrow, xa_part >> 1) + 4 * (xa_part & 1);
  }
  f32x4 xreg[4][2];
  auto load_x = [&](int slot, int ch) {
    ch = ch < nchunk ? ch : nchunk - 1;
#pragma unroll
    for (int j = 0; j < 2; ++j) xreg[slot][j] = *reinterpret_cast<const f32x4*>(xsrc[j] + ch * BK);
  };
  typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
  auto store_x = [&](int slot, int buf) {  // fp32 -> three bf16 planes -> LDS
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      bf16x4 h, m, l;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        __bf16 hh, mm, ll;
        split3(xreg[slot][j][c], hh, mm, ll);
        h[c] = hh; m[c] = mm; l[c] = ll;
      }
      __bf16* dst = As + buf * (3 * ROWS * BK) + x_dst[j];
      *reinterpret_cast<bf16x4*>(dst) = h;
      *reinterpret_cast<bf16x4*>(dst + ROWS * BK) = m;
      *reinterpret_cast<bf16x4*>(dst + 2 * ROWS * BK) = l;
    }
  };
  f32x16 acc[2];  // wave tile: 32 rows x 64 columns = two 32 x 32 accumulators
#pragma unroll
  for (int tt = 0; tt < 2; ++tt)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[tt][r] = 0.f;

#pragma unroll
  for (int c = 0; c < 4; ++c) load_x(c, c);
  load_w(0, 0);
  load_w(1, 1);
  MEM_FENCE();
  store_w(0, 0);
  store_x(0, 0);
  load_w(0, 2);  // slot s holds chunk c with c % 2 == s: chunk 0 is staged, its slot takes chunk 2
  load_x(0, 4);
  MEM_FENCE();
  __syncthreads();
  // fragments: lane (row or column l31, k half hk) of k-step ks reads the 16-byte slot 2 ks + hk of its row
  const int fx = (l31 >> 2) & 3;
  const int a_off = (32 * rw + l31) * BK, w_off = (64 * cw + l31) * BK;
  for (int ch0 = 0; ch0 < nchunk; ch0 += 4) {  // nchunk % 4 == 0 (launcher)
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int ch = ch0 + u, buf = u & 1;
      const __bf16* al = As + buf * (3 * ROWS * BK) + a_off;
      const __bf16* wl = Ws + buf * (3 * 128 * BK) + w_off;
      constexpr int TA[6] = {1, 2, 0, 1, 0, 0}, TB[6] = {1, 0, 2, 0, 1, 0};  // (mid,mid) (lo,hi) (hi,lo) (mid,hi) (hi,mid) (hi,hi): smallest first
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const int so = 8 * ((2 * ks + hk) ^ fx);
        bf16x8 a[3], b[2][3];
#pragma unroll
        for (int p = 0; p < 3; ++p) {
          a[p] = *reinterpret_cast<const bf16x8*>(al + (p * ROWS) * BK + so);
#pragma unroll
          for (int tt = 0; tt < 2; ++tt) b[tt][p] = *reinterpret_cast<const bf16x8*>(wl + (p * 128 + 32 * tt) * BK + so);
        }
#pragma unroll
        for (int term = 0; term < 6; ++term)
#pragma unroll
          for (int tt = 0; tt < 2; ++tt)
            acc[tt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[TA[term]], b[tt][TB[term]], acc[tt], 0, 0, 0);
      }
      // stage chunk ch + 1 into the other buffers (its loads were issued two / four iterations ago), then refill the ring slots
      store_w((u + 1) & 1, buf ^ 1);
      store_x((u + 1) & 3, buf ^ 1);
      load_w((u + 1) & 1, ch + 3);
      load_x((u + 1) & 3, ch + 5);
      MEM_FENCE();
      __syncthreads();
    }
  }
  // D 32x32: column = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5); 32 lanes = 128 contiguous bytes of a row
  const bool table = bias_idx != nullptr || bias_div > 0;
#pragma unroll
  for (int tt = 0; tt < 2; ++tt) {
    const int col = 64 * cw + 32 * tt + l31;
    float bv = (bias && !table) ? bias[col] : 0.0f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = m0 + 32 * rw + (r & 3) + 8 * (r >> 2) + 4 * hk;
      if (row >= M) continue;
      if (table) {
        const int64_t bi = bias_idx ? bias_idx[row] : row / bias_div;
        bv = bias[bi * 128 + col];
      }
      float o = acc[tt][r] + bv;
      if (RELU) o = fmaxf(o, 0.f);
      Y[static_cast<int64_t>(row) * ldy + col] = o;
    }
  }
}

size_t rowgemm128_b6_scratch_bytes(int Kd) { return static_cast<size_t>(3) * 128 * Kd * sizeof(__bf16); }

bool rowgemm128_b6_ok(const float* X, int ldx, const float* Y, int ldy, int M, int Kd) {
  return Kd % (4 * BK) == 0 && ldx % 4 == 0 && ldy % 4 == 0 && (reinterpret_cast<uintptr_t>(X) & 15) == 0 &&
         (reinterpret_cast<uintptr_t>(Y) & 15) == 0 && M >= 1;
}

// W[128 x Kd] (rows ldw floats apart) -> split planes for rowgemm128_b6p (rowgemm128_b6_scratch_bytes(Kd) bytes, 16-byte aligned)
int launch_wsplit128(const float* W, int ldw, int Kd, void* planes, hipStream_t st) {
  DIFFAB_REQUIRE(W && planes && (reinterpret_cast<uintptr_t>(planes) & 15) == 0 && Kd % BK == 0, DIFFAB_ERR_ARG, "wsplit128: bad operands");
  hipLaunchKernelGGL(wsplit128_kernel, dim3((128 * Kd + 255) / 256), dim3(256), 0, st, W, ldw, Kd, static_cast<__bf16*>(planes));
  DIFFAB_LAUNCH_CHECK();
  return DIFFAB_OK;
}

// Y[M x 128] = act(X[:, 0:Kd] W[:, 0:Kd]^T + bias row) with W given as split planes (launch_wsplit128)
int launch_rowgemm128_b6p(const float* X, int ldx, const void* planes, const float* bias, const int64_t* bias_idx, int bias_div, float* Y,
                          int ldy, int M, int Kd, bool relu, hipStream_t st) {
  DIFFAB_REQUIRE(rowgemm128_b6_ok(X, ldx, Y, ldy, M, Kd) && planes && (reinterpret_cast<uintptr_t>(planes) & 15) == 0, DIFFAB_ERR_ARG,
                 "rowgemm128_b6: unsupported operands");
  const __bf16* Wc = static_cast<const __bf16*>(planes);
  // 128-row work-groups when they fill the chip (measured at 256 groups: 47 us against 53 for 64-row groups, K = 1024), 64-row
  // groups below that (B <= 128 patches of 128 residues per GPU: twice the groups); DIFFAB_B6_ROWS=64|128 forces one
  static const int rows_force = [] {
    const char* e = getenv("DIFFAB_B6_ROWS");
    return e ? atoi(e) : 0;
  }();
  const int rows_env = rows_force ? rows_force : ((M + 127) / 128 >= 256 ? 128 : 64);
#define B6_LAUNCH(RELU_, ROWS_)                                                                                                         \
  do {                                                                                                                                  \
    DIFFAB_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(rowgemm128_b6_kernel<RELU_, ROWS_>),                              \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, b6_lds_bytes<ROWS_>()));                           \
    hipLaunchKernelGGL((rowgemm128_b6_kernel<RELU_, ROWS_>), dim3((M + ROWS_ - 1) / ROWS_), dim3(ROWS_ * 4), b6_lds_bytes<ROWS_>(), st, X, \
                       ldx, Wc, bias, bias_idx, bias_div, Y, ldy, M, Kd);                                                               \
  } while (0)
  if (rows_env == 128) {
    if (relu) B6_LAUNCH(true, 128);
    else B6_LAUNCH(false, 128);
  } else {
    if (relu) B6_LAUNCH(true, 64);
    else B6_LAUNCH(false, 64);
  }
#undef B6_LAUNCH
  DIFFAB_LAUNCH_CHECK();
  return DIFFAB_OK;
}

// split + product in one call; `scratch`: rowgemm128_b6_scratch_bytes(Kd) bytes, 16-byte aligned
int launch_rowgemm128_b6(const float* X, int ldx, const float* W, int ldw, const float* bias, const int64_t* bias_idx, int bias_div, float* Y,
                         int ldy, int M, int Kd, bool relu, void* scratch, hipStream_t st) {
  if (int rc = launch_wsplit128(W, ldw, Kd, scratch, st)) return rc;
  return launch_rowgemm128_b6p(X, ldx, scratch, bias, bias_idx, bias_div, Y, ldy, M, Kd, relu, st);
}

// ================================================================== six IPA projections + local->global frames (bf16x6)
// proj[:, 0:1344] = x [Wq_s; Wk_s; Wv_s; Wq_p; Wk_p; Wv_p]^T, the three point blocks mapped to the global frame (x R + t,
// diffab_pytorch.py:324) before they are stored - the bf16x6 form of proj_frames_kernel (denoiser_fast.hip), same decomposition:
// x-stationary (a wave keeps its 32 x 128 slab of x as SPLIT A fragments: 96 VGPRs), 14 blocks of 96 output columns, MFMA n index
// permuted so that a lane ends up with three consecutive output columns (one point) per row.  The weights arrive pre-split
// (pjsplit_kernel) in stage order - stage = (block, k half): 3 planes x 96 LDS rows x 64 k, 36 KiB contiguous - and are staged
// through a ring of two LDS buffers with rows padded to 160 bytes (conflict-free ds_read_b128 for the 16-row B fragment).
namespace {
constexpr int PJ_NP = 1344, PJ_GQ = 768, PJ_GK = 960, PJ_GV = 1152;  // column map of the projection buffer (denoiser_fast.hip: ANP, OFF_*)
constexpr int PJ_B = 96, PJ_NB = PJ_NP / PJ_B, PJ_ROWS = 128;
constexpr int PJ_LD = 80;                                 // bf16 per staged row: 64 k + 16 pad (160 bytes)
constexpr int PJ_STAGE_ELEMS = 3 * PJ_B * 64;             // bf16 per stage in global memory (36 864 bytes)
constexpr int PJ_STAGE_LDS = 3 * PJ_B * PJ_LD;            // bf16 per stage in LDS (46 080 bytes)
constexpr int PJ_LDS_BYTES = 2 * PJ_STAGE_LDS * 2 + PJ_ROWS * 12 * 4;
struct __attribute__((packed, aligned(4))) pjb_f3 { float x, y, z; };
}  // namespace

// stage-ordered split weights: out[((blk * 2 + kh) * 3 + plane) * 96 + l][kk], l = 48 cw + 16 tt + j <-> output column
// 96 blk + 48 cw + 3 j + tt, k = 64 kh + kk
__global__ void pjsplit_kernel(const float* __restrict__ W0, const float* __restrict__ W1, const float* __restrict__ W2,
                               const float* __restrict__ W3, const float* __restrict__ W4, const float* __restrict__ W5,
                               __bf16* __restrict__ out) {
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;  // (output column gc, k)
  if (gid >= PJ_NP * 128) return;
  const int gc = gid >> 7, k = gid & 127;
  const float* Wp;
  int row;
  if (gc < PJ_GQ) {
    Wp = gc < 256 ? W0 : (gc < 512 ? W1 : W2);
    row = gc & 255;
  } else {
    Wp = gc < PJ_GK ? W3 : (gc < PJ_GV ? W4 : W5);
    row = gc - (gc < PJ_GK ? PJ_GQ : (gc < PJ_GV ? PJ_GK : PJ_GV));
  }
  __bf16 h, m, l;
  split3(Wp[row * 128 + k], h, m, l);
  const int blk = gc / PJ_B, rem = gc % PJ_B, cwl = rem / 48, r48 = rem % 48, j = r48 / 3, tt = r48 % 3;
  const int lrow = 48 * cwl + 16 * tt + j, kh = k >> 6, kk = k & 63;
  const size_t base = (static_cast<size_t>(blk * 2 + kh) * 3 * PJ_B + lrow) * 64 + kk;
  out[base] = h;
  out[base + PJ_B * 64] = m;
  out[base + 2 * PJ_B * 64] = l;
}

template <bool FULL>  // FULL: M is a multiple of 128, no row guards
__global__ __launch_bounds__(512) void proj_frames_b6_kernel(const float* __restrict__ X, const __bf16* __restrict__ Wc,
                                                             const float* __restrict__ R, const float* __restrict__ t,
                                                             float* __restrict__ Y, int M) {
  extern __shared__ __attribute__((aligned(16))) __bf16 pj_lds[];  // [2][3][96][PJ_LD] weights, then [128][12] frames (fp32)
  float* Rt = reinterpret_cast<float*>(pj_lds + 2 * PJ_STAGE_LDS);
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int l15 = lane & 15, g = lane >> 4, rw = wv & 3, cw = wv >> 2;
  const int m0 = blockIdx.x * PJ_ROWS;
  float* ybase = Y + static_cast<int64_t>(m0 + 32 * rw + 4 * g) * PJ_NP + 48 * cw + 3 * l15;

  // weight staging: a stage is 2304 16-byte pieces, piece idx -> (plane idx / 768, row (idx % 768) / 8, part idx % 8); thread tid takes
  // idx = tid + 512 i (i = 0..3) and 2048 + (tid & 255) - the two halves of the work-group duplicate the last 256 (no branch)
  int st_src[5], st_dst[5];
#pragma unroll
  for (int i = 0; i < 5; ++i) {
    const int idx = i < 4 ? tid + 512 * i : 2048 + (tid & 255);
    const int pl = idx / 768, rem = idx % 768;
    st_src[i] = idx * 8;
    st_dst[i] = (pl * PJ_B + (rem >> 3)) * PJ_LD + (rem & 7) * 8;
  }
  f32x4 wreg[5];
  constexpr int NSTAGE = 2 * PJ_NB;
  auto load_w = [&](int stg) {
    stg = stg < NSTAGE ? stg : NSTAGE - 1;
    const __bf16* src = Wc + static_cast<size_t>(stg) * PJ_STAGE_ELEMS;
#pragma unroll
    for (int i = 0; i < 5; ++i) wreg[i] = *reinterpret_cast<const f32x4*>(src + st_src[i]);
  };
  auto store_w = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 5; ++i) *reinterpret_cast<f32x4*>(pj_lds + buf * PJ_STAGE_LDS + st_dst[i]) = wreg[i];
  };
  load_w(0);
  // A fragments (v_mfma_f32_16x16x32_bf16: lane = row l15, k group g): a[mt][q][plane] = split(x[m0 + 32 rw + 16 mt + l15][32 q + 8 g .. + 7])
  bf16x8 a[2][4][3];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
    const int row = m0 + 32 * rw + 16 * mt + l15;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      f32x4 v0 = {0.f, 0.f, 0.f, 0.f}, v1 = v0;
      if (FULL || row < M) {
        const float* xp = X + static_cast<int64_t>(row) * 128 + 32 * q + 8 * g;
        v0 = *reinterpret_cast<const f32x4*>(xp);
        v1 = *reinterpret_cast<const f32x4*>(xp + 4);
      }
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        __bf16 hh, mm, ll;
        split3(c < 4 ? v0[c & 3] : v1[c & 3], hh, mm, ll);
        a[mt][q][0][c] = hh; a[mt][q][1][c] = mm; a[mt][q][2][c] = ll;
      }
    }
  }
  for (int idx = tid; idx < PJ_ROWS * 12; idx += 512) {
    const int row = idx / 12, cc = idx % 12, gr = m0 + row;
    float v = 0.0f;
    if (FULL || gr < M) v = cc < 9 ? R[static_cast<int64_t>(gr) * 9 + cc] : t[static_cast<int64_t>(gr) * 3 + (cc - 9)];
    Rt[idx] = v;
  }
  MEM_FENCE();
  store_w(0);
  load_w(1);
  MEM_FENCE();
  __syncthreads();

  constexpr int TA[6] = {1, 2, 0, 1, 0, 0}, TB[6] = {1, 0, 2, 0, 1, 0};  // (mid,mid) (lo,hi) (hi,lo) (mid,hi) (hi,mid) (hi,hi)
  // one (mt, r) slice of a finished block: 3 consecutive columns of one row per lane
  auto epilogue_piece = [&](const f32x4 (&acc)[2][3], int blk, int piece) {
    const int mt = piece >> 2, r = piece & 3;
    const int lrow = 32 * rw + 16 * mt + 4 * g + r;
    float vx = acc[mt][0][r], vy = acc[mt][1][r], vz = acc[mt][2][r];
    if (blk >= PJ_GQ / PJ_B) {  // point columns: local -> global frame
      const f32x4* F = reinterpret_cast<const f32x4*>(Rt + lrow * 12);
      const f32x4 f0 = F[0], f1 = F[1], f2 = F[2];  // R row-major 0..8, t 9..11
      const float ox = (vx * f0[0] + vy * f0[3] + vz * f1[2]) + f2[1];
      const float oy = (vx * f0[1] + vy * f1[0] + vz * f1[3]) + f2[2];
      const float oz = (vx * f0[2] + vy * f1[1] + vz * f2[0]) + f2[3];
      vx = ox; vy = oy; vz = oz;
    }
#ifdef PJB6_NOSTORE
    if (M < 0) {
#else
    if (FULL || m0 + lrow < M) {
#endif
      pjb_f3 o{vx, vy, vz};
      *reinterpret_cast<pjb_f3*>(ybase + (16 * mt + r) * PJ_NP + PJ_B * blk) = o;
    }
  };
  // block `blk` into `cur`; the previous block's epilogue (`prev`) is issued between the MFMA groups of the first k half, so the
  // stores drain while the matrix pipe works (all waves storing at once in front of the barrier left it idle)
  auto run_block = [&](f32x4 (&cur)[2][3], const f32x4 (&prev)[2][3], int blk) {
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int tt = 0; tt < 3; ++tt) cur[mt][tt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kh = 0; kh < 2; ++kh) {
      // stage s = 2 blk + kh is in buffer kh; stage s + 1 (loaded during stage s - 1) goes to buffer kh ^ 1, then s + 2 is requested
      // - before this stage issues any global store (a wait for loads behind stores in flight degenerates to vmcnt(0))
#ifndef PJB6_NOSTAGE
      store_w(kh ^ 1);
      load_w(2 * blk + kh + 2);
#endif
      MEM_FENCE();
      const __bf16* wl = pj_lds + kh * PJ_STAGE_LDS + (48 * cw + l15) * PJ_LD + 8 * g;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        bf16x8 b[3][3];
        constexpr int PORD[3] = {1, 0, 2};  // planes in the order the terms need them (mid, hi, lo): the first MFMAs wait for 3 reads, not 9
#pragma unroll
        for (int pi = 0; pi < 3; ++pi)
#pragma unroll
          for (int tt = 0; tt < 3; ++tt)
            b[tt][PORD[pi]] = *reinterpret_cast<const bf16x8*>(wl + (PORD[pi] * PJ_B + 16 * tt) * PJ_LD + 32 * ks);
#pragma unroll
        for (int term = 0; term < 6; ++term) {
#pragma unroll
          for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int tt = 0; tt < 3; ++tt)
#ifdef PJB6_NOMFMA
              cur[mt][tt][term & 3] += static_cast<float>(a[mt][2 * kh + ks][TA[term]][0]) + static_cast<float>(b[tt][TB[term]][1]);
#else
              cur[mt][tt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[mt][2 * kh + ks][TA[term]], b[tt][TB[term]], cur[mt][tt], 0, 0, 0);
#endif
          if (kh == 0 && blk > 0 && term >= 1 && term <= 4) {
            epilogue_piece(prev, blk - 1, 4 * ks + term - 1);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      }
      __syncthreads();
    }
  };
  f32x4 accA[2][3], accB[2][3];
  static_assert(PJ_NB % 2 == 0, "two blocks per iteration");
  for (int blk = 0; blk < PJ_NB; blk += 2) {
    run_block(accA, accB, blk);
    run_block(accB, accA, blk + 1);
  }
#pragma unroll
  for (int piece = 0; piece < 8; ++piece) epilogue_piece(accB, PJ_NB - 1, piece);
}

size_t proj_frames_b6_scratch_bytes() { return static_cast<size_t>(2 * PJ_NB) * PJ_STAGE_ELEMS * sizeof(__bf16); }

// W6 = {wq_s, wk_s, wv_s, wq_p, wk_p, wv_p} -> stage-ordered split planes (proj_frames_b6_scratch_bytes() bytes, 16-byte aligned)
int launch_pjsplit(const float* const* W6, void* planes, hipStream_t st) {
  DIFFAB_REQUIRE(planes && (reinterpret_cast<uintptr_t>(planes) & 15) == 0, DIFFAB_ERR_ARG, "pjsplit: bad operands");
  hipLaunchKernelGGL(pjsplit_kernel, dim3((PJ_NP * 128 + 255) / 256), dim3(256), 0, st, W6[0], W6[1], W6[2], W6[3], W6[4], W6[5],
                     static_cast<__bf16*>(planes));
  DIFFAB_LAUNCH_CHECK();
  return DIFFAB_OK;
}

// the six projections of one IPA layer (D = 128) into proj[rows x 1344], weights given as split planes (launch_pjsplit)
int launch_proj_frames_b6p(const float* x, const void* planes, const float* R, const float* t, float* proj, int rows, hipStream_t st) {
  DIFFAB_REQUIRE(planes && (reinterpret_cast<uintptr_t>(planes) & 15) == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0 &&
                     (reinterpret_cast<uintptr_t>(proj) & 3) == 0 && rows >= 1,
                 DIFFAB_ERR_ARG, "proj_frames_b6: unsupported operands");
  const __bf16* Wc = static_cast<const __bf16*>(planes);
  const dim3 grid((rows + PJ_ROWS - 1) / PJ_ROWS);
  if (rows % PJ_ROWS == 0) {
    DIFFAB_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(proj_frames_b6_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                         PJ_LDS_BYTES));
    hipLaunchKernelGGL(proj_frames_b6_kernel<true>, grid, dim3(512), PJ_LDS_BYTES, st, x, Wc, R, t, proj, rows);
  } else {
    DIFFAB_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(proj_frames_b6_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                         PJ_LDS_BYTES));
    hipLaunchKernelGGL(proj_frames_b6_kernel<false>, grid, dim3(512), PJ_LDS_BYTES, st, x, Wc, R, t, proj, rows);
  }
  DIFFAB_LAUNCH_CHECK();
  return DIFFAB_OK;
}

}  // namespace diffab
