// denoiser_backward.hip - backward of the hot-path training step (reference: autograd through
// Denoiser.forward diffab_pytorch.py:558-607 and the loss assembly :856-880; BASELINE config 4).
//
// Correctness-first kernels for ANY model dims: the softmax of each query row is recomputed from the saved projections
// (nothing of size K x K is stored), per-row results are written directly and per-key / per-parameter sums are added
// with float atomics (sized by bytes per Guideline 12: ~0.4 MB of adds per patch-layer).  Saved activations ("tape"):
// per layer the layer input, the projection buffer and the feature rows - the same buffers the forward kernels write.
#include <mutex>

#include "common.h"
#include "denoiser_internal.h"
#include "so3_math.h"

namespace diffab {

// ------------------------------------------------------------------ GEMMs (generic, LDS-tiled VALU)
constexpr int TB = 64, TK = 16;

// C[M,N] (+)= A[M,K] B[K,N]           (dX = dY W)
template <bool ACC>
__global__ __launch_bounds__(256) void gemm_nn_kernel(const float* __restrict__ A, int lda, const float* __restrict__ Bm, int ldb,
                                                      float* __restrict__ C, int ldc, int M, int N, int K) {
  __shared__ float As[TK][TB + 4];
  __shared__ float Bs[TK][TB + 4];
  const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
  const int m0 = blockIdx.y * TB, n0 = blockIdx.x * TB;
  float acc[4][4] = {};
  for (int k0 = 0; k0 < K; k0 += TK) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int idx = tid + r * 256;
      {  // A tile: 64 rows x 16 k
        const int row = idx >> 4, kk = idx & 15;
        const int gm = m0 + row, gk = k0 + kk;
        As[kk][row] = (gm < M && gk < K) ? A[static_cast<int64_t>(gm) * lda + gk] : 0.0f;
      }
      {  // B tile: 16 k x 64 cols
        const int kk = idx >> 6, col = idx & 63;
        const int gk = k0 + kk, gn = n0 + col;
        Bs[kk][col] = (gk < K && gn < N) ? Bm[static_cast<int64_t>(gk) * ldb + gn] : 0.0f;
      }
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < TK; ++kk) {
      float a[4], b[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) { a[i] = As[kk][ty * 4 + i]; b[i] = Bs[kk][tx * 4 + i]; }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] += a[i] * b[j];
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int gm = m0 + ty * 4 + i;
    if (gm >= M) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int gn = n0 + tx * 4 + j;
      if (gn >= N) continue;
      float* c = C + static_cast<int64_t>(gm) * ldc + gn;
      *c = ACC ? (*c + acc[i][j]) : acc[i][j];
    }
  }
}

// db[n] += sum_m dY[m][n]; optionally dY *= (act > 0) first (ReLU backward, in place)
__global__ void relu_mask_kernel(float* __restrict__ dY, const float* __restrict__ act, int64_t n) {
  const int64_t i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x;
  if (i < n && !(act[i] > 0.0f)) dY[i] = 0.0f;
}
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ dY, int ld, int M, int N, int m_chunk, float* __restrict__ db) {
  __shared__ float part[4][64];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int n = blockIdx.x * 64 + tx;
  const int lo = blockIdx.y * m_chunk, hi = min(M, lo + m_chunk);
  float s = 0.f;
  if (n < N)
    for (int m = lo + ty; m < hi; m += 4) s += dY[static_cast<int64_t>(m) * ld + n];
  part[ty][tx] = s;
  __syncthreads();
  if (ty == 0 && n < N) atomicAdd(db + n, (part[0][tx] + part[1][tx]) + (part[2][tx] + part[3][tx]));
}

struct GemmSegs {  // the six IPA projection matrices side by side: segment s covers columns [n_end[s-1], n_end[s]) of the wide operand
  float* p[6];
  int n_end[6];
  int nseg;  // 0: one plain matrix
};
__device__ __forceinline__ int seg_of(const GemmSegs& sg, int col, int* begin) {
  int s_ = 0, beg = 0;
  while (s_ + 1 < sg.nseg && col >= sg.n_end[s_]) { beg = sg.n_end[s_]; ++s_; }
  *begin = beg;
  return s_;
}
// MFMA version of C[M,N] (+)= A[M,K] B[K,N] for 16-byte-aligned operands with lda, ldb, ldc, N multiples of 4: a wave owns 16 rows
// x 64 columns.  A is read along its contiguous (k) direction - one float4 per lane = four k-steps, k order permuted identically
// for both operands - and B along ITS contiguous (n) direction: lane (l15, g) loads B[k0 + 4 g + s][n0 + 4 l15 ..], whose four
// components are column 4 l15 + c of four accumulators, so a lane ends with four consecutive columns per row (float4 stores).
typedef float nn_f32x4 __attribute__((ext_vector_type(4)));
template <bool ACC>
__global__ __launch_bounds__(256) void gemm_nn_mfma_kernel(const float* __restrict__ A, int lda, const float* __restrict__ Bm, int ldb,
                                                           float* __restrict__ C, int ldc, int M, int N, int K, GemmSegs segs,
                                                           const float* __restrict__ relu_act = nullptr, int ld_act = 0) {
  // relu_act (nullable): C = (A B) masked by relu_act > 0 - the ReLU backward of the layer below, fused into this product's epilogue
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, l15 = lane & 15, g = lane >> 4;
  const int m0 = blockIdx.y * 64 + 16 * wv, n0 = blockIdx.x * 64;
  const int arow = min(m0 + l15, M - 1);  // rows past M are clamped (never stored)
  const bool col_ok = n0 + 4 * l15 < N;
  nn_f32x4 acc[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) acc[c] = nn_f32x4{0.f, 0.f, 0.f, 0.f};
  const float* ap = A + static_cast<int64_t>(arow) * lda + 4 * g;
  const float* bp = Bm + static_cast<int64_t>(4 * g) * ldb + n0 + 4 * l15;
  for (int k0 = 0; k0 < K; k0 += 16) {
    if (segs.nseg > 0) {  // rows k of B live in the segment's own matrix (segments are multiples of 16 deep: uniform per k-group)
      int beg;
      const float* sb = segs.p[seg_of(segs, k0, &beg)];
      bp = sb + static_cast<int64_t>(4 * g - beg) * ldb + n0 + 4 * l15;
    }
    nn_f32x4 a = {0.f, 0.f, 0.f, 0.f}, b[4];
    if (k0 + 4 * g + 3 < K) {
      a = *reinterpret_cast<const nn_f32x4*>(ap + k0);
    } else {
#pragma unroll
      for (int s_ = 0; s_ < 4; ++s_)
        if (k0 + 4 * g + s_ < K) a[s_] = ap[k0 + s_];
    }
#pragma unroll
    for (int s_ = 0; s_ < 4; ++s_) {
      b[s_] = nn_f32x4{0.f, 0.f, 0.f, 0.f};
      if (col_ok && k0 + 4 * g + s_ < K) b[s_] = *reinterpret_cast<const nn_f32x4*>(bp + static_cast<int64_t>(k0 + s_) * ldb);
    }
#pragma unroll
    for (int s_ = 0; s_ < 4; ++s_)
#pragma unroll
      for (int c = 0; c < 4; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s_], b[s_][c], acc[c], 0, 0, 0);
  }
  // acc[c][r]: row 4 g + r of the wave's 16, column 4 l15 + c of the block's 64
  if (!col_ok) return;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int gm = m0 + 4 * g + r;
    if (gm >= M) continue;
    nn_f32x4* cp = reinterpret_cast<nn_f32x4*>(C + static_cast<int64_t>(gm) * ldc + n0 + 4 * l15);
    nn_f32x4 v = {acc[0][r], acc[1][r], acc[2][r], acc[3][r]};
    if (relu_act != nullptr) {
      const nn_f32x4 av = *reinterpret_cast<const nn_f32x4*>(relu_act + static_cast<int64_t>(gm) * ld_act + n0 + 4 * l15);
#pragma unroll
      for (int c = 0; c < 4; ++c) v[c] = av[c] > 0.0f ? v[c] : 0.0f;
    }
    if (ACC) v += *cp;
    *cp = v;
  }
}

static int relu_mask(float* dY, const float* act, int64_t n, hipStream_t st);
// ---- weight-gradient products beside the backward chain (round 6) -----------------------------------------------------------------
// Nothing on the backward chain reads a dW or a db, and each of the 23 weight-gradient products of a training step fills 64..700 of
// the chip's work-group slots for 20-46 us when it runs alone (0.9 ms of a 7 ms step, profiles/r05_train_kernel_stats.csv).  They go to a
// second stream of the library's own: begin() orders it behind everything the caller's stream holds so far, reads(buf) remembers that the
// side work enqueued since then reads buf, before_write(buf) makes the caller's stream wait for exactly that work before a kernel
// overwrites buf, join() before the entry point returns - so from outside the call is still ordered by ONE stream.  (Two streams of one
// process are safe since the packed-fp32 form of profiles/r06_lanes_48_63.md is out of the library.)  No side stream (creation failed, or
// the caller's stream is being captured into a graph): everything runs on the caller's stream as before.
class SideRun {
 public:
  explicit SideRun(hipStream_t main) : main_(main) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(main, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) { (void)hipGetLastError(); return; }
    st_ = &pool()[dev & 31];
    st_->mu.lock();
    if (st_->side == nullptr) {
      // lowest priority: the weight-gradient products fill the chain's gaps and tails, they must not hold CUs (64 KiB of LDS per
      // work-group) that the chain's next kernel is waiting for
      int lo = 0, hi = 0;
      if (hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess) { (void)hipGetLastError(); lo = 0; }
      bool ok = hipStreamCreateWithPriority(&st_->side, hipStreamNonBlocking, lo) == hipSuccess;
      ok = ok && hipEventCreateWithFlags(&st_->fork, hipEventDisableTiming) == hipSuccess;
      for (int i = 0; i < kEv && ok; ++i) ok = hipEventCreateWithFlags(&st_->done[i], hipEventDisableTiming) == hipSuccess;
      if (!ok) { (void)hipGetLastError(); st_->side = nullptr; st_->broken = true; }
    }
    if (st_->broken) { st_->mu.unlock(); st_ = nullptr; }
  }
  ~SideRun() {
    if (st_ == nullptr) return;
    join();
    st_->mu.unlock();
  }
  SideRun(const SideRun&) = delete;
  SideRun& operator=(const SideRun&) = delete;
  hipStream_t begin() {
    if (st_ == nullptr) return main_;
    if (hipEventRecord(st_->fork, main_) == hipSuccess) (void)hipStreamWaitEvent(st_->side, st_->fork, 0);
    return st_->side;
  }
  void reads(const void* buf) {
    if (st_ == nullptr) return;
    int slot = -1;
    for (int i = 0; i < kEv; ++i)
      if (buf_[i] == nullptr) { slot = i; break; }
    if (slot < 0) { before_write(buf_[0]); slot = 0; }  // table full: retire the oldest entry
    if (hipEventRecord(st_->done[slot], st_->side) == hipSuccess) buf_[slot] = buf;
  }
  void before_write(const void* buf) {
    if (st_ == nullptr || buf == nullptr) return;
    for (int i = 0; i < kEv; ++i)
      if (buf_[i] == buf) {
        (void)hipStreamWaitEvent(main_, st_->done[i], 0);
        buf_[i] = nullptr;
      }
  }
  void join() {
    if (st_ == nullptr) return;
    for (int i = 0; i < kEv; ++i) buf_[i] = nullptr;
    if (hipEventRecord(st_->fork, st_->side) == hipSuccess) (void)hipStreamWaitEvent(main_, st_->fork, 0);
  }

 private:
  static constexpr int kEv = 16;
  struct State {
    std::mutex mu;
    hipStream_t side = nullptr;
    hipEvent_t fork = nullptr, done[kEv] = {};
    bool broken = false;
  };
  static State* pool() {
    static State s[32];
    return s;
  }
  hipStream_t main_;
  State* st_ = nullptr;
  const void* buf_[kEv] = {};
};

static int gemm_nn(const float* A, int lda, const float* B, int ldb, float* C, int ldc, int M, int N, int K, bool acc, hipStream_t st,
                   const GemmSegs* segs = nullptr, const float* relu_act = nullptr, int ld_act = 0) {
  GemmSegs sg{};
  if (segs) sg = *segs;
  const bool mfma = lda % 4 == 0 && ldb % 4 == 0 && ldc % 4 == 0 && N % 4 == 0 && (reinterpret_cast<uintptr_t>(A) & 15) == 0 &&
                    (reinterpret_cast<uintptr_t>(B) & 15) == 0 && (reinterpret_cast<uintptr_t>(C) & 15) == 0 &&
                    (relu_act == nullptr || (ld_act % 4 == 0 && (reinterpret_cast<uintptr_t>(relu_act) & 15) == 0));
  if (relu_act != nullptr && (!mfma || acc)) {  // the fused mask exists on the aligned, non-accumulating path: otherwise two launches
    DIFFAB_REQUIRE(!acc && ld_act == N && ldc == N, DIFFAB_ERR_ARG, "gemm_nn: masked product needs dense operands off the aligned path");
    if (int rc = gemm_nn(A, lda, B, ldb, C, ldc, M, N, K, false, st, segs)) return rc;
    return relu_mask(C, relu_act, static_cast<int64_t>(M) * N, st);
  }
  dim3 grid((N + TB - 1) / TB, (M + TB - 1) / TB);
  DIFFAB_REQUIRE(sg.nseg == 0 || mfma, DIFFAB_ERR_ARG, "gemm_nn: segmented operand needs the aligned path");
  if (mfma) {
    if (acc) hipLaunchKernelGGL(gemm_nn_mfma_kernel<true>, grid, dim3(256), 0, st, A, lda, B, ldb, C, ldc, M, N, K, sg, nullptr, 0);
    else hipLaunchKernelGGL(gemm_nn_mfma_kernel<false>, grid, dim3(256), 0, st, A, lda, B, ldb, C, ldc, M, N, K, sg, relu_act, ld_act);
  } else {
    if (acc) hipLaunchKernelGGL(gemm_nn_kernel<true>, grid, dim3(256), 0, st, A, lda, B, ldb, C, ldc, M, N, K);
    else hipLaunchKernelGGL(gemm_nn_kernel<false>, grid, dim3(256), 0, st, A, lda, B, ldb, C, ldc, M, N, K);
  }
  DIFFAB_LAUNCH_CHECK();
  return DIFFAB_OK;
}
// MFMA version of the weight-gradient product for 16-byte-aligned operands and N1, N2 multiples of 64: one wave per 64 x 64 tile
// of C and per chunk of M.  Both operands are read along their contiguous (n) direction: lane (l15, g) loads the float4
// A[m + g][a0 + 4 l15 ..] and B[m + g][b0 + 4 l15 ..]; component c of the A vector is row 4 l15 + c of the tile and component c' of
// the B vector column 4 l15 + c', so the 16 MFMAs (c, c') of a step fill 16 accumulators whose (row, col) = (l15, l15') element
// is C[4 l15 + c][4 l15' + c'].  Two 1 KiB loads feed 16 MFMAs - no LDS staging needed.
typedef float tn_f32x4 __attribute__((ext_vector_type(4)));
template <bool ALIGNED>  // false: any N1, N2, lda, ldb (per-element guarded loads and stores); true: the vector path described above
__global__ __launch_bounds__(256) void gemm_tn_mfma_kernel(const float* __restrict__ A, int lda, const float* __restrict__ Bm, int ldb,
                                                           float* __restrict__ C, int ldc, int M, int m_chunk, int N1, int N2,
                                                           GemmSegs segs) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, l15 = lane & 15, g = lane >> 4;
  const int a0 = blockIdx.y * 64, b0 = blockIdx.x * 64;
  const int m_lo = min((blockIdx.z * 4 + wv) * m_chunk, M), m_hi = min(M, m_lo + m_chunk);  // past M: an empty range (zeros)
  tn_f32x4 acc[4][4];
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int d = 0; d < 4; ++d) acc[c][d] = tn_f32x4{0.f, 0.f, 0.f, 0.f};
  const float* ap = A + static_cast<int64_t>(m_lo + g) * lda + a0 + 4 * l15;
  const float* bp = Bm + static_cast<int64_t>(m_lo + g) * ldb + b0 + 4 * l15;
  constexpr int U = 4;  // steps (of 4 rows of M) in flight
  tn_f32x4 va[U], vb[U];
  auto load = [&](int u, int m) {
    const bool ok = m + g < m_hi;
    if (ALIGNED) {
      va[u] = ok ? *reinterpret_cast<const tn_f32x4*>(ap + static_cast<int64_t>(m - m_lo) * lda) : tn_f32x4{0.f, 0.f, 0.f, 0.f};
      vb[u] = ok ? *reinterpret_cast<const tn_f32x4*>(bp + static_cast<int64_t>(m - m_lo) * ldb) : tn_f32x4{0.f, 0.f, 0.f, 0.f};
    } else {
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        va[u][c] = (ok && a0 + 4 * l15 + c < N1) ? ap[static_cast<int64_t>(m - m_lo) * lda + c] : 0.0f;
        vb[u][c] = (ok && b0 + 4 * l15 + c < N2) ? bp[static_cast<int64_t>(m - m_lo) * ldb + c] : 0.0f;
      }
    }
  };
#pragma unroll
  for (int u = 0; u < U; ++u) load(u, m_lo + 4 * u);
  for (int m = m_lo; m < m_hi; m += 4 * U) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const tn_f32x4 xa = va[u], xb = vb[u];
      load(u, m + 4 * (U + u));  // rows past m_hi load zeros
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int d = 0; d < 4; ++d) acc[c][d] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[c], xb[d], acc[c][d], 0, 0, 0);
    }
  }
  // The four waves of a work-group hold partial sums of the SAME tile (consecutive chunks of M): they are added through LDS and
  // wave w finishes accumulator row group c = w, so the tile costs one set of atomics per work-group instead of four.
  __shared__ float red[4 * 64 * 64];  // [wave][value (c, d, r)][lane]
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int d = 0; d < 4; ++d)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[(wv * 64 + (c * 4 + d) * 4 + r) * 64 + lane] = acc[c][d][r];
  __syncthreads();
  // D layout: acc[c][d][r] = tile row 4 (4 g + r) + c, tile column 4 l15 + d
  const int c = wv;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int crow_i = a0 + 4 * (4 * g + r) + c;
    if (!ALIGNED && crow_i >= N1) continue;
    float* cbase = C;
    int crow_l = crow_i;
    if (segs.nseg > 0) {  // rows of C live in the segment's own matrix (segments are multiples of 64 wide: uniform per tile)
      int beg;
      cbase = segs.p[seg_of(segs, a0, &beg)];
      crow_l = crow_i - beg;
    }
    float* crow = cbase + static_cast<int64_t>(crow_l) * ldc + b0 + 4 * l15;
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      const int v = (c * 4 + d) * 4 + r;
      const float sum = (red[(0 * 64 + v) * 64 + lane] + red[(1 * 64 + v) * 64 + lane]) + (red[(2 * 64 + v) * 64 + lane] + red[(3 * 64 + v) * 64 + lane]);
      if (ALIGNED || b0 + 4 * l15 + d < N2) atomicAdd(crow + d, sum);
    }
  }
}

// db / db_done: the bias gradient db[N1] += column sums of A rides along when the bf16x6 kernel takes the product (*db_done = true)
static int gemm_tn(const float* A, int lda, const float* B, int ldb, float* C, int ldc, int M, int N1, int N2, hipStream_t st,
                   const GemmSegs* segs = nullptr, float* db = nullptr, bool* db_done = nullptr) {
  GemmSegs sg{};
  if (segs) sg = *segs;
  if (use_b6_gemm() && gemm_tn_b6_ok(A, lda, B, ldb, M, N1, N2) && M >= 128) {
    bool seg_ok = sg.nseg == 0 || (lda % 4 == 0 && N1 % 16 == 0 && (reinterpret_cast<uintptr_t>(A) & 15) == 0);
    for (int i = 0; i < sg.nseg; ++i) seg_ok = seg_ok && sg.n_end[i] % 64 == 0;
    if (seg_ok) {
      if (db_done) *db_done = db != nullptr;
      if (tn_h3_enabled()) return launch_gemm_tn_h3(A, lda, B, ldb, C, ldc, M, N1, N2, db_done ? db : nullptr, sg.p, sg.n_end, sg.nseg, st);
      return launch_gemm_tn_b6(A, lda, B, ldb, C, ldc, M, N1, N2, db_done ? db : nullptr, sg.p, sg.n_end, sg.nseg, st);
    }
  }
  const bool aligned = N1 % 64 == 0 && N2 % 64 == 0 && lda % 4 == 0 && ldb % 4 == 0 && (reinterpret_cast<uintptr_t>(A) & 15) == 0 &&
                       (reinterpret_cast<uintptr_t>(B) & 15) == 0;
  // enough (tile, M-chunk) waves to fill the chip: ~2048 waves, chunks of at least 256 rows
  const int t1 = (N1 + 63) / 64, t2 = (N2 + 63) / 64, tiles = t1 * t2;
  int splits = (2048 + tiles - 1) / tiles;  // M chunks wanted
  int m_chunk = (M + splits - 1) / splits;
  m_chunk = ((max(m_chunk, 256) + 15) / 16) * 16;
  const int nchunks = (M + m_chunk - 1) / m_chunk;
  dim3 grid(t2, t1, (nchunks + 3) / 4);
  DIFFAB_REQUIRE(sg.nseg == 0 || aligned, DIFFAB_ERR_ARG, "gemm_tn: segmented output needs the aligned path");
  if (aligned) hipLaunchKernelGGL(gemm_tn_mfma_kernel<true>, grid, dim3(256), 0, st, A, lda, B, ldb, C, ldc, M, m_chunk, N1, N2, sg);
  else hipLaunchKernelGGL(gemm_tn_mfma_kernel<false>, grid, dim3(256), 0, st, A, lda, B, ldb, C, ldc, M, m_chunk, N1, N2, sg);
  DIFFAB_LAUNCH_CHECK();
  return DIFFAB_OK;
}
static int colsum(const float* dY, int ld, int M, int N, float* db, hipStream_t st) {
  const int m_chunk = 128;
  hipLaunchKernelGGL(colsum_kernel, dim3((N + 63) / 64, (M + m_chunk - 1) / m_chunk), dim3(256), 0, st, dY, ld, M, N, m_chunk, db);
  DIFFAB_LAUNCH_CHECK();
  return DIFFAB_OK;
}
static int relu_mask(float* dY, const float* act, int64_t n, hipStream_t st) {
  hipLaunchKernelGGL(relu_mask_kernel, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, st, dY, act, n);
  DIFFAB_LAUNCH_CHECK();
  return DIFFAB_OK;
}

// Y = act(X W^T + b) backward: dW += dY^T X, db += colsum dY, dX (+)= dY W.   dY must already carry the activation mask.
// side (nullable): the weight / bias gradients run on its stream, beside the chain; the caller must side->before_write(dY) ahead of the
// next kernel that overwrites dY
static int linear_bwd(const float* dY, int ldy, const float* X, int ldx, const float* W, float* dW, float* db, float* dX, int lddx, int M,
                      int N, int Kd, bool acc_dx, hipStream_t st, SideRun* side = nullptr) {
  bool db_done = false;
  hipStream_t sw = side ? side->begin() : st;
  if (int rc = gemm_tn(dY, ldy, X, ldx, dW, Kd, M, N, Kd, sw, nullptr, db, &db_done)) return rc;
  if (db && !db_done)
    if (int rc = colsum(dY, ldy, M, N, db, sw)) return rc;
  if (side) side->reads(dY);
  if (dX) return gemm_nn(dY, ldy, W, Kd, dX, lddx, M, Kd, N, acc_dx, st);
  return DIFFAB_OK;
}

// shared with context_kernels.hip (the encode_context backward is the same chain of linear / ReLU backward steps)
int bwd_linear(const float* dY, int ldy, const float* X, int ldx, const float* W, float* dW, float* db, float* dX, int lddx, int M, int N,
               int Kd, bool acc_dx, hipStream_t st) {
  return linear_bwd(dY, ldy, X, ldx, W, dW, db, dX, lddx, M, N, Kd, acc_dx, st);
}
int bwd_relu_mask(float* dY, const float* act, int64_t n, hipStream_t st) { return relu_mask(dY, act, n, st); }
// dX = (dY W) masked by relu_act > 0 (the ReLU below the layer), dW += dY^T X, db += colsum dY: bwd_linear + bwd_relu_mask in one product
int bwd_linear_masked(const float* dY, int ldy, const float* X, int ldx, const float* W, float* dW, float* db, float* dX, int lddx, int M,
                      int N, int Kd, const float* relu_act, hipStream_t st) {
  bool db_done = false;
  if (int rc = gemm_tn(dY, ldy, X, ldx, dW, Kd, M, N, Kd, st, nullptr, db, &db_done)) return rc;
  if (db && !db_done)
    if (int rc = colsum(dY, ldy, M, N, db, st)) return rc;
  return gemm_nn(dY, ldy, W, Kd, dX, lddx, M, Kd, N, false, st, nullptr, relu_act, lddx);
}
int bwd_gemm_nn_masked(const float* A, int lda, const float* B, int ldb, float* C, int ldc, int M, int N, int K, const float* relu_act,
                       hipStream_t st) {
  return gemm_nn(A, lda, B, ldb, C, ldc, M, N, K, false, st, nullptr, relu_act, ldc);
}
// ---- C_p[64 x 64] += A_p^T B_p for up to four operand pairs of [M x 64] rows (M very large): the PairEmbedding backward's weight
// gradients, M = 491 k pair rows per chunk.  The 128 x 128-tile bf16x6 kernel spends 196 us on each (three quarters of its tile and all
// of its split-plane staging wasted on a 64 x 64 output: 1.3 TB/s); here a work-group streams 32-row slabs of both operands through
// LDS and every thread keeps a 4 x 4 block of the product in registers (plain fp32 FMAs: 8 GFLOP per product is nothing), partial
// sums joined by atomics; db_p (nullable) += column sums of A_p.
struct Tn64Set {
  const float* A[4];
  const float* B[4];
  float* C[4];
  float* db[4];
  int ldc[4];
  int n;
};
__global__ __launch_bounds__(256) void tn64_kernel(Tn64Set set, int M, int rows_per_group) {
  __shared__ __attribute__((aligned(16))) float As[32 * 64], Bs[32 * 64];
  const int pi = blockIdx.y;
  const float* __restrict__ A = pi == 0 ? set.A[0] : pi == 1 ? set.A[1] : pi == 2 ? set.A[2] : set.A[3];
  const float* __restrict__ B = pi == 0 ? set.B[0] : pi == 1 ? set.B[1] : pi == 2 ? set.B[2] : set.B[3];
  float* __restrict__ C = pi == 0 ? set.C[0] : pi == 1 ? set.C[1] : pi == 2 ? set.C[2] : set.C[3];
  float* __restrict__ db = pi == 0 ? set.db[0] : pi == 1 ? set.db[1] : pi == 2 ? set.db[2] : set.db[3];
  const int ldc = pi == 0 ? set.ldc[0] : pi == 1 ? set.ldc[1] : pi == 2 ? set.ldc[2] : set.ldc[3];
  const int tid = threadIdx.x, ty = tid >> 4, tx = tid & 15;  // thread (ty, tx): rows 4 ty .. of C (columns of A), columns 4 tx .. of C
  const int64_t m_lo = static_cast<int64_t>(blockIdx.x) * rows_per_group;
  const int64_t m_hi = m_lo + rows_per_group < M ? m_lo + rows_per_group : M;
  float acc[4][4], cs[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
  typedef float v4 __attribute__((ext_vector_type(4)));
  for (int64_t m = m_lo; m < m_hi; m += 32) {
    // slab of 32 rows x 64 floats = 512 float4 per operand: two per thread (rows past m_hi are zero)
#pragma unroll
    for (int rep = 0; rep < 2; ++rep) {
      const int idx = tid + 256 * rep, r = idx >> 4, c4 = idx & 15;
      v4 va = {0.f, 0.f, 0.f, 0.f}, vb = va;
      if (m + r < m_hi) {
        va = *reinterpret_cast<const v4*>(A + (m + r) * 64 + 4 * c4);
        vb = *reinterpret_cast<const v4*>(B + (m + r) * 64 + 4 * c4);
      }
      *reinterpret_cast<v4*>(As + r * 64 + 4 * c4) = va;
      *reinterpret_cast<v4*>(Bs + r * 64 + 4 * c4) = vb;
    }
    __syncthreads();
#pragma unroll 8
    for (int r = 0; r < 32; ++r) {
      const v4 a = *reinterpret_cast<const v4*>(As + r * 64 + 4 * ty);
      const v4 b = *reinterpret_cast<const v4*>(Bs + r * 64 + 4 * tx);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        cs[i] += a[i];
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] += a[i] * b[j];
      }
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
#pragma unroll
    for (int j = 0; j < 4; ++j) atomicAdd(C + static_cast<int64_t>(4 * ty + i) * ldc + 4 * tx + j, acc[i][j]);
    if (db != nullptr && tx == 0) atomicAdd(db + 4 * ty + i, cs[i]);
  }
}
// n <= 4 products C_p[64 x 64] (rows ldc_p apart) += A_p^T B_p, A_p, B_p dense [M x 64], 16-byte aligned; db_p nullable
int bwd_tn64_set(int n, const float* const* A, const float* const* B, float* const* C, const int* ldc, float* const* db, int64_t M,
                 hipStream_t st) {
  DIFFAB_REQUIRE(n >= 1 && n <= 4 && M >= 1 && M < (1ll << 31), DIFFAB_ERR_ARG, "tn64: bad arguments");
  Tn64Set set{};
  set.n = n;
  for (int i = 0; i < n; ++i) {
    DIFFAB_REQUIRE(A[i] && B[i] && C[i] && (reinterpret_cast<uintptr_t>(A[i]) & 15) == 0 && (reinterpret_cast<uintptr_t>(B[i]) & 15) == 0,
                   DIFFAB_ERR_ARG, "tn64: null / misaligned operand");
    set.A[i] = A[i]; set.B[i] = B[i]; set.C[i] = C[i]; set.ldc[i] = ldc[i]; set.db[i] = db ? db[i] : nullptr;
  }
  const int groups = 512;  // 512 x n work-groups, two or more per CU: the slab loads of one hide behind the FMAs of the other
  int rpg = static_cast<int>((M + groups - 1) / groups);
  rpg = (rpg + 31) / 32 * 32;
  hipLaunchKernelGGL(tn64_kernel, dim3(static_cast<unsigned>((M + rpg - 1) / rpg), n), dim3(256), 0, st, set, static_cast<int>(M), rpg);
  DIFFAB_LAUNCH_CHECK();
  return DIFFAB_OK;
}
int bwd_gemm_tn(const float* A, int lda, const float* B, int ldb, float* C, int ldc, int M, int N1, int N2, float* db, hipStream_t st) {
  bool db_done = false;
  if (int rc = gemm_tn(A, lda, B, ldb, C, ldc, M, N1, N2, st, nullptr, db, &db_done)) return rc;
  if (db && !db_done) return colsum(A, lda, M, N1, db, st);
  return DIFFAB_OK;
}
int bwd_gemm_nn(const float* A, int lda, const float* B, int ldb, float* C, int ldc, int M, int N, int K, bool acc, hipStream_t st) {
  return gemm_nn(A, lda, B, ldb, C, ldc, M, N, K, acc, st);
}

// ------------------------------------------------------------------ loss backward (per residue)
// upstream scalars g = (g_seq, g_x, g_o); N = #masked residues (device scalar computed here by one block)
__global__ void count_mask_kernel(const uint8_t* __restrict__ gm, const uint8_t* __restrict__ rm, int64_t n, float* __restrict__ out) {
  __shared__ float red[1024];
  float c = 0.f;
  for (int64_t i = threadIdx.x; i < n; i += blockDim.x) c += (gm[i] && rm[i]) ? 1.0f : 0.0f;
  red[threadIdx.x] = c;
  __syncthreads();
  for (int s = blockDim.x / 2; s > 0; s >>= 1) {
    if (static_cast<int>(threadIdx.x) < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = red[0];
}

// ------------------------------------------------------------------ attention backward in two atomic-free passes
// Pass 1, one work-group per (patch, query residue i): recompute the row's softmax from the saved projections, form dA and
// the logit gradient, write this row's own gradients (q_s, q points, de[i], and dog = d o_g) directly, and store the
// probabilities A and g = scale_t * dlogit TRANSPOSED ([b][h][j][i]) for pass 2.  dWb / dgamma are per-row partial sums added
// with one atomic each (H*C + H per row).
// Pass 2, one work-group per (patch, key residue j): reduce over the query rows for k_s, k points, v_s, v points.
// proj / dproj rows: [q_s | k_s | v_s | gq | gk | gv] (points in the global frame; dproj holds gradients w.r.t. the GLOBAL points,
// points_bwd_kernel turns them into local-point gradients afterwards).
__global__ __launch_bounds__(256) void ipa_attn_bwd_rows_kernel(const float* __restrict__ proj, const float* __restrict__ e,
                                                                const float* __restrict__ R, const float* __restrict__ Wb,
                                                                const float* __restrict__ gamma, const float* __restrict__ feat,
                                                                const float* __restrict__ dfeat, float* __restrict__ dproj,
                                                                float* __restrict__ de, float* __restrict__ dWb, float* __restrict__ dgamma,
                                                                float* __restrict__ At, float* __restrict__ Gt, float* __restrict__ dogbuf,
                                                                float* __restrict__ wb_part, int K, int C, int H, int DS, int PQ, int PV,
                                                                int vec) {
  // vec: DS, C, 3 PQ, 3 PV are multiples of 4 and e is 16-byte aligned: the (h, j) dot products run on 8 lanes x float4 (every
  // load instruction covers 128-byte pieces of a key row) instead of one lane per (h, j) walking its own row (64 lines per load
  // instruction: the texture-address unit, not the math, set the 10 ms this kernel used to take per layer).
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int b = blockIdx.x / K, i = blockIdx.x % K;
  const int NP = 3 * H * DS + 2 * H * PQ * 3 + H * PV * 3;
  const int F = H * DS + H * C + H * PV * 3 + H * PV;
  const int off_ks = H * DS, off_vs = 2 * H * DS, off_gq = 3 * H * DS, off_gk = off_gq + H * PQ * 3, off_gv = off_gk + H * PQ * 3;
  const int n_os = H * DS, n_oe = H * C, n_og = H * PV * 3;
  float* attn = smem;                        // H*K   probabilities
  float* gl = attn + H * K;                  // H*K   dA, then scale_t * dlogit
  float* d2 = gl + H * K;                    // H*K   squared point distances
  float* qrow = d2 + H * K;                  // H*DS + H*PQ*3
  float* dog = qrow + H * DS + H * PQ * 3;   // H*PV*3  gradient w.r.t. the global value-point sums
  float* red = dog + H * PV * 3;             // H  row sums
  float* dfl = red + H;                      // F  this row's feature gradient
  const int64_t row_i = static_cast<int64_t>(b) * K + i;
  const float* prow = proj + row_i * NP;
  const float* frow = feat + row_i * F;
  const float* dfrow = dfeat + row_i * F;
  const float* erow = e + row_i * K * C;
  float* derow = de ? de + row_i * K * C : nullptr;
  const float* Rr = R + row_i * 9;
  for (int d = threadIdx.x; d < H * DS; d += blockDim.x) qrow[d] = prow[d];
  for (int d = threadIdx.x; d < H * PQ * 3; d += blockDim.x) qrow[H * DS + d] = prow[off_gq + d];
  for (int d = threadIdx.x; d < F; d += blockDim.x) dfl[d] = dfrow[d];
  // value-point sums: o_l = (o_g - t) R^T, o_n = |o_l|  ->  d o_g[k] = sum_c (do_l[c] + do_n o_l[c]/o_n) R[c][k]
  for (int hp = threadIdx.x; hp < H * PV; hp += blockDim.x) {
    const float on = frow[n_os + n_oe + n_og + hp], don = dfrow[n_os + n_oe + n_og + hp];
    float dl[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float ol = frow[n_os + n_oe + hp * 3 + c];
      dl[c] = dfrow[n_os + n_oe + hp * 3 + c] + (on > 0.0f ? don * ol / on : 0.0f);
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const float v = dl[0] * Rr[0 * 3 + k] + dl[1] * Rr[1 * 3 + k] + dl[2] * Rr[2 * 3 + k];
      dog[hp * 3 + k] = v;
      dogbuf[row_i * n_og + hp * 3 + k] = v;
    }
  }
  __syncthreads();
  const float scale_s = 1.0f / sqrtf(static_cast<float>(DS));
  const float scale_p = -0.5f / sqrtf(4.5f * PQ);
  const float scale_t = 1.0f / sqrtf(C > 0 ? 3.0f : 2.0f);  // 3 independent logits with the pair bias, 2 without (C == 0)
  // ---- recompute logits
  typedef float v4 __attribute__((ext_vector_type(4)));
  const int lo = threadIdx.x & 7, oct = threadIdx.x >> 3, noct = blockDim.x >> 3;
  if (vec) {
    for (int idx = oct; idx < H * K; idx += noct) {
      const int h = idx / K, j = idx % K;
      const float* krow = proj + (static_cast<int64_t>(b) * K + j) * NP;
      float ls = 0.f, lb = 0.f, lp = 0.f;
      for (int d = 4 * lo; d < DS; d += 32) {
        const v4 kv = *reinterpret_cast<const v4*>(krow + off_ks + h * DS + d), qv = *reinterpret_cast<const v4*>(qrow + h * DS + d);
        ls += (qv[0] * kv[0] + qv[1] * kv[1]) + (qv[2] * kv[2] + qv[3] * kv[3]);
      }
      for (int c = 4 * lo; c < C; c += 32) {
        const v4 ev = *reinterpret_cast<const v4*>(erow + static_cast<int64_t>(j) * C + c), wv = *reinterpret_cast<const v4*>(Wb + h * C + c);
        lb += (ev[0] * wv[0] + ev[1] * wv[1]) + (ev[2] * wv[2] + ev[3] * wv[3]);
      }
      for (int p = 4 * lo; p < PQ * 3; p += 32) {
        const v4 kv = *reinterpret_cast<const v4*>(krow + off_gk + h * PQ * 3 + p);
        const v4 qv = *reinterpret_cast<const v4*>(qrow + H * DS + h * PQ * 3 + p);
        const v4 dd = qv - kv;
        lp += (dd[0] * dd[0] + dd[1] * dd[1]) + (dd[2] * dd[2] + dd[3] * dd[3]);
      }
#pragma unroll
      for (int o = 1; o < 8; o <<= 1) {
        ls += __shfl_xor(ls, o);
        lb += __shfl_xor(lb, o);
        lp += __shfl_xor(lp, o);
      }
      if (lo == 0) {
        d2[idx] = lp;
        attn[idx] = scale_t * ((ls * scale_s + lb) + (scale_p * gamma[h]) * lp);
      }
    }
  } else {
    for (int idx = threadIdx.x; idx < H * K; idx += blockDim.x) {
      const int h = idx / K, j = idx % K;
      const float* krow = proj + (static_cast<int64_t>(b) * K + j) * NP;
      float ls = 0.f, lb = 0.f, lp = 0.f;
      for (int d = 0; d < DS; ++d) ls += qrow[h * DS + d] * krow[off_ks + h * DS + d];
      for (int c = 0; c < C; ++c) lb += erow[static_cast<int64_t>(j) * C + c] * Wb[h * C + c];
      for (int p = 0; p < PQ * 3; ++p) {
        const float dd = qrow[H * DS + h * PQ * 3 + p] - krow[off_gk + h * PQ * 3 + p];
        lp += dd * dd;
      }
      d2[idx] = lp;
      attn[idx] = scale_t * ((ls * scale_s + lb) + (scale_p * gamma[h]) * lp);
    }
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwave = blockDim.x >> 6;
  for (int h = wave; h < H; h += nwave) {
    float m = -INFINITY;
    for (int j = lane; j < K; j += 64) m = fmaxf(m, attn[h * K + j]);
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    float s = 0.f;
    for (int j = lane; j < K; j += 64) {
      const float ex = expf(attn[h * K + j] - m);
      attn[h * K + j] = ex;
      s += ex;
    }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    const float inv = 1.0f / s;
    for (int j = lane; j < K; j += 64) attn[h * K + j] *= inv;
  }
  __syncthreads();
  // ---- dA[h][j] = do_s . v_s[j] + do_e . e[i][j] + do_g . gv[j]
  if (vec) {
    for (int idx = oct; idx < H * K; idx += noct) {
      const int h = idx / K, j = idx % K;
      const float* vrow = proj + (static_cast<int64_t>(b) * K + j) * NP;
      float sacc = 0.f;
      for (int d = 4 * lo; d < DS; d += 32) {
        const v4 a = *reinterpret_cast<const v4*>(dfl + h * DS + d), v = *reinterpret_cast<const v4*>(vrow + off_vs + h * DS + d);
        sacc += (a[0] * v[0] + a[1] * v[1]) + (a[2] * v[2] + a[3] * v[3]);
      }
      for (int c = 4 * lo; c < C; c += 32) {
        const v4 a = *reinterpret_cast<const v4*>(dfl + n_os + h * C + c), v = *reinterpret_cast<const v4*>(erow + static_cast<int64_t>(j) * C + c);
        sacc += (a[0] * v[0] + a[1] * v[1]) + (a[2] * v[2] + a[3] * v[3]);
      }
      for (int p = 4 * lo; p < PV * 3; p += 32) {
        const v4 a = *reinterpret_cast<const v4*>(dog + h * PV * 3 + p), v = *reinterpret_cast<const v4*>(vrow + off_gv + h * PV * 3 + p);
        sacc += (a[0] * v[0] + a[1] * v[1]) + (a[2] * v[2] + a[3] * v[3]);
      }
#pragma unroll
      for (int o = 1; o < 8; o <<= 1) sacc += __shfl_xor(sacc, o);
      if (lo == 0) gl[idx] = sacc;
    }
  } else {
    for (int idx = threadIdx.x; idx < H * K; idx += blockDim.x) {
      const int h = idx / K, j = idx % K;
      const float* vrow = proj + (static_cast<int64_t>(b) * K + j) * NP;
      float sacc = 0.f;
      for (int d = 0; d < DS; ++d) sacc += dfl[h * DS + d] * vrow[off_vs + h * DS + d];
      for (int c = 0; c < C; ++c) sacc += dfl[n_os + h * C + c] * erow[static_cast<int64_t>(j) * C + c];
      for (int p = 0; p < PV * 3; ++p) sacc += dog[h * PV * 3 + p] * vrow[off_gv + h * PV * 3 + p];
      gl[idx] = sacc;
    }
  }
  __syncthreads();
  // ---- softmax backward: dlogit = A (dA - sum_j A dA); keep g = scale_t * dlogit
  for (int h = wave; h < H; h += nwave) {
    float s = 0.f;
    for (int j = lane; j < K; j += 64) s += attn[h * K + j] * gl[h * K + j];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) red[h] = s;
  }
  __syncthreads();
  for (int idx = threadIdx.x; idx < H * K; idx += blockDim.x) {
    const int h = idx / K, j = idx % K;
    const float g = scale_t * attn[idx] * (gl[idx] - red[h]);
    gl[idx] = g;
    const int64_t o = ((static_cast<int64_t>(b) * H + h) * K + j) * K + i;  // transposed: [b][h][j][i]
    At[o] = attn[idx];
    Gt[o] = g;
  }
  __syncthreads();
  for (int h = wave; h < H; h += nwave) {  // dgamma_h += sum_j g scale_p d2
    float s = 0.f;
    for (int j = lane; j < K; j += 64) s += gl[h * K + j] * scale_p * d2[h * K + j];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) wb_part[row_i * (H * C + H) + H * C + h] = s;  // per-row partial; summed over rows afterwards (no contended atomics)
  }
  // ---- query-side gradients (this row only)
  float* dqrow = dproj + row_i * NP;
  for (int o = threadIdx.x; o < H * DS + H * PQ * 3; o += blockDim.x) {
    float acc = 0.f;
    if (o < H * DS) {
      const int h = o / DS;
      for (int j = 0; j < K; ++j) acc += gl[h * K + j] * proj[(static_cast<int64_t>(b) * K + j) * NP + off_ks + o];
      dqrow[o] = acc * scale_s;
    } else {
      const int oo = o - H * DS, h = oo / (PQ * 3);
      const float cp = 2.0f * scale_p * gamma[h];
      for (int j = 0; j < K; ++j)
        acc += gl[h * K + j] * (qrow[H * DS + oo] - proj[(static_cast<int64_t>(b) * K + j) * NP + off_gk + oo]);
      dqrow[off_gq + oo] = cp * acc;
    }
  }
  // ---- pair embedding: de[i][j][c] += sum_h (A do_e[h][c] + g Wb[h][c]);   dWb[h][c] += sum_j g e[i][j][c]
  if (derow) {
    for (int idx = threadIdx.x; idx < K * C; idx += blockDim.x) {
      const int j = idx / C, c = idx % C;
      float s = 0.f;
      for (int h = 0; h < H; ++h) s += attn[h * K + j] * dfl[n_os + h * C + c] + gl[h * K + j] * Wb[h * C + c];
      derow[idx] += s;
    }
  }
  for (int o = threadIdx.x; o < H * C; o += blockDim.x) {
    const int h = o / C, c = o % C;
    float s = 0.f;
    for (int j = 0; j < K; ++j) s += gl[h * K + j] * erow[static_cast<int64_t>(j) * C + c];
    wb_part[row_i * (H * C + H) + o] = s;
  }
}

// Same pass for RR consecutive query rows per work-group (vector path only: DS, C, 3 PQ, 3 PV multiples of 4, K % RR == 0).
// The one-row kernel above re-reads the whole key side of the patch (k_s, k_pts, v_s, v_pts: 448 KiB at the benchmark geometry,
// twice for k) from L2 for every query row - 0.9 MB per row, 15 GB per layer at B = 128 - and that traffic, not the arithmetic,
// sets its 2.3 ms.  Here every key-side value is loaded once and used for RR rows; the per-row state lives in RR LDS slots.
// HAVE_P: the probabilities and the squared point distances come from global memory (Pn, D2g: [b][h][i][j], written by
// launch_attention_probs - the MFMA forward kernels re-run), instead of being recomputed here on the VALU.
// MODE 0: everything here on the VALU.  1 (HAVE_P): see above.  2 (HAVE_G): g itself comes from global memory too (dAkv then holds g,
// written by ipa_pair_stream_bwd_kernel together with the d gamma / d w_bias partials): only the transposed copies and d e remain.
template <int RR, int MODE>
__global__ __launch_bounds__(512) void ipa_attn_bwd_rows_mr_kernel(const float* __restrict__ proj, const float* __restrict__ e,
                                                                   const float* __restrict__ R, const float* __restrict__ Wb,
                                                                   const float* __restrict__ gamma, const float* __restrict__ feat,
                                                                   const float* __restrict__ dfeat, float* __restrict__ dproj,
                                                                   float* __restrict__ de, float* __restrict__ At, float* __restrict__ Gt,
                                                                   float* __restrict__ dogbuf, float* __restrict__ wb_part, int K, int C,
                                                                   int H, int DS, int PQ, int PV, float* __restrict__ Pn,
                                                                   const float* __restrict__ D2g, const float* __restrict__ dAkv) {
  // HAVE_P additionally: the key/value part of dA comes from dAkv (ipa_attn_bwd_dakv_mfma_kernel), g is written back over the
  // probabilities in Pn ([b][h][i][j], for the query-side MFMA pass) and the query-side gradients are not computed here.
  extern __shared__ __attribute__((aligned(16))) float smem[];
  typedef float v4 __attribute__((ext_vector_type(4)));
  constexpr bool HAVE_P = MODE >= 1, HAVE_G = MODE == 2;
  const int nblk = K / RR;
  const int b = blockIdx.x / nblk, i0 = (blockIdx.x % nblk) * RR;
  const int NP = 3 * H * DS + 2 * H * PQ * 3 + H * PV * 3;
  const int F = H * DS + H * C + H * PV * 3 + H * PV;
  const int off_ks = H * DS, off_vs = 2 * H * DS, off_gq = 3 * H * DS, off_gk = off_gq + H * PQ * 3, off_gv = off_gk + H * PQ * 3;
  const int n_os = H * DS, n_oe = H * C, n_og = H * PV * 3;
  const int HK = H * K, NQ = H * DS + H * PQ * 3;
  const int PR = ((3 * HK + NQ + n_og + H + F) + 3) & ~3;  // floats per row slot
  auto attn_of = [&](int rr) { return smem + rr * PR; };            // H*K probabilities
  auto gl_of = [&](int rr) { return smem + rr * PR + HK; };         // H*K dA, then scale_t * dlogit
  auto d2_of = [&](int rr) { return smem + rr * PR + 2 * HK; };     // H*K squared point distances
  auto q_of = [&](int rr) { return smem + rr * PR + 3 * HK; };      // q_s | q_pts of the row
  auto dog_of = [&](int rr) { return smem + rr * PR + 3 * HK + NQ; };
  auto red_of = [&](int rr) { return smem + rr * PR + 3 * HK + NQ + n_og; };
  auto dfl_of = [&](int rr) { return smem + rr * PR + 3 * HK + NQ + n_og + H; };
  const int64_t row0 = static_cast<int64_t>(b) * K + i0;
  const int tid = threadIdx.x, nthr = blockDim.x;
  for (int rr = 0; rr < RR; ++rr) {
    const int64_t row_i = row0 + rr;
    const float* prow = proj + row_i * NP;
    const float* frow = feat + row_i * F;
    const float* dfrow = dfeat + row_i * F;
    const float* Rr = R + row_i * 9;
    float* qrow = q_of(rr);
    float* dfl = dfl_of(rr);
    float* dog = dog_of(rr);
    for (int d = tid; d < H * DS; d += nthr) qrow[d] = prow[d];
    for (int d = tid; d < H * PQ * 3; d += nthr) qrow[H * DS + d] = prow[off_gq + d];
    for (int d = tid; d < F; d += nthr) dfl[d] = dfrow[d];
    for (int hp = tid; hp < H * PV; hp += nthr) {  // d o_g[k] = sum_c (do_l[c] + do_n o_l[c]/o_n) R[c][k]
      const float on = frow[n_os + n_oe + n_og + hp], don = dfrow[n_os + n_oe + n_og + hp];
      float dl[3];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float ol = frow[n_os + n_oe + hp * 3 + c];
        dl[c] = dfrow[n_os + n_oe + hp * 3 + c] + (on > 0.0f ? don * ol / on : 0.0f);
      }
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const float v = dl[0] * Rr[0 * 3 + k] + dl[1] * Rr[1 * 3 + k] + dl[2] * Rr[2 * 3 + k];
        dog[hp * 3 + k] = v;
        dogbuf[row_i * n_og + hp * 3 + k] = v;
      }
    }
  }
  __syncthreads();
  const float scale_s = 1.0f / sqrtf(static_cast<float>(DS));
  const float scale_p = -0.5f / sqrtf(4.5f * PQ);
  const float scale_t = 1.0f / sqrtf(C > 0 ? 3.0f : 2.0f);  // 3 independent logits with the pair bias, 2 without (C == 0)
  const int lo = tid & 7, oct = tid >> 3, noct = nthr >> 3;
  const float* e0 = e + row0 * K * C;  // + rr * K * C
  // sum over the 8 lanes of an octet of 8 per-lane values, lane `lo` ending with total number `lo` (reduce-scatter: 7 shuffles)
  auto octet_reduce_scatter = [&](const float (&v)[8]) {
    float w4[4], w2[2];
    const bool b2 = (lo & 4) != 0, b1 = (lo & 2) != 0, b0 = (lo & 1) != 0;
#pragma unroll
    for (int k_ = 0; k_ < 4; ++k_) {
      const float send = b2 ? v[k_] : v[k_ + 4], keep = b2 ? v[k_ + 4] : v[k_];
      w4[k_] = keep + __shfl_xor(send, 4);
    }
#pragma unroll
    for (int k_ = 0; k_ < 2; ++k_) {
      const float send = b1 ? w4[k_] : w4[k_ + 2], keep = b1 ? w4[k_ + 2] : w4[k_];
      w2[k_] = keep + __shfl_xor(send, 2);
    }
    const float send = b0 ? w2[0] : w2[1], keep = b0 ? w2[1] : w2[0];
    return keep + __shfl_xor(send, 1);
  };
  const int lane = tid & 63, wave = tid >> 6, nwave = nthr >> 6;
  if constexpr (HAVE_P) {
    for (int idx = tid; idx < HK; idx += nthr) {
      const int h = idx / K, j = idx % K;
#pragma unroll
      for (int rr = 0; rr < RR; ++rr) {
        const int64_t o = ((static_cast<int64_t>(b) * H + h) * K + i0 + rr) * K + j;
        attn_of(rr)[idx] = Pn[o];
        if (!HAVE_G) d2_of(rr)[idx] = D2g[o];
        gl_of(rr)[idx] = dAkv[o];
      }
    }
    __syncthreads();
  } else {
  // ---- recompute logits: key-side vectors loaded once per (h, j), used for the RR rows
  for (int idx = oct; idx < HK; idx += noct) {
    const int h = idx / K, j = idx % K;
    const float* krow = proj + (static_cast<int64_t>(b) * K + j) * NP;
    float ls[RR], lp[RR];
#pragma unroll
    for (int rr = 0; rr < RR; ++rr) ls[rr] = lp[rr] = 0.f;
    for (int d = 4 * lo; d < DS; d += 32) {
      const v4 kv = *reinterpret_cast<const v4*>(krow + off_ks + h * DS + d);
#pragma unroll
      for (int rr = 0; rr < RR; ++rr) {
        const v4 qv = *reinterpret_cast<const v4*>(q_of(rr) + h * DS + d);
        ls[rr] += (qv[0] * kv[0] + qv[1] * kv[1]) + (qv[2] * kv[2] + qv[3] * kv[3]);
      }
    }
    for (int p = 4 * lo; p < PQ * 3; p += 32) {
      const v4 kv = *reinterpret_cast<const v4*>(krow + off_gk + h * PQ * 3 + p);
#pragma unroll
      for (int rr = 0; rr < RR; ++rr) {
        const v4 dd = *reinterpret_cast<const v4*>(q_of(rr) + H * DS + h * PQ * 3 + p) - kv;
        lp[rr] += (dd[0] * dd[0] + dd[1] * dd[1]) + (dd[2] * dd[2] + dd[3] * dd[3]);
      }
    }
#pragma unroll
    for (int rr = 0; rr < RR; ++rr) {
#pragma unroll
      for (int o = 1; o < 8; o <<= 1) {
        ls[rr] += __shfl_xor(ls[rr], o);
        lp[rr] += __shfl_xor(lp[rr], o);
      }
      if (lo == 0) {
        d2_of(rr)[idx] = lp[rr];
        attn_of(rr)[idx] = scale_t * (ls[rr] * scale_s + (scale_p * gamma[h]) * lp[rr]);  // the pair bias is added below
      }
    }
  }
  __syncthreads();
  // pair bias: one octet per (row, key) reads e[i][j][:] ONCE and forms the 8 heads' dot products (H <= 8: lane lo <-> head lo);
  // per (head, key) octets would read every pair row H times
  for (int idx = oct; idx < RR * K; idx += noct) {
    const int rr = idx / K, j = idx % K;
    float part[8];
#pragma unroll
    for (int hh = 0; hh < 8; ++hh) part[hh] = 0.f;
    for (int c = 4 * lo; c < C; c += 32) {
      const v4 ev = *reinterpret_cast<const v4*>(e0 + (static_cast<int64_t>(rr) * K + j) * C + c);
#pragma unroll
      for (int hh = 0; hh < 8; ++hh) {
        if (hh < H) {
          const v4 wv = *reinterpret_cast<const v4*>(Wb + hh * C + c);
          part[hh] += (ev[0] * wv[0] + ev[1] * wv[1]) + (ev[2] * wv[2] + ev[3] * wv[3]);
        }
      }
    }
    const float tot = octet_reduce_scatter(part);
    if (lo < H) attn_of(rr)[lo * K + j] += scale_t * tot;
  }
  __syncthreads();
  for (int rh = wave; rh < RR * H; rh += nwave) {  // softmax per (row, head)
    float* a = attn_of(rh / H) + (rh % H) * K;
    float m = -INFINITY;
    for (int j = lane; j < K; j += 64) m = fmaxf(m, a[j]);
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    float sacc = 0.f;
    for (int j = lane; j < K; j += 64) {
      const float ex = expf(a[j] - m);
      a[j] = ex;
      sacc += ex;
    }
    for (int o = 32; o > 0; o >>= 1) sacc += __shfl_xor(sacc, o);
    const float inv = 1.0f / sacc;
    for (int j = lane; j < K; j += 64) a[j] *= inv;
  }
  __syncthreads();
  }
  if constexpr (!HAVE_P) {
  // ---- dA[h][j] = do_s . v_s[j] + do_e . e[i][j] + do_g . gv[j]
  for (int idx = oct; idx < HK; idx += noct) {
    const int h = idx / K, j = idx % K;
    const float* vrow = proj + (static_cast<int64_t>(b) * K + j) * NP;
    float sacc[RR];
#pragma unroll
    for (int rr = 0; rr < RR; ++rr) sacc[rr] = 0.f;
    for (int d = 4 * lo; d < DS; d += 32) {
      const v4 v = *reinterpret_cast<const v4*>(vrow + off_vs + h * DS + d);
#pragma unroll
      for (int rr = 0; rr < RR; ++rr) {
        const v4 a = *reinterpret_cast<const v4*>(dfl_of(rr) + h * DS + d);
        sacc[rr] += (a[0] * v[0] + a[1] * v[1]) + (a[2] * v[2] + a[3] * v[3]);
      }
    }
    for (int p = 4 * lo; p < PV * 3; p += 32) {
      const v4 v = *reinterpret_cast<const v4*>(vrow + off_gv + h * PV * 3 + p);
#pragma unroll
      for (int rr = 0; rr < RR; ++rr) {
        const v4 a = *reinterpret_cast<const v4*>(dog_of(rr) + h * PV * 3 + p);
        sacc[rr] += (a[0] * v[0] + a[1] * v[1]) + (a[2] * v[2] + a[3] * v[3]);
      }
    }
#pragma unroll
    for (int rr = 0; rr < RR; ++rr) {
#pragma unroll
      for (int o = 1; o < 8; o <<= 1) sacc[rr] += __shfl_xor(sacc[rr], o);
      if (lo == 0) gl_of(rr)[idx] = sacc[rr];
    }
  }
  __syncthreads();
  }
  if constexpr (!HAVE_G) {
  for (int idx = oct; idx < RR * K; idx += noct) {  // the do_e . e[i][j] term of dA, all heads from one read of the pair row
    const int rr = idx / K, j = idx % K;
    float part[8];
#pragma unroll
    for (int hh = 0; hh < 8; ++hh) part[hh] = 0.f;
    for (int c = 4 * lo; c < C; c += 32) {
      const v4 ev = *reinterpret_cast<const v4*>(e0 + (static_cast<int64_t>(rr) * K + j) * C + c);
#pragma unroll
      for (int hh = 0; hh < 8; ++hh) {
        if (hh < H) {
          const v4 a = *reinterpret_cast<const v4*>(dfl_of(rr) + n_os + hh * C + c);
          part[hh] += (a[0] * ev[0] + a[1] * ev[1]) + (a[2] * ev[2] + a[3] * ev[3]);
        }
      }
    }
    const float tot = octet_reduce_scatter(part);
    if (lo < H) gl_of(rr)[lo * K + j] += tot;
  }
  __syncthreads();
  }
  if constexpr (!HAVE_G) {
  // ---- softmax backward: dlogit = A (dA - sum_j A dA); keep g = scale_t * dlogit
  for (int rh = wave; rh < RR * H; rh += nwave) {
    const float* a = attn_of(rh / H) + (rh % H) * K;
    const float* gg = gl_of(rh / H) + (rh % H) * K;
    float sacc = 0.f;
    for (int j = lane; j < K; j += 64) sacc += a[j] * gg[j];
    for (int o = 32; o > 0; o >>= 1) sacc += __shfl_xor(sacc, o);
    if (lane == 0) red_of(rh / H)[rh % H] = sacc;
  }
  __syncthreads();
  }
  for (int idx = tid; idx < HK; idx += nthr) {
    const int h = idx / K, j = idx % K;
    const int64_t o = ((static_cast<int64_t>(b) * H + h) * K + j) * K + i0;  // transposed: [b][h][j][i0 .. i0 + RR)
#pragma unroll
    for (int rr = 0; rr < RR; ++rr) {
      const float a = attn_of(rr)[idx];
      float g;
      if constexpr (HAVE_G) {
        g = gl_of(rr)[idx];
      } else {
        g = scale_t * a * (gl_of(rr)[idx] - red_of(rr)[h]);
        gl_of(rr)[idx] = g;
      }
      At[o + rr] = a;
      Gt[o + rr] = g;
      if constexpr (HAVE_P && !HAVE_G) Pn[((static_cast<int64_t>(b) * H + h) * K + i0 + rr) * K + j] = g;
    }
  }
  __syncthreads();
  if constexpr (!HAVE_G) {
  for (int rh = wave; rh < RR * H; rh += nwave) {  // dgamma_h partial of the row: sum_j g scale_p d2
    const int rr = rh / H, h = rh % H;
    float sacc = 0.f;
    for (int j = lane; j < K; j += 64) sacc += gl_of(rr)[h * K + j] * scale_p * d2_of(rr)[h * K + j];
    for (int o = 32; o > 0; o >>= 1) sacc += __shfl_xor(sacc, o);
    if (lane == 0) wb_part[(row0 + rr) * (H * C + H) + H * C + h] = sacc;
  }
  }
  if constexpr (!HAVE_P) {
  // ---- query-side gradients: each key value is loaded once for the RR rows
  for (int o = tid; o < NQ; o += nthr) {
    float acc[RR];
#pragma unroll
    for (int rr = 0; rr < RR; ++rr) acc[rr] = 0.f;
    if (o < H * DS) {
      const int h = o / DS;
      for (int j = 0; j < K; ++j) {
        const float kv = proj[(static_cast<int64_t>(b) * K + j) * NP + off_ks + o];
#pragma unroll
        for (int rr = 0; rr < RR; ++rr) acc[rr] += gl_of(rr)[h * K + j] * kv;
      }
#pragma unroll
      for (int rr = 0; rr < RR; ++rr) dproj[(row0 + rr) * NP + o] = acc[rr] * scale_s;
    } else {
      const int oo = o - H * DS, h = oo / (PQ * 3);
      const float cp = 2.0f * scale_p * gamma[h];
      float qv[RR];
#pragma unroll
      for (int rr = 0; rr < RR; ++rr) qv[rr] = q_of(rr)[H * DS + oo];
      for (int j = 0; j < K; ++j) {
        const float kv = proj[(static_cast<int64_t>(b) * K + j) * NP + off_gk + oo];
#pragma unroll
        for (int rr = 0; rr < RR; ++rr) acc[rr] += gl_of(rr)[h * K + j] * (qv[rr] - kv);
      }
#pragma unroll
      for (int rr = 0; rr < RR; ++rr) dproj[(row0 + rr) * NP + off_gq + oo] = cp * acc[rr];
    }
  }
  }
  // ---- pair embedding: de[i][j][c] += sum_h (A do_e[h][c] + g Wb[h][c]);   dWb[h][c] partial = sum_j g e[i][j][c]
  for (int rr = 0; rr < RR; ++rr) {
    const float* attn = attn_of(rr);
    const float* gl = gl_of(rr);
    const float* dfl = dfl_of(rr);
    const float* erow = e0 + static_cast<int64_t>(rr) * K * C;
    if (de) {
      float* derow = de + (row0 + rr) * K * C;
      for (int idx = tid; idx < K * C; idx += nthr) {
        const int j = idx / C, c = idx % C;
        float sacc = 0.f;
        for (int h = 0; h < H; ++h) sacc += attn[h * K + j] * dfl[n_os + h * C + c] + gl[h * K + j] * Wb[h * C + c];
        derow[idx] += sacc;
      }
    }
    if constexpr (!HAVE_G) {
    for (int o = tid; o < H * C; o += nthr) {
      const int h = o / C, c = o % C;
      float sacc = 0.f;
      for (int j = 0; j < K; ++j) sacc += gl[h * K + j] * erow[static_cast<int64_t>(j) * C + c];
      wb_part[(row0 + rr) * (H * C + H) + o] = sacc;
    }
    }
  }
}

// Pass 2: key-side gradients of residue j = sums over the query rows i, read contiguously from the transposed A / g images.
__global__ __launch_bounds__(256) void ipa_attn_bwd_keys_kernel(const float* __restrict__ proj, const float* __restrict__ gamma,
                                                                const float* __restrict__ dfeat, const float* __restrict__ At,
                                                                const float* __restrict__ Gt, const float* __restrict__ dogbuf,
                                                                float* __restrict__ dproj, int K, int C, int H, int DS, int PQ, int PV) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int b = blockIdx.x / K, j = blockIdx.x % K;
  const int NP = 3 * H * DS + 2 * H * PQ * 3 + H * PV * 3;
  const int F = H * DS + H * C + H * PV * 3 + H * PV;
  const int off_ks = H * DS, off_vs = 2 * H * DS, off_gq = 3 * H * DS, off_gk = off_gq + H * PQ * 3, off_gv = off_gk + H * PQ * 3;
  const int n_og = H * PV * 3;
  float* a = smem;        // H*K  A[h][i] for this key
  float* g = a + H * K;   // H*K  g[h][i]
  const int64_t base = static_cast<int64_t>(b) * H * K * K + static_cast<int64_t>(j) * K;
  for (int idx = threadIdx.x; idx < H * K; idx += blockDim.x) {
    const int h = idx / K, i = idx % K;
    a[idx] = At[base + static_cast<int64_t>(h) * K * K + i];
    g[idx] = Gt[base + static_cast<int64_t>(h) * K * K + i];
  }
  __syncthreads();
  const float scale_s = 1.0f / sqrtf(static_cast<float>(DS));
  const float scale_p = -0.5f / sqrtf(4.5f * PQ);
  const int64_t row_j = static_cast<int64_t>(b) * K + j;
  const float* prow_j = proj + row_j * NP;
  float* drow = dproj + row_j * NP;
  const int n_ks = H * DS, n_gk = H * PQ * 3, n_vs = H * DS, n_gv = H * PV * 3;
  for (int o = threadIdx.x; o < n_ks + n_gk + n_vs + n_gv; o += blockDim.x) {
    float acc = 0.f;
    if (o < n_ks) {  // dk_s[j][h][d] = scale_s sum_i g q_s[i][h][d]
      const int h = o / DS;
      for (int i = 0; i < K; ++i) acc += g[h * K + i] * proj[(static_cast<int64_t>(b) * K + i) * NP + o];
      drow[off_ks + o] = acc * scale_s;
    } else if (o < n_ks + n_gk) {  // dgk = -2 scale_p gamma sum_i g (gq[i] - gk[j])
      const int oo = o - n_ks, h = oo / (PQ * 3);
      const float kj = prow_j[off_gk + oo];
      for (int i = 0; i < K; ++i) acc += g[h * K + i] * (proj[(static_cast<int64_t>(b) * K + i) * NP + off_gq + oo] - kj);
      drow[off_gk + oo] = -2.0f * scale_p * gamma[h] * acc;
    } else if (o < n_ks + n_gk + n_vs) {  // dv_s[j][h][d] = sum_i A do_s[i][h][d]
      const int oo = o - n_ks - n_gk, h = oo / DS;
      for (int i = 0; i < K; ++i) acc += a[h * K + i] * dfeat[(static_cast<int64_t>(b) * K + i) * F + oo];
      drow[off_vs + oo] = acc;
    } else {  // dgv[j][h][p][k] = sum_i A dog[i][h][p][k]
      const int oo = o - n_ks - n_gk - n_vs, h = oo / (PV * 3);
      for (int i = 0; i < K; ++i) acc += a[h * K + i] * dogbuf[(static_cast<int64_t>(b) * K + i) * n_og + oo];
      drow[off_gv + oo] = acc;
    }
  }
}

// Key-side pass for JJ consecutive keys per work-group: every query-side value (q_s, q_pts, do_s, dog of row i) is loaded once and
// used for the JJ keys; the one-key kernel above re-reads that 0.45 MB per key.
template <int JJ>
__global__ __launch_bounds__(256) void ipa_attn_bwd_keys_mr_kernel(const float* __restrict__ proj, const float* __restrict__ gamma,
                                                                   const float* __restrict__ dfeat, const float* __restrict__ At,
                                                                   const float* __restrict__ Gt, const float* __restrict__ dogbuf,
                                                                   float* __restrict__ dproj, int K, int C, int H, int DS, int PQ, int PV) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int nblk = K / JJ;
  const int b = blockIdx.x / nblk, j0 = (blockIdx.x % nblk) * JJ;
  const int NP = 3 * H * DS + 2 * H * PQ * 3 + H * PV * 3;
  const int F = H * DS + H * C + H * PV * 3 + H * PV;
  const int off_ks = H * DS, off_vs = 2 * H * DS, off_gq = 3 * H * DS, off_gk = off_gq + H * PQ * 3, off_gv = off_gk + H * PQ * 3;
  const int n_og = H * PV * 3, HK = H * K;
  float* a = smem;            // [JJ][H*K]  A[h][i] of key j0 + jj
  float* g = a + JJ * HK;     // [JJ][H*K]  g[h][i]
  for (int idx = threadIdx.x; idx < JJ * HK; idx += blockDim.x) {
    const int jj = idx / HK, hi = idx % HK, h = hi / K, i = hi % K;
    const int64_t o = ((static_cast<int64_t>(b) * H + h) * K + j0 + jj) * K + i;
    a[idx] = At[o];
    g[idx] = Gt[o];
  }
  __syncthreads();
  const float scale_s = 1.0f / sqrtf(static_cast<float>(DS));
  const float scale_p = -0.5f / sqrtf(4.5f * PQ);
  const int64_t row_j0 = static_cast<int64_t>(b) * K + j0;
  const int n_ks = H * DS, n_gk = H * PQ * 3, n_vs = H * DS, n_gv = H * PV * 3;
  for (int o = threadIdx.x; o < n_ks + n_gk + n_vs + n_gv; o += blockDim.x) {
    float acc[JJ];
#pragma unroll
    for (int jj = 0; jj < JJ; ++jj) acc[jj] = 0.f;
    if (o < n_ks) {  // dk_s[j][h][d] = scale_s sum_i g q_s[i][h][d]
      const int h = o / DS;
      for (int i = 0; i < K; ++i) {
        const float v = proj[(static_cast<int64_t>(b) * K + i) * NP + o];
#pragma unroll
        for (int jj = 0; jj < JJ; ++jj) acc[jj] += g[jj * HK + h * K + i] * v;
      }
#pragma unroll
      for (int jj = 0; jj < JJ; ++jj) dproj[(row_j0 + jj) * NP + off_ks + o] = acc[jj] * scale_s;
    } else if (o < n_ks + n_gk) {  // dgk = -2 scale_p gamma sum_i g (gq[i] - gk[j])
      const int oo = o - n_ks, h = oo / (PQ * 3);
      float kj[JJ];
#pragma unroll
      for (int jj = 0; jj < JJ; ++jj) kj[jj] = proj[(row_j0 + jj) * NP + off_gk + oo];
      for (int i = 0; i < K; ++i) {
        const float v = proj[(static_cast<int64_t>(b) * K + i) * NP + off_gq + oo];
#pragma unroll
        for (int jj = 0; jj < JJ; ++jj) acc[jj] += g[jj * HK + h * K + i] * (v - kj[jj]);
      }
#pragma unroll
      for (int jj = 0; jj < JJ; ++jj) dproj[(row_j0 + jj) * NP + off_gk + oo] = -2.0f * scale_p * gamma[h] * acc[jj];
    } else if (o < n_ks + n_gk + n_vs) {  // dv_s[j][h][d] = sum_i A do_s[i][h][d]
      const int oo = o - n_ks - n_gk, h = oo / DS;
      for (int i = 0; i < K; ++i) {
        const float v = dfeat[(static_cast<int64_t>(b) * K + i) * F + oo];
#pragma unroll
        for (int jj = 0; jj < JJ; ++jj) acc[jj] += a[jj * HK + h * K + i] * v;
      }
#pragma unroll
      for (int jj = 0; jj < JJ; ++jj) dproj[(row_j0 + jj) * NP + off_vs + oo] = acc[jj];
    } else {  // dgv[j][h][p][k] = sum_i A dog[i][h][p][k]
      const int oo = o - n_ks - n_gk - n_vs, h = oo / (PV * 3);
      for (int i = 0; i < K; ++i) {
        const float v = dogbuf[(static_cast<int64_t>(b) * K + i) * n_og + oo];
#pragma unroll
        for (int jj = 0; jj < JJ; ++jj) acc[jj] += a[jj * HK + h * K + i] * v;
      }
#pragma unroll
      for (int jj = 0; jj < JJ; ++jj) dproj[(row_j0 + jj) * NP + off_gv + oo] = acc[jj];
    }
  }
}

// Key-side pass on the MFMA for the benchmark head geometry (DS = 32, 3 PQ = 3 PV = 24), one work-group per (patch, head, 64 keys),
// one wave per 16 keys.  The transposed images At / Gt [b][h][j][i] are the A operand (rows j, k = i contiguous: one float4 per
// lane = four k-steps); the query-side rows are staged once per work-group in LDS as [i][64]:
//   PART 0:  [q_s (32) | q_pts (24) | 1 | 0...]  x Gt  ->  dk_s = scale_s sum_i g q_s,  sum_i g q_pts,  sum_i g  (for d k_pts)
//   PART 1:  [do_s (32) | dog (24) | 0...]       x At  ->  dv_s,  d v_pts
//   PART 2 (query side, same shape with the roles of i and j swapped; A = g [b][h][i][j], staged rows are the KEY side):
//            [k_s (32) | k_pts (24) | 1 | 0...]  x g   ->  dq_s = scale_s sum_j g k_s,  sum_j g k_pts,  sum_j g  (for d q_pts)
constexpr int KM_LD = 68;  // LDS row stride (floats)
// TSRC: the image is stored transposed ([b][h][j][i]: one float4 per lane along k); otherwise [b][h][i][j] is read with the key as
// the lane index and the query row as k (four 4-byte loads per lane and 16 k, 64 contiguous bytes per 16 lanes) - no transposed copy
template <int PART, bool TSRC = true>
__global__ __launch_bounds__(256) void ipa_attn_bwd_keys_mfma_kernel(const float* __restrict__ proj, const float* __restrict__ gamma,
                                                                     const float* __restrict__ dfeat, const float* __restrict__ AGt,
                                                                     const float* __restrict__ dogbuf, float* __restrict__ dproj, int K) {
  typedef float v4 __attribute__((ext_vector_type(4)));
  extern __shared__ __attribute__((aligned(16))) float smem[];  // [K][KM_LD]
  constexpr int H = 8, DS = 32, NPT = 24;
  constexpr int NP = 3 * H * DS + 3 * H * NPT, F = H * DS + H * 64 + H * NPT + H * 8;
  constexpr int off_ks = H * DS, off_vs = 2 * H * DS, off_gq = 3 * H * DS, off_gk = off_gq + H * NPT, off_gv = off_gk + H * NPT;
  const int nkb = K / 64;
  const int kb = blockIdx.x % nkb, h = (blockIdx.x / nkb) % H, b = blockIdx.x / (nkb * H);
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, l15 = lane & 15, g = lane >> 4;
  const int64_t prow0 = static_cast<int64_t>(b) * K;
  for (int idx = tid; idx < K * 16; idx += 256) {  // 16 float4 per staged row
    const int i = idx >> 4, c4 = idx & 15;
    v4 v = {0.f, 0.f, 0.f, 0.f};
    if (PART == 0) {
      if (c4 < 8) v = *reinterpret_cast<const v4*>(proj + (prow0 + i) * NP + h * DS + 4 * c4);
      else if (c4 < 14) v = *reinterpret_cast<const v4*>(proj + (prow0 + i) * NP + off_gq + h * NPT + 4 * (c4 - 8));
      else if (c4 == 14) v[0] = 1.0f;  // column 56: sum_i g
    } else if (PART == 2) {
      if (c4 < 8) v = *reinterpret_cast<const v4*>(proj + (prow0 + i) * NP + off_ks + h * DS + 4 * c4);
      else if (c4 < 14) v = *reinterpret_cast<const v4*>(proj + (prow0 + i) * NP + off_gk + h * NPT + 4 * (c4 - 8));
      else if (c4 == 14) v[0] = 1.0f;  // column 56: sum_j g
    } else {
      if (c4 < 8) v = *reinterpret_cast<const v4*>(dfeat + (prow0 + i) * F + h * DS + 4 * c4);
      else if (c4 < 14) v = *reinterpret_cast<const v4*>(dogbuf + (prow0 + i) * (H * NPT) + h * NPT + 4 * (c4 - 8));
    }
    *reinterpret_cast<v4*>(smem + i * KM_LD + 4 * c4) = v;
  }
  __syncthreads();
  const int j0 = kb * 64 + 16 * wv;
  const float* arow = TSRC ? AGt + ((static_cast<int64_t>(b) * H + h) * K + j0 + l15) * K + 4 * g   // [j][i]: + 16 grp
                           : AGt + ((static_cast<int64_t>(b) * H + h) * K + 4 * g) * K + j0 + l15;  // [i][j]: + 16 grp rows
  v4 acc[4];
#pragma unroll
  for (int t_ = 0; t_ < 4; ++t_) acc[t_] = v4{0.f, 0.f, 0.f, 0.f};
  for (int grp = 0; grp < K / 16; ++grp) {
    v4 a;  // A[j = l15][i = 16 grp + 4 g + s]
    if (TSRC) {
      a = *reinterpret_cast<const v4*>(arow + 16 * grp);
    } else {
#pragma unroll
      for (int s_ = 0; s_ < 4; ++s_) a[s_] = arow[static_cast<int64_t>(16 * grp + s_) * K];
    }
#pragma unroll
    for (int s_ = 0; s_ < 4; ++s_) {
      const float* brow = smem + (16 * grp + 4 * g + s_) * KM_LD + l15;
#pragma unroll
      for (int t_ = 0; t_ < 4; ++t_) acc[t_] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s_], brow[16 * t_], acc[t_], 0, 0, 0);
    }
  }
  // acc[t][r]: key j0 + 4 g + r, column 16 t + l15
  const float scale_s = 0.17677669529663687f, scale_p = -0.5f * 0.16666666666666666f;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int64_t row_j = prow0 + j0 + 4 * g + r;
    float* drow = dproj + row_j * NP;
    if (PART == 0) {
      drow[off_ks + h * DS + l15] = acc[0][r] * scale_s;
      drow[off_ks + h * DS + 16 + l15] = acc[1][r] * scale_s;
      const float colsum = __shfl(acc[3][r], (lane & 48) | 8);  // column 56 = 48 + 8
      const float cpg = -2.0f * scale_p * gamma[h];
      const float* kj = proj + row_j * NP + off_gk + h * NPT;
      drow[off_gk + h * NPT + l15] = cpg * (acc[2][r] - kj[l15] * colsum);             // columns 32..47 -> point component 0..15
      if (l15 < 8) drow[off_gk + h * NPT + 16 + l15] = cpg * (acc[3][r] - kj[16 + l15] * colsum);  // columns 48..55 -> 16..23
    } else if (PART == 2) {  // "row_j" is the query row here
      drow[h * DS + l15] = acc[0][r] * scale_s;
      drow[h * DS + 16 + l15] = acc[1][r] * scale_s;
      const float rowsum = __shfl(acc[3][r], (lane & 48) | 8);
      const float cpq = 2.0f * scale_p * gamma[h];
      const float* qi = proj + row_j * NP + off_gq + h * NPT;
      drow[off_gq + h * NPT + l15] = cpq * (qi[l15] * rowsum - acc[2][r]);
      if (l15 < 8) drow[off_gq + h * NPT + 16 + l15] = cpq * (qi[16 + l15] * rowsum - acc[3][r]);
    } else {
      drow[off_vs + h * DS + l15] = acc[0][r];
      drow[off_vs + h * DS + 16 + l15] = acc[1][r];
      drow[off_gv + h * NPT + l15] = acc[2][r];
      if (l15 < 8) drow[off_gv + h * NPT + 16 + l15] = acc[3][r];
    }
  }
}

// The key-side pass on the bf16 matrix cores (six-term split, gemm_bf16x6.hip), K = 128, as two kernels of small work-groups.  The
// products are latency chains (load, split, product, store); with the whole K x K image resident (144 KiB of LDS, one work-group per
// CU, the three products one after the other: 112 us per layer at 128 patches, measured, and every single stage ablated changed
// nothing) nothing overlaps them - so the image streams through a 24 KiB slab instead and 4-5 work-groups share a CU.
//   ipa_attn_bwd_keys_tn_b6_kernel  (patch, head, image):  dk = g^T [q_s | q_pts | 1]  and  dv = P^T [do_s | dog]
//       contraction over the rows i of image and operand: 32-row slabs in the layout of gemm_tn_b6_kernel (three bf16 planes, 16-byte
//       chunks XOR-swizzled by row), both fragments through the transposing LDS read
//   ipa_attn_bwd_keys_nn_b6_kernel  (patch, head):  dq = g [k_s | k_pts | 1]
//       A fragments straight from global memory (a lane's 8 k values are 32 contiguous bytes of an image row, split in registers),
//       the operand as planes in LDS, read through the transposing read
// 96 matrix-pipe cycles per 16 x 16 x 32 block instead of 256 (f32 MFMA).  Same epilogues as ipa_attn_bwd_keys_mfma_kernel.
namespace {
typedef short kb_s16x4 __attribute__((ext_vector_type(4)));
typedef short kb_s16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 kb_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 kb_bf16x4 __attribute__((ext_vector_type(4)));
constexpr int KB_IMG = 32 * 128, KB_OPD = 32 * 64;  // bf16 elements of one plane of one slab (image | operand)
__device__ __forceinline__ int kb_off128(int row, int chunk) { return row * 128 + 8 * (chunk ^ (((row & 3) << 2) | ((row >> 2) & 3))); }
// 64-column rows are 128 bytes: rows of equal parity share their banks, so the 32-byte column-tile of rows {r, r + 2, r + 8, r + 10}
// (one transposing read) goes to four different chunk pairs
__device__ __forceinline__ int kb_off64(int row, int chunk) { return row * 64 + 8 * (chunk ^ ((((row >> 1) & 1) | (((row >> 3) & 1) << 1)) << 1)); }
__device__ __forceinline__ void kb_split3(float x, __bf16& h, __bf16& m, __bf16& l) {
  h = static_cast<__bf16>(x);
  const float r = x - static_cast<float>(h);
  m = static_cast<__bf16>(r);
  l = static_cast<__bf16>(r - static_cast<float>(m));
}
typedef float kb_v4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void kb_store_planes(__bf16* dst, int plane, kb_v4 v) {  // 4 consecutive elements of a row -> the three planes
  kb_bf16x4 hh, mm, ll;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    __bf16 a, b, d;
    kb_split3(v[c], a, b, d);
    hh[c] = a; mm[c] = b; ll[c] = d;
  }
  *reinterpret_cast<kb_bf16x4*>(dst) = hh;
  *reinterpret_cast<kb_bf16x4*>(dst + plane) = mm;
  *reinterpret_cast<kb_bf16x4*>(dst + 2 * plane) = ll;
}
__device__ __forceinline__ kb_bf16x8 kb_frag_tr(const __bf16* base, int off0, int off1) {
  const kb_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((kb_s16x4 __attribute__((address_space(3)))*)(base + off0));
  const kb_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((kb_s16x4 __attribute__((address_space(3)))*)(base + off1));
  const kb_s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(kb_bf16x8, v);
}
constexpr int KB_TA[6] = {1, 2, 0, 1, 0, 0}, KB_TB[6] = {1, 0, 2, 0, 1, 0};  // (mid,mid) (lo,hi) (hi,lo) (mid,hi) (hi,mid) (hi,hi): smallest first
constexpr int KB_H = 8, KB_DS = 32, KB_NPT = 24, KB_K = 128;
constexpr int KB_NP = 3 * KB_H * KB_DS + 3 * KB_H * KB_NPT, KB_F = KB_H * KB_DS + KB_H * 64 + KB_H * KB_NPT + KB_H * 8;
constexpr int KB_OFF_KS = KB_H * KB_DS, KB_OFF_VS = 2 * KB_H * KB_DS, KB_OFF_GQ = 3 * KB_H * KB_DS, KB_OFF_GK = KB_OFF_GQ + KB_H * KB_NPT,
              KB_OFF_GV = KB_OFF_GK + KB_H * KB_NPT;
// blockIdx -> (patch, item): the `per` items of a patch (its heads, or (head, image) pairs) write neighbouring pieces of the SAME d proj
// rows - 128- and 96-byte pieces of one line - so they must meet in ONE L2: consecutive work-groups go round-robin over the 8 XCDs, so
// the patch is the index that varies fastest in steps of 8 and a patch's items are 8 work-groups apart (measured: see DESIGN 7)
__device__ __forceinline__ void kb_patch_item(int bid, int B, int per, int& b, int& item) {
  if (B % 8 == 0) {
    b = (bid / (8 * per)) * 8 + (bid & 7);
    item = (bid >> 3) % per;
  } else {
    b = bid / per;
    item = bid % per;
  }
}
// float4 c4 (0..15) of row `i` of a 64-column operand: 0 = [q_s | q_pts | 1], 1 = [do_s | dog | 0], 2 = [k_s | k_pts | 1]
__device__ __forceinline__ kb_v4 kb_operand(int which, const float* __restrict__ proj, const float* __restrict__ dfeat,
                                            const float* __restrict__ dogbuf, int64_t row, int h, int c4) {
  kb_v4 v = {0.f, 0.f, 0.f, 0.f};
  if (which == 1) {
    if (c4 < 8) v = *reinterpret_cast<const kb_v4*>(dfeat + row * KB_F + h * KB_DS + 4 * c4);
    else if (c4 < 14) v = *reinterpret_cast<const kb_v4*>(dogbuf + row * (KB_H * KB_NPT) + h * KB_NPT + 4 * (c4 - 8));
  } else {
    const float* pr = proj + row * KB_NP;
    if (c4 < 8) v = *reinterpret_cast<const kb_v4*>(pr + (which == 0 ? 0 : KB_OFF_KS) + h * KB_DS + 4 * c4);
    else if (c4 < 14) v = *reinterpret_cast<const kb_v4*>(pr + (which == 0 ? KB_OFF_GQ : KB_OFF_GK) + h * KB_NPT + 4 * (c4 - 8));
    else if (c4 == 14) v[0] = 1.0f;  // column 56: the plain sum over the contraction index
  }
  return v;
}
// d k / d q epilogue of one accumulator pair (columns 32..47 | 48..63 of the product: points and the plain sum in column 56)
// own0 / own1: the row's own point components l15 / 16 + l15 (requested with the kernel's first loads: at the tail their latency was 20
// of the transposed kernel's 89 us)
__device__ __forceinline__ void kb_store_points(float* drow, float own0, float own1, float cp, float a0, float a1, int lane, int l15) {
  const float tot = __shfl(a1, (lane & 48) | 8);  // column 56 = 48 + 8
  drow[l15] = cp * (a0 - own0 * tot);
  if (l15 < 8) drow[16 + l15] = cp * (a1 - own1 * tot);
}
}  // namespace

// blockIdx.x = (patch, head, image): image 0 = g with [q_s | q_pts | 1] -> d k; image 1 = P with [do_s | dog] -> d v.  256 threads,
// wave w: output rows (= image columns = keys) 32 w .. 32 w + 31, all 64 columns.  LDS: one slab of image and operand planes (30 KiB).
__global__ __launch_bounds__(256) void ipa_attn_bwd_keys_tn_b6_kernel(const float* __restrict__ proj, const float* __restrict__ gamma,
                                                                      const float* __restrict__ dfeat, const float* __restrict__ G,
                                                                      const float* __restrict__ P, const float* __restrict__ dogbuf,
                                                                      float* __restrict__ dproj, int B) {
  __shared__ __attribute__((aligned(16))) __bf16 img[3 * KB_IMG];
  __shared__ __attribute__((aligned(16))) __bf16 opd[3 * KB_OPD];
  int b, item;
  kb_patch_item(blockIdx.x, B, 2 * KB_H, b, item);
  const int which = item & 1, h = item >> 1;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, l15 = lane & 15, g = lane >> 4;
  const int64_t prow0 = static_cast<int64_t>(b) * KB_K;
  const float* src = (which ? P : G) + (static_cast<int64_t>(b) * KB_H + h) * KB_K * KB_K;
  // the whole image and operand are requested up front (one latency), then staged slab by slab: image thread -> rows tid / 32 + 8 j
  // (j = 0..3) of each slab, float4 tid % 32; operand thread -> float4 tid + 256 k of the slab's [32][16]
  const int s_row = tid >> 5, s_f4 = tid & 31;
  kb_v4 ri[4][4], ro[4][2];
#pragma unroll
  for (int sl = 0; sl < 4; ++sl) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
      ri[sl][j] = __builtin_nontemporal_load(reinterpret_cast<const kb_v4*>(src + (32 * sl + s_row + 8 * j) * KB_K + 4 * s_f4));
#pragma unroll
    for (int k2 = 0; k2 < 2; ++k2) {
      const int idx = tid + 256 * k2;
      ro[sl][k2] = kb_operand(which, proj, dfeat, dogbuf, prow0 + 32 * sl + (idx >> 4), h, idx & 15);
    }
  }
  float own[2][4][2];  // d k: the key rows' own points (epilogue)
#pragma unroll
  for (int at = 0; at < 2; ++at)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float* o = proj + (prow0 + 16 * (2 * wv + at) + 4 * g + r) * KB_NP + KB_OFF_GK + h * KB_NPT;
      own[at][r][0] = which == 0 ? o[l15] : 0.f;
      own[at][r][1] = (which == 0 && l15 < 8) ? o[16 + l15] : 0.f;
    }
  const int q = l15 >> 2, pp = l15 & 3;
  int offA[2][2], offB[4][2];
#pragma unroll
  for (int rd = 0; rd < 2; ++rd) {
    const int row = 8 * g + 4 * rd + q;
#pragma unroll
    for (int at = 0; at < 2; ++at) offA[at][rd] = kb_off128(row, 2 * (2 * wv + at) + (pp >> 1)) + 4 * (pp & 1);
#pragma unroll
    for (int bt = 0; bt < 4; ++bt) offB[bt][rd] = kb_off64(row, 2 * bt + (pp >> 1)) + 4 * (pp & 1);
  }
  kb_v4 acc[2][4];
#pragma unroll
  for (int at = 0; at < 2; ++at)
#pragma unroll
    for (int bt = 0; bt < 4; ++bt) acc[at][bt] = kb_v4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int sl = 0; sl < 4; ++sl) {
    if (sl > 0) __syncthreads();  // every wave is done with the previous slab
#pragma unroll
    for (int j = 0; j < 4; ++j) kb_store_planes(img + kb_off128(s_row + 8 * j, s_f4 >> 1) + 4 * (s_f4 & 1), KB_IMG, ri[sl][j]);
#pragma unroll
    for (int k2 = 0; k2 < 2; ++k2) {
      const int idx = tid + 256 * k2;
      kb_store_planes(opd + kb_off64(idx >> 4, (idx & 15) >> 1) + 4 * (idx & 1), KB_OPD, ro[sl][k2]);
    }
    __syncthreads();
    kb_bf16x8 fa[2][3], fb[4][3];
#pragma unroll
    for (int p = 0; p < 3; ++p) {
#pragma unroll
      for (int at = 0; at < 2; ++at) fa[at][p] = kb_frag_tr(img + p * KB_IMG, offA[at][0], offA[at][1]);
#pragma unroll
      for (int bt = 0; bt < 4; ++bt) fb[bt][p] = kb_frag_tr(opd + p * KB_OPD, offB[bt][0], offB[bt][1]);
    }
#pragma unroll
    for (int term = 0; term < 6; ++term)
#pragma unroll
      for (int at = 0; at < 2; ++at)
#pragma unroll
        for (int bt = 0; bt < 4; ++bt)
          acc[at][bt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[at][KB_TA[term]], fb[bt][KB_TB[term]], acc[at][bt], 0, 0, 0);
  }
  // acc[at][bt][r]: key 16 (2 wv + at) + 4 g + r, column 16 bt + l15
  const float scale_s = 0.17677669529663687f, scale_p = -0.5f * 0.16666666666666666f;
  const float cp = -2.0f * scale_p * gamma[h];
#pragma unroll
  for (int at = 0; at < 2; ++at)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int64_t row = prow0 + 16 * (2 * wv + at) + 4 * g + r;
      float* drow = dproj + row * KB_NP;
      if (which == 0) {
        drow[KB_OFF_KS + h * KB_DS + l15] = acc[at][0][r] * scale_s;
        drow[KB_OFF_KS + h * KB_DS + 16 + l15] = acc[at][1][r] * scale_s;
        kb_store_points(drow + KB_OFF_GK + h * KB_NPT, own[at][r][0], own[at][r][1], cp, acc[at][2][r], acc[at][3][r], lane, l15);
      } else {
        drow[KB_OFF_VS + h * KB_DS + l15] = acc[at][0][r];
        drow[KB_OFF_VS + h * KB_DS + 16 + l15] = acc[at][1][r];
        drow[KB_OFF_GV + h * KB_NPT + l15] = acc[at][2][r];
        if (l15 < 8) drow[KB_OFF_GV + h * KB_NPT + 16 + l15] = acc[at][3][r];
      }
    }
}

// blockIdx.x = (patch, head): d q = g [k_s | k_pts | 1].  256 threads, wave w: query rows 32 w .. 32 w + 31.  LDS: the operand's planes
// [4 slabs of 32 keys][3][32][64] (48 KiB).
__global__ __launch_bounds__(256) void ipa_attn_bwd_keys_nn_b6_kernel(const float* __restrict__ proj, const float* __restrict__ gamma,
                                                                      const float* __restrict__ G, float* __restrict__ dproj, int B) {
  __shared__ __attribute__((aligned(16))) __bf16 opd[4 * 3 * KB_OPD];
  int b, h;
  kb_patch_item(blockIdx.x, B, KB_H, b, h);
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, l15 = lane & 15, g = lane >> 4;
  const int64_t prow0 = static_cast<int64_t>(b) * KB_K;
  const float* Gi = G + (static_cast<int64_t>(b) * KB_H + h) * KB_K * KB_K;
  // A: lane (row l15 of row tile at, k group g) of k step ks holds g[32 wv + 16 at + l15][32 ks + 8 g .. + 7]: two float4
  kb_v4 ra[4][2][2];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks)
#pragma unroll
    for (int at = 0; at < 2; ++at) {
      const float* p = Gi + (32 * wv + 16 * at + l15) * KB_K + 32 * ks + 8 * g;
      ra[ks][at][0] = __builtin_nontemporal_load(reinterpret_cast<const kb_v4*>(p));
      ra[ks][at][1] = __builtin_nontemporal_load(reinterpret_cast<const kb_v4*>(p + 4));
    }
  {
    kb_v4 ro[8];
#pragma unroll
    for (int k8 = 0; k8 < 8; ++k8) {
      const int idx = tid + 256 * k8;
      ro[k8] = kb_operand(2, proj, nullptr, nullptr, prow0 + (idx >> 4), h, idx & 15);
    }
#pragma unroll
    for (int k8 = 0; k8 < 8; ++k8) {
      const int idx = tid + 256 * k8, j = idx >> 4;
      kb_store_planes(opd + (j >> 5) * (3 * KB_OPD) + kb_off64(j & 31, (idx & 15) >> 1) + 4 * (idx & 1), KB_OPD, ro[k8]);
    }
  }
  float own[2][4][2];  // the query rows' own points (epilogue)
#pragma unroll
  for (int at = 0; at < 2; ++at)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float* o = proj + (prow0 + 32 * wv + 16 * at + 4 * g + r) * KB_NP + KB_OFF_GQ + h * KB_NPT;
      own[at][r][0] = o[l15];
      own[at][r][1] = l15 < 8 ? o[16 + l15] : 0.f;
    }
  const int q = l15 >> 2, pp = l15 & 3;
  int offB[4][2];
#pragma unroll
  for (int rd = 0; rd < 2; ++rd)
#pragma unroll
    for (int bt = 0; bt < 4; ++bt) offB[bt][rd] = kb_off64(8 * g + 4 * rd + q, 2 * bt + (pp >> 1)) + 4 * (pp & 1);
  kb_v4 acc[2][4];
#pragma unroll
  for (int at = 0; at < 2; ++at)
#pragma unroll
    for (int bt = 0; bt < 4; ++bt) acc[at][bt] = kb_v4{0.f, 0.f, 0.f, 0.f};
  __syncthreads();
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    kb_bf16x8 fa[2][3], fb[4][3];
#pragma unroll
    for (int at = 0; at < 2; ++at)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        __bf16 a, bq, cq;
        kb_split3(ra[ks][at][e >> 2][e & 3], a, bq, cq);
        fa[at][0][e] = a; fa[at][1][e] = bq; fa[at][2][e] = cq;
      }
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
      for (int bt = 0; bt < 4; ++bt) fb[bt][p] = kb_frag_tr(opd + (ks * 3 + p) * KB_OPD, offB[bt][0], offB[bt][1]);
#pragma unroll
    for (int term = 0; term < 6; ++term)
#pragma unroll
      for (int at = 0; at < 2; ++at)
#pragma unroll
        for (int bt = 0; bt < 4; ++bt)
          acc[at][bt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[at][KB_TA[term]], fb[bt][KB_TB[term]], acc[at][bt], 0, 0, 0);
  }
  const float scale_s = 0.17677669529663687f, scale_p = -0.5f * 0.16666666666666666f;
  const float cp = -2.0f * scale_p * gamma[h];  // d q_pts = 2 scale_p gamma (q sum g - sum g k) = cp (sum g k - q sum g)
#pragma unroll
  for (int at = 0; at < 2; ++at)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int64_t row = prow0 + 32 * wv + 16 * at + 4 * g + r;
      float* drow = dproj + row * KB_NP;
      drow[h * KB_DS + l15] = acc[at][0][r] * scale_s;
      drow[h * KB_DS + 16 + l15] = acc[at][1][r] * scale_s;
      kb_store_points(drow + KB_OFF_GQ + h * KB_NPT, own[at][r][0], own[at][r][1], cp, acc[at][2][r], acc[at][3][r], lane, l15);
    }
}

// dA_kv[b][h][i][j] = do_s[i] . v_s[j] + dog[i] . v_pts[j] on the MFMA (benchmark head geometry): work-group = (patch, head, 64
// query rows), the value side [j][v_s (32) | v_pts (24) | 0...] staged once in LDS, A = [do_s | dog] of 16 query rows per wave in
// registers (K-contiguous float4s, k order permuted identically for both operands).  The pair term do_e . e is added by the row pass.
__global__ __launch_bounds__(256) void ipa_attn_bwd_dakv_mfma_kernel(const float* __restrict__ proj, const float* __restrict__ dfeat,
                                                                     const float* __restrict__ dogbuf, float* __restrict__ dA, int K) {
  typedef float v4 __attribute__((ext_vector_type(4)));
  extern __shared__ __attribute__((aligned(16))) float smem[];  // [K][KM_LD]
  constexpr int H = 8, DS = 32, NPT = 24;
  constexpr int NP = 3 * H * DS + 3 * H * NPT, F = H * DS + H * 64 + H * NPT + H * 8;
  constexpr int off_vs = 2 * H * DS, off_gv = 3 * H * DS + 2 * H * NPT;
  const int nrb = K / 64;
  const int rb = blockIdx.x % nrb, h = (blockIdx.x / nrb) % H, b = blockIdx.x / (nrb * H);
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, l15 = lane & 15, g = lane >> 4;
  const int64_t prow0 = static_cast<int64_t>(b) * K;
  for (int idx = tid; idx < K * 16; idx += 256) {
    const int j = idx >> 4, c4 = idx & 15;
    v4 v = {0.f, 0.f, 0.f, 0.f};
    if (c4 < 8) v = *reinterpret_cast<const v4*>(proj + (prow0 + j) * NP + off_vs + h * DS + 4 * c4);
    else if (c4 < 14) v = *reinterpret_cast<const v4*>(proj + (prow0 + j) * NP + off_gv + h * NPT + 4 * (c4 - 8));
    *reinterpret_cast<v4*>(smem + j * KM_LD + 4 * c4) = v;
  }
  const int i0 = rb * 64 + 16 * wv;
  const int64_t row_i = prow0 + i0 + l15;
  v4 a[4];  // A[i = l15][k = 16 grp + 4 g + s]: do_s (grp 0, 1), dog 0..15 (grp 2), dog 16..23 | 0 (grp 3)
  a[0] = *reinterpret_cast<const v4*>(dfeat + row_i * F + h * DS + 4 * g);
  a[1] = *reinterpret_cast<const v4*>(dfeat + row_i * F + h * DS + 16 + 4 * g);
  a[2] = *reinterpret_cast<const v4*>(dogbuf + row_i * (H * NPT) + h * NPT + 4 * g);
  a[3] = g < 2 ? *reinterpret_cast<const v4*>(dogbuf + row_i * (H * NPT) + h * NPT + 16 + 4 * g) : v4{0.f, 0.f, 0.f, 0.f};
  __syncthreads();
  float* drow = dA + ((static_cast<int64_t>(b) * H + h) * K + i0 + 4 * g) * K + l15;  // + r rows, + 16 jt keys
  for (int jt = 0; jt < K / 16; ++jt) {
    v4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int grp = 0; grp < 4; ++grp) {
      const v4 bv = *reinterpret_cast<const v4*>(smem + (16 * jt + l15) * KM_LD + 16 * grp + 4 * g);  // B[k][n = key 16 jt + l15]
#pragma unroll
      for (int s_ = 0; s_ < 4; ++s_) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[grp][s_], bv[s_], acc, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) drow[static_cast<int64_t>(r) * K + 16 * jt] = acc[r];
  }
}

// d o_g (gradient w.r.t. the attention-weighted global value-point sums) of every (row, head, point):
// o_l = (o_g - t) R^T, o_n = |o_l|  ->  d o_g[k] = sum_c (do_l[c] + do_n o_l[c]/o_n) R[c][k]   (same expression as the row pass)
__global__ void ipa_dog_kernel(const float* __restrict__ feat, const float* __restrict__ dfeat, const float* __restrict__ R, int H, int C,
                               int DS, int PV, int64_t rows, float* __restrict__ dogbuf) {
  const int64_t gid = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x;
  const int HP = H * PV;
  if (gid >= rows * HP) return;
  const int64_t row = gid / HP;
  const int hp = static_cast<int>(gid % HP);
  const int F = H * DS + H * C + H * PV * 3 + H * PV, n_os = H * DS, n_oe = H * C, n_og = H * PV * 3;
  const float* frow = feat + row * F;
  const float* dfrow = dfeat + row * F;
  const float* Rr = R + row * 9;
  const float on = frow[n_os + n_oe + n_og + hp], don = dfrow[n_os + n_oe + n_og + hp];
  float dl[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float ol = frow[n_os + n_oe + hp * 3 + c];
    dl[c] = dfrow[n_os + n_oe + hp * 3 + c] + (on > 0.0f ? don * ol / on : 0.0f);
  }
#pragma unroll
  for (int k = 0; k < 3; ++k) dogbuf[row * n_og + hp * 3 + k] = dl[0] * Rr[0 * 3 + k] + dl[1] * Rr[1 * 3 + k] + dl[2] * Rr[2 * 3 + k];
}

// gradient w.r.t. global points -> local points, in place: g = p R + t  =>  dp[k] = sum_c dg[c] R[k][c]
__global__ void points_bwd_kernel(float* __restrict__ dproj, int ld, int col0, int n_pts, const float* __restrict__ R, int64_t rows) {
  const int64_t gid = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x;
  if (gid >= rows * n_pts) return;
  const int64_t r = gid / n_pts;
  float* q = dproj + r * ld + col0 + (gid % n_pts) * 3;
  const float* Rr = R + r * 9;
  const float x = q[0], y = q[1], z = q[2];
  q[0] = x * Rr[0] + y * Rr[1] + z * Rr[2];
  q[1] = x * Rr[3] + y * Rr[4] + z * Rr[5];
  q[2] = x * Rr[6] + y * Rr[7] + z * Rr[8];
}

// d seq_emb[s] += dcat2[:, D:2D];  d res_ctx = dcat2[:, 0:D]
__global__ void embed_bwd_kernel(const float* __restrict__ dcat2, const int64_t* __restrict__ seq, int D, int64_t rows,
                                 float* __restrict__ d_res_ctx, float* __restrict__ d_emb) {
  const int64_t r = blockIdx.x;
  if (r >= rows) return;
  const int64_t s = seq[r];
  for (int d = threadIdx.x; d < D; d += blockDim.x) {
    if (d_res_ctx) d_res_ctx[r * D + d] = dcat2[r * 2 * D + d];
    atomicAdd(d_emb + s * D + d, dcat2[r * 2 * D + D + d]);
  }
}

// ------------------------------------------------------------------ orchestration
size_t train_tape_floats(const diffab_dims* d) {
  const size_t rows = static_cast<size_t>(d->B) * d->K, D = d->D;
  const size_t NP = 3 * d->H * d->DS + 2 * d->H * d->PQ * 3 + d->H * d->PV * 3;
  const size_t F = d->H * d->DS + d->H * d->C + d->H * d->PV * 3 + d->H * d->PV;
  const size_t keep = (fast_path_supported(d) && attention_split_supported(d)) ? 2 * static_cast<size_t>(d->B) * d->H * d->K * d->K : 0;
  return rows * (2 * D + D + D) + static_cast<size_t>(d->NL) * (rows * (NP + F + D) + keep) + rows * (D + 3) + 6 * rows * D + rows * 3 +
         rows * d->V + 1024 + (fast_path_supported(d) ? ipa_layer_planes_bytes() / sizeof(float) + 64 : 0);
}

TrainTape carve_tape(const diffab_dims* d, float* base) {
  const size_t rows = static_cast<size_t>(d->B) * d->K, D = d->D;
  const size_t NP = 3 * d->H * d->DS + 2 * d->H * d->PQ * 3 + d->H * d->PV * 3;
  const size_t F = d->H * d->DS + d->H * d->C + d->H * d->PV * 3 + d->H * d->PV;
  TrainTape t;
  float* p = base;
  auto take = [&](size_t n) { float* r = p; p += n; return r; };
  t.cat2 = take(rows * 2 * D);
  t.h1 = take(rows * D);
  t.x[0] = take(rows * D);
  const bool keep = fast_path_supported(d) && attention_split_supported(d);
  const size_t HKK = static_cast<size_t>(d->B) * d->H * d->K * d->K;
  for (int l = 0; l < kMaxLayers; ++l) t.sp[l] = t.d2[l] = nullptr;
  for (int l = 0; l < d->NL; ++l) {
    t.ipa_ws[l] = take(rows * (NP + F));
    t.x[l + 1] = take(rows * D);
    if (keep) {
      t.sp[l] = take(HKK);
      t.d2[l] = take(HKK);
    }
  }
  t.cat3 = take(rows * (D + 3));
  for (int hd = 0; hd < 3; ++hd) { t.t1[hd] = take(rows * D); t.t2[hd] = take(rows * D); }
  t.vbuf = take(rows * 3);
  t.logits = take(rows * d->V);
  t.scratch = take(1024);  // (the +1024 of train_tape_floats)
  t.planes = nullptr;
  if (fast_path_supported(d)) {  // split bf16 planes of one layer's dense weights, re-filled by every layer's forward (stream order)
    float* raw = take(ipa_layer_planes_bytes() / sizeof(float) + 64);
    t.planes = reinterpret_cast<void*>((reinterpret_cast<uintptr_t>(raw) + 255) & ~static_cast<uintptr_t>(255));
  }
  return t;
}

// split bf16 planes of one transposed weight operand at a time (bf16x6 input-gradient products; stream-ordered reuse)
static size_t bwd_planes_floats(const diffab_dims* d) {
  if (d->D != 128) return 0;
  const size_t NP = 3 * d->H * d->DS + 2 * d->H * d->PQ * 3 + d->H * d->PV * 3;
  const size_t F = d->H * d->DS + d->H * d->C + d->H * d->PV * 3 + d->H * d->PV;
  const size_t kmax = ((NP > F ? NP : F) + 31) / 32 * 32;
  const size_t a = rowgemm128_b6_scratch_bytes(static_cast<int>(kmax)), b = xstat_b6_scratch_bytes(static_cast<int>(F)),
               c = xstat_h3_scratch_bytes(static_cast<int>(F));
  return (a > b ? (a > c ? a : c) : (b > c ? b : c)) / sizeof(float) + 128;
}
size_t train_bwd_workspace_floats(const diffab_dims* d) {
  const size_t rows = static_cast<size_t>(d->B) * d->K, D = d->D;
  const size_t NP = 3 * d->H * d->DS + 2 * d->H * d->PQ * 3 + d->H * d->PV * 3;
  const size_t F = d->H * d->DS + d->H * d->C + d->H * d->PV * 3 + d->H * d->PV;
  const size_t HKK = static_cast<size_t>(d->B) * d->H * d->K * d->K;
  return rows * (d->V + 3 + 3) + rows * (D + 3) + 6 * rows * D + 2 * rows * D + rows * F + 2 * rows * NP + rows * 2 * D + 64 + 2 * HKK +
         rows * d->H * d->PV * 3 + rows * (d->H * d->C + d->H) + 64 +
         ((fast_path_supported(d) && attention_split_supported(d)) ? 3 * HKK : 0) +  // probabilities / g, squared distances, dA_kv
         ((fast_path_supported(d) && attention_split_supported(d) && d->NL <= kDeLayersMax)
              ? static_cast<size_t>(d->NL) * (HKK + rows * d->H * d->C) + 64 : 0) +  // g and d o_e of every layer (d pair_ctx in one pass)
         bwd_planes_floats(d);
}

// One backward driver, three roots:
//   BWD_LOSSES      the three masked losses of diffab_pytorch.py:856-880 with upstream gradients upstream3 (the training step)
//   BWD_COTANGENTS  arbitrary cotangents of the Denoiser outputs (cot_eps, cot_O0, cot_post; null = zero): Denoiser.forward under autograd
//   BWD_LAYER       one IPA layer from d y (layer_dy) to d x (layer_dx): InvariantPointAttentionLayer.forward under autograd; `w` / `g`
//                   then carry only layers[0] and `tp` is a one-layer tape
enum { BWD_LOSSES = 0, BWD_COTANGENTS = 1, BWD_LAYER = 2 };
static int run_backward(int mode, const diffab_dims* d, const diffab_denoiser_weights* w, const diffab_denoiser_weights* g, const TrainTape& tp,
                        const int64_t* seq_t, const float* x_t, const float* O_t, const float* pair_ctx, const float* eps_hat,
                        const float* O0_hat, const float* post_hat, const float* true_post, const float* true_eps, const float* true_O0,
                        const uint8_t* gm, const uint8_t* rm, const float* upstream3, const float* cot_eps, const float* cot_O0,
                        const float* cot_post, const float* layer_dy, float* layer_dx, float* d_res_ctx, float* d_pair_ctx, float* ws,
                        hipStream_t st, float* d_x_t = nullptr, float* d_O_t = nullptr) {
  // d_x_t (rows x 3) / d_O_t (rows x 9), nullable: gradients with respect to the frames (translations_t, orientations_t), which the
  // reference's forward is differentiable in (:315-336, :594-596); the training step never asks for them
  const int rows = d->B * d->K, D = d->D, H = d->H, DS = d->DS, PQ = d->PQ, PV = d->PV, C = d->C, V = d->V;
  const int NP = 3 * H * DS + 2 * H * PQ * 3 + H * PV * 3;
  const int F = H * DS + H * C + H * PV * 3 + H * PV;
  float* p = ws;
  auto take = [&](size_t n) { float* r = p; p += n; return r; };
  float* cnt = take(64);
  float* d_logits = take(static_cast<size_t>(rows) * V);
  float* d_eps = take(static_cast<size_t>(rows) * 3);
  float* d_v = take(static_cast<size_t>(rows) * 3);
  float* dcat3 = take(static_cast<size_t>(rows) * (D + 3));
  float *dt2s[3], *dt1s[3];  // per head: the weight-gradient products beside the chain read them while the next head runs
  for (int hd = 0; hd < 3; ++hd) {
    dt2s[hd] = take(static_cast<size_t>(rows) * D);
    dt1s[hd] = take(static_cast<size_t>(rows) * D);
  }
  float* dxa = take(static_cast<size_t>(rows) * D);
  float* dxb = take(static_cast<size_t>(rows) * D);
  float* dfeat = take(static_cast<size_t>(rows) * F);
  float* dprojs[2] = {take(static_cast<size_t>(rows) * NP), take(static_cast<size_t>(rows) * NP)};  // by layer parity (same reason)
  float* dcat2 = take(static_cast<size_t>(rows) * 2 * D);
  const size_t HKK = static_cast<size_t>(d->B) * H * d->K * d->K;
  float* At = take(HKK);
  float* Gt = take(HKK);
  float* dogbuf = take(static_cast<size_t>(rows) * H * PV * 3);
  float* wb_part = take(static_cast<size_t>(rows) * (H * C + H));  // per-row partials of d w_bias (H*C) and d gamma (H)
  const bool mfma_probs = fast_path_supported(d) && attention_split_supported(d);
  float* Pn = mfma_probs ? take(HKK) : nullptr;
  float* D2g = mfma_probs ? take(HKK) : nullptr;
  float* dAkv = mfma_probs ? take(HKK) : nullptr;
  // d pair_ctx in ONE pass behind the layer loop (launch_pair_de_layers) instead of a read-modify-write of the whole gradient per layer:
  // needs g and the o_e columns of d feat of every layer kept, and the probabilities of every layer on the tape
  bool defer_de = mfma_probs && d_pair_ctx != nullptr && mode != BWD_LAYER && d->NL <= kDeLayersMax && C == 64 && H == 8 &&
                  fast_path_supported(d) && attention_split_supported(d);
  for (int l = 0; l < d->NL && defer_de; ++l) defer_de = tp.sp[l] != nullptr;
  float* g_keep = nullptr;
  float* doe_keep = nullptr;
  if (mfma_probs && fast_path_supported(d) && attention_split_supported(d) && d->NL <= kDeLayersMax) {  // (the slots exist either way: the size
    g_keep = take(static_cast<size_t>(d->NL) * HKK);                                                     //  function does not know d_pair_ctx)
    doe_keep = take(static_cast<size_t>(d->NL) * rows * H * C + 64);
  }
  void* planes = nullptr;  // bf16x6 input-gradient products (D = 128): the transposed weights as split planes, one operand at a time
  if (bwd_planes_floats(d) > 0 && use_b6_gemm())
    planes = reinterpret_cast<void*>((reinterpret_cast<uintptr_t>(take(bwd_planes_floats(d))) + 255) & ~static_cast<uintptr_t>(255));

  float* dcur = dxa;
  float* dnxt = dxb;
  SideRun side(st);  // weight / bias gradients beside the chain; joined when it goes out of scope (every return path)
  if (mode == BWD_LOSSES) {
    hipLaunchKernelGGL(count_mask_kernel, dim3(1), dim3(1024), 0, st, gm, rm, static_cast<int64_t>(rows), cnt);
    DIFFAB_LAUNCH_CHECK();
    if (int rc = launch_losses_bwd(post_hat, true_post, eps_hat, true_eps, O0_hat, true_O0, O_t, tp.vbuf, gm, rm, cnt, upstream3, V,
                                   static_cast<int64_t>(rows), d_logits, d_eps, d_v, st))
      return rc;
  } else if (mode == BWD_COTANGENTS) {
    if (int rc = launch_heads_cotangent(post_hat, cot_post, cot_eps, cot_O0, O_t, tp.vbuf, V, static_cast<int64_t>(rows), d_logits, d_eps, d_v,
                                        d_O_t, st))
      return rc;
  } else {
    DIFFAB_HIP_CHECK(hipMemcpyAsync(dcur, layer_dy, sizeof(float) * rows * D, hipMemcpyDeviceToDevice, st));
  }
  if (d_O_t && mode != BWD_COTANGENTS) DIFFAB_HIP_CHECK(hipMemsetAsync(d_O_t, 0, sizeof(float) * rows * 9, st));  // (the heads kernel wrote it above)
  if (d_x_t) DIFFAB_HIP_CHECK(hipMemsetAsync(d_x_t, 0, sizeof(float) * rows * 3, st));
  // ---- heads (Linear-ReLU-Linear-ReLU-Linear), gradients into cat3 accumulate over the three heads
  const diffab_mlp3_weights* hw[3] = {&w->coord, &w->orient, &w->seq};
  const diffab_mlp3_weights* hg[3] = {&g->coord, &g->orient, &g->seq};
  const float* dy[3] = {d_eps, d_v, d_logits};
  const int nout[3] = {3, 3, V};
  for (int hd = 0; hd < 3 && mode != BWD_LAYER; ++hd) {
    float *dt2 = dt2s[hd], *dt1 = dt1s[hd];
    if (int rc = linear_bwd(dy[hd], nout[hd], tp.t2[hd], D, hw[hd]->w4, const_cast<float*>(hg[hd]->w4), const_cast<float*>(hg[hd]->b4), dt2,
                            D, rows, nout[hd], D, false, st, &side)) return rc;
    if (int rc = relu_mask(dt2, tp.t2[hd], static_cast<int64_t>(rows) * D, st)) return rc;
    if (int rc = linear_bwd(dt2, D, tp.t1[hd], D, hw[hd]->w2, const_cast<float*>(hg[hd]->w2), const_cast<float*>(hg[hd]->b2), dt1, D, rows,
                            D, D, false, st, &side)) return rc;
    if (int rc = relu_mask(dt1, tp.t1[hd], static_cast<int64_t>(rows) * D, st)) return rc;
    if (int rc = linear_bwd(dt1, D, tp.cat3, D + 3, hw[hd]->w0, const_cast<float*>(hg[hd]->w0), const_cast<float*>(hg[hd]->b0), dcat3,
                            D + 3, rows, D, D + 3, hd > 0, st, &side)) return rc;
  }
  // dh = dcat3[:, :D] (leading dimension D+3)
  if (mode != BWD_LAYER)
    DIFFAB_HIP_CHECK(hipMemcpy2DAsync(dcur, sizeof(float) * D, dcat3, sizeof(float) * (D + 3), sizeof(float) * D, rows,
                                      hipMemcpyDeviceToDevice, st));
  // ---- IPA layers, last to first
  for (int l = d->NL - 1; l >= 0; --l) {
    const diffab_ipa_layer_weights* lw = &w->layers[l];
    const diffab_ipa_layer_weights* lg = &g->layers[l];
    const float* proj = tp.ipa_ws[l];
    const float* feat = tp.ipa_ws[l] + static_cast<size_t>(rows) * NP;
    const float* xin = tp.x[l];
    float* dproj = dprojs[l & 1];
    side.before_write(dproj);  // the projection weight gradient of layer l + 2 (beside the chain) read this buffer
    // to_out
    if (planes && (reinterpret_cast<uintptr_t>(dcur) & 15) == 0) {  // d feat = d y W_out on the x-stationary kernel (fp16 x 3; bf16 x 6 under the A/B switch)
      if (int rc = linear_bwd(dcur, D, feat, F, lw->w_out, const_cast<float*>(lg->w_out), const_cast<float*>(lg->b_out), nullptr, F, rows, D,
                              F, false, st, &side)) return rc;
      if (dense_h3_enabled()) {
        if (int rc = launch_xstat_h3(dcur, lw->w_out, 1, F, dfeat, F, rows, F, planes, st)) return rc;
      } else if (int rc = launch_xstat_b6(dcur, lw->w_out, 1, F, dfeat, F, rows, F, planes, st)) {
        return rc;
      }
    } else if (int rc = linear_bwd(dcur, D, feat, F, lw->w_out, const_cast<float*>(lg->w_out), const_cast<float*>(lg->b_out), dfeat, F, rows,
                                   D, F, false, st, &side)) {
      return rc;
    }
    const size_t lds = (3 * static_cast<size_t>(H) * d->K + H * DS + H * PQ * 3 + H * PV * 3 + H + F) * sizeof(float);
    const int vec = (DS % 4 == 0 && C % 4 == 0 && (PQ * 3) % 4 == 0 && (PV * 3) % 4 == 0 && H % 4 == 0 &&
                     (reinterpret_cast<uintptr_t>(pair_ctx) & 15) == 0 && (reinterpret_cast<uintptr_t>(lw->w_bias) & 15) == 0) ? 1 : 0;
    DIFFAB_REQUIRE(lds <= 160 * 1024, DIFFAB_ERR_UNSUPPORTED, "attention backward: H*K too large for LDS (%zu bytes)", lds);
    if (lds > 64 * 1024)
      DIFFAB_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(ipa_attn_bwd_rows_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
    bool keys_done = false;  // the MFMA path below also does the key-side pass
    int wb_rows = rows;       // rows of the d w_bias / d gamma partials (the MFMA path writes one per 32 query rows)
    constexpr int RRm = 4;  // query rows per work-group of the multi-row kernel
    const size_t slot = ((3 * static_cast<size_t>(H) * d->K + H * DS + H * PQ * 3 + H * PV * 3 + H + F) + 3) & ~static_cast<size_t>(3);
    const size_t lds_mr = RRm * slot * sizeof(float);
    if (vec && H <= 8 && d->K % RRm == 0 && lds_mr <= 160 * 1024) {
      if (mfma_probs) {
        // MFMA path: probabilities and squared distances by the forward's kernels, the key/value part of dA as a batched GEMM; the
        // row pass keeps what needs the pair row (do_e . e, softmax backward, d e, d w_bias, d gamma) and hands g to two more GEMMs
        const size_t lds_km = static_cast<size_t>(d->K) * KM_LD * sizeof(float);
        const dim3 grid_km(d->B * H * (d->K / 64));
        const int64_t ndog = static_cast<int64_t>(rows) * H * PV;
        hipLaunchKernelGGL(ipa_dog_kernel, dim3(static_cast<unsigned>((ndog + 255) / 256)), dim3(256), 0, st, feat, dfeat, O_t, H, C, DS, PV,
                           static_cast<int64_t>(rows), dogbuf);
        DIFFAB_LAUNCH_CHECK();
        if (tp.sp[l] != nullptr) {  // saved by the taped forward
          Pn = tp.sp[l];
          D2g = tp.d2[l];
        } else if (int rc = launch_attention_probs(d, proj, pair_ctx, lw->w_bias, lw->gamma, Pn, D2g, st)) {
          return rc;
        }
        float* Gl = defer_de ? g_keep + static_cast<size_t>(l) * HKK : dAkv;  // d A_kv, then g, of this layer
        hipLaunchKernelGGL(ipa_attn_bwd_dakv_mfma_kernel, grid_km, dim3(256), lds_km, st, proj, dfeat, dogbuf, Gl, d->K);
        DIFFAB_LAUNCH_CHECK();
        // Gl holds g afterwards.  d pair_ctx: accumulated by the same kernel (a read-modify-write of the whole gradient per layer), or -
        // defer_de - left to ONE pass behind the layer loop, for which this layer's g (Gl is its own slot) and d o_e are kept
        if (int rc = launch_pair_stream_bwd(d, pair_ctx, Pn, Gl, D2g, dfeat, wb_part, lw->w_bias, defer_de ? nullptr : d_pair_ctx, st)) return rc;
        if (defer_de)
          DIFFAB_HIP_CHECK(hipMemcpy2DAsync(doe_keep + static_cast<size_t>(l) * rows * H * C, sizeof(float) * H * C, dfeat + H * DS,
                                            sizeof(float) * F, sizeof(float) * H * C, rows, hipMemcpyDeviceToDevice, st));
        keys_done = true;
        DIFFAB_REQUIRE(rows % 32 == 0, DIFFAB_ERR_UNSUPPORTED, "attention backward (MFMA path): B K = %d must be a multiple of 32", rows);
        wb_rows = rows / 32;  // one partial row per 32 query rows (launch_pair_stream_bwd)
        // key side straight from the [b][h][i][j] images (g in dAkv, P in Pn): no transposed copies, no VALU row pass at all
        if (d->K == 128) {  // bf16 matrix cores: d k and d v (transposed products), d q
          hipLaunchKernelGGL(ipa_attn_bwd_keys_tn_b6_kernel, dim3(d->B * H * 2), dim3(256), 0, st, proj, lw->gamma, dfeat, Gl, Pn, dogbuf,
                             dproj, d->B);
          hipLaunchKernelGGL(ipa_attn_bwd_keys_nn_b6_kernel, dim3(d->B * H), dim3(256), 0, st, proj, lw->gamma, Gl, dproj, d->B);
        } else {
        hipLaunchKernelGGL((ipa_attn_bwd_keys_mfma_kernel<0, false>), grid_km, dim3(256), lds_km, st, proj, lw->gamma, dfeat, Gl, dogbuf,
                           dproj, d->K);
        hipLaunchKernelGGL((ipa_attn_bwd_keys_mfma_kernel<1, false>), grid_km, dim3(256), lds_km, st, proj, lw->gamma, dfeat, Pn, dogbuf,
                           dproj, d->K);
        hipLaunchKernelGGL(ipa_attn_bwd_keys_mfma_kernel<2>, grid_km, dim3(256), lds_km, st, proj, lw->gamma, dfeat, Gl, dogbuf, dproj, d->K);
        }
      } else {
        DIFFAB_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(ipa_attn_bwd_rows_mr_kernel<RRm, 0>),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds_mr)));
        hipLaunchKernelGGL((ipa_attn_bwd_rows_mr_kernel<RRm, 0>), dim3(rows / RRm), dim3(512), lds_mr, st, proj, pair_ctx, O_t,
                           lw->w_bias, lw->gamma, feat, dfeat, dproj, d_pair_ctx, At, Gt, dogbuf, wb_part, d->K, C, H, DS, PQ, PV, nullptr,
                           nullptr, nullptr);
      }
    } else
    {
      hipLaunchKernelGGL(ipa_attn_bwd_rows_kernel, dim3(rows), dim3(256), lds, st, proj, pair_ctx, O_t, lw->w_bias, lw->gamma, feat, dfeat,
                         dproj, d_pair_ctx, const_cast<float*>(lg->w_bias), const_cast<float*>(lg->gamma), At, Gt, dogbuf, wb_part, d->K, C, H,
                         DS, PQ, PV, vec);
    }
    DIFFAB_LAUNCH_CHECK();
    // d w_bias[h][c] += sum_rows partial, d gamma[h] += sum_rows partial (one column sum over the [rows][H*C + H] partials)
    if (keys_done && C > 0) {  // both sums in ONE launch (per-work-group partial rows -> two destinations; two colsum launches of 12.5 us before)
      PartsSegs sg{};
      sg.nseg = 2;
      sg.off[0] = 0; sg.n[0] = H * C; sg.cols[0] = H * C; sg.ld[0] = H * C; sg.out[0] = const_cast<float*>(lg->w_bias);
      sg.off[1] = H * C; sg.n[1] = H; sg.cols[1] = H; sg.ld[1] = H; sg.out[1] = const_cast<float*>(lg->gamma);
      if (int rc = launch_parts_reduce(wb_part, wb_rows, H * C + H, sg, st)) return rc;
    } else {
      if (C > 0)  // (use_pair_bias = False: no pair bias, no d w_bias)
        if (int rc = colsum(wb_part, H * C + H, wb_rows, H * C, const_cast<float*>(lg->w_bias), st)) return rc;
      if (int rc = colsum(wb_part + H * C, H * C + H, wb_rows, H, const_cast<float*>(lg->gamma), st)) return rc;
    }
    const size_t lds2 = 2 * static_cast<size_t>(H) * d->K * sizeof(float);
    constexpr int JJm = 4;  // keys per work-group of the multi-key kernel
    if (keys_done) {
    } else if (fast_path_supported(d) && d->K % 64 == 0 && static_cast<size_t>(d->K) * KM_LD * sizeof(float) <= 64 * 1024) {
      const size_t lds_km = static_cast<size_t>(d->K) * KM_LD * sizeof(float);
      const dim3 grid_km(d->B * H * (d->K / 64));
      hipLaunchKernelGGL(ipa_attn_bwd_keys_mfma_kernel<0>, grid_km, dim3(256), lds_km, st, proj, lw->gamma, dfeat, Gt, dogbuf, dproj, d->K);
      hipLaunchKernelGGL(ipa_attn_bwd_keys_mfma_kernel<1>, grid_km, dim3(256), lds_km, st, proj, lw->gamma, dfeat, At, dogbuf, dproj, d->K);
    } else if (d->K % JJm == 0 && JJm * lds2 <= 64 * 1024) {
      hipLaunchKernelGGL((ipa_attn_bwd_keys_mr_kernel<JJm>), dim3(rows / JJm), dim3(256), JJm * lds2, st, proj, lw->gamma, dfeat, At, Gt,
                         dogbuf, dproj, d->K, C, H, DS, PQ, PV);
    } else {
      hipLaunchKernelGGL(ipa_attn_bwd_keys_kernel, dim3(rows), dim3(256), lds2, st, proj, lw->gamma, dfeat, At, Gt, dogbuf, dproj, d->K, C, H,
                         DS, PQ, PV);
    }
    DIFFAB_LAUNCH_CHECK();
    if (d_x_t || d_O_t) {  // frame gradients of this layer, from the GLOBAL point gradients (before points_bwd_kernel rewrites dproj)
      if (int rc = launch_ipa_frames_bwd(proj, dproj, NP, 3 * H * DS, 2 * H * PQ + H * PV, feat, dfeat, F, H * DS + H * C,
                                         H * DS + H * C + H * PV * 3, H * PV, O_t, x_t, d_O_t, d_x_t, rows, st))
        return rc;
    }
    // global-point gradients -> local-point gradients (three point blocks)
    {  // the three point blocks (q, k, v) are adjacent columns of dproj: one launch over all of a row's points
      const int npts = 2 * H * PQ + H * PV;
      const int64_t n = static_cast<int64_t>(rows) * npts;
      hipLaunchKernelGGL(points_bwd_kernel, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, st, dproj, NP, 3 * H * DS, npts, O_t,
                         static_cast<int64_t>(rows));
      DIFFAB_LAUNCH_CHECK();
    }
    // projections: dx = sum_seg dproj_seg W_seg, dW_seg += dproj_seg^T x_in
    const float* Ws[6] = {lw->wq_s, lw->wk_s, lw->wv_s, lw->wq_p, lw->wk_p, lw->wv_p};
    float* dWs[6] = {const_cast<float*>(lg->wq_s), const_cast<float*>(lg->wk_s), const_cast<float*>(lg->wv_s),
                     const_cast<float*>(lg->wq_p), const_cast<float*>(lg->wk_p), const_cast<float*>(lg->wv_p)};
    const int Ns[6] = {H * DS, H * DS, H * DS, H * PQ * 3, H * PQ * 3, H * PV * 3};
    bool segs_ok = D % 4 == 0 && NP % 4 == 0 && D % 64 == 0 && (reinterpret_cast<uintptr_t>(xin) & 15) == 0 &&
                   (reinterpret_cast<uintptr_t>(dproj) & 15) == 0 && (reinterpret_cast<uintptr_t>(dnxt) & 15) == 0;
    GemmSegs gw{}, gd{};
    gw.nseg = gd.nseg = 6;
    int col = 0;
    for (int q = 0; q < 6; ++q) {
      col += Ns[q];
      gw.p[q] = const_cast<float*>(Ws[q]);
      gd.p[q] = dWs[q];
      gw.n_end[q] = gd.n_end[q] = col;
      segs_ok = segs_ok && Ns[q] % 64 == 0 && (reinterpret_cast<uintptr_t>(Ws[q]) & 15) == 0;
    }
    side.before_write(dnxt);  // (to_out's weight gradient of layer l + 1 read it as its d y)
    if (segs_ok) {  // the six projections as ONE weight-gradient product and ONE input-gradient product over the 1344-wide dproj
      if (int rc = gemm_tn(dproj, NP, xin, D, nullptr, D, rows, NP, D, side.begin(), &gd)) return rc;
      side.reads(dproj);
      if (planes && NP % 32 == 0 && rowgemm128_b6_ok(dproj, NP, dnxt, D, rows, NP)) {
        // dx = dproj [W_q_s; ...; W_v_p] as Y = X W'^T with W'[n][k] = W_seg[k - k0][n]: six strided splits into one set of planes
        if (int rc = launch_wsplit128_segs(Ws, gw.n_end, 6, planes, st)) return rc;
        if (int rc = launch_rowgemm128_b6p(dproj, NP, planes, nullptr, nullptr, 0, dnxt, D, rows, NP, false, st)) return rc;
      } else if (int rc = gemm_nn(dproj, NP, nullptr, D, dnxt, D, rows, D, NP, false, st, &gw)) {
        return rc;
      }
    } else {
      col = 0;
      for (int q = 0; q < 6; ++q) {
        if (int rc = linear_bwd(dproj + col, NP, xin, D, Ws[q], dWs[q], nullptr, dnxt, D, rows, Ns[q], D, q > 0, st)) return rc;  // (generic dims: on the chain)
        col += Ns[q];
      }
    }
    float* tmp = dcur; dcur = dnxt; dnxt = tmp;
  }
  if (defer_de) {  // d pair_ctx += the pair-bias and o_e terms of all layers, one pass (attention_split.hip)
    const float* Ps[kDeLayersMax];
    const float* Gs[kDeLayersMax];
    const float* Es[kDeLayersMax];
    const float* Wbs[kDeLayersMax];
    for (int l = 0; l < d->NL; ++l) {
      Ps[l] = tp.sp[l];
      Gs[l] = g_keep + static_cast<size_t>(l) * HKK;
      Es[l] = doe_keep + static_cast<size_t>(l) * rows * H * C;
      Wbs[l] = w->layers[l].w_bias;
    }
    if (int rc = launch_pair_de_layers(d, d->NL, Ps, Gs, Es, Wbs, d_pair_ctx, st)) return rc;
  }
  if (mode == BWD_LAYER) {
    DIFFAB_HIP_CHECK(hipMemcpyAsync(layer_dx, dcur, sizeof(float) * rows * D, hipMemcpyDeviceToDevice, st));
    return DIFFAB_OK;
  }
  // ---- to_res_emb (Linear-ReLU-Linear) and the sequence embedding
  side.before_write(dnxt);
  if (int rc = linear_bwd(dcur, D, tp.h1, D, w->res_w2, const_cast<float*>(g->res_w2), const_cast<float*>(g->res_b2), dnxt, D, rows, D, D,
                          false, st, &side)) return rc;
  if (int rc = relu_mask(dnxt, tp.h1, static_cast<int64_t>(rows) * D, st)) return rc;
  if (int rc = linear_bwd(dnxt, D, tp.cat2, 2 * D, w->res_w0, const_cast<float*>(g->res_w0), const_cast<float*>(g->res_b0), dcat2, 2 * D,
                          rows, D, 2 * D, false, st, &side)) return rc;
  hipLaunchKernelGGL(embed_bwd_kernel, dim3(rows), dim3(128), 0, st, dcat2, seq_t, D, static_cast<int64_t>(rows), d_res_ctx,
                     const_cast<float*>(g->seq_emb));
  DIFFAB_LAUNCH_CHECK();
  return DIFFAB_OK;
}

int train_step_bwd(const diffab_dims* d, const diffab_denoiser_weights* w, const diffab_denoiser_weights* g, const TrainTape& tp,
                   const int64_t* seq_t, const float* x_t, const float* O_t, const float* pair_ctx, const float* eps_hat,
                   const float* O0_hat, const float* post_hat, const float* true_post, const float* true_eps, const float* true_O0,
                   const uint8_t* gm, const uint8_t* rm, const float* upstream3, float* d_res_ctx, float* d_pair_ctx, float* ws,
                   hipStream_t st) {
  return run_backward(BWD_LOSSES, d, w, g, tp, seq_t, x_t, O_t, pair_ctx, eps_hat, O0_hat, post_hat, true_post, true_eps, true_O0, gm, rm,
                      upstream3, nullptr, nullptr, nullptr, nullptr, nullptr, d_res_ctx, d_pair_ctx, ws, st);
}

// Denoiser.forward backward from arbitrary cotangents of (eps-hat, O0-hat, posterior); null cotangents are zeros
int denoise_step_bwd(const diffab_dims* d, const diffab_denoiser_weights* w, const diffab_denoiser_weights* g, const TrainTape& tp,
                     const int64_t* seq_t, const float* x_t, const float* O_t, const float* pair_ctx, const float* post_hat,
                     const float* cot_eps, const float* cot_O0, const float* cot_post, float* d_res_ctx, float* d_pair_ctx, float* ws,
                     hipStream_t st, float* d_x_t, float* d_O_t) {
  return run_backward(BWD_COTANGENTS, d, w, g, tp, seq_t, x_t, O_t, pair_ctx, nullptr, nullptr, post_hat, nullptr, nullptr, nullptr, nullptr,
                      nullptr, nullptr, cot_eps, cot_O0, cot_post, nullptr, nullptr, d_res_ctx, d_pair_ctx, ws, st, d_x_t, d_O_t);
}

// One IPA layer backward: tp is the one-layer tape of the taped layer forward (x[0] = the layer input), dy -> dx, d pair_ctx += ..
int ipa_layer_bwd(const diffab_dims* d1, const diffab_ipa_layer_weights* lw, const diffab_ipa_layer_weights* lg, const TrainTape& tp,
                  const float* R, const float* t, const float* pair_ctx, const float* dy, float* dx, float* d_pair_ctx, float* ws,
                  hipStream_t st, float* d_R, float* d_t) {
  diffab_denoiser_weights w{}, g{};
  w.layers = lw;
  g.layers = lg;
  return run_backward(BWD_LAYER, d1, &w, &g, tp, nullptr, t, R, pair_ctx, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
                      nullptr, nullptr, nullptr, nullptr, nullptr, dy, dx, nullptr, d_pair_ctx, ws, st, d_t, d_R);
}

}  // namespace diffab
