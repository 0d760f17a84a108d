// api.hip - C-ABI entry points for the denoise step, the IPA layer and the reverse sampling loop,
// plus library plumbing (version, last error, device probe).
#include <atomic>
#include <cstring>
#include <mutex>
#include <vector>

#include "common.h"
#include "denoiser_internal.h"
#include "mlp_chain_tile.h"

namespace diffab {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

// ---- cross-stream ordering guard (common.h StreamOrder), OFF by default since round 6 ---------------------------------------------
// History (profiles/r04_two_queue.md, r05_two_queue.md): with two library pipelines on two streams of one process, heads_finish_kernel and
// reverse_update_philox_kernel - small VALU-only kernels - computed wrong values in lanes 48-63 while bf16 x 6 GEMM work-groups of the OTHER
// stream were resident.  Round 6 found the cause (profiles/r06_lanes_48_63.md, tools/hwtests/pkmul_two_streams.hip): hipcc's SLP vectoriser
// had packed their scalar code into v_pk_{mul,fma}_f32 ... op_sel:[0,1], a form gfx950 miscomputes in lanes 48-63 while f16 / bf16 MFMAs of
// ANY wave - another kernel's included - are in flight on the SIMD.  The form is gone from every kernel of the library (Makefile NOSLP,
// tools/isa_hazard_lint.py), two pipelines on two streams measure bitwise the sequential runs with the guard off, so the guard is now an
// opt-in (diffab_set_stream_guard(1)): one state per device = {last stream, event}.
namespace {
struct OrderState {
  std::recursive_mutex mu;
  hipStream_t last = nullptr;
  bool have = false;
  hipEvent_t ev = nullptr;
  int depth = 0;
};
OrderState g_order[32];
std::atomic<bool> g_order_on{false};
}  // namespace

StreamOrder::StreamOrder(void* stream) : dev_(-1) {
  if (!g_order_on.load(std::memory_order_relaxed)) return;  // guard off (default): nothing is held, calls of host threads do not serialise
  if (hipGetDevice(&dev_) != hipSuccess) dev_ = 0;
  OrderState& o = g_order[dev_ & 31];
  o.mu.lock();
  if (o.depth++ > 0) return;  // an entry point called from another entry point: already ordered
  hipStream_t st = as_stream(stream);
  if (o.have && o.last != st) {
    // everything enqueued on the previous stream so far (the library's last call, and whatever the caller put behind it) comes first
    if (o.ev == nullptr && hipEventCreateWithFlags(&o.ev, hipEventDisableTiming) != hipSuccess) o.ev = nullptr;
    if (o.ev != nullptr && hipEventRecord(o.ev, o.last) == hipSuccess) (void)hipStreamWaitEvent(st, o.ev, 0);
    (void)hipGetLastError();  // a stream the caller destroyed meanwhile: nothing left to order against
  }
  o.last = st;
  o.have = true;
}
StreamOrder::~StreamOrder() {
  if (dev_ < 0) return;
  OrderState& o = g_order[dev_ & 31];
  --o.depth;
  o.mu.unlock();
}
void set_stream_order(bool on) { g_order_on.store(on); }

// ---- opt-in launch timer for the dominant kernel (bench.py's roofline leg) ---------------------------------
// When enabled, the attention-kernel launchers bracket each launch with a hipEvent pair recorded on the launch
// stream; diffab_kernel_timer_read() synchronises the events and returns the launch count and the summed time.
struct KernelTimer {
  bool on = false;
  std::vector<hipEvent_t> ev;  // start/stop pairs
  size_t used = 0;
};
static KernelTimer g_timer;

void timer_begin(hipStream_t st) {
  if (!g_timer.on) return;
  if (g_timer.used + 2 > g_timer.ev.size()) {
    for (int i = 0; i < 2; ++i) {
      hipEvent_t e;
      if (hipEventCreate(&e) != hipSuccess) return;
      g_timer.ev.push_back(e);
    }
  }
  (void)hipEventRecord(g_timer.ev[g_timer.used], st);
}
bool kernel_timer_enabled() { return g_timer.on; }
void timer_end(hipStream_t st) {
  if (!g_timer.on || g_timer.used + 2 > g_timer.ev.size()) return;
  (void)hipEventRecord(g_timer.ev[g_timer.used + 1], st);
  g_timer.used += 2;
}

static int check_dims(const diffab_dims* d, const char* who) {
  DIFFAB_REQUIRE(d != nullptr, DIFFAB_ERR_ARG, "%s: dims is null", who);
  // C == 0: an IPA layer built with use_pair_bias = False (reference :348-385) - no pair bias, two independent logits, no o_pair block
  // in the feature row; the layer entries take it on the any-dims path (e and w_bias may be NULL), the Denoiser always has C > 0
  DIFFAB_REQUIRE(d->B > 0 && d->K > 0 && d->D > 0 && d->C >= 0 && d->H > 0 && d->DS > 0 && d->PQ > 0 && d->PV > 0 && d->NL >= 0 && d->V > 0,
                 DIFFAB_ERR_ARG, "%s: non-positive dimension (B=%d K=%d D=%d C=%d H=%d DS=%d PQ=%d PV=%d NL=%d V=%d)", who, d->B, d->K, d->D,
                 d->C, d->H, d->DS, d->PQ, d->PV, d->NL, d->V);
  DIFFAB_REQUIRE(static_cast<int64_t>(d->B) * d->K < (1ll << 31), DIFFAB_ERR_UNSUPPORTED, "%s: B*K must be < 2^31", who);
  return DIFFAB_OK;
}

struct StepBuffers {
  float *cat2, *h1, *hA, *hB, *cat3, *t1, *t2, *vbuf, *logits, *ipa, *emb_tab, *beta_tab;
  char* planes;  // split bf16 planes of the dense weights (MFMA path): NL x ipa_layer_planes_bytes(), then 11 MLP matrices
  float* pair;   // fp16 planes of the pair embedding (launch_pair_split), null when the fused kernel cannot take them
  size_t bytes;
};
static size_t mlp_planes_bytes() { return (rowgemm128_b6_scratch_bytes(128) + 255) & ~static_cast<size_t>(255); }

static StepBuffers carve_step(const diffab_dims* d, void* ws) {
  Carver c(ws);
  const size_t rows = static_cast<size_t>(d->B) * d->K;
  StepBuffers b;
  b.cat2 = c.take<float>(rows * 2 * d->D);
  b.h1 = c.take<float>(rows * d->D);
  b.hA = c.take<float>(rows * d->D);
  b.hB = c.take<float>(rows * d->D);
  b.cat3 = c.take<float>(rows * (d->D + 3));
  b.t1 = c.take<float>(rows * d->D);
  b.t2 = c.take<float>(rows * d->D);
  b.vbuf = c.take<float>(rows * 3);
  b.logits = c.take<float>(rows * d->V);
  b.emb_tab = c.take<float>(static_cast<size_t>(25) * d->D);
  b.beta_tab = c.take<float>(static_cast<size_t>(3) * d->B * d->D);
  size_t ipa_floats = ipa_generic_workspace_floats(d);
  if (fast_path_supported(d)) ipa_floats = ipa_floats > ipa_fast_workspace_floats(d) ? ipa_floats : ipa_fast_workspace_floats(d);
  b.ipa = c.take<float>(ipa_floats);
  b.planes = fast_path_supported(d) ? c.take<char>(d->NL * ipa_layer_planes_bytes() + 11 * mlp_planes_bytes()) : nullptr;
  b.pair = pair_planes_supported(d) ? c.take<float>(pair_planes_floats(d)) : nullptr;
  b.bytes = c.bytes();
  return b;
}

static int ipa_layer_dispatch(const diffab_dims* d, const diffab_ipa_layer_weights* w, const float* x, const float* e, const float* R,
                              const float* t, float* y, float* ws, uint32_t flags, hipStream_t st, float* sp_keep = nullptr,
                              float* d2_keep = nullptr, const void* planes = nullptr, const float* pair_planes = nullptr,
                              bool taped = false,  // taped: ws is a slot of the training tape (the backward reads proj and feat)
                              const unsigned char* tile_needed = nullptr) {
  DIFFAB_REQUIRE(w && w->gamma && w->wq_s && w->wk_s && w->wv_s && (w->w_bias || d->C == 0) && w->wq_p && w->wk_p && w->wv_p && w->w_out &&
                     w->b_out,
                 DIFFAB_ERR_ARG, "ipa layer: null weight pointer");
  if (!(flags & DIFFAB_FLAG_FORCE_GENERIC) && fast_path_supported(d))
    return ipa_layer_fast(d, w, x, e, R, t, y, ws, st, sp_keep, d2_keep, planes, pair_planes, (flags & DIFFAB_FLAG_FP32_GEMM) != 0,
                          taped, tile_needed);
  return ipa_layer_generic(d, w, x, e, R, t, y, ws, st);
}

static int mlp3(const diffab_dims* d, const diffab_mlp3_weights* w, const float* cat3, float* t1, float* t2, float* out, int n_out,
                hipStream_t st) {
  DIFFAB_REQUIRE(w->w0 && w->b0 && w->w2 && w->b2 && w->w4 && w->b4, DIFFAB_ERR_ARG, "denoiser head: null weight pointer");
  const int rows = d->B * d->K, D = d->D;
  if (int rc = launch_linear(cat3, D + 3, w->w0, w->b0, t1, D, rows, D, D + 3, true, st)) return rc;
  if (int rc = launch_linear(t1, D, w->w2, w->b2, t2, D, rows, D, D, true, st)) return rc;
  return launch_linear(t2, D, w->w4, w->b4, out, n_out, rows, n_out, D, false, st);
}

static bool use_pair_planes(const diffab_dims* d, uint32_t flags, const float* pair_ctx, const StepBuffers& b) {
  const uint32_t other = DIFFAB_FLAG_FORCE_GENERIC | DIFFAB_FLAG_PAIR_F32;
  return (flags & DIFFAB_FLAG_PAIR_PLANES) && !(flags & other) && b.pair != nullptr && pair_planes_supported(d) &&
         (reinterpret_cast<uintptr_t>(pair_ctx) & 15) == 0;
}

// Everything on the folded MFMA path that depends on the weights only: the sequence-embedding bias table and the split bf16 planes
// of every dense weight matrix.  Once per denoise_step call - or once per trajectory (diffab_sample_loop).
static int prepare_weights(const diffab_dims* d, const diffab_denoiser_weights* w, const StepBuffers& b, uint32_t flags, hipStream_t st) {
  DIFFAB_REQUIRE(w->coord.w0 && w->coord.b0 && w->orient.w0 && w->orient.b0 && w->seq.w0 && w->seq.b0 && w->coord.w2 && w->orient.w2 &&
                     w->seq.w2,
                 DIFFAB_ERR_ARG, "denoiser head: null weight pointer");
  if (int rc = launch_fold_tables(d, w, nullptr, b.emb_tab, nullptr, st)) return rc;
  if (!use_b6_gemm(flags)) return DIFFAB_OK;
  const int D = d->D;
  for (int l = 0; l < d->NL; ++l)
    if (int rc = ipa_layer_split_weights(&w->layers[l], b.planes + l * ipa_layer_planes_bytes(), st)) return rc;
  char* mlp = b.planes + d->NL * ipa_layer_planes_bytes();
  const diffab_mlp3_weights* hw[3] = {&w->coord, &w->orient, &w->seq};
  if (int rc = launch_wsplit128(w->res_w0, 2 * D, D, mlp, st)) return rc;                        // slot 0: res_ctx half of to_res_emb[0]
  if (int rc = launch_wsplit128(w->res_w2, D, D, mlp + mlp_planes_bytes(), st)) return rc;      // slot 1
  for (int hd = 0; hd < 3; ++hd) {                                                              // slots 2 + 2 hd, 3 + 2 hd
    if (int rc = launch_wsplit128(hw[hd]->w0, D + 3, D, mlp + (2 + 2 * hd) * mlp_planes_bytes(), st)) return rc;
    if (int rc = launch_wsplit128(hw[hd]->w2, D, D, mlp + (3 + 2 * hd) * mlp_planes_bytes(), st)) return rc;
    DIFFAB_REQUIRE(hw[hd]->w4, DIFFAB_ERR_ARG, "denoiser head: null weight pointer");
    const int nout = hd == 2 ? d->V : 3;                                                          // slot 8 + hd: the narrow last layer
    if (int rc = launch_wsplit128(hw[hd]->w4, D, D, mlp + (8 + hd) * mlp_planes_bytes(), st, nout)) return rc;
  }
  return DIFFAB_OK;
}

static int denoise_step(const diffab_dims* d, const diffab_denoiser_weights* w, const int64_t* seq_t, const float* x_t, const float* O_t,
                        const float* res_ctx, const float* pair_ctx, const float* beta, float* out_eps, float* out_O0, float* out_post,
                        float* out_logits, float* out_res_emb, void* ws, uint32_t flags, hipStream_t st, bool weights_prepared = false,
                        bool pair_prepared = false, const float* sched_beta = nullptr, int t_step = 0, const int* t_dev = nullptr,
                        const unsigned char* last_layer_tiles = nullptr,  // row tiles of the LAST layer whose outputs are read
                        bool skip_heads_finish = false,  // the caller finishes the heads itself from b.vbuf / b.logits (reverse sampler)
                        const float* beta_traj = nullptr, int traj_rows = 0) {  // [3 heads][traj_rows steps][D]: the heads' folded beta
                                                                               // columns of EVERY step (row t_step is this step's)
  // sched_beta (reverse sampler): every patch is at step t_step (or *t_dev): the folded head tables take beta from the schedule and
  // `beta` is only read by the unfolded path
  const StepBuffers b = carve_step(d, ws);
  const int rows = d->B * d->K, D = d->D;
  // DIFFAB_FLAG_PAIR_PLANES: the pair embedding as two fp16 planes (same bytes), attention's pair-tile products on the f16 matrix
  // cores.  The reverse sampler splits once per trajectory (pair_prepared); a single call splits here, per call.
  const float* pair_planes = nullptr;
  if (use_pair_planes(d, flags, pair_ctx, b)) {
    if (!pair_prepared)
      if (int rc = launch_pair_split(d, pair_ctx, b.pair, st)) return rc;
    pair_planes = b.pair;
  }
  // Folded concatenations (MFMA path): the sequence-embedding half of to_res_emb[0] and the beta-embedding columns of the three
  // head MLPs become bias tables, so neither cat[res_ctx, E[s]] nor cat[h, tau] is written or re-read.
  const bool fold = !(flags & DIFFAB_FLAG_FORCE_GENERIC) && fast_path_supported(d) && rowgemm128_ok(res_ctx, D, b.h1, D, rows, D);
  // dense N = 128 layers of the folded path: bf16x6 from the prepared planes (slot), or the fp32 kernel; each MLP as one row-resident
  // kernel (mlp_chain_b6_kernel), one launch per dense layer only where the chain does not apply
  const bool b6 = fold && use_b6_gemm(flags) && rowgemm128_b6_ok(res_ctx, D, b.h1, D, rows, D);
  const bool chain = b6 && d->V <= 128;
  if (!chain) beta_traj = nullptr;  // (the per-call table of every step's head columns is read by the chain kernel only)
  if (fold) {
    DIFFAB_REQUIRE(w->coord.w0 && w->coord.b0 && w->orient.w0 && w->orient.b0 && w->seq.w0 && w->seq.b0, DIFFAB_ERR_ARG,
                   "denoiser head: null weight pointer");
    if (!weights_prepared)
      if (int rc = prepare_weights(d, w, b, flags, st)) return rc;
    if (beta_traj == nullptr)
      if (int rc = launch_fold_tables(d, w, beta, b.emb_tab, b.beta_tab, st, true, sched_beta, t_step, t_dev)) return rc;
  }
  const char* mlp = b6 ? b.planes + d->NL * ipa_layer_planes_bytes() : nullptr;
  auto dense128 = [&](int slot, const float* X, const float* W, int ldw, const float* bias, const int64_t* bias_idx, int bias_div, float* Y,
                      bool relu) -> int {
    if (b6) return launch_rowgemm128_b6p(X, D, mlp + slot * mlp_planes_bytes(), bias, bias_idx, bias_div, Y, D, rows, D, relu, st);
    return launch_rowgemm128(X, D, W, ldw, bias, bias_idx, bias_div, Y, D, rows, D, relu, st);
  };
  // DIFFAB_FLAG_PERSISTENT_MODULE: the NL layers as one patch-resident launch (ipa_persistent.hip) - the same tile bodies, bitwise the
  // same result; needs the prepared planes of all layers, the pair planes, and K = 128.  Where the MLP chains apply too, the embedding
  // MLP and the three heads run as phases of that launch (fused_mlps): one launch per step in front of the state update.
  const bool persistent = (flags & DIFFAB_FLAG_PERSISTENT_MODULE) && fold && use_b6_gemm(flags) && pair_planes != nullptr &&
                          ipa_module_persistent_supported(d) && last_layer_tiles == nullptr;
  const bool fused_mlps = persistent && chain && D == 128 && out_res_emb == nullptr;
  float* logits = out_logits ? out_logits : b.logits;
  MlpChainSet emb_set{}, head_set{};
  if (fold && chain) {
    const void* pl[3] = {mlp, mlp + mlp_planes_bytes(), nullptr};
    const float* bs[3] = {b.emb_tab, w->res_b2, nullptr};
    float* ys[1] = {b.hA};
    const int nout1[1] = {D};
    if (int rc = make_mlp_chain_set(&emb_set, 1, pl, bs, seq_t, 0, 2, nout1, ys, nout1)) return rc;
    if (!fused_mlps)
      if (int rc = launch_mlp_chain_b6(res_ctx, D, pl, bs, seq_t, 0, 2, D, b.hA, D, rows, st)) return rc;
  } else if (fold) {
    if (int rc = dense128(0, res_ctx, w->res_w0, 2 * D, b.emb_tab, seq_t, 0, b.h1, true)) return rc;
    if (int rc = dense128(1, b.h1, w->res_w2, D, w->res_b2, nullptr, 0, b.hA, false)) return rc;
  } else {
    if (int rc = launch_embed_concat(res_ctx, w->seq_emb, seq_t, D, rows, b.cat2, st)) return rc;
    if (int rc = launch_linear(b.cat2, 2 * D, w->res_w0, w->res_b0, b.h1, D, rows, D, 2 * D, true, st)) return rc;
    if (int rc = launch_linear(b.h1, D, w->res_w2, w->res_b2, b.hA, D, rows, D, D, false, st)) return rc;
  }
  float *cur = b.hA, *nxt = b.hB;
  const diffab_mlp3_weights* hw[3] = {&w->coord, &w->orient, &w->seq};
  float* outs[3] = {out_eps, b.vbuf, logits};
  const int nout[3] = {3, 3, d->V};
  const void* hpl[9];
  const float* hbs[9];
  if (fold && chain) {  // the three heads read the same rows: one launch (blockIdx.y = head), or three phases of the module launch
    for (int hd = 0; hd < 3; ++hd) {
      DIFFAB_REQUIRE(hw[hd]->w2 && hw[hd]->b2 && hw[hd]->w4 && hw[hd]->b4, DIFFAB_ERR_ARG, "denoiser head: null weight pointer");
      hpl[3 * hd] = mlp + (2 + 2 * hd) * mlp_planes_bytes();
      hpl[3 * hd + 1] = mlp + (3 + 2 * hd) * mlp_planes_bytes();
      hpl[3 * hd + 2] = mlp + (8 + hd) * mlp_planes_bytes();
      hbs[3 * hd] = beta_traj ? beta_traj + (static_cast<size_t>(hd) * traj_rows + t_step) * D : b.beta_tab + static_cast<size_t>(hd) * d->B * D;
      hbs[3 * hd + 1] = hw[hd]->b2;
      hbs[3 * hd + 2] = hw[hd]->b4;
    }
    // (beta_traj: one table row for every patch - "row / rows" is 0 for all of them)
    if (int rc = make_mlp_chain_set(&head_set, 3, hpl, hbs, nullptr, beta_traj ? rows : d->K, 3, nout, outs, nout)) return rc;
  }
  if (persistent) {
    if (int rc = launch_ipa_module_persistent(d, b.hA, b.hB, O_t, x_t, b.ipa, b.planes, pair_planes, st, fused_mlps ? res_ctx : nullptr,
                                              &emb_set, &head_set))
      return rc;
    cur = (d->NL & 1) ? b.hB : b.hA;
  }
  for (int l = 0; l < d->NL && !persistent; ++l) {
    const void* planes = (fold && use_b6_gemm(flags)) ? b.planes + l * ipa_layer_planes_bytes() : nullptr;
    if (int rc = ipa_layer_dispatch(d, &w->layers[l], cur, pair_ctx, O_t, x_t, nxt, b.ipa, flags, st, nullptr, nullptr, planes, pair_planes,
                                    false, l == d->NL - 1 ? last_layer_tiles : nullptr))
      return rc;
    float* tmp = cur; cur = nxt; nxt = tmp;
  }
  if (out_res_emb) DIFFAB_HIP_CHECK(hipMemcpyAsync(out_res_emb, cur, sizeof(float) * rows * D, hipMemcpyDeviceToDevice, st));
  if (fold) {
    for (int hd = 0; hd < 3; ++hd)
      DIFFAB_REQUIRE(hw[hd]->w2 && hw[hd]->b2 && hw[hd]->w4 && hw[hd]->b4, DIFFAB_ERR_ARG, "denoiser head: null weight pointer");
    if (chain && !fused_mlps)
      if (int rc = launch_mlp_chains_b6(cur, D, 3, hpl, hbs, nullptr, beta_traj ? rows : d->K, 3, nout, outs, nout, rows, st)) return rc;
    for (int hd = 0; hd < 3 && !chain; ++hd) {
      if (int rc = dense128(2 + 2 * hd, cur, hw[hd]->w0, D + 3, b.beta_tab + static_cast<size_t>(hd) * d->B * D, nullptr, d->K, b.t1, true))
        return rc;
      if (int rc = dense128(3 + 2 * hd, b.t1, hw[hd]->w2, D, hw[hd]->b2, nullptr, 0, b.t2, true)) return rc;
      if (int rc = launch_linear(b.t2, D, hw[hd]->w4, hw[hd]->b4, outs[hd], nout[hd], rows, nout[hd], D, false, st)) return rc;
    }
  } else {
    if (int rc = launch_beta_concat(cur, beta, D, d->K, rows, b.cat3, st)) return rc;
    if (int rc = mlp3(d, &w->coord, b.cat3, b.t1, b.t2, out_eps, 3, st)) return rc;
    if (int rc = mlp3(d, &w->orient, b.cat3, b.t1, b.t2, b.vbuf, 3, st)) return rc;
    if (int rc = mlp3(d, &w->seq, b.cat3, b.t1, b.t2, logits, d->V, st)) return rc;
  }
  if (skip_heads_finish) return DIFFAB_OK;
  return launch_heads_finish(b.vbuf, O_t, logits, d->V, rows, out_O0, out_post, st);
}

// The taped forwards and their backwards are one unit: where the MFMA path applies the backward reads the probabilities and squared
// distances the three-launch attention left on the tape, so a taped forward must not be diverted to the generic kernels (which do not
// write them) - DIFFAB_FLAG_FORCE_GENERIC is ignored by diffab_train_step_fwd / diffab_denoise_step_fwd_taped /
// diffab_ipa_layer_fwd_taped (unit dims take the generic kernels on both sides anyway); the arithmetic selectors (FP32_GEMM) pass.
static uint32_t taped_flags(uint32_t flags) { return flags & ~(DIFFAB_FLAG_FORCE_GENERIC | DIFFAB_FLAG_PAIR_PLANES); }

// Training forward: the same launches as denoise_step, but every intermediate lands in its own slot of the tape.
static int denoise_step_taped(const diffab_dims* d, const diffab_denoiser_weights* w, const int64_t* seq_t, const float* x_t,
                              const float* O_t, const float* res_ctx, const float* pair_ctx, const float* beta, float* out_eps,
                              float* out_O0, float* out_post, const TrainTape& tp, uint32_t flags, hipStream_t st) {
  const int rows = d->B * d->K, D = d->D;
  if (int rc = launch_embed_concat(res_ctx, w->seq_emb, seq_t, D, rows, tp.cat2, st)) return rc;
  if (int rc = launch_linear(tp.cat2, 2 * D, w->res_w0, w->res_b0, tp.h1, D, rows, D, 2 * D, true, st)) return rc;
  if (int rc = launch_linear(tp.h1, D, w->res_w2, w->res_b2, tp.x[0], D, rows, D, D, false, st)) return rc;
  const bool b6 = tp.planes && use_b6_gemm(flags) && !(flags & DIFFAB_FLAG_FORCE_GENERIC) && fast_path_supported(d);
  for (int l = 0; l < d->NL; ++l) {
    if (b6)
      if (int rc = ipa_layer_split_weights(&w->layers[l], tp.planes, st)) return rc;
    if (int rc = ipa_layer_dispatch(d, &w->layers[l], tp.x[l], pair_ctx, O_t, x_t, tp.x[l + 1], tp.ipa_ws[l], flags, st, tp.sp[l], tp.d2[l],
                                    b6 ? tp.planes : nullptr, nullptr, true))
      return rc;
  }
  if (int rc = launch_beta_concat(tp.x[d->NL], beta, D, d->K, rows, tp.cat3, st)) return rc;
  const diffab_mlp3_weights* hw[3] = {&w->coord, &w->orient, &w->seq};
  float* outs[3] = {out_eps, tp.vbuf, tp.logits};
  const int nout[3] = {3, 3, d->V};
  for (int hd = 0; hd < 3; ++hd) {
    DIFFAB_REQUIRE(hw[hd]->w0 && hw[hd]->b0 && hw[hd]->w2 && hw[hd]->b2 && hw[hd]->w4 && hw[hd]->b4, DIFFAB_ERR_ARG,
                   "denoiser head: null weight pointer");
    if (int rc = launch_linear(tp.cat3, D + 3, hw[hd]->w0, hw[hd]->b0, tp.t1[hd], D, rows, D, D + 3, true, st)) return rc;
    if (int rc = launch_linear(tp.t1[hd], D, hw[hd]->w2, hw[hd]->b2, tp.t2[hd], D, rows, D, D, true, st)) return rc;
    if (int rc = launch_linear(tp.t2[hd], D, hw[hd]->w4, hw[hd]->b4, outs[hd], nout[hd], rows, nout[hd], D, false, st)) return rc;
  }
  return launch_heads_finish(tp.vbuf, O_t, tp.logits, d->V, rows, out_O0, out_post, st);
}

static int check_denoiser_weights(const diffab_dims* d, const diffab_denoiser_weights* w) {
  DIFFAB_REQUIRE(w && w->seq_emb && w->res_w0 && w->res_b0 && w->res_w2 && w->res_b2 && (d->NL == 0 || w->layers), DIFFAB_ERR_ARG,
                 "denoiser: null weight pointer");
  DIFFAB_REQUIRE(d->C > 0, DIFFAB_ERR_ARG, "denoiser: C must be positive (the Denoiser's IPA layers use the pair bias, :478-492)");
  return DIFFAB_OK;
}

constexpr int kTrajRows = 1025;  // schedules up to T = 1024 get their per-step head tables built once per call
struct SampleBuffers {
  float *beta, *eps, *O0, *post;
  int* t_dev;  // the current timestep in device memory (graph replay)
  unsigned char* tiles;  // [B][K / 16]: row tiles with a generated residue (DIFFAB_FLAG_SKIP_UNUSED_ROWS)
  float* beta_traj;      // [3 heads][kTrajRows][D]: the heads' folded beta columns of every step of a schedule with T < kTrajRows
  void* step;
  size_t bytes;
};

static SampleBuffers carve_sample(const diffab_dims* d, void* ws) {
  Carver c(ws);
  const size_t rows = static_cast<size_t>(d->B) * d->K;
  SampleBuffers s;
  s.beta = c.take<float>(d->B);
  s.eps = c.take<float>(rows * 3);
  s.O0 = c.take<float>(rows * 9);
  s.post = c.take<float>(rows * d->V);
  s.t_dev = c.take<int>(64);
  s.tiles = c.take<unsigned char>(static_cast<size_t>(d->B) * ((d->K + 15) / 16));
  s.beta_traj = c.take<float>(static_cast<size_t>(3) * kTrajRows * d->D);
  const size_t step_bytes = carve_step(d, nullptr).bytes;
  s.step = c.take<char>(step_bytes);
  s.bytes = c.bytes();
  return s;
}

}  // namespace diffab

using namespace diffab;

extern "C" {

const char* diffab_version(void) { return "diffab_hip 0.1.0 (gfx950)"; }
const char* diffab_last_error(void) { return g_err; }

int diffab_device_ok(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return 0;
  hipDeviceProp_t p;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&p, dev) != hipSuccess) return 0;
  return std::strncmp(p.gcnArchName, "gfx950", 6) == 0 ? 1 : 0;
}

int diffab_debug_linear128(const float* X, const float* W, const float* bias, float* Y, int64_t M, int32_t Kd, int32_t mode, void* scratch,
                           size_t scratch_bytes, void* stream) {
  StreamOrder order_(stream);
  DIFFAB_REQUIRE(X && W && Y && M >= 1 && M < (1LL << 31) && Kd >= 32 && Kd % 32 == 0, DIFFAB_ERR_ARG, "debug_linear128: bad operands");
  hipStream_t st = as_stream(stream);
  if (mode == 0) return launch_linear(X, Kd, W, bias, Y, 128, static_cast<int>(M), 128, Kd, false, st);  // rowgemm128 / tiled f32 MFMA
  if (mode == 2) {  // fp16 x 3 (gemm_f16x3.hip): planes | 1 / scale of the 128 weight rows
    const size_t pb = (rowgemm128_h3_planes_bytes(Kd) + 255) & ~static_cast<size_t>(255);
    DIFFAB_REQUIRE(Kd % 64 == 0, DIFFAB_ERR_ARG, "debug_linear128: mode 2 (fp16 x 3) needs Kd %% 64 == 0 (its chunks of 32 k are joined in pairs)");
    DIFFAB_REQUIRE(scratch && scratch_bytes >= pb + 512 && rowgemm128_b6_ok(X, Kd, Y, 128, static_cast<int>(M), Kd) &&
                       (reinterpret_cast<uintptr_t>(scratch) & 15) == 0,
                   DIFFAB_ERR_ARG, "debug_linear128: mode 2 needs 16-byte aligned operands and %zu bytes of scratch", pb + 512);
    float* wis = reinterpret_cast<float*>(static_cast<char*>(scratch) + pb);
    if (int rc = launch_wsplit128_h3(W, Kd, Kd, scratch, wis, st)) return rc;
    return launch_rowgemm128_h3p(X, Kd, scratch, wis, bias, nullptr, 0, Y, 128, static_cast<int>(M), Kd, false, st, nullptr);
  }
  DIFFAB_REQUIRE(mode == 1 && scratch && scratch_bytes >= rowgemm128_b6_scratch_bytes(Kd) && rowgemm128_b6_ok(X, Kd, Y, 128, static_cast<int>(M), Kd),
                 DIFFAB_ERR_ARG, "debug_linear128: mode 1 needs 16-byte aligned operands and %zu bytes of scratch", rowgemm128_b6_scratch_bytes(Kd));
  return launch_rowgemm128_b6(X, Kd, W, Kd, bias, nullptr, 0, Y, 128, static_cast<int>(M), Kd, false, scratch, st);
}

int diffab_debug_xstat128(const float* X, const float* W, float* Y, int64_t M, int32_t N, int32_t mode, void* scratch, size_t scratch_bytes,
                          void* stream) {
  StreamOrder order_(stream);
  DIFFAB_REQUIRE(X && W && Y && scratch && M >= 1 && M < (1LL << 31) && N >= 1 && (mode == 1 || mode == 2), DIFFAB_ERR_ARG,
                 "debug_xstat128: bad operands");
  const size_t need = mode == 1 ? xstat_b6_scratch_bytes(N) : xstat_h3_scratch_bytes(N);
  DIFFAB_REQUIRE(scratch_bytes >= need && (reinterpret_cast<uintptr_t>(scratch) & 15) == 0 && (reinterpret_cast<uintptr_t>(X) & 15) == 0,
                 DIFFAB_ERR_ARG, "debug_xstat128: needs 16-byte aligned X / scratch and %zu bytes of scratch", need);
  hipStream_t st = as_stream(stream);
  if (mode == 1) return launch_xstat_b6(X, W, 1, N, Y, N, static_cast<int>(M), N, scratch, st);
  return launch_xstat_h3(X, W, 1, N, Y, N, static_cast<int>(M), N, scratch, st);
}

int diffab_debug_gemm_tn(const float* A, const float* B, float* C, float* db, int64_t M, int32_t N1, int32_t N2, int32_t mode, void* stream) {
  StreamOrder order_(stream);
  DIFFAB_REQUIRE(A && B && C && M >= 1 && M < (1LL << 31) && N1 >= 1 && N2 >= 1 && (mode == 1 || mode == 2), DIFFAB_ERR_ARG,
                 "debug_gemm_tn: bad operands");
  hipStream_t st = as_stream(stream);
  if (mode == 1) return launch_gemm_tn_b6(A, N1, B, N2, C, N2, static_cast<int>(M), N1, N2, db, nullptr, nullptr, 0, st);
  return launch_gemm_tn_h3(A, N1, B, N2, C, N2, static_cast<int>(M), N1, N2, db, nullptr, nullptr, 0, st);
}

int diffab_set_stream_guard(int on) {
  set_stream_order(on != 0);
  return DIFFAB_OK;
}

int diffab_debug_set_module_stagger(int32_t ticks_10ns, int32_t classes) {
  set_module_stagger(ticks_10ns, classes);
  return DIFFAB_OK;
}

int diffab_debug_set_module_stamps(void* device_buffer) {
  set_module_stamps(device_buffer);
  return DIFFAB_OK;
}

int diffab_debug_set_attn_variant(int32_t v) {
  set_attn_variant(v);
  return DIFFAB_OK;
}

int diffab_debug_set_attn_stamps(void* device_buffer) {
  set_attn_stamps(device_buffer);
  return DIFFAB_OK;
}

int diffab_kernel_timer_enable(int on) {
  g_timer.on = on != 0;
  g_timer.used = 0;
  return DIFFAB_OK;
}

int diffab_kernel_timer_read(int64_t* launches, double* total_ms) {
  DIFFAB_REQUIRE(launches && total_ms, DIFFAB_ERR_ARG, "kernel_timer_read: null pointer");
  double tot = 0.0;
  for (size_t i = 0; i + 1 < g_timer.used; i += 2) {
    DIFFAB_HIP_CHECK(hipEventSynchronize(g_timer.ev[i + 1]));
    float ms = 0.f;
    DIFFAB_HIP_CHECK(hipEventElapsedTime(&ms, g_timer.ev[i], g_timer.ev[i + 1]));
    tot += ms;
  }
  *launches = static_cast<int64_t>(g_timer.used / 2);
  *total_ms = tot;
  g_timer.used = 0;
  return DIFFAB_OK;
}

size_t diffab_denoise_workspace_bytes(const diffab_dims* d) {
  if (check_dims(d, "denoise_workspace_bytes")) return 0;
  return carve_step(d, nullptr).bytes;
}

size_t diffab_sample_workspace_bytes(const diffab_dims* d) {
  if (check_dims(d, "sample_workspace_bytes")) return 0;
  return carve_sample(d, nullptr).bytes;
}

int diffab_ipa_layer_fwd(const diffab_dims* d, const diffab_ipa_layer_weights* w, const float* x, const float* e, const float* R,
                         const float* t, float* y, void* workspace, size_t workspace_bytes, uint32_t flags, void* stream) {
  StreamOrder order_(stream);
  if (int rc = check_dims(d, "ipa_layer_fwd")) return rc;
  DIFFAB_REQUIRE(x && (e || d->C == 0) && R && t && y && workspace, DIFFAB_ERR_ARG, "ipa_layer_fwd: null pointer");
  const StepBuffers b = carve_step(d, workspace);
  DIFFAB_REQUIRE(workspace_bytes >= b.bytes, DIFFAB_ERR_WORKSPACE, "ipa_layer_fwd: workspace %zu < %zu bytes", workspace_bytes, b.bytes);
  const float* pair_planes = nullptr;
  if (use_pair_planes(d, flags, e, b)) {
    if (int rc = launch_pair_split(d, e, b.pair, as_stream(stream))) return rc;
    pair_planes = b.pair;
  }
  return ipa_layer_dispatch(d, w, x, e, R, t, y, b.ipa, flags, as_stream(stream), nullptr, nullptr, nullptr, pair_planes);
}

int diffab_denoise_step_fwd(const diffab_dims* d, const diffab_denoiser_weights* w, const int64_t* seq_t, const float* x_t, const float* O_t,
                            const float* res_ctx, const float* pair_ctx, const float* beta, float* out_eps, float* out_O0,
                            float* out_posterior, float* out_logits, float* out_res_emb, void* workspace, size_t workspace_bytes,
                            uint32_t flags, void* stream) {
  StreamOrder order_(stream);
  if (int rc = check_dims(d, "denoise_step_fwd")) return rc;
  if (int rc = check_denoiser_weights(d, w)) return rc;
  DIFFAB_REQUIRE(seq_t && x_t && O_t && res_ctx && pair_ctx && beta && out_eps && out_O0 && out_posterior && workspace, DIFFAB_ERR_ARG,
                 "denoise_step_fwd: null pointer");
  const size_t need = carve_step(d, nullptr).bytes;
  DIFFAB_REQUIRE(workspace_bytes >= need, DIFFAB_ERR_WORKSPACE, "denoise_step_fwd: workspace %zu < %zu bytes", workspace_bytes, need);
  return denoise_step(d, w, seq_t, x_t, O_t, res_ctx, pair_ctx, beta, out_eps, out_O0, out_posterior, out_logits, out_res_emb, workspace,
                      flags, as_stream(stream));
}

size_t diffab_train_tape_bytes(const diffab_dims* d) {
  if (check_dims(d, "train_tape_bytes") || d->NL > kMaxLayers) return 0;
  return train_tape_floats(d) * sizeof(float);
}

size_t diffab_train_workspace_bytes(const diffab_dims* d) {
  if (check_dims(d, "train_workspace_bytes")) return 0;
  return train_bwd_workspace_floats(d) * sizeof(float);
}

int diffab_train_step_fwd(const diffab_dims* d, const diffab_denoiser_weights* w, const int64_t* seq_t, const float* x_t, const float* O_t,
                          const float* res_ctx, const float* pair_ctx, const float* beta, const float* true_post, const float* true_eps,
                          const float* true_O0, const uint8_t* gen_mask, const uint8_t* res_mask, float* out_eps, float* out_O0,
                          float* out_posterior, float* losses3, void* tape, size_t tape_bytes, uint32_t flags, void* stream) {
  StreamOrder order_(stream);
  if (int rc = check_dims(d, "train_step_fwd")) return rc;
  if (int rc = check_denoiser_weights(d, w)) return rc;
  DIFFAB_REQUIRE(d->NL <= kMaxLayers, DIFFAB_ERR_UNSUPPORTED, "train_step_fwd: at most %d IPA layers", kMaxLayers);
  DIFFAB_REQUIRE(seq_t && x_t && O_t && res_ctx && pair_ctx && beta && true_post && true_eps && true_O0 && gen_mask && res_mask && out_eps &&
                     out_O0 && out_posterior && losses3 && tape,
                 DIFFAB_ERR_ARG, "train_step_fwd: null pointer");
  DIFFAB_REQUIRE(tape_bytes >= train_tape_floats(d) * sizeof(float), DIFFAB_ERR_WORKSPACE, "train_step_fwd: tape %zu < %zu bytes", tape_bytes,
                 train_tape_floats(d) * sizeof(float));
  const TrainTape tp = carve_tape(d, static_cast<float*>(tape));
  hipStream_t st = as_stream(stream);
  flags = taped_flags(flags);
  if (int rc = denoise_step_taped(d, w, seq_t, x_t, O_t, res_ctx, pair_ctx, beta, out_eps, out_O0, out_posterior, tp, flags, st)) return rc;
  return launch_losses_fwd(out_posterior, true_post, out_eps, true_eps, out_O0, true_O0, gen_mask, res_mask, d->B, d->K, d->V, losses3, st,
                           tp.scratch);
}

int diffab_train_step_bwd(const diffab_dims* d, const diffab_denoiser_weights* w, const diffab_denoiser_weights* grads,
                          const int64_t* seq_t, const float* x_t, const float* O_t, const float* pair_ctx, const float* out_eps,
                          const float* out_O0, const float* out_posterior, const float* true_post, const float* true_eps,
                          const float* true_O0, const uint8_t* gen_mask, const uint8_t* res_mask, const float* upstream3, float* d_res_ctx,
                          float* d_pair_ctx, const void* tape, size_t tape_bytes, void* workspace, size_t workspace_bytes, void* stream) {
  StreamOrder order_(stream);
  if (int rc = check_dims(d, "train_step_bwd")) return rc;
  if (int rc = check_denoiser_weights(d, w)) return rc;
  if (int rc = check_denoiser_weights(d, grads)) return rc;
  DIFFAB_REQUIRE(d->NL <= kMaxLayers, DIFFAB_ERR_UNSUPPORTED, "train_step_bwd: at most %d IPA layers", kMaxLayers);
  DIFFAB_REQUIRE(seq_t && x_t && O_t && pair_ctx && out_eps && out_O0 && out_posterior && true_post && true_eps && true_O0 && gen_mask &&
                     res_mask && upstream3 && tape && workspace,
                 DIFFAB_ERR_ARG, "train_step_bwd: null pointer");
  DIFFAB_REQUIRE(tape_bytes >= train_tape_floats(d) * sizeof(float), DIFFAB_ERR_WORKSPACE, "train_step_bwd: tape too small");
  DIFFAB_REQUIRE(workspace_bytes >= train_bwd_workspace_floats(d) * sizeof(float), DIFFAB_ERR_WORKSPACE, "train_step_bwd: workspace %zu < %zu",
                 workspace_bytes, train_bwd_workspace_floats(d) * sizeof(float));
  const TrainTape tp = carve_tape(d, static_cast<float*>(const_cast<void*>(tape)));
  return train_step_bwd(d, w, grads, tp, seq_t, x_t, O_t, pair_ctx, out_eps, out_O0, out_posterior, true_post, true_eps, true_O0, gen_mask,
                        res_mask, upstream3, d_res_ctx, d_pair_ctx, static_cast<float*>(workspace), as_stream(stream));
}

/* ---- Denoiser.forward / InvariantPointAttentionLayer.forward under autograd: taped forward + backward from arbitrary cotangents ---- */
int diffab_denoise_step_fwd_taped(const diffab_dims* d, const diffab_denoiser_weights* w, const int64_t* seq_t, const float* x_t,
                                  const float* O_t, const float* res_ctx, const float* pair_ctx, const float* beta, float* out_eps,
                                  float* out_O0, float* out_posterior, void* tape, size_t tape_bytes, uint32_t flags, void* stream) {
  StreamOrder order_(stream);
  if (int rc = check_dims(d, "denoise_step_fwd_taped")) return rc;
  if (int rc = check_denoiser_weights(d, w)) return rc;
  DIFFAB_REQUIRE(d->NL <= kMaxLayers, DIFFAB_ERR_UNSUPPORTED, "denoise_step_fwd_taped: at most %d IPA layers", kMaxLayers);
  DIFFAB_REQUIRE(seq_t && x_t && O_t && res_ctx && pair_ctx && beta && out_eps && out_O0 && out_posterior && tape, DIFFAB_ERR_ARG,
                 "denoise_step_fwd_taped: null pointer");
  DIFFAB_REQUIRE(tape_bytes >= train_tape_floats(d) * sizeof(float), DIFFAB_ERR_WORKSPACE, "denoise_step_fwd_taped: tape %zu < %zu bytes",
                 tape_bytes, train_tape_floats(d) * sizeof(float));
  const TrainTape tp = carve_tape(d, static_cast<float*>(tape));
  return denoise_step_taped(d, w, seq_t, x_t, O_t, res_ctx, pair_ctx, beta, out_eps, out_O0, out_posterior, tp, taped_flags(flags), as_stream(stream));
}

int diffab_denoise_step_bwd(const diffab_dims* d, const diffab_denoiser_weights* w, const diffab_denoiser_weights* grads,
                            const int64_t* seq_t, const float* x_t, const float* O_t, const float* pair_ctx, const float* out_posterior,
                            const float* d_eps, const float* d_O0, const float* d_posterior, float* d_res_ctx, float* d_pair_ctx,
                            float* d_x_t, float* d_O_t, const void* tape, size_t tape_bytes, void* workspace, size_t workspace_bytes,
                            void* stream) {
  StreamOrder order_(stream);
  if (int rc = check_dims(d, "denoise_step_bwd")) return rc;
  if (int rc = check_denoiser_weights(d, w)) return rc;
  if (int rc = check_denoiser_weights(d, grads)) return rc;
  DIFFAB_REQUIRE(d->NL <= kMaxLayers, DIFFAB_ERR_UNSUPPORTED, "denoise_step_bwd: at most %d IPA layers", kMaxLayers);
  DIFFAB_REQUIRE(seq_t && x_t && O_t && pair_ctx && out_posterior && d_res_ctx && tape && workspace, DIFFAB_ERR_ARG,
                 "denoise_step_bwd: null pointer");
  DIFFAB_REQUIRE(tape_bytes >= train_tape_floats(d) * sizeof(float), DIFFAB_ERR_WORKSPACE, "denoise_step_bwd: tape too small");
  DIFFAB_REQUIRE(workspace_bytes >= train_bwd_workspace_floats(d) * sizeof(float), DIFFAB_ERR_WORKSPACE, "denoise_step_bwd: workspace %zu < %zu",
                 workspace_bytes, train_bwd_workspace_floats(d) * sizeof(float));
  const TrainTape tp = carve_tape(d, static_cast<float*>(const_cast<void*>(tape)));
  return denoise_step_bwd(d, w, grads, tp, seq_t, x_t, O_t, pair_ctx, out_posterior, d_eps, d_O0, d_posterior, d_res_ctx, d_pair_ctx,
                          static_cast<float*>(workspace), as_stream(stream), d_x_t, d_O_t);
}

static diffab_dims one_layer(const diffab_dims* d) {
  diffab_dims d1 = *d;
  d1.NL = 1;
  return d1;
}
size_t diffab_ipa_layer_tape_bytes(const diffab_dims* d) {
  if (check_dims(d, "ipa_layer_tape_bytes")) return 0;
  const diffab_dims d1 = one_layer(d);
  return train_tape_floats(&d1) * sizeof(float);
}
size_t diffab_ipa_layer_bwd_workspace_bytes(const diffab_dims* d) {
  if (check_dims(d, "ipa_layer_bwd_workspace_bytes")) return 0;
  const diffab_dims d1 = one_layer(d);
  return train_bwd_workspace_floats(&d1) * sizeof(float);
}

int diffab_ipa_layer_fwd_taped(const diffab_dims* d, const diffab_ipa_layer_weights* w, const float* x, const float* e, const float* R,
                               const float* t, float* y, void* tape, size_t tape_bytes, uint32_t flags, void* stream) {
  StreamOrder order_(stream);
  if (int rc = check_dims(d, "ipa_layer_fwd_taped")) return rc;
  DIFFAB_REQUIRE(x && (e || d->C == 0) && R && t && y && tape, DIFFAB_ERR_ARG, "ipa_layer_fwd_taped: null pointer");
  const diffab_dims d1 = one_layer(d);
  DIFFAB_REQUIRE(tape_bytes >= train_tape_floats(&d1) * sizeof(float), DIFFAB_ERR_WORKSPACE, "ipa_layer_fwd_taped: tape %zu < %zu bytes", tape_bytes,
                 train_tape_floats(&d1) * sizeof(float));
  const TrainTape tp = carve_tape(&d1, static_cast<float*>(tape));
  hipStream_t st = as_stream(stream);
  const size_t nb = sizeof(float) * static_cast<size_t>(d->B) * d->K * d->D;
  DIFFAB_HIP_CHECK(hipMemcpyAsync(tp.x[0], x, nb, hipMemcpyDeviceToDevice, st));
  flags = taped_flags(flags);
  const bool b6 = tp.planes && use_b6_gemm(flags) && fast_path_supported(&d1);
  if (b6)
    if (int rc = ipa_layer_split_weights(w, tp.planes, st)) return rc;
  if (int rc = ipa_layer_dispatch(&d1, w, tp.x[0], e, R, t, tp.x[1], tp.ipa_ws[0], flags, st, tp.sp[0], tp.d2[0], b6 ? tp.planes : nullptr,
                                  nullptr, true))
    return rc;
  DIFFAB_HIP_CHECK(hipMemcpyAsync(y, tp.x[1], nb, hipMemcpyDeviceToDevice, st));
  return DIFFAB_OK;
}

int diffab_ipa_layer_bwd(const diffab_dims* d, const diffab_ipa_layer_weights* w, const diffab_ipa_layer_weights* grads, const float* e,
                         const float* R, const float* t, const float* dy, float* dx, float* d_e, float* d_R, float* d_t, const void* tape,
                         size_t tape_bytes, void* workspace, size_t workspace_bytes, void* stream) {
  StreamOrder order_(stream);
  if (int rc = check_dims(d, "ipa_layer_bwd")) return rc;
  DIFFAB_REQUIRE(w && grads && (e || d->C == 0) && R && t && dy && dx && tape && workspace, DIFFAB_ERR_ARG, "ipa_layer_bwd: null pointer");
  const diffab_dims d1 = one_layer(d);
  DIFFAB_REQUIRE(tape_bytes >= train_tape_floats(&d1) * sizeof(float), DIFFAB_ERR_WORKSPACE, "ipa_layer_bwd: tape too small");
  DIFFAB_REQUIRE(workspace_bytes >= train_bwd_workspace_floats(&d1) * sizeof(float), DIFFAB_ERR_WORKSPACE, "ipa_layer_bwd: workspace %zu < %zu",
                 workspace_bytes, train_bwd_workspace_floats(&d1) * sizeof(float));
  const TrainTape tp = carve_tape(&d1, static_cast<float*>(const_cast<void*>(tape)));
  return ipa_layer_bwd(&d1, w, grads, tp, R, t, e, dy, dx, d_e, static_cast<float*>(workspace), as_stream(stream), d_R, d_t);
}

int diffab_sample_loop(const diffab_dims* d, const diffab_denoiser_weights* w, const diffab_sched* s, const diffab_igso3* rev_tab,
                       int64_t* seq, float* x, float* O, const float* res_ctx, const float* pair_ctx, const uint8_t* gen_mask, uint64_t seed,
                       int64_t first_patch, int32_t t_start, int32_t t_stop, void* workspace, size_t workspace_bytes, uint32_t flags,
                       void* stream) {
  StreamOrder order_(stream);
  if (int rc = check_dims(d, "sample_loop")) return rc;
  if (int rc = check_denoiser_weights(d, w)) return rc;
  DIFFAB_REQUIRE(s && s->T > 0 && s->alpha && s->beta && s->one_minus_alpha_bar_sqrt, DIFFAB_ERR_ARG, "sample_loop: bad schedule");
  DIFFAB_REQUIRE(rev_tab && rev_tab->sigmas && rev_tab->cdf && rev_tab->n_sigmas >= s->T + 1 && rev_tab->n_bins > 0, DIFFAB_ERR_ARG,
                 "sample_loop: reverse IGSO3 table must have T+1 rows");
  DIFFAB_REQUIRE(seq && x && O && res_ctx && pair_ctx && gen_mask && workspace, DIFFAB_ERR_ARG, "sample_loop: null pointer");
  DIFFAB_REQUIRE(t_start <= s->T && t_stop >= 0 && t_stop <= t_start, DIFFAB_ERR_ARG, "sample_loop: need T >= t_start >= t_stop >= 0");
  const SampleBuffers sb = carve_sample(d, workspace);
  DIFFAB_REQUIRE(workspace_bytes >= sb.bytes, DIFFAB_ERR_WORKSPACE, "sample_loop: workspace %zu < %zu bytes", workspace_bytes, sb.bytes);
  hipStream_t st = as_stream(stream);
  // the folded sequence-embedding table depends on the weights only: once per call, not once per step (same condition as denoise_step)
  const StepBuffers b0 = carve_step(d, sb.step);
  const bool fold = !(flags & DIFFAB_FLAG_FORCE_GENERIC) && fast_path_supported(d) &&
                    rowgemm128_ok(res_ctx, d->D, b0.h1, d->D, d->B * d->K, d->D);
  if (fold)
    if (int rc = prepare_weights(d, w, b0, flags, st)) return rc;
  // the pair embedding is the same tensor in all T x NL attention launches of a trajectory: its fp16 planes are built once here
  if (!(flags & DIFFAB_FLAG_PAIR_F32)) flags |= DIFFAB_FLAG_PAIR_PLANES;
  const bool pair_ready = use_pair_planes(d, flags, pair_ctx, b0);
  if (pair_ready)
    if (int rc = launch_pair_split(d, pair_ctx, b0.pair, st)) return rc;
  // The IPA module as ONE patch-resident launch per step (ipa_persistent.hip; bitwise the 3 NL launches it replaces, so the choice never
  // shows in the samples) when the batch fills the chip with one work-group per patch: B = 256 is 2.60 ms per step against 2.70.  Fewer
  // patches than CUs leave CUs idle for the whole module (B = 8: 2.07 ms against 0.47), a ragged last round of patches costs a module
  // time for a few of them - those shapes keep the per-layer launches.
  if (!(flags & DIFFAB_FLAG_MULTI_LAUNCH) && pair_ready && fold && use_b6_gemm(flags) && ipa_module_persistent_supported(d)) {
    int dev = 0, ncu = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
    const int rounds = (d->B + ncu - 1) / ncu;
    // (K = 256 - two dense tiles and sixteen two-chunk attention items per patch in the same launch, round 6 - measures SLOWER than its
    // per-layer launches at B = 512: 19.5 against 18.6 ms per step; it is bitwise the same and stays behind the explicit flag)
    if (d->K == 128 && d->B >= ncu && static_cast<double>(d->B) >= 0.85 * rounds * ncu) flags |= DIFFAB_FLAG_PERSISTENT_MODULE;
  }
  // DIFFAB_FLAG_SKIP_UNUSED_ROWS: the step's outputs (eps, O0, posterior) are read for GENERATED residues only (reverse_update leaves
  // the others alone), so the last layer's attention is needed only for row tiles that contain one; every other layer feeds keys and
  // values of all rows to the next.  Same trajectory, bit for bit; the work skipped depends on the mask, so bench.py's headline keeps it off.
  const unsigned char* tiles = nullptr;
  if ((flags & DIFFAB_FLAG_SKIP_UNUSED_ROWS) && fold && d->K % 16 == 0) {
    if (int rc = launch_tiles_needed(gen_mask, d->B, d->K, sb.tiles, st)) return rc;
    tiles = sb.tiles;
  }
  // The heads' folded beta columns depend on (step, head, column) only - every patch of a reverse step has the same beta - so the table
  // of ALL steps is built once per call (the same kernel and formula as the per-step table, "patch" = step) instead of once per step.
  // Eager loop only: under graph replay the step index lives in device memory and the chain's bias pointer is a launch argument.
  const bool chain_path = fold && use_b6_gemm(flags) && d->V <= 128;  // (denoise_step's `chain`)
  const float* beta_traj = nullptr;
  if (chain_path && s->T + 1 <= kTrajRows && !(flags & DIFFAB_FLAG_GRAPH_SAMPLER)) {
    diffab_dims dt = *d;
    dt.B = s->T + 1;
    if (int rc = launch_fold_tables(&dt, w, s->beta, nullptr, sb.beta_traj, st, true)) return rc;
    beta_traj = sb.beta_traj;
  }
  auto one_step = [&](int t, const int* t_dev) -> int {
    if (!fold)  // (the folded head tables read the schedule themselves: one launch less per step)
      if (int rc = launch_fill_beta(s, t, d->B, sb.beta, st, t_dev)) return rc;
    if (int rc = denoise_step(d, w, seq, x, O, res_ctx, pair_ctx, sb.beta, sb.eps, sb.O0, sb.post, nullptr, nullptr, sb.step, flags, st, fold,
                              pair_ready, fold ? s->beta : nullptr, t, t_dev, tiles, true, t_dev ? nullptr : beta_traj, s->T + 1))
      return rc;
    // (the heads' epilogue - O0 = O_t exp(hat(v)), the posterior's softmax - runs inside the update kernel, for the generated rows)
    return launch_reverse_update_philox(s, rev_tab, t, seq, x, O, sb.eps, sb.O0, sb.post, gen_mask, seed, first_patch, d->B, d->K, d->V, st,
                                        t_dev, b0.vbuf, b0.logits);
  };
  // DIFFAB_FLAG_GRAPH_SAMPLER: a step is ~45 launches; at B = 1 (BASELINE config 1) their host cost (3-4 us each) is several times
  // the kernels' own time.  The first step runs eagerly (it also performs the one-time function-attribute calls), the second is
  // captured into a hipGraph that takes its timestep from device memory, and the graph is replayed for every remaining step: one host
  // call per step instead of 45.  Same kernels, same order, same arguments: bitwise the same trajectory (tested).
  const int n_steps = t_start - t_stop;
  const bool graph = (flags & DIFFAB_FLAG_GRAPH_SAMPLER) && n_steps >= 3 && !kernel_timer_enabled();
  if (!graph) {
    for (int t = t_start; t > t_stop; --t)
      if (int rc = one_step(t, nullptr)) return rc;
    return DIFFAB_OK;
  }
  if (int rc = one_step(t_start, nullptr)) return rc;
  if (int rc = launch_set_int(sb.t_dev, t_start - 1, st)) return rc;
  // Capture and replay run on a private stream (the caller's may be the legacy default stream, which cannot be captured), ordered
  // behind the caller's stream by an event; the private stream is drained before the call returns, so later work on the caller's
  // stream sees the finished trajectory, and the executable graph outlives its launches (the only synchronisation in this library;
  // the eager path stays fully asynchronous).
  hipStream_t side = nullptr;
  hipEvent_t ev = nullptr;
  hipGraph_t g = nullptr;
  hipGraphExec_t ge = nullptr;
  DIFFAB_HIP_CHECK(hipStreamCreateWithFlags(&side, hipStreamNonBlocking));
  hipError_t ei = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
  if (ei == hipSuccess) ei = hipEventRecord(ev, st);
  if (ei == hipSuccess) ei = hipStreamWaitEvent(side, ev, 0);
  int rc = DIFFAB_OK;
  if (ei == hipSuccess) {
    hipStream_t caller = st;
    st = side;  // one_step enqueues on `st`
    ei = hipStreamBeginCapture(side, hipStreamCaptureModeThreadLocal);
    if (ei == hipSuccess) {
      rc = one_step(t_start - 1, sb.t_dev);
      if (rc == DIFFAB_OK) rc = launch_dec_int(sb.t_dev, side);
      ei = hipStreamEndCapture(side, &g);
    }
    st = caller;
  }
  if (ei == hipSuccess && rc == DIFFAB_OK) ei = hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
  if (ei == hipSuccess && rc == DIFFAB_OK)
    for (int t = t_start - 1; t > t_stop && ei == hipSuccess; --t) ei = hipGraphLaunch(ge, side);
  if (ei == hipSuccess) ei = hipStreamSynchronize(side);
  if (ge) (void)hipGraphExecDestroy(ge);
  if (g) (void)hipGraphDestroy(g);
  if (ev) (void)hipEventDestroy(ev);
  (void)hipStreamDestroy(side);
  if (rc != DIFFAB_OK) return rc;
  if (ei != hipSuccess) {
    set_error("sample_loop: graph capture / replay failed: %s", hipGetErrorString(ei));
    return DIFFAB_ERR_HIP;
  }
  return DIFFAB_OK;
}

}  // extern "C"
