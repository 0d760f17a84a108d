// denoiser_internal.h - launchers shared between denoiser_generic.hip, denoiser_fast.hip and api.hip.
#pragma once
#include "common.h"

namespace diffab {

// generic (any dims)
int launch_linear_generic(const float* X, int ldx, const float* W, const float* bias, float* Y, int ldy, int M, int N, int Kd, bool relu,
                          hipStream_t st);
size_t ipa_generic_workspace_floats(const diffab_dims* d);
int ipa_layer_generic(const diffab_dims* d, const diffab_ipa_layer_weights* w, const float* x, const float* e, const float* R, const float* t,
                      float* y, float* ws, hipStream_t st);
int launch_embed_concat(const float* res_ctx, const float* emb, const int64_t* seq, int D, int64_t rows, float* out, hipStream_t st);
int launch_beta_concat(const float* h, const float* beta, int D, int K, int64_t rows, float* out, hipStream_t st);
int launch_heads_finish(const float* v, const float* O_t, const float* logits, int V, int64_t rows, float* O0, float* post, hipStream_t st);
// noslp_kernels.hip (compiled without the SLP vectoriser: no packed fp32 arithmetic from scalar code)
int launch_proj_frames_f32(const float* x, const float* const* W6, const float* R, const float* t, float* proj, int rows, hipStream_t st);
int launch_losses_bwd(const float* post, const float* tpost, const float* eps, const float* teps, const float* O0, const float* tO,
                      const float* O_t, const float* v, const uint8_t* gm, const uint8_t* rm, const float* count, const float* up, int V,
                      int64_t rows, float* d_logits, float* d_eps, float* d_v, hipStream_t st);
int launch_heads_cotangent(const float* post, const float* c_post, const float* c_eps, const float* c_O0, const float* O_t, const float* v,
                           int V, int64_t rows, float* d_logits, float* d_eps, float* d_v, float* d_Ot, hipStream_t st);
int launch_ipa_frames_bwd(const float* proj, const float* dproj, int NP, int pt_col0, int n_pts, const float* feat, const float* dfeat, int F,
                          int ol_col0, int on_col0, int n_vpts, const float* R, const float* t, float* dR, float* dt, int rows,
                          hipStream_t st);

// MFMA path for the benchmark geometry (D=128, C=64, H=8, DS=32, PQ=PV=8, K % 16 == 0)
bool fast_path_supported(const diffab_dims* d);
size_t ipa_fast_workspace_floats(const diffab_dims* d);
int ipa_layer_fast(const diffab_dims* d, const diffab_ipa_layer_weights* w, const float* x, const float* e, const float* R, const float* t,
                   float* y, float* ws, hipStream_t st,
                   float* sp_keep = nullptr, float* d2_keep = nullptr,  // training tape: three launches, P and d2 kept in these buffers
                   const void* planes = nullptr,  // ipa_layer_split_weights() output; nullptr: split per call into the workspace tail
                   const float* pair_planes = nullptr,  // launch_pair_split() output: attention's pair-tile products on f16 MFMA
                   bool fp32_gemm = false,
                   bool taped = false,   // ws is a slot of the training tape: proj and feat are read by the backward (no scratch use)
                   const unsigned char* tile_needed = nullptr);  // [B][K / 16]: row tiles whose outputs are read (nullptr: all)  // DIFFAB_FLAG_FP32_GEMM: dense products on the f32-input MFMA kernels
// fp16 planes of the pair embedding for the fused attention kernel (K = 64 / 128): pair_planes_floats(d) floats, 256-byte aligned
bool pair_planes_supported(const diffab_dims* d);
size_t pair_planes_floats(const diffab_dims* d);
int launch_pair_split(const diffab_dims* d, const float* e, float* planes, hipStream_t st);
const float* pair_row_scales(const diffab_dims* d, const float* planes);  // {s_i, 1 / s_i} per pair row (b, i) of a launch_pair_split() buffer
// Y = act(X W^T + b) on MFMA; requires Kd % 4 == 0 (falls back to the generic kernel otherwise)
int launch_linear(const float* X, int ldx, const float* W, const float* bias, float* Y, int ldy, int M, int N, int Kd, bool relu,
                  hipStream_t st);

// Y[M x 128] = act(X[:, 0:Kd] W[:, 0:Kd]^T + bias row), W rows ldw floats apart (4-byte aligned), bias row = bias (vector) or
// bias[128 * bias_idx[row]] or bias[128 * (row / bias_div)]; requires rowgemm128_ok()
bool rowgemm128_ok(const float* X, int ldx, const float* Y, int ldy, int M, int Kd);
int launch_rowgemm128(const float* X, int ldx, const float* W, int ldw, const float* bias, const int64_t* bias_idx, int bias_div, float* Y,
                      int ldy, int M, int Kd, bool relu, hipStream_t st);
// gemm_bf16x6.hip: the same products on the bf16 matrix cores, fp32-accurate (three-way bf16 split of both operands, six partial
// products, fp32 accumulation).  The weights are split once into bf16 planes (launch_wsplit128 / launch_pjsplit: per call, or
// once per trajectory by the reverse sampler) and the ...p launchers take the planes.
size_t rowgemm128_b6_scratch_bytes(int Kd);
bool rowgemm128_b6_ok(const float* X, int ldx, const float* Y, int ldy, int M, int Kd);
int launch_wsplit128(const float* W, int ldw, int Kd, void* planes, hipStream_t st, int nrows = 128);  // rows >= nrows: zero planes
// X[M x 128] through 2 or 3 dense 128-wide layers in ONE kernel (ReLU between them, the last layer n_out <= 128 columns wide,
// activations stay in LDS); layer 0's bias may be a table indexed per row like launch_rowgemm128's
int launch_mlp_chains_b6(const float* X, int ldx, int nchains, const void* const* planes, const float* const* bias, const int64_t* bias_idx0,
                         int bias_div0, int nlayers, const int* n_out, float* const* Y, const int* ldy, int M, hipStream_t st);
struct MlpChainSet;  // mlp_chain_tile.h
// the descriptors of up to three chains (the by-value kernel argument of mlp_chain_b6_kernel; also handed to the module kernel)
int make_mlp_chain_set(MlpChainSet* out, int nchains, const void* const* planes, const float* const* bias, const int64_t* bias_idx0,
                       int bias_div0, int nlayers, const int* n_out, float* const* Y, const int* ldy);
int launch_mlp_chain_b6(const float* X, int ldx, const void* const* planes, const float* const* bias, const int64_t* bias_idx0, int bias_div0,
                        int nlayers, int n_out, float* Y, int ldy, int M, hipStream_t st);
int launch_wsplit128_strided(const float* W, int64_t sn, int64_t sk, int kseg, int k0, void* planes, hipStream_t st);  // (n, k) = W[n sn + k sk]
int launch_wsplit128_segs(const float* const* W, const int* k_end, int nseg, void* planes, hipStream_t st);  // W_q[k][n] stacked along k
int launch_rowgemm128_b6p(const float* X, int ldx, const void* planes, const float* bias, const int64_t* bias_idx, int bias_div, float* Y,
                          int ldy, int M, int Kd, bool relu, hipStream_t st, float* parts = nullptr);
size_t rowgemm128_b6_parts_floats(int M, int Kd);  // scratch of the k-parts form (few row tiles)
size_t rowgemm128_h3_parts_floats(int M, int Kd);
bool value_planes_enabled();  // true: diffab_debug_set_attn_variant(16)
void ipa_ws_value_planes(const diffab_dims* d, float* ws, float** vpl, float** vsc);  // the value planes / scales inside a layer workspace
bool tn_h3_enabled();     // false: diffab_debug_set_attn_variant(8 | 32): weight-gradient products in the six-term bf16 form
bool dense_h3_enabled();  // false: diffab_debug_set_attn_variant(8), the six-term bf16 form of the projections and to_out
int launch_rowgemm128_b6(const float* X, int ldx, const float* W, int ldw, const float* bias, const int64_t* bias_idx, int bias_div, float* Y,
                         int ldy, int M, int Kd, bool relu, void* scratch, hipStream_t st);
bool use_b6_gemm(uint32_t flags = 0);  // false with DIFFAB_FLAG_FP32_GEMM (experimental builds: or DIFFAB_FP32_GEMM=1 in the environment)
// the six IPA projections + frames; W6 = {wq_s, wk_s, wv_s, wq_p, wk_p, wv_p}
size_t proj_frames_b6_scratch_bytes();
int launch_pjsplit(const float* const* W6, void* planes, hipStream_t st);
int launch_proj_frames_b6p(const float* x, const void* planes, const float* R, const float* t, float* proj, int rows, hipStream_t st);
// the same x-stationary kernel as a plain product Y[rows x N] = X[rows x 128] W'^T, (n, k) of W' = W[n sn + k sk] (to_out input gradient)
size_t xstat_b6_scratch_bytes(int N);
// gemm_f16x3.hip: C[N1 x N2] += A^T B as three fp16 terms (launch_gemm_tn_b6's contract)
int launch_gemm_tn_h3(const float* A, int lda, const float* B, int ldb, float* C, int ldc, int M, int N1, int N2, float* db,
                      float* const* seg_ptrs, const int* seg_ends, int nseg, hipStream_t st);
size_t xstat_h3_scratch_bytes(int N);  // gemm_f16x3.hip: the same product as three fp16 terms
int launch_xstat_h3(const float* X, const float* W, int64_t sn, int64_t sk, float* Y, int ldy, int rows, int N, void* scratch, hipStream_t st);
int launch_xstat_b6(const float* X, const float* W, int64_t sn, int64_t sk, float* Y, int ldy, int rows, int N, void* scratch, hipStream_t st);
// weight-gradient product C[N1 x N2] += A[M x N1]^T B[M x N2] (transposing LDS reads), any widths / row strides; rows of C optionally
// spread over nseg matrices
bool gemm_tn_b6_ok(const float* A, int lda, const float* B, int ldb, int M, int N1, int N2);
int launch_gemm_tn_b6(const float* A, int lda, const float* B, int ldb, float* C, int ldc, int M, int N1, int N2, float* db,
                      float* const* seg_ptrs, const int* seg_ends, int nseg, hipStream_t st);  // db (nullable): += column sums of A
// split planes of one IPA layer's projection and to_out weights: ipa_layer_planes_bytes() bytes, 256-byte aligned
size_t ipa_layer_planes_bytes();
size_t ipa_layer_out_planes_offset();  // the to_out planes inside a layer's block
size_t ipa_layer_small_offset();       // fp32 copies of w_bias [8][64], gamma [8] (padded to 64), b_out [128] behind the planes
size_t ipa_layer_h3_pj_offset();       // gemm_f16x3.hip: the projections' two-piece fp16 planes,
size_t ipa_layer_h3_out_offset();      //   to_out's,
size_t ipa_layer_h3_wis_offset();      //   and 1 / scale per output column: [1344 | 128] floats
// gemm_f16x3.hip: the two big dense products of a layer as three-term fp16 split products (half the matrix-pipe work of bf16x6)
size_t rowgemm128_h3_planes_bytes(int Kd);
size_t proj_frames_h3_planes_bytes();
int launch_wsplit128_h3(const float* W, int ldw, int Kd, void* planes, float* wis, hipStream_t st, int nrows = 128);
int launch_pjsplit_h3(const float* const* W6, void* planes, float* wis, hipStream_t st);
int launch_rowgemm128_h3p(const float* X, int ldx, const void* planes, const float* wis, const float* bias, const int64_t* bias_idx, int bias_div,
                          float* Y, int ldy, int M, int Kd, bool relu, hipStream_t st, float* parts = nullptr);
int launch_proj_frames_h3p(const float* x, const void* planes, const float* wis, const float* R, const float* t, float* proj, int rows,
                           hipStream_t st);
size_t proj_value_planes_floats(int64_t rows);   // (attn_planes_tile.h; consumer: ipa_attn_tile.h phase 3, VPL)
size_t proj_value_scales_floats(int64_t rows);
int ipa_layer_split_weights(const diffab_ipa_layer_weights* w, void* planes, hipStream_t st);
// ipa_persistent.hip: all NL layers of the IPA module for K = 128 patches as ONE patch-resident launch (one work-group owns a patch
// through projections -> 8 attention row tiles -> to_out, layer after layer; no inter-CU synchronisation).  xa: input, the result is in
// (NL odd ? xb : xa).  planes: NL x ipa_layer_planes_bytes(); pair_planes: launch_pair_split(); ws: proj | feat (ipa_fast_workspace_floats)
bool ipa_module_persistent_supported(const diffab_dims* d);
int launch_ipa_module_persistent(const diffab_dims* d, float* xa, float* xb, const float* R, const float* t, float* ws, const void* planes,
                                 const float* pair_planes, hipStream_t st, const float* emb_X = nullptr, const MlpChainSet* emb = nullptr,
                                 const MlpChainSet* heads = nullptr);
void set_module_stagger(int ticks, int classes);  // diagnostics: start-up stagger of the persistent module kernel (10 ns ticks)
void set_module_stamps(void* device_buffer);      // diagnostics: phase stamps of the persistent module kernel
// bias tables of the folded concatenations: emb_tab[25][D] and beta_tab[3 heads][B][D] (see denoiser_fast.hip)
int launch_fold_tables(const diffab_dims* d, const diffab_denoiser_weights* w, const float* beta, float* emb_tab, float* beta_tab,
                       hipStream_t st, bool emb_tab_ready = false,
                       const float* sched_beta = nullptr, int t = 0, const int* t_dev = nullptr);  // beta = sched_beta[t] for every patch  // beta == nullptr: the weights-only embedding table alone

// pair_embed_fused.hip: PairEmbedding forward as one kernel (C = 64, K % 128 == 0, A <= 16); context_kernels.hip falls back to its
// unfused launches elsewhere.  prep: pair_embed_fused_prep_floats(d) floats (prepared planes and tables, rebuilt per call); tapes
// (all or none): the four hidden activations [nrows][64] of rows [row0, row0 + nrows) for the backward; out == nullptr: tapes only
bool pair_embed_fused_supported(const diffab_ctx_dims* d);
size_t pair_embed_fused_prep_floats(const diffab_ctx_dims* d);
int launch_pair_embed_fused(const diffab_ctx_dims* d, const diffab_pair_emb_weights* w, const int64_t* seq_idx, const float* distmat,
                            const float* xyz, const float* pairwise_dihedrals, const int64_t* residue_idx, int32_t residue_idx_batch_stride,
                            const int64_t* chain_idx, const float* atom_mask, const uint8_t* sequence_context_mask, float* out, float* prep,
                            float* tape_h1, float* tape_df, float* tape_m1, float* tape_m2, int64_t row0, int64_t nrows, hipStream_t st,
                            const float** coef_sp_out = nullptr);
int launch_softplus_table(const float* coefw, int n, float* out, hipStream_t st);  // softplus(pair2distcoef) (context_kernels.hip)

// attention_split.hip: the attention of one IPA layer as three launches (logits | pair stream | P x V) exchanging the
// (B, 8, K, K) logits / probabilities through SP - the form the training tape keeps; single key chunk only (K = 64, 128)
bool attention_split_supported(const diffab_dims* d);
size_t attention_split_workspace_floats(const diffab_dims* d);
int launch_ipa_logits(const diffab_dims* d, const float* proj, const float* gamma, float* SP, hipStream_t st);  // S[b][h][i][j] only
int launch_attention_probs(const diffab_dims* d, const float* proj, const float* e, const float* Wb, const float* gamma, float* P,
                           float* D2, hipStream_t st);  // training backward: normalised probabilities + squared point distances
// d pair_ctx of all layers in one pass (attention_split.hip): P / G [B][8][K][K], doe [B K][512], Wb [8][64] per layer
constexpr int kDeLayersMax = 6;
int launch_pair_de_layers(const diffab_dims* d, int nl, const float* const* P, const float* const* G, const float* const* doe,
                          const float* const* Wb, float* de, hipStream_t st);
int launch_pair_stream_bwd(const diffab_dims* d, const float* e, const float* P, float* G /* in dA_kv, out g */, const float* D2,
                           const float* dfeat, float* wb_part, const float* Wb, float* de /* nullable: += d pair_ctx */,
                           hipStream_t st);  // training backward: g, d gamma / d w_bias partials, d e
int launch_attention_split(const diffab_dims* d, const float* proj, const float* e, const float* R, const float* t, const float* Wb,
                           const float* gamma, float* feat, float* SP, hipStream_t st, float* D2 = nullptr);

void set_stream_order(bool on);  // api.hip: the cross-stream ordering guard (common.h StreamOrder)
void set_attn_variant(int v);  // diagnostics: 1 = the four-wave / two-groups-per-CU form of the plane attention kernel
void set_attn_stamps(void* device_buffer);  // diagnostics: per-wave s_memtime stamps of the attention kernel's phases

// api.hip: opt-in hipEvent bracket around the dominant (attention) kernel
void timer_begin(hipStream_t st);
void timer_end(hipStream_t st);
bool kernel_timer_enabled();

// denoiser_backward.hip: saved activations of one training forward (every buffer written by the forward kernels themselves)
constexpr int kMaxLayers = 16;
struct TrainTape {
  float *cat2, *h1, *x[kMaxLayers + 1], *ipa_ws[kMaxLayers], *cat3, *t1[3], *t2[3], *vbuf, *logits;
  float* scratch;  // 1024 floats: partial sums of the loss reduction
  void* planes;    // ipa_layer_planes_bytes(): the current layer's split weights for the bf16x6 forward GEMMs (null off the MFMA path)
  // per layer, benchmark geometry with K = 64 / 128 only (else null): the attention probabilities and the squared point distances
  // [b][h][i][j], saved by the three-launch forward so that the backward does not recompute them
  float *sp[kMaxLayers], *d2[kMaxLayers];
};
size_t train_tape_floats(const diffab_dims* d);
TrainTape carve_tape(const diffab_dims* d, float* base);
size_t train_bwd_workspace_floats(const diffab_dims* d);
int train_step_bwd(const diffab_dims* d, const diffab_denoiser_weights* w, const diffab_denoiser_weights* g, const TrainTape& tp,
                   const int64_t* seq_t, const float* x_t, const float* O_t, const float* pair_ctx, const float* eps_hat,
                   const float* O0_hat, const float* post_hat, const float* true_post, const float* true_eps, const float* true_O0,
                   const uint8_t* gm, const uint8_t* rm, const float* upstream3, float* d_res_ctx, float* d_pair_ctx, float* ws,
                   hipStream_t st);
// the same backward from arbitrary cotangents of the Denoiser outputs (null = zero), and one IPA layer's backward (one-layer tape)
int denoise_step_bwd(const diffab_dims* d, const diffab_denoiser_weights* w, const diffab_denoiser_weights* g, const TrainTape& tp,
                     const int64_t* seq_t, const float* x_t, const float* O_t, const float* pair_ctx, const float* post_hat,
                     const float* cot_eps, const float* cot_O0, const float* cot_post, float* d_res_ctx, float* d_pair_ctx, float* ws,
                     hipStream_t st, float* d_x_t = nullptr, float* d_O_t = nullptr);  // frame gradients (nullable)
int ipa_layer_bwd(const diffab_dims* d1, const diffab_ipa_layer_weights* lw, const diffab_ipa_layer_weights* lg, const TrainTape& tp,
                  const float* R, const float* t, const float* pair_ctx, const float* dy, float* dx, float* d_pair_ctx, float* ws,
                  hipStream_t st, float* d_R = nullptr, float* d_t = nullptr);  // frame gradients (nullable)
// Y = act(X W^T + b) backward: dW += dY^T X (W is N x Kd, row-major), db += colsum dY (nullable), dX (+)= dY W (nullable).
// dY must already carry the activation mask (bwd_relu_mask: dY *= act > 0, in place).  bwd_gemm_nn: C (+)= A[M,K] B[K,N].
int bwd_linear(const float* dY, int ldy, const float* X, int ldx, const float* W, float* dW, float* db, float* dX, int lddx, int M, int N,
               int Kd, bool acc_dx, hipStream_t st);
int bwd_relu_mask(float* dY, const float* act, int64_t n, hipStream_t st);
// bwd_linear whose input gradient is masked by relu_act > 0 (rows lddx apart) in the product's epilogue; C = (A B) masked likewise
int bwd_linear_masked(const float* dY, int ldy, const float* X, int ldx, const float* W, float* dW, float* db, float* dX, int lddx, int M,
                      int N, int Kd, const float* relu_act, hipStream_t st);
// pair_chain_bwd.hip: the four 64-wide layers at the end of the PairEmbedding backward as one launch per chunk of pair rows
size_t pair_chain_bwd_prep_floats();
size_t pair_chain_bwd_part_floats();
int launch_pair_dist_bwd_fused(const int64_t* seq, const uint8_t* seq_m, const float* distmat, const float* xyz, const float* din, const float* dh1,
                               const float* W, int ld_w, int K, int A, int n_aa, int unk, int64_t row0, int64_t nrows, int ld, float* g_sp,
                               float* prep, hipStream_t st);
bool pair_dist_bwd_mfma_supported(int K, int A, int64_t row0, int64_t nrows, int ld, int n_aa);
int launch_pair_dist_bwd_mfma(const int64_t* seq, const uint8_t* seq_m, const float* distmat, const float* xyz, const float* din, const float* ddin,
                              int K, int A, int n_aa, int unk, int64_t row0, int64_t nrows, int ld, float* g_sp, hipStream_t st);
bool pair_table_mfma_supported(int C, int K, int64_t nrows, int n_aa, int max_dist);
int launch_pair_table_mfma(const float* g, const int64_t* seq, const uint8_t* seq_m, const int64_t* resid, int resid_bstride, const int64_t* chain,
                           int K, int max_dist, int n_aa, int unk, int64_t row0, int64_t nrows, float* G1, float* part, hipStream_t st);
// per-work-group partial sums -> their destinations: out[q][(i / cols[q]) ld[q] + i % cols[q]] += sum_p parts[p stride + off[q] + i], i < n[q]
struct PartsSegs { int nseg; int off[8]; int n[8]; int cols[8]; int ld[8]; float* out[8]; };
int launch_parts_reduce(const float* parts, int nparts, int64_t stride, const PartsSegs& sg, hipStream_t st);
bool pair_chain_bwd_supported(int C, int K, int64_t nrows);
bool pair_chain_bwd_enabled();  // false: diffab_debug_set_attn_variant bit 6 (64): the separate launches (A/B, tests)
int launch_pair_chain_bwd(const float* d_out, const float* amask, int K, int A, int ca, int64_t row0, int64_t nrows, const float* const* X,
                          const float* const* W, const int* ldw, float* dC, float* dh1, float* const* gW, const int* ldg, float* const* gb,
                          float* prep, float* part, hipStream_t st);
int bwd_gemm_nn_masked(const float* A, int lda, const float* B, int ldb, float* C, int ldc, int M, int N, int K, const float* relu_act,
                       hipStream_t st);
// n <= 4 products C_p[64 x 64] (rows ldc_p apart) += A_p^T B_p of dense [M x 64] operands in one launch; db_p (nullable) += colsum A_p
int bwd_tn64_set(int n, const float* const* A, const float* const* B, float* const* C, const int* ldc, float* const* db, int64_t M,
                 hipStream_t st);
// C[N1 x N2] (rows ldc apart) += A[M x N1]^T B[M x N2]; db (nullable) += column sums of A
int bwd_gemm_tn(const float* A, int lda, const float* B, int ldb, float* C, int ldc, int M, int N1, int N2, float* db, hipStream_t st);
int bwd_gemm_nn(const float* A, int lda, const float* B, int ldb, float* C, int ldc, int M, int N, int K, bool acc, hipStream_t st);
int launch_losses_fwd(const float* pp, const float* tp, const float* pe, const float* te, const float* pO, const float* tO, const uint8_t* gm,
                      const uint8_t* rm, int B, int K, int V, float* out3, hipStream_t st,
                      float* scratch = nullptr);  // scratch: 1024 floats -> the multi-work-group two-stage reduction

// diffusion_kernels.hip
int launch_reverse_update_philox(const diffab_sched* s, const diffab_igso3* tab, int t, int64_t* seq, float* x, float* O,
                                 const float* eps_hat, float* O0_hat, float* post, const uint8_t* gm, uint64_t seed,
                                 int64_t first_patch, int B, int K, int V, hipStream_t st, const int* t_dev = nullptr,
                                 const float* head_v = nullptr, const float* head_logits = nullptr);  // heads' epilogue done in the kernel  // t_dev: read the timestep from device memory (graph replay)
int launch_fill_beta(const diffab_sched* s, int t, int B, float* out, hipStream_t st, const int* t_dev = nullptr);
int launch_tiles_needed(const uint8_t* gm, int B, int K, unsigned char* out, hipStream_t st);  // [B][K / 16]: any generated residue in the tile
int launch_set_int(int* p, int v, hipStream_t st);
int launch_dec_int(int* p, hipStream_t st);

}  // namespace diffab
