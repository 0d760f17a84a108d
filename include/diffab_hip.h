/*
 * diffab_hip.h - C ABI of libdiffab_hip.so: the MI355X (gfx950) engine under the
 * DiffAb diffusion / denoise hot path.
 *
 * The reference (dohlee/diffab-pytorch) is pure Python on stock ATen ops and has
 * no FFI of its own; its boundary for this path is the
 * Python class surface of diffab_pytorch.DiffAb.  This header is the C ABI a
 * maintainer binds underneath that surface (ctypes stub: INTEGRATION.md).  Each
 * entry point cites the reference code it replaces, file:line relative to the
 * reference repository root.
 *
 * Conventions (all entry points):
 *   - plain device pointers and sizes; no torch / HIP types in signatures
 *     (`stream` is a hipStream_t passed as void*, NULL = default stream);
 *   - row-major, contiguous, float32 unless stated; residue/aa indices and
 *     timesteps are int64 (torch.LongTensor), masks are 1 byte per element
 *     (torch.bool);
 *   - the caller owns every buffer including the workspace (size from the
 *     matching *_workspace_bytes query); kernels are enqueued on `stream` and
 *     never synchronise on the host; no hidden global state beyond the opt-in stream guard below; the
 *     library never reads the environment.  Two DIAGNOSTIC entry points keep
 *     process-global state and are off by default: diffab_kernel_timer_enable/read
 *     (an event list) and diffab_debug_set_attn_stamps / diffab_debug_set_module_stamps / _stagger (stamp-buffer pointers, two ints);
 *     they are not thread-safe and must not be left enabled in production.
 *     DIFFAB_FLAG_GRAPH_SAMPLER makes diffab_sample_loop drain a private stream
 *     before it returns.  (Kernel variants that were measured and not adopted, the environment
 *     switches used to A/B them and the timing-ablation hooks are patches under experiments/.)
 *   - Streams: calls on ONE stream are ordered by the stream, as usual; calls on DIFFERENT streams are independent and may overlap on
 *     the device (no state of the library is shared between two calls).  History: rounds 3-5 saw a small elementwise kernel compute wrong
 *     values in lanes 48-63 while kernels of a second library pipeline ran on another stream, and serialised the library's calls across
 *     streams by default.  Round 6 found the cause (profiles/r06_lanes_48_63.md; reproducer tools/hwtests/pkmul_two_streams.hip): gfx950
 *     returns a wrong low result in lanes 48-63 for v_pk_{mul,add,fma}_f32 ... op_sel:[0,1] while f16 / bf16 MFMAs of ANY wave on the
 *     SIMD - another kernel's included - are in flight, and hipcc's SLP vectoriser had formed that instruction in 14 VALU-only kernels.
 *     No kernel of the library contains the form any more (enforced at build time and by tests/test_isa_lint.py), two pipelines on two
 *     streams are bitwise the sequential runs, and the ordering guard is OFF by default.  diffab_set_stream_guard(1) switches it on:
 *     before a call enqueues on stream B, everything enqueued so far on the stream of the library's previous call (same device) is
 *     ordered in front of it (hipEventRecord + hipStreamWaitEvent; no host synchronisation; one mutex per device).  With the guard on,
 *     a stream handed to the library must stay alive until the library's next call on that device has been made, and library calls must
 *     not be captured into a hipGraph from a stream other than the last one used.  NOTE for callers: the hardware behaviour applies to
 *     YOUR kernels too - a caller's own VALU kernel holding that packed form can miscompute beside this library's f16 / bf16 MFMA
 *     kernels on another stream (tools/isa_hazard_lint.py checks any gfx950 object file or shared library).
 *   - empty problems (a count or extent of 0) return 0 before any pointer is looked at: an empty tensor's data pointer is NULL;
 *   - return 0 on success, a negative DIFFAB_ERR_* otherwise (never throws);
 *     diffab_last_error() gives the thread's last message.
 */
#ifndef DIFFAB_HIP_H
#define DIFFAB_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DIFFAB_OK 0
#define DIFFAB_ERR_ARG (-1)         /* null pointer / non-positive size / inconsistent dims */
#define DIFFAB_ERR_UNSUPPORTED (-2) /* dims outside what the kernels cover */
#define DIFFAB_ERR_HIP (-3)         /* a HIP runtime call failed */
#define DIFFAB_ERR_WORKSPACE (-4)   /* workspace too small */

/* flags */
#define DIFFAB_FLAG_FORCE_GENERIC 1u /* skip the MFMA kernels specialised for D=128,C=64,H=8,DS=32,P=8 */
/* (bits 2u, 4u, 8u selected attention variants that were measured slower than the fused kernel and are no longer part of this
   library: experiments/README.md; the three-launch attention survives as the form the training tape keeps) */

#define DIFFAB_FLAG_PAIR_PLANES 32u /* K = 64 / 128, default attention kernel: the pair embedding is first rewritten as two fp16 planes (e s =
                                      h1 + h2 to 2^-23 of the tensor maximum, same bytes, in the workspace) and the two products on
                                      the pair tile run on the f16 matrix cores as three exact partial products each, fp32
                                      accumulation.  diffab_sample_loop always does this (once per trajectory); for single calls
                                      the flag adds the rewrite (2x the pair embedding in HBM traffic) to every call. */

#define DIFFAB_FLAG_PAIR_F32 64u /* diffab_sample_loop: keep the fp32 pair stream (do not build the fp16 planes); for single calls simply
                                    do not pass DIFFAB_FLAG_PAIR_PLANES.  The plain-fp32 reference form of the attention kernel. */
#define DIFFAB_FLAG_FP32_GEMM 128u /* forward paths: the dense products (projections, to_out, MLPs) on the f32-input MFMA kernels instead
                                      of the split-precision products (projections, to_out: three-term fp16 under power-of-two
                                      scales; MLPs: six-term bf16); same results to fp32 rounding (the plain-fp32 reference form) */

#define DIFFAB_FLAG_GRAPH_SAMPLER 16u /* diffab_sample_loop: capture one reverse step into a hipGraph (timestep read from device memory)
                                         and replay it for the remaining steps - one host call per step instead of ~45.  Bitwise the
                                         eager trajectory.  The call drains its private replay stream before it returns (the graph
                                         must outlive its launches).  Measured at B = 1, K = 128: no gain (the host already runs
                                         ahead of the device; a step is 45 dependent small-grid kernels), hence opt-in. */
#define DIFFAB_FLAG_PERSISTENT_MODULE 512u /* MFMA path with pair planes, K = 128 or 256 (diffab_sample_loop, or a single call with
                                         DIFFAB_FLAG_PAIR_PLANES): the NL layers of the IPA module (reference diffab_pytorch.py:494-498)
                                         run as ONE patch-resident launch - a 512-thread work-group owns a patch through projections,
                                         eight attention row tiles and to_out, layer after layer, with no inter-CU synchronisation
                                         (patches never exchange data) - instead of 3 NL chip-wide launches.  Same tile bodies:
                                         bitwise the multi-launch result.  Where the MLP chains apply (D == 128, V <= 128, no
                                         out_res_emb) the launch also runs the embedding MLP of the patch's rows in front of layer 0
                                         and the three heads behind the last layer: one launch per denoiser forward.  Ignored where
                                         it does not apply.  diffab_sample_loop
                                         chooses it by itself at K = 128 when the batch fills the chip (B >= number of CUs), see
                                         DIFFAB_FLAG_MULTI_LAUNCH; with DIFFAB_FLAG_SKIP_UNUSED_ROWS the per-layer launches stay.  K = 256
                                         (two dense tiles, sixteen two-chunk attention items per patch; round 6) only on request:
                                         it measures 4 % slower than its per-layer launches at B = 512. */
#define DIFFAB_FLAG_MULTI_LAUNCH 1024u /* diffab_sample_loop: keep one launch per kernel of an IPA layer even where the patch-resident module
                                          launch would be chosen (B >= number of CUs, K = 128); the two forms are bitwise equal */
#define DIFFAB_FLAG_SKIP_UNUSED_ROWS 256u /* diffab_sample_loop (MFMA path, K % 16 == 0): a step's outputs are read for GENERATED residues
                                         only (diffab_reverse_update leaves the others alone), so the LAST layer's attention runs only
                                         for the 16-row tiles that contain one (every other layer feeds all rows' keys and values to the
                                         next).  Bitwise the same trajectory; the work skipped depends on the mask - with one CDR-like
                                         segment per patch 5-7 of the 8 row tiles of the last layer - so it is opt-in and bench.py's
                                         headline keeps it off (reported separately). */

/* Model and batch geometry.  Reference ctor: diffab_pytorch.py:629-647. */
typedef struct {
  int32_t B;  /* patches in this call */
  int32_t K;  /* residues per patch */
  int32_t D;  /* d_residue_emb */
  int32_t C;  /* d_pair_emb */
  int32_t H;  /* n_head */
  int32_t DS; /* d_scalar_per_head */
  int32_t PQ; /* n_query_point_per_head */
  int32_t PV; /* n_value_point_per_head */
  int32_t NL; /* n_ipa_layers */
  int32_t V;  /* aa vocabulary (21; reference diffusion.py:47) */
} diffab_dims;

/* One InvariantPointAttentionLayer's parameters in nn.Linear layout (out x in),
 * i.e. pointers straight into the reference's state_dict tensors
 * (diffab_pytorch.py:354-379; keys: SURVEY Appendix B.3). */
typedef struct {
  const float* gamma;  /* (H)                     raw, no softplus (:373) */
  const float* wq_s;   /* (H*DS, D)  to_q_scalar.weight */
  const float* wk_s;   /* (H*DS, D)  to_k_scalar.weight */
  const float* wv_s;   /* (H*DS, D)  to_v_scalar.weight */
  const float* w_bias; /* (H, C)     to_pair_bias.weight */
  const float* wq_p;   /* (H*PQ*3, D) to_q_point.weight */
  const float* wk_p;   /* (H*PQ*3, D) to_k_point.weight */
  const float* wv_p;   /* (H*PV*3, D) to_v_point.weight */
  const float* w_out;  /* (D, H*DS + H*C + H*PV*3 + H*PV) to_out.weight */
  const float* b_out;  /* (D) to_out.bias */
} diffab_ipa_layer_weights;

/* Linear-ReLU-Linear-ReLU-Linear head (diffab_pytorch.py:533-556). */
typedef struct {
  const float *w0, *b0; /* (D, D+3), (D) */
  const float *w2, *b2; /* (D, D), (D) */
  const float *w4, *b4; /* (n_out, D), (n_out) */
} diffab_mlp3_weights;

/* Denoiser parameters (diffab_pytorch.py:501-556). `layers` is a HOST array of NL. */
typedef struct {
  const float* seq_emb;           /* (25, D) sequence_embedding.weight (:514) */
  const float *res_w0, *res_b0;   /* (D, 2D), (D)  to_res_emb.0 */
  const float *res_w2, *res_b2;   /* (D, D), (D)   to_res_emb.2 */
  const diffab_ipa_layer_weights* layers;
  diffab_mlp3_weights coord;      /* coordinate_denoising  -> 3 */
  diffab_mlp3_weights orient;     /* orientation_denoising -> 3 */
  diffab_mlp3_weights seq;        /* sequence_denoising    -> V (softmax applied by the kernel) */
} diffab_denoiser_weights;

/* Variance schedule on the device: five (T+1) float arrays
 * (diffusion.py:11-35; keys alpha, alpha_bar, alpha_bar_sqrt,
 * one_minus_alpha_bar_sqrt, beta). */
typedef struct {
  int32_t T;
  const float* alpha;
  const float* alpha_bar;
  const float* alpha_bar_sqrt;
  const float* one_minus_alpha_bar_sqrt;
  const float* beta;
} diffab_sched;

/* IGSO3 tables on the device (so3.py:9-72): one row per sigma. */
typedef struct {
  int32_t n_sigmas;
  int32_t n_bins;
  const float* sigmas;   /* (n_sigmas) */
  const float* cdf;      /* (n_sigmas, n_bins) normalised inclusive prefix sums of the pdf rows */
  float sigma_threshold; /* histogram branch iff sigma < threshold (so3.py:122-125) */
} diffab_igso3;

const char* diffab_version(void);
const char* diffab_last_error(void);
/* 1 if a gfx950 device is visible to this process, else 0 (no error). */
int diffab_device_ok(void);

/* Opt-in diagnostics for bench.py's roofline leg: while enabled, every launch of the dominant kernel (the IPA
 * attention kernel) is bracketed by a hipEvent pair recorded on its launch stream.  read() waits for the events,
 * returns the number of launches and their summed duration, and resets the counter.  Not thread-safe; off by default. */
int diffab_kernel_timer_enable(int on);
/* Diagnostics only: while a device buffer of (work-groups x 8 waves x 8) uint64 is registered, the fused attention kernel
 * writes s_memtime stamps at its phase boundaries into it (tools/attn_phase_profile.py).  NULL (default) disables it. */
int diffab_debug_set_attn_stamps(void* device_buffer);
/* Diagnostics of the patch-resident module kernel (DIFFAB_FLAG_PERSISTENT_MODULE): its start-up stagger (work-groups of class
 * (index / 8) % classes start class x ticks late, ticks of 10 ns; default 8 x 1000: eight classes 10 us apart),
 * and a stamp buffer of (B NL 8 tiles x 8 waves x 8) + (B NL 4) uint64 filled with 100 MHz s_memrealtime stamps (NULL: off). */
int diffab_debug_set_attn_variant(int32_t v); /* A/B switches (tests, tools; process-global): bit 0 = four-wave work-groups in the plane
                                                 attention kernel (two per CU; measured slower), bit 2 (4) = the PairEmbedding forward / backward
                                                 as their unfused launches where the fused kernel would apply, 8 (alone) = the six
                                                 projections and to_out as six-term bf16 split products (rounds 3-4) instead of the
                                                 three-term fp16 ones (and with them the backward's d feat and weight-gradient products) -
                                                 per-layer launches only, bit 5 (32) = only the weight-gradient products of the training
                                                 backward in the six-term bf16 form, bit 6 (64) = the PairEmbedding backward's matrix-core
                                                 kernels (csrc/pair_chain_bwd.hip: 64-wide chain, one-hot table / coefficient sums) as the
                                                 separate launches they replaced, bit 4 (16) = value planes: a pass after the
                                                 projections cuts the value side (v_s, global value points) into two fp16 planes and phase 3
                                                 of the attention tile (P x V) runs on the f16 matrix cores (parity-green, measured slower
                                                 overall: profiles/r06_attention.md).  0 = defaults. */
int diffab_debug_set_module_stagger(int32_t ticks_10ns, int32_t classes);
int diffab_debug_set_module_stamps(void* device_buffer);
/* The cross-stream ordering guard described under "Streams" above: on / off (default since round 6), process-wide. */
int diffab_set_stream_guard(int on);
/* Diagnostics / accuracy tests: Y[M x 128] = X[M x Kd] W[128 x Kd]^T + bias through ONE of the two dense kernels of the MFMA path -
 * mode 0: f32-input MFMA (rowgemm128_kernel), mode 1: bf16 matrix cores, six-term split (rowgemm128_b6_kernel; scratch >=
 * 3 * 128 * Kd * 2 bytes, 16-byte aligned operands), mode 2: f16 matrix cores, three-term split under power-of-two scales
 * (rowgemm128_h3_kernel; scratch >= 2 * 128 * Kd * 2 + 768 bytes; Kd a multiple of 64).  Modes 0 and 1: Kd a multiple of 32.  Lets a test measure them against float64. */
int diffab_debug_linear128(const float* X, const float* W, const float* bias, float* Y, int64_t M, int32_t Kd, int32_t mode, void* scratch,
                           size_t scratch_bytes, void* stream);
/* Diagnostics / accuracy tests: Y[M x N] = X[M x 128] W[128 x N] (row-major, 16-byte aligned X) through one of the two x-stationary kernels
 * the training backward uses for d feat = d y W_out (reference: autograd of diffab_pytorch.py:459-464) - mode 1: bf16 matrix cores, six-term
 * split (proj_frames_b6_kernel without frames), mode 2: f16 matrix cores, three-term split under power-of-two scales (xstat_h3_kernel).
 * Any N >= 1, any M >= 1; scratch: 16-byte aligned, >= 3 * 2 * ceil(N / 96) * 96 * 128 * 2 + ceil(N / 96) * 96 * 4 bytes. */
int diffab_debug_xstat128(const float* X, const float* W, float* Y, int64_t M, int32_t N, int32_t mode, void* scratch, size_t scratch_bytes,
                          void* stream);
/* Diagnostics / accuracy tests: C[N1 x N2] += A[M x N1]^T B[M x N2] (row-major, contraction over the rows) through one of the two
 * weight-gradient kernels of the training backward (reference: autograd of every nn.Linear, e.g. diffab_pytorch.py:375-379, :459-464) -
 * mode 1: bf16 matrix cores, six-term split (gemm_tn_b6_kernel), mode 2: f16 matrix cores, three-term split with one power-of-two
 * scale per (32-row slab, operand) (gemm_tn_h3_kernel).  db (nullable): db[N1] += column sums of A.  Any M, N1, N2 >= 1. */
int diffab_debug_gemm_tn(const float* A, const float* B, float* C, float* db, int64_t M, int32_t N1, int32_t N2, int32_t mode, void* stream);
int diffab_kernel_timer_read(int64_t* launches, double* total_ms);
/* ---- SO(3) maps, n matrices/vectors each --------------------------------- */
/* so3.py:146-162  log R = theta/(2 sin theta) (R - R^T); NaN at theta = 0 like the reference */
int diffab_so3_log(const float* R, float* S, int64_t n, void* stream);
/* so3.py:219-237  exp of a skew-symmetric matrix (Rodrigues); NaN at |v| = 0 like the reference */
int diffab_so3_exp(const float* S, float* R, int64_t n, void* stream);
/* so3.py:173-182 */
int diffab_so3_matrix_to_rotvec(const float* R, float* v, int64_t n, void* stream);
/* so3.py:207-216 */
int diffab_so3_rotvec_to_matrix(const float* v, float* R, int64_t n, void* stream);
/* so3.py:240-259  exp(k log R); k has n/per_k entries, k[i / per_k] scales matrix i */
int diffab_so3_scale_rot(const float* R, const float* k, float* out, int64_t n, int64_t per_k, void* stream);

/* ---- IGSO3 ---------------------------------------------------------------- */
/* so3.py:52-72  pdf[n_sigmas][n_bins] at the bin centres, series of num_iters terms, NaN->0, <0 -> 0.  The reference's table:
 * every term formed with the reference's own fp32 roundings (its rounding noise, rectified by the clamp, is part of the
 * distribution it samples from); the terms are added in fp32 in the order of torch's CPU cascade sum for this reduction (ATen
 * SumKernel multi_row_sum: four levels), csrc/diffusion_kernels.hip igso3_pdf_kernel<true>. */
int diffab_igso3_table_build(const float* sigmas, int32_t n_sigmas, int32_t n_bins, int32_t num_iters, float* pdf, void* stream);
/* opt-in variant: the whole series in float64, rounded once - the exact density, NOT what the reference samples from */
int diffab_igso3_table_build_accurate(const float* sigmas, int32_t n_sigmas, int32_t n_bins, int32_t num_iters, float* pdf,
                                      void* stream);
/* build-defined: cdf rows = normalised inclusive prefix sums (float64 accumulate) of the pdf rows */
int diffab_igso3_cdf_build(const float* pdf, int32_t n_sigmas, int32_t n_bins, float* cdf, void* stream);
/* so3.py:98-126  rot-vectors (B,K,3) = normalize(axis_raw) * theta, theta from the histogram row
 * sigma_idx[b] (inverse CDF on u_bin, uniform in the bin by u_in) if sigma < threshold, else
 * (2 sigma + sigma z) mod pi.  axis_raw (B,K,3), u_bin/u_in/z (B,K). */
int diffab_igso3_sample(const diffab_igso3* tab, const int64_t* sigma_idx, int32_t B, int32_t K, const float* axis_raw,
                        const float* u_bin, const float* u_in, const float* z, float* rotvec, void* stream);

/* so3.py:78  `torch.multinomial(probs, num_samples)` draws the K bins of a patch WITHOUT replacement.  bins (B,K) int32 = the K
 * bins of histogram row sigma_idx[b] with the largest pdf[bin] / race[b][bin], largest first, ties by lower bin: with race (B, n_bins)
 * ~ Exp(1) this is a draw without replacement in draw order (the exponential race torch itself uses on a GPU).  n_bins <= 16384.
 * sigmas (n_sigmas, nullable) + sigma_threshold: rows with sigma >= threshold use the Gaussian angle and never read their bins
 * (so3.py:122-125) - they are skipped (bins 0) instead of sorted; NULL sorts every row.  Cost of a sorted row: one 1024-thread
 * work-group, a bitonic network over the 8192 keys in LDS. */
int diffab_igso3_bins_without_replacement(const float* pdf, int32_t n_sigmas, int32_t n_bins, const int64_t* sigma_idx, int32_t B,
                                          int32_t K, const float* race, int32_t* bins, const float* sigmas, float sigma_threshold,
                                          void* stream);
/* diffab_igso3_sample with the histogram bins given (bins (B,K) from diffab_igso3_bins_without_replacement) instead of u_bin */
int diffab_igso3_sample_bins(const diffab_igso3* tab, const int64_t* sigma_idx, int32_t B, int32_t K, const float* axis_raw,
                             const int32_t* bins, const float* u_in, const float* z, float* rotvec, void* stream);

/* ---- forward (noising) process, explicit noise ----------------------------- */
/* diffusion.py:38-41  out[b, ...] = w1[b] p1[b, ...] + w2[b] p2[b, ...]; n elements in all, per_b of them per patch */
int diffab_weighted_multinomial(const float* p1, const float* p2, const float* w1, const float* w2, int64_t n, int64_t per_b,
                                float* out, void* stream);
/* diffusion.py:49-79 (mode 0: q(s_t|s_{t-1}), beta_t), :105-135 (mode 1: q(s_t|s_0), alpha_bar_t) */
int diffab_seq_forward_prob(const diffab_sched* s, int mode, const int64_t* seq, const int64_t* t, const uint8_t* mask,
                            int32_t B, int32_t K, float* prob /* (B,K,21) */, void* stream);
/* diffusion.py:168-192 */
int diffab_seq_posterior(const diffab_sched* s, const int64_t* seq_t, const int64_t* seq_0, const int64_t* t,
                         const uint8_t* mask, int32_t B, int32_t K, float* post /* (B,K,21) */, void* stream);
/* diffusion.py:156-158 (torch.multinomial replaced by inverse CDF on the given uniforms) */
int diffab_categorical_sample(const float* prob, const float* u, int64_t n_rows, int32_t V, int64_t* out, void* stream);
/* diffusion.py:199-236 */
int diffab_coord_forward(const diffab_sched* s, const float* x0, const int64_t* t, const uint8_t* mask, const float* eps,
                         int32_t B, int32_t K, float* xt, void* stream);
/* diffusion.py:262-294 */
int diffab_orient_forward(const diffab_sched* s, const float* O0, const uint8_t* mask, const int64_t* t,
                          const float* rotvec, int32_t B, int32_t K, float* Ot, void* stream);

/* ---- counter-based noise (Philox4x32-10) ------------------------------------ */
/* out[(b*K + k)*4 + c], c = 0..3: normals (kind 0) or uniforms in (0,1) (kind 1) for
 * counter (residue k, patch first_patch + b, step, stream_id), key = seed. */
int diffab_philox_fill(uint64_t seed, int64_t first_patch, int32_t B, int32_t K, int32_t step, int32_t stream_id, int kind,
                       float* out, void* stream);

/* ---- IPA / denoiser --------------------------------------------------------- */
size_t diffab_denoise_workspace_bytes(const diffab_dims* d);
size_t diffab_sample_workspace_bytes(const diffab_dims* d); /* for diffab_sample_loop */
/* diffab_pytorch.py:389-465  one InvariantPointAttentionLayer.forward */
int diffab_ipa_layer_fwd(const diffab_dims* d, const diffab_ipa_layer_weights* w, const float* x /* (B,K,D) */,
                         const float* e /* (B,K,K,C) */, const float* R /* (B,K,3,3) */, const float* t /* (B,K,3) */,
                         float* y /* (B,K,D) */, void* workspace, size_t workspace_bytes, uint32_t flags, void* stream);
/* diffab_pytorch.py:558-607  Denoiser.forward == DiffAb.denoise (:726-768).
 * out_logits (B,K,V) pre-softmax and out_res_emb (B,K,D) post-IPA are optional (NULL to skip). */
int diffab_denoise_step_fwd(const diffab_dims* d, const diffab_denoiser_weights* w, const int64_t* seq_t, const float* x_t,
                            const float* O_t, const float* res_ctx, const float* pair_ctx, const float* beta /* (B) */,
                            float* out_eps, float* out_O0, float* out_posterior, float* out_logits, float* out_res_emb,
                            void* workspace, size_t workspace_bytes, uint32_t flags, void* stream);

/* ---- losses (diffab_pytorch.py:610-625, 856-880) ----------------------------- */
/* losses[3] = (seq KL, translation MSE, orientation) each summed over masked residues / #masked residues.
 * One work-group, fixed-order tree reduction: bitwise reproducible. */
int diffab_losses_fwd(const float* pred_post, const float* true_post, const float* pred_eps, const float* true_eps,
                      const float* pred_O0, const float* true_O0, const uint8_t* gen_mask, const uint8_t* res_mask,
                      int32_t B, int32_t K, int32_t V, float* losses3, void* stream);

/* ---- training step of the hot path (diffab_pytorch.py:808-880 minus encode_context; BASELINE config 4) -------------------
 * fwd: Denoiser.forward with every intermediate saved in `tape` (size: diffab_train_tape_bytes) + the three masked losses.
 * bwd: gradients of  upstream3 . (seq KL, translation MSE, orientation loss)  w.r.t. every denoiser parameter, the residue
 *      context and (optionally) the pair context.  `grads` is the weights struct again, pointing at ZERO-INITIALISED buffers
 *      of the parameters' shapes; they, and d_pair_ctx (NULL to skip, else zero-initialised (B,K,K,C)), are accumulated into
 *      (float atomics: sums over residues are order-dependent in the last bits).  d_res_ctx (B,K,D) may be NULL.
 *      upstream3 is a DEVICE pointer to 3 floats (autograd's incoming gradient; no host sync). */
size_t diffab_train_tape_bytes(const diffab_dims* d);
size_t diffab_train_workspace_bytes(const diffab_dims* d);
int diffab_train_step_fwd(const diffab_dims* d, const diffab_denoiser_weights* w, const int64_t* seq_t, const float* x_t,
                          const float* O_t, const float* res_ctx, const float* pair_ctx, const float* beta, const float* true_post,
                          const float* true_eps, const float* true_O0, const uint8_t* gen_mask, const uint8_t* res_mask,
                          float* out_eps, float* out_O0, float* out_posterior, float* losses3, void* tape, size_t tape_bytes,
                          uint32_t flags, void* stream);
int diffab_train_step_bwd(const diffab_dims* d, const diffab_denoiser_weights* w, const diffab_denoiser_weights* grads,
                          const int64_t* seq_t, const float* x_t, const float* O_t, const float* pair_ctx, const float* out_eps,
                          const float* out_O0, const float* out_posterior, const float* true_post, const float* true_eps,
                          const float* true_O0, const uint8_t* gen_mask, const uint8_t* res_mask, const float* upstream3,
                          float* d_res_ctx, float* d_pair_ctx, const void* tape, size_t tape_bytes, void* workspace,
                          size_t workspace_bytes, void* stream);

/* diffab_pytorch.py:610-625  OrientationLoss: elems (n,3,3) = (pred^T target - I)^2 and/or their total (either may be NULL) */
int diffab_orientation_loss(const float* pred, const float* target, int64_t n, float* elems, float* sum1, void* stream);
/* its backward (autograd of :620-625): cotangent per element (g_elems, (n,3,3)) or one device scalar applied to every element
 * (g_total: upstream / (9 n) for reduction "mean", upstream for "sum"); d_pred / d_target (n,3,3), either may be NULL */
int diffab_orientation_loss_bwd(const float* pred, const float* target, int64_t n, const float* g_elems, const float* g_total,
                                float* d_pred, float* d_target, void* stream);

/* diffab_pytorch.py:315-324 euclidean_transform: out = x R + t, and :327-336 inverse_euclidean_transform: out = (x - t) R^T, for
 * points x (B, N heads, L, P, 3), frames R (B, L, 3, 3), t (B, L, 3) broadcast over the heads (row-vector convention).  t may be
 * NULL (rotation only: the x-gradient of the opposite direction). */
int diffab_frames_apply(const float* x, const float* R, const float* t, float* out, int32_t B, int32_t N, int32_t L, int32_t P, void* stream);
int diffab_frames_invert(const float* x, const float* R, const float* t, float* out, int32_t B, int32_t N, int32_t L, int32_t P, void* stream);
/* d R (B, L, 3, 3) and d t (B, L, 3) of the same two maps from the cotangent g_out of their output (each nullable; written, not accumulated):
 * invert = 0: euclidean_transform, 1: inverse_euclidean_transform.  The x-gradient is the other map with t = NULL. */
int diffab_frames_bwd(const float* x, const float* R, const float* t, const float* g_out, int32_t invert, float* dR, float* dt, int32_t B,
                      int32_t N, int32_t L, int32_t P, void* stream);
/* diffab_pytorch.py:20-54 AngularEncoding.forward: n input values -> n x (4 num_funcs + 1) outputs [x, sin(f x), cos(f x)],
 * f = [1 .. num_funcs, 1/1 .. 1/num_funcs] */
int diffab_angular_encoding(const float* x, int64_t n, int32_t num_funcs, float* out, void* stream);
/* its backward (the reference module is plain differentiable torch code): enc = the forward's output, g_out its cotangent, both
 * n x (4 num_funcs + 1); dx[n] = g[0] + sum_k f_k (cos(f_k x) g_sin[k] - sin(f_k x) g_cos[k]) */
int diffab_angular_encoding_bwd(const float* enc, const float* g_out, int64_t n, int32_t num_funcs, float* dx, void* stream);

/* ---- Denoiser.forward / InvariantPointAttentionLayer.forward under autograd (reference :558-607, :389-465 are differentiable) ----
 * Taped forwards (same outputs as diffab_denoise_step_fwd / diffab_ipa_layer_fwd, activations kept in `tape`) and backwards from
 * ARBITRARY cotangents.  Gradient buffers in `grads` and d_pair_ctx / d_e must be zero-filled by the caller (they are accumulated
 * into); NULL cotangents mean zero.  Tape: diffab_train_tape_bytes(d) / diffab_ipa_layer_tape_bytes(d); workspace:
 * diffab_train_workspace_bytes(d) / diffab_ipa_layer_bwd_workspace_bytes(d).  d_x_t (B,K,3) / d_O_t (B,K,3,3) and d_R / d_t: the
 * gradients with respect to the frames (reference: euclidean_transform / inverse_euclidean_transform :315-336 and O_t @ exp(v) :594-596
 * are differentiable in them), WRITTEN by the call, each nullable (the training step never needs them).  They are the gradients of
 * the formulas with R^T = R^-1, i.e. for rotation frames, as on the whole path.  DIFFAB_FLAG_FORCE_GENERIC is ignored by the taped
 * forwards (their backward reads the tape the MFMA path writes). */
int diffab_denoise_step_fwd_taped(const diffab_dims* d, const diffab_denoiser_weights* w, const int64_t* seq_t, const float* x_t,
                                  const float* O_t, const float* res_ctx, const float* pair_ctx, const float* beta, float* out_eps,
                                  float* out_O0, float* out_posterior, void* tape, size_t tape_bytes, uint32_t flags, void* stream);
int diffab_denoise_step_bwd(const diffab_dims* d, const diffab_denoiser_weights* w, const diffab_denoiser_weights* grads,
                            const int64_t* seq_t, const float* x_t, const float* O_t, const float* pair_ctx, const float* out_posterior,
                            const float* d_eps, const float* d_O0, const float* d_posterior, float* d_res_ctx, float* d_pair_ctx,
                            float* d_x_t, float* d_O_t, const void* tape, size_t tape_bytes, void* workspace, size_t workspace_bytes,
                            void* stream);
size_t diffab_ipa_layer_tape_bytes(const diffab_dims* d);
size_t diffab_ipa_layer_bwd_workspace_bytes(const diffab_dims* d);
int diffab_ipa_layer_fwd_taped(const diffab_dims* d, const diffab_ipa_layer_weights* w, const float* x, const float* e, const float* R,
                               const float* t, float* y, void* tape, size_t tape_bytes, uint32_t flags, void* stream);
int diffab_ipa_layer_bwd(const diffab_dims* d, const diffab_ipa_layer_weights* w, const diffab_ipa_layer_weights* grads, const float* e,
                         const float* R, const float* t, const float* dy, float* dx, float* d_e, float* d_R, float* d_t, const void* tape,
                         size_t tape_bytes, void* workspace, size_t workspace_bytes, void* stream);

/* ---- encode_context (reference diffab_pytorch.py:57-312, 680-724) -----------------------------------------
 * Runs once per sample.  atom_mask is float32 (B,K,A) (1 = atom present); context masks are 1 byte per residue, NULL = "not
 * given" (the reference passes None when generate_structure / generate_sequence is False). */
typedef struct {
  int32_t B, K;
  int32_t A;        /* atoms per residue (n_atoms, 15) */
  int32_t D;        /* d_residue_emb */
  int32_t C;        /* d_pair_emb */
  int32_t max_dist; /* max_dist_to_consider (32) */
} diffab_ctx_dims;

typedef struct { /* ResidueEmbedding parameters (:57-79), nn.Linear layout */
  const float *aa_emb, *chain_emb;        /* (21, D), (10, D) */
  const float *w0, *b0, *w2, *b2, *w4, *b4, *w6, *b6; /* mlp: (2D, 2D+21*A*3+39) (D,2D) (D,D) (D,D) */
} diffab_residue_emb_weights;

typedef struct { /* PairEmbedding parameters (:186-218) */
  const float *aa_pair_emb, *relpos_emb, *pair2distcoef; /* (441, C), (2*max_dist+1, C), (441, A*A) */
  const float *dw0, *db0, *dw2, *db2;                    /* distance_embedding: (C, A*A), (C, C) */
  const float *mw0, *mb0, *mw2, *mb2, *mw4, *mb4;        /* mlp: (C, 3C+18), (C, C), (C, C) */
} diffab_pair_emb_weights;

size_t diffab_residue_embedding_workspace_bytes(const diffab_ctx_dims* d);
/* ResidueEmbedding.forward (:81-183) -> out (B,K,D) */
int diffab_residue_embedding_fwd(const diffab_ctx_dims* d, const diffab_residue_emb_weights* w, const int64_t* seq_idx,
                                 const float* xyz /* (B,K,A,3) */, const float* orientations, const float* dihedrals /* (B,K,3) */,
                                 const int64_t* chain_idx, const float* atom_mask, const uint8_t* structure_context_mask,
                                 const uint8_t* sequence_context_mask, float* out, void* workspace, size_t workspace_bytes,
                                 void* stream);
size_t diffab_pair_embedding_workspace_bytes(const diffab_ctx_dims* d);
/* PairEmbedding.forward (:220-312) -> out (B,K,K,C).  residue_idx is (B,K) with batch stride K, or (1,K) with batch stride 0.
 * The structure-context mask has no effect on this module's output in the reference (:292-301) and is not a parameter. */
int diffab_pair_embedding_fwd(const diffab_ctx_dims* d, const diffab_pair_emb_weights* w, const int64_t* seq_idx,
                              const float* distmat /* (B,K,K,A,A) */, const float* pairwise_dihedrals /* (B,K,K,2) */,
                              const int64_t* residue_idx, int32_t residue_idx_batch_stride, const int64_t* chain_idx,
                              const float* atom_mask, const uint8_t* sequence_context_mask, float* out, void* workspace,
                              size_t workspace_bytes, void* stream);
/* Same, with the atom-atom distances taken from the coordinates xyz (B,K,A,3) inside the kernel instead of a materialised
 * distmat (K*K*A*A*4 = 14.7 MB per K=128 patch).  The reference's data layer computes that tensor with protstruc
 * (data.py:76, preprocess_pdb.py:61) and then leaves it out of its batches; d = |xyz[b,i,a] - xyz[b,j,a']|. */
int diffab_pair_embedding_xyz_fwd(const diffab_ctx_dims* d, const diffab_pair_emb_weights* w, const int64_t* seq_idx,
                                  const float* xyz /* (B,K,A,3) */, const float* pairwise_dihedrals /* (B,K,K,2) */,
                                  const int64_t* residue_idx, int32_t residue_idx_batch_stride, const int64_t* chain_idx,
                                  const float* atom_mask, const uint8_t* sequence_context_mask, float* out, void* workspace,
                                  size_t workspace_bytes, void* stream);

/* Featurisation from coordinates (what the reference's data layer computes with protstruc: data.py:75-82, preprocess_pdb.py:60-65):
 * backbone orientations (B,K,3,3) (rows = the residue's local axes: x along CA->C, y in the N-CA-C plane towards N; `global = local @ R
 * + t`), backbone dihedrals (B,K,3) = (phi, psi, omega) with their validity mask (B,K,3), and pairwise dihedrals (B,K,K,2):
 * phi_ij = (C_i, N_j, CA_j, C_j), psi_ij = (N_i, CA_i, C_i, N_j), IUPAC sign.  Any output may be NULL; chain_idx / residue_mask
 * may be NULL (one chain, every residue present).  xyz is (B,K,A,3) with atoms N, CA, C in slots 0, 1, 2.
 * protstruc is not part of the reference tree: these are the geometric definitions, parity with protstruc is unpinned. */
int diffab_featurize_xyz(const float* xyz, const int64_t* chain_idx, const uint8_t* residue_mask, int32_t B, int32_t K, int32_t A,
                         float* orientations, float* backbone_dihedrals, uint8_t* backbone_dihedrals_mask,
                         float* pairwise_dihedrals, void* stream);

/* Backward of the two context encoders (training through encode_context, diffab_pytorch.py:843-854 under autograd).
 * d_out is the gradient w.r.t. the module output; parameter gradients ACCUMULATE (+=) into the buffers of `g`, which has the
 * layout of the weight struct (the caller zero-fills them).  Inputs other than parameters take no gradient.  Nothing is taped:
 * the backward recomputes the forward of each chunk.  PairEmbedding: the reference's own autograd fails on an in-place product
 * (:295-301); this is the gradient of the same forward with that product out of place (it does not reach the output). */
size_t diffab_residue_embedding_bwd_workspace_bytes(const diffab_ctx_dims* d);
int diffab_residue_embedding_bwd(const diffab_ctx_dims* d, const diffab_residue_emb_weights* w, const diffab_residue_emb_weights* g,
                                 const int64_t* seq_idx, const float* xyz, const float* orientations, const float* dihedrals,
                                 const int64_t* chain_idx, const float* atom_mask, const uint8_t* structure_context_mask,
                                 const uint8_t* sequence_context_mask, const float* d_out /* (B,K,D) */, void* workspace,
                                 size_t workspace_bytes, void* stream);
size_t diffab_pair_embedding_bwd_workspace_bytes(const diffab_ctx_dims* d);
/* exactly one of distmat (B,K,K,A,A) / xyz (B,K,A,3) is non-null, as in the two forward entries */
int diffab_pair_embedding_bwd(const diffab_ctx_dims* d, const diffab_pair_emb_weights* w, const diffab_pair_emb_weights* g,
                              const int64_t* seq_idx, const float* distmat, const float* xyz, const float* pairwise_dihedrals,
                              const int64_t* residue_idx, int32_t residue_idx_batch_stride, const int64_t* chain_idx,
                              const float* atom_mask, const uint8_t* sequence_context_mask, const float* d_out /* (B,K,K,C) */,
                              void* workspace, size_t workspace_bytes, void* stream);
/* Taped form of the PairEmbedding pair (round 6; where diffab_pair_embedding_tape_bytes(d) > 0: C = 64, K a multiple of 128, A <= 16):
 * the forward also leaves the four hidden activations of every pair row on `tape` (4 x B K K C floats: 8.6 GB at B = 128, K = 128 - the
 * 288 GB of an MI355X hold it), and the backward reads them instead of recomputing the forward chunk by chunk (-2 ms of 12 at B = 128).
 * Same arguments and results as diffab_pair_embedding_fwd / _xyz_fwd (exactly one of distmat / xyz non-null) and diffab_pair_embedding_bwd;
 * the tape must be 16-byte aligned and stay untouched between the two calls.  Reference: autograd's saved tensors of
 * PairEmbedding.forward, diffab_pytorch.py:186-312. */
size_t diffab_pair_embedding_tape_bytes(const diffab_ctx_dims* d);
int diffab_pair_embedding_fwd_taped(const diffab_ctx_dims* d, const diffab_pair_emb_weights* w, const int64_t* seq_idx, const float* distmat,
                                    const float* xyz, const float* pairwise_dihedrals, const int64_t* residue_idx,
                                    int32_t residue_idx_batch_stride, const int64_t* chain_idx, const float* atom_mask,
                                    const uint8_t* sequence_context_mask, float* out, float* tape, size_t tape_bytes, void* workspace,
                                    size_t workspace_bytes, void* stream);
int diffab_pair_embedding_bwd_taped(const diffab_ctx_dims* d, const diffab_pair_emb_weights* w, const diffab_pair_emb_weights* g,
                                    const int64_t* seq_idx, const float* distmat, const float* xyz, const float* pairwise_dihedrals,
                                    const int64_t* residue_idx, int32_t residue_idx_batch_stride, const int64_t* chain_idx,
                                    const float* atom_mask, const uint8_t* sequence_context_mask, const float* d_out, const float* tape,
                                    size_t tape_bytes, void* workspace, size_t workspace_bytes, void* stream);

/* ---- reverse process (build-defined; reference stub diffab_pytorch.py:770-776) -- */
/* One update t -> t-1 from denoiser outputs with explicit noise (z (B,K,3), rotvec (B,K,3), u_seq (B,K)),
 * in place on (seq, x, O), only where gen_mask is set. */
int diffab_reverse_update(const diffab_sched* s, int32_t t, int64_t* seq, float* x, float* O, const float* eps_hat,
                          const float* O0_hat, const float* posterior, const uint8_t* gen_mask, const float* z,
                          const float* rotvec, const float* u_seq, int32_t B, int32_t K, int32_t V, void* stream);
/* The whole reverse trajectory t = t_start .. t_stop+1 (normally T .. 1) for B patches, all launches
 * enqueued on `stream` with no host synchronisation: denoise step + Philox noise + IGSO3 draw (table
 * over sqrt(beta)) + update.  State (seq, x, O) is updated in place.  Noise is keyed by
 * (seed, first_patch + b, residue, t), so any sharding of a batch over ranks gives identical samples. */
int diffab_sample_loop(const diffab_dims* d, const diffab_denoiser_weights* w, const diffab_sched* s,
                       const diffab_igso3* rev_tab, int64_t* seq, float* x, float* O, const float* res_ctx,
                       const float* pair_ctx, const uint8_t* gen_mask, uint64_t seed, int64_t first_patch, int32_t t_start,
                       int32_t t_stop, void* workspace, size_t workspace_bytes, uint32_t flags, void* stream);
/* Initial state at t = T on generated residues: x ~ N(0,I), O ~ uniform SO(3), s ~ U{0..19} (Philox, step = T+1). */
int diffab_sample_init(int64_t* seq, float* x, float* O, const uint8_t* gen_mask, uint64_t seed, int64_t first_patch,
                       int32_t B, int32_t K, int32_t T, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DIFFAB_HIP_H */
