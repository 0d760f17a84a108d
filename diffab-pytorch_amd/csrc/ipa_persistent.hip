// ipa_persistent.hip - the IPA module (all NL layers) of a batch of K = 128 or K = 256 patches as ONE patch-resident launch.
//
// BASELINE.json's execution model: one 512-thread work-group per CDR patch.  A work-group owns its patch from the first layer's
// projections to the last layer's to_out - per layer: the six projections of its 128 rows (proj_frames_h3_tile.h), the eight 16-row
// attention tiles (ipa_attn_tile.h), to_out (rowgemm_h3_tile.h) - and the next patch of its queue after that.  Patches never exchange
// data on this path (every einsum of diffab_pytorch.py:416-457 carries `b`; the layer loop :494-498 is per sample), so there is NO
// inter-CU synchronisation: a phase hands its rows to the next one through global memory written and read by the SAME work-group
// (one CU, one vector L1: work-group scope; s_waitcnt vmcnt(0) + s_barrier), and the CUs are free to drift apart - which is the point:
// started together by 18 separate launches, all 256 CUs stream their pair rows in the same 13-15 us and leave HBM idle for the next 25
// (profiles/r03_lockstep.md); here the first items are staggered once per launch and nothing ever re-aligns them.
//
// Same tile bodies, same arithmetic, same summation orders as the multi-launch path: results are bitwise the same (tested), and with
// them the sampler's shard invariance.
#include "common.h"
#include "denoiser_internal.h"
#define AT_STAMP_REALTIME 1  // the diagnostic stamps of this file's kernels use the chip-wide 100 MHz counter (comparable between CUs)
#include "ipa_attn_tile.h"
#include "attn_planes_tile.h"
#include "proj_frames_h3_tile.h"
#include "rowgemm_h3_tile.h"
#include "mlp_chain_tile.h"

namespace diffab {

namespace {
constexpr size_t cmax(size_t a, size_t b) { return a > b ? a : b; }
constexpr size_t kModuleLdsBytes = cmax(cmax(ipa_attn_lds_bytes(8), static_cast<size_t>(kChainLdsBytes)),
                                        cmax(static_cast<size_t>(pjh3::PJ_LDS_BYTES), static_cast<size_t>(h3tile::lds_bytes<128>())));

struct ModuleArgs {
  float* xa;                 // [B K][128]: the module's input (layer 0 reads it), then every odd layer's output
  float* xb;                 // [B K][128]: every even layer's output; the result is in (NL odd ? xb : xa)
  float* proj;               // [B K][1344] workspace
  float* feat;               // [B K][1024] workspace
  const float* pair;         // fp16 planes of the pair embedding (launch_pair_split)
  const float* esc;          // {s, 1 / s} per pair row
  float* vpl;                // value planes of the P x V product (proj_frames_h3_tile.h), null: phase 3 from the fp32 value columns
  float* vsc;                // their scales
  const float* R;            // [B K][9]
  const float* t;            // [B K][3]
  const char* planes;        // per layer: ipa_layer_planes_bytes() (projection planes | to_out planes | w_bias, gamma, b_out)
  size_t layer_stride, pj_off, out_off, wis_off, small_off;  // offsets of the fp16 planes, 1 / scale vectors and small vectors in a layer's block
  unsigned long long* stamps;  // diagnostics (null in production): [item][wave][8] of the attention tiles + [B][NL][4] phase stamps behind them
  // the denoiser's MLPs as phases of the same launch (null emb_X: not fused): the embedding MLP of the patch's rows in front of layer 0
  // (emb_X -> xa), the three heads behind the last layer (module output -> heads.Y[]); mlp_chain_tile.h, bitwise mlp_chain_b6_kernel
  const float* emb_X;
  MlpChainSet emb, heads;
  int B, NL;
  int stagger_ticks, stagger_classes;  // the work-groups of class c = (blockIdx / 8) % classes start c * ticks (10 ns each) late
};
}  // namespace

// KRES: residues per patch - 128 (one 128-row dense tile per patch, single-chunk attention items) or 256 (BASELINE config 5: two dense
// tiles, sixteen attention items of two 128-key chunks with the online softmax across them - ipa_attn_tile<8, true, ...>)
template <bool VPL, int KRES>
__global__ __launch_bounds__(512) void ipa_module_persistent_kernel(const ModuleArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int K = KRES, NTILE = K / TI, DT = K / 128;  // DT: dense 128-row tiles per patch
  static_assert(KRES == 128 || KRES == 256, "patch-resident module: K = 128 or 256");
  const int M = a.B * K;
  if (a.stagger_ticks > 0 && a.stagger_classes > 1) {
    // (every wave waits for itself: no barrier needed, the first phase starts with loads only)
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long wait = static_cast<unsigned long long>((blockIdx.x >> 3) % a.stagger_classes) * a.stagger_ticks;
    while (__builtin_amdgcn_s_memrealtime() - t0 < wait) __builtin_amdgcn_s_sleep(32);
  }
  unsigned long long* pstamps = a.stamps ? a.stamps + static_cast<size_t>(a.B) * a.NL * NTILE * 64 : nullptr;
  auto pstamp = [&](int b, int l, int k) {
    if (pstamps != nullptr && threadIdx.x == 0) pstamps[(static_cast<size_t>(b) * a.NL + l) * 4 + k] = __builtin_amdgcn_s_memrealtime();
  };
#pragma unroll 1
  for (int b = blockIdx.x; b < a.B; b += gridDim.x) {
    if (a.emb_X != nullptr) {  // ---- to_res_emb of the patch's rows (diffab_pytorch.py:519-523, folded concatenation)
#pragma unroll 1
      for (int dt = 0; dt < DT; ++dt) {
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
        chaintile::mlp_chain_tile(reinterpret_cast<__bf16*>(lds), tid, b * DT + dt, a.emb_X, 128, a.emb.c[0].planes[0], a.emb.c[0].planes[1],
                                  nullptr, a.emb.c[0].bias[0], a.emb.c[0].bias[1], nullptr, a.emb.c[0].bias_idx0, a.emb.c[0].bias_div0, 2, 128,
                                  a.xa, 128, M);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
      }
    }
#pragma unroll 1
    for (int l = 0; l < a.NL; ++l) {
      const char* lp = a.planes + static_cast<size_t>(l) * a.layer_stride;
      const float* small = reinterpret_cast<const float*>(lp + a.small_off);  // [w_bias 8 x 64][gamma 8, pad to 64][b_out 128]
      const float* xin = (l & 1) ? a.xb : a.xa;
      float* xout = (l & 1) ? a.xa : a.xb;
      pstamp(b, l, 0);
#pragma unroll 1
      for (int dt = 0; dt < DT; ++dt) {  // ---- the six projections + frames of the patch's rows, 128 at a time
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));  // (an opaque copy per phase: lane-constant addresses must not stay live across the phases)
        pjh3::proj_frames_h3_tile<true, false>(reinterpret_cast<_Float16*>(lds), tid, b * DT + dt, 0, 1, xin,
                                               reinterpret_cast<const _Float16*>(lp + a.pj_off), reinterpret_cast<const float*>(lp + a.wis_off),
                                               a.R, a.t, a.proj, M);
        if (dt + 1 < DT) {  // the next tile's weight stages overwrite this one's LDS
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          __syncthreads();
        }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if constexpr (VPL) {  // ---- the value side of the patch as fp16 planes for phase 3 of its attention items (attn_planes_tile.h)
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
        aplanes::attn_value_planes_tile(lds, tid, b, a.proj, a.t, K, reinterpret_cast<_Float16*>(a.vpl), a.vsc);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        // the scales were written (vector stores) over the previous layer's at the same addresses and are read through the SCALAR cache
        // by phase 3: drop its lines (the vector L1 needs nothing: same CU, written through)
        __builtin_amdgcn_s_dcache_inv();
      }
      pstamp(b, l, 1);
      // ---- attention: the eight row tiles of the patch
#pragma unroll 1
      for (int tile = 0; tile < NTILE; ++tile) {
        ipa_attn_tile<8, (KRES > 128), true, false, 8, VPL>(lds, b, tile, static_cast<unsigned>((b * a.NL + l) * NTILE + tile), a.proj, a.pair, a.R,
                                                            a.t, small, small + 512, a.feat, K / 128, a.stamps, a.esc, nullptr, nullptr,
                                                            reinterpret_cast<const f32x4*>(a.vpl), a.vsc);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();  // the next tile's phase 1 overwrites the image; the last tile's feature rows are complete
      }
      pstamp(b, l, 2);
#pragma unroll 1
      for (int dt = 0; dt < DT; ++dt) {  // ---- to_out
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
        h3tile::rowgemm128_h3_tile<false, 128>(reinterpret_cast<_Float16*>(lds), tid, b * DT + dt, a.feat, AF,
                                               reinterpret_cast<const _Float16*>(lp + a.out_off),
                                               reinterpret_cast<const float*>(lp + a.wis_off) + ANP, small + 576, nullptr, 0, xout, 128, M, AF);
        if (dt + 1 < DT) {
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          __syncthreads();
        }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      pstamp(b, l, 3);
    }
    if (a.emb_X != nullptr) {  // ---- the three heads on the module's output rows (:525-533, beta columns folded into bias tables)
      const float* xfin = (a.NL & 1) ? a.xb : a.xa;
#pragma unroll 1
      for (int hdt = 0; hdt < 3 * DT; ++hdt) {
        const int hd = hdt / DT, dt = hdt % DT;
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
#define HSEL(f) (hd == 0 ? a.heads.c[0].f : hd == 1 ? a.heads.c[1].f : a.heads.c[2].f)
        chaintile::mlp_chain_tile(reinterpret_cast<__bf16*>(lds), tid, b * DT + dt, xfin, 128, HSEL(planes[0]), HSEL(planes[1]), HSEL(planes[2]),
                                  HSEL(bias[0]), HSEL(bias[1]), HSEL(bias[2]), HSEL(bias_idx0), HSEL(bias_div0), 3, HSEL(n_out),
                                  hd == 0 ? a.heads.Y[0] : hd == 1 ? a.heads.Y[1] : a.heads.Y[2],
                                  hd == 0 ? a.heads.ldy[0] : hd == 1 ? a.heads.ldy[1] : a.heads.ldy[2], M);
#undef HSEL
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();  // the next chain (or the next patch's first phase) overwrites the image
      }
    }
  }
}

namespace {
int g_stagger_ticks = 1000, g_stagger_classes = 8;  // 8 classes x 10 us = two attention-tile periods (swept in round 5: 0 | 8 x 2.5 | 8 x 5 |
                                                     // 8 x 10 | 8 x 20 | 16 x 10 | 8 x 30 us -> 2.47 2.48 2.47 2.43 2.44 2.46 2.50 ms per step)
unsigned long long* g_module_stamps = nullptr;
}  // namespace
void set_module_stagger(int ticks, int classes) {
  g_stagger_ticks = ticks;
  g_stagger_classes = classes;
}
void set_module_stamps(void* p) { g_module_stamps = static_cast<unsigned long long*>(p); }

bool ipa_module_persistent_supported(const diffab_dims* d) {
  return fast_path_supported(d) && (d->K == 128 || d->K == 256) && d->NL >= 1 && dense_h3_enabled();  // (the kernel holds the fp16 tiles only)
}

// planes: d->NL x ipa_layer_planes_bytes() (ipa_layer_split_weights); pair_planes: launch_pair_split(); xa in, result in (NL odd ? xb : xa)
// emb_X (optional, with emb and heads): the embedding MLP's input rows - the launch then also runs the embedding MLP (-> xa) and the three
// heads (module output -> heads->Y[]) of every patch
int launch_ipa_module_persistent(const diffab_dims* d, float* xa, float* xb, const float* R, const float* t, float* ws, const void* planes,
                                 const float* pair_planes, hipStream_t st, const float* emb_X, const MlpChainSet* emb,
                                 const MlpChainSet* heads) {
  DIFFAB_REQUIRE(ipa_module_persistent_supported(d) && xa && xb && R && t && ws && planes && pair_planes, DIFFAB_ERR_ARG,
                 "ipa_module_persistent: unsupported operands");
  const size_t rows = static_cast<size_t>(d->B) * d->K;
  ModuleArgs a{};
  a.xa = xa;
  a.xb = xb;
  a.proj = ws;
  a.feat = ws + rows * ANP;
  a.pair = pair_planes + 64;
  a.esc = pair_row_scales(d, pair_planes);
  const bool vpl_on = value_planes_enabled();
  if (vpl_on) ipa_ws_value_planes(d, ws, &a.vpl, &a.vsc);
  a.R = R;
  a.t = t;
  a.planes = static_cast<const char*>(planes);
  a.layer_stride = ipa_layer_planes_bytes();
  a.pj_off = ipa_layer_h3_pj_offset();
  a.out_off = ipa_layer_h3_out_offset();
  a.wis_off = ipa_layer_h3_wis_offset();
  a.small_off = ipa_layer_small_offset();
  a.stamps = g_module_stamps;
  if (emb_X != nullptr) {
    DIFFAB_REQUIRE(emb && heads && d->D == 128 && (reinterpret_cast<uintptr_t>(emb_X) & 15) == 0, DIFFAB_ERR_ARG,
                   "ipa_module_persistent: the fused MLP phases need both chain sets");
    a.emb_X = emb_X;
    a.emb = *emb;
    a.heads = *heads;
  }
  a.B = d->B;
  a.NL = d->NL;
  a.stagger_ticks = g_stagger_ticks;
  a.stagger_classes = g_stagger_classes;
  int dev = 0, ncu = 256;
  if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
  const int grid = d->B < ncu ? d->B : ncu;  // one work-group per CU (149 KiB of LDS each); more patches than CUs: a work-group walks its queue
#define MODULE_LAUNCH(VPL_, K_)                                                                                                     \
  do {                                                                                                                            \
    DIFFAB_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(ipa_module_persistent_kernel<VPL_, K_>),                   \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(kModuleLdsBytes)));        \
    timer_begin(st);                                                                                                              \
    hipLaunchKernelGGL((ipa_module_persistent_kernel<VPL_, K_>), dim3(grid), dim3(512), kModuleLdsBytes, st, a);                  \
    timer_end(st);                                                                                                                \
  } while (0)
  if (d->K == 128) {
    if (vpl_on) MODULE_LAUNCH(true, 128);
    else MODULE_LAUNCH(false, 128);
  } else {
    if (vpl_on) MODULE_LAUNCH(true, 256);
    else MODULE_LAUNCH(false, 256);
  }
#undef MODULE_LAUNCH
  DIFFAB_LAUNCH_CHECK();
  return DIFFAB_OK;
}

}  // namespace diffab
