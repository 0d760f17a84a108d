/* diffab_hip_experimental.h - entry points that exist only in the EXPERIMENTAL build of libdiffab_hip.so
 * (`make -C diffab-pytorch_amd/csrc EXPERIMENTAL=1` -> diffab-pytorch_amd/build_exp/libdiffab_hip.so; tools/ select it with
 * DIFFAB_HIP_LIB).  That build adds, to everything include/diffab_hip.h declares: the kernel variants that were measured and not
 * adopted (csrc/attention_flash.hip, csrc/attention_pipe.hip, csrc/proj_planes.hip: profiles/r03_operand_planes.md) and the
 * environment switches that select them for A/B timing: DIFFAB_ATTN_FLASH, DIFFAB_FLASH_WAVES, DIFFAB_OPERAND_PLANES,
 * DIFFAB_ATTN_PIPE, DIFFAB_FP32_GEMM, DIFFAB_MLP_UNFUSED, DIFFAB_B6_ROWS (read once per process).  Nothing here is part of the
 * drop-in boundary. */
#ifndef DIFFAB_HIP_EXPERIMENTAL_H
#define DIFFAB_HIP_EXPERIMENTAL_H

#include "diffab_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Diagnostics / accuracy tests: the projection kernel of the operand-plane attention path alone (csrc/proj_planes.hip;
 * InvariantPointAttentionLayer.forward, diffab_pytorch.py:391-413 + the frame transform :324): from x (B K x 128), the frames
 * (R, t) and one layer's weights it writes qk_out = the query / key operand planes of the logits product (B K x 1536 floats:
 * three bf16 planes of 64 slots per residue and head, layout in the header of that file) and the v_s / global value-point
 * columns of proj_out (B K x 1344 floats, the other columns untouched).  Benchmark geometry only (D=128, H=8, DS=32, P=8).
 * diffab_debug_proj_planes_scratch_bytes() bytes of scratch, 256-byte aligned. */
size_t diffab_debug_proj_planes_scratch_bytes(const diffab_dims* d);
int diffab_debug_proj_planes(const diffab_dims* d, const diffab_ipa_layer_weights* w, const float* x, const float* R, const float* t,
                             float* qk_out, float* proj_out, void* scratch, size_t scratch_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif
