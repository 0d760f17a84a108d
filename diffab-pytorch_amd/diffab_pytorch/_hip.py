"""ctypes binding of libdiffab_hip.so (C ABI: include/diffab_hip.h).

torch is used here only as the owner of device memory and streams: every call
passes raw device pointers, sizes and the current HIP stream handle.  There is
no CPU or ATen fallback: if the library or a gfx950 device is missing, calls
raise ``HipUnavailable`` loudly.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, Optional, Sequence

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# DIFFAB_HIP_LIB: developer override used by tools/ and experiments/ to load another build of the same C ABI (ablations, variants)
LIB_PATH = os.environ.get("DIFFAB_HIP_LIB") or os.path.join(os.path.dirname(_HERE), "lib", "libdiffab_hip.so")

FLAG_FORCE_GENERIC = 1
FLAG_PAIR_F32 = 64  # sample_loop: keep the fp32 pair stream (no fp16 planes)
FLAG_FP32_GEMM = 128  # forward paths: dense products on the f32-input MFMA kernels instead of the bf16 split
FLAG_PAIR_PLANES = 32  # K = 64 / 128: pair embedding as two fp16 planes, pair-tile products on the f16 matrix cores (always on in sample_loop)
FLAG_GRAPH_SAMPLER = 16  # sample_loop: one captured step replayed as a hipGraph (launch-bound small batches)
FLAG_PERSISTENT_MODULE = 512  # MFMA path, K = 128 / 256, pair planes: the IPA module as one patch-resident launch (bitwise the multi-launch result)
FLAG_MULTI_LAUNCH = 1024  # sample_loop: never choose the patch-resident module launch (bitwise the same samples either way)
FLAG_SKIP_UNUSED_ROWS = 256  # sample_loop: the last layer's attention only for row tiles with a generated residue (same trajectory)


class HipUnavailable(RuntimeError):
    pass


class DiffabHipError(RuntimeError):
    pass


class Dims(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("B", "K", "D", "C", "H", "DS", "PQ", "PV", "NL", "V")]


_fp = C.c_void_p  # device pointers travel as void*


class IpaLayerWeights(C.Structure):
    _fields_ = [(n, _fp) for n in ("gamma", "wq_s", "wk_s", "wv_s", "w_bias", "wq_p", "wk_p", "wv_p", "w_out", "b_out")]


class Mlp3Weights(C.Structure):
    _fields_ = [(n, _fp) for n in ("w0", "b0", "w2", "b2", "w4", "b4")]


class DenoiserWeights(C.Structure):
    _fields_ = [
        ("seq_emb", _fp), ("res_w0", _fp), ("res_b0", _fp), ("res_w2", _fp), ("res_b2", _fp),
        ("layers", C.POINTER(IpaLayerWeights)),
        ("coord", Mlp3Weights), ("orient", Mlp3Weights), ("seq", Mlp3Weights),
    ]


class CtxDims(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("B", "K", "A", "D", "C", "max_dist")]


class ResidueEmbWeights(C.Structure):
    _fields_ = [(n, _fp) for n in ("aa_emb", "chain_emb", "w0", "b0", "w2", "b2", "w4", "b4", "w6", "b6")]


class PairEmbWeights(C.Structure):
    _fields_ = [(n, _fp) for n in ("aa_pair_emb", "relpos_emb", "pair2distcoef", "dw0", "db0", "dw2", "db2", "mw0", "mb0", "mw2", "mb2",
                                   "mw4", "mb4")]


class Sched(C.Structure):
    _fields_ = [("T", C.c_int32), ("alpha", _fp), ("alpha_bar", _fp), ("alpha_bar_sqrt", _fp),
                ("one_minus_alpha_bar_sqrt", _fp), ("beta", _fp)]


class Igso3(C.Structure):
    _fields_ = [("n_sigmas", C.c_int32), ("n_bins", C.c_int32), ("sigmas", _fp), ("cdf", _fp), ("sigma_threshold", C.c_float)]


# every symbol include/diffab_hip.h declares: name -> (restype, argtypes)
_i32, _i64, _u32, _u64, _sz = C.c_int32, C.c_int64, C.c_uint32, C.c_uint64, C.c_size_t
_PD, _PS, _PI = C.POINTER(Dims), C.POINTER(Sched), C.POINTER(Igso3)
SYMBOLS = {
    "diffab_version": (C.c_char_p, []),
    "diffab_last_error": (C.c_char_p, []),
    "diffab_device_ok": (C.c_int, []),
    "diffab_kernel_timer_enable": (C.c_int, [C.c_int]),
    "diffab_debug_set_attn_stamps": (C.c_int, [_fp]),
    "diffab_debug_set_attn_variant": (C.c_int, [_i32]),
    "diffab_debug_set_module_stagger": (C.c_int, [_i32, _i32]),
    "diffab_debug_set_module_stamps": (C.c_int, [_fp]),
    "diffab_set_stream_guard": (C.c_int, [C.c_int]),
    "diffab_debug_linear128": (C.c_int, [_fp, _fp, _fp, _fp, C.c_int64, C.c_int32, C.c_int32, _fp, C.c_size_t, _fp]),
    "diffab_debug_gemm_tn": (C.c_int, [_fp, _fp, _fp, _fp, C.c_int64, C.c_int32, C.c_int32, C.c_int32, _fp]),
    "diffab_debug_xstat128": (C.c_int, [_fp, _fp, _fp, C.c_int64, C.c_int32, C.c_int32, _fp, C.c_size_t, _fp]),
    "diffab_kernel_timer_read": (C.c_int, [C.POINTER(C.c_int64), C.POINTER(C.c_double)]),
    "diffab_so3_log": (C.c_int, [_fp, _fp, _i64, _fp]),
    "diffab_so3_exp": (C.c_int, [_fp, _fp, _i64, _fp]),
    "diffab_so3_matrix_to_rotvec": (C.c_int, [_fp, _fp, _i64, _fp]),
    "diffab_so3_rotvec_to_matrix": (C.c_int, [_fp, _fp, _i64, _fp]),
    "diffab_so3_scale_rot": (C.c_int, [_fp, _fp, _fp, _i64, _i64, _fp]),
    "diffab_igso3_table_build": (C.c_int, [_fp, _i32, _i32, _i32, _fp, _fp]),
    "diffab_igso3_table_build_accurate": (C.c_int, [_fp, _i32, _i32, _i32, _fp, _fp]),
    "diffab_igso3_cdf_build": (C.c_int, [_fp, _i32, _i32, _fp, _fp]),
    "diffab_igso3_sample": (C.c_int, [_PI, _fp, _i32, _i32, _fp, _fp, _fp, _fp, _fp, _fp]),
    "diffab_igso3_bins_without_replacement": (C.c_int, [_fp, _i32, _i32, _fp, _i32, _i32, _fp, _fp, _fp, C.c_float, _fp]),
    "diffab_igso3_sample_bins": (C.c_int, [_PI, _fp, _i32, _i32, _fp, _fp, _fp, _fp, _fp, _fp]),
    "diffab_weighted_multinomial": (C.c_int, [_fp, _fp, _fp, _fp, _i64, _i64, _fp, _fp]),
    "diffab_seq_forward_prob": (C.c_int, [_PS, C.c_int, _fp, _fp, _fp, _i32, _i32, _fp, _fp]),
    "diffab_seq_posterior": (C.c_int, [_PS, _fp, _fp, _fp, _fp, _i32, _i32, _fp, _fp]),
    "diffab_categorical_sample": (C.c_int, [_fp, _fp, _i64, _i32, _fp, _fp]),
    "diffab_coord_forward": (C.c_int, [_PS, _fp, _fp, _fp, _fp, _i32, _i32, _fp, _fp]),
    "diffab_orient_forward": (C.c_int, [_PS, _fp, _fp, _fp, _fp, _i32, _i32, _fp, _fp]),
    "diffab_philox_fill": (C.c_int, [_u64, _i64, _i32, _i32, _i32, _i32, C.c_int, _fp, _fp]),
    "diffab_denoise_workspace_bytes": (_sz, [_PD]),
    "diffab_sample_workspace_bytes": (_sz, [_PD]),
    "diffab_ipa_layer_fwd": (C.c_int, [_PD, C.POINTER(IpaLayerWeights), _fp, _fp, _fp, _fp, _fp, _fp, _sz, _u32, _fp]),
    "diffab_denoise_step_fwd": (C.c_int, [_PD, C.POINTER(DenoiserWeights), _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp,
                                          _fp, _sz, _u32, _fp]),
    "diffab_losses_fwd": (C.c_int, [_fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _i32, _i32, _i32, _fp, _fp]),
    "diffab_train_tape_bytes": (_sz, [_PD]),
    "diffab_train_workspace_bytes": (_sz, [_PD]),
    "diffab_train_step_fwd": (C.c_int, [_PD, C.POINTER(DenoiserWeights)] + [_fp] * 16 + [_sz, _u32, _fp]),
    "diffab_train_step_bwd": (C.c_int, [_PD, C.POINTER(DenoiserWeights), C.POINTER(DenoiserWeights)] + [_fp] * 16 + [_sz, _fp, _sz, _fp]),
    "diffab_residue_embedding_workspace_bytes": (_sz, [C.POINTER(CtxDims)]),
    "diffab_residue_embedding_fwd": (C.c_int, [C.POINTER(CtxDims), C.POINTER(ResidueEmbWeights)] + [_fp] * 10 + [_sz, _fp]),
    "diffab_pair_embedding_workspace_bytes": (_sz, [C.POINTER(CtxDims)]),
    "diffab_pair_embedding_fwd": (C.c_int, [C.POINTER(CtxDims), C.POINTER(PairEmbWeights), _fp, _fp, _fp, _fp, _i32, _fp, _fp, _fp, _fp,
                                            _fp, _sz, _fp]),
    "diffab_pair_embedding_xyz_fwd": (C.c_int, [C.POINTER(CtxDims), C.POINTER(PairEmbWeights), _fp, _fp, _fp, _fp, _i32, _fp, _fp, _fp, _fp,
                                                _fp, _sz, _fp]),
    "diffab_residue_embedding_bwd_workspace_bytes": (_sz, [C.POINTER(CtxDims)]),
    "diffab_residue_embedding_bwd": (C.c_int, [C.POINTER(CtxDims), C.POINTER(ResidueEmbWeights), C.POINTER(ResidueEmbWeights)] + [_fp] * 10
                                     + [_sz, _fp]),
    "diffab_pair_embedding_bwd_workspace_bytes": (_sz, [C.POINTER(CtxDims)]),
    "diffab_pair_embedding_bwd": (C.c_int, [C.POINTER(CtxDims), C.POINTER(PairEmbWeights), C.POINTER(PairEmbWeights), _fp, _fp, _fp, _fp, _fp,
                                            _i32, _fp, _fp, _fp, _fp, _fp, _sz, _fp]),
    "diffab_pair_embedding_tape_bytes": (_sz, [C.POINTER(CtxDims)]),
    # (d, w, seq, distmat, xyz, pdih, resid, stride, chain, amask, mask, out, tape, tape_bytes, ws, ws_bytes, stream)
    "diffab_pair_embedding_fwd_taped": (C.c_int, [C.POINTER(CtxDims), C.POINTER(PairEmbWeights), _fp, _fp, _fp, _fp, _fp, _i32, _fp, _fp, _fp,
                                                  _fp, _fp, _sz, _fp, _sz, _fp]),
    # (d, w, g, seq, distmat, xyz, pdih, resid, stride, chain, amask, mask, d_out, tape, tape_bytes, ws, ws_bytes, stream)
    "diffab_pair_embedding_bwd_taped": (C.c_int, [C.POINTER(CtxDims), C.POINTER(PairEmbWeights), C.POINTER(PairEmbWeights), _fp, _fp, _fp, _fp,
                                                  _fp, _i32, _fp, _fp, _fp, _fp, _fp, _sz, _fp, _sz, _fp]),
    "diffab_featurize_xyz": (C.c_int, [_fp, _fp, _fp, _i32, _i32, _i32, _fp, _fp, _fp, _fp, _fp]),
    "diffab_orientation_loss": (C.c_int, [_fp, _fp, _i64, _fp, _fp, _fp]),
    "diffab_orientation_loss_bwd": (C.c_int, [_fp, _fp, _i64, _fp, _fp, _fp, _fp, _fp]),
    "diffab_frames_apply": (C.c_int, [_fp, _fp, _fp, _fp, _i32, _i32, _i32, _i32, _fp]),
    "diffab_frames_invert": (C.c_int, [_fp, _fp, _fp, _fp, _i32, _i32, _i32, _i32, _fp]),
    "diffab_frames_bwd": (C.c_int, [_fp, _fp, _fp, _fp, _i32, _fp, _fp, _i32, _i32, _i32, _i32, _fp]),
    "diffab_angular_encoding": (C.c_int, [_fp, _i64, _i32, _fp, _fp]),
    "diffab_angular_encoding_bwd": (C.c_int, [_fp, _fp, _i64, _i32, _fp, _fp]),
    "diffab_denoise_step_fwd_taped": (C.c_int, [_PD, C.POINTER(DenoiserWeights)] + [_fp] * 10 + [_sz, _u32, _fp]),
    "diffab_denoise_step_bwd": (C.c_int, [_PD, C.POINTER(DenoiserWeights), C.POINTER(DenoiserWeights)] + [_fp] * 13 + [_sz, _fp, _sz, _fp]),
    "diffab_ipa_layer_tape_bytes": (_sz, [_PD]),
    "diffab_ipa_layer_bwd_workspace_bytes": (_sz, [_PD]),
    "diffab_ipa_layer_fwd_taped": (C.c_int, [_PD, C.POINTER(IpaLayerWeights), _fp, _fp, _fp, _fp, _fp, _fp, _sz, _u32, _fp]),
    "diffab_ipa_layer_bwd": (C.c_int, [_PD, C.POINTER(IpaLayerWeights), C.POINTER(IpaLayerWeights), _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp,
                                       _sz, _fp, _sz, _fp]),
    "diffab_reverse_update": (C.c_int, [_PS, _i32, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _i32, _i32, _i32, _fp]),
    "diffab_sample_loop": (C.c_int, [_PD, C.POINTER(DenoiserWeights), _PS, _PI, _fp, _fp, _fp, _fp, _fp, _fp, _u64, _i64, _i32,
                                     _i32, _fp, _sz, _u32, _fp]),
    "diffab_sample_init": (C.c_int, [_fp, _fp, _fp, _fp, _u64, _i64, _i32, _i32, _i32, _fp]),
}

_lib: Optional[C.CDLL] = None


def load_library() -> C.CDLL:
    """dlopen libdiffab_hip.so and type every declared symbol.  Needs no GPU."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise HipUnavailable(
                f"{LIB_PATH} is missing - build it with `make -C diffab-pytorch_amd/csrc` "
                "(or __graft_entry__.build()); there is no CPU fallback for this path")
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(lib, name)  # AttributeError if the header and the library drift apart
            fn.restype, fn.argtypes = res, args
        _lib = lib
    return _lib


_device_checked = False


def lib() -> C.CDLL:
    """The library, after checking that a gfx950 device is usable (compute calls only)."""
    global _device_checked
    l = load_library()
    if not _device_checked:
        if not torch.cuda.is_available() or not l.diffab_device_ok():
            raise HipUnavailable("no gfx950 (MI355X) device visible: the DiffAb hot path runs on HIP only")
        _device_checked = True
    return l


def check(rc: int, what: str) -> None:
    if rc != 0:
        raise DiffabHipError(f"{what} failed (code {rc}): {load_library().diffab_last_error().decode()}")


def stream_ptr() -> C.c_void_p:
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def device() -> torch.device:
    return torch.device("cuda", torch.cuda.current_device())


def ptr(t: Optional[torch.Tensor]) -> C.c_void_p:
    return C.c_void_p(0 if t is None else t.data_ptr())


def dev_f32(t: torch.Tensor) -> torch.Tensor:
    """float32, contiguous, on the current HIP device (copying only if needed)."""
    return t.detach().to(device=device(), dtype=torch.float32).contiguous()


def dev_i64(t: torch.Tensor) -> torch.Tensor:
    return t.detach().to(device=device(), dtype=torch.int64).contiguous()


def dev_mask(t: torch.Tensor) -> torch.Tensor:
    return t.detach().to(device=device()).ne(0).contiguous()  # torch.bool, 1 byte per element


def make_dims(B, K, D, C_, H, DS, PQ, PV, NL, V=21) -> Dims:
    return Dims(int(B), int(K), int(D), int(C_), int(H), int(DS), int(PQ), int(PV), int(NL), int(V))


def workspace(nbytes: int) -> torch.Tensor:
    # torch's caching allocator owns the buffer; it is stream-ordered with the launches that use it.
    return torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device())


class SchedOnDevice:
    """The five schedule tables (host-computed, see diffusion.cosine_variance_schedule) uploaded once per device."""

    def __init__(self, sched: Dict[str, torch.Tensor]):
        self.T = int(sched["beta"].numel() - 1)
        self.tensors = {k: dev_f32(v) for k, v in sched.items()}
        t = self.tensors
        self.struct = Sched(self.T, ptr(t["alpha"]), ptr(t["alpha_bar"]), ptr(t["alpha_bar_sqrt"]),
                            ptr(t["one_minus_alpha_bar_sqrt"]), ptr(t["beta"]))


def ipa_layer_weights(params: Dict[str, torch.Tensor], keep: list) -> IpaLayerWeights:
    """params: the layer's own named parameters (reference names, diffab_pytorch.py:354-379)."""
    order = ("gamma", "to_q_scalar.weight", "to_k_scalar.weight", "to_v_scalar.weight", "to_pair_bias.weight",
             "to_q_point.weight", "to_k_point.weight", "to_v_point.weight", "to_out.weight", "to_out.bias")
    # (a layer built with use_pair_bias=False has no to_pair_bias: NULL in the struct, C = 0 in its dims)
    ts = [dev_f32(params[k]) if k in params else None for k in order]
    keep.extend(t for t in ts if t is not None)
    return IpaLayerWeights(*[ptr(t) for t in ts])


def mlp3_weights(params: Dict[str, torch.Tensor], prefix: str, keep: list) -> Mlp3Weights:
    ts = [dev_f32(params[prefix + k]) for k in ("0.weight", "0.bias", "2.weight", "2.bias", "4.weight", "4.bias")]
    keep.extend(ts)
    return Mlp3Weights(*[ptr(t) for t in ts])


class DenoiserWeightsOnDevice:
    """Pointer table over a Denoiser's parameters (no copies when they already live on the device)."""

    def __init__(self, params: Dict[str, torch.Tensor], n_layers: int):
        self.keep: list = []
        g = lambda k: dev_f32(params[k])
        base = [g(k) for k in ("sequence_embedding.weight", "to_res_emb.0.weight", "to_res_emb.0.bias", "to_res_emb.2.weight",
                               "to_res_emb.2.bias")]
        self.keep.extend(base)
        self.layers = (IpaLayerWeights * max(n_layers, 1))()
        for l in range(n_layers):
            pre = f"ipa.layers.{l}."
            sub = {k[len(pre):]: v for k, v in params.items() if k.startswith(pre)}
            self.layers[l] = ipa_layer_weights(sub, self.keep)
        self.struct = DenoiserWeights(
            *[ptr(t) for t in base], C.cast(self.layers, C.POINTER(IpaLayerWeights)),
            mlp3_weights(params, "coordinate_denoising.", self.keep),
            mlp3_weights(params, "orientation_denoising.", self.keep),
            mlp3_weights(params, "sequence_denoising.", self.keep),
        )
