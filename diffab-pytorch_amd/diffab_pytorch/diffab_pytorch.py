"""DiffAb module surface over the HIP engine (libdiffab_hip.so, gfx950).

Mirrors the reference's ``diffab_pytorch/diffab_pytorch.py`` for the diffusion / denoise hot path:
InvariantPointAttentionLayer (:339-465), InvariantPointAttentionModule (:468-498), Denoiser (:501-607),
OrientationLoss (:610-625) and DiffAb (:628-931) keep their constructor signatures, method names,
output dict keys and ``state_dict`` keys/shapes, so a checkpoint of the
reference loads unchanged and callers need no edits.  The ``nn.Module`` objects only own the
parameters; every forward is one C-ABI call into the HIP library - there is no ATen fallback.

The context encoders ResidueEmbedding / PairEmbedding (the step just before the hot path) also run
on HIP, forward and backward (the reference itself cannot back-propagate through PairEmbedding: in-place product at
diffab_pytorch.py:295-301; the backward here is the gradient of the same forward with that product out of place), so
DiffAb.training_step on a reference batch dict trains all 2 538 468 parameters.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, Optional

import torch
import torch.nn as nn

from . import _hip
from .diffusion import CoordinateDiffuser, OrientationDiffuser, SequenceDiffuser, cosine_variance_schedule
from . import so3 as _so3
from . import features as _features

try:  # LightningModule hooks when Lightning is installed; a plain nn.Module otherwise
    import pytorch_lightning as pl

    _ModuleBase = pl.LightningModule
except ImportError:  # pragma: no cover - this image has no pytorch_lightning

    class _ModuleBase(nn.Module):
        def log_dict(self, *args, **kwargs):
            pass

        def log(self, *args, **kwargs):
            pass


CA_IDX = 1  # protstruc.general.ATOM.CA (reference diffab_pytorch.py:9, :820)


def _named(module: nn.Module) -> Dict[str, torch.Tensor]:
    return dict(module.named_parameters())


def _wants_grad(module: Optional[nn.Module], *tensors) -> bool:
    """True when the caller expects a differentiable result: autograd is on and an input or a parameter requires grad."""
    if not torch.is_grad_enabled():
        return False
    if any(torch.is_tensor(t_) and t_.requires_grad for t_ in tensors):
        return True
    return module is not None and any(p.requires_grad for p in module.parameters())


class _AngularEncodingFn(torch.autograd.Function):
    """AngularEncoding on HIP; differentiable in x like the reference's plain torch code (diffab_angular_encoding_bwd)."""

    @staticmethod
    def forward(ctx, x, num_funcs: int):
        lib = _hip.lib()
        xd = _hip.dev_f32(x)
        out = torch.empty(*xd.shape[:-1], xd.shape[-1] * (4 * num_funcs + 1), dtype=torch.float32, device=xd.device)
        _hip.check(lib.diffab_angular_encoding(_hip.ptr(xd), xd.numel(), num_funcs, _hip.ptr(out), _hip.stream_ptr()),
                   "diffab_angular_encoding")
        ctx.save_for_backward(out)
        ctx.num_funcs, ctx.x_shape, ctx.x_device, ctx.x_dtype = num_funcs, tuple(x.shape), x.device, x.dtype
        return out.to(x.device)

    @staticmethod
    def backward(ctx, g):
        (out,) = ctx.saved_tensors
        lib = _hip.lib()
        gd = _hip.dev_f32(g)
        dx = torch.empty(ctx.x_shape, dtype=torch.float32, device=out.device)
        _hip.check(lib.diffab_angular_encoding_bwd(_hip.ptr(out), _hip.ptr(gd), dx.numel(), ctx.num_funcs, _hip.ptr(dx), _hip.stream_ptr()),
                   "diffab_angular_encoding_bwd")
        return dx.to(device=ctx.x_device, dtype=ctx.x_dtype), None


class AngularEncoding(nn.Module):
    """[x, sin(f x), cos(f x)] with f = [1..n, 1/1..1/n] per input value (reference diffab_pytorch.py:20-54); one HIP kernel, and
    differentiable in x as the reference's torch expression is."""

    def __init__(self, num_funcs=3):
        super().__init__()
        self.num_funcs = num_funcs
        self.freq_bands = torch.tensor([i + 1.0 for i in range(num_funcs)] + [1.0 / (i + 1.0) for i in range(num_funcs)]).float()

    def get_output_dimension(self, d_in):
        return d_in * (self.num_funcs * 2 * 2 + 1)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return _AngularEncodingFn.apply(x, self.num_funcs)


class _FramesFn(torch.autograd.Function):
    """euclidean_transform / inverse_euclidean_transform on HIP; differentiable in the points (the opposite rotation of the cotangent,
    same kernel with t = NULL) and in the frames r, t (diffab_frames_bwd), as the reference's einsums are (:315-336)."""

    @staticmethod
    def forward(ctx, x, r, t, invert: bool):
        lib = _hip.lib()
        xd, rd, td = _hip.dev_f32(x), _hip.dev_f32(r), _hip.dev_f32(t)
        B, N, L, P = xd.shape[:4]
        out = torch.empty_like(xd)
        fn = lib.diffab_frames_invert if invert else lib.diffab_frames_apply
        _hip.check(fn(_hip.ptr(xd), _hip.ptr(rd), _hip.ptr(td), _hip.ptr(out), B, N, L, P, _hip.stream_ptr()), "diffab_frames")
        ctx.invert, ctx.shape, ctx.devs = invert, (B, N, L, P), (x.device, r.device, t.device)
        ctx.save_for_backward(xd, rd, td)
        return out.to(x.device)

    @staticmethod
    def backward(ctx, g):
        lib = _hip.lib()
        xd, rd, td = ctx.saved_tensors
        gd = _hip.dev_f32(g)
        B, N, L, P = ctx.shape
        dx = dr = dt = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(gd)
            fn = lib.diffab_frames_apply if ctx.invert else lib.diffab_frames_invert  # d x = g R^T (apply) | g R (invert)
            _hip.check(fn(_hip.ptr(gd), _hip.ptr(rd), None, _hip.ptr(dx), B, N, L, P, _hip.stream_ptr()), "diffab_frames (backward)")
            dx = dx.to(ctx.devs[0])
        if ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
            dr = torch.empty_like(rd) if ctx.needs_input_grad[1] else None
            dt = torch.empty_like(td) if ctx.needs_input_grad[2] else None
            _hip.check(lib.diffab_frames_bwd(_hip.ptr(xd), _hip.ptr(rd), _hip.ptr(td), _hip.ptr(gd), int(ctx.invert), _hip.ptr(dr), _hip.ptr(dt),
                                             B, N, L, P, _hip.stream_ptr()), "diffab_frames_bwd")
            dr = dr.to(ctx.devs[1]) if dr is not None else None
            dt = dt.to(ctx.devs[2]) if dt is not None else None
        return dx, dr, dt, None


def euclidean_transform(x, r, t):
    """global = x R + t for points x (b, n heads, l, p, 3), r (b, l, 3, 3), t (b, l, 3) (reference diffab_pytorch.py:315-324)."""
    return _FramesFn.apply(x, r, t, False)


def inverse_euclidean_transform(x, r, t):
    """local = (x - t) R^T (reference diffab_pytorch.py:327-336)."""
    return _FramesFn.apply(x, r, t, True)


class _IpaLayerFn(torch.autograd.Function):
    """InvariantPointAttentionLayer.forward under autograd: taped HIP forward + HIP backward from d y (diffab_ipa_layer_fwd_taped /
    diffab_ipa_layer_bwd): gradients of the layer's ten parameters, of x and of the pair embedding."""

    @staticmethod
    def forward(ctx, layer, flags, x, e, r, t, *params):
        lib = _hip.lib()
        names = [n for n, _ in layer.named_parameters()]
        xd, ed, rd, td = (_hip.dev_f32(a) for a in (x, e, r, t))
        B, K = xd.shape[:2]
        d = layer.dims
        dims = _hip.make_dims(B, K, d["D"], d["C"], d["H"], d["DS"], d["PQ"], d["PV"], 1)
        keep: list = []
        w = _hip.ipa_layer_weights(dict(zip(names, params)), keep)
        tape = _hip.workspace(lib.diffab_ipa_layer_tape_bytes(C.byref(dims)))
        y = torch.empty_like(xd)
        _hip.check(lib.diffab_ipa_layer_fwd_taped(C.byref(dims), C.byref(w), _hip.ptr(xd), _hip.ptr(ed), _hip.ptr(rd), _hip.ptr(td), _hip.ptr(y),
                                                  _hip.ptr(tape), tape.numel(), flags, _hip.stream_ptr()), "diffab_ipa_layer_fwd_taped")
        ctx.layer, ctx.names, ctx.dims = layer, names, dims
        ctx.need_e = ctx.needs_input_grad[3]
        ctx.need_frames = (ctx.needs_input_grad[4], ctx.needs_input_grad[5])  # d r, d t (reference :315-336 is differentiable in them)
        ctx.devs = (x.device, e.device, [p.device for p in params], r.device, t.device)
        ctx.save_for_backward(ed, rd, td, tape, *params)
        return y.to(x.device)

    @staticmethod
    def backward(ctx, dy):
        lib = _hip.lib()
        ed, rd, td, tape = ctx.saved_tensors[:4]
        params = ctx.saved_tensors[4:]
        dims = ctx.dims
        keep: list = []
        w = _hip.ipa_layer_weights(dict(zip(ctx.names, params)), keep)
        grads, ctx.layer._flat_grad = _zero_grads_like(params)
        g = _hip.ipa_layer_weights(dict(zip(ctx.names, grads)), keep)
        dyd = _hip.dev_f32(dy)
        dx = torch.empty_like(dyd)
        de = torch.zeros_like(ed) if ctx.need_e else None
        dr = torch.empty_like(rd) if ctx.need_frames[0] else None
        dt = torch.empty_like(td) if ctx.need_frames[1] else None
        ws = _hip.workspace(lib.diffab_ipa_layer_bwd_workspace_bytes(C.byref(dims)))
        _hip.check(lib.diffab_ipa_layer_bwd(C.byref(dims), C.byref(w), C.byref(g), _hip.ptr(ed), _hip.ptr(rd), _hip.ptr(td), _hip.ptr(dyd),
                                            _hip.ptr(dx), _hip.ptr(de), _hip.ptr(dr), _hip.ptr(dt), _hip.ptr(tape), tape.numel(), _hip.ptr(ws),
                                            ws.numel(), _hip.stream_ptr()), "diffab_ipa_layer_bwd")
        x_dev, e_dev, p_devs, r_dev, t_dev = ctx.devs
        return (None, None, dx.to(x_dev), de.to(e_dev) if ctx.need_e else None, dr.to(r_dev) if dr is not None else None,
                dt.to(t_dev) if dt is not None else None) + tuple(gr.to(dv) for gr, dv in zip(grads, p_devs))


class InvariantPointAttentionLayer(nn.Module):
    """Reference IPA layer: no LayerNorm/residual/transition, raw gamma, unmasked (diffab_pytorch.py:339-465)."""

    def __init__(self, d_residue_emb, d_pair_emb, d_scalar_per_head=16, n_query_point_per_head=4, n_value_point_per_head=4, n_head=8,
                 use_pair_bias=True):
        super().__init__()
        self.n_head = n_head
        self.use_pair_bias = use_pair_bias
        # use_pair_bias=False (reference :348-385, not on the DiffAb path): no to_pair_bias, two independent logits, no pair block in
        # to_out's input; on the HIP side that is C = 0 on the any-dims kernels (forward and backward), e is not read
        self.dims = dict(D=d_residue_emb, C=d_pair_emb if use_pair_bias else 0, H=n_head, DS=d_scalar_per_head, PQ=n_query_point_per_head,
                         PV=n_value_point_per_head)
        d_scalar = d_scalar_per_head * n_head
        # creation order = the reference's, so a seeded construction draws identical initial weights
        self.to_q_scalar = nn.Linear(d_residue_emb, d_scalar, bias=False)
        self.to_k_scalar = nn.Linear(d_residue_emb, d_scalar, bias=False)
        self.to_v_scalar = nn.Linear(d_residue_emb, d_scalar, bias=False)
        if use_pair_bias:
            self.to_pair_bias = nn.Linear(d_pair_emb, n_head, bias=False)
        self.to_q_point = nn.Linear(d_residue_emb, n_query_point_per_head * 3 * n_head, bias=False)
        self.to_k_point = nn.Linear(d_residue_emb, n_query_point_per_head * 3 * n_head, bias=False)
        self.to_v_point = nn.Linear(d_residue_emb, n_value_point_per_head * 3 * n_head, bias=False)
        self.gamma = nn.Parameter(torch.log(torch.exp(torch.ones(n_head)) - 1.0))
        self.to_out = nn.Linear(d_scalar + (d_pair_emb * n_head if use_pair_bias else 0) + n_value_point_per_head * 3 * n_head +
                                n_value_point_per_head * n_head, d_residue_emb)

    def forward(self, x, e, r, t, *, flags: int = 0):
        lib = _hip.lib()
        if x.shape[0] == 0 or x.shape[1] == 0:  # empty batch / empty patch: nothing to launch (the reference's einsums return empty too)
            return torch.zeros(x.shape, dtype=torch.float32, device=x.device)
        if _wants_grad(self, x, e, r, t):  # differentiable like the reference's forward (:389-465): taped HIP forward + HIP backward
            return _IpaLayerFn.apply(self, flags & ~_hip.FLAG_PAIR_PLANES, x, e, r, t, *[p for _, p in self.named_parameters()])
        xd, ed, rd, td = (_hip.dev_f32(a) for a in (x, e, r, t))
        B, K = xd.shape[:2]
        d = self.dims
        dims = _hip.make_dims(B, K, d["D"], d["C"], d["H"], d["DS"], d["PQ"], d["PV"], 1)
        keep: list = []
        w = _hip.ipa_layer_weights(_named(self), keep)
        ws = _hip.workspace(lib.diffab_denoise_workspace_bytes(C.byref(dims)))
        y = torch.empty_like(xd)
        _hip.check(lib.diffab_ipa_layer_fwd(C.byref(dims), C.byref(w), _hip.ptr(xd), _hip.ptr(ed), _hip.ptr(rd), _hip.ptr(td), _hip.ptr(y),
                                            _hip.ptr(ws), ws.numel(), flags, _hip.stream_ptr()), "diffab_ipa_layer_fwd")
        return y.to(x.device)


class InvariantPointAttentionModule(nn.Module):
    """x <- layer(x, e, R, t) for each layer, same e/R/t (diffab_pytorch.py:468-498)."""

    def __init__(self, n_layers, d_residue_emb, d_pair_emb, d_scalar_per_head, n_query_point_per_head, n_value_point_per_head, n_head):
        super().__init__()
        self.layers = nn.ModuleList([
            InvariantPointAttentionLayer(d_residue_emb, d_pair_emb, d_scalar_per_head, n_query_point_per_head, n_value_point_per_head, n_head)
            for _ in range(n_layers)
        ])

    def forward(self, res_emb, pair_emb, orientations, translations, *, flags: int = 0):
        dev = res_emb.device
        x, e, r, t = (_hip.dev_f32(a) for a in (res_emb, pair_emb, orientations, translations))
        for layer in self.layers:
            x = layer(x, e, r, t, flags=flags)
        return x.to(dev)


class Denoiser(nn.Module):
    """eps-hat, O0-hat and the aa posterior from (s_t, x_t, O_t, contexts, beta) (diffab_pytorch.py:501-607)."""

    def __init__(self, d_residue_emb, d_pair_emb, n_ipa_layers, d_scalar_per_head, n_query_point_per_head, n_value_point_per_head, n_head,
                 aa_vocab_size):
        super().__init__()
        D = d_residue_emb
        self.dims = dict(D=D, C=d_pair_emb, H=n_head, DS=d_scalar_per_head, PQ=n_query_point_per_head, PV=n_value_point_per_head,
                         NL=n_ipa_layers, V=aa_vocab_size)
        self.sequence_embedding = nn.Embedding(25, D)
        self.to_res_emb = nn.Sequential(nn.Linear(D * 2, D), nn.ReLU(), nn.Linear(D, D))
        self.ipa = InvariantPointAttentionModule(n_ipa_layers, D, d_pair_emb, d_scalar_per_head, n_query_point_per_head,
                                                 n_value_point_per_head, n_head)

        def head(n_out, softmax=False):
            mods = [nn.Linear(D + 3, D), nn.ReLU(), nn.Linear(D, D), nn.ReLU(), nn.Linear(D, n_out)]
            if softmax:
                mods.append(nn.Softmax(dim=-1))
            return nn.Sequential(*mods)

        self.coordinate_denoising = head(3)
        self.orientation_denoising = head(3)
        self.sequence_denoising = head(aa_vocab_size, softmax=True)

    def hip_weights(self) -> _hip.DenoiserWeightsOnDevice:
        return _hip.DenoiserWeightsOnDevice(_named(self), self.dims["NL"])

    def hip_dims(self, B: int, K: int) -> _hip.Dims:
        d = self.dims
        return _hip.make_dims(B, K, d["D"], d["C"], d["H"], d["DS"], d["PQ"], d["PV"], d["NL"], d["V"])

    def forward(self, seq_idx_t, translations_t, orientations_t, res_context_emb, pair_context_emb, beta, generation_mask=None,
                residue_mask=None, *, return_logits: bool = False, flags: int = 0) -> Dict[str, torch.Tensor]:
        # generation_mask / residue_mask are accepted and ignored, exactly like the reference (:566-567).
        lib = _hip.lib()
        if seq_idx_t.shape[0] == 0 or seq_idx_t.shape[1] == 0:  # empty batch: empty outputs, like the reference
            B0, K0 = seq_idx_t.shape[:2]
            dev0, V0 = translations_t.device, self.dims["V"]
            out0 = {"translations_eps": torch.zeros(B0, K0, 3, device=dev0), "orientations_t0": torch.zeros(B0, K0, 3, 3, device=dev0),
                    "seq_posterior": torch.zeros(B0, K0, V0, device=dev0)}
            if return_logits:
                out0["aa_logits"] = torch.zeros(B0, K0, V0, device=dev0)
                out0["res_emb"] = torch.zeros(B0, K0, self.dims["D"], device=dev0)
            return out0
        if not return_logits and _wants_grad(self, res_context_emb, pair_context_emb, translations_t, orientations_t):
            # differentiable like the reference's forward (:558-607): taped HIP forward + HIP backward from the outputs' cotangents
            # (return_logits=True is an inference-only diagnostic of this package: detached outputs)
            eps, O0, post = _DenoiserFn.apply(self, flags & ~_hip.FLAG_PAIR_PLANES, seq_idx_t, translations_t, orientations_t, beta,
                                              res_context_emb, pair_context_emb, *[p for _, p in self.named_parameters()])
            return {"translations_eps": eps, "orientations_t0": O0, "seq_posterior": post}
        out_dev = translations_t.device
        seq = _hip.dev_i64(seq_idx_t)
        x, O, rc, pc, bt = (_hip.dev_f32(a) for a in (translations_t, orientations_t, res_context_emb, pair_context_emb, beta))
        B, K = seq.shape
        dims = self.hip_dims(B, K)
        w = self.hip_weights()
        ws = _hip.workspace(lib.diffab_denoise_workspace_bytes(C.byref(dims)))
        dev = seq.device
        eps = torch.empty(B, K, 3, dtype=torch.float32, device=dev)
        O0 = torch.empty(B, K, 3, 3, dtype=torch.float32, device=dev)
        post = torch.empty(B, K, dims.V, dtype=torch.float32, device=dev)
        logits = torch.empty(B, K, dims.V, dtype=torch.float32, device=dev) if return_logits else None
        h = torch.empty(B, K, dims.D, dtype=torch.float32, device=dev) if return_logits else None
        _hip.check(lib.diffab_denoise_step_fwd(C.byref(dims), C.byref(w.struct), _hip.ptr(seq), _hip.ptr(x), _hip.ptr(O), _hip.ptr(rc),
                                               _hip.ptr(pc), _hip.ptr(bt), _hip.ptr(eps), _hip.ptr(O0), _hip.ptr(post), _hip.ptr(logits),
                                               _hip.ptr(h), _hip.ptr(ws), ws.numel(), flags, _hip.stream_ptr()), "diffab_denoise_step_fwd")
        out = {"translations_eps": eps.to(out_dev), "orientations_t0": O0.to(out_dev), "seq_posterior": post.to(out_dev)}
        if return_logits:
            out["aa_logits"] = logits.to(out_dev)
            out["res_emb"] = h.to(out_dev)
        return out


class _DenoiserFn(torch.autograd.Function):
    """Denoiser.forward under autograd: diffab_denoise_step_fwd_taped, then diffab_denoise_step_bwd from the cotangents of
    (eps-hat, O0-hat, posterior): gradients of every denoiser parameter and of the two context embeddings."""

    @staticmethod
    def forward(ctx, denoiser, flags, seq_t, x_t, O_t, beta, res_ctx, pair_ctx, *params):
        lib = _hip.lib()
        names = [n for n, _ in denoiser.named_parameters()]
        seq = _hip.dev_i64(seq_t)
        x, O, bt, rc, pc = (_hip.dev_f32(a) for a in (x_t, O_t, beta, res_ctx, pair_ctx))
        B, K = seq.shape
        dims = denoiser.hip_dims(B, K)
        w = _hip.DenoiserWeightsOnDevice(dict(zip(names, params)), denoiser.dims["NL"])
        dev = seq.device
        eps = torch.empty(B, K, 3, dtype=torch.float32, device=dev)
        O0 = torch.empty(B, K, 3, 3, dtype=torch.float32, device=dev)
        post = torch.empty(B, K, dims.V, dtype=torch.float32, device=dev)
        tape = _hip.workspace(lib.diffab_train_tape_bytes(C.byref(dims)))
        _hip.check(lib.diffab_denoise_step_fwd_taped(C.byref(dims), C.byref(w.struct), _hip.ptr(seq), _hip.ptr(x), _hip.ptr(O), _hip.ptr(rc),
                                                     _hip.ptr(pc), _hip.ptr(bt), _hip.ptr(eps), _hip.ptr(O0), _hip.ptr(post), _hip.ptr(tape),
                                                     tape.numel(), flags, _hip.stream_ptr()), "diffab_denoise_step_fwd_taped")
        ctx.denoiser, ctx.names, ctx.dims = denoiser, names, dims
        ctx.need = (ctx.needs_input_grad[6], ctx.needs_input_grad[7])
        ctx.need_frames = (ctx.needs_input_grad[3], ctx.needs_input_grad[4])  # d translations_t, d orientations_t
        ctx.devs = (res_ctx.device, pair_ctx.device, [p.device for p in params], x_t.device, O_t.device)
        ctx.save_for_backward(seq, x, O, pc, post, tape, *params)
        out_dev = x_t.device
        return eps.to(out_dev), O0.to(out_dev), post.to(out_dev)

    @staticmethod
    def backward(ctx, g_eps, g_O0, g_post):
        lib = _hip.lib()
        seq, x, O, pc, post, tape = ctx.saved_tensors[:6]
        params = ctx.saved_tensors[6:]
        dims = ctx.dims
        B, K = seq.shape
        w = _hip.DenoiserWeightsOnDevice(dict(zip(ctx.names, params)), ctx.denoiser.dims["NL"])
        grads, ctx.denoiser._flat_grad = _zero_grads_like(params)
        g = _hip.DenoiserWeightsOnDevice(dict(zip(ctx.names, grads)), ctx.denoiser.dims["NL"])
        ce, cO, cp = (None if t_ is None else _hip.dev_f32(t_) for t_ in (g_eps, g_O0, g_post))
        d_rc = torch.empty(B, K, dims.D, dtype=torch.float32, device=seq.device)
        d_pc = torch.zeros_like(pc) if ctx.need[1] else None
        d_x = torch.empty_like(x) if ctx.need_frames[0] else None
        d_O = torch.empty_like(O) if ctx.need_frames[1] else None
        ws = _hip.workspace(lib.diffab_train_workspace_bytes(C.byref(dims)))
        _hip.check(lib.diffab_denoise_step_bwd(C.byref(dims), C.byref(w.struct), C.byref(g.struct), _hip.ptr(seq), _hip.ptr(x), _hip.ptr(O),
                                               _hip.ptr(pc), _hip.ptr(post), _hip.ptr(ce), _hip.ptr(cO), _hip.ptr(cp), _hip.ptr(d_rc),
                                               _hip.ptr(d_pc), _hip.ptr(d_x), _hip.ptr(d_O), _hip.ptr(tape), tape.numel(), _hip.ptr(ws), ws.numel(),
                                               _hip.stream_ptr()), "diffab_denoise_step_bwd")
        rc_dev, pc_dev, p_devs, x_dev, O_dev = ctx.devs
        return (None, None, None, d_x.to(x_dev) if d_x is not None else None, d_O.to(O_dev) if d_O is not None else None, None,
                d_rc.to(rc_dev) if ctx.need[0] else None, d_pc.to(pc_dev) if ctx.need[1] else None) + \
            tuple(gr.to(dv) for gr, dv in zip(grads, p_devs))


class _OrientationLossFn(torch.autograd.Function):
    """OrientationLoss under autograd: diffab_orientation_loss, then diffab_orientation_loss_bwd."""

    @staticmethod
    def forward(ctx, pred, target, reduction):
        lib = _hip.lib()
        p, t = _hip.dev_f32(pred), _hip.dev_f32(target)
        n = p.numel() // 9
        elems = torch.empty_like(p) if reduction == "none" else None
        total = torch.empty(1, dtype=torch.float32, device=p.device)
        _hip.check(lib.diffab_orientation_loss(_hip.ptr(p), _hip.ptr(t), n, _hip.ptr(elems), _hip.ptr(total), _hip.stream_ptr()),
                   "diffab_orientation_loss")
        ctx.reduction, ctx.n = reduction, n
        ctx.devs = (pred.device, target.device)
        ctx.save_for_backward(p, t)
        out = elems if reduction == "none" else (total[0] / float(9 * n) if reduction == "mean" else total[0])
        return out.to(device=pred.device, dtype=pred.dtype)

    @staticmethod
    def backward(ctx, g):
        lib = _hip.lib()
        p, t = ctx.saved_tensors
        gd = _hip.dev_f32(g)
        ge, gt = (gd, None) if ctx.reduction == "none" else (None, (gd / float(9 * ctx.n) if ctx.reduction == "mean" else gd).reshape(1))
        dp = torch.empty_like(p) if ctx.needs_input_grad[0] else None
        dt = torch.empty_like(t) if ctx.needs_input_grad[1] else None
        _hip.check(lib.diffab_orientation_loss_bwd(_hip.ptr(p), _hip.ptr(t), ctx.n, _hip.ptr(ge), _hip.ptr(gt), _hip.ptr(dp), _hip.ptr(dt),
                                                   _hip.stream_ptr()), "diffab_orientation_loss_bwd")
        return (None if dp is None else dp.to(ctx.devs[0]), None if dt is None else dt.to(ctx.devs[1]), None)


class _HotpathTrainStep(torch.autograd.Function):
    """(seq KL, translation MSE, orientation loss) of one noised batch, differentiable w.r.t. every denoiser parameter and
    the two context embeddings.  Forward = diffab_train_step_fwd (Denoiser forward with a saved-activation tape + the masked
    losses of reference diffab_pytorch.py:856-880), backward = diffab_train_step_bwd.  Both are single C-ABI calls."""

    @staticmethod
    def forward(ctx, denoiser, seq_t, x_t, O_t, beta, true_post, true_eps, true_O0, gen_mask, res_mask, res_ctx, pair_ctx, *params):
        lib = _hip.lib()
        names = [n for n, _ in denoiser.named_parameters()]
        seq = _hip.dev_i64(seq_t)
        x, O, bt, tp, te, tO, rc, pc = (_hip.dev_f32(a) for a in (x_t, O_t, beta, true_post, true_eps, true_O0, res_ctx, pair_ctx))
        gm, rm = _hip.dev_mask(gen_mask), _hip.dev_mask(res_mask)
        B, K = seq.shape
        dims = denoiser.hip_dims(B, K)
        w = _hip.DenoiserWeightsOnDevice(dict(zip(names, params)), denoiser.dims["NL"])
        dev = seq.device
        eps = torch.empty(B, K, 3, dtype=torch.float32, device=dev)
        O0 = torch.empty(B, K, 3, 3, dtype=torch.float32, device=dev)
        post = torch.empty(B, K, dims.V, dtype=torch.float32, device=dev)
        losses = torch.empty(3, dtype=torch.float32, device=dev)
        tape = _hip.workspace(lib.diffab_train_tape_bytes(C.byref(dims)))
        _hip.check(lib.diffab_train_step_fwd(C.byref(dims), C.byref(w.struct), _hip.ptr(seq), _hip.ptr(x), _hip.ptr(O), _hip.ptr(rc),
                                             _hip.ptr(pc), _hip.ptr(bt), _hip.ptr(tp), _hip.ptr(te), _hip.ptr(tO), _hip.ptr(gm), _hip.ptr(rm),
                                             _hip.ptr(eps), _hip.ptr(O0), _hip.ptr(post), _hip.ptr(losses), _hip.ptr(tape), tape.numel(), 0,
                                             _hip.stream_ptr()), "diffab_train_step_fwd")
        ctx.denoiser, ctx.names, ctx.dims = denoiser, names, dims
        ctx.need = (ctx.needs_input_grad[10], ctx.needs_input_grad[11])
        ctx.out_devs = (res_ctx.device, pair_ctx.device, [p.device for p in params])
        ctx.save_for_backward(seq, x, O, pc, eps, O0, post, tp, te, tO, gm, rm, tape, *params)
        ctx.mark_non_differentiable(eps, O0, post)
        return losses, eps, O0, post

    @staticmethod
    def backward(ctx, g_losses, _g_eps, _g_O0, _g_post):
        lib = _hip.lib()
        seq, x, O, pc, eps, O0, post, tp, te, tO, gm, rm, tape = ctx.saved_tensors[:13]
        params = ctx.saved_tensors[13:]
        dims = ctx.dims
        B, K = seq.shape
        w = _hip.DenoiserWeightsOnDevice(dict(zip(ctx.names, params)), ctx.denoiser.dims["NL"])
        # one zero-filled flat buffer, one view per parameter (256-byte aligned): a fill per parameter is 100+ tiny launches per step;
        # kept on the module so that the data-parallel all-reduce can run on the bucket itself (DiffAb.gradient_buckets)
        grads, ctx.denoiser._flat_grad = _zero_grads_like(params)
        g = _hip.DenoiserWeightsOnDevice(dict(zip(ctx.names, grads)), ctx.denoiser.dims["NL"])
        up = _hip.dev_f32(g_losses)
        d_rc = torch.empty(B, K, dims.D, dtype=torch.float32, device=seq.device)
        d_pc = torch.zeros_like(pc) if ctx.need[1] else None
        ws = _hip.workspace(lib.diffab_train_workspace_bytes(C.byref(dims)))
        _hip.check(lib.diffab_train_step_bwd(C.byref(dims), C.byref(w.struct), C.byref(g.struct), _hip.ptr(seq), _hip.ptr(x), _hip.ptr(O),
                                             _hip.ptr(pc), _hip.ptr(eps), _hip.ptr(O0), _hip.ptr(post), _hip.ptr(tp), _hip.ptr(te), _hip.ptr(tO),
                                             _hip.ptr(gm), _hip.ptr(rm), _hip.ptr(up), _hip.ptr(d_rc), _hip.ptr(d_pc), _hip.ptr(tape),
                                             tape.numel(), _hip.ptr(ws), ws.numel(), _hip.stream_ptr()), "diffab_train_step_bwd")
        rc_dev, pc_dev, p_devs = ctx.out_devs
        out_params = tuple(gr.to(dv) for gr, dv in zip(grads, p_devs))
        return (None,) * 10 + (d_rc.to(rc_dev) if ctx.need[0] else None, d_pc.to(pc_dev) if ctx.need[1] else None) + out_params


class OrientationLoss(nn.Module):
    """(pred^T target - I)^2; reduction 'none' | 'mean' | 'sum' (diffab_pytorch.py:610-625)."""

    def __init__(self, reduction="mean"):
        super().__init__()
        self.reduction = reduction

    def forward(self, pred_rotmat: torch.Tensor, target_rotmat: torch.Tensor) -> torch.Tensor:
        lib = _hip.lib()
        if _wants_grad(None, pred_rotmat, target_rotmat):
            return _OrientationLossFn.apply(pred_rotmat, target_rotmat, self.reduction)
        p, t = _hip.dev_f32(pred_rotmat), _hip.dev_f32(target_rotmat)
        n = p.numel() // 9
        elems = torch.empty_like(p) if self.reduction == "none" else None
        total = torch.empty(1, dtype=torch.float32, device=p.device)
        _hip.check(lib.diffab_orientation_loss(_hip.ptr(p), _hip.ptr(t), n, _hip.ptr(elems), _hip.ptr(total), _hip.stream_ptr()),
                   "diffab_orientation_loss")
        if self.reduction == "none":
            out = elems
        elif self.reduction == "mean":
            out = total[0] / float(9 * n)
        else:
            out = total[0]
        return out.to(device=pred_rotmat.device, dtype=pred_rotmat.dtype)


def _opt_mask(m):
    return None if m is None else _hip.dev_mask(m)


_RES_KEYS = ("amino_acid_type_embedding.weight", "chain_embedding.weight", "mlp.0.weight", "mlp.0.bias", "mlp.2.weight", "mlp.2.bias",
             "mlp.4.weight", "mlp.4.bias", "mlp.6.weight", "mlp.6.bias")
_PAIR_KEYS = ("aa_pair_type_embedding.weight", "relpos_embedding.weight", "pair2distcoef.weight", "distance_embedding.0.weight",
              "distance_embedding.0.bias", "distance_embedding.2.weight", "distance_embedding.2.bias", "mlp.0.weight", "mlp.0.bias",
              "mlp.2.weight", "mlp.2.bias", "mlp.4.weight", "mlp.4.bias")


def _zero_grads_like(params):
    """One zero-filled flat buffer with a (256-byte aligned) view per parameter: the HIP backward accumulates into it."""
    offs, total = [], 0
    for p in params:
        offs.append(total)
        total += (p.numel() + 63) // 64 * 64
    flat = torch.zeros(total, dtype=torch.float32, device=_hip.device())
    return [flat[o:o + p.numel()].view(p.shape) for o, p in zip(offs, params)], flat


class _ResidueEmbeddingFn(torch.autograd.Function):
    """ResidueEmbedding forward / backward as two C-ABI calls (the backward recomputes the forward: nothing is taped)."""

    @staticmethod
    def forward(ctx, owner, dims, seq, x, O, dh, ch, am, sm, qm, *params):
        lib = _hip.lib()
        ctx.owner = owner
        ts = [_hip.dev_f32(p) for p in params]
        w = _hip.ResidueEmbWeights(*[_hip.ptr(t_) for t_ in ts])
        ws = _hip.workspace(lib.diffab_residue_embedding_workspace_bytes(C.byref(dims)))
        out = torch.empty(dims.B, dims.K, dims.D, dtype=torch.float32, device=seq.device)
        _hip.check(lib.diffab_residue_embedding_fwd(C.byref(dims), C.byref(w), _hip.ptr(seq), _hip.ptr(x), _hip.ptr(O), _hip.ptr(dh),
                                                    _hip.ptr(ch), _hip.ptr(am), _hip.ptr(sm), _hip.ptr(qm), _hip.ptr(out), _hip.ptr(ws),
                                                    ws.numel(), _hip.stream_ptr()), "diffab_residue_embedding_fwd")
        ctx.dims, ctx.masks = dims, (sm, qm)
        ctx.p_devs = [p.device for p in params]
        ctx.save_for_backward(seq, x, O, dh, ch, am, *ts)
        return out

    @staticmethod
    def backward(ctx, d_out):
        lib = _hip.lib()
        seq, x, O, dh, ch, am = ctx.saved_tensors[:6]
        ts = list(ctx.saved_tensors[6:])
        sm, qm = ctx.masks
        dims = ctx.dims
        grads, ctx.owner._flat_grad = _zero_grads_like(ts)
        w = _hip.ResidueEmbWeights(*[_hip.ptr(t_) for t_ in ts])
        g = _hip.ResidueEmbWeights(*[_hip.ptr(t_) for t_ in grads])
        ws = _hip.workspace(lib.diffab_residue_embedding_bwd_workspace_bytes(C.byref(dims)))
        do = _hip.dev_f32(d_out)
        _hip.check(lib.diffab_residue_embedding_bwd(C.byref(dims), C.byref(w), C.byref(g), _hip.ptr(seq), _hip.ptr(x), _hip.ptr(O),
                                                    _hip.ptr(dh), _hip.ptr(ch), _hip.ptr(am), _hip.ptr(sm), _hip.ptr(qm), _hip.ptr(do),
                                                    _hip.ptr(ws), ws.numel(), _hip.stream_ptr()), "diffab_residue_embedding_bwd")
        return (None,) * 10 + tuple(gr.to(dv) for gr, dv in zip(grads, ctx.p_devs))


class _PairEmbeddingFn(torch.autograd.Function):
    """PairEmbedding forward / backward as two C-ABI calls.  The backward is the gradient of the reference's forward with its
    in-place mask product (diffab_pytorch.py:295-301, which makes the reference's own autograd fail) taken out of place."""

    @staticmethod
    def forward(ctx, owner, dims, from_xyz, seq, dm, dh, ri, ri_stride, ch, am, qm, *params):
        lib = _hip.lib()
        ctx.owner = owner
        ts = [_hip.dev_f32(p) for p in params]
        w = _hip.PairEmbWeights(*[_hip.ptr(t_) for t_ in ts])
        ws = _hip.workspace(lib.diffab_pair_embedding_workspace_bytes(C.byref(dims)))
        out = torch.empty(dims.B, dims.K, dims.K, dims.C, dtype=torch.float32, device=seq.device)
        # Taped form (C ABI "Taped form of the PairEmbedding pair"): when a backward will follow and the device has the room, the forward
        # leaves its four hidden activations (8.6 GB at B = 128, K = 128) and the backward does not recompute it.  DIFFAB_PAIR_TAPE=0: off.
        ctx.tape = None
        tape_bytes = lib.diffab_pair_embedding_tape_bytes(C.byref(dims))
        if (tape_bytes and getattr(owner, "_tape_wanted", False) and any(ctx.needs_input_grad) and os.environ.get("DIFFAB_PAIR_TAPE", "1") != "0"
                and torch.cuda.mem_get_info(seq.device)[0] > 2 * tape_bytes):
            ctx.tape = torch.empty(tape_bytes // 4, dtype=torch.float32, device=seq.device)
            _hip.check(lib.diffab_pair_embedding_fwd_taped(C.byref(dims), C.byref(w), _hip.ptr(seq), _hip.ptr(None if from_xyz else dm),
                                                           _hip.ptr(dm if from_xyz else None), _hip.ptr(dh), _hip.ptr(ri), ri_stride, _hip.ptr(ch),
                                                           _hip.ptr(am), _hip.ptr(qm), _hip.ptr(out), _hip.ptr(ctx.tape), tape_bytes, _hip.ptr(ws),
                                                           ws.numel(), _hip.stream_ptr()), "diffab_pair_embedding_fwd_taped")
        else:
            entry = lib.diffab_pair_embedding_xyz_fwd if from_xyz else lib.diffab_pair_embedding_fwd
            _hip.check(entry(C.byref(dims), C.byref(w), _hip.ptr(seq), _hip.ptr(dm), _hip.ptr(dh), _hip.ptr(ri), ri_stride, _hip.ptr(ch),
                             _hip.ptr(am), _hip.ptr(qm), _hip.ptr(out), _hip.ptr(ws), ws.numel(), _hip.stream_ptr()),
                       "diffab_pair_embedding_xyz_fwd" if from_xyz else "diffab_pair_embedding_fwd")
        ctx.dims, ctx.from_xyz, ctx.ri_stride, ctx.qm = dims, from_xyz, ri_stride, qm
        ctx.p_devs = [p.device for p in params]
        ctx.save_for_backward(seq, dm, dh, ri, ch, am, *ts)
        return out

    @staticmethod
    def backward(ctx, d_out):
        lib = _hip.lib()
        seq, dm, dh, ri, ch, am = ctx.saved_tensors[:6]
        ts = list(ctx.saved_tensors[6:])
        dims = ctx.dims
        grads, ctx.owner._flat_grad = _zero_grads_like(ts)
        w = _hip.PairEmbWeights(*[_hip.ptr(t_) for t_ in ts])
        g = _hip.PairEmbWeights(*[_hip.ptr(t_) for t_ in grads])
        ws = _hip.workspace(lib.diffab_pair_embedding_bwd_workspace_bytes(C.byref(dims)))
        do = _hip.dev_f32(d_out)
        if ctx.tape is not None:
            tape, ctx.tape = ctx.tape, None
            _hip.check(lib.diffab_pair_embedding_bwd_taped(C.byref(dims), C.byref(w), C.byref(g), _hip.ptr(seq),
                                                           _hip.ptr(None if ctx.from_xyz else dm), _hip.ptr(dm if ctx.from_xyz else None),
                                                           _hip.ptr(dh), _hip.ptr(ri), ctx.ri_stride, _hip.ptr(ch), _hip.ptr(am), _hip.ptr(ctx.qm),
                                                           _hip.ptr(do), _hip.ptr(tape), tape.numel() * 4, _hip.ptr(ws), ws.numel(),
                                                           _hip.stream_ptr()), "diffab_pair_embedding_bwd_taped")
            del tape
        else:
            _hip.check(lib.diffab_pair_embedding_bwd(C.byref(dims), C.byref(w), C.byref(g), _hip.ptr(seq),
                                                     _hip.ptr(None if ctx.from_xyz else dm), _hip.ptr(dm if ctx.from_xyz else None), _hip.ptr(dh),
                                                     _hip.ptr(ri), ctx.ri_stride, _hip.ptr(ch), _hip.ptr(am), _hip.ptr(ctx.qm), _hip.ptr(do),
                                                     _hip.ptr(ws), ws.numel(), _hip.stream_ptr()), "diffab_pair_embedding_bwd")
        return (None,) * 11 + tuple(gr.to(dv) for gr, dv in zip(grads, ctx.p_devs))


class ResidueEmbedding(nn.Module):
    """Per-residue context embedding (reference diffab_pytorch.py:57-183): same parameters, creation order and forward
    signature; forward and backward are one C-ABI call each (feature gather kernel + four MFMA linears; the backward recomputes them)."""

    def __init__(self, max_n_atoms_per_residue, d_feat):
        super().__init__()
        self.max_n_aa_types = 21
        self.max_n_atoms_per_residue = max_n_atoms_per_residue
        self.d_feat = d_feat
        self.amino_acid_type_embedding = nn.Embedding(self.max_n_aa_types, d_feat)
        self.chain_embedding = nn.Embedding(10, d_feat, padding_idx=0)
        d_in = d_feat + self.max_n_aa_types * max_n_atoms_per_residue * 3 + 3 * (3 * 2 * 2 + 1) + d_feat
        self.mlp = nn.Sequential(nn.Linear(d_in, d_feat * 2), nn.ReLU(), nn.Linear(d_feat * 2, d_feat), nn.ReLU(),
                                 nn.Linear(d_feat, d_feat), nn.ReLU(), nn.Linear(d_feat, d_feat))

    def forward(self, seq_idx, xyz, orientation, dihedrals, chain_idx, atom_mask, structure_context_mask=None,
                sequence_context_mask=None):
        lib = _hip.lib()
        out_dev = xyz.device
        seq, ch = _hip.dev_i64(seq_idx), _hip.dev_i64(chain_idx)
        x, O, dh, am = (_hip.dev_f32(a) for a in (xyz, orientation, dihedrals, atom_mask))
        sm, qm = _opt_mask(structure_context_mask), _opt_mask(sequence_context_mask)
        B, K = seq.shape
        dims = _hip.CtxDims(B, K, self.max_n_atoms_per_residue, self.d_feat, 1, 32)
        p = _named(self)
        out = _ResidueEmbeddingFn.apply(self, dims, seq, x, O, dh, ch, am, sm, qm, *[p[k] for k in _RES_KEYS])
        return out.to(out_dev)


class PairEmbedding(nn.Module):
    """Residue-pair context embedding (reference diffab_pytorch.py:186-312), forward and backward on HIP.  Reference quirks kept: the
    same-chain mask is a product of chain ids (:279) and the structure-context mask never reaches the output (:292-301)."""

    def __init__(self, max_n_atoms_per_residue, d_feat, max_dist_to_consider=32):
        super().__init__()
        self.d_feat = d_feat
        self.max_dist_to_consider = max_dist_to_consider
        self.max_n_atoms_per_residue = max_n_atoms_per_residue
        self.max_n_aa_types = 21
        self.aa_pair_type_embedding = nn.Embedding(self.max_n_aa_types**2, d_feat)
        self.relpos_embedding = nn.Embedding(2 * max_dist_to_consider + 1, d_feat)
        self.pair2distcoef = nn.Embedding(self.max_n_aa_types**2, max_n_atoms_per_residue**2)
        nn.init.zeros_(self.pair2distcoef.weight)
        self.distance_embedding = nn.Sequential(nn.Linear(max_n_atoms_per_residue**2, d_feat), nn.ReLU(), nn.Linear(d_feat, d_feat),
                                                nn.ReLU())
        self.mlp = nn.Sequential(nn.Linear(3 * d_feat + 2 * (2 * 2 * 2 + 1), d_feat), nn.ReLU(), nn.Linear(d_feat, d_feat), nn.ReLU(),
                                 nn.Linear(d_feat, d_feat))

    def forward(self, seq_idx, distmat, dihedrals, residue_idx, chain_idx, atom_mask, structure_context_mask, sequence_context_mask, *,
                xyz=None):
        """distmat (B,K,K,A,A) as in the reference; or distmat=None and xyz=(B,K,A,3): the atom-atom distances are then computed
        inside the kernel and the 14.7 MB/patch tensor is never built."""
        lib = _hip.lib()
        if distmat is None and xyz is None:
            raise ValueError("PairEmbedding.forward needs distmat or xyz")
        from_xyz = distmat is None
        out_dev = (xyz if from_xyz else distmat).device
        seq, ch, ri = _hip.dev_i64(seq_idx), _hip.dev_i64(chain_idx), _hip.dev_i64(residue_idx)
        dm, dh, am = (_hip.dev_f32(a) for a in (xyz if from_xyz else distmat, dihedrals, atom_mask))
        qm = _opt_mask(sequence_context_mask)
        B, K = seq.shape
        A = self.max_n_atoms_per_residue
        dims = _hip.CtxDims(B, K, A, 1, self.d_feat, self.max_dist_to_consider)
        p = _named(self)
        if ri.shape[0] not in (1, B):
            raise ValueError("residue_idx must be (1, K) or (B, K)")
        # (the grad mode is read HERE: inside an autograd Function's forward it is always off, and needs_input_grad ignores it)
        self._tape_wanted = torch.is_grad_enabled()
        out = _PairEmbeddingFn.apply(self, dims, from_xyz, seq, dm, dh, ri, K if ri.shape[0] == B else 0, ch, am, qm,
                                     *[p[k] for k in _PAIR_KEYS])
        return out.to(out_dev)


class DiffAb(_ModuleBase):
    """Drop-in for ``diffab_pytorch.DiffAb`` on the diffusion hot path (reference diffab_pytorch.py:628-931)."""

    def __init__(self, d_residue_emb, d_pair_emb, n_ipa_layers, d_scalar_per_head, n_query_point_per_head, n_value_point_per_head, n_head,
                 T=100, s=0.01, beta_max=0.999, n_atoms=15, aa_vocab_size=21, max_dist_to_consider=32, lr=1e-4, weight_decay=0.0,
                 betas=(0.9, 0.999), *, igso3_without_replacement: bool = True):
        """The reference's constructor (diffab_pytorch.py:629-660).  `igso3_without_replacement` (keyword-only, build-defined switch, default =
        the reference's behaviour): the forward orientation noise draws a patch's K histogram bins without replacement, as
        torch.multinomial does at so3.py:78; False selects independent inverse-CDF draws (what the build-defined reverse sampler uses)."""
        super().__init__()
        self.sched = cosine_variance_schedule(T=T, s=s, beta_max=beta_max)
        self.residue_context_embedding = ResidueEmbedding(n_atoms, d_residue_emb)
        self.pair_context_embedding = PairEmbedding(n_atoms, d_pair_emb, max_dist_to_consider)
        self.denoiser = Denoiser(d_residue_emb, d_pair_emb, n_ipa_layers, d_scalar_per_head, n_query_point_per_head, n_value_point_per_head,
                                 n_head, aa_vocab_size)
        self.seq_diffuser = SequenceDiffuser(T, s, beta_max, aa_vocab_size)
        self.coordinate_diffuser = CoordinateDiffuser(T, s, beta_max)
        self.orientation_diffuser = OrientationDiffuser(T, s, beta_max, igso3_without_replacement=igso3_without_replacement)
        self.aa_loss = nn.KLDivLoss(reduction="none")
        self.coordinate_loss = nn.MSELoss(reduction="none")
        self.orientation_loss = OrientationLoss(reduction="none")
        self.T = T
        self.lr = lr
        self.weight_decay = weight_decay
        self.betas = betas
        self._sched_dev: Optional[_hip.SchedOnDevice] = None
        self._rev_so3: Optional[_so3.SO3] = None

    # ------------------------------------------------------------------ device-side tables
    def _sched_on_device(self) -> _hip.SchedOnDevice:
        if self._sched_dev is None or self._sched_dev.tensors["beta"].device != _hip.device():
            self._sched_dev = _hip.SchedOnDevice(self.sched)
        return self._sched_dev

    def _reverse_so3(self) -> _so3.SO3:
        """IGSO3 table over sigma_t = sqrt(beta_t) for the reverse step (build-defined)."""
        if self._rev_so3 is None or self._rev_so3.histograms.device != _hip.device():
            # (the device sampler of the reverse loop draws by inverse CDF: one table lookup per residue inside reverse_update)
            self._rev_so3 = _so3.SO3(self.sched["beta"].sqrt(), sigma_threshold=0.1, n_bins=8192, num_iters=1024, without_replacement=False)
        return self._rev_so3

    # ------------------------------------------------------------------ reference API
    def encode_context(self, seq_idx_t0, xyz_t0, orientations_t0, backbone_dihedrals, distmat, pairwise_dihedrals, atom_mask, chain_idx,
                       residue_idx, generation_mask, residue_mask, generate_structure: bool = True, generate_sequence: bool = True):
        """Residue and pair context embeddings of the non-generated residues (reference diffab_pytorch.py:680-724)."""
        context_mask = residue_mask.bool() & (~generation_mask.bool())
        structure_context_mask = context_mask if generate_structure else None
        sequence_context_mask = context_mask if generate_sequence else None
        res_context_emb = self.residue_context_embedding(seq_idx_t0, xyz_t0, orientations_t0, backbone_dihedrals, chain_idx, atom_mask,
                                                         structure_context_mask, sequence_context_mask)
        # distmat=None: distances come from xyz_t0 inside the kernel (the reference's batches do not carry distmat, data.py:93-94)
        pair_context_emb = self.pair_context_embedding(seq_idx_t0, distmat, pairwise_dihedrals, residue_idx, chain_idx, atom_mask,
                                                       structure_context_mask, sequence_context_mask,
                                                       xyz=xyz_t0 if distmat is None else None)
        return res_context_emb, pair_context_emb

    def denoise(self, seq_idx_t, translations_t, orientations_t, res_context_emb, pair_context_emb, beta, generation_mask, residue_mask
                ) -> Dict[str, torch.Tensor]:
        """seq_posterior, translations_eps, orientations_t0 for a noisy state (diffab_pytorch.py:726-768)."""
        return self.denoiser(seq_idx_t, translations_t, orientations_t, res_context_emb, pair_context_emb, beta, generation_mask,
                             residue_mask)

    def _add_noise(self, seq_idx_t0, translations_t0, orientations_t0, generation_mask, t) -> Dict[str, torch.Tensor]:
        """Forward-noise all three modalities to timestep t (diffab_pytorch.py:778-806)."""
        seq_idx_t, seq_posterior = self.seq_diffuser.diffuse_from_t0(seq_idx_t0, t, generation_mask, return_posterior=True)
        translations_t, translations_eps = self.coordinate_diffuser.diffuse_from_t0(translations_t0, t, generation_mask, return_eps=True)
        orientations_t = self.orientation_diffuser.diffuse_from_t0(orientations_t0, generation_mask, t)
        return {"seq_idx_t": seq_idx_t, "seq_posterior": seq_posterior, "translations_t": translations_t,
                "translations_eps": translations_eps, "orientations_t": orientations_t}

    def hotpath_losses(self, denoised, noised, orientations_t0, generation_mask, residue_mask):
        """(seq KL, translation MSE, orientation) each over masked residues / #masked residues (diffab_pytorch.py:856-880)."""
        lib = _hip.lib()
        pp, tp = _hip.dev_f32(denoised["seq_posterior"]), _hip.dev_f32(noised["seq_posterior"])
        pe, te = _hip.dev_f32(denoised["translations_eps"]), _hip.dev_f32(noised["translations_eps"])
        pO, tO = _hip.dev_f32(denoised["orientations_t0"]), _hip.dev_f32(orientations_t0)
        gm, rm = _hip.dev_mask(generation_mask), _hip.dev_mask(residue_mask)
        B, K, V = pp.shape
        out = torch.empty(3, dtype=torch.float32, device=pp.device)
        _hip.check(lib.diffab_losses_fwd(_hip.ptr(pp), _hip.ptr(tp), _hip.ptr(pe), _hip.ptr(te), _hip.ptr(pO), _hip.ptr(tO), _hip.ptr(gm),
                                         _hip.ptr(rm), B, K, V, _hip.ptr(out), _hip.stream_ptr()), "diffab_losses_fwd")
        return out[0], out[1], out[2]

    def hotpath_train_losses(self, noised, res_context_emb, pair_context_emb, beta, orientations_t0, generation_mask, residue_mask):
        """Differentiable (seq, translation, orientation) losses of a noised batch: denoise + losses in one taped HIP forward,
        gradients for the denoiser parameters and both contexts in one HIP backward (reference :843-880 under autograd)."""
        params = [p for _, p in self.denoiser.named_parameters()]
        losses, *_ = _HotpathTrainStep.apply(self.denoiser, noised["seq_idx_t"], noised["translations_t"], noised["orientations_t"], beta,
                                             noised["seq_posterior"], noised["translations_eps"], orientations_t0, generation_mask,
                                             residue_mask, res_context_emb, pair_context_emb, *params)
        out_dev = noised["translations_t"].device
        return losses[0].to(out_dev), losses[1].to(out_dev), losses[2].to(out_dev)

    def _shared_step(self, batch, batch_idx):
        """t ~ U[1,T]; noise; denoise; three losses (diffab_pytorch.py:808-880).  The contexts come from encode_context on
        the reference's batch dict, or from batch['res_context_emb'] / batch['pair_context_emb'] when a caller
        has them already (the hot-path benchmarks and gradient goldens, where contexts are leaf inputs)."""
        dev_in = batch["generation_mask"].device
        bsz = batch["generation_mask"].size(0)
        t_host = torch.randint(low=1, high=self.T + 1, size=(bsz,))  # CPU generator, as the reference (:813); the schedule lookup stays on
        beta = self.sched["beta"][t_host].to(dev_in)                  # the host: no device -> host copy (a stream drain) per step
        t = t_host.to(dev_in)
        xyz_t0 = batch["xyz"]
        if "orientations" not in batch and xyz_t0.dim() == 4:  # frames from the backbone atoms
            batch = dict(batch, **_features.featurize(xyz_t0, orientations=True, backbone_dihedrals=False, pairwise_dihedrals=False))
        translations_t0 = xyz_t0[:, :, CA_IDX] if xyz_t0.dim() == 4 else xyz_t0
        noised = self._add_noise(batch["seq_idx"], translations_t0, batch["orientations"], batch["generation_mask"], t)
        if "res_context_emb" in batch and "pair_context_emb" in batch:
            res_ctx, pair_ctx = batch["res_context_emb"], batch["pair_context_emb"]
        else:
            if "backbone_dihedrals" not in batch or "pairwise_dihedrals" not in batch:  # from xyz, on the device
                batch = dict(batch, **_features.featurize(xyz_t0, batch["chain_idx"], batch["residue_mask"], orientations=False,
                                                          backbone_dihedrals="backbone_dihedrals" not in batch,
                                                          pairwise_dihedrals="pairwise_dihedrals" not in batch))
            res_ctx, pair_ctx = self.encode_context(batch["seq_idx"], xyz_t0, batch["orientations"], batch["backbone_dihedrals"],
                                                    batch.get("distmat"), batch["pairwise_dihedrals"], batch["atom_mask"], batch["chain_idx"],
                                                    batch["residue_idx"], batch["generation_mask"], batch["residue_mask"])
        if torch.is_grad_enabled():
            return self.hotpath_train_losses(noised, res_ctx, pair_ctx, beta, batch["orientations"], batch["generation_mask"],
                                             batch["residue_mask"])
        denoised = self.denoise(noised["seq_idx_t"], noised["translations_t"], noised["orientations_t"], res_ctx, pair_ctx, beta,
                                batch["generation_mask"], batch["residue_mask"])
        return self.hotpath_losses(denoised, noised, batch["orientations"], batch["generation_mask"], batch["residue_mask"])

    def training_step(self, batch, batch_idx):
        seq_loss, translations_loss, orientations_loss = self._shared_step(batch, batch_idx)
        loss = seq_loss + translations_loss + orientations_loss
        self.log_dict({"train/seq_loss": seq_loss, "train/translations_loss": translations_loss,
                       "train/orientations_loss": orientations_loss, "train/loss": loss}, on_step=True, on_epoch=True, prog_bar=True,
                      logger=True)
        return loss

    @torch.no_grad()
    def validation_step(self, batch, batch_idx):
        seq_loss, translations_loss, orientations_loss = self._shared_step(batch, batch_idx)
        loss = seq_loss + translations_loss + orientations_loss
        self.log_dict({"val/seq_loss": seq_loss, "val/translations_loss": translations_loss, "val/orientations_loss": orientations_loss,
                       "val/loss": loss}, on_step=False, on_epoch=True, prog_bar=False, logger=True)
        return loss

    def gradient_buckets(self):
        """The flat fp32 buffers the last HIP backward wrote the parameter gradients into (denoiser, residue encoder, pair encoder):
        after zero_grad(set_to_none=True) + backward every p.grad is a view of one of them.  distributed.allreduce_gradients
        reduces them in place."""
        return [getattr(m, "_flat_grad", None) for m in (self.denoiser, self.residue_context_embedding, self.pair_context_embedding)]

    def configure_optimizers(self):
        # reference :925-931: Adam with these hyper-parameters.  On the device the update runs as torch's fused kernel (one launch over
        # all parameters instead of ~8 multi-tensor launches per step: 0.17 -> 0.03 ms of a 7.6 ms step); same update rule.
        params = list(self.parameters())
        fused = bool(params) and all(p.is_cuda and p.is_floating_point() for p in params)
        return torch.optim.Adam(params, lr=self.lr, weight_decay=self.weight_decay, betas=self.betas, fused=fused)

    # ------------------------------------------------------------------ reverse process (the reference has a stub, :770-776)
    @torch.no_grad()
    def sample(self, seq_idx: torch.LongTensor, xyz: torch.FloatTensor, orientations: torch.FloatTensor, *, generation_mask=None,
               res_context_emb=None, pair_context_emb=None, residue_mask=None, backbone_dihedrals=None, pairwise_dihedrals=None,
               distmat=None, atom_mask=None, chain_idx=None, residue_idx=None, generate_structure: bool = True,
               generate_sequence: bool = True, seed: Optional[int] = None, first_patch: int = 0, t_start: Optional[int] = None,
               t_stop: int = 0, init: bool = True, flags: int = 0, graph: Optional[bool] = None,
               skip_unused_rows: bool = False) -> Dict[str, torch.Tensor]:
        """Reverse diffusion t_start .. t_stop+1 (default T .. 1) on the generated residues (the reference's `sample` is a stub,
        diffab_pytorch.py:770-776; the loop is build-defined).

        seq_idx (B,K), xyz (B,K,3) CA translations or (B,K,A,3) atoms, orientations (B,K,3,3): the ground-truth
        context; generated residues are re-initialised (x ~ N(0,I), O ~ U(SO3), s ~ U{0..19}) when ``init``.
        Contexts: pass ``res_context_emb`` / ``pair_context_emb``, or the remaining fields of the reference's batch dict
        (SURVEY B.2): atom_mask and chain_idx; residue_idx and residue_mask default to arange(K) and all-true; distmat and the two
        dihedral features are taken from xyz on the device when absent (features.featurize) - and `encode_context` runs first, once.
        Noise is Philox keyed by (seed, first_patch + b, residue, t): any sharding of a batch over ranks gives
        the same samples.  All T steps are enqueued on the current stream by ONE C-ABI call, no host sync.
        ``graph=True``: replay one captured step as a hipGraph instead of ~45 launches per step (same kernels, bitwise the same
        result; the call then waits for the trajectory).  Off by default: measured at BASELINE config 1 (B = 1, K = 128, 100 steps)
        it changes nothing - 145 ms eager, 146 ms replayed - because the host already runs ahead of the device there; a step is a
        chain of ~45 dependent kernels on 8-work-group grids (1.45 ms), not 45 launch overheads.
        ``skip_unused_rows=True``: a step's outputs are used for generated residues only, so the LAST layer's attention runs only for the
        16-row tiles that contain one (`DIFFAB_FLAG_SKIP_UNUSED_ROWS`): bitwise the same samples, less work when few residues are
        generated (one CDR: 5-7 of the 8 row tiles of the last layer are skipped)."""
        if generation_mask is None:
            raise ValueError("sample() needs generation_mask: which residues to generate")
        if res_context_emb is None or pair_context_emb is None:
            need = {"atom_mask": atom_mask, "chain_idx": chain_idx}
            missing = [k for k, v in need.items() if v is None]
            if missing or xyz.dim() != 4:
                raise ValueError("sample(): without res_context_emb / pair_context_emb the contexts are computed by encode_context, "
                                 f"which needs all-atom xyz (B,K,A,3) and the batch fields {sorted(need)}; missing: "
                                 f"{missing if missing else 'xyz is not (B,K,A,3)'}")
            Bq, Kq = seq_idx.shape
            if residue_mask is None:
                residue_mask = torch.ones(Bq, Kq, dtype=torch.bool, device=seq_idx.device)
            if backbone_dihedrals is None or pairwise_dihedrals is None:  # dihedral features from the coordinates, on the device
                feats = _features.featurize(xyz, chain_idx, residue_mask, orientations=False, backbone_dihedrals=backbone_dihedrals is None,
                                            pairwise_dihedrals=pairwise_dihedrals is None)
                backbone_dihedrals = feats.get("backbone_dihedrals", backbone_dihedrals)
                pairwise_dihedrals = feats.get("pairwise_dihedrals", pairwise_dihedrals)
            if residue_idx is None:
                residue_idx = torch.arange(Kq, device=seq_idx.device).unsqueeze(0)  # data.py:91
            res_context_emb, pair_context_emb = self.encode_context(seq_idx, xyz, orientations, backbone_dihedrals, distmat,
                                                                    pairwise_dihedrals, atom_mask, chain_idx, residue_idx,
                                                                    generation_mask, residue_mask, generate_structure, generate_sequence)
        lib = _hip.lib()
        out_dev = seq_idx.device
        seq = _hip.dev_i64(seq_idx).clone()
        x = _hip.dev_f32(xyz[:, :, CA_IDX] if xyz.dim() == 4 else xyz).clone()
        O = _hip.dev_f32(orientations).clone()
        rc, pc, gm = _hip.dev_f32(res_context_emb), _hip.dev_f32(pair_context_emb), _hip.dev_mask(generation_mask)
        B, K = seq.shape
        seed = _so3._draw_seed() if seed is None else int(seed)
        t_start = self.T if t_start is None else int(t_start)
        dims = self.denoiser.hip_dims(B, K)
        w = self.denoiser.hip_weights()
        sd = self._sched_on_device()
        tab = self._reverse_so3().struct()
        ws = _hip.workspace(lib.diffab_sample_workspace_bytes(C.byref(dims)))
        if graph:
            flags |= _hip.FLAG_GRAPH_SAMPLER
        if skip_unused_rows:
            flags |= _hip.FLAG_SKIP_UNUSED_ROWS
        if init:
            _hip.check(lib.diffab_sample_init(_hip.ptr(seq), _hip.ptr(x), _hip.ptr(O), _hip.ptr(gm), seed, first_patch, B, K, self.T,
                                              _hip.stream_ptr()), "diffab_sample_init")
        _hip.check(lib.diffab_sample_loop(C.byref(dims), C.byref(w.struct), C.byref(sd.struct), C.byref(tab), _hip.ptr(seq), _hip.ptr(x),
                                          _hip.ptr(O), _hip.ptr(rc), _hip.ptr(pc), _hip.ptr(gm), seed, first_patch, t_start, t_stop,
                                          _hip.ptr(ws), ws.numel(), flags, _hip.stream_ptr()), "diffab_sample_loop")
        return {"seq_idx": seq.to(out_dev), "translations": x.to(out_dev), "orientations": O.to(out_dev)}
