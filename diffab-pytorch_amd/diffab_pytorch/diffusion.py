"""Forward (noising) process of DiffAb on HIP: variance schedule, sequence / translation /
orientation diffusers.  Same classes and method signatures as the reference's
``diffab_pytorch/diffusion.py`` (cosine_variance_schedule :11, SequenceDiffuser :44,
CoordinateDiffuser :195, OrientationDiffuser :239).

The schedule (101 floats) is init-time host arithmetic and is computed with the same torch
expressions as the reference so the five tables are bit-identical; everything per residue runs in
libdiffab_hip.so.  Random draws come from a Philox stream whose seed is taken from torch's default
generator (so ``torch.manual_seed`` reproduces a call), or from explicit noise tensors passed by
keyword (the parity tests inject the reference's own draws that way).
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Dict, Optional

import torch

from . import _hip
from . import so3 as so3

V_AA = 21  # the reference ignores aa_vocab_size and hard-codes 21 (diffusion.py:47)


def cosine_variance_schedule(T: int, s: float = 8e-3, beta_max: float = 0.999) -> Dict[str, torch.Tensor]:
    """Cosine schedule, keys alpha / alpha_bar / alpha_bar_sqrt / one_minus_alpha_bar_sqrt / beta,
    each (T+1,).  alpha_bar is f_t/f_0, NOT the running product of the clipped alphas
    (reference diffusion.py:11-35)."""
    steps = torch.arange(T + 1)
    f_t = torch.cos((steps / T + s) / (1 + s) * math.pi / 2.0).square()
    alpha_bar = f_t / f_t[0]
    beta = torch.cat([torch.tensor([0.0]), torch.clip(1 - alpha_bar[1:] / alpha_bar[:-1], min=1e-5, max=beta_max)])
    return {
        "alpha": 1 - beta,
        "alpha_bar": alpha_bar,
        "alpha_bar_sqrt": alpha_bar.sqrt(),
        "one_minus_alpha_bar_sqrt": (1 - alpha_bar).sqrt(),
        "beta": beta,
    }


def weighted_multinomial(p1: torch.Tensor, p2: torch.Tensor, w1: torch.Tensor, w2: torch.Tensor) -> torch.Tensor:
    """w1 * p1 + w2 * p2 with the (B,) weights broadcast over the trailing (K, V) dimensions (reference diffusion.py:38-41;
    p1 may be an int64 one-hot, promoted to float32 as upstream)."""
    a, b = _hip.dev_f32(p1), _hip.dev_f32(p2)
    if a.shape != b.shape or a.dim() != 3:
        a, b = torch.broadcast_tensors(a, b)
        a, b = a.contiguous(), b.contiguous()
        if a.dim() != 3:
            raise ValueError("weighted_multinomial expects (B, K, V) probabilities")
    wa, wb = _hip.dev_f32(w1), _hip.dev_f32(w2)
    B = a.shape[0]
    if wa.shape != (B,) or wb.shape != (B,):
        raise ValueError("weighted_multinomial expects (B,) weights")
    out = torch.empty_like(a)
    _hip.check(_hip.lib().diffab_weighted_multinomial(_hip.ptr(a), _hip.ptr(b), _hip.ptr(wa), _hip.ptr(wb), a.numel(),
                                                      a.numel() // max(B, 1), _hip.ptr(out), _hip.stream_ptr()), "diffab_weighted_multinomial")
    return out.to(p2.device if torch.is_tensor(p2) else out.device)


class _Diffuser:
    def __init__(self, T, s, beta_max):
        self.sched = cosine_variance_schedule(T, s=s, beta_max=beta_max)
        self._dev: Optional[_hip.SchedOnDevice] = None

    def _sched_dev(self) -> _hip.SchedOnDevice:
        if self._dev is None or self._dev.tensors["beta"].device != _hip.device():
            self._dev = _hip.SchedOnDevice(self.sched)
        return self._dev


def _philox(seed: int, B: int, K: int, stream_id: int, kind: int) -> torch.Tensor:
    out = torch.empty(B, K, 4, dtype=torch.float32, device=_hip.device())
    _hip.check(_hip.lib().diffab_philox_fill(seed, 0, B, K, 0, stream_id, kind, _hip.ptr(out), _hip.stream_ptr()), "diffab_philox_fill")
    return out


class SequenceDiffuser(_Diffuser):
    def __init__(self, T, s=0.01, beta_max=0.999, aa_vocab_size=21):
        super().__init__(T, s, beta_max)
        self.aa_vocab_size = V_AA

    def _prob(self, mode: int, seq_idx, t, generation_mask):
        lib = _hip.lib()
        sd = self._sched_dev()
        seq, tt, m = _hip.dev_i64(seq_idx), _hip.dev_i64(t), _hip.dev_mask(generation_mask)
        B, K = seq.shape
        out = torch.empty(B, K, V_AA, dtype=torch.float32, device=seq.device)
        _hip.check(lib.diffab_seq_forward_prob(C.byref(sd.struct), mode, _hip.ptr(seq), _hip.ptr(tt), _hip.ptr(m), B, K, _hip.ptr(out),
                                               _hip.stream_ptr()), "diffab_seq_forward_prob")
        return out.to(seq_idx.device)

    def forward_prob_single_step(self, seq_idx, t, generation_mask):
        """q(s_t | s_{t-1})  (diffusion.py:49-79)."""
        return self._prob(0, seq_idx, t, generation_mask)

    def forward_prob_from_t0(self, seq_idx_t0, t, generation_mask):
        """q(s_t | s_0)  (diffusion.py:105-135)."""
        return self._prob(1, seq_idx_t0, t, generation_mask)

    def _draw(self, p: torch.Tensor, u: Optional[torch.Tensor]) -> torch.Tensor:
        lib = _hip.lib()
        pd = _hip.dev_f32(p)
        B, K = pd.shape[:2]
        ud = _philox(so3._draw_seed(), B, K, 0, 1)[..., 0].contiguous() if u is None else _hip.dev_f32(u)
        out = torch.empty(B, K, dtype=torch.int64, device=pd.device)
        _hip.check(lib.diffab_categorical_sample(_hip.ptr(pd), _hip.ptr(ud), B * K, V_AA, _hip.ptr(out), _hip.stream_ptr()),
                   "diffab_categorical_sample")
        return out.to(p.device)

    def diffuse_single_step(self, seq_idx, t, generation_mask, *, u=None):
        """One forward step s_{t-1} -> s_t  (diffusion.py:81-103; its stray print is not reproduced)."""
        return self._draw(self.forward_prob_single_step(seq_idx, t, generation_mask), u)

    def diffuse_from_t0(self, seq_idx_t0, t, generation_mask, return_posterior: bool = True, *, u=None):
        """s_t ~ q(s_t | s_0) and, optionally, q(s_{t-1} | s_t, s_0)  (diffusion.py:137-166)."""
        seq_idx_t = self._draw(self.forward_prob_from_t0(seq_idx_t0, t, generation_mask), u)
        if return_posterior:
            return seq_idx_t, self.posterior_single_step(seq_idx_t, seq_idx_t0, t, generation_mask)
        return seq_idx_t

    def posterior_single_step(self, seq_idx_t, seq_idx_t0, t, generation_mask):
        """q(s_{t-1} | s_t, s_0)  (diffusion.py:168-192)."""
        lib = _hip.lib()
        sd = self._sched_dev()
        st, s0, tt, m = _hip.dev_i64(seq_idx_t), _hip.dev_i64(seq_idx_t0), _hip.dev_i64(t), _hip.dev_mask(generation_mask)
        B, K = st.shape
        out = torch.empty(B, K, V_AA, dtype=torch.float32, device=st.device)
        _hip.check(lib.diffab_seq_posterior(C.byref(sd.struct), _hip.ptr(st), _hip.ptr(s0), _hip.ptr(tt), _hip.ptr(m), B, K, _hip.ptr(out),
                                            _hip.stream_ptr()), "diffab_seq_posterior")
        return out.to(seq_idx_t.device)


class CoordinateDiffuser(_Diffuser):
    def __init__(self, T, s=0.01, beta_max=0.999):
        super().__init__(T, s, beta_max)

    def diffuse_from_t0(self, translations_t0, t, generation_mask, return_eps: bool = True, *, eps=None):
        """x_t = sqrt(abar_t) x_0 + sqrt(1 - abar_t) eps on generated residues; eps returned unmasked
        (diffusion.py:199-236)."""
        lib = _hip.lib()
        sd = self._sched_dev()
        x0, tt, m = _hip.dev_f32(translations_t0), _hip.dev_i64(t), _hip.dev_mask(generation_mask)
        B, K = x0.shape[:2]
        e = _philox(so3._draw_seed(), B, K, 1, 0)[..., :3].contiguous() if eps is None else _hip.dev_f32(eps)
        xt = torch.empty_like(x0)
        _hip.check(lib.diffab_coord_forward(C.byref(sd.struct), _hip.ptr(x0), _hip.ptr(tt), _hip.ptr(m), _hip.ptr(e), B, K, _hip.ptr(xt),
                                            _hip.stream_ptr()), "diffab_coord_forward")
        xt = xt.to(translations_t0.device)
        if return_eps:
            return xt, e.to(translations_t0.device)
        return xt


class OrientationDiffuser(_Diffuser):
    def __init__(self, T: int, s: float = 0.01, beta_max: float = 0.999, *, igso3_without_replacement: bool = True):
        super().__init__(T, s, beta_max)
        # IGSO3 table over sigma_t = sqrt(1 - abar_t)  (diffusion.py:254-260).  igso3_without_replacement (default, the reference's
        # behaviour): a patch's K histogram bins drawn as torch.multinomial draws them (so3.py:78; see so3.SO3); False = inverse CDF
        self.so3 = so3.SO3(
            sigmas_to_consider=self.sched["one_minus_alpha_bar_sqrt"],
            cache_prefix=".cache/so3_histograms",
            sigma_threshold=0.1,
            n_bins=8192,
            num_iters=1024,
            without_replacement=igso3_without_replacement,
        )

    def diffuse_from_t0(self, orientations_t0, generation_mask, t, *, rotvec=None):
        """O_t = scale_rot(O_0, sqrt(abar_t)) @ exp(IGSO3 rot-vector, sigma index t) on generated residues
        (diffusion.py:262-294)."""
        lib = _hip.lib()
        sd = self._sched_dev()
        O0, tt, m = _hip.dev_f32(orientations_t0), _hip.dev_i64(t), _hip.dev_mask(generation_mask)
        B, K = O0.shape[:2]
        rv = self.so3.sample_isotropic_gaussian(tt, K) if rotvec is None else rotvec
        rv = _hip.dev_f32(rv)
        Ot = torch.empty_like(O0)
        _hip.check(lib.diffab_orient_forward(C.byref(sd.struct), _hip.ptr(O0), _hip.ptr(m), _hip.ptr(tt), _hip.ptr(rv), B, K, _hip.ptr(Ot),
                                             _hip.stream_ptr()), "diffab_orient_forward")
        return Ot.to(orientations_t0.device)
