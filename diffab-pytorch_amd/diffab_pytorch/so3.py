"""SO(3) maps and the IGSO3 sampler of the DiffAb hot path, on HIP (gfx950).

Same importable names and call signatures as the reference's ``diffab_pytorch/so3.py``
(SO3 :9-126, uniform :129, tensor_trace :142, log_rotmat :146, skew_symmetric_mat_to_vector :165,
rotation_matrix_to_vector :173, vector_to_skew_symmetric_mat :185, vector_to_rotation_matrix :207,
exp_skew_symmetric_mat :219, scale_rot :240).  Tensors may arrive on any device; the math runs in
libdiffab_hip.so on the current HIP device and results come back on the caller's device.

Differences from the reference, all documented in DESIGN.md:
  * the IGSO3 table is built on the GPU in milliseconds and never cached on disk
    (``cache_prefix`` is accepted and ignored; the reference's cache key never hit anyway, so3.py:18); it is the reference's
    table (fp32 terms, see SO3.__init__), with the float64 series as the opt-in ``accurate=True``;
  * random draws come from a Philox stream seeded from torch's default generator
    (``torch.manual_seed`` still makes calls reproducible) - or from explicit noise tensors;
  * histogram bins: like the reference's ``torch.multinomial(probs, num_samples)`` (so3.py:78, replacement=False by default) the K
    bins of one patch are drawn WITHOUT replacement - by the exponential race torch itself uses on a GPU
    (``diffab_igso3_bins_without_replacement``; only rows with sigma below the threshold are sorted, the others take the Gaussian
    angle).  ``without_replacement=False`` (constructor or per call) selects the inverse CDF instead, i.e. independent draws per
    residue, which follow the tabulated density also where a row's mass sits in fewer than ~K bins (small sigma: there the
    reference's joint draw pushes most of a patch's angles into the tail) - the build-defined reverse sampler uses that form.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from . import _hip


def _draw_seed() -> int:
    return int(torch.randint(0, 2**62, (1,), dtype=torch.int64).item())


def _back(out: torch.Tensor, like: torch.Tensor) -> torch.Tensor:
    return out.to(like.device)


# ------------------------------------------------------------------ free functions
def tensor_trace(T: torch.Tensor) -> torch.Tensor:
    return T.diagonal(offset=0, dim1=-2, dim2=-1).sum(dim=-1)


def skew_symmetric_mat_to_vector(S: torch.Tensor) -> torch.Tensor:
    return torch.stack([S[..., 2, 1], S[..., 0, 2], S[..., 1, 0]], dim=-1)


def vector_to_skew_symmetric_mat(v: torch.Tensor) -> torch.Tensor:
    z = torch.zeros_like(v[..., 0])
    x, y, w = v[..., 0], v[..., 1], v[..., 2]
    return torch.stack([torch.stack([z, -w, y], -1), torch.stack([w, z, -x], -1), torch.stack([-y, x, z], -1)], -2)


def _mat_op(fn_name: str, M: torch.Tensor, out_last: tuple) -> torch.Tensor:
    lib = _hip.lib()
    inp = _hip.dev_f32(M)
    lead = inp.shape[:-2] if out_last != "from_vec" else inp.shape[:-1]
    n = 1
    for s in lead:
        n *= int(s)
    out = torch.empty(*lead, *(out_last if out_last != "from_vec" else (3, 3)), dtype=torch.float32, device=inp.device)
    _hip.check(getattr(lib, fn_name)(_hip.ptr(inp), _hip.ptr(out), n, _hip.stream_ptr()), fn_name)
    return _back(out, M)


def log_rotmat(R: torch.Tensor) -> torch.Tensor:
    """theta/(2 sin theta) (R - R^T); NaN at theta = 0, like the reference (so3.py:146-162)."""
    return _mat_op("diffab_so3_log", R, (3, 3))


def exp_skew_symmetric_mat(S: torch.Tensor) -> torch.Tensor:
    """Rodrigues; NaN at |v| = 0, like the reference (so3.py:219-237)."""
    return _mat_op("diffab_so3_exp", S, (3, 3))


def rotation_matrix_to_vector(R: torch.Tensor) -> torch.Tensor:
    return _mat_op("diffab_so3_matrix_to_rotvec", R, (3,))


def vector_to_rotation_matrix(v: torch.Tensor) -> torch.Tensor:
    return _mat_op("diffab_so3_rotvec_to_matrix", v, "from_vec")


def scale_rot(R: torch.Tensor, k: torch.Tensor) -> torch.Tensor:
    """exp(k log R), k broadcast from the left (so3.py:240-259)."""
    if k.ndim > R.ndim:
        raise ValueError(f"Dimension of k ({k.ndim}) cannot be larger than that of R ({R.ndim})")
    lib = _hip.lib()
    Rd = _hip.dev_f32(R)
    lead = Rd.shape[:-2]
    if tuple(k.shape) != tuple(lead[: k.ndim]):
        kk = k
        for _ in range(len(lead) - k.ndim):
            kk = kk.unsqueeze(-1)
        kd = _hip.dev_f32(kk.expand(lead))
        per_k = 1
    else:
        kd = _hip.dev_f32(k)
        per_k = 1
        for s in lead[k.ndim:]:
            per_k *= int(s)
    n = 1
    for s in lead:
        n *= int(s)
    out = torch.empty_like(Rd)
    _hip.check(lib.diffab_so3_scale_rot(_hip.ptr(Rd), _hip.ptr(kd), _hip.ptr(out), n, per_k, _hip.stream_ptr()), "diffab_so3_scale_rot")
    return _back(out, R)


def uniform(*size) -> torch.Tensor:
    """Uniform random rotations of shape (*size) with size[-2:] == (3, 3)  (so3.py:129-139)."""
    assert len(size) >= 2, "size must be at least 2-dimensional"
    assert size[-2] == size[-1] == 3, "last two dimensions must be 3"
    lib = _hip.lib()
    n = 1
    for s in size[:-2]:
        n *= int(s)
    dev = _hip.device()
    O = torch.empty(n, 3, 3, dtype=torch.float32, device=dev)
    seq = torch.empty(n, dtype=torch.int64, device=dev)
    x = torch.empty(n, 3, dtype=torch.float32, device=dev)
    m = torch.ones(n, dtype=torch.bool, device=dev)
    chunk = 1 << 20
    for lo in range(0, n, chunk):  # counter field "residue" is 32-bit; patches index the chunks
        hi = min(lo + chunk, n)
        _hip.check(lib.diffab_sample_init(_hip.ptr(seq[lo:hi]), _hip.ptr(x[lo:hi]), _hip.ptr(O[lo:hi]), _hip.ptr(m[lo:hi]),
                                          _draw_seed(), lo // chunk, 1, hi - lo, 1, _hip.stream_ptr()), "diffab_sample_init")
    return O.view(*size).cpu()


# ------------------------------------------------------------------ IGSO3
class SO3:
    """IGSO3 angle table + axis-angle sampler (so3.py:9-126)."""

    def __init__(self, sigmas_to_consider, cache_prefix=".cache/so3_histograms", sigma_threshold=0.1, n_bins=8192, num_iters=1024, *,
                 accurate: bool = False, without_replacement: bool = True):
        """``accurate=False`` (what DiffAb uses): the reference's table - every series term with the reference's own fp32
        roundings, so its rectified rounding noise (~1e-4 of spurious tail mass on the small-sigma rows) is part of the table,
        as it is part of what the reference samples from.  ``accurate=True``: the series in float64, i.e. the exact density."""
        lib = _hip.lib()
        self.accurate = bool(accurate)
        self.without_replacement = bool(without_replacement)  # so3.py:78 (see the module docstring)
        self.n_bins = int(n_bins)
        self.num_iters = int(num_iters)
        self.sigma_threshold = float(sigma_threshold)
        self.sigmas_to_consider = sigmas_to_consider
        self._sigmas = _hip.dev_f32(torch.as_tensor(sigmas_to_consider))
        n = int(self._sigmas.numel())
        self.histograms = torch.empty(n, self.n_bins, dtype=torch.float32, device=self._sigmas.device)
        self._cdf = torch.empty_like(self.histograms)
        build = lib.diffab_igso3_table_build_accurate if self.accurate else lib.diffab_igso3_table_build
        _hip.check(build(_hip.ptr(self._sigmas), n, self.n_bins, self.num_iters, _hip.ptr(self.histograms), _hip.stream_ptr()),
                   "diffab_igso3_table_build")
        _hip.check(lib.diffab_igso3_cdf_build(_hip.ptr(self.histograms), n, self.n_bins, _hip.ptr(self._cdf), _hip.stream_ptr()),
                   "diffab_igso3_cdf_build")

    def struct(self, threshold: Optional[float] = None) -> _hip.Igso3:
        thr = self.sigma_threshold if threshold is None else threshold
        return _hip.Igso3(int(self._sigmas.numel()), self.n_bins, _hip.ptr(self._sigmas), _hip.ptr(self._cdf), C.c_float(thr))

    def _noise(self, n: int, s: int, seed: Optional[int]):
        """(axis_raw, u_bin, u_in, z) from Philox: step 0, patches 0..n-1."""
        lib = _hip.lib()
        seed = _draw_seed() if seed is None else seed
        dev = _hip.device()
        ax = torch.empty(n, s, 4, dtype=torch.float32, device=dev)
        un = torch.empty(n, s, 4, dtype=torch.float32, device=dev)
        _hip.check(lib.diffab_philox_fill(seed, 0, n, s, 0, 2, 0, _hip.ptr(ax), _hip.stream_ptr()), "diffab_philox_fill")
        _hip.check(lib.diffab_philox_fill(seed, 0, n, s, 0, 3, 1, _hip.ptr(un), _hip.stream_ptr()), "diffab_philox_fill")
        nz = torch.empty(n, s, 4, dtype=torch.float32, device=dev)
        _hip.check(lib.diffab_philox_fill(seed, 0, n, s, 0, 3, 0, _hip.ptr(nz), _hip.stream_ptr()), "diffab_philox_fill")
        return ax[..., :3].contiguous(), un[..., 0].contiguous(), un[..., 1].contiguous(), nz[..., 2].contiguous()

    def draw_bins_without_replacement(self, sigma_idx, num_samples, *, race=None, seed=None, threshold=None) -> torch.Tensor:
        """(n, num_samples) int32 histogram bins of rows sigma_idx, each row's draws WITHOUT replacement, in draw order - the joint
        distribution of `torch.multinomial(self.histograms[sigma_idx], num_samples)` (so3.py:78).  `race` (n, n_bins): Exp(1) draws
        (default: -log of Philox uniforms).  ``threshold``: rows with sigma >= threshold are not sorted (their bins come back 0: the
        sampler takes the Gaussian angle there); None sorts every row."""
        lib = _hip.lib()
        idx = _hip.dev_i64(torch.as_tensor(sigma_idx))
        n, s = int(idx.numel()), int(num_samples)
        if race is None:
            seed = _draw_seed() if seed is None else seed
            u = torch.empty(n, (self.n_bins + 3) // 4, 4, dtype=torch.float32, device=_hip.device())
            _hip.check(lib.diffab_philox_fill(seed, 0, n, u.shape[1], 0, 5, 1, _hip.ptr(u), _hip.stream_ptr()), "diffab_philox_fill")  # stream 5: the race
            race = -torch.log(u.view(n, u.shape[1] * 4)[:, :self.n_bins].clamp_min(1e-38)).contiguous()  # (n may be 0: no view(n, -1))
        race = _hip.dev_f32(race)
        bins = torch.empty(n, s, dtype=torch.int32, device=idx.device)
        _hip.check(lib.diffab_igso3_bins_without_replacement(_hip.ptr(self.histograms), int(self._sigmas.numel()), self.n_bins, _hip.ptr(idx), n,
                                                             s, _hip.ptr(race), _hip.ptr(bins),
                                                             _hip.ptr(self._sigmas) if threshold is not None else None,
                                                             C.c_float(float(threshold) if threshold is not None else 0.0), _hip.stream_ptr()),
                   "diffab_igso3_bins_without_replacement")
        return bins

    def _sample(self, sigma_idx, num_samples, threshold, axis_raw=None, u_bin=None, u_in=None, z=None, seed=None, without_replacement=None,
                race=None):
        lib = _hip.lib()
        idx = _hip.dev_i64(torch.as_tensor(sigma_idx))
        n, s = int(idx.numel()), int(num_samples)
        norepl = self.without_replacement if without_replacement is None else bool(without_replacement)
        if norepl and u_bin is None:  # explicit u_bin (the parity tests' captured draws) always means inverse CDF
            if axis_raw is None or u_in is None or z is None:
                a0, _, c0, d0 = self._noise(n, s, seed)
                axis_raw = a0 if axis_raw is None else axis_raw
                u_in = c0 if u_in is None else u_in
                z = d0 if z is None else z
            thr = self.sigma_threshold if threshold is None else threshold
            bins = self.draw_bins_without_replacement(idx, s, race=race, seed=seed, threshold=None if thr == float("inf") else thr)
            axis_raw, u_in, z = (_hip.dev_f32(t) for t in (axis_raw, u_in, z))
            out = torch.empty(n, s, 3, dtype=torch.float32, device=idx.device)
            tab = self.struct(threshold)
            _hip.check(lib.diffab_igso3_sample_bins(C.byref(tab), _hip.ptr(idx), n, s, _hip.ptr(axis_raw), _hip.ptr(bins), _hip.ptr(u_in),
                                                    _hip.ptr(z), _hip.ptr(out), _hip.stream_ptr()), "diffab_igso3_sample_bins")
            return out
        if axis_raw is None or u_bin is None or u_in is None or z is None:
            a0, b0, c0, d0 = self._noise(n, s, seed)
            axis_raw = a0 if axis_raw is None else axis_raw
            u_bin = b0 if u_bin is None else u_bin
            u_in = c0 if u_in is None else u_in
            z = d0 if z is None else z
        axis_raw, u_bin, u_in, z = (_hip.dev_f32(t) for t in (axis_raw, u_bin, u_in, z))
        out = torch.empty(n, s, 3, dtype=torch.float32, device=idx.device)
        tab = self.struct(threshold)
        _hip.check(lib.diffab_igso3_sample(C.byref(tab), _hip.ptr(idx), n, s, _hip.ptr(axis_raw), _hip.ptr(u_bin), _hip.ptr(u_in),
                                           _hip.ptr(z), _hip.ptr(out), _hip.stream_ptr()), "diffab_igso3_sample")
        return out

    def sample_from_histogram(self, sigma_idx, num_samples, *, u_bin=None, u_in=None, without_replacement=None, race=None):
        """Angles (n, num_samples) from the histogram rows (so3.py:74-84)."""
        n = int(torch.as_tensor(sigma_idx).numel())
        ax = torch.tensor([1.0, 0.0, 0.0]).expand(n, int(num_samples), 3)
        r = self._sample(sigma_idx, num_samples, float("inf"), axis_raw=ax, u_bin=u_bin, u_in=u_in, without_replacement=without_replacement,
                         race=race)
        return _back(r[..., 0], torch.as_tensor(sigma_idx))

    def sample_from_gaussian(self, sigma_idx, num_samples, *, z=None):
        """Angles (n, num_samples) = (2 sigma + sigma N(0,1)) mod pi (so3.py:86-96)."""
        n = int(torch.as_tensor(sigma_idx).numel())
        ax = torch.tensor([1.0, 0.0, 0.0]).expand(n, int(num_samples), 3)
        r = self._sample(sigma_idx, num_samples, float("-inf"), axis_raw=ax, z=z)
        return _back(r[..., 0], torch.as_tensor(sigma_idx))

    def sample_isotropic_gaussian(self, sigma_idx: torch.LongTensor, num_samples: int, *, axis_raw=None, u_bin=None, u_in=None, z=None,
                                  seed=None, without_replacement=None, race=None) -> torch.FloatTensor:
        """Rotation vectors (n, num_samples, 3) ~ IGSO3(sigma[sigma_idx])  (so3.py:98-126)."""
        r = self._sample(sigma_idx, num_samples, None, axis_raw, u_bin, u_in, z, seed, without_replacement, race)
        return _back(r, torch.as_tensor(sigma_idx))
