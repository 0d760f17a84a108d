"""Seeded synthetic K-residue patches and denoiser weights.

The reference ships no structures (only SAbDab ids) and its own tests feed
unseeded random tensors (reference tests/test_modules.py:203-248).  This module
is the seeded equivalent used by the parity tests, the golden-vector generator
and ``bench.py``: everything is generated with numpy's PCG64 on the host, so the
same (seed, shape) gives the same tensors in the build container and on the GPU
box.  It depends on numpy/torch only and imports nothing else from the package
(the golden generator loads it by file path next to the real reference).

Shapes follow SURVEY.md section 8(d) / reference diffab_pytorch.py:558-568.
"""
from __future__ import annotations

from typing import Dict

import numpy as np
import torch

BENCH_DIMS = dict(D=128, C=64, NL=6, DS=32, H=8, PQ=8, PV=8, V=21)  # reference train.py:62-70
UNIT_DIMS = dict(D=32, C=16, NL=4, DS=12, H=8, PQ=4, PV=4, V=21)  # reference tests/test_modules.py:203-211


def _rng(seed: int, *tags: int) -> np.random.Generator:
    return np.random.Generator(np.random.PCG64([int(seed), *[int(t) for t in tags]]))


def random_rotations(rng: np.random.Generator, n: int) -> np.ndarray:
    """Uniform SO(3) from normalised 4-normals (quaternion w,x,y,z)."""
    q = rng.standard_normal((n, 4))
    q /= np.linalg.norm(q, axis=-1, keepdims=True)
    w, x, y, z = q.T
    R = np.stack(
        [
            1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w),
            2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w),
            2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y),
        ],
        axis=-1,
    ).reshape(n, 3, 3)
    return R


def denoiser_state_dict(dims: Dict[str, int], seed: int = 0, prefix: str = "denoiser.") -> Dict[str, torch.Tensor]:
    """Synthetic parameters with the reference's state_dict keys and shapes
    (reference diffab_pytorch.py:501-556, 339-387).
    Weights ~ N(0, 1/fan_in), biases ~ N(0, 0.1^2), gamma = log(e-1) + N(0, 0.1^2)
    (per-head distinct so a head-indexing slip cannot hide)."""
    D, C, NL, DS, H, PQ, PV, V = (dims[k] for k in ("D", "C", "NL", "DS", "H", "PQ", "PV", "V"))
    rng = _rng(seed, 1001)
    sd: Dict[str, np.ndarray] = {}

    def lin(name, out_f, in_f, bias=True):
        sd[name + ".weight"] = rng.standard_normal((out_f, in_f)) / np.sqrt(in_f)
        if bias:
            sd[name + ".bias"] = 0.1 * rng.standard_normal(out_f)

    sd["sequence_embedding.weight"] = rng.standard_normal((25, D))
    lin("to_res_emb.0", D, 2 * D)
    lin("to_res_emb.2", D, D)
    for l in range(NL):
        p = f"ipa.layers.{l}."
        sd[p + "gamma"] = np.log(np.e - 1.0) + 0.1 * rng.standard_normal(H)
        lin(p + "to_q_scalar", H * DS, D, bias=False)
        lin(p + "to_k_scalar", H * DS, D, bias=False)
        lin(p + "to_v_scalar", H * DS, D, bias=False)
        lin(p + "to_pair_bias", H, C, bias=False)
        lin(p + "to_q_point", H * PQ * 3, D, bias=False)
        lin(p + "to_k_point", H * PQ * 3, D, bias=False)
        lin(p + "to_v_point", H * PV * 3, D, bias=False)
        lin(p + "to_out", D, H * DS + H * C + H * PV * 3 + H * PV)
    for head, nout in (("coordinate_denoising", 3), ("orientation_denoising", 3), ("sequence_denoising", V)):
        lin(head + ".0", D, D + 3)
        lin(head + ".2", D, D)
        lin(head + ".4", nout, D)
    return {prefix + k: torch.from_numpy(v.astype(np.float32)) for k, v in sd.items()}


def patches(B: int, K: int, dims: Dict[str, int], seed: int = 0, coord_sigma: float = 10.0,
            first_patch: int = 0) -> Dict[str, torch.Tensor]:
    """B synthetic patches.  Patch p's tensors depend only on (seed, first_patch + p),
    so any sharding of a batch over ranks reproduces the unsharded tensors."""
    D, C = dims["D"], dims["C"]
    out = {k: [] for k in ("res_context_emb", "pair_context_emb", "translations", "orientations",
                           "seq_idx", "generation_mask", "residue_mask")}
    for p in range(first_patch, first_patch + B):
        rng = _rng(seed, 2002, p)
        out["res_context_emb"].append(rng.standard_normal((K, D), dtype=np.float32))
        out["pair_context_emb"].append(rng.standard_normal((K, K, C), dtype=np.float32))
        x = coord_sigma * rng.standard_normal((K, 3))
        out["translations"].append((x - x.mean(0, keepdims=True)).astype(np.float32))
        out["orientations"].append(random_rotations(rng, K).astype(np.float32))
        out["seq_idx"].append(rng.integers(0, 20, size=K, dtype=np.int64))
        seg = int(rng.integers(5, 21))
        seg = min(seg, K)
        start = int(rng.integers(0, K - seg + 1))
        m = np.zeros(K, dtype=bool)
        m[start:start + seg] = True
        out["generation_mask"].append(m)
        out["residue_mask"].append(np.ones(K, dtype=bool))
    return {k: torch.from_numpy(np.stack(v)) for k, v in out.items()}
