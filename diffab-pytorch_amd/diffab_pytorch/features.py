"""Featurisation from coordinates on the device: the per-patch structural features the reference's data
layer computes with ``protstruc`` before a batch reaches the model (data.py:75-82, preprocess_pdb.py:60-65) - backbone
orientations, backbone dihedrals + mask and the pairwise phi / psi dihedrals - from the all-atom coordinates ``xyz`` alone, so a
patch needs no feature files besides ``xyz`` (the atom-atom distances are already taken from ``xyz`` inside the pair kernel).

``protstruc`` is not in the reference tree, so these follow the geometric definitions (``include/diffab_hip.h``:
``diffab_featurize_xyz``); parity with protstruc's conventions (frame axes, dihedral sign, masking at chain breaks) is UNPINNED.
The oracle restates the same definitions in float64 (``oracle/diffab_oracle.py::featurize_xyz``).
"""
from __future__ import annotations

from typing import Dict, Optional

import torch

from . import _hip


def featurize(xyz: torch.Tensor, chain_idx: Optional[torch.Tensor] = None, residue_mask: Optional[torch.Tensor] = None, *,
              orientations: bool = True, backbone_dihedrals: bool = True, pairwise_dihedrals: bool = True) -> Dict[str, torch.Tensor]:
    """xyz (B,K,A,3) with atoms N, CA, C in slots 0, 1, 2 -> the requested keys of the reference batch dict:
    ``orientations`` (B,K,3,3), ``backbone_dihedrals`` (B,K,3) + ``backbone_dihedrals_mask`` (B,K,3) bool, ``pairwise_dihedrals``
    (B,K,K,2).  One C-ABI call; results on xyz's device."""
    lib = _hip.lib()
    x = _hip.dev_f32(xyz)
    if x.dim() != 4 or x.shape[-1] != 3 or x.shape[2] < 3:
        raise ValueError("featurize expects xyz (B, K, A >= 3, 3)")
    B, K, A = x.shape[:3]
    ch = None if chain_idx is None else _hip.dev_i64(chain_idx)
    rm = None if residue_mask is None else _hip.dev_mask(residue_mask)
    dev = x.device
    O = torch.empty(B, K, 3, 3, dtype=torch.float32, device=dev) if orientations else None
    dh = torch.empty(B, K, 3, dtype=torch.float32, device=dev) if backbone_dihedrals else None
    dm = torch.empty(B, K, 3, dtype=torch.bool, device=dev) if backbone_dihedrals else None
    pd = torch.empty(B, K, K, 2, dtype=torch.float32, device=dev) if pairwise_dihedrals else None
    _hip.check(lib.diffab_featurize_xyz(_hip.ptr(x), _hip.ptr(ch), _hip.ptr(rm), B, K, A, _hip.ptr(O), _hip.ptr(dh), _hip.ptr(dm), _hip.ptr(pd),
                                        _hip.stream_ptr()), "diffab_featurize_xyz")
    out = {}
    if orientations:
        out["orientations"] = O.to(xyz.device)
    if backbone_dihedrals:
        out["backbone_dihedrals"] = dh.to(xyz.device)
        out["backbone_dihedrals_mask"] = dm.to(xyz.device)
    if pairwise_dihedrals:
        out["pairwise_dihedrals"] = pd.to(xyz.device)
    return out
