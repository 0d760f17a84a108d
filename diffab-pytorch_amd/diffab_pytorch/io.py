"""Output side of the path: backbone atoms from the sampled frames, a PDB writer and a
round-trippable sample file.  The reference stops at frames (x, O) - it has no reconstruction or writer - so this is new,
build-defined functionality (nothing here is on the timed path); the atom reconstruction of frames that live on the device runs on
the HIP frame kernel, file writing is host code.

Frame convention (the one the hot path uses everywhere: reference ``euclidean_transform`` diffab_pytorch.py:315-324,
``global = local @ R + t`` with row vectors): a residue's frame has its origin at CA, and in LOCAL coordinates
C lies on +x and N in the xy-plane with positive y - the Gram-Schmidt frame of AlphaFold-style backbones.
``frames_from_backbone`` and ``backbone_from_frames`` are exact inverses on ideal backbones.  Whether this is the convention
of ``protstruc.StructureBatch.backbone_orientations`` (which produces the reference's orientations, data.py:82) cannot be
checked here (protstruc is not in the tree): parity with it is UNPINNED.
"""
from __future__ import annotations

from typing import Dict, Optional, Sequence

import torch

# ideal local coordinates (Angstrom) of the backbone atoms in the residue frame above (AlphaFold 2 supplementary table 2 geometry)
IDEAL_BACKBONE = {
    "N": (-0.525, 1.363, 0.000),
    "CA": (0.000, 0.000, 0.000),
    "C": (1.526, 0.000, 0.000),
    "O": (2.153, -1.062, 0.000),   # carbonyl oxygen for an extended chain (psi-independent placement in the N-CA-C plane)
    "CB": (-0.529, -0.774, -1.205),
}
BACKBONE_ATOMS = ("N", "CA", "C", "O", "CB")
AA3 = ("ALA", "ARG", "ASN", "ASP", "CYS", "GLN", "GLU", "GLY", "HIS", "ILE", "LEU", "LYS", "MET", "PHE", "PRO", "SER", "THR", "TRP",
       "TYR", "VAL", "UNK")  # index order of the 20 + UNK vocabulary (reference AA.UNK = 20)


def backbone_from_frames(translations: torch.Tensor, orientations: torch.Tensor, atoms: Sequence[str] = BACKBONE_ATOMS) -> torch.Tensor:
    """(…,3) CA positions and (…,3,3) orientations -> (…, len(atoms), 3) atom coordinates: local @ R + t.
    Frames on the device (the sampler's output) go through the HIP frame kernel - the same `diffab_frames_apply` that implements
    the reference's euclidean_transform (diffab_pytorch.py:315-324), with the ideal backbone as the local points; host tensors
    (files being written) use the identical expression in torch."""
    local = torch.tensor([IDEAL_BACKBONE[a] for a in atoms], dtype=torch.float32)  # (A,3)
    if translations.is_cuda:
        from . import _hip

        lib = _hip.lib()
        t = _hip.dev_f32(translations).reshape(-1, 3)
        R = _hip.dev_f32(orientations).reshape(-1, 3, 3)
        L, A = t.shape[0], len(atoms)
        pts = local.to(t.device).expand(L, A, 3).contiguous()  # (B=1, N=1, L, A, 3)
        out = torch.empty_like(pts)
        _hip.check(lib.diffab_frames_apply(_hip.ptr(pts), _hip.ptr(R), _hip.ptr(t), _hip.ptr(out), 1, 1, L, A, _hip.stream_ptr()),
                   "diffab_frames_apply")
        return out.view(*translations.shape[:-1], A, 3).to(translations.dtype)
    local = local.to(dtype=translations.dtype)
    return torch.einsum("ak,...kc->...ac", local, orientations) + translations.unsqueeze(-2)


def frames_from_backbone(n: torch.Tensor, ca: torch.Tensor, c: torch.Tensor):
    """Inverse of ``backbone_from_frames`` on (N, CA, C): Gram-Schmidt, rows of R are the local axes in global coordinates."""
    e1 = torch.nn.functional.normalize(c - ca, dim=-1)
    u2 = (n - ca) - (e1 * (n - ca)).sum(-1, keepdim=True) * e1
    e2 = torch.nn.functional.normalize(u2, dim=-1)
    e3 = torch.cross(e1, e2, dim=-1)
    return ca, torch.stack([e1, e2, e3], dim=-2)


def write_pdb(path: str, seq_idx: torch.Tensor, translations: torch.Tensor, orientations: torch.Tensor,
              chain_idx: Optional[torch.Tensor] = None, residue_idx: Optional[torch.Tensor] = None, residue_mask: Optional[torch.Tensor] = None,
              b_factor: Optional[torch.Tensor] = None, atoms: Sequence[str] = ("N", "CA", "C", "O")) -> int:
    """One patch (K residues) as ATOM records (glycine gets no CB).  chain ids 1, 2, 3 ... -> A, B, C ...; returns the atom count."""
    seq_idx, translations, orientations = seq_idx.cpu(), translations.cpu().float(), orientations.cpu().float()
    K = seq_idx.shape[0]
    xyz = backbone_from_frames(translations, orientations, atoms)
    lines, serial = [], 1
    for i in range(K):
        if residue_mask is not None and not bool(residue_mask[i]):
            continue
        aa = AA3[int(seq_idx[i])] if 0 <= int(seq_idx[i]) < len(AA3) else "UNK"
        ch = "A" if chain_idx is None else chr(ord("A") + max(int(chain_idx[i]) - 1, 0) % 26)
        rn = i + 1 if residue_idx is None else int(residue_idx[i]) + 1
        bf = 0.0 if b_factor is None else float(b_factor[i])
        for a, name in enumerate(atoms):
            if name == "CB" and aa == "GLY":
                continue
            x, y, z = (float(v) for v in xyz[i, a])
            lines.append(f"ATOM  {serial:5d} {name:<4s} {aa:>3s} {ch}{rn:4d}    {x:8.3f}{y:8.3f}{z:8.3f}{1.0:6.2f}{bf:6.2f}          {name[0]:>2s}")
            serial += 1
    lines.append("END")
    with open(path, "w") as f:
        f.write("\n".join(lines) + "\n")
    return serial - 1


def save_samples(path: str, samples: Dict[str, torch.Tensor], **meta) -> None:
    """``DiffAb.sample`` output (seq_idx, translations, orientations, ...) + free-form metadata, bit-exact round trip."""
    torch.save({"samples": {k: v.detach().cpu() for k, v in samples.items()}, "meta": meta}, path)


def load_samples(path: str):
    d = torch.load(path, map_location="cpu")
    return d["samples"], d.get("meta", {})
