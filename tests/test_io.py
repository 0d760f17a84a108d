"""Output side: host-only helpers."""
import numpy as np
import torch

from diffab_pytorch import io as dio
from diffab_pytorch import synthetic as syn


def _frames(n, seed=0):
    rng = np.random.Generator(np.random.PCG64(seed))
    return torch.from_numpy(10 * rng.standard_normal((n, 3))).float(), torch.from_numpy(syn.random_rotations(rng, n)).float()


def test_backbone_round_trip_and_geometry():
    x, O = _frames(32)
    bb = dio.backbone_from_frames(x, O)
    assert bb.shape == (32, 5, 3)
    n, ca, c = bb[:, 0], bb[:, 1], bb[:, 2]
    assert torch.allclose(ca, x)
    # ideal bond lengths / angle survive any rigid frame
    assert torch.allclose((n - ca).norm(dim=-1), torch.full((32,), 1.4607), atol=2e-4)
    assert torch.allclose((c - ca).norm(dim=-1), torch.full((32,), 1.526), atol=2e-4)
    cosang = ((n - ca) * (c - ca)).sum(-1) / ((n - ca).norm(dim=-1) * (c - ca).norm(dim=-1))
    assert torch.allclose(torch.rad2deg(torch.acos(cosang)), torch.full((32,), 111.07), atol=0.05)
    t2, R2 = dio.frames_from_backbone(n, ca, c)
    assert torch.allclose(t2, x) and torch.allclose(R2, O, atol=2e-6)
    # same convention as the hot path's euclidean_transform (global = local @ R + t): batched leading dims work too
    bb2 = dio.backbone_from_frames(x.view(4, 8, 3), O.view(4, 8, 3, 3))
    assert torch.equal(bb2.view(32, 5, 3), bb)


def test_pdb_writer_and_sample_file(tmp_path):
    x, O = _frames(6, seed=3)
    seq = torch.tensor([0, 7, 19, 20, 3, 7])  # ALA GLY VAL UNK ASP GLY
    n = dio.write_pdb(str(tmp_path / "p.pdb"), seq, x, O, chain_idx=torch.tensor([1, 1, 1, 2, 2, 2]), residue_idx=torch.arange(6),
                      residue_mask=torch.tensor([1, 1, 1, 1, 0, 1], dtype=torch.bool), atoms=("N", "CA", "C", "O", "CB"))
    lines = open(tmp_path / "p.pdb").read().splitlines()
    atoms = [l for l in lines if l.startswith("ATOM")]
    assert n == len(atoms) == 3 * 5 + 2 * 4  # five residues kept, the two glycines without CB
    assert lines[-1] == "END" and all(len(l) == 78 for l in atoms)
    ca = [l for l in atoms if l[12:16].strip() == "CA"]
    assert [l[17:20] for l in ca] == ["ALA", "GLY", "VAL", "UNK", "GLY"] and [l[21] for l in ca] == ["A", "A", "A", "B", "B"]
    got = torch.tensor([[float(l[30:38]), float(l[38:46]), float(l[46:54])] for l in ca])
    assert torch.allclose(got, x[[0, 1, 2, 3, 5]], atol=5e-4)
    s = {"seq_idx": seq, "translations": x, "orientations": O}
    dio.save_samples(str(tmp_path / "s.pt"), s, seed=5, t_stop=0)
    s2, meta = dio.load_samples(str(tmp_path / "s.pt"))
    assert all(torch.equal(s[k], s2[k]) for k in s) and meta == {"seed": 5, "t_stop": 0}
