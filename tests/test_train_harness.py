"""Trainer harness: host logic on CPU; one short synthetic run on the GPU."""
import json
import os
import sys

import numpy as np
import pytest
import torch

from diffab_pytorch import synthetic as syn
from diffab_pytorch import train as T


def test_flags_mirror_reference_parse_argument():
    # reference train.py:14-43: option strings, types and defaults
    a = T.parse_argument(["--synthetic", "4"])
    assert (a.val_pct, a.bsz, a.epochs, a.learning_rate, a.seed, a.no_wandb) == (0.1, 128, 60, 0.01, 42, False)
    a = T.parse_argument(["--meta", "m.csv", "--data-dir", "d", "--val-pct", "0.2", "--cdrs", "H1", "H3", "-b", "8", "-e", "2", "-l", "1e-3",
                          "-s", "7", "--no-wandb"])
    assert (a.meta, a.data_dir, a.val_pct, a.cdrs, a.bsz, a.epochs, a.learning_rate, a.seed, a.no_wandb) == \
        ("m.csv", "d", 0.2, ["H1", "H3"], 8, 2, 1e-3, 7, True)
    assert T.parse_argument(["--gradient-clip-val", "1.0"]).gradient_clip_val == 1.0  # what train.py:102 reads
    assert list(T.MODEL_HPARAMS.values()) == [128, 64, 6, 32, 8, 8, 8]  # train.py:62-80


def test_reference_data_route_is_refused_loudly():
    a = T.parse_argument(["--meta", "m.csv", "--data-dir", "d"])
    with pytest.raises(SystemExit, match="protstruc"):
        T.PatchSource(a, 0, 1, "train")


def test_pairwise_atom_distances_matches_loop():
    g = torch.Generator().manual_seed(0)
    xyz = torch.randn(2, 3, 4, 3, generator=g)
    d = T.pairwise_atom_distances(xyz)
    assert d.shape == (2, 3, 3, 4, 4)
    for b, i, j, p, q in [(0, 0, 0, 0, 0), (1, 2, 0, 3, 1), (0, 1, 2, 2, 2)]:
        assert torch.allclose(d[b, i, j, p, q], (xyz[b, i, p] - xyz[b, j, q]).norm(), atol=1e-6)
    assert torch.equal(d, d.permute(0, 2, 1, 4, 3))  # symmetric under (i,a) <-> (j,a')


def test_synthetic_source_shards_partition_every_batch():
    a = T.parse_argument(["--synthetic", "22", "-b", "8", "--k", "8", "--val-pct", "0.25"])
    whole = T.PatchSource(a, 0, 1, "train")
    assert whole.n == 17 and len(whole) == 3 and T.PatchSource(a, 0, 1, "val").n == 5
    full = [b["seq_idx"] for b in whole.batches(epoch=3)]
    parts = [[b["seq_idx"] for b in T.PatchSource(a, r, 3, "train").batches(epoch=3)] for r in range(3)]
    assert [len(p) for p in parts] == [2, 2, 2]  # the 1-patch tail cannot feed 3 ranks: dropped on every rank alike
    for k, fb in enumerate(full[:2]):  # rank shards, concatenated in rank order, are the single-process batch
        assert torch.equal(torch.cat([parts[r][k] for r in range(3)]), fb)
    e4 = [b["seq_idx"] for b in whole.batches(epoch=4)]
    assert not all(torch.equal(x, y) for x, y in zip(full, e4))  # reshuffled per epoch


def test_patch_files_round_trip_and_distmat(tmp_path):
    b = syn.context_batch(3, 6, seed=5)
    for p in range(3):
        one = {k: (b[k][p:p + 1] if k != "residue_idx" else b[k]) for k in T.PATCH_KEYS if k != "backbone_dihedrals_mask"}
        one["backbone_dihedrals_mask"] = torch.ones(1, 6, 3, dtype=torch.bool)
        one["generation_mask"] = b["generation_mask"][p:p + 1]
        torch.save(one, tmp_path / f"patch{p}.pt")
    a = T.parse_argument(["--patch-dir", str(tmp_path), "-b", "4", "--val-pct", "0.0"])
    src = T.PatchSource(a, 0, 1, "train")
    assert src.n == 3
    (batch,) = list(src.batches(0))
    assert batch["xyz"].shape == (3, 6, 15, 3) and batch["residue_idx"].shape == (1, 6)
    # distmat is not stored (preprocess_pdb.py leaves it out) and not rebuilt on the host: the pair kernel takes it from xyz.
    # pairwise_atom_distances states what that kernel computes; same numbers as the generator's float64 distances.
    assert "distmat" not in batch
    order = [int(os.path.basename(f)[5]) for f in src.files]
    epoch_order = torch.randperm(3, generator=torch.Generator().manual_seed(a.seed + 0)).tolist()
    idx = [order[i] for i in epoch_order]
    np.testing.assert_allclose(T.pairwise_atom_distances(batch["xyz"]).numpy(), b["distmat"][idx].numpy(), rtol=0, atol=2e-5)
    with pytest.raises(KeyError, match="preprocess_pdb"):
        torch.save({"xyz": b["xyz"][:1]}, tmp_path / "bad.pt")
        T.load_patch(str(tmp_path / "bad.pt"))
    # a patch exactly as the reference writes it has no CDR mask: refused loudly (an all-false mask would give 0/0 losses and
    # zero gradients), as is a mask that selects nothing; --cdr-mask-key names another key
    ref_fmt = torch.load(tmp_path / "patch0.pt")
    mask = ref_fmt.pop("generation_mask")
    torch.save(ref_fmt, tmp_path / "nomask.pt")
    with pytest.raises(KeyError, match="generation_mask"):
        T.load_patch(str(tmp_path / "nomask.pt"))
    torch.save(dict(ref_fmt, h3=mask), tmp_path / "named.pt")
    assert torch.equal(T.load_patch(str(tmp_path / "named.pt"), "h3")["generation_mask"], mask.bool())
    torch.save(dict(ref_fmt, generation_mask=torch.zeros_like(mask)), tmp_path / "empty.pt")
    with pytest.raises(ValueError, match="no valid residue"):
        T.load_patch(str(tmp_path / "empty.pt"))


def test_checkpoint_layout_round_trip(tmp_path):
    m = torch.nn.Linear(3, 2)
    opt = torch.optim.Adam(m.parameters(), lr=1e-3)
    m(torch.ones(1, 3)).sum().backward()
    opt.step()
    T.save_checkpoint(str(tmp_path / "c" / "last.ckpt"), m, opt, epoch=2, global_step=17)
    ck = torch.load(tmp_path / "c" / "last.ckpt")
    assert {"state_dict", "optimizer", "epoch", "global_step"} <= set(ck)  # Lightning layout: weights under "state_dict"
    m2 = torch.nn.Linear(3, 2)
    opt2 = torch.optim.Adam(m2.parameters(), lr=1e-3)
    assert T.load_checkpoint(str(tmp_path / "c" / "last.ckpt"), m2, opt2) == {"epoch": 2, "global_step": 17}
    assert all(torch.equal(a, b) for a, b in zip(m.state_dict().values(), m2.state_dict().values()))
    assert opt2.state_dict()["state"][0]["step"] == opt.state_dict()["state"][0]["step"]


@pytest.mark.gpu
def test_two_synthetic_steps_train_and_resume(tmp_path):
    log = tmp_path / "log.jsonl"
    rc = T.main(["--synthetic", "6", "--k", "16", "-b", "3", "-e", "1", "--val-pct", "0.34", "--gradient-clip-val", "1.0",
                 "--ckpt-dir", str(tmp_path / "ck"), "--log", str(log)])
    assert rc == 0
    recs = [json.loads(l) for l in open(log)]
    train = [r for r in recs if "train/loss" in r]
    val = [r for r in recs if "val/loss" in r]
    assert len(train) == 2 and len(val) == 1  # 4 training patches in batches of 3 -> 2 steps; 2 validation patches
    for r in train:  # the reference's logging keys, diffab_pytorch.py:889-902
        assert {"train/seq_loss", "train/translations_loss", "train/orientations_loss", "train/loss"} <= set(r)
        assert np.isfinite(r["train/loss"])
    ck = torch.load(tmp_path / "ck" / "last.ckpt")
    assert ck["global_step"] == 2 and any(k.startswith("denoiser.ipa.layers.0.") for k in ck["state_dict"])
    # the context encoders are trained too (encode_context has a HIP backward): their weights differ from the seeded initialisation
    from diffab_pytorch import DiffAb

    torch.manual_seed(42)  # the harness's default --seed
    fresh = DiffAb(*T.MODEL_HPARAMS.values()).state_dict()
    for k in ("residue_context_embedding.mlp.0.weight", "residue_context_embedding.amino_acid_type_embedding.weight",
              "pair_context_embedding.mlp.4.weight", "pair_context_embedding.pair2distcoef.weight", "denoiser.ipa.layers.5.to_out.weight"):
        assert not torch.equal(ck["state_dict"][k].cpu(), fresh[k].cpu()), k
    # resume: one more epoch continues the step count from the checkpoint
    rc = T.main(["--synthetic", "6", "--k", "16", "-b", "3", "-e", "2", "--val-pct", "0.34", "--resume", str(tmp_path / "ck" / "last.ckpt"),
                 "--log", str(log)])
    assert rc == 0
    assert max(json.loads(l)["step"] for l in open(log)) == 4
