"""CPU: the C-ABI library loads and exports every symbol include/diffab_hip.h declares (no compute calls),
plus the host logic of the boundary package: schedule, state_dict layout, seeded-init parity with the
reference's creation order, loud failure without a GPU."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from conftest import REPO
from diffab_pytorch import _hip, synthetic as syn


def header_symbols():
    src = open(os.path.join(REPO, "include", "diffab_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(diffab_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    names = header_symbols()
    assert len(names) >= 25
    lib = ctypes.CDLL(_hip.LIB_PATH)
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/diffab_hip.h but not exported by libdiffab_hip.so"
    assert sorted(_hip.SYMBOLS) == names, "ctypes table and header drifted apart"
    l = _hip.load_library()
    assert b"gfx950" in l.diffab_version()


def test_struct_layouts_match_header():
    assert ctypes.sizeof(_hip.Dims) == 40
    assert ctypes.sizeof(_hip.IpaLayerWeights) == 80
    assert ctypes.sizeof(_hip.Mlp3Weights) == 48
    assert ctypes.sizeof(_hip.DenoiserWeights) == 5 * 8 + 8 + 3 * 48
    assert ctypes.sizeof(_hip.Sched) == 48 and ctypes.sizeof(_hip.Igso3) == 32


def test_schedule_is_bit_identical_to_reference(golden):
    from diffab_pytorch.diffusion import cosine_variance_schedule

    g = golden("schedule")
    for T, s in ((100, 0.01), (200, 0.01), (100, 8e-3)):
        mine = cosine_variance_schedule(T, s=s, beta_max=0.999)
        assert set(mine) == {"alpha", "alpha_bar", "alpha_bar_sqrt", "one_minus_alpha_bar_sqrt", "beta"}
        for k, v in mine.items():
            # bit-identical on the CPU that generated the goldens; torch.cos may differ by an ulp on another CPU model
            np.testing.assert_allclose(v.numpy(), g[f"T{T}_s{s}_{k}"], rtol=4e-7, atol=1e-30, err_msg=f"{T} {s} {k}")


@pytest.mark.skipif(torch.cuda.is_available(), reason="CPU-box behaviour")
def test_compute_fails_loudly_without_gpu():
    from diffab_pytorch import so3

    with pytest.raises(_hip.HipUnavailable):
        so3.log_rotmat(torch.eye(3).expand(2, 2, 3, 3))


def test_denoiser_state_dict_layout_and_seeded_init():
    """Keys/shapes of SURVEY.md Appendix B.3 and the reference's parameter creation order."""
    from diffab_pytorch.diffab_pytorch import Denoiser

    d = syn.BENCH_DIMS
    torch.manual_seed(0)
    den = Denoiser(d["D"], d["C"], d["NL"], d["DS"], d["PQ"], d["PV"], d["H"], 21)
    sd = den.state_dict()
    want = syn.denoiser_state_dict(d, prefix="")
    assert list(sd) == list(want) or set(sd) == set(want)
    for k, v in want.items():
        assert tuple(sd[k].shape) == tuple(v.shape), k
    assert sum(p.numel() for p in den.parameters()) == 1_978_827
    assert len(list(den.buffers())) == 0
    den.load_state_dict(want, strict=True)
    # gamma init = log(e - 1), raw (reference diffab_pytorch.py:373)
    torch.manual_seed(0)
    den2 = Denoiser(d["D"], d["C"], 1, d["DS"], d["PQ"], d["PV"], d["H"], 21)
    assert torch.allclose(den2.ipa.layers[0].gamma, torch.full((8,), float(np.log(np.e - 1.0))))


def test_diffab_constructor_surface(monkeypatch):
    """DiffAb() needs the GPU for its IGSO3 table; on a CPU box check the class surface only."""
    import inspect

    from diffab_pytorch import DiffAb

    sig = inspect.signature(DiffAb.__init__)
    extra = [n for n, q in sig.parameters.items() if q.kind is inspect.Parameter.KEYWORD_ONLY]
    # build-defined keyword; its default is the reference's behaviour (torch.multinomial without replacement, so3.py:78)
    assert extra == ["igso3_without_replacement"] and sig.parameters["igso3_without_replacement"].default is True
    assert [n for n in list(sig.parameters)[1:] if n not in extra] == ["d_residue_emb", "d_pair_emb", "n_ipa_layers", "d_scalar_per_head", "n_query_point_per_head",
                                        "n_value_point_per_head", "n_head", "T", "s", "beta_max", "n_atoms", "aa_vocab_size",
                                        "max_dist_to_consider", "lr", "weight_decay", "betas"]
    for m in ("encode_context", "denoise", "sample", "_add_noise", "_shared_step", "training_step", "validation_step",
              "configure_optimizers"):
        assert callable(getattr(DiffAb, m))
    assert list(inspect.signature(DiffAb.denoise).parameters)[1:] == [
        "seq_idx_t", "translations_t", "orientations_t", "res_context_emb", "pair_context_emb", "beta", "generation_mask", "residue_mask"]
    assert list(inspect.signature(DiffAb.sample).parameters)[1:4] == ["seq_idx", "xyz", "orientations"]


def test_synthetic_patches_are_shard_invariant():
    d = syn.UNIT_DIMS
    full = syn.patches(4, 16, d, seed=3)
    lo = syn.patches(2, 16, d, seed=3, first_patch=0)
    hi = syn.patches(2, 16, d, seed=3, first_patch=2)
    for k in full:
        assert torch.equal(full[k], torch.cat([lo[k], hi[k]])), k
    R = full["orientations"]
    eye = torch.eye(3).expand_as(R)
    assert torch.allclose(R.transpose(-1, -2) @ R, eye, atol=1e-5)
    assert (full["generation_mask"].sum(-1) >= 5).all() and (full["generation_mask"].sum(-1) <= 20).all()

