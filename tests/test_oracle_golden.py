"""CPU: the oracle restatement against every golden vector generated from the real reference
(oracle/gen_golden.py).  Runs wherever the tests run - the reference itself never travels."""
import numpy as np
import pytest
import torch

import diffab_oracle as orc
from conftest import maxrel
from diffab_pytorch import synthetic as syn

T = torch.from_numpy


def test_schedule_bit_exact(golden):
    g = golden("schedule")
    for Tn, s in ((100, 0.01), (200, 0.01), (100, 8e-3)):
        mine = orc.cosine_variance_schedule(Tn, s=s, beta_max=0.999)
        for k, v in mine.items():
            # bit-identical on the CPU that generated the goldens; torch.cos may differ by an ulp on another CPU model
            np.testing.assert_allclose(v.numpy(), g[f"T{Tn}_s{s}_{k}"], rtol=4e-7, atol=1e-30, err_msg=f"{Tn} {s} {k}")


def test_schedule_known_answers():
    # SURVEY.md Appendix B.4 (measured from the reference)
    s = orc.cosine_variance_schedule(100, s=0.01, beta_max=0.999)
    np.testing.assert_allclose(s["beta"][[1, 2, 50, 99, 100]].numpy(),
                               [7.25448132e-4, 1.20937824e-3, 3.06271315e-2, 0.749942124, 0.999000013], rtol=1e-6)
    np.testing.assert_allclose(s["alpha_bar"][[1, 2, 50, 99, 100]].numpy(),
                               [0.999274552, 0.998066068, 0.492343277, 2.41914851e-4, 1.91114773e-15], rtol=1e-5)
    assert abs(float(orc.cosine_variance_schedule(100)["beta"][1]) - 6.31272793e-4) < 1e-9


def test_so3(golden):
    g = golden("so3")
    R, k, v = T(g["R"]), T(g["k"]), T(g["v"])
    assert maxrel(orc.log_so3(R), g["log"]) < 1e-6
    assert maxrel(orc.matrix_to_rotvec(R), g["rotvec"]) < 1e-6
    assert maxrel(orc.exp_so3(orc.log_so3(R)), g["explog"]) < 1e-6
    assert maxrel(orc.scale_rot(R, k), g["scaled"]) < 1e-6
    assert maxrel(orc.rotvec_to_matrix(v), g["expv"]) < 1e-6
    assert np.array_equal(orc.hat(v).numpy(), g["hat"])
    with pytest.raises(ValueError):
        orc.scale_rot(R[0, 0], torch.ones(2, 2, 2, 2))


def test_sequence_diffuser(golden):
    g = golden("seqdiff")
    s = orc.cosine_variance_schedule(100, s=0.01, beta_max=0.999)
    seq0, seqt, t, m = T(g["seq0"]), T(g["seqt"]), T(g["t"]), T(g["mask"])
    assert maxrel(orc.seq_forward_prob_single_step(seqt, t, m, s), g["single"]) < 1e-7
    assert maxrel(orc.seq_forward_prob_from_t0(seq0, t, m, s), g["from_t0"]) < 1e-7
    post = orc.seq_posterior_single_step(seqt, seq0, t, m, s)
    assert maxrel(post, g["posterior"]) < 1e-6
    assert torch.allclose(post.sum(-1), torch.ones_like(post[..., 0]), atol=1e-6)


def test_weighted_multinomial(golden):
    g = golden("seqdiff")
    p1 = torch.nn.functional.one_hot(T(g["seq0"]), 21)
    assert np.array_equal(orc.weighted_multinomial(p1, T(g["wm_p2"]), T(g["wm_w1"]), T(g["wm_w2"])).numpy(), g["wm_out"])


def test_coordinate_diffuser(golden):
    g = golden("coorddiff")
    s = orc.cosine_variance_schedule(100, s=0.01, beta_max=0.999)
    xt = orc.coord_diffuse_from_t0(T(g["x0"]), T(g["t"]), T(g["mask"]), T(g["eps"]), s)
    assert maxrel(xt, g["xt"]) < 1e-7


def test_igso3_table_and_sampler(golden):
    g = golden("igso3")
    sig = T(g["sigmas"])
    rows = g["rows"].tolist()
    tab = orc.igso3_table(sig[rows], 8192, 1024)
    assert maxrel(tab[1:, ::16], g["probe_every16"][1:]) < 1e-6  # row 0 (sigma = 0) is finite garbage upstream
    assert (np.asarray(tab.argmax(-1))[1:] == g["row_argmax"][rows][1:]).all()
    # captured reference draws -> same rotation vectors
    tt = T(g["samp_t"])
    Ks = g["samp_bin"].shape[1]
    th_h = orc.igso3_theta_from_hist(T(g["samp_bin"]), T(g["samp_u"]))
    th_g = orc.igso3_theta_from_gaussian(sig[tt][:, None].expand(-1, Ks), T(g["samp_z"]))
    rv = orc.igso3_rotvec(T(g["samp_axis_raw"]), th_h, th_g, sig[tt])
    assert maxrel(rv, g["samp_rotvec"]) < 1e-6
    s = orc.cosine_variance_schedule(100, s=0.01, beta_max=0.999)
    Ot = orc.orient_diffuse_from_t0(T(g["od_O0"]), T(g["od_mask"]), tt, T(g["samp_rotvec"]), s)
    assert maxrel(Ot, g["od_Ot"]) < 2e-6


CASES = ["unit_wide", "unit_tight", "unit_ragged", "bench_wide", "bench_tight", "bench_k256"]


def case_inputs(g):
    B, K, seed, D, C, NL, DS, H, PQ, PV = [int(v) for v in g["meta"]]
    dims = dict(D=D, C=C, NL=NL, DS=DS, H=H, PQ=PQ, PV=PV, V=21)
    sd = syn.denoiser_state_dict(dims, seed=seed)
    inp = syn.patches(B, K, dims, seed=seed, coord_sigma=float(g["coord_sigma"]))
    chk = np.array([float(inp[k].double().sum()) for k in ("res_context_emb", "pair_context_emb", "translations", "orientations")])
    assert np.allclose(chk, g["input_checksum"], rtol=1e-9, atol=1e-6), "synthetic inputs drifted from the fixture"
    assert np.isclose(float(sum(v.double().sum() for v in sd.values())), float(g["weight_checksum"]), rtol=1e-9)
    return dims, sd, inp


@pytest.mark.parametrize("name", CASES)
def test_denoiser_cases(golden, name):
    g = golden("denoiser_" + name)
    dims, sd, inp = case_inputs(g)
    out = orc.denoiser(sd, inp["seq_idx"], inp["translations"], inp["orientations"], inp["res_context_emb"], inp["pair_context_emb"],
                       T(g["beta"]), dims["NL"], dims["H"])
    for k in ("translations_eps", "orientations_t0", "seq_posterior", "aa_logits", "res_emb"):
        assert maxrel(out[k], g[k]) < 2e-5, k
    l0 = orc.ipa_layer(inp["res_context_emb"], inp["pair_context_emb"], inp["orientations"], inp["translations"], sd,
                       "denoiser.ipa.layers.0.", dims["H"])
    assert maxrel(l0, g["ipa_layer0"]) < 2e-5


def test_losses(golden):
    g = golden("losses_grads")
    den = {"seq_posterior": T(g["out_post"]), "translations_eps": T(g["out_eps"]), "orientations_t0": T(g["out_O0"])}
    B, K, seed = [int(v) for v in g["meta"][:3]]
    dims = dict(zip(("D", "C", "NL", "DS", "H", "PQ", "PV"), [int(v) for v in g["meta"][3:]]), V=21)
    inp = syn.patches(B, K, dims, seed=seed, coord_sigma=4.0)
    ls = orc.hotpath_losses(den, T(g["post"]), T(g["eps"]), inp["orientations"], T(g["gen"]), T(g["resm"]))
    np.testing.assert_allclose([float(x) for x in ls], g["losses"], rtol=2e-6)


def test_philox_known_answers():
    # Random123 kat_vectors, philox4x32-10
    kat = [
        ((0, 0, 0, 0), (0, 0), (0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8)),
        ((0xFFFFFFFF,) * 4, (0xFFFFFFFF,) * 2, (0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD)),
        ((0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344), (0xA4093822, 0x299F31D0), (0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1)),
    ]
    for ctr, key, want in kat:
        got = orc.philox4x32(*[np.array([c]) for c in ctr], *key)
        assert tuple(int(x[0]) for x in got) == want


def test_categorical_and_reverse_update_properties():
    torch.manual_seed(0)
    p = torch.rand(4, 7, 21).softmax(-1)
    u = torch.rand(4, 7)
    s = orc.categorical_from_uniform(p, u)
    c = p.cumsum(-1)
    lo = torch.cat([torch.zeros_like(c[..., :1]), c[..., :-1]], -1).gather(-1, s[..., None])[..., 0]
    hi = c.gather(-1, s[..., None])[..., 0]
    assert ((u * p.sum(-1) >= lo - 1e-6) & (u * p.sum(-1) <= hi + 1e-6)).all()
    # at t = 1 the reverse update adds no noise and keeps un-generated residues
    sched = orc.cosine_variance_schedule(100, s=0.01, beta_max=0.999)
    x = torch.randn(2, 5, 3)
    O = orc.rotvec_to_matrix(torch.randn(2, 5, 3))
    seq = torch.randint(0, 20, (2, 5))
    den = {"translations_eps": torch.randn(2, 5, 3), "orientations_t0": orc.rotvec_to_matrix(torch.randn(2, 5, 3)),
           "seq_posterior": torch.rand(2, 5, 21).softmax(-1)}
    m = torch.tensor([[1, 0, 1, 0, 1], [0, 0, 1, 1, 1]]).bool()
    s1, x1, O1 = orc.reverse_update(1, seq, x, O, den, m, sched, torch.full_like(x, 1e9), torch.full_like(x, 1.0), torch.rand(2, 5))
    assert torch.equal(x1[~m], x[~m]) and torch.equal(O1[~m], O[~m]) and torch.equal(s1[~m], seq[~m])
    assert torch.equal(O1[m], den["orientations_t0"][m]) and x1.abs().max() < 1e6


def test_encode_context_oracle_vs_golden(golden):
    """ResidueEmbedding + PairEmbedding through DiffAb.encode_context, 4 flag combinations."""
    g = golden("encode_context")
    Bc, Kc, A_, D_, C_, seed = [int(v) for v in g["meta"]]
    sd = syn.context_state_dict(D_, C_, A_, 32, seed=seed)
    cb = syn.context_batch(Bc, Kc, A_, seed=seed)
    for gs in (True, False):
        for gq in (True, False):
            res, pair = orc.encode_context(sd, cb, gs, gq)
            assert maxrel(res, g[f"res_{int(gs)}{int(gq)}"]) < 2e-6
            assert maxrel(pair, g[f"pair_{int(gs)}{int(gq)}"]) < 2e-6
    # the flags matter (sequence masking changes both outputs, structure masking only the residue embedding)
    assert not np.allclose(g["res_11"], g["res_01"]) and not np.allclose(g["res_11"], g["res_10"])
    assert not np.allclose(g["pair_11"], g["pair_10"]) and np.array_equal(g["pair_11"], g["pair_01"])


def test_featurize_xyz_definitions():
    """SURVEY 8 row f2 (parity with protstruc UNPINNED): the oracle's geometric definitions against known answers - frames are
    the exact inverse of io.backbone_from_frames, a trans / cis / +90 degree dihedral, and the chain-break mask."""
    from diffab_pytorch import io

    g = torch.Generator().manual_seed(4)
    B, K = 2, 7
    t = 10 * torch.randn(B, K, 3, generator=g, dtype=torch.float64)
    O = orc.uniform_rotation_from_normals(torch.randn(B, K, 4, generator=g)).double()
    xyz = io.backbone_from_frames(t, O)  # (B,K,5,3): N, CA, C, O, CB
    chain = torch.tensor([[1, 1, 1, 2, 2, 2, 2], [1, 1, 1, 1, 1, 1, 1]])
    rmask = torch.ones(B, K, dtype=torch.bool)
    rmask[1, 4] = False
    f = orc.featurize_xyz(xyz, chain, rmask)
    assert torch.allclose(f["orientations"], O, atol=1e-6)  # (O itself is orthonormal to fp32 only) exact inverse of the reconstruction used on the output side
    m = f["backbone_dihedrals_mask"]
    assert m[0, :, 0].tolist() == [False, True, True, False, True, True, True]   # phi needs residue l-1 in the same chain
    assert m[0, :, 1].tolist() == [True, True, False, True, True, True, False]   # psi / omega need residue l+1
    assert m[1, :, 1].tolist() == [True, True, True, False, False, True, False]  # a missing residue breaks both neighbours
    assert torch.equal(m[..., 1], m[..., 2]) and (f["backbone_dihedrals"][~m] == 0).all()
    p = lambda *v: torch.tensor(v, dtype=torch.float64)
    assert abs(float(orc.dihedral(p(1, 1, 0), p(0, 1, 0), p(0, 0, 0), p(-1, 0, 0))) - np.pi) < 1e-12  # trans
    assert abs(float(orc.dihedral(p(1, 1, 0), p(0, 1, 0), p(0, 0, 0), p(1, 0, 0)))) < 1e-12          # cis
    assert abs(abs(float(orc.dihedral(p(1, 1, 0), p(0, 1, 0), p(0, 0, 0), p(0, 0, 1)))) - np.pi / 2) < 1e-12
    # pairwise: the (i, i+1) psi entry is the backbone psi, the (i-1, i) phi entry the backbone phi (same four atoms)
    pd, bd = f["pairwise_dihedrals"], f["backbone_dihedrals"]
    for l in range(K - 1):
        assert abs(float(pd[1, l, l + 1, 1]) - float(orc.dihedral(xyz[1, l, 0], xyz[1, l, 1], xyz[1, l, 2], xyz[1, l + 1, 0]))) < 1e-12
        if m[1, l, 1]:
            assert abs(float(pd[1, l, l + 1, 1]) - float(bd[1, l, 1])) < 1e-12
        if m[1, l + 1, 0]:
            assert abs(float(pd[1, l, l + 1, 0]) - float(bd[1, l + 1, 0])) < 1e-12


def test_oracle_bins_without_replacement_is_multinomial_without_replacement():
    """The exponential race of oracle.igso3_bins_without_replacement has the joint distribution of torch.multinomial(probs, K)
    (replacement=False, the reference's call at so3.py:78): checked on a small pmf where the inclusion probabilities are far from
    K * p - the frequency of every bin among the K draws, and of every bin as FIRST draw, against torch.multinomial's own."""
    import torch

    import diffab_oracle as orc

    torch.manual_seed(0)
    p = torch.tensor([0.5, 0.2, 0.1, 0.1, 0.05, 0.03, 0.02, 0.0])
    n, K = 40000, 3
    race = -torch.log(torch.rand(n, len(p)).clamp_min(1e-30))
    mine = orc.igso3_bins_without_replacement(p.expand(n, -1), race, K)
    ref = torch.multinomial(p.expand(n, -1), K)
    assert int((mine == 7).sum()) == 0  # a zero-mass bin is never drawn while enough positive ones exist
    for r in range(n // 1000):
        assert len(set(mine[r].tolist())) == K
    inc_m = torch.bincount(mine.flatten(), minlength=8).double() / n
    inc_r = torch.bincount(ref.flatten(), minlength=8).double() / n
    assert (inc_m - inc_r).abs().max() < 0.012, (inc_m, inc_r)
    first_m = torch.bincount(mine[:, 0], minlength=8).double() / n
    assert (first_m - p.double()).abs().max() < 0.01
    second_m = torch.bincount(mine[:, 1], minlength=8).double() / n
    second_r = torch.bincount(ref[:, 1], minlength=8).double() / n
    assert (second_m - second_r).abs().max() < 0.012
