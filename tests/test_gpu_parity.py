"""GPU parity tests proper: every C-ABI entry of the hot path (called through the boundary package, i.e.
through ctypes into libdiffab_hip.so) against the oracle on the same seeded inputs and against the golden
vectors generated from the real reference.  Tolerances are max|a-b|/max|b| in fp32; BASELINE.json's bar is
1e-4 for aa-type logits and translations - the tests hold every tensor to that or tighter."""
import numpy as np
import os

import pytest
import torch

import diffab_oracle as orc
from conftest import elemrel, elemrel_by_decade, maxrel
from diffab_pytorch import _hip, synthetic as syn

pytestmark = pytest.mark.gpu
T = torch.from_numpy
TOL = 1e-4  # BASELINE.json: "aa-type logits and translations within 1e-4 rel fp32"


@pytest.fixture(scope="module")
def hip():
    lib = _hip.lib()  # raises HipUnavailable when there is no gfx950 / no library: never a silent fallback
    assert lib.diffab_device_ok() == 1
    return lib


def well_conditioned(R):
    """The reference's own rule (tests/test_so3.py:56-59): log / scale_rot parity is defined only away from
    theta in {0, pi}, where theta/(2 sin theta) turns a 1-ulp acos difference into an O(1e-4) change."""
    cos = (R.diagonal(dim1=-2, dim2=-1).sum(-1) - 1) / 2
    return ((cos - 1).abs() >= 1e-2) & ((cos + 1).abs() >= 1e-2)


# ------------------------------------------------------------------ SO(3)
def test_so3_maps_vs_reference_goldens(hip, golden):
    from diffab_pytorch import so3

    g = golden("so3")
    R, k, v = T(g["R"]), T(g["k"]), T(g["v"])
    ok = well_conditioned(R)
    assert ok.float().mean() > 0.8
    S = so3.log_rotmat(R.cuda())
    assert S.is_cuda and maxrel(S.cpu()[ok], T(g["log"])[ok]) < 1e-5
    assert torch.allclose(S, -S.transpose(2, 3))  # reference tests/test_so3.py:31
    assert maxrel(so3.rotation_matrix_to_vector(R)[ok], T(g["rotvec"])[ok]) < 1e-5  # CPU in -> CPU out
    assert maxrel(so3.exp_skew_symmetric_mat(T(g["log"])), g["explog"]) < 2e-6
    assert maxrel(so3.scale_rot(R, k)[ok], T(g["scaled"])[ok]) < 1e-5
    assert maxrel(so3.vector_to_rotation_matrix(v), g["expv"]) < 2e-6
    assert np.array_equal(so3.vector_to_skew_symmetric_mat(v).numpy(), g["hat"])
    assert np.array_equal(so3.tensor_trace(R).numpy(), g["trace"])
    assert np.array_equal(so3.skew_symmetric_mat_to_vector(T(g["log"])).numpy(), g["rotvec"])
    with pytest.raises(ValueError):
        so3.scale_rot(R[0, 0], torch.ones(2, 2, 2, 2))


def test_so3_properties_like_reference_tests(hip):
    """reference tests/test_so3.py:44-93 at its own sizes (bsz=32, L=100)."""
    from diffab_pytorch import so3

    torch.manual_seed(0)
    R = so3.uniform(32, 100, 3, 3)
    assert R.shape == (32, 100, 3, 3)
    eye = torch.eye(3).expand_as(R)
    assert torch.allclose(R.transpose(2, 3) @ R, eye, rtol=1e-5, atol=1e-5)
    assert torch.allclose(torch.linalg.det(R), torch.ones(32, 100), atol=1e-5)
    rec = so3.exp_skew_symmetric_mat(so3.log_rotmat(R))
    cos = (so3.tensor_trace(R) - 1) / 2
    ok = ((cos - 1).abs() >= 1e-2) & ((cos + 1).abs() >= 1e-2)
    err = (R - rec).abs().sum((-1, -2))[ok]
    want = (R - orc.exp_so3(orc.log_so3(R))).abs().sum((-1, -2))[ok]  # the same test on the oracle
    print("exp(log R) sum|diff|: hip max %.3e, oracle max %.3e, hip > 1e-4: %d of %d" % (err.max(), want.max(), (err >= 1e-4).sum(), err.numel()))
    # the reference's own threshold is 1e-4 (tests/test_so3.py:61); on 3200 random rotations its formulation itself can
    # exceed it slightly near the mask edge, so hold the HIP path to "no worse than the reference on the same inputs"
    assert err.max() < max(1e-4, 1.25 * float(want.max()))
    Rs = so3.scale_rot(R, torch.rand(32))
    good = torch.isfinite(Rs).all(-1).all(-1)
    assert good.float().mean() > 0.99
    assert torch.allclose((Rs.transpose(2, 3) @ Rs)[good], eye[good], rtol=1e-5, atol=1e-5)
    # singular inputs give NaN exactly like the reference (so3.py:157-162, :235)
    assert torch.isnan(so3.log_rotmat(torch.eye(3).view(1, 1, 3, 3))).any()
    assert torch.isnan(so3.vector_to_rotation_matrix(torch.zeros(1, 1, 3))).any()


# ------------------------------------------------------------------ diffusers
def test_sequence_diffuser_vs_golden(hip, golden):
    from diffab_pytorch.diffusion import SequenceDiffuser

    g = golden("seqdiff")
    sd = SequenceDiffuser(T=100, s=0.01, beta_max=0.999)
    seq0, seqt, t, m = T(g["seq0"]), T(g["seqt"]), T(g["t"]), T(g["mask"])
    # (the schedule is recomputed on this host's CPU; torch.cos may differ by an ulp between CPU models)
    assert maxrel(sd.forward_prob_single_step(seqt, t, m), g["single"]) < 3e-7
    assert maxrel(sd.forward_prob_from_t0(seq0, t, m), g["from_t0"]) < 3e-7
    post = sd.posterior_single_step(seqt, seq0, t, m)
    assert maxrel(post, g["posterior"]) < 2e-6
    # sampling: matches the oracle's inverse-CDF draw on the same uniforms, keeps un-generated residues
    u = torch.rand(seq0.shape)
    st, post2 = sd.diffuse_from_t0(seq0, t, m, return_posterior=True, u=u)
    sched = orc.cosine_variance_schedule(100, s=0.01, beta_max=0.999)
    want = orc.categorical_from_uniform(orc.seq_forward_prob_from_t0(seq0, t, m, sched), u)
    assert torch.equal(st, want) and torch.equal(st[~m], seq0[~m])
    assert maxrel(post2, orc.seq_posterior_single_step(st, seq0, t, m, sched)) < 2e-6


def test_weighted_multinomial_and_single_step(hip, golden):
    """diffusion.py:38-41 against the reference's own output (bit-exact: two products and one sum per element), and
    diffuse_single_step (:81-103) against the oracle's inverse-CDF draw on the same uniforms."""
    from diffab_pytorch.diffusion import SequenceDiffuser, weighted_multinomial

    g = golden("seqdiff")
    p1 = torch.nn.functional.one_hot(T(g["seq0"]), 21)  # int64 one-hot, as the reference's call sites pass it (:74, :130)
    out = weighted_multinomial(p1, T(g["wm_p2"]), T(g["wm_w1"]), T(g["wm_w2"]))
    assert out.dtype == torch.float32 and np.array_equal(out.numpy(), g["wm_out"])
    out_dev = weighted_multinomial(p1.cuda(), T(g["wm_p2"]).cuda(), T(g["wm_w1"]).cuda(), T(g["wm_w2"]).cuda())
    assert out_dev.is_cuda and np.array_equal(out_dev.cpu().numpy(), g["wm_out"])
    with pytest.raises(ValueError):
        weighted_multinomial(p1, T(g["wm_p2"]), T(g["wm_w1"])[:2], T(g["wm_w2"]))
    sd = SequenceDiffuser(T=100, s=0.01, beta_max=0.999)
    seqt, t, m = T(g["seqt"]), T(g["t"]), T(g["mask"])
    u = torch.rand(seqt.shape, generator=torch.Generator().manual_seed(3))
    nxt = sd.diffuse_single_step(seqt, t, m, u=u)
    sched = orc.cosine_variance_schedule(100, s=0.01, beta_max=0.999)
    want = orc.categorical_from_uniform(orc.seq_forward_prob_single_step(seqt, t, m, sched), u)
    assert nxt.dtype == torch.int64 and torch.equal(nxt, want) and torch.equal(nxt[~m], seqt[~m])
    torch.manual_seed(11)  # production draw (Philox seeded from torch's generator): reproducible, and a valid residue type
    a = sd.diffuse_single_step(seqt, t, m)
    torch.manual_seed(11)
    assert torch.equal(a, sd.diffuse_single_step(seqt, t, m)) and int(a.min()) >= 0 and int(a.max()) <= 20


def test_sequence_diffuser_reference_properties(hip):
    """reference tests/test_diffusion.py:16-103 at its own sizes."""
    from diffab_pytorch.diffusion import SequenceDiffuser

    sd = SequenceDiffuser(T=100, s=0.01, beta_max=0.999)
    torch.manual_seed(1)
    bsz, L = 32, 100
    seq = torch.randint(0, 20, (bsz, L))
    allm = torch.ones(bsz, L).bool()
    one, ninety = torch.ones(bsz).long(), torch.full((bsz,), 90).long()
    for fn in (sd.forward_prob_single_step, sd.forward_prob_from_t0):
        p1, p90 = fn(seq, one, allm), fn(seq, ninety, allm)
        assert p1.shape == p90.shape == (bsz, L, 21)
        assert (p1.gather(-1, seq[..., None]) > p90.gather(-1, seq[..., None])).all()
    gm = torch.randint(0, 2, (bsz, L)).bool()
    ten = torch.full((bsz,), 10).long()
    s10 = sd.diffuse_from_t0(seq, ten, gm, return_posterior=False)
    post = sd.posterior_single_step(s10, seq, ten, gm)
    assert (post.gather(-1, seq[..., None]) > 1 / 20.0).all()
    s2 = sd.diffuse_from_t0(seq, torch.full((bsz,), 2).long(), allm, return_posterior=False)
    s99 = sd.diffuse_from_t0(seq, torch.full((bsz,), 99).long(), allm, return_posterior=False)
    assert (s2 != seq).sum() < (s99 != seq).sum()


def test_coordinate_diffuser_vs_golden(hip, golden):
    from diffab_pytorch.diffusion import CoordinateDiffuser

    g = golden("coorddiff")
    cd = CoordinateDiffuser(T=100, s=0.01, beta_max=0.999)
    xt, eps = cd.diffuse_from_t0(T(g["x0"]), T(g["t"]), T(g["mask"]), return_eps=True, eps=T(g["eps"]))
    assert maxrel(xt, g["xt"]) < 3e-7 and np.array_equal(eps.numpy(), g["eps"])
    sched = orc.cosine_variance_schedule(100, s=0.01, beta_max=0.999)  # same host CPU as the diffuser's schedule -> bit-exact
    assert torch.equal(xt, orc.coord_diffuse_from_t0(T(g["x0"]), T(g["t"]), T(g["mask"]), T(g["eps"]), sched))
    # own noise: eps ~ N(0,1), returned unmasked, seeded by torch's generator
    torch.manual_seed(7)
    x1, e1 = cd.diffuse_from_t0(torch.zeros(64, 128, 3), torch.full((64,), 50), torch.ones(64, 128).bool())
    torch.manual_seed(7)
    x2, e2 = cd.diffuse_from_t0(torch.zeros(64, 128, 3), torch.full((64,), 50), torch.ones(64, 128).bool())
    assert torch.equal(x1, x2) and abs(float(e1.mean())) < 0.02 and abs(float(e1.std()) - 1) < 0.02


def test_igso3_table_vs_golden(hip, golden):
    """The table DiffAb samples from is the REFERENCE's table (so3.py:52-72): series terms with the reference's fp32 roundings,
    including the rounding noise its clamp rectifies into ~1e-4 of tail mass on the small-sigma rows.  CDF within 5e-6 on every
    row >= 1 (row 0, sigma = 0, is finite garbage upstream: SURVEY B.4).  The float64 series is the opt-in accurate=True."""
    from diffab_pytorch import so3
    from diffab_pytorch.diffusion import OrientationDiffuser

    g = golden("igso3")
    od = OrientationDiffuser(T=100, s=0.01, beta_max=0.999)
    assert not od.so3.accurate
    tab = od.so3.histograms.cpu()
    assert tab.shape == (101, 8192) and torch.isfinite(tab).all() and (tab >= 0).all()
    rows = g["rows"].tolist()
    worst, same = 0.0, []
    for i, r in enumerate(rows):
        if r == 0:
            continue
        ref = T(g["probe_every16"][i])
        err = float((tab[r, ::16] - ref).abs().max() / ref.max())
        worst = max(worst, err)
        same.append(float((tab[r, ::16] == ref).float().mean()))
        # per bin: bit-identical where torch's SLEEF cos / sin / exp are correctly rounded (~95 % of the evaluations); a one-ulp
        # difference in cos(theta) alone moves a = (1 - cos theta)/pi by 6e-8 / (1 - cos theta), i.e. up to 1e-4 of the row
        # maximum on the narrowest row (measured with the same algorithm on the CPU: 1.1e-4 on row 1, 5e-6 on row 7)
        assert err < (2e-4 if r <= 3 else 3e-5), (r, err)
    sums = tab.double().sum(-1).numpy()[1:]
    cdf = od.so3._cdf.cpu()
    cdf_err = np.abs(cdf[1:, 511::512].numpy() - g["cdf_every512"][1:]).max()
    nz = (tab > 0).sum(-1).numpy()
    print("igso3 faithful table: worst pdf err / row max %.2e, bit-identical probes per row %s, row-sum rel %.2e, cdf abs %.2e, "
          "non-zero bins (rows 1..7) %s vs %s" % (worst, ["%.2f" % f for f in same], np.abs(sums / g["row_sums"][1:] - 1).max(), cdf_err,
                                                   nz[1:8].tolist(), g["row_nonzero"][1:8].tolist()))
    assert min(same) > 0.3  # a different series order or precision leaves no bin bit-identical
    assert np.allclose(sums, g["row_sums"][1:], rtol=5e-6)
    assert cdf_err < 5e-6  # what the sampler reads: the reference's CDF on EVERY row, the small-sigma rows 1..7 included
    assert np.abs(nz[1:8] - g["row_nonzero"][1:8]).max() <= 64  # support of the rectified tail noise
    assert (cdf[:, 1:] >= cdf[:, :-1]).all() and (cdf[:, -1] == 1).all()
    # opt-in exact density: differs from the reference by the reference's own fp32 error (row 1: ~3e-5 of the maximum, 2.5e-4 of mass)
    acc = so3.SO3(od.sched["one_minus_alpha_bar_sqrt"], accurate=True)
    tab_a, cdf_a = acc.histograms.cpu(), acc._cdf.cpu()
    for i, r in enumerate(rows):
        if r:
            assert float((tab_a[r, ::16] - T(g["probe_every16"][i])).abs().max() / T(g["probe_every16"][i]).max()) < 1e-4
    assert np.allclose(tab_a.double().sum(-1).numpy()[1:], np.pi and 8192 / np.pi, rtol=2e-6)  # exact normalisation n_bins / pi
    assert np.allclose(cdf_a[1:, 511::512].numpy(), g["cdf_every512"][1:], atol=2e-4)
    assert np.allclose(cdf_a[8:, 511::512].numpy(), g["cdf_every512"][8:], atol=5e-6)


def test_igso3_sampler_and_orientation_diffuser_vs_golden(hip, golden):
    from diffab_pytorch.diffusion import OrientationDiffuser

    g = golden("igso3")
    od = OrientationDiffuser(T=100, s=0.01, beta_max=0.999)
    tt = T(g["samp_t"])
    cdf = od.so3._cdf.cpu()
    bins = T(g["samp_bin"])
    rows = cdf[tt]  # (n, 8192)
    hi = rows.gather(1, bins)
    lo = torch.where(bins > 0, rows.gather(1, (bins - 1).clamp_min(0)), torch.zeros_like(hi))
    u_bin = (lo + hi) / 2  # a uniform that inverse-CDFs to the reference's captured bin
    ok = hi > lo
    rv = od.so3.sample_isotropic_gaussian(tt, bins.shape[1], axis_raw=T(g["samp_axis_raw"]), u_bin=u_bin, u_in=T(g["samp_u"]),
                                          z=T(g["samp_z"]))
    use = ok | (T(g["sigmas"])[tt] >= 0.1)[:, None]
    assert use.float().mean() > 0.95
    assert maxrel(rv[use], T(g["samp_rotvec"])[use]) < 2e-6
    # inverse-CDF draw equals the oracle's searchsorted on the same table
    u = torch.rand(len(tt), 12)
    th = od.so3.sample_from_histogram(tt, 12, u_bin=u, u_in=torch.zeros_like(u))
    want = orc.igso3_theta_from_hist(orc.igso3_bin_from_cdf(rows, u), torch.zeros_like(u))
    assert torch.allclose(th, want, atol=1e-6)
    Ot = od.diffuse_from_t0(T(g["od_O0"]), T(g["od_mask"]), tt, rotvec=T(g["samp_rotvec"]))
    okO = well_conditioned(T(g["od_O0"]))
    assert okO.float().mean() > 0.8 and maxrel(Ot[okO], T(g["od_Ot"])[okO]) < 1e-5 and torch.isfinite(Ot).all()
    # reference tests/test_diffusion.py:122-134 (shape only; non-rotation input allowed)
    out = od.diffuse_from_t0(torch.randn(32, 100, 3, 3), torch.randint(0, 2, (32, 100)).bool(), torch.full((32,), 50).long())
    assert out.shape == (32, 100, 3, 3)


def test_igso3_angle_distribution(hip):
    """SO(3) samples are distribution-equivalent to the reference sampler: the histogram branch follows the table's
    CDF (KS distance), the Gaussian branch is (2 sigma + sigma z) mod pi."""
    from diffab_pytorch.diffusion import OrientationDiffuser

    od = OrientationDiffuser(T=100, s=0.01, beta_max=0.999)
    torch.manual_seed(3)
    assert od.so3.without_replacement is True  # the default joint draw is the reference's (torch.multinomial without replacement)
    for t in (2, 5):
        def ks_of(th):
            cdf = od.so3._cdf[t].double().cpu()
            emp = torch.sort(th.flatten().double()).values
            idx = (emp / (np.pi / 8192)).long().clamp_max(8191)
            return float((cdf[idx] - torch.arange(1, len(emp) + 1) / len(emp)).abs().max())
        # independent draws (inverse CDF) follow the table row; so does the FIRST bin of each without-replacement draw - the joint draw
        # of 512 bins from a row whose mass sits in a few hundred bins does not (so3.py:78 has that property: the spread test below)
        assert ks_of(od.so3.sample_from_histogram(torch.full((64,), t), 512, without_replacement=False)) < 0.02, t
        assert ks_of(od.so3.sample_from_histogram(torch.full((4096,), t), 1)) < 0.03, t
    rv = od.so3.sample_isotropic_gaussian(torch.full((64,), 50), 512)
    ang = rv.norm(dim=-1)
    sg = float(od.sched["one_minus_alpha_bar_sqrt"][50])
    want = orc.igso3_theta_from_gaussian(torch.tensor(sg), torch.randn(200000))
    assert abs(float(ang.mean()) - float(want.mean())) < 0.02 and abs(float(ang.std()) - float(want.std())) < 0.02
    assert (ang < np.pi + 1e-5).all()
    ax = rv / ang[..., None]
    assert ax.mean((0, 1)).abs().max() < 0.02


def test_igso3_bins_without_replacement(hip):
    """so3.py:78 draws the K bins of a patch with torch.multinomial's default, i.e. WITHOUT replacement.  The HIP race kernel against
    the oracle's restatement on the same Exp(1) draws (bit-exact bin sequences), no bin twice in a patch, the first draw of a patch
    distributed like the table row, and - the signature of a draw without replacement - a row whose mass sits in a few bins spreads
    over more than those bins, as the reference's own torch.multinomial does."""
    from diffab_pytorch.diffusion import OrientationDiffuser

    od = OrientationDiffuser(T=100, s=0.01, beta_max=0.999)
    so3 = od.so3
    g = torch.Generator().manual_seed(5)
    rows = torch.tensor([1, 2, 5, 9, 9, 30, 2, 1])
    for K in (19, 128, 256):
        race = -torch.log(torch.rand(len(rows), so3.n_bins, generator=g).clamp_min(1e-30))
        bins = so3.draw_bins_without_replacement(rows, K, race=race).cpu().long()
        want = orc.igso3_bins_without_replacement(so3.histograms[rows].cpu(), race, K)
        assert torch.equal(bins, want), (K, int((bins != want).sum()))
        for r in range(len(rows)):
            assert len(set(bins[r].tolist())) == K  # no bin twice
    # first draw of a row ~ the row's pmf (KS against the CDF); 4096 independent races of row 5
    n = 4096
    torch.manual_seed(11)
    first = so3.draw_bins_without_replacement(torch.full((n,), 5), 4)[:, 0].cpu().long()
    cdf = so3._cdf[5].double().cpu()
    emp = torch.sort(first).values
    ks = (cdf[emp] - torch.arange(1, n + 1) / n).abs().max()
    assert ks < 0.035, float(ks)
    # joint behaviour against torch.multinomial itself (the reference's call) on a peaked row: number of distinct bins among the K
    # draws is K for both, and the spread (95 % quantile of the drawn angles) agrees - with replacement it would be far narrower
    t, K = 1, 128
    p = so3.histograms[t].cpu()
    ref = torch.multinomial(p.expand(256, -1), K)  # (256, K), each row without replacement
    mine = so3.draw_bins_without_replacement(torch.full((256,), t), K).cpu().long()
    q_ref, q_mine = ref.double().quantile(0.95), mine.double().quantile(0.95)
    with_repl = torch.multinomial(p.expand(256, -1), K, replacement=True).double().quantile(0.95)
    assert abs(float(q_ref - q_mine)) < 0.05 * float(q_ref), (float(q_ref), float(q_mine))
    assert float(q_ref) > 1.02 * float(with_repl) or float(q_ref - with_repl) >= 1.0, (float(q_ref), float(with_repl))
    # the sampler entry with the bins given: theta inside the drawn bin
    th = so3.sample_from_histogram(rows, 128, without_replacement=True)
    assert th.shape == (len(rows), 128) and bool(((th >= 0) & (th < np.pi)).all())
    rv = so3.sample_isotropic_gaussian(torch.tensor([2, 60]), 64, without_replacement=True)
    assert rv.shape == (2, 64, 3) and bool(torch.isfinite(rv).all())


# ------------------------------------------------------------------ denoiser
CASES = ["unit_wide", "unit_tight", "unit_ragged", "bench_wide", "bench_tight", "bench_k256"]


def build_case(g):
    from diffab_pytorch.diffab_pytorch import Denoiser

    B, K, seed, D, C, NL, DS, H, PQ, PV = [int(v) for v in g["meta"]]
    dims = dict(D=D, C=C, NL=NL, DS=DS, H=H, PQ=PQ, PV=PV, V=21)
    den = Denoiser(D, C, NL, DS, PQ, PV, H, 21)
    den.load_state_dict(syn.denoiser_state_dict(dims, seed=seed, prefix=""), strict=True)
    den = den.cuda().requires_grad_(False)  # inference kernels (the autograd route: tests/test_gpu_autograd.py)
    inp = {k: v.cuda() for k, v in syn.patches(B, K, dims, seed=seed, coord_sigma=float(g["coord_sigma"])).items()}
    return dims, den, inp


@pytest.mark.parametrize("flags", [0, _hip.FLAG_FORCE_GENERIC, _hip.FLAG_FP32_GEMM, _hip.FLAG_PAIR_PLANES],
                         ids=["dispatch", "generic", "fp32gemm", "pairplanes"])
@pytest.mark.parametrize("name", CASES)
def test_denoiser_vs_reference_goldens(hip, golden, name, flags):
    g = golden("denoiser_" + name)
    dims, den, inp = build_case(g)
    out = den(inp["seq_idx"], inp["translations"], inp["orientations"], inp["res_context_emb"], inp["pair_context_emb"],
              T(g["beta"]).cuda(), inp["generation_mask"], inp["residue_mask"], return_logits=True, flags=flags)
    assert set(out) >= {"translations_eps", "orientations_t0", "seq_posterior"}
    for k in ("res_emb", "aa_logits", "translations_eps", "orientations_t0", "seq_posterior"):
        assert torch.isfinite(out[k]).all(), k
        assert maxrel(out[k], g[k]) < TOL, (name, k, maxrel(out[k], g[k]))
    # element-wise form of the bar (north_star: "aa-type logits and translations within 1e-4 rel"): every element above 1 % of the
    # tensor's maximum individually (conftest.elemrel: below that the reference's own fp32-vs-fp64 noise exceeds 1e-4 of the
    # element), not only the tensor-global norm; the worst element-wise error per decade of magnitude is printed
    for k in ("aa_logits", "translations_eps"):
        assert elemrel(out[k], g[k]) < TOL, (name, k, elemrel(out[k], g[k]))
        print(f"{name} flags={flags} {k}: worst element-wise relative error per decade of |ref| / max|ref|:",
              {f"1e-{d_}": f"{v_:.1e}" for d_, v_ in elemrel_by_decade(out[k], g[k]).items()})
    l0 = den.ipa.layers[0](inp["res_context_emb"], inp["pair_context_emb"], inp["orientations"], inp["translations"], flags=flags)
    assert maxrel(l0, g["ipa_layer0"]) < TOL
    # masks are ignored by the denoiser exactly like the reference (diffab_pytorch.py:566-567)
    out2 = den(inp["seq_idx"], inp["translations"], inp["orientations"], inp["res_context_emb"], inp["pair_context_emb"],
               T(g["beta"]).cuda(), ~inp["generation_mask"], ~inp["residue_mask"], flags=flags)
    assert torch.equal(out2["translations_eps"], out["translations_eps"])


@pytest.mark.parametrize("name", ["bench_wide", "bench_tight", "bench_k256"])
def test_value_planes_variant_vs_reference_goldens(hip, golden, name):
    """diffab_debug_set_attn_variant(16) (round 6, opt-in): the projection tile writes the value side (v_s, global value points relative
    to the patch's first translation) as two fp16 planes under a bound-derived power-of-two scale, and phase 3 of the attention tile runs
    P x V on the f16 matrix cores (three exact partial products, the probability mass through a ones column).  Same bar as every other
    form: the reference goldens at the benchmark geometry (K = 128 wide / tight patches, K = 256 chunked), pair planes on (the variant
    applies to the plane kernels), every output < 1e-4 on the tensor-global and the element-wise norm - and NOT bitwise the default form."""
    g = golden("denoiser_" + name)
    dims, den, inp = build_case(g)
    args = (inp["seq_idx"], inp["translations"], inp["orientations"], inp["res_context_emb"], inp["pair_context_emb"], T(g["beta"]).cuda(),
            inp["generation_mask"], inp["residue_mask"])
    base = den(*args, return_logits=True, flags=_hip.FLAG_PAIR_PLANES)
    try:
        hip.diffab_debug_set_attn_variant(16)
        out = den(*args, return_logits=True, flags=_hip.FLAG_PAIR_PLANES)
        l0 = den.ipa.layers[0](inp["res_context_emb"], inp["pair_context_emb"], inp["orientations"], inp["translations"], flags=_hip.FLAG_PAIR_PLANES)
    finally:
        hip.diffab_debug_set_attn_variant(0)
    for k in ("res_emb", "aa_logits", "translations_eps", "orientations_t0", "seq_posterior"):
        assert torch.isfinite(out[k]).all(), k
        assert maxrel(out[k], g[k]) < TOL, (name, k, maxrel(out[k], g[k]))
    for k in ("aa_logits", "translations_eps"):
        assert elemrel(out[k], g[k]) < TOL, (name, k, elemrel(out[k], g[k]))
    assert maxrel(l0, g["ipa_layer0"]) < TOL
    assert not torch.equal(out["res_emb"], base["res_emb"])  # the variant really ran (different arithmetic, same answer)
    print(f"value planes {name}: res_emb vs golden {maxrel(out['res_emb'], g['res_emb']):.1e} (default form {maxrel(base['res_emb'], g['res_emb']):.1e})")


def test_denoiser_reference_test_shapes(hip):
    """reference tests/test_modules.py:143-248: unseeded random inputs, non-rotation 'orientations', shapes only."""
    from diffab_pytorch.diffab_pytorch import Denoiser, InvariantPointAttentionLayer, InvariantPointAttentionModule

    ipa = InvariantPointAttentionLayer(32, 16, 16, 4, 4, 8).cuda().requires_grad_(False)
    x, e = torch.rand(32, 16, 32), torch.rand(32, 16, 16, 16)
    r, t = torch.rand(32, 16, 3, 3), torch.rand(32, 16, 3)
    y = ipa(x, e, r, t)
    assert y.shape == (32, 16, 32) and not y.is_cuda
    want = orc.ipa_layer(x, e, r, t, {k: v.cpu() for k, v in ipa.state_dict().items()}, "", 8)
    assert maxrel(y, want) < TOL
    mod = InvariantPointAttentionModule(4, 32, 16, 16, 4, 4, 8).cuda().requires_grad_(False)
    assert mod(x, torch.randn(32, 16, 16, 16), r, t).shape == (32, 16, 32)
    den = Denoiser(32, 16, 4, 12, 4, 4, 8, aa_vocab_size=21).cuda().requires_grad_(False)
    out = den(torch.randint(0, 20, (32, 16)), t, r, x, torch.randn(32, 16, 16, 16), torch.rand(32), torch.randint(0, 2, (32, 16)),
              torch.randint(0, 2, (32, 16)))
    assert out["translations_eps"].shape == (32, 16, 3)
    assert out["orientations_t0"].shape == (32, 16, 3, 3)
    assert out["seq_posterior"].shape == (32, 16, 21)


def test_denoiser_vs_oracle_batch_and_permutation(hip):
    """B > 1 at the benchmark geometry against the oracle, and patch-permutation equivariance (bitwise)."""
    from diffab_pytorch.diffab_pytorch import Denoiser

    dims = dict(syn.BENCH_DIMS, NL=2)
    den = Denoiser(dims["D"], dims["C"], dims["NL"], dims["DS"], dims["PQ"], dims["PV"], dims["H"], 21)
    sd = syn.denoiser_state_dict(dims, seed=5, prefix="")
    den.load_state_dict(sd)
    den = den.cuda().requires_grad_(False)
    inp = syn.patches(3, 128, dims, seed=5, coord_sigma=6.0)
    beta = torch.tensor([0.01, 0.3, 0.9])
    args = [inp[k] for k in ("seq_idx", "translations", "orientations", "res_context_emb", "pair_context_emb")]
    out = den(*[a.cuda() for a in args], beta.cuda(), None, None, return_logits=True)
    want = orc.denoiser({"denoiser." + k: v for k, v in sd.items()}, *args, beta, dims["NL"], dims["H"])
    for k in ("aa_logits", "translations_eps", "orientations_t0", "seq_posterior"):
        assert maxrel(out[k], want[k]) < TOL, (k, maxrel(out[k], want[k]))
    perm = torch.tensor([2, 0, 1])
    outp = den(*[a[perm].cuda() for a in args], beta[perm].cuda(), None, None)
    for k in ("translations_eps", "orientations_t0", "seq_posterior"):
        assert torch.equal(outp[k].cpu(), out[k].cpu()[perm]), k


@pytest.mark.parametrize("K", [64, 192, 256])
def test_fast_path_key_chunks_vs_generic_and_oracle(hip, K):
    """K = 64 (one 64-key chunk), 192 (three 64-key chunks) and 256 (two 128-key chunks): the online-softmax chunk loop of the
    MFMA attention kernel against the generic kernels (same device) and the oracle."""
    from diffab_pytorch.diffab_pytorch import Denoiser

    dims = dict(syn.BENCH_DIMS, NL=2)
    den = Denoiser(dims["D"], dims["C"], dims["NL"], dims["DS"], dims["PQ"], dims["PV"], dims["H"], 21)
    sd = syn.denoiser_state_dict(dims, seed=6, prefix="")
    den.load_state_dict(sd)
    den = den.cuda().requires_grad_(False)
    inp = syn.patches(2, K, dims, seed=60 + K, coord_sigma=5.0)
    beta = torch.tensor([0.02, 0.6])
    args = [inp[k] for k in ("seq_idx", "translations", "orientations", "res_context_emb", "pair_context_emb")]
    fast = den(*[a.cuda() for a in args], beta.cuda(), None, None, return_logits=True)
    gen = den(*[a.cuda() for a in args], beta.cuda(), None, None, return_logits=True, flags=_hip.FLAG_FORCE_GENERIC)
    want = orc.denoiser({"denoiser." + k: v for k, v in sd.items()}, *args, beta, dims["NL"], dims["H"])
    for k in ("res_emb", "aa_logits", "translations_eps", "orientations_t0", "seq_posterior"):
        assert maxrel(fast[k], want[k]) < TOL, (K, k, maxrel(fast[k], want[k]))
        assert maxrel(fast[k], gen[k]) < TOL, (K, k, maxrel(fast[k], gen[k]))
    # the arithmetic selectors (fp32 dense kernels, fp16 pair planes) at this K: 4-tile instantiations for K = 64, chunked ones for 192 / 256
    for fl in (_hip.FLAG_FP32_GEMM, _hip.FLAG_PAIR_PLANES):
        alt = den(*[a.cuda() for a in args], beta.cuda(), None, None, return_logits=True, flags=fl)
        for k in ("res_emb", "aa_logits", "translations_eps", "orientations_t0", "seq_posterior"):
            assert maxrel(alt[k], want[k]) < TOL, (K, fl, k, maxrel(alt[k], want[k]))


def test_denoiser_translation_offset_robustness(hip):
    """The reference feeds raw PDB coordinates (diffab_pytorch.py:820); a 150 A offset must not break parity
    (reference fp32 noise floor rises to 6e-6 there)."""
    from diffab_pytorch.diffab_pytorch import Denoiser

    dims = dict(syn.BENCH_DIMS, NL=1)
    den = Denoiser(dims["D"], dims["C"], 1, dims["DS"], dims["PQ"], dims["PV"], dims["H"], 21)
    sd = syn.denoiser_state_dict(dims, seed=9, prefix="")
    den.load_state_dict(sd)
    den = den.cuda().requires_grad_(False)
    inp = syn.patches(1, 128, dims, seed=9, coord_sigma=8.0)
    x = inp["translations"] + torch.tensor([150.0, -90.0, 40.0])
    args = [inp["seq_idx"], x, inp["orientations"], inp["res_context_emb"], inp["pair_context_emb"]]
    beta = torch.tensor([0.05])
    out = den(*[a.cuda() for a in args], beta.cuda(), None, None, return_logits=True)
    want64 = orc.denoiser({"denoiser." + k: v.double() for k, v in sd.items()}, args[0], x.double(), args[2].double(), args[3].double(),
                          args[4].double(), beta.double(), 1, dims["H"])
    for k in ("aa_logits", "translations_eps"):
        assert maxrel(out[k], want64[k]) < TOL, (k, maxrel(out[k], want64[k]))


# ------------------------------------------------------------------ losses, reverse step, sampler
def test_losses_vs_golden(hip, golden):
    from diffab_pytorch import DiffAb
    from diffab_pytorch.diffab_pytorch import OrientationLoss

    g = golden("losses_grads")
    B, K, seed = [int(v) for v in g["meta"][:3]]
    dims = dict(zip(("D", "C", "NL", "DS", "H", "PQ", "PV"), [int(v) for v in g["meta"][3:]]), V=21)
    model = DiffAb(dims["D"], dims["C"], dims["NL"], dims["DS"], dims["PQ"], dims["PV"], dims["H"]).cuda()
    model.denoiser.load_state_dict(syn.denoiser_state_dict(dims, seed=seed, prefix=""))
    inp = syn.patches(B, K, dims, seed=seed, coord_sigma=4.0)
    den = model.denoise(T(g["seq_t"]).cuda(), T(g["x_t"]).cuda(), T(g["O_t"]).cuda(), inp["res_context_emb"].cuda(),
                        inp["pair_context_emb"].cuda(), T(g["beta"]).cuda(), None, None)
    for k, gk in (("translations_eps", "out_eps"), ("orientations_t0", "out_O0"), ("seq_posterior", "out_post")):
        assert maxrel(den[k], g[gk]) < TOL
    noised = {"seq_posterior": T(g["post"]), "translations_eps": T(g["eps"])}
    ls = model.hotpath_losses(den, noised, inp["orientations"], T(g["gen"]), T(g["resm"]))
    np.testing.assert_allclose([float(x) for x in ls], g["losses"], rtol=5e-5)
    # reference tests/test_loss.py:9-21 (float64 input, loss(R, R) ~ 0)
    R = orc.rotvec_to_matrix(torch.randn(16, 20, 3).double())
    loss = OrientationLoss(reduction="mean")(R, R)
    assert loss.shape == () and loss.dtype == torch.float64 and float(loss) == pytest.approx(0.0, abs=1e-9)
    el = OrientationLoss(reduction="none")(T(g["out_O0"]), inp["orientations"])
    assert maxrel(el, orc.orientation_loss_elems(T(g["out_O0"]), inp["orientations"])) < 1e-5
    # state_dict of the full module: reference keys, parameters only
    sd = model.state_dict()
    assert "residue_context_embedding.mlp.0.weight" in sd and "pair_context_embedding.pair2distcoef.weight" in sd
    assert len(list(model.buffers())) == 0


def test_philox_matches_oracle_bitwise(hip):
    out = torch.empty(3, 16, 4, dtype=torch.float32, device="cuda")
    seed = 0x1234_5678_9ABC_DEF0
    _hip.check(hip.diffab_philox_fill(seed, 7, 3, 16, 42, 2, 1, _hip.ptr(out), _hip.stream_ptr()), "philox")
    patch = (7 + np.arange(3))[:, None] + np.zeros((3, 16), dtype=np.int64)
    res = np.zeros((3, 16), dtype=np.int64) + np.arange(16)[None, :]
    want = np.stack(orc.philox_uniform4(seed, patch, res, 42, 2), -1)
    assert np.array_equal(out.cpu().numpy(), want)
    _hip.check(hip.diffab_philox_fill(seed, 7, 3, 16, 42, 2, 0, _hip.ptr(out), _hip.stream_ptr()), "philox")
    wantn = np.stack(orc.philox_normal4(seed, patch, res, 42, 2), -1)
    assert np.abs(out.cpu().numpy() - wantn).max() < 2e-6
    big = torch.empty(256, 128, 4, dtype=torch.float32, device="cuda")
    _hip.check(hip.diffab_philox_fill(1, 0, 256, 128, 1, 1, 0, _hip.ptr(big), _hip.stream_ptr()), "philox")
    assert abs(float(big.mean())) < 0.01 and abs(float(big.std()) - 1) < 0.01


def _unit_model(NL=2, seed=17):
    from diffab_pytorch import DiffAb

    dims = dict(syn.UNIT_DIMS, NL=NL)
    model = DiffAb(dims["D"], dims["C"], dims["NL"], dims["DS"], dims["PQ"], dims["PV"], dims["H"]).cuda()
    sd = syn.denoiser_state_dict(dims, seed=seed, prefix="")
    model.denoiser.load_state_dict(sd)
    return dims, model, {"denoiser." + k: v for k, v in sd.items()}


def test_reverse_step_teacher_forced_vs_oracle(hip):
    """One reverse step t -> t-1 (denoise + Philox noise + IGSO3 draw + update) against the oracle, teacher-forced
    at realistic coordinates for several t, including the histogram branch of the reverse table (small sqrt(beta))
    and t = 1 (no noise)."""
    dims, model, sd = _unit_model()
    sched = orc.cosine_variance_schedule(100, s=0.01, beta_max=0.999)
    B, K, seed = 3, 16, 991
    inp = syn.patches(B, K, dims, seed=4, coord_sigma=5.0)
    gm = inp["generation_mask"]
    rev = model._reverse_so3()
    sig = sched["beta"].sqrt()
    for t in (100, 57, 8, 2, 1):
        got = model.sample(inp["seq_idx"], inp["translations"], inp["orientations"], res_context_emb=inp["res_context_emb"],
                           pair_context_emb=inp["pair_context_emb"], generation_mask=gm, seed=seed, first_patch=10, t_start=t,
                           t_stop=t - 1, init=False)
        patch = (10 + np.arange(B))[:, None] + np.zeros((B, K), dtype=np.int64)
        res = np.zeros((B, K), dtype=np.int64) + np.arange(K)[None, :]
        z = torch.from_numpy(np.stack(orc.philox_normal4(seed, patch, res, t, orc.STREAM_TRANS)[:3], -1))
        ax = torch.from_numpy(np.stack(orc.philox_normal4(seed, patch, res, t, orc.STREAM_AXIS)[:3], -1))
        ua = orc.philox_uniform4(seed, patch, res, t, orc.STREAM_ANGLE)
        na = orc.philox_normal4(seed, patch, res, t, orc.STREAM_ANGLE)
        us = torch.from_numpy(orc.philox_uniform4(seed, patch, res, t, orc.STREAM_SEQ)[0])
        cdf_row = rev._cdf[t].cpu()[None, None, :].expand(B, K, -1)
        th_h = orc.igso3_theta_from_hist(orc.igso3_bin_from_cdf(cdf_row, torch.from_numpy(ua[0])), torch.from_numpy(ua[1]))
        th_g = orc.igso3_theta_from_gaussian(sig[t].expand(B, K), torch.from_numpy(na[2]))
        rotvec = orc.igso3_rotvec(ax, th_h, th_g, sig[t].expand(B))
        den = orc.denoiser(sd, inp["seq_idx"], inp["translations"], inp["orientations"], inp["res_context_emb"], inp["pair_context_emb"],
                           sched["beta"][t].expand(B), dims["NL"], dims["H"])
        s1, x1, O1 = orc.reverse_update(t, inp["seq_idx"], inp["translations"], inp["orientations"], den, gm, sched, z, rotvec, us)
        assert maxrel(got["translations"], x1) < TOL, t
        assert maxrel(got["orientations"], O1) < TOL, t
        diff = got["seq_idx"] != s1  # a draw can flip only when u sits on an edge of the posterior's CDF
        if diff.any():
            edge = (den["seq_posterior"].double().cumsum(-1) - us.double()[..., None]).abs().min(dim=-1).values
            assert float(edge[diff].max()) < 1e-5, (t, int(diff.sum()), float(edge[diff].max()))
        assert torch.equal(got["translations"][~gm], inp["translations"][~gm])
        assert torch.equal(got["seq_idx"][~gm], inp["seq_idx"][~gm])


@pytest.mark.parametrize("K", [128, 256])
def test_reverse_step_teacher_forced_at_benchmark_geometry(hip, K):
    """The path bench.py times - diffab_sample_loop on the MFMA kernels with prepared weight planes, the fp16 pair planes and
    reverse_update_philox - teacher-forced against the oracle at the benchmark dims: K = 128 (single-chunk planes kernel) and
    K = 256 (chunked planes kernel), B = 2, NL = 2, t in {100, 57, 8, 2, 1}.  x and O within 1e-4; every sequence draw that differs
    from the oracle's must sit on an edge of the posterior's CDF (the draw is u < cumsum(p): a 1e-6 difference in p flips it only
    there), and their number is printed."""
    from diffab_pytorch import DiffAb

    dims = dict(syn.BENCH_DIMS, NL=2)
    torch.manual_seed(0)
    model = DiffAb(dims["D"], dims["C"], dims["NL"], dims["DS"], dims["PQ"], dims["PV"], dims["H"]).cuda()
    sd0 = syn.denoiser_state_dict(dims, seed=19, prefix="")
    model.denoiser.load_state_dict(sd0)
    sd = {"denoiser." + k: v for k, v in sd0.items()}
    sched = orc.cosine_variance_schedule(100, s=0.01, beta_max=0.999)
    B, seed = 2, 4242
    inp = syn.patches(B, K, dims, seed=40 + K, coord_sigma=6.0)
    gm = inp["generation_mask"].clone()
    gm[:, : K // 2] = True  # half of every patch is generated: enough draws to see CDF-edge flips
    rev = model._reverse_so3()
    sig = sched["beta"].sqrt()
    flips = 0
    for t in (100, 57, 8, 2, 1):
        got = model.sample(inp["seq_idx"], inp["translations"], inp["orientations"], res_context_emb=inp["res_context_emb"],
                           pair_context_emb=inp["pair_context_emb"], generation_mask=gm, seed=seed, first_patch=3, t_start=t,
                           t_stop=t - 1, init=False)
        patch = (3 + np.arange(B))[:, None] + np.zeros((B, K), dtype=np.int64)
        res = np.zeros((B, K), dtype=np.int64) + np.arange(K)[None, :]
        z = torch.from_numpy(np.stack(orc.philox_normal4(seed, patch, res, t, orc.STREAM_TRANS)[:3], -1))
        ax = torch.from_numpy(np.stack(orc.philox_normal4(seed, patch, res, t, orc.STREAM_AXIS)[:3], -1))
        ua = orc.philox_uniform4(seed, patch, res, t, orc.STREAM_ANGLE)
        na = orc.philox_normal4(seed, patch, res, t, orc.STREAM_ANGLE)
        us = torch.from_numpy(orc.philox_uniform4(seed, patch, res, t, orc.STREAM_SEQ)[0])
        cdf_row = rev._cdf[t].cpu()[None, None, :].expand(B, K, -1)
        th_h = orc.igso3_theta_from_hist(orc.igso3_bin_from_cdf(cdf_row, torch.from_numpy(ua[0])), torch.from_numpy(ua[1]))
        th_g = orc.igso3_theta_from_gaussian(sig[t].expand(B, K), torch.from_numpy(na[2]))
        rotvec = orc.igso3_rotvec(ax, th_h, th_g, sig[t].expand(B))
        den = orc.denoiser(sd, inp["seq_idx"], inp["translations"], inp["orientations"], inp["res_context_emb"], inp["pair_context_emb"],
                           sched["beta"][t].expand(B), dims["NL"], dims["H"])
        s1, x1, O1 = orc.reverse_update(t, inp["seq_idx"], inp["translations"], inp["orientations"], den, gm, sched, z, rotvec, us)
        assert maxrel(got["translations"], x1) < TOL, (K, t, maxrel(got["translations"], x1))
        assert maxrel(got["orientations"], O1) < TOL, (K, t, maxrel(got["orientations"], O1))
        diff = (got["seq_idx"].cpu() != s1)
        if diff.any():  # each flipped draw: u within 1e-5 of a cumulative-probability edge of the oracle's posterior
            cdf = den["seq_posterior"].double().cumsum(-1)
            edge = (cdf - us.double()[..., None]).abs().min(dim=-1).values
            assert float(edge[diff].max()) < 1e-5, (K, t, float(edge[diff].max()))
            flips += int(diff.sum())
        assert torch.equal(got["translations"].cpu()[~gm], inp["translations"][~gm])
        assert torch.equal(got["seq_idx"].cpu()[~gm], inp["seq_idx"][~gm])
    print(f"teacher-forced reverse steps at the benchmark geometry, K={K}: {flips} of {5 * int(gm.sum())} sequence draws on a CDF edge")


def test_explicit_noise_reverse_update_vs_oracle(hip):
    import ctypes as C

    dims, model, sd = _unit_model(NL=1)
    sched = orc.cosine_variance_schedule(100, s=0.01, beta_max=0.999)
    B, K, t = 2, 16, 33
    torch.manual_seed(0)
    x, O = torch.randn(B, K, 3) * 5, orc.rotvec_to_matrix(torch.randn(B, K, 3))
    seq = torch.randint(0, 20, (B, K))
    den = {"translations_eps": torch.randn(B, K, 3), "orientations_t0": orc.rotvec_to_matrix(torch.randn(B, K, 3)),
           "seq_posterior": torch.rand(B, K, 21).softmax(-1)}
    gm = torch.rand(B, K) < 0.5
    z, rv, u = torch.randn(B, K, 3), torch.randn(B, K, 3) * 0.3, torch.rand(B, K)
    s1, x1, O1 = orc.reverse_update(t, seq, x, O, den, gm, sched, z, rv, u)
    sq, xq, Oq = seq.cuda(), x.cuda(), O.cuda()
    dv = {k: v.cuda().contiguous() for k, v in den.items()}
    gq, zq, rq, uq = gm.cuda(), z.cuda(), rv.cuda(), u.cuda()  # keep the device buffers alive across the launch
    sdv = model._sched_on_device()
    _hip.check(hip.diffab_reverse_update(C.byref(sdv.struct), t, _hip.ptr(sq), _hip.ptr(xq), _hip.ptr(Oq), _hip.ptr(dv["translations_eps"]),
                                         _hip.ptr(dv["orientations_t0"]), _hip.ptr(dv["seq_posterior"]), _hip.ptr(gq),
                                         _hip.ptr(zq), _hip.ptr(rq), _hip.ptr(uq), B, K, 21, _hip.stream_ptr()), "rev")
    assert maxrel(xq, x1) < 1e-6 and maxrel(Oq, O1) < 1e-6 and torch.equal(sq.cpu(), s1)


def test_sample_loop_shard_invariance_and_determinism(hip):
    """100-step reverse loop: finite, reproducible, context untouched, and identical under any sharding of the
    batch (noise is keyed by the GLOBAL patch id) - the N>1 correctness property on one device."""
    dims, model, _ = _unit_model()
    B, K = 6, 16
    inp = syn.patches(B, K, dims, seed=8, coord_sigma=5.0)
    kw = dict(res_context_emb=inp["res_context_emb"], pair_context_emb=inp["pair_context_emb"], generation_mask=inp["generation_mask"])
    full = model.sample(inp["seq_idx"], inp["translations"], inp["orientations"], seed=123, **kw)
    again = model.sample(inp["seq_idx"], inp["translations"], inp["orientations"], seed=123, **kw)
    gm = inp["generation_mask"]
    for k in full:
        assert torch.equal(full[k], again[k]), k
    assert torch.isfinite(full["translations"]).all() and torch.isfinite(full["orientations"]).all()
    assert torch.equal(full["translations"][~gm], inp["translations"][~gm])
    assert torch.equal(full["orientations"][~gm], inp["orientations"][~gm])
    assert ((full["seq_idx"] >= 0) & (full["seq_idx"] < 21)).all()
    Og = full["orientations"][gm]
    assert torch.allclose(Og.transpose(-1, -2) @ Og, torch.eye(3).expand_as(Og), atol=1e-3)
    parts = []
    for lo, hi in ((0, 2), (2, 3), (3, 6)):
        sl = slice(lo, hi)
        parts.append(model.sample(inp["seq_idx"][sl], inp["translations"][sl], inp["orientations"][sl], seed=123, first_patch=lo,
                                  res_context_emb=inp["res_context_emb"][sl], pair_context_emb=inp["pair_context_emb"][sl],
                                  generation_mask=gm[sl]))
    for k in full:
        assert torch.equal(torch.cat([p[k] for p in parts]), full[k]), k
    other = model.sample(inp["seq_idx"], inp["translations"], inp["orientations"], seed=124, **kw)
    assert not torch.equal(other["translations"], full["translations"])


def test_skip_unused_rows_sampler_is_bitwise_the_full_sampler(hip):
    """DIFFAB_FLAG_SKIP_UNUSED_ROWS: the last layer's attention only for row tiles with a generated residue = the full loop, bit for bit
    (MFMA path, K = 128 and the chunked K = 256; a patch without any generated residue, one with all of them; eager and graph replay)."""
    from diffab_pytorch import DiffAb

    bd = dict(syn.BENCH_DIMS, NL=2)
    torch.manual_seed(0)
    big = DiffAb(bd["D"], bd["C"], bd["NL"], bd["DS"], bd["PQ"], bd["PV"], bd["H"]).cuda()
    for B_, K_ in ((5, 128), (2, 256)):
        bi = {k: v.cuda() for k, v in syn.patches(B_, K_, bd, seed=31).items()}
        gm = bi["generation_mask"].clone()
        gm[0] = False   # nothing to generate: every tile of the last layer is skipped
        gm[1] = True    # everything generated: nothing is skipped
        skipped = 1.0 - gm.view(B_, K_ // 16, 16).any(-1).float().mean().item()
        assert 0.3 < skipped < 0.95, skipped
        kw = dict(res_context_emb=bi["res_context_emb"], pair_context_emb=bi["pair_context_emb"], generation_mask=gm, seed=9, t_start=40,
                  t_stop=28)
        full = big.sample(bi["seq_idx"], bi["translations"], bi["orientations"], **kw)
        for graph in (False, True):
            lean = big.sample(bi["seq_idx"], bi["translations"], bi["orientations"], skip_unused_rows=True, graph=graph, **kw)
            for k in full:
                assert torch.equal(full[k], lean[k]), (K_, graph, k)
        assert torch.isfinite(full["translations"]).all()
        assert torch.equal(full["translations"][0], bi["translations"][0])  # (the patch without generated residues is untouched)


def test_graph_sampler_is_bitwise_the_eager_sampler(hip):
    """DIFFAB_FLAG_GRAPH_SAMPLER: one captured reverse step (timestep in device memory) replayed as a hipGraph = the eager loop, bit
    for bit, on the generic path (unit dims) and on the MFMA path at BASELINE config 1's shape (B = 1, K = 128, 100 steps)."""
    from diffab_pytorch import DiffAb

    dims, model, _ = _unit_model()
    inp = syn.patches(3, 16, dims, seed=21, coord_sigma=5.0)
    kw = dict(res_context_emb=inp["res_context_emb"], pair_context_emb=inp["pair_context_emb"], generation_mask=inp["generation_mask"], seed=77)
    eager = model.sample(inp["seq_idx"], inp["translations"], inp["orientations"], graph=False, **kw)
    graph = model.sample(inp["seq_idx"], inp["translations"], inp["orientations"], graph=True, **kw)
    for k in eager:
        assert torch.equal(eager[k], graph[k]), k
    part = model.sample(inp["seq_idx"], inp["translations"], inp["orientations"], graph=True, t_start=60, t_stop=55, init=False, **kw)
    want = model.sample(inp["seq_idx"], inp["translations"], inp["orientations"], graph=False, t_start=60, t_stop=55, init=False, **kw)
    for k in part:
        assert torch.equal(part[k], want[k]), k
    bd = dict(syn.BENCH_DIMS)
    torch.manual_seed(0)
    big = DiffAb(bd["D"], bd["C"], bd["NL"], bd["DS"], bd["PQ"], bd["PV"], bd["H"]).cuda()
    bi = {k: v.cuda() for k, v in syn.patches(1, 128, bd, seed=22).items()}
    kw = dict(res_context_emb=bi["res_context_emb"], pair_context_emb=bi["pair_context_emb"], generation_mask=bi["generation_mask"], seed=5)
    import time
    res = {}
    for mode in (False, True, False, True):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = big.sample(bi["seq_idx"], bi["translations"], bi["orientations"], graph=mode, **kw)
        torch.cuda.synchronize()
        res[mode] = (out, time.perf_counter() - t0)
    for k in res[False][0]:
        assert torch.equal(res[False][0][k], res[True][0][k]), k
    assert torch.isfinite(res[True][0]["translations"]).all()
    print("config 1 (B=1, K=128, 100 steps): eager %.1f ms, graph replay %.1f ms" % (1e3 * res[False][1], 1e3 * res[True][1]))


def test_shared_step_and_add_noise(hip):
    dims, model, sd = _unit_model()
    B, K = 4, 16
    inp = syn.patches(B, K, dims, seed=2, coord_sigma=5.0)
    t = torch.tensor([1, 30, 60, 100])
    torch.manual_seed(0)
    nz = model._add_noise(inp["seq_idx"], inp["translations"], inp["orientations"], inp["generation_mask"], t)
    assert set(nz) == {"seq_idx_t", "seq_posterior", "translations_t", "translations_eps", "orientations_t"}
    gm = inp["generation_mask"]
    assert torch.equal(nz["translations_t"][~gm], inp["translations"][~gm])
    assert torch.equal(nz["orientations_t"][~gm], inp["orientations"][~gm])
    Ot = nz["orientations_t"]
    assert torch.allclose(Ot.transpose(-1, -2) @ Ot, torch.eye(3).expand_as(Ot), atol=1e-4)
    sched = orc.cosine_variance_schedule(100, s=0.01, beta_max=0.999)
    assert maxrel(nz["translations_t"], orc.coord_diffuse_from_t0(inp["translations"], t, gm, nz["translations_eps"], sched)) < 1e-6
    batch = {"seq_idx": inp["seq_idx"], "xyz": inp["translations"], "orientations": inp["orientations"],
             "generation_mask": gm, "residue_mask": inp["residue_mask"], "res_context_emb": inp["res_context_emb"],
             "pair_context_emb": inp["pair_context_emb"]}
    torch.manual_seed(1)
    l1 = [float(x) for x in model._shared_step(batch, 0)]
    torch.manual_seed(1)
    l2 = [float(x) for x in model._shared_step(batch, 0)]
    assert l1 == l2 and all(np.isfinite(l1)) and all(v >= 0 for v in l1)
    loss = model.validation_step(batch, 0)
    assert torch.isfinite(loss) and not loss.requires_grad


def test_argument_errors_are_reported_not_crashed(hip):
    import ctypes as C

    d = _hip.make_dims(0, 16, 32, 16, 8, 12, 4, 4, 1)
    assert hip.diffab_denoise_workspace_bytes(C.byref(d)) == 0
    assert b"non-positive" in hip.diffab_last_error()
    rc = hip.diffab_so3_log(None, None, 4, None)
    assert rc == -1 and b"so3_log" in hip.diffab_last_error()
    d = _hip.make_dims(1, 16, 32, 16, 8, 12, 4, 4, 1)
    w = _hip.IpaLayerWeights()
    x = torch.zeros(16, device="cuda")
    rc = hip.diffab_ipa_layer_fwd(C.byref(d), C.byref(w), _hip.ptr(x), _hip.ptr(x), _hip.ptr(x), _hip.ptr(x), _hip.ptr(x), _hip.ptr(x), 8, 0, None)
    assert rc == -4  # workspace too small is caught before anything is launched
    assert hip.diffab_orientation_loss(_hip.ptr(x), _hip.ptr(x), 0, None, None, None) == -1


def test_full_size_properties_b256(hip):
    """BASELINE configs[1] size (B=256, K=128, benchmark model): size-independent properties."""
    from diffab_pytorch.diffab_pytorch import Denoiser

    d = syn.BENCH_DIMS
    den = Denoiser(d["D"], d["C"], d["NL"], d["DS"], d["PQ"], d["PV"], d["H"], 21)
    den.load_state_dict(syn.denoiser_state_dict(d, seed=0, prefix=""))
    den = den.cuda().requires_grad_(False)
    B, K = 256, 128
    g = torch.Generator(device="cuda").manual_seed(0)
    rc = torch.randn(B, K, d["D"], device="cuda", generator=g)
    pc = torch.randn(B, K, K, d["C"], device="cuda", generator=g)
    x = 10 * torch.randn(B, K, 3, device="cuda", generator=g)
    q = torch.nn.functional.normalize(torch.randn(B, K, 4, device="cuda", generator=g), dim=-1)
    O = orc.uniform_rotation_from_normals(q.cpu()).cuda()
    seq = torch.randint(0, 20, (B, K), device="cuda", generator=g)
    beta = torch.rand(B, device="cuda", generator=g) * 0.9 + 0.01
    out = den(seq, x, O, rc, pc, beta, None, None)
    assert all(torch.isfinite(v).all() for v in out.values())
    assert torch.allclose(out["seq_posterior"].sum(-1), torch.ones(B, K, device="cuda"), atol=1e-5)
    O0 = out["orientations_t0"]
    assert torch.allclose(O0.transpose(-1, -2) @ O0, torch.eye(3, device="cuda").expand_as(O0), atol=1e-4)
    # a sub-batch gives bitwise the same rows as the full batch (patches are independent; sharding-safe)
    sl = slice(100, 108)
    sub = den(seq[sl], x[sl], O[sl], rc[sl], pc[sl], beta[sl], None, None)
    for k in out:
        assert torch.equal(sub[k], out[k][sl]), k
    # IPA is invariant to a global rigid motion of the patch frame (translation): eps-hat changes only by fp32 noise
    shifted = den(seq[sl], x[sl] + torch.tensor([30.0, -20.0, 10.0], device="cuda"), O[sl], rc[sl], pc[sl], beta[sl], None, None)
    assert maxrel(shifted["translations_eps"], sub["translations_eps"]) < 1e-3


# ------------------------------------------------------------------ training step: gradients vs the reference's autograd
def test_hotpath_gradients_vs_reference_goldens(hip, golden):
    """Loss values and gradients of the hot-path training step (noised state -> denoise -> 3 losses, contexts as leaf inputs)
    against autograd of the REAL reference (tests/golden/losses_grads.npz, generated by oracle/gen_golden.py)."""
    from diffab_pytorch import DiffAb

    g = golden("losses_grads")
    B, K, seed = [int(v) for v in g["meta"][:3]]
    dims = dict(zip(("D", "C", "NL", "DS", "H", "PQ", "PV"), [int(v) for v in g["meta"][3:]]), V=21)
    model = DiffAb(dims["D"], dims["C"], dims["NL"], dims["DS"], dims["PQ"], dims["PV"], dims["H"]).cuda()
    model.denoiser.load_state_dict(syn.denoiser_state_dict(dims, seed=seed, prefix=""))
    inp = syn.patches(B, K, dims, seed=seed, coord_sigma=4.0)
    res_ctx = inp["res_context_emb"].cuda().requires_grad_(True)
    pair_ctx = inp["pair_context_emb"].cuda().requires_grad_(True)
    noised = {"seq_idx_t": T(g["seq_t"]).cuda(), "translations_t": T(g["x_t"]).cuda(), "orientations_t": T(g["O_t"]).cuda(),
              "seq_posterior": T(g["post"]).cuda(), "translations_eps": T(g["eps"]).cuda()}
    ls = model.hotpath_train_losses(noised, res_ctx, pair_ctx, T(g["beta"]).cuda(), inp["orientations"].cuda(), T(g["gen"]).cuda(),
                                    T(g["resm"]).cuda())
    np.testing.assert_allclose([float(x) for x in ls], g["losses"], rtol=5e-5)
    (ls[0] + ls[1] + ls[2]).backward()
    assert maxrel(res_ctx.grad, g["grad_res_ctx"]) < 2e-4, maxrel(res_ctx.grad, g["grad_res_ctx"])
    assert maxrel(pair_ctx.grad, g["grad_pair_ctx"]) < 2e-4, maxrel(pair_ctx.grad, g["grad_pair_ctx"])
    worst = ("", 0.0)
    for n, p in model.denoiser.named_parameters():
        want = g["grad/" + n]
        assert p.grad is not None and p.grad.shape == want.shape, n
        r = maxrel(p.grad, want)
        worst = max(worst, (n, r), key=lambda t_: t_[1])
        assert r < 2e-4, (n, r)
    print("worst parameter-gradient max-rel:", worst)
    # a second backward of the same batch accumulates like autograd does
    ls2 = model.hotpath_train_losses(noised, res_ctx, pair_ctx, T(g["beta"]).cuda(), inp["orientations"].cuda(), T(g["gen"]).cuda(),
                                     T(g["resm"]).cuda())
    (2.0 * ls2[1]).backward()
    assert torch.isfinite(model.denoiser.to_res_emb[0].weight.grad).all()


def test_bench_geometry_gradients_vs_reference_goldens(hip, golden):
    """The MFMA forward tape + MFMA backward kernels (benchmark geometry: D=128, C=64, H=8, DS=32, P=8, K=128, NL=2, B=2) against
    autograd of the REAL reference (tests/golden/bench_grads.npz): EVERY parameter and both contexts, norm and a strided
    subsample of <= 512 elements each, at the 2e-4 bar of the unit-dims goldens."""
    from diffab_pytorch import DiffAb

    g = golden("bench_grads")
    B, K, seed = [int(v) for v in g["meta"][:3]]
    dims = dict(zip(("D", "C", "NL", "DS", "H", "PQ", "PV"), [int(v) for v in g["meta"][3:]]), V=21)
    model = DiffAb(dims["D"], dims["C"], dims["NL"], dims["DS"], dims["PQ"], dims["PV"], dims["H"]).cuda()
    model.denoiser.load_state_dict(syn.denoiser_state_dict(dims, seed=seed, prefix=""))
    inp = syn.patches(B, K, dims, seed=seed, coord_sigma=float(g["coord_sigma"]))
    res_ctx = inp["res_context_emb"].cuda().requires_grad_(True)
    pair_ctx = inp["pair_context_emb"].cuda().requires_grad_(True)
    noised = {"seq_idx_t": T(g["seq_t"]).cuda(), "translations_t": T(g["x_t"]).cuda(), "orientations_t": T(g["O_t"]).cuda(),
              "seq_posterior": T(g["post"]).cuda(), "translations_eps": T(g["eps"]).cuda()}
    ls = model.hotpath_train_losses(noised, res_ctx, pair_ctx, T(g["beta"]).cuda(), inp["orientations"].cuda(), T(g["gen"]).cuda(),
                                    T(g["resm"]).cuda())
    np.testing.assert_allclose([float(x) for x in ls], g["losses"], rtol=5e-5)
    (ls[0] + ls[1] + ls[2]).backward()
    grads = {"res_ctx": res_ctx.grad, "pair_ctx": pair_ctx.grad}
    grads.update({n: p.grad for n, p in model.denoiser.named_parameters()})
    names = [k[4:] for k in g if k.startswith("sub/")]
    assert set(names) == set(grads), set(names) ^ set(grads)
    worst = ("", 0.0)
    for n in names:
        numel, stride, off, norm, amax = g["info/" + n]
        got = grads[n].detach().reshape(-1)
        assert got.numel() == int(numel), n
        sub = got[int(off)::int(stride)][:512].double().cpu()
        r = float((sub - T(g["sub/" + n]).double()).abs().max() / amax)  # max |a-b| / max |b| with the FULL tensor's maximum
        rn = abs(float(got.double().norm()) - norm) / norm
        worst = max(worst, (n, max(r, rn)), key=lambda t_: t_[1])
        assert r < 2e-4 and rn < 2e-4, (n, r, rn)
    print("bench-geometry worst gradient (subsample max-rel | norm rel):", worst)


def test_training_step_runs_and_learns(hip):
    """DiffAb.training_step + Adam (configure_optimizers, reference :925-931) on a fixed synthetic batch: loss decreases."""
    dims, model, _ = _unit_model()
    B, K = 4, 16
    inp = syn.patches(B, K, dims, seed=3, coord_sigma=5.0)
    batch = {"seq_idx": inp["seq_idx"].cuda(), "xyz": inp["translations"].cuda(), "orientations": inp["orientations"].cuda(),
             "generation_mask": inp["generation_mask"].cuda(), "residue_mask": inp["residue_mask"].cuda(),
             "res_context_emb": inp["res_context_emb"].cuda(), "pair_context_emb": inp["pair_context_emb"].cuda()}
    model.lr = 1e-3
    opt = model.configure_optimizers()
    first = last = None
    for it in range(12):
        torch.manual_seed(42)  # same t and noise every iteration: a fixed objective
        opt.zero_grad()
        loss = model.training_step(batch, it)
        assert loss.requires_grad and torch.isfinite(loss)
        loss.backward()
        opt.step()
        first = float(loss) if first is None else first
        last = float(loss)
    assert last < first, (first, last)
    # the fast (MFMA) forward feeds the same backward at the benchmark geometry
    from diffab_pytorch import DiffAb

    bd = dict(syn.BENCH_DIMS, NL=2)
    big = DiffAb(bd["D"], bd["C"], bd["NL"], bd["DS"], bd["PQ"], bd["PV"], bd["H"]).cuda()
    sd = syn.denoiser_state_dict(bd, seed=4, prefix="")
    big.denoiser.load_state_dict(sd)
    # (1, 128), (2, 64): both key-tile counts of the MFMA attention backward (NT = 8, 4), B = 2 crosses a patch boundary;
    # (1, 256): chunked MFMA forward + the multi-row VALU backward (no MFMA backward / tape slots beyond one key chunk)
    for Bb, Kb in ((1, 128), (2, 64), (1, 256)):
        bi = syn.patches(Bb, Kb, bd, seed=4, coord_sigma=6.0)
        rc = bi["res_context_emb"].cuda().requires_grad_(True)
        pcg = bi["pair_context_emb"].cuda().requires_grad_(True)  # exercises the d pair_ctx product of the backward stream kernel
        t = torch.tensor([40] * Bb)
        torch.manual_seed(0)
        nz = big._add_noise(bi["seq_idx"].cuda(), bi["translations"].cuda(), bi["orientations"].cuda(), bi["generation_mask"].cuda(),
                            t.cuda())
        big.zero_grad()
        ls = big.hotpath_train_losses(nz, rc, pcg, big.sched["beta"][t].cuda(), bi["orientations"].cuda(), bi["generation_mask"].cuda(),
                                      bi["residue_mask"].cuda())
        sum(ls).backward()
        # oracle autograd on the same noised state
        rco = bi["res_context_emb"].clone().requires_grad_(True)
        pco = bi["pair_context_emb"].clone().requires_grad_(True)
        sdo = {"denoiser." + k: v.clone().requires_grad_(True) for k, v in sd.items()}
        den = orc.denoiser(sdo, nz["seq_idx_t"].cpu(), nz["translations_t"].cpu(), nz["orientations_t"].cpu(), rco, pco, big.sched["beta"][t],
                           bd["NL"], bd["H"])
        lo = orc.hotpath_losses(den, nz["seq_posterior"].cpu(), nz["translations_eps"].cpu(), bi["orientations"], bi["generation_mask"],
                                bi["residue_mask"])
        sum(lo).backward()
        np.testing.assert_allclose([float(x) for x in ls], [float(x) for x in lo], rtol=1e-4)
        assert maxrel(rc.grad, rco.grad) < 5e-4, (Kb, maxrel(rc.grad, rco.grad))
        assert maxrel(pcg.grad, pco.grad) < 5e-4, (Kb, maxrel(pcg.grad, pco.grad))
        for n in ("ipa.layers.0.to_q_scalar.weight", "ipa.layers.1.gamma", "ipa.layers.0.gamma", "ipa.layers.0.to_pair_bias.weight",
                  "ipa.layers.1.to_pair_bias.weight", "to_res_emb.0.weight", "sequence_denoising.4.weight", "ipa.layers.1.to_out.weight",
                  "ipa.layers.0.to_k_point.weight", "ipa.layers.0.to_q_point.weight", "ipa.layers.1.to_v_point.weight",
                  "ipa.layers.0.to_v_scalar.weight", "ipa.layers.1.to_k_scalar.weight"):
            got = dict(big.denoiser.named_parameters())[n].grad
            assert maxrel(got, sdo["denoiser." + n].grad) < 5e-4, (Kb, n, maxrel(got, sdo["denoiser." + n].grad))


# ------------------------------------------------------------------ encode_context
def test_encode_context_vs_reference_goldens(hip, golden):
    from diffab_pytorch import DiffAb

    g = golden("encode_context")
    Bc, Kc, A_, D_, C_, seed = [int(v) for v in g["meta"]]
    model = DiffAb(D_, C_, 1, 12, 4, 4, 8).cuda()
    missing = model.load_state_dict(syn.context_state_dict(D_, C_, A_, 32, seed=seed), strict=False)
    assert not missing.unexpected_keys
    cb = syn.context_batch(Bc, Kc, A_, seed=seed)
    for gs in (True, False):
        for gq in (True, False):
            res, pair = model.encode_context(cb["seq_idx"], cb["xyz"], cb["orientations"], cb["backbone_dihedrals"], cb["distmat"],
                                             cb["pairwise_dihedrals"], cb["atom_mask"], cb["chain_idx"], cb["residue_idx"],
                                             cb["generation_mask"], cb["residue_mask"], generate_structure=gs, generate_sequence=gq)
            assert res.shape == (Bc, Kc, D_) and pair.shape == (Bc, Kc, Kc, C_)
            assert maxrel(res, g[f"res_{int(gs)}{int(gq)}"]) < 1e-5, (gs, gq, maxrel(res, g[f"res_{int(gs)}{int(gq)}"]))
            assert maxrel(pair, g[f"pair_{int(gs)}{int(gq)}"]) < 1e-5, (gs, gq, maxrel(pair, g[f"pair_{int(gs)}{int(gq)}"]))
            # distmat=None: atom-atom distances taken from xyz inside the pair kernel - same goldens, which were
            # produced by the real reference from the materialised distance tensor
            res_x, pair_x = model.encode_context(cb["seq_idx"], cb["xyz"], cb["orientations"], cb["backbone_dihedrals"], None,
                                                 cb["pairwise_dihedrals"], cb["atom_mask"], cb["chain_idx"], cb["residue_idx"],
                                                 cb["generation_mask"], cb["residue_mask"], generate_structure=gs, generate_sequence=gq)
            assert torch.equal(res_x, res)
            assert maxrel(pair_x, g[f"pair_{int(gs)}{int(gq)}"]) < 1e-5, (gs, gq, maxrel(pair_x, g[f"pair_{int(gs)}{int(gq)}"]))


def test_encode_context_gradients_vs_goldens(hip, golden):
    """Training through encode_context: every residue_context_embedding.* gradient against autograd of the REAL reference, every
    pair_context_embedding.* gradient against autograd of the oracle restatement (the reference's own PairEmbedding backward
    raises on its in-place product, diffab_pytorch.py:295-301 - asserted when the fixture is generated), for the materialised
    distance tensor and for distances taken from xyz."""
    from diffab_pytorch import DiffAb

    g = golden("encode_context_grads")
    Bc, Kc, A_, D_, C_, seed = [int(v) for v in g["meta"]]
    model = DiffAb(D_, C_, 1, 12, 4, 4, 8).cuda()
    model.load_state_dict(syn.context_state_dict(D_, C_, A_, 32, seed=seed), strict=False)
    cb = {k: v.cuda() for k, v in syn.context_batch(Bc, Kc, A_, seed=seed).items()}
    G1, G2 = T(g["G1"]).cuda(), T(g["G2"]).cuda()
    names = [k[5:] for k in g if k.startswith("grad/")]
    assert len(names) == 23 and {n.split(".")[0] for n in names} == {"residue_context_embedding", "pair_context_embedding"}
    params = dict(model.named_parameters())
    for distmat in (cb["distmat"], None):
        model.zero_grad()
        res, pair = model.encode_context(cb["seq_idx"], cb["xyz"], cb["orientations"], cb["backbone_dihedrals"], distmat,
                                         cb["pairwise_dihedrals"], cb["atom_mask"], cb["chain_idx"], cb["residue_idx"],
                                         cb["generation_mask"], cb["residue_mask"])
        assert res.requires_grad and pair.requires_grad
        ((res * G1).sum() + (pair * G2).sum()).backward()
        worst = ("", 0.0)
        for n in names:
            want = g["grad/" + n]
            got = params[n].grad
            assert got is not None and got.shape == want.shape, n
            r = maxrel(got, want)
            worst = max(worst, (n, r), key=lambda t_: t_[1])
            assert r < 2e-4, (n, r, "xyz" if distmat is None else "distmat")
        print("encode_context worst parameter-gradient max-rel (%s):" % ("xyz" if distmat is None else "distmat"), worst)
    # padding_idx of the chain embedding takes no gradient (nn.Embedding(10, D, padding_idx=0), reference :65)
    assert float(params["residue_context_embedding.chain_embedding.weight"].grad[0].abs().max()) == 0.0


def test_fused_pair_embedding_k128_vs_unfused_and_oracle(hip):
    """The one-kernel PairEmbedding forward and its folded backward (csrc/pair_embed_fused.hip: C = 64, K % 128 == 0) at the benchmark
    model, K = 128: forward against the oracle, forward and every pair_context_embedding.* gradient against the unfused HIP launches
    (which the K = 12 goldens pin to the reference), and the gradients against autograd of the oracle on one patch; distances from
    xyz and from the materialised tensor."""
    from diffab_pytorch import DiffAb

    d = syn.BENCH_DIMS
    B, K, A_ = 2, 128, 15
    model = DiffAb(d["D"], d["C"], 1, d["DS"], d["PQ"], d["PV"], d["H"]).cuda()
    sd = syn.context_state_dict(d["D"], d["C"], A_, 32, seed=11)
    sd["pair_context_embedding.pair2distcoef.weight"] = 0.3 * torch.randn(441, A_ * A_, generator=torch.Generator().manual_seed(5))
    model.load_state_dict(sd, strict=False)
    cbc = syn.context_batch(B, K, A_, seed=12)
    cb = {k: v.cuda() for k, v in cbc.items()}
    g = torch.Generator(device="cuda").manual_seed(2)
    Gp = torch.randn(B, K, K, d["C"], device="cuda", generator=g)
    names = [n for n, _ in model.named_parameters() if n.startswith("pair_context_embedding.")]
    assert len(names) == 13

    def run(distmat, variant):
        hip.diffab_debug_set_attn_variant(variant)
        try:
            model.zero_grad()
            _, pair = model.encode_context(cb["seq_idx"], cb["xyz"], cb["orientations"], cb["backbone_dihedrals"], distmat,
                                           cb["pairwise_dihedrals"], cb["atom_mask"], cb["chain_idx"], cb["residue_idx"],
                                           cb["generation_mask"], cb["residue_mask"])
            (pair * Gp).sum().backward()
            return pair.detach().clone(), {n: dict(model.named_parameters())[n].grad.detach().clone() for n in names}
        finally:
            hip.diffab_debug_set_attn_variant(0)

    for distmat in (None, cb["distmat"]):
        pf, gf = run(distmat, 0)
        pu, gu = run(distmat, 4)
        assert maxrel(pf, pu) < 2e-6, maxrel(pf, pu)
        worst = max(((n, maxrel(gf[n], gu[n])) for n in names), key=lambda t_: t_[1])
        print("fused vs unfused PairEmbedding (%s): forward %.1e, worst gradient %s %.1e" % ("xyz" if distmat is None else "distmat",
                                                                                            maxrel(pf, pu), worst[0], worst[1]))
        assert worst[1] < 1e-4, worst
        # round 6: the 64-wide tail of the backward as one launch per chunk (csrc/pair_chain_bwd.hip) against its separate launches
        # (variant 64: mask kernel, four masked d x products, the four-product weight-gradient launch) - both fp32-accurate
        ps, gs = run(distmat, 64)
        assert torch.equal(pf, ps)
        worst = max(((n, maxrel(gf[n], gs[n])) for n in names), key=lambda t_: t_[1])
        print("  one-launch chain vs separate launches: worst gradient %s %.1e" % worst)
        assert worst[1] < 2e-5, worst
        # the taped form (the default when a backward follows: diffab_pair_embedding_fwd_taped / _bwd_taped) against the recomputing one:
        # the same activations, read from the tape instead of recomputed per chunk - equal up to the order of the gradient atomics
        os.environ["DIFFAB_PAIR_TAPE"] = "0"
        try:
            pr, gr = run(distmat, 0)
        finally:
            del os.environ["DIFFAB_PAIR_TAPE"]
        assert torch.equal(pf, pr)
        worst = max(((n, maxrel(gf[n], gr[n])) for n in names), key=lambda t_: t_[1])
        print("  taped vs recomputing backward: worst gradient %s %.1e" % worst)
        assert worst[1] < 2e-6, worst
    # the oracle (and its autograd) on patch 0, distances from the materialised tensor
    csd = {k: v.clone().requires_grad_(k.startswith("pair_context_embedding.")) for k, v in sd.items()}
    b1 = {k: (v[:1] if v.shape[0] == B else v) for k, v in cbc.items()}
    _, pair_o = orc.encode_context(csd, b1, True, True)
    assert maxrel(pf[:1], pair_o.detach()) < 1e-5, maxrel(pf[:1], pair_o.detach())
    (pair_o * Gp[:1].cpu()).sum().backward()
    hip.diffab_debug_set_attn_variant(0)
    model.zero_grad()
    cb1 = {k: (v[:1].contiguous() if v.shape[0] == B else v) for k, v in cb.items()}
    _, pair1 = model.encode_context(cb1["seq_idx"], cb1["xyz"], cb1["orientations"], cb1["backbone_dihedrals"], cb1["distmat"],
                                    cb1["pairwise_dihedrals"], cb1["atom_mask"], cb1["chain_idx"], cb1["residue_idx"], cb1["generation_mask"],
                                    cb1["residue_mask"])
    (pair1 * Gp[:1]).sum().backward()
    for n in names:
        r = maxrel(dict(model.named_parameters())[n].grad, csd[n].grad)
        assert r < 2e-4, (n, r)


def test_pair_embedding_backward_k256_matrix_core_kernels_vs_separate_launches(hip):
    """K = 256 (two 128-row tiles per (patch, i) group: the second tile starts at j = 128): the round-6 backward kernels
    (csrc/pair_chain_bwd.hip: chain, one-hot table sums, coefficient gradient with d din formed in the kernel - all walk 128-row tiles of
    consecutive j) against the separate launches (variant 64) that the K = 128 test pins to the oracle's autograd, taped
    and recomputing."""
    from diffab_pytorch import DiffAb

    d = syn.BENCH_DIMS
    B, K, A_ = 1, 256, 15
    model = DiffAb(d["D"], d["C"], 1, d["DS"], d["PQ"], d["PV"], d["H"]).cuda()
    sd = syn.context_state_dict(d["D"], d["C"], A_, 32, seed=21)
    sd["pair_context_embedding.pair2distcoef.weight"] = 0.3 * torch.randn(441, A_ * A_, generator=torch.Generator().manual_seed(6))
    model.load_state_dict(sd, strict=False)
    cb = {k: v.cuda() for k, v in syn.context_batch(B, K, A_, seed=22).items()}
    Gp = torch.randn(B, K, K, d["C"], device="cuda", generator=torch.Generator(device="cuda").manual_seed(4))
    names = [n for n, _ in model.named_parameters() if n.startswith("pair_context_embedding.")]

    def run(variant, tape):
        hip.diffab_debug_set_attn_variant(variant)
        os.environ["DIFFAB_PAIR_TAPE"] = tape
        try:
            model.zero_grad()
            _, pair = model.encode_context(cb["seq_idx"], cb["xyz"], cb["orientations"], cb["backbone_dihedrals"], None, cb["pairwise_dihedrals"],
                                           cb["atom_mask"], cb["chain_idx"], cb["residue_idx"], cb["generation_mask"], cb["residue_mask"])
            (pair * Gp).sum().backward()
            return pair.detach().clone(), {n: dict(model.named_parameters())[n].grad.detach().clone() for n in names}
        finally:
            hip.diffab_debug_set_attn_variant(0)
            del os.environ["DIFFAB_PAIR_TAPE"]

    p0, g0 = run(0, "1")
    for variant, tape in ((64, "0"), (0, "0")):
        p1, g1 = run(variant, tape)
        assert torch.equal(p0, p1)
        worst = max(((n, maxrel(g0[n], g1[n])) for n in names), key=lambda t_: t_[1])
        print(f"K = 256, variant {variant}, tape {tape}: worst gradient {worst[0]} {worst[1]:.1e}")
        assert worst[1] < 2e-5, worst
    assert all(torch.isfinite(v).all() and float(v.abs().max()) > 0 for v in g0.values())


def test_full_training_step_updates_every_parameter(hip):
    """DiffAb.training_step on the reference's batch dict (no precomputed contexts): encode_context is part of the graph, so one
    Adam step moves all parameters - the 559 641 of the two context encoders included."""
    from diffab_pytorch import DiffAb

    d = syn.BENCH_DIMS
    torch.manual_seed(0)
    model = DiffAb(d["D"], d["C"], 2, d["DS"], d["PQ"], d["PV"], d["H"]).cuda()
    model.load_state_dict(syn.context_state_dict(d["D"], d["C"], 15, 32, seed=3), strict=False)
    batch = {k: v.cuda() for k, v in syn.context_batch(2, 32, 15, seed=3).items()}
    batch.pop("distmat")
    n_params = sum(p.numel() for p in model.parameters())
    before = {n: p.detach().clone() for n, p in model.named_parameters()}
    opt = model.configure_optimizers()
    torch.manual_seed(1)
    loss = model.training_step(batch, 0)
    assert loss.requires_grad and torch.isfinite(loss)
    loss.backward()
    opt.step()
    # pair2distcoef is zero-initialised upstream and takes a gradient all the same; relpos rows never indexed stay put
    moved = {n: not torch.equal(p.detach(), before[n]) for n, p in model.named_parameters()}
    still = [n for n, m in moved.items() if not m]
    assert not still, still
    enc = sum(p.numel() for n, p in model.named_parameters() if n.startswith(("residue_context_embedding", "pair_context_embedding")))
    print(f"full training step: {n_params} parameters, {enc} of them in the context encoders, all moved")


def test_featurize_xyz_vs_oracle_and_minimal_batch(hip):
    """orientations, backbone dihedrals (+ mask) and pairwise dihedrals from xyz on the device against the oracle's
    float64 statement of the same definitions (parity with protstruc itself is unpinned: it is not in the reference tree); then a
    training step and a sampling call on a batch that carries nothing but coordinates, sequence, masks and chain ids."""
    from diffab_pytorch import DiffAb, features, io

    cb = syn.context_batch(3, 64, 15, seed=11)
    rmask = cb["residue_mask"].clone()
    rmask[0, 10] = False
    f = features.featurize(cb["xyz"].cuda(), cb["chain_idx"].cuda(), rmask.cuda())
    want = orc.featurize_xyz(cb["xyz"], cb["chain_idx"], rmask)
    # fp32 differences of coordinates tens of Angstrom from the origin carry ~3e-6 relative error before the Gram-Schmidt step
    assert f["orientations"].is_cuda and maxrel(f["orientations"], want["orientations"]) < 5e-5
    Rd = f["orientations"].double().cpu()
    orth = (Rd @ Rd.transpose(-1, -2) - torch.eye(3, dtype=torch.float64)).abs().flatten(-2).max(-1).values
    # (the synthetic atoms are scattered at random: a few N-CA-C triples are nearly collinear, where Gram-Schmidt in fp32 loses digits)
    assert float(orth.median()) < 1e-6 and float(orth.max()) < 2e-4, (float(orth.median()), float(orth.max()))
    assert torch.equal(f["backbone_dihedrals_mask"].cpu(), want["backbone_dihedrals_mask"])
    for k in ("backbone_dihedrals", "pairwise_dihedrals"):  # compare on the circle: +-pi are the same angle
        d = (f[k].cpu().double() - want[k]).abs()
        d = torch.minimum(d, 2 * np.pi - d)
        assert float(d.max()) < 2e-4, (k, float(d.max()))  # fp32 atan2 of fp32 cross products: near-planar quadruples lose a few digits
        assert float(d.median()) < 1e-6, (k, float(d.median()))
    # frames are the exact inverse of the output side's reconstruction
    back = io.backbone_from_frames(cb["xyz"][:, :, 1].cuda(), f["orientations"], atoms=("N", "CA", "C"))
    n_ca = (cb["xyz"][:, :, 0] - cb["xyz"][:, :, 1]).norm(dim=-1).cuda()
    assert torch.allclose((back[:, :, 2] - back[:, :, 1]).norm(dim=-1), torch.full_like(n_ca, 1.526), atol=1e-4)
    # minimal batch: no orientations, no dihedral features, no distance tensor
    d = syn.BENCH_DIMS
    torch.manual_seed(0)
    model = DiffAb(d["D"], d["C"], 1, d["DS"], d["PQ"], d["PV"], d["H"]).cuda()
    batch = {k: cb[k].cuda() for k in ("seq_idx", "xyz", "atom_mask", "chain_idx", "residue_idx", "residue_mask", "generation_mask")}
    loss = model.training_step(batch, 0)
    assert torch.isfinite(loss) and loss.requires_grad
    out = model.sample(batch["seq_idx"], batch["xyz"], f["orientations"], generation_mask=batch["generation_mask"],
                       atom_mask=batch["atom_mask"], chain_idx=batch["chain_idx"], residue_mask=batch["residue_mask"], seed=3, t_start=100, t_stop=96)
    assert torch.isfinite(out["translations"]).all()


def test_encode_context_benchmark_dims_and_end_to_end(hip):
    """Benchmark model (D=128, C=64, A=15), K=64, vs the oracle; then the whole chain on the device: encode_context ->
    _shared_step (no precomputed contexts) -> sample."""
    from diffab_pytorch import DiffAb

    d = syn.BENCH_DIMS
    torch.manual_seed(0)
    model = DiffAb(d["D"], d["C"], 2, d["DS"], d["PQ"], d["PV"], d["H"]).cuda()
    csd = syn.context_state_dict(d["D"], d["C"], 15, 32, seed=7)
    model.load_state_dict(csd, strict=False)
    B, K = 3, 64
    cb = syn.context_batch(B, K, 15, seed=7)
    dev = {k: v.cuda() for k, v in cb.items()}
    res, pair = model.encode_context(dev["seq_idx"], dev["xyz"], dev["orientations"], dev["backbone_dihedrals"], dev["distmat"],
                                     dev["pairwise_dihedrals"], dev["atom_mask"], dev["chain_idx"], dev["residue_idx"],
                                     dev["generation_mask"], dev["residue_mask"])
    assert res.is_cuda and pair.is_cuda
    res_o, pair_o = orc.encode_context(csd, cb, True, True)
    assert maxrel(res, res_o) < 1e-5 and maxrel(pair, pair_o) < 1e-5, (maxrel(res, res_o), maxrel(pair, pair_o))
    # (B, K) residue_idx gives the same result as the broadcast (1, K) form
    dev2 = dict(dev, residue_idx=dev["residue_idx"].expand(B, K).contiguous())
    _, pair2 = model.encode_context(dev2["seq_idx"], dev2["xyz"], dev2["orientations"], dev2["backbone_dihedrals"], dev2["distmat"],
                                    dev2["pairwise_dihedrals"], dev2["atom_mask"], dev2["chain_idx"], dev2["residue_idx"],
                                    dev2["generation_mask"], dev2["residue_mask"])
    assert torch.equal(pair2, pair)
    batch = dict(dev)  # reference batch dict: no precomputed contexts
    batch.pop("distmat")  # ... and, as upstream (data.py:93-94), no distance tensor: taken from xyz on the device
    with torch.no_grad():
        torch.manual_seed(3)
        ls = model._shared_step(batch, 0)
    assert all(torch.isfinite(x) for x in ls)
    out = model.sample(dev["seq_idx"], dev["xyz"], dev["orientations"], res_context_emb=res, pair_context_emb=pair,
                       generation_mask=dev["generation_mask"], seed=5, t_start=100, t_stop=90)
    assert torch.isfinite(out["translations"]).all() and out["translations"].shape == (B, K, 3)


@pytest.mark.parametrize("Kd", [128, 1024, 1344])
def test_split_precision_gemm_is_fp32_accurate(hip, Kd):
    """The bf16x6 dense kernel (three exact bf16 pieces per operand, six exact partial products, fp32 accumulation) and the fp16x3 one
    (two fp16 pieces under power-of-two scales - one per weight row, one per (row, 64 k) of X - three partial products) against float64,
    side by side with the f32-input MFMA kernel they replace: its error must be of the same size, on operands whose
    magnitudes span six decades (the split must not care) and at the three contraction lengths of the model."""
    g = torch.Generator(device="cuda").manual_seed(Kd)
    M = 4096 + 37  # ragged last work-group
    X = torch.randn(M, Kd, device="cuda", generator=g) * torch.exp(3 * torch.randn(M, 1, device="cuda", generator=g))
    W = torch.randn(128, Kd, device="cuda", generator=g) * 0.1 * torch.exp(2 * torch.randn(128, 1, device="cuda", generator=g))
    b = torch.randn(128, device="cuda", generator=g)
    want = X.double() @ W.double().T + b.double()
    scale = (X.double().abs() @ W.double().abs().T) + b.double().abs()  # the natural error scale of a dot product: sum |x w|
    scratch = torch.empty(3 * 128 * Kd * 2 + 256, dtype=torch.uint8, device="cuda")
    errs = {}
    for mode in (0, 1, 2):
        Y = torch.empty(M, 128, device="cuda")
        rc = hip.diffab_debug_linear128(_hip.ptr(X), _hip.ptr(W), _hip.ptr(b), _hip.ptr(Y), M, Kd, mode, _hip.ptr(scratch), scratch.numel(),
                                        _hip.stream_ptr())
        assert rc == 0, hip.diffab_last_error()
        errs[mode] = float(((Y.double() - want).abs() / scale).max())
    print(f"Kd={Kd}: max |err| / sum|x w|: f32 MFMA {errs[0]:.2e}, bf16x6 {errs[1]:.2e}, fp16x3 {errs[2]:.2e}")
    assert errs[0] < 2e-6 and errs[1] < 2e-6 and errs[2] < 2e-6, errs  # all at fp32 accumulation noise (a plain bf16 product would be ~4e-3)
    assert errs[1] < 4 * errs[0] + 1e-7, errs           # and the split forms are not worse than the fp32 kernel by more than noise
    assert errs[2] < 4 * errs[0] + 4e-7, errs           # (fp16 x 3 carries 22 bits per operand: 2^-22 = 2.4e-7 on top)


@pytest.mark.parametrize("M,N", [(16384, 1024), (4096 + 37, 1024), (300, 100), (129, 1344)])
def test_x_stationary_backward_product_is_fp32_accurate(hip, M, N):
    """d feat = d y W_out of the training backward (autograd of diffab_pytorch.py:459-464; 128 -> N columns, x-stationary): the fp16 x 3
    kernel (round 6: the projection tile without frames, any N) beside the bf16 x 6 one it replaces, against float64 on rows and weight
    columns whose magnitudes span decades - gradients are small and uneven; per-row and per-column power-of-two scales must absorb it.
    Shapes: the training step's own (16 384 x 1024), ragged rows, a partial last block (N = 100: one block of 96 + 4), one row tile + 1."""
    g = torch.Generator(device="cuda").manual_seed(M + N)
    X = torch.randn(M, 128, device="cuda", generator=g) * 1e-4 * torch.exp(3 * torch.randn(M, 1, device="cuda", generator=g))
    X[7] = 0.0
    W = torch.randn(128, N, device="cuda", generator=g) * 0.1 * torch.exp(2 * torch.randn(1, N, device="cuda", generator=g))
    want = X.double() @ W.double()
    scale = X.double().abs() @ W.double().abs() + 1e-300
    nb = (N + 95) // 96
    scratch = torch.empty(3 * 2 * nb * 96 * 128 * 2 + nb * 96 * 4 + 256, dtype=torch.uint8, device="cuda")
    errs = {}
    for mode in (1, 2):
        Y = torch.full((M + 1, N), 7.0, device="cuda")  # (one guard row behind the product)
        rc = hip.diffab_debug_xstat128(_hip.ptr(X), _hip.ptr(W), _hip.ptr(Y), M, N, mode, _hip.ptr(scratch), scratch.numel(), _hip.stream_ptr())
        assert rc == 0, hip.diffab_last_error()
        assert torch.isfinite(Y).all() and bool((Y[M] == 7.0).all()) and bool((Y[7] == 0.0).all())
        errs[mode] = float(((Y[:M].double() - want).abs() / scale).max())
    print(f"M={M} N={N}: max |err| / sum|x w|: bf16x6 {errs[1]:.2e}, fp16x3 {errs[2]:.2e}")
    assert errs[1] < 2e-6 and errs[2] < 2e-6, errs
    assert errs[2] < 4 * errs[1] + 4e-7, errs


@pytest.mark.parametrize("M,N1,N2", [(16384, 1344, 128), (16384, 128, 1024), (4096 + 37, 131, 128), (777, 20, 131), (128, 3, 128)])
def test_weight_gradient_product_is_fp32_accurate(hip, M, N1, N2):
    """d W += d y^T x of every nn.Linear's backward (contraction over the B K rows): the fp16 x 3 kernel (round 6: one power-of-two scale
    per 32-row slab and operand, found from wave maxima that travel one barrier ahead of the data) beside the bf16 x 6 one, against
    float64.  Rows whose magnitudes span decades in BOTH operands and change from slab to slab (masked rows are zero, gradients are
    1e-6 .. 1e-2): the per-slab scales must follow.  Shapes: the projections' and to_out's of the training step, the 131-wide head
    input (scalar loads), the 20- and 3-wide head outputs, a ragged last slab; C accumulates (+=); db = column sums of A."""
    g = torch.Generator(device="cuda").manual_seed(M + 3 * N1 + N2)
    rowmag = torch.exp(4 * torch.randn(M, 1, device="cuda", generator=g))
    A = torch.randn(M, N1, device="cuda", generator=g) * 1e-4 * rowmag
    Bm = torch.randn(M, N2, device="cuda", generator=g) * torch.exp(2 * torch.randn(M, 1, device="cuda", generator=g))
    A[M // 3: M // 3 + 70] = 0.0  # masked rows: whole slabs of zeros
    C0 = torch.randn(N1, N2, device="cuda", generator=g) * 1e-3
    want = C0.double() + A.double().T @ Bm.double()
    scale = A.double().abs().T @ Bm.double().abs() + C0.double().abs()
    want_db = A.double().sum(0)
    errs = {}
    for mode in (1, 2):
        Cm, db = C0.clone(), torch.zeros(N1, device="cuda")
        rc = hip.diffab_debug_gemm_tn(_hip.ptr(A), _hip.ptr(Bm), _hip.ptr(Cm), _hip.ptr(db), M, N1, N2, mode, _hip.stream_ptr())
        assert rc == 0, hip.diffab_last_error()
        assert torch.isfinite(Cm).all()
        errs[mode] = float(((Cm.double() - want).abs() / scale).max())
        assert float((db.double() - want_db).abs().max()) <= 1e-5 * float(A.double().abs().sum(0).max()), mode
    print(f"M={M} N1={N1} N2={N2}: max |err| / sum|a b|: bf16x6 {errs[1]:.2e}, fp16x3 {errs[2]:.2e}")
    assert errs[1] < 2e-6 and errs[2] < 2e-6, errs
    assert errs[2] < 4 * errs[1] + 4e-7, errs


def test_fp16x3_gemm_shape_guard_and_tiny_rows(hip):
    """Round-5 advisor findings on the fp16 x 3 row GEMM: (1) it joins its 32-k chunks in pairs, so a contraction length that is not a
    multiple of 64 must be REFUSED (Kd = 96 used to drop the last chunk silently); (2) a row whose largest magnitude is a normal number
    below 2^-119 used to get an infinite scale (NaN outputs): such rows are now left unscaled, like zero rows - the answer is the fp32
    one (about 0 beside the bias)."""
    g = torch.Generator(device="cuda").manual_seed(7)
    M = 256
    scratch = torch.empty(3 * 128 * 128 * 2 + 1024, dtype=torch.uint8, device="cuda")
    X96, W96, Y = torch.randn(M, 96, device="cuda", generator=g), torch.randn(128, 96, device="cuda", generator=g), torch.empty(M, 128, device="cuda")
    b = torch.randn(128, device="cuda", generator=g)
    rc = hip.diffab_debug_linear128(_hip.ptr(X96), _hip.ptr(W96), _hip.ptr(b), _hip.ptr(Y), M, 96, 2, _hip.ptr(scratch), scratch.numel(),
                                    _hip.stream_ptr())
    assert rc != 0 and b"64" in hip.diffab_last_error()
    X = torch.randn(M, 128, device="cuda", generator=g)
    X[3] *= 1e-37   # largest magnitude ~ 3e-37: biased exponent 5..6
    X[4] *= 3e-38   # ... and at the edge of the subnormals
    X[5] = 0.0
    W = torch.randn(128, 128, device="cuda", generator=g)
    rc = hip.diffab_debug_linear128(_hip.ptr(X), _hip.ptr(W), _hip.ptr(b), _hip.ptr(Y), M, 128, 2, _hip.ptr(scratch), scratch.numel(),
                                    _hip.stream_ptr())
    assert rc == 0, hip.diffab_last_error()
    want = X.double() @ W.double().T + b.double()
    assert torch.isfinite(Y).all()
    assert float((Y.double() - want).abs().max()) < 1e-5, float((Y.double() - want).abs().max())
    assert torch.equal(Y[5], b)


@pytest.mark.parametrize("scale", [1e-6, 1.0, 3e4])
def test_pair_planes_scale_robustness(hip, scale):
    """fp16 planes of the pair embedding (DIFFAB_FLAG_PAIR_PLANES) against the fp32-pair kernel when the tensor's magnitude is far from
    one and a single outlier sets the maximum: the power-of-two rescaling keeps both fp16 pieces in range, and the bias weights shrink
    by the same factor so that the logits keep their size (otherwise the softmax would saturate and hide any error)."""
    from diffab_pytorch.diffab_pytorch import InvariantPointAttentionLayer

    d = syn.BENCH_DIMS
    torch.manual_seed(1)
    layer = InvariantPointAttentionLayer(d["D"], d["C"], d["DS"], d["PQ"], d["PV"], d["H"]).cuda().requires_grad_(False)
    inp = {k: v.cuda() for k, v in syn.patches(2, 128, d, seed=9, coord_sigma=6.0).items()}
    e = inp["pair_context_emb"] * scale
    e[0, 3, 5, 7] = 37.0 * scale  # one element far above the rest: the scale is chosen for it, everything else sits 5 binades lower
    with torch.no_grad():
        layer.to_pair_bias.weight.mul_(1.0 / scale)
    args = (inp["res_context_emb"], e, inp["orientations"], inp["translations"])
    ref = layer(*args)
    got = layer(*args, flags=_hip.FLAG_PAIR_PLANES)
    assert torch.isfinite(got).all()
    assert maxrel(got, ref) < 2e-6, (scale, maxrel(got, ref))


def test_pair_planes_outlier_stays_in_its_row(hip):
    """A trained PairEmbedding has no norm layer: one element 1e4 x the rest must not cost the other rows their precision.  The planes
    carry one power-of-two scale per PAIR ROW (b, i) (csrc/denoiser_fast.hip, pair_rowscale_kernel): rows that do not contain the
    outlier keep the full 2^-23 of their own maximum.  Compared with the fp32-pair kernel on the same inputs: every output row of a
    query residue whose pair row is outlier-free agrees to 2e-6 of the output maximum; the row that holds the outlier (query 3 of patch
    0: its attention is one-hot on the outlier's key for the heads with a non-zero bias weight, in both kernels) is held to the 1e-4
    bar."""
    from diffab_pytorch.diffab_pytorch import InvariantPointAttentionLayer

    d = syn.BENCH_DIMS
    torch.manual_seed(1)
    layer = InvariantPointAttentionLayer(d["D"], d["C"], d["DS"], d["PQ"], d["PV"], d["H"]).cuda().requires_grad_(False)
    inp = {k: v.cuda() for k, v in syn.patches(2, 128, d, seed=9, coord_sigma=6.0).items()}
    e = inp["pair_context_emb"].clone()
    e[0, 3, 5, 7] = 1.0e4
    args = (inp["res_context_emb"], e, inp["orientations"], inp["translations"])
    ref = layer(*args)
    got = layer(*args, flags=_hip.FLAG_PAIR_PLANES)
    assert torch.isfinite(got).all()
    clean = torch.ones(2, 128, dtype=torch.bool, device=got.device)
    clean[0, 3] = False
    scale = float(ref.abs().max())
    err_clean = float((got - ref)[clean].abs().max()) / scale
    err_row = float((got - ref)[~clean].abs().max()) / scale
    print(f"pair planes with a 1e4 x outlier: outlier-free rows {err_clean:.2e}, the outlier's row {err_row:.2e} (of the output maximum)")
    assert err_clean < 2e-6, err_clean
    assert err_row < TOL, err_row

