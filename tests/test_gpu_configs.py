"""BASELINE.json configs at their STATED sizes on one MI355X (the 8-GPU legs of configs 3 / 4 run their per-GPU share here):
size-independent properties of the domain - finiteness, context preservation, orthonormal frames, determinism and bitwise
invariance under sharding of the batch (what makes the N > 1 runs correct by construction) - on the MFMA path.

  config 2: B = 256, K = 128, 100 reverse steps (T = 100)            -> test_config2_...
  config 3: B = 2048 over 8 GPUs = 256 per GPU, SAbDab-shaped RAW batch (15 atoms, chain ids, no distmat, no contexts):
            encode_context + 100 steps at the per-GPU share                 -> test_config3_...
  config 4: training step, B = 1024 over 8 GPUs = 128 per GPU, NL = 6 -> test_config4_...
  config 5: B = 512, K = 256, 200 reverse steps on the T = 200 schedule -> test_config5_...
"""
import time

import numpy as np
import pytest
import torch

import diffab_oracle as orc
from diffab_pytorch import _hip, synthetic as syn

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    lib = _hip.lib()
    assert lib.diffab_device_ok() == 1
    return lib


def device_patches(B, K, dims, seed):
    """Seeded synthetic patches of SURVEY 8(d)'s shapes, generated on the device (the 8.6 GB pair context of config 5 would take
    minutes through the host generator): N(0,1) contexts, N(0,10^2) A translations, uniform rotations, one CDR-like segment
    of 5..20 generated residues per patch."""
    g = torch.Generator(device="cuda").manual_seed(seed)
    out = {
        "res_context_emb": torch.randn(B, K, dims["D"], device="cuda", generator=g),
        "pair_context_emb": torch.randn(B, K, K, dims["C"], device="cuda", generator=g),
        "translations": 10 * torch.randn(B, K, 3, device="cuda", generator=g),
        "seq_idx": torch.randint(0, 20, (B, K), device="cuda", generator=g),
    }
    q = torch.randn(B, K, 4, device="cuda", generator=g)
    out["orientations"] = orc.uniform_rotation_from_normals(q.cpu()).cuda()
    start = torch.randint(0, K - 20, (B, 1), device="cuda", generator=g)
    length = torch.randint(5, 21, (B, 1), device="cuda", generator=g)
    pos = torch.arange(K, device="cuda")[None]
    out["generation_mask"] = (pos >= start) & (pos < start + length)
    return out


def bench_model(T_steps, NL=None):
    from diffab_pytorch import DiffAb

    d = dict(syn.BENCH_DIMS)
    if NL is not None:
        d["NL"] = NL
    torch.manual_seed(0)
    model = DiffAb(d["D"], d["C"], d["NL"], d["DS"], d["PQ"], d["PV"], d["H"], T=T_steps).cuda()
    model.denoiser.load_state_dict(syn.denoiser_state_dict(d, seed=0, prefix=""))
    return d, model


def check_trajectory(model, inp, seed, shard, n_steps_expected, kw=None):
    """Full reverse trajectory + the properties listed in the module docstring; `shard` = (lo, hi) patches re-run alone.
    kw(slice) -> the keyword arguments of DiffAb.sample for those patches (default: precomputed contexts)."""
    if kw is None:
        kw = lambda sl: dict(res_context_emb=inp["res_context_emb"][sl], pair_context_emb=inp["pair_context_emb"][sl],
                             generation_mask=inp["generation_mask"][sl])
    all_ = slice(None)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    full = model.sample(inp["seq_idx"], inp["translations"], inp["orientations"], seed=seed, **kw(all_))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    B, K = inp["seq_idx"].shape
    assert model.T == n_steps_expected
    gm = inp["generation_mask"]
    assert torch.isfinite(full["translations"]).all() and torch.isfinite(full["orientations"]).all()
    x_ctx = inp["translations"] if inp["translations"].dim() == 3 else inp["translations"][:, :, 1]
    assert torch.equal(full["translations"][~gm], x_ctx[~gm])  # context residues are never touched
    assert torch.equal(full["orientations"][~gm], inp["orientations"][~gm])
    assert torch.equal(full["seq_idx"][~gm], inp["seq_idx"][~gm])
    assert ((full["seq_idx"] >= 0) & (full["seq_idx"] < 21)).all()
    x_in = inp["translations"] if inp["translations"].dim() == 3 else inp["translations"][:, :, 1]  # CA of an all-atom batch
    assert not torch.equal(full["translations"][gm], x_in[gm])
    Og = full["orientations"][gm]
    err = (Og.transpose(-1, -2) @ Og - torch.eye(3, device=Og.device)).abs().max()
    assert err < 1e-3, float(err)  # every step re-derives O from exp maps: stays a rotation over the whole trajectory
    lo, hi = shard
    sl = slice(lo, hi)
    part = model.sample(inp["seq_idx"][sl], inp["translations"][sl], inp["orientations"][sl], seed=seed, first_patch=lo, **kw(sl))
    for k in full:  # bitwise: a rank that owns patches [lo, hi) of the global batch produces exactly these rows
        assert torch.equal(part[k], full[k][sl]), k
    again = model.sample(inp["seq_idx"][sl], inp["translations"][sl], inp["orientations"][sl], seed=seed, first_patch=lo, **kw(sl))
    for k in full:
        assert torch.equal(again[k], part[k]), k
    other = model.sample(inp["seq_idx"][sl], inp["translations"][sl], inp["orientations"][sl], seed=seed + 1, first_patch=lo, **kw(sl))
    assert not torch.equal(other["translations"], part["translations"])
    return B * K * model.T / dt, dt


def oracle_reverse_step(model, inp, sl, seed, t):
    """The oracle's reverse step t -> t-1 for the patches `sl` (a slice) of a batch, with the Philox noise of their GLOBAL patch ids."""
    sd = {"denoiser." + k: v.detach().cpu() for k, v in model.denoiser.state_dict().items()}
    sched = orc.cosine_variance_schedule(model.T, s=0.01, beta_max=0.999)
    c = {k: v[sl].cpu() for k, v in inp.items()}
    B, K = c["seq_idx"].shape
    patch = (sl.start + np.arange(B))[:, None] + np.zeros((B, K), dtype=np.int64)
    res = np.zeros((B, K), dtype=np.int64) + np.arange(K)[None, :]
    z = torch.from_numpy(np.stack(orc.philox_normal4(seed, patch, res, t, orc.STREAM_TRANS)[:3], -1))
    ax = torch.from_numpy(np.stack(orc.philox_normal4(seed, patch, res, t, orc.STREAM_AXIS)[:3], -1))
    ua = orc.philox_uniform4(seed, patch, res, t, orc.STREAM_ANGLE)
    na = orc.philox_normal4(seed, patch, res, t, orc.STREAM_ANGLE)
    us = torch.from_numpy(orc.philox_uniform4(seed, patch, res, t, orc.STREAM_SEQ)[0])
    sig = sched["beta"].sqrt()
    cdf_row = model._reverse_so3()._cdf[t].cpu()[None, None, :].expand(B, K, -1)
    th_h = orc.igso3_theta_from_hist(orc.igso3_bin_from_cdf(cdf_row, torch.from_numpy(ua[0])), torch.from_numpy(ua[1]))
    th_g = orc.igso3_theta_from_gaussian(sig[t].expand(B, K), torch.from_numpy(na[2]))
    rotvec = orc.igso3_rotvec(ax, th_h, th_g, sig[t].expand(B))
    den = orc.denoiser(sd, c["seq_idx"], c["translations"], c["orientations"], c["res_context_emb"], c["pair_context_emb"],
                       sched["beta"][t].expand(B), model.denoiser.dims["NL"], model.denoiser.dims["H"])
    # distance of every sequence draw's uniform from the nearest edge of the oracle posterior's CDF (a draw can flip only on an edge)
    edge = (den["seq_posterior"].double().cumsum(-1) - us.double()[..., None]).abs().min(dim=-1).values
    return orc.reverse_update(t, c["seq_idx"], c["translations"], c["orientations"], den, c["generation_mask"], sched, z, rotvec, us) + (edge,)


def test_patch_resident_module_is_bitwise_the_per_layer_launches(hip):
    """DIFFAB_FLAG_PERSISTENT_MODULE (one work-group owns a patch through the six layers of the IPA module, the form diffab_sample_loop
    chooses by itself when the batch fills the chip) against the 3 NL launches per step it replaces: same tile bodies, so the samples
    are equal bit for bit - forced on at a small batch (8 work-groups), chosen automatically at B = 256 against DIFFAB_FLAG_MULTI_LAUNCH,
    with more patches than CUs (a work-group walks two patches), eager and graph-replayed."""
    dims, model = bench_model(100)
    for B, fl_a, fl_b, steps in ((8, _hip.FLAG_PERSISTENT_MODULE, 0, 6), (256, 0, _hip.FLAG_MULTI_LAUNCH, 4),
                                 (300, _hip.FLAG_PERSISTENT_MODULE, _hip.FLAG_MULTI_LAUNCH, 2)):
        inp = device_patches(B, 128, dims, seed=40 + B)
        kw = dict(res_context_emb=inp["res_context_emb"], pair_context_emb=inp["pair_context_emb"], generation_mask=inp["generation_mask"],
                  seed=5, t_stop=100 - steps)
        a = model.sample(inp["seq_idx"], inp["translations"], inp["orientations"], flags=fl_a, **kw)
        b = model.sample(inp["seq_idx"], inp["translations"], inp["orientations"], flags=fl_b, **kw)
        for k in a:
            assert torch.equal(a[k], b[k]), (B, k)
        assert not torch.equal(a["translations"], inp["translations"])
        if B == 8:
            g = model.sample(inp["seq_idx"], inp["translations"], inp["orientations"], flags=fl_a, graph=True, **kw)
            for k in a:
                assert torch.equal(a[k], g[k]), ("graph", k)
        if B == 256:  # and equal to itself, run after run (how the packed-fp32 hazard of profiles/r05_pk_opsel_hazard.md showed: a
            for rep in range(4):  # handful of patches per step differed between two runs of the SAME launch)
                a2 = model.sample(inp["seq_idx"], inp["translations"], inp["orientations"], flags=fl_a, **kw)
                for k in a:
                    assert torch.equal(a[k], a2[k]), ("run-to-run", rep, k)
    # K = 256 (round 6: two dense tiles and sixteen two-chunk attention items per patch inside the same launch; BASELINE config 5)
    inp = device_patches(6, 256, dims, seed=46)
    kw = dict(res_context_emb=inp["res_context_emb"], pair_context_emb=inp["pair_context_emb"], generation_mask=inp["generation_mask"],
              seed=5, t_stop=97)
    a = model.sample(inp["seq_idx"], inp["translations"], inp["orientations"], flags=_hip.FLAG_PERSISTENT_MODULE, **kw)
    b = model.sample(inp["seq_idx"], inp["translations"], inp["orientations"], flags=0, **kw)
    for k in a:
        assert torch.equal(a[k], b[k]), ("K = 256", k)
    assert not torch.equal(a["translations"], inp["translations"])
    # the opt-in value-plane form (diffab_debug_set_attn_variant(16), profiles/r06_attention.md) shares its tile bodies between the two launch
    # forms as well: bitwise equal to each other (and different bits from the default form: other arithmetic in phase 3)
    inp = device_patches(8, 128, dims, seed=48)
    kw = dict(res_context_emb=inp["res_context_emb"], pair_context_emb=inp["pair_context_emb"], generation_mask=inp["generation_mask"],
              seed=5, t_stop=97)
    d0 = model.sample(inp["seq_idx"], inp["translations"], inp["orientations"], flags=0, **kw)
    try:
        hip.diffab_debug_set_attn_variant(16)
        a = model.sample(inp["seq_idx"], inp["translations"], inp["orientations"], flags=_hip.FLAG_PERSISTENT_MODULE, **kw)
        b = model.sample(inp["seq_idx"], inp["translations"], inp["orientations"], flags=0, **kw)
    finally:
        hip.diffab_debug_set_attn_variant(0)
    for k in a:
        assert torch.equal(a[k], b[k]), ("value planes", k)
    assert not torch.equal(a["translations"], d0["translations"])
    assert float((a["translations"] - d0["translations"]).abs().max()) < 1e-2  # three reverse steps of the same noise: the same trajectory
    del inp
    torch.cuda.empty_cache()


def test_ipa_layer_bits_do_not_depend_on_the_launch_form(hip):
    """One IPA layer on the first patches of batches of 1, 8, 40, 128 and 256 patches: the dense products pick a launch form by batch
    size - (row tile, 64-k part) work-groups, 64-row groups, 128-row groups, projections split over column blocks - and all of them
    have to produce the same bits for the same patch (the sampler's shard invariance rests on it)."""
    from diffab_pytorch.diffab_pytorch import InvariantPointAttentionLayer

    d, K = syn.BENCH_DIMS, 128
    torch.manual_seed(0)
    layer = InvariantPointAttentionLayer(d["D"], d["C"], d["DS"], d["PQ"], d["PV"], d["H"]).cuda().requires_grad_(False)
    inp = device_patches(256, K, d, seed=77)
    outs = {}
    for B in (1, 8, 40, 128, 256):
        args = [inp[k][:B].contiguous() for k in ("res_context_emb", "pair_context_emb", "orientations", "translations")]
        outs[B] = layer(*args, flags=_hip.FLAG_PAIR_PLANES)[: min(B, 8)].clone()
    for B, o in outs.items():
        assert torch.isfinite(o).all()
        assert torch.equal(o, outs[256][: o.shape[0]]), (B, int((o != outs[256][: o.shape[0]]).sum()))


def test_config2_b256_k128_100_steps(hip):
    dims, model = bench_model(100)
    inp = device_patches(256, 128, dims, seed=2)
    rate, dt = check_trajectory(model, inp, seed=11, shard=(64, 72), n_steps_expected=100)
    print(f"config 2: B=256 K=128 x 100 steps in {dt:.3f} s (incl. launch + sync) = {rate / 1e6:.2f} M residue-steps/s")
    # DIRECT parity at the instantiations this batch size selects (128-row dense tiles, the non-SPLIT projection kernel - forms that
    # smaller batches never run): one teacher-forced reverse step of all 256 patches, an 8-patch slice of it against the oracle
    from conftest import maxrel

    for t in (57, 3):
        got = model.sample(inp["seq_idx"], inp["translations"], inp["orientations"], res_context_emb=inp["res_context_emb"],
                           pair_context_emb=inp["pair_context_emb"], generation_mask=inp["generation_mask"], seed=29, t_start=t, t_stop=t - 1,
                           init=False)
        sl = slice(120, 128)
        s1, x1, O1, edge = oracle_reverse_step(model, inp, sl, 29, t)
        assert maxrel(got["translations"][sl], x1) < 1e-4, (t, maxrel(got["translations"][sl], x1))
        assert maxrel(got["orientations"][sl], O1) < 1e-4, (t, maxrel(got["orientations"][sl], O1))
        diff = got["seq_idx"][sl].cpu() != s1  # every flipped draw: the oracle's uniform within 1e-5 of a cumulative-probability edge
        if diff.any():
            assert float(edge[diff].max()) < 1e-5, (t, int(diff.sum()), float(edge[diff].max()))


def test_config5_b512_k256_200_steps_T200(hip):
    dims, model = bench_model(200)
    assert model.sched["beta"].numel() == 201 and model._reverse_so3().histograms.shape == (201, 8192)
    inp = device_patches(512, 256, dims, seed=5)
    rate, dt = check_trajectory(model, inp, seed=12, shard=(300, 304), n_steps_expected=200)
    print(f"config 5: B=512 K=256 x 200 steps (T=200 schedule) in {dt:.3f} s = {rate / 1e6:.2f} M residue-steps/s")


def test_config4_training_step_b128_nl6(hip):
    """Per-GPU share of config 4 (1024 patches over 8 GPUs): noise + taped forward + 3 losses + HIP backward + Adam at
    B = 128, K = 128, NL = 6.  Finite loss, a finite non-zero gradient on every denoiser parameter, parameters move."""
    dims, model = bench_model(100)
    inp = device_patches(128, 128, dims, seed=4)
    batch = {"seq_idx": inp["seq_idx"], "xyz": inp["translations"], "orientations": inp["orientations"],
             "generation_mask": inp["generation_mask"], "residue_mask": torch.ones_like(inp["generation_mask"]),
             "res_context_emb": inp["res_context_emb"], "pair_context_emb": inp["pair_context_emb"]}
    opt = model.configure_optimizers()
    before = {n: p.detach().clone() for n, p in model.denoiser.named_parameters()}
    times = []
    for it in range(3):
        torch.manual_seed(100 + it)
        opt.zero_grad(set_to_none=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        loss = model.training_step(batch, it)
        loss.backward()
        opt.step()
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
        assert torch.isfinite(loss)
    for n, p in model.denoiser.named_parameters():
        assert p.grad is not None and torch.isfinite(p.grad).all() and float(p.grad.abs().max()) > 0, n
        assert not torch.equal(p.detach(), before[n]), n
    print(f"config 4 (per-GPU share): B=128 K=128 NL=6 training step {1e3 * min(times):.2f} ms = "
          f"{128 * 128 / min(times) / 1e6:.2f} M residue-steps/s")
    # GRADIENT parity at this size and depth (NL = 6, the 128-patch forms of every backward kernel): the loss masks are restricted to ONE
    # patch of the batch, so the batch loss and every gradient are that patch's alone and the oracle's autograd on that single patch (CPU,
    # seconds) is the whole answer: contexts as leaves, both context rows of the patch and three parameters, < 2e-4 of the tensor maximum;
    # the other 127 patches' context gradients must be exactly zero
    p = 77
    model.load_state_dict({"denoiser." + k: v for k, v in before.items()}, strict=False)
    gen1 = torch.zeros_like(inp["generation_mask"])
    gen1[p] = inp["generation_mask"][p]
    resm = torch.ones_like(gen1)
    t = torch.full((128,), 40)
    torch.manual_seed(9)
    noised = model._add_noise(inp["seq_idx"], inp["translations"], inp["orientations"], gen1, t.cuda())
    res_ctx = inp["res_context_emb"].clone().requires_grad_(True)
    pair_ctx = inp["pair_context_emb"].clone().requires_grad_(True)
    model.zero_grad(set_to_none=True)
    beta = model.sched["beta"][t].cuda()
    ls = model.hotpath_train_losses(noised, res_ctx, pair_ctx, beta, inp["orientations"], gen1, resm)
    (ls[0] + ls[1] + ls[2]).backward()
    sl = slice(p, p + 1)
    c = lambda v: v[sl].detach().cpu()
    o_res, o_pair = c(inp["res_context_emb"]).requires_grad_(True), c(inp["pair_context_emb"]).requires_grad_(True)
    names = ["to_res_emb.0.weight", "ipa.layers.5.to_out.weight", "coordinate_denoising.4.weight"]
    sd = {"denoiser." + k: v.detach().cpu() for k, v in before.items()}
    for n in names:
        sd["denoiser." + n] = sd["denoiser." + n].clone().requires_grad_(True)
    den = orc.denoiser(sd, c(noised["seq_idx_t"]), c(noised["translations_t"]), c(noised["orientations_t"]), o_res, o_pair,
                       model.sched["beta"][t[sl]], 6, dims["H"])
    ol = orc.hotpath_losses(den, c(noised["seq_posterior"]), c(noised["translations_eps"]), c(inp["orientations"]), c(gen1), c(resm))
    for a_, b_ in zip(ls, ol):
        assert abs(float(a_) - float(b_)) < 5e-5 * max(1.0, abs(float(b_))), (float(a_), float(b_))
    (ol[0] + ol[1] + ol[2]).backward()
    from conftest import maxrel

    worst = {"res_ctx": maxrel(res_ctx.grad[sl], o_res.grad), "pair_ctx": maxrel(pair_ctx.grad[sl], o_pair.grad)}
    worst.update({n: maxrel(dict(model.denoiser.named_parameters())[n].grad, sd["denoiser." + n].grad) for n in names})
    print("config 4 gradients at B=128, NL=6 vs the oracle's autograd on patch 77:", {k: f"{v:.1e}" for k, v in worst.items()})
    assert max(worst.values()) < 2e-4, worst
    others = torch.ones(128, dtype=torch.bool, device="cuda")
    others[p] = False
    assert float(res_ctx.grad[others].abs().max()) == 0.0 and float(pair_ctx.grad[others].abs().max()) == 0.0


def test_config3_share_raw_sabdab_batch_b256_k128(hip):
    """Per-GPU share of config 3 (2048 SAbDab-shaped patches over 8 GPUs): the RAW batch of the reference's data module - all-atom
    xyz (A = 15), chain ids in {1, 2, 3}, atom mask, no distance tensor, no precomputed contexts (data.py:82-96) - at B = 256,
    K = 128: DiffAb.sample runs encode_context itself (features from xyz on the device), then 100 reverse steps.  Trajectory
    properties as for config 2, shard invariance included (what the 8-GPU all-gather relies on), and encode_context at K = 128
    against the oracle on a 4-patch slice."""
    from conftest import maxrel

    dims, model = bench_model(100)
    csd = syn.context_state_dict(dims["D"], dims["C"], 15, 32, seed=3)
    model.load_state_dict(csd, strict=False)
    B, K = 256, 128
    cb = syn.context_batch(B, K, 15, seed=3, with_distmat=False)
    dev = {k: v.cuda() for k, v in cb.items() if k != "distmat"}
    inp = {"seq_idx": dev["seq_idx"], "translations": dev["xyz"], "orientations": dev["orientations"], "generation_mask": dev["generation_mask"]}
    kw = lambda sl: dict(generation_mask=dev["generation_mask"][sl], atom_mask=dev["atom_mask"][sl], chain_idx=dev["chain_idx"][sl],
                         residue_mask=dev["residue_mask"][sl])
    rate, dt = check_trajectory(model, inp, seed=13, shard=(96, 100), n_steps_expected=100, kw=kw)
    print(f"config 3 (per-GPU share): raw batch B=256 K=128 -> encode_context + 100 steps in {dt:.3f} s = {rate / 1e6:.2f} M residue-steps/s")
    # encode_context at K = 128 against the oracle (4 patches; the oracle takes the distance tensor, built here on the host)
    sl = slice(8, 12)
    xyz4 = cb["xyz"][sl].double()
    dist4 = (xyz4[:, :, None, :, None, :] - xyz4[:, None, :, None, :, :]).norm(dim=-1).float()
    feats = __import__("diffab_pytorch").features.featurize(dev["xyz"][sl], dev["chain_idx"][sl], dev["residue_mask"][sl], orientations=False)
    b4 = {k: (v if k == "residue_idx" else v[sl]) for k, v in cb.items() if k != "distmat"}
    b4["distmat"] = dist4
    b4["backbone_dihedrals"], b4["pairwise_dihedrals"] = feats["backbone_dihedrals"].cpu(), feats["pairwise_dihedrals"].cpu()
    with torch.no_grad():
        res, pair = model.encode_context(dev["seq_idx"][sl], dev["xyz"][sl], dev["orientations"][sl], feats["backbone_dihedrals"], None,
                                         feats["pairwise_dihedrals"], dev["atom_mask"][sl], dev["chain_idx"][sl], dev["residue_idx"],
                                         dev["generation_mask"][sl], dev["residue_mask"][sl])
    res_o, pair_o = orc.encode_context(csd, b4, True, True)
    assert maxrel(res, res_o) < 2e-5 and maxrel(pair, pair_o) < 2e-5, (maxrel(res, res_o), maxrel(pair, pair_o))


@pytest.mark.parametrize("guard", [0, 1])
def test_two_sampler_pipelines_on_two_streams_are_bitwise_the_sequential_runs(hip, guard):
    """Two reverse-sampling pipelines enqueued on two streams at the same time (200 steps each: the T = 200 schedule) against the same two
    calls one after the other: bitwise equal, and the denoise step repeated while the other stream is kept busy is bitwise the solo step.
    Rounds 3-5 saw rows of O_t exp(v) differ here (wrong values in lanes 48-63 of heads_finish_kernel / reverse_update_philox_kernel while
    bf16 x 6 GEMM work-groups of the other pipeline were resident) and serialised the library's calls across streams (csrc/common.h
    StreamOrder).  Round 6 (profiles/r06_lanes_48_63.md): the cause was the packed-fp32 form v_pk_*_f32 op_sel:[0,1] that hipcc's SLP
    vectoriser had put into those kernels - gfx950 miscomputes it while ANOTHER wave's f16 / bf16 MFMAs are in flight on the SIMD
    (tools/hwtests/pkmul_two_streams.hip) - and the form is gone from the library (tests/test_isa_lint.py).  guard = 0 is the library's
    default now and the real test: overlapping pipelines give the sequential bits; guard = 1 keeps the opt-in ordering working."""
    assert hip.diffab_set_stream_guard(guard) == 0
    dims, model = bench_model(200)
    B = 64
    inp = device_patches(2 * B, 128, dims, seed=21)
    halves = [slice(0, B), slice(B, 2 * B)]

    def run(sl, lo):
        return model.sample(inp["seq_idx"][sl], inp["translations"][sl], inp["orientations"][sl], seed=5, first_patch=lo,
                            res_context_emb=inp["res_context_emb"][sl], pair_context_emb=inp["pair_context_emb"][sl],
                            generation_mask=inp["generation_mask"][sl])

    torch.cuda.synchronize()
    seq = []
    for i, sl in enumerate(halves):  # one after the other, default stream
        seq.append(run(sl, i * B))
        torch.cuda.synchronize()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    for rep in range(6 if guard == 0 else 2):
        con = [None, None]
        with torch.cuda.stream(s1):
            con[0] = run(halves[0], 0)
        with torch.cuda.stream(s2):
            con[1] = run(halves[1], B)
        torch.cuda.synchronize()
        for i in range(2):
            for k in seq[i]:
                n_bad = int((con[i][k] != seq[i][k]).sum())
                assert n_bad == 0, f"pipeline {i} repetition {rep}: {n_bad} elements of {k} differ from the sequential run"
    # the single denoise step (heads_finish_kernel was where the wrong rows appeared), repeated while the other stream runs the same work
    args = lambda sl: (inp["seq_idx"][sl], inp["translations"][sl], inp["orientations"][sl], inp["res_context_emb"][sl],
                       inp["pair_context_emb"][sl], torch.full((B,), 0.01, device="cuda"), inp["generation_mask"][sl],
                       torch.ones(B, 128, dtype=torch.bool, device="cuda"))
    with torch.no_grad():
        ref = {k: v.clone() for k, v in model.denoise(*args(halves[0])).items()}
        torch.cuda.synchronize()
        bad = torch.zeros(3, dtype=torch.int64, device="cuda")
        for rep in range(150):
            with torch.cuda.stream(s2):
                model.denoise(*args(halves[1]))
            with torch.cuda.stream(s1):
                y = model.denoise(*args(halves[0]))
                for j, k in enumerate(ref):
                    bad[j] += (y[k] != ref[k]).sum()
        torch.cuda.synchronize()
    assert hip.diffab_set_stream_guard(0) == 0
    assert int(bad.sum()) == 0, dict(zip(ref, bad.tolist()))
