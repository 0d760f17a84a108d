"""Golden-vector generator (BUILD CONTAINER ONLY - needs /root/reference).

Imports the real reference (dohlee/diffab-pytorch) with the recipe of SURVEY.md
Appendix B.1 - stand-ins for the two absent third-party modules
(`pytorch_lightning`, `protstruc.general`) are registered in sys.modules, the
reference's own sources run unmodified - then

  1. runs every hot-path function of the reference on seeded synthetic inputs,
  2. asserts that the oracle restatement (oracle/diffab_oracle.py) reproduces it,
  3. writes inputs-by-seed + expected outputs as small fixtures under
     tests/golden/ (data only - no reference source text).

Run from anywhere:  python oracle/gen_golden.py
The reference writes `.cache/so3_histograms/` into the CWD, so the script
chdirs to a scratch directory first and sets PYTHONDONTWRITEBYTECODE.
"""
from __future__ import annotations

import enum
import importlib.util
import os
import sys
import tempfile
import types

os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
sys.dont_write_bytecode = True

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
GOLD = os.path.join(REPO, "tests", "golden")
REF = "/root/reference"


def _load(path, name):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def install_standins():
    pl = types.ModuleType("pytorch_lightning")

    class LightningModule(nn.Module):
        def log_dict(self, *a, **k):
            pass

        def log(self, *a, **k):
            pass

    class LightningDataModule:
        pass

    pl.LightningModule = LightningModule
    pl.LightningDataModule = LightningDataModule
    sys.modules["pytorch_lightning"] = pl
    ps = types.ModuleType("protstruc")
    psg = types.ModuleType("protstruc.general")

    class ATOM(enum.IntEnum):
        N = 0
        CA = 1
        C = 2
        O = 3
        CB = 4

    class AA(enum.IntEnum):
        UNK = 20

    psg.ATOM, psg.AA = ATOM, AA
    ps.general = psg
    ps.StructureBatch = None
    ps.AntibodyStructureBatch = None
    sys.modules["protstruc"] = ps
    sys.modules["protstruc.general"] = psg


def maxrel(a, b):
    a, b = a.double(), b.double()
    return float(((a - b).abs().max() / b.abs().max().clamp_min(1e-30)))


def check(name, got, want, tol):
    r = maxrel(got, want)
    flag = "ok " if r <= tol else "BAD"
    print(f"  [{flag}] oracle vs reference {name:40s} max-rel {r:.2e} (tol {tol:.0e})")
    if r > tol:
        raise SystemExit(f"oracle restatement disagrees with the reference on {name}")


def npf(t):
    return t.detach().cpu().numpy()


def main():
    scratch = tempfile.mkdtemp(prefix="diffab_gold_")
    os.chdir(scratch)
    install_standins()
    sys.path.insert(0, REF)
    import diffab_pytorch.diffab_pytorch as rmod  # the REAL reference
    import diffab_pytorch.diffusion as rdiff
    import diffab_pytorch.so3 as rso3

    assert rmod.__file__.startswith(REF)
    orc = _load(os.path.join(HERE, "diffab_oracle.py"), "diffab_oracle")
    syn = _load(os.path.join(REPO, "diffab-pytorch_amd", "diffab_pytorch", "synthetic.py"), "diffab_synthetic")
    os.makedirs(GOLD, exist_ok=True)
    torch.set_num_threads(8)

    # ---------------------------------------------------------------- schedule
    print("schedule")
    out = {}
    for T, s in ((100, 0.01), (200, 0.01), (100, 8e-3)):
        ref = rdiff.cosine_variance_schedule(T, s=s, beta_max=0.999)
        mine = orc.cosine_variance_schedule(T, s=s, beta_max=0.999)
        for k in ref:
            check(f"sched T={T} s={s} {k}", mine[k], ref[k], 0.0)
            out[f"T{T}_s{s}_{k}"] = npf(ref[k])
    np.savez_compressed(os.path.join(GOLD, "schedule.npz"), **out)
    sched = rdiff.cosine_variance_schedule(100, s=0.01, beta_max=0.999)

    # ---------------------------------------------------------------- so3 free functions
    print("so3")
    rng = np.random.Generator(np.random.PCG64([7, 1]))
    Bq, Lq = 4, 25
    R = torch.from_numpy(syn.random_rotations(rng, Bq * Lq).astype(np.float32)).view(Bq, Lq, 3, 3)
    kk = torch.from_numpy(rng.random(Bq).astype(np.float32))
    v = torch.from_numpy((rng.standard_normal((Bq, Lq, 3)) * 0.8).astype(np.float32))
    ref = {
        "R": R, "k": kk, "v": v,
        "log": rso3.log_rotmat(R),
        "rotvec": rso3.rotation_matrix_to_vector(R),
        "explog": rso3.exp_skew_symmetric_mat(rso3.log_rotmat(R)),
        "scaled": rso3.scale_rot(R, kk),
        "hat": rso3.vector_to_skew_symmetric_mat(v),
        "expv": rso3.vector_to_rotation_matrix(v),
        "trace": rso3.tensor_trace(R),
    }
    check("log_rotmat", orc.log_so3(R), ref["log"], 1e-6)
    check("rotation_matrix_to_vector", orc.matrix_to_rotvec(R), ref["rotvec"], 1e-6)
    check("exp(log R)", orc.exp_so3(orc.log_so3(R)), ref["explog"], 1e-6)
    check("scale_rot", orc.scale_rot(R, kk), ref["scaled"], 1e-6)
    check("hat", orc.hat(v), ref["hat"], 0.0)
    check("vector_to_rotation_matrix", orc.rotvec_to_matrix(v), ref["expv"], 1e-6)
    np.savez_compressed(os.path.join(GOLD, "so3.npz"), **{k: npf(x) for k, x in ref.items()})

    # ---------------------------------------------------------------- sequence diffuser
    print("sequence diffuser")
    sdiff = rdiff.SequenceDiffuser(T=100, s=0.01, beta_max=0.999)
    Bs, Ls = 6, 40
    seq0 = torch.from_numpy(rng.integers(0, 20, (Bs, Ls)))
    seqt = torch.from_numpy(rng.integers(0, 21, (Bs, Ls)))
    t = torch.tensor([1, 2, 10, 50, 99, 100])
    mask = torch.from_numpy(rng.random((Bs, Ls)) < 0.5)
    seqt = torch.where(mask, seqt, seq0)  # un-generated residues keep s_0 (as diffuse_from_t0 guarantees)
    ref = {
        "seq0": seq0, "seqt": seqt, "t": t, "mask": mask,
        "single": sdiff.forward_prob_single_step(seqt, t, mask),
        "from_t0": sdiff.forward_prob_from_t0(seq0, t, mask),
        "posterior": sdiff.posterior_single_step(seqt, seq0, t, mask),
    }
    osched = orc.cosine_variance_schedule(100, s=0.01, beta_max=0.999)
    check("forward_prob_single_step", orc.seq_forward_prob_single_step(seqt, t, mask, osched), ref["single"], 1e-7)
    check("forward_prob_from_t0", orc.seq_forward_prob_from_t0(seq0, t, mask, osched), ref["from_t0"], 1e-7)
    check("posterior_single_step", orc.seq_posterior_single_step(seqt, seq0, t, mask, osched), ref["posterior"], 1e-6)
    # weighted_multinomial (diffusion.py:38-41) with an int64 one-hot as p1, as its two call sites pass it
    wm_p1 = torch.nn.functional.one_hot(seq0, 21)
    rng_wm = np.random.Generator(np.random.PCG64([7, 2]))  # its own stream: the fixtures below keep their round-1 draws
    wm_p2 = torch.from_numpy(rng_wm.random((Bs, Ls, 21)).astype(np.float32))
    wm_w1, wm_w2 = torch.from_numpy(rng_wm.random(Bs).astype(np.float32)), torch.from_numpy(rng_wm.random(Bs).astype(np.float32))
    ref.update(wm_p2=wm_p2, wm_w1=wm_w1, wm_w2=wm_w2, wm_out=rdiff.weighted_multinomial(wm_p1, wm_p2, wm_w1, wm_w2))
    check("weighted_multinomial", orc.weighted_multinomial(wm_p1, wm_p2, wm_w1, wm_w2), ref["wm_out"], 0.0)
    np.savez_compressed(os.path.join(GOLD, "seqdiff.npz"), **{k: npf(x) for k, x in ref.items()})

    # ---------------------------------------------------------------- coordinate diffuser
    print("coordinate diffuser")
    cdiff = rdiff.CoordinateDiffuser(T=100, s=0.01, beta_max=0.999)
    x0 = torch.from_numpy((10 * rng.standard_normal((Bs, Ls, 3))).astype(np.float32))
    torch.manual_seed(1234)
    xt, eps = cdiff.diffuse_from_t0(x0, t, mask, return_eps=True)
    check("coord diffuse_from_t0", orc.coord_diffuse_from_t0(x0, t, mask, eps, osched), xt, 1e-7)
    np.savez_compressed(os.path.join(GOLD, "coorddiff.npz"), x0=npf(x0), t=npf(t), mask=npf(mask), eps=npf(eps), xt=npf(xt))

    # ---------------------------------------------------------------- IGSO3 table + orientation diffuser
    print("IGSO3 table / orientation diffuser (reference builds 101x8192 table, ~3 s)")
    odiff = rdiff.OrientationDiffuser(T=100, s=0.01, beta_max=0.999)
    table = odiff.so3.histograms  # (101, 8192)
    sig = odiff.sched["one_minus_alpha_bar_sqrt"]
    rows = [0, 1, 2, 3, 4, 5, 6, 7, 50, 100]
    mine_rows = orc.igso3_table(sig[rows], 8192, 1024)
    for i, r_ in enumerate(rows):
        check(f"igso3 pdf row {r_}", mine_rows[i], table[r_], 1e-6 if r_ else 1e-4)
    gold = {
        "sigmas": npf(sig),
        "rows": np.array(rows),
        "probe_full": npf(table[rows]),  # 10 x 8192 fp32 = 320 KB raw (compresses poorly) -> subsample below
    }
    gold["probe_every16"] = gold.pop("probe_full")[:, ::16]
    gold["row_sums"] = npf(table.double().sum(-1))
    gold["row_argmax"] = npf(table.argmax(-1))
    gold["row_nonzero"] = npf((table > 0).sum(-1))
    cdf = table.double().cumsum(-1)
    cdf = cdf / cdf[:, -1:]
    gold["cdf_every512"] = npf(cdf[:, 511::512])
    # reference sampler with captured draws (RNG order: so3.py:114, :78, :83, :93)
    tt = torch.tensor([1, 3, 5, 6, 40, 100])
    Ks = 12
    torch.manual_seed(99)
    rotvec_ref = odiff.so3.sample_isotropic_gaussian(tt, Ks)
    torch.manual_seed(99)
    axis_raw = torch.randn(len(tt), Ks, 3)
    bin_idx = torch.multinomial(table[tt], num_samples=Ks)
    u_bin = torch.rand(bin_idx.shape)
    z_g = torch.randn(len(tt), Ks)
    th_h = orc.igso3_theta_from_hist(bin_idx, u_bin)
    th_g = orc.igso3_theta_from_gaussian(sig[tt][:, None].expand(-1, Ks), z_g)
    check("sample_isotropic_gaussian (captured draws)", orc.igso3_rotvec(axis_raw, th_h, th_g, sig[tt]), rotvec_ref, 1e-6)
    gold.update(samp_t=npf(tt), samp_axis_raw=npf(axis_raw), samp_bin=npf(bin_idx), samp_u=npf(u_bin), samp_z=npf(z_g),
                samp_rotvec=npf(rotvec_ref))
    # orientation diffuser end-to-end with the same captured rot-vector
    Bo, Lo = len(tt), Ks
    O0 = torch.from_numpy(syn.random_rotations(rng, Bo * Lo).astype(np.float32)).view(Bo, Lo, 3, 3)
    omask = torch.from_numpy(rng.random((Bo, Lo)) < 0.6)
    torch.manual_seed(99)
    Ot_ref = odiff.diffuse_from_t0(O0, omask, tt)
    check("orientation diffuse_from_t0", orc.orient_diffuse_from_t0(O0, omask, tt, rotvec_ref, osched), Ot_ref, 2e-6)
    gold.update(od_O0=npf(O0), od_mask=npf(omask), od_Ot=npf(Ot_ref))
    np.savez_compressed(os.path.join(GOLD, "igso3.npz"), **gold)

    # ---------------------------------------------------------------- IPA layer / module / denoiser
    def build_ref_denoiser(dims, seed):
        sd = syn.denoiser_state_dict(dims, seed=seed)
        den = rmod.Denoiser(dims["D"], dims["C"], dims["NL"], dims["DS"], dims["PQ"], dims["PV"], dims["H"], aa_vocab_size=21)
        den.load_state_dict({k[len("denoiser."):]: v for k, v in sd.items()}, strict=True)
        return den.eval(), sd

    def run_denoiser(den, inp, beta):
        cap = {}
        hk = den.sequence_denoising[4].register_forward_hook(lambda m, i, o: cap.__setitem__("logits", o.detach()))
        hk2 = den.ipa.register_forward_hook(lambda m, i, o: cap.__setitem__("res_emb", o.detach()))
        with torch.no_grad():
            out = den(inp["seq_idx"], inp["translations"], inp["orientations"], inp["res_context_emb"],
                      inp["pair_context_emb"], beta, inp["generation_mask"], inp["residue_mask"])
        hk.remove()
        hk2.remove()
        out = dict(out)
        out["aa_logits"] = cap["logits"]
        out["res_emb"] = cap["res_emb"]
        return out

    cases = [
        # name, dims, B, K, synth seed, coord sigma
        ("unit_wide", dict(syn.UNIT_DIMS, NL=2), 3, 16, 11, 10.0),
        ("unit_tight", dict(syn.UNIT_DIMS, NL=2), 3, 16, 12, 1.0),
        ("unit_ragged", dict(syn.UNIT_DIMS, NL=1, DS=16), 2, 19, 13, 3.0),  # K not a multiple of 16
        ("bench_wide", dict(syn.BENCH_DIMS), 1, 128, 21, 10.0),
        ("bench_tight", dict(syn.BENCH_DIMS, NL=2), 1, 128, 22, 1.5),
        ("bench_k256", dict(syn.BENCH_DIMS, NL=1), 1, 256, 23, 12.0),
    ]
    for name, dims, B, K, seed, sigma in cases:
        print(f"denoiser case {name}: dims={dims} B={B} K={K}")
        den, sd = build_ref_denoiser(dims, seed)
        inp = syn.patches(B, K, dims, seed=seed, coord_sigma=sigma)
        beta = sched["beta"][torch.tensor([(7 * (i + 1)) % 100 + 1 for i in range(B)])]
        ref = run_denoiser(den, inp, beta)
        mine = orc.denoiser(sd, inp["seq_idx"], inp["translations"], inp["orientations"], inp["res_context_emb"],
                            inp["pair_context_emb"], beta, dims["NL"], dims["H"])
        for k_ in ("translations_eps", "orientations_t0", "seq_posterior", "aa_logits", "res_emb"):
            check(f"{name} {k_}", mine[k_], ref[k_], 5e-5)
        # one IPA layer alone (layer 0) on the raw res ctx
        with torch.no_grad():
            l0 = den.ipa.layers[0](inp["res_context_emb"], inp["pair_context_emb"], inp["orientations"], inp["translations"])
        check(f"{name} ipa layer0", orc.ipa_layer(inp["res_context_emb"], inp["pair_context_emb"], inp["orientations"],
                                                  inp["translations"], sd, "denoiser.ipa.layers.0.", dims["H"]), l0, 5e-5)
        g = {k_: npf(v_) for k_, v_ in ref.items()}
        g["ipa_layer0"] = npf(l0)
        g["beta"] = npf(beta)
        g["meta"] = np.array([B, K, seed, dims["D"], dims["C"], dims["NL"], dims["DS"], dims["H"], dims["PQ"], dims["PV"]])
        g["coord_sigma"] = np.array(sigma)
        g["input_checksum"] = np.array([float(inp[k_].double().sum()) for k_ in
                                        ("res_context_emb", "pair_context_emb", "translations", "orientations")])
        g["weight_checksum"] = np.array(float(sum(v_.double().sum() for v_ in sd.values())))
        np.savez_compressed(os.path.join(GOLD, f"denoiser_{name}.npz"), **g)

    # ---------------------------------------------------------------- losses (+ hot-path gradients)
    print("losses + hot-path gradients (unit dims)")
    dims = dict(syn.UNIT_DIMS, NL=2)
    B, K, seed = 3, 16, 31
    den, sd = build_ref_denoiser(dims, seed)
    den.train()
    inp = syn.patches(B, K, dims, seed=seed, coord_sigma=4.0)
    tt = torch.tensor([3, 40, 97])
    beta = sched["beta"][tt]
    gen = inp["generation_mask"]
    resm = inp["residue_mask"].clone()
    resm[0, :3] = False  # make residue_mask matter
    gen[0, :5] = True
    # noised state from the reference diffusers with captured draws
    torch.manual_seed(5)
    seq_t, post = sdiff.diffuse_from_t0(inp["seq_idx"], tt, gen, return_posterior=True)
    x_t, eps = cdiff.diffuse_from_t0(inp["translations"], tt, gen, return_eps=True)
    O_t = odiff.diffuse_from_t0(inp["orientations"], gen, tt)
    res_ctx = inp["res_context_emb"].clone().requires_grad_(True)
    pair_ctx = inp["pair_context_emb"].clone().requires_grad_(True)
    out = den(seq_t, x_t, O_t, res_ctx, pair_ctx, beta, gen, resm)
    kl = nn.KLDivLoss(reduction="none")(out["seq_posterior"].log(), post)
    mse = nn.MSELoss(reduction="none")(out["translations_eps"], eps)
    ol = rmod.OrientationLoss(reduction="none")(out["orientations_t0"], inp["orientations"])
    lm = gen & resm
    denom = lm.sum()
    l_seq = (kl * lm[..., None]).sum() / denom
    l_x = (mse * lm[..., None]).sum() / denom
    l_o = (ol * lm[..., None, None]).sum() / denom
    (l_seq + l_x + l_o).backward()
    mine = orc.denoiser(sd, seq_t, x_t, O_t, inp["res_context_emb"], inp["pair_context_emb"], beta, dims["NL"], dims["H"])
    ml = orc.hotpath_losses(mine, post, eps, inp["orientations"], gen, resm)
    check("loss seq", ml[0], l_seq.detach(), 2e-5)
    check("loss translations", ml[1], l_x.detach(), 2e-5)
    check("loss orientations", ml[2], l_o.detach(), 2e-5)
    g = dict(
        meta=np.array([B, K, seed, dims["D"], dims["C"], dims["NL"], dims["DS"], dims["H"], dims["PQ"], dims["PV"]]),
        t=npf(tt), beta=npf(beta), gen=npf(gen), resm=npf(resm), seq_t=npf(seq_t), post=npf(post), x_t=npf(x_t), eps=npf(eps),
        O_t=npf(O_t), losses=np.array([float(l_seq), float(l_x), float(l_o)]),
        out_eps=npf(out["translations_eps"]), out_O0=npf(out["orientations_t0"]), out_post=npf(out["seq_posterior"]),
        grad_res_ctx=npf(res_ctx.grad), grad_pair_ctx=npf(pair_ctx.grad),
    )
    for n_, p_ in den.named_parameters():
        g["grad/" + n_] = npf(p_.grad)
    np.savez_compressed(os.path.join(GOLD, "losses_grads.npz"), **g)

    # ---------------------------------------------------------------- hot-path gradients at the benchmark geometry (MFMA kernels)
    # Autograd of the REAL reference at D=128, C=64, H=8, DS=32, P=8, K=128, NL=2, B=2.  The full gradients are 5 MB (pair context
    # alone 8 MB), so the fixture holds, for every parameter and both contexts: the L2 norm, the max |g|, and a strided subsample
    # (<= 512 elements, stride and offset stored), which a wrong kernel cannot match by accident.
    print("hot-path gradients at the benchmark geometry (K=128, NL=2, B=2; reference autograd, ~1 min)")
    dims = dict(syn.BENCH_DIMS, NL=2)
    B, K, seed = 2, 128, 33
    den, sd = build_ref_denoiser(dims, seed)
    den.train()
    inp = syn.patches(B, K, dims, seed=seed, coord_sigma=6.0)
    tt = torch.tensor([12, 71])
    beta = sched["beta"][tt]
    gen, resm = inp["generation_mask"].clone(), inp["residue_mask"].clone()
    resm[1, :2] = False
    gen[1, :4] = True
    torch.manual_seed(6)
    seq_t, post = sdiff.diffuse_from_t0(inp["seq_idx"], tt, gen, return_posterior=True)
    x_t, eps = cdiff.diffuse_from_t0(inp["translations"], tt, gen, return_eps=True)
    O_t = odiff.diffuse_from_t0(inp["orientations"], gen, tt)
    res_ctx = inp["res_context_emb"].clone().requires_grad_(True)
    pair_ctx = inp["pair_context_emb"].clone().requires_grad_(True)
    out = den(seq_t, x_t, O_t, res_ctx, pair_ctx, beta, gen, resm)
    kl = nn.KLDivLoss(reduction="none")(out["seq_posterior"].log(), post)
    mse = nn.MSELoss(reduction="none")(out["translations_eps"], eps)
    ol = rmod.OrientationLoss(reduction="none")(out["orientations_t0"], inp["orientations"])
    lm = gen & resm
    denom = lm.sum()
    l_seq = (kl * lm[..., None]).sum() / denom
    l_x = (mse * lm[..., None]).sum() / denom
    l_o = (ol * lm[..., None, None]).sum() / denom
    (l_seq + l_x + l_o).backward()
    mine = orc.denoiser(sd, seq_t, x_t, O_t, inp["res_context_emb"], inp["pair_context_emb"], beta, dims["NL"], dims["H"])
    ml = orc.hotpath_losses(mine, post, eps, inp["orientations"], gen, resm)
    check("bench-geometry loss seq", ml[0], l_seq.detach(), 2e-5)
    check("bench-geometry loss translations", ml[1], l_x.detach(), 2e-5)
    check("bench-geometry loss orientations", ml[2], l_o.detach(), 2e-5)
    g = dict(
        meta=np.array([B, K, seed, dims["D"], dims["C"], dims["NL"], dims["DS"], dims["H"], dims["PQ"], dims["PV"]]),
        coord_sigma=np.array(6.0), t=npf(tt), beta=npf(beta), gen=npf(gen), resm=npf(resm), seq_t=npf(seq_t), post=npf(post),
        x_t=npf(x_t), eps=npf(eps), O_t=npf(O_t), losses=np.array([float(l_seq), float(l_x), float(l_o)]),
    )

    def put_grad(name, gr):
        flat = gr.detach().reshape(-1)
        n = flat.numel()
        stride = max(1, n // 512)
        off = (7 * len(name)) % stride
        g["sub/" + name] = npf(flat[off::stride][:512])
        g["info/" + name] = np.array([n, stride, off, float(flat.double().norm()), float(flat.abs().max())])

    put_grad("res_ctx", res_ctx.grad)
    put_grad("pair_ctx", pair_ctx.grad)
    for n_, p_ in den.named_parameters():
        put_grad(n_, p_.grad)
    np.savez_compressed(os.path.join(GOLD, "bench_grads.npz"), **g)

    # ---------------------------------------------------------------- encode_context through the real DiffAb
    print("encode_context (ResidueEmbedding + PairEmbedding), 4 flag combinations")
    D_, C_, A_, Kc, Bc = 32, 16, 15, 12, 2
    ref_model = rmod.DiffAb(D_, C_, 1, 12, 4, 4, 8).eval()
    csd = syn.context_state_dict(D_, C_, A_, 32, seed=41)
    missing = ref_model.load_state_dict(csd, strict=False)
    assert not missing.unexpected_keys and all(k.startswith("denoiser.") for k in missing.missing_keys)
    cb = syn.context_batch(Bc, Kc, A_, seed=41)
    g = {"meta": np.array([Bc, Kc, A_, D_, C_, 41])}
    for gs in (True, False):
        for gq in (True, False):
            with torch.no_grad():
                res_ref, pair_ref = ref_model.encode_context(cb["seq_idx"], cb["xyz"], cb["orientations"], cb["backbone_dihedrals"],
                                                             cb["distmat"].clone(), cb["pairwise_dihedrals"], cb["atom_mask"], cb["chain_idx"],
                                                             cb["residue_idx"], cb["generation_mask"], cb["residue_mask"],
                                                             generate_structure=gs, generate_sequence=gq)
            res_o, pair_o = orc.encode_context(csd, cb, gs, gq)
            check(f"encode_context res  gs={gs} gq={gq}", res_o, res_ref, 2e-6)
            check(f"encode_context pair gs={gs} gq={gq}", pair_o, pair_ref, 2e-6)
            g[f"res_{int(gs)}{int(gq)}"] = npf(res_ref)
            g[f"pair_{int(gs)}{int(gq)}"] = npf(pair_ref)
    np.savez_compressed(os.path.join(GOLD, "encode_context.npz"), **g)

    # ---------------------------------------------------------------- encode_context gradients (training through the encoders)
    # loss = <res_ctx, G1> + <pair_ctx, G2> with fixed upstream gradients.  ResidueEmbedding: autograd of the REAL reference.
    # PairEmbedding: the reference's own backward raises (in-place product on a tensor saved for backward, diffab_pytorch.py:295-301 -
    # checked below), so its gradients come from autograd of the oracle restatement, whose forward equals the reference's (above).
    print("encode_context gradients (residue: reference autograd; pair: oracle autograd, the reference's raises)")
    rng_g = np.random.Generator(np.random.PCG64([7, 3]))
    G1 = torch.from_numpy(rng_g.standard_normal((Bc, Kc, D_)).astype(np.float32))
    G2 = torch.from_numpy(rng_g.standard_normal((Bc, Kc, Kc, C_)).astype(np.float32))
    ctx_mask = cb["residue_mask"].bool() & (~cb["generation_mask"].bool())
    ref_model.train()
    ref_model.zero_grad()
    res_ref = ref_model.residue_context_embedding(cb["seq_idx"], cb["xyz"], cb["orientations"], cb["backbone_dihedrals"], cb["chain_idx"],
                                                  cb["atom_mask"], ctx_mask, ctx_mask)
    (res_ref * G1).sum().backward()
    gg = {"G1": npf(G1), "G2": npf(G2), "meta": np.array([Bc, Kc, A_, D_, C_, 41])}
    for n_, p_ in ref_model.residue_context_embedding.named_parameters():
        gg["grad/residue_context_embedding." + n_] = npf(p_.grad)
    raised = False
    try:
        pr = ref_model.pair_context_embedding(cb["seq_idx"], cb["distmat"].clone(), cb["pairwise_dihedrals"], cb["residue_idx"],
                                              cb["chain_idx"], cb["atom_mask"], ctx_mask, ctx_mask)
        (pr * G2).sum().backward()
    except RuntimeError as err:
        raised = "inplace" in str(err) or "in-place" in str(err)
    assert raised, "the reference's PairEmbedding backward was expected to fail on its in-place product"
    sdo = {k_: v_.clone().requires_grad_(v_.is_floating_point()) for k_, v_ in csd.items()}
    res_o, pair_o = orc.encode_context(sdo, cb, True, True)
    ((res_o * G1).sum() + (pair_o * G2).sum()).backward()
    for n_, p_ in ref_model.residue_context_embedding.named_parameters():
        check("grad residue_context_embedding." + n_, sdo["residue_context_embedding." + n_].grad, p_.grad, 2e-5)
    for k_, v_ in sdo.items():
        if k_.startswith("pair_context_embedding."):
            gg["grad/" + k_] = npf(v_.grad)
    np.savez_compressed(os.path.join(GOLD, "encode_context_grads.npz"), **gg)

    # ---------------------------------------------------------------- frames, AngularEncoding (module-level functions of the hot path)
    def subsample(name, gr, store, full):
        flat = gr.detach().reshape(-1)
        if full or flat.numel() <= 4096:
            store["grad/" + name] = npf(gr)
            return
        n = flat.numel()
        stride = max(1, n // 512)
        off = (7 * len(name)) % stride
        store["sub/" + name] = npf(flat[off::stride][:512])
        store["info/" + name] = np.array([n, stride, off, float(flat.double().norm()), float(flat.abs().max())])

    print("euclidean_transform / inverse_euclidean_transform / AngularEncoding")
    gf = torch.Generator().manual_seed(51)
    xpts = 5.0 * torch.randn(2, 3, 5, 4, 3, generator=gf)
    Rf = torch.from_numpy(syn.random_rotations(np.random.Generator(np.random.PCG64(52)), 10).astype(np.float32)).view(2, 5, 3, 3)
    tf = 20.0 * torch.randn(2, 5, 3, generator=gf)
    fwd = rmod.euclidean_transform(xpts, Rf, tf)
    inv = rmod.inverse_euclidean_transform(xpts, Rf, tf)
    check("euclidean_transform", orc.to_global(xpts, Rf, tf), fwd, 1e-6)
    check("inverse_euclidean_transform", orc.to_local(xpts, Rf, tf), inv, 1e-6)
    check("inverse o forward", rmod.inverse_euclidean_transform(fwd, Rf, tf), xpts, 2e-6)
    xa = 3.0 * torch.randn(2, 5, 3, generator=gf)
    enc = rmod.AngularEncoding(num_funcs=3)(xa)
    check("AngularEncoding", orc.angular_encoding(xa, 3), enc, 0.0)
    # autograd of the two transforms with respect to the points
    xg = xpts.clone().requires_grad_(True)
    cg = torch.randn(2, 3, 5, 4, 3, generator=gf)
    (rmod.euclidean_transform(xg, Rf, tf) * cg).sum().backward()
    g_fwd = xg.grad.clone()
    xg.grad = None
    (rmod.inverse_euclidean_transform(xg, Rf, tf) * cg).sum().backward()
    # ... and with respect to the frames (the einsums are differentiable in r and t as well)
    fr = {}
    for key, fn in (("fwd", rmod.euclidean_transform), ("inv", rmod.inverse_euclidean_transform)):
        Rg, tg = Rf.clone().requires_grad_(True), tf.clone().requires_grad_(True)
        (fn(xpts, Rg, tg) * cg).sum().backward()
        fr[f"grad_{key}_R"], fr[f"grad_{key}_t"] = npf(Rg.grad), npf(tg.grad)
    np.savez_compressed(os.path.join(GOLD, "frames.npz"), x=npf(xpts), R=npf(Rf), t=npf(tf), fwd=npf(fwd), inv=npf(inv), xa=npf(xa), enc=npf(enc),
                        cot=npf(cg), grad_fwd=npf(g_fwd), grad_inv=npf(xg.grad), **fr)

    print("InvariantPointAttentionLayer(use_pair_bias=False): forward + autograd (reference :348-385, :422-459)")
    for tag, dims, B, K, seed, sigma in (("unit", dict(syn.UNIT_DIMS), 2, 16, 81, 4.0), ("bench", dict(syn.BENCH_DIMS), 1, 64, 82, 6.0)):
        torch.manual_seed(seed)
        lay = rmod.InvariantPointAttentionLayer(dims["D"], dims["C"], dims["DS"], dims["PQ"], dims["PV"], dims["H"], use_pair_bias=False)
        with torch.no_grad():
            lay.gamma.copy_(torch.rand(dims["H"]) + 0.2)
        assert not hasattr(lay, "to_pair_bias")
        inp = syn.patches(B, K, dims, seed=seed, coord_sigma=sigma)
        xl = inp["res_context_emb"].clone().requires_grad_(True)
        Rl = inp["orientations"].clone().requires_grad_(True)
        tl = inp["translations"].clone().requires_grad_(True)
        yl = lay(xl, inp["pair_context_emb"], Rl, tl)
        sd_l = {k: v.detach().clone() for k, v in lay.state_dict().items()}
        check(f"ipa_layer no pair bias ({tag})", orc.ipa_layer(xl.detach(), inp["pair_context_emb"], Rl.detach(), tl.detach(), sd_l, "", dims["H"],
                                                            use_pair_bias=False), yl.detach(), 2e-6)
        c_y = torch.randn(B, K, dims["D"], generator=torch.Generator().manual_seed(seed))
        (yl * c_y).sum().backward()
        gl = dict(meta=np.array([B, K, seed, dims["D"], dims["C"], dims["DS"], dims["H"], dims["PQ"], dims["PV"]]), coord_sigma=np.array(sigma),
                  y=npf(yl), c_y=npf(c_y))  # weights: torch.manual_seed(seed) construction + the gamma draw, as above
        subsample("x", xl.grad, gl, tag == "unit")
        subsample("R", Rl.grad, gl, True)
        subsample("t", tl.grad, gl, True)
        for n_, p_ in lay.named_parameters():
            subsample(n_, p_.grad, gl, tag == "unit")
        np.savez_compressed(os.path.join(GOLD, f"ipa_layer_no_pair_bias_{tag}.npz"), **gl)

    # ---------------------------------------------------------------- autograd through the module forwards from arbitrary cotangents
    # (Denoiser.forward :558-607, InvariantPointAttentionLayer.forward :389-465, OrientationLoss :610-625 are differentiable upstream;
    # a caller with a loss of their own on model.denoise() needs these gradients)
    for tag, dims, B, K, seed, sigma in (("unit", dict(syn.UNIT_DIMS, NL=2), 2, 16, 61, 4.0), ("bench", dict(syn.BENCH_DIMS, NL=2), 1, 128, 62, 6.0)):
        print(f"module autograd from cotangents ({tag} dims, reference autograd)")
        full = tag == "unit"
        den, sd = build_ref_denoiser(dims, seed)
        den.train()
        inp = syn.patches(B, K, dims, seed=seed, coord_sigma=sigma)
        tt = torch.tensor([9, 63][:B])
        beta = sched["beta"][tt]
        gc = torch.Generator().manual_seed(seed)
        c_eps = torch.randn(B, K, 3, generator=gc)
        c_O0 = torch.randn(B, K, 3, 3, generator=gc)
        c_post = torch.randn(B, K, 21, generator=gc)
        res_ctx = inp["res_context_emb"].clone().requires_grad_(True)
        pair_ctx = inp["pair_context_emb"].clone().requires_grad_(True)
        # the frames are differentiable inputs of the reference's forward too (euclidean_transform :315-336, O_t @ exp(v) :594-596)
        x_t = inp["translations"].clone().requires_grad_(True)
        O_t = inp["orientations"].clone().requires_grad_(True)
        out = den(inp["seq_idx"], x_t, O_t, res_ctx, pair_ctx, beta, None, None)
        ((out["translations_eps"] * c_eps).sum() + (out["orientations_t0"] * c_O0).sum() + (out["seq_posterior"] * c_post).sum()).backward()
        g = dict(meta=np.array([B, K, seed, dims["D"], dims["C"], dims["NL"], dims["DS"], dims["H"], dims["PQ"], dims["PV"]]),
                 coord_sigma=np.array(sigma), beta=npf(beta), c_eps=npf(c_eps), c_O0=npf(c_O0), c_post=npf(c_post),
                 out_eps=npf(out["translations_eps"]), out_post=npf(out["seq_posterior"]))
        subsample("res_ctx", res_ctx.grad, g, full)
        subsample("pair_ctx", pair_ctx.grad, g, full)
        subsample("x_t", x_t.grad, g, True)
        subsample("O_t", O_t.grad, g, True)
        for n_, p_ in den.named_parameters():
            subsample(n_, p_.grad, g, full)
        # one IPA layer alone: d y random -> d x, d e, parameter gradients
        layer = den.ipa.layers[0]
        for p_ in layer.parameters():
            p_.grad = None
        xl = inp["res_context_emb"].clone().requires_grad_(True)
        el = inp["pair_context_emb"].clone().requires_grad_(True)
        c_y = torch.randn(B, K, dims["D"], generator=gc)
        Rl = inp["orientations"].clone().requires_grad_(True)
        tl = inp["translations"].clone().requires_grad_(True)
        yl = layer(xl, el, Rl, tl)
        (yl * c_y).sum().backward()
        g["layer/c_y"] = npf(c_y)
        g["layer/y"] = npf(yl)
        subsample("layer/x", xl.grad, g, full)
        subsample("layer/e", el.grad, g, full)
        subsample("layer/R", Rl.grad, g, True)
        subsample("layer/t", tl.grad, g, True)
        for n_, p_ in layer.named_parameters():
            subsample("layer/" + n_, p_.grad, g, full)
        np.savez_compressed(os.path.join(GOLD, f"module_autograd_{tag}.npz"), **g)

    print("OrientationLoss autograd")
    go = torch.Generator().manual_seed(71)
    Rp = torch.from_numpy(syn.random_rotations(np.random.Generator(np.random.PCG64(72)), 12).astype(np.float32)).view(3, 4, 3, 3)
    Rp = Rp + 0.05 * torch.randn(3, 4, 3, 3, generator=go)  # a prediction is not exactly a rotation
    Rt = torch.from_numpy(syn.random_rotations(np.random.Generator(np.random.PCG64(73)), 12).astype(np.float32)).view(3, 4, 3, 3)
    ol = {}
    for red in ("mean", "sum", "none"):
        pr, tr = Rp.clone().requires_grad_(True), Rt.clone().requires_grad_(True)
        val = rmod.OrientationLoss(reduction=red)(pr, tr)
        cot = torch.randn(val.shape, generator=go) if red == "none" else torch.tensor(1.7)
        (val * cot).sum().backward()
        ol[f"{red}/value"], ol[f"{red}/cot"], ol[f"{red}/d_pred"], ol[f"{red}/d_target"] = npf(val), npf(cot), npf(pr.grad), npf(tr.grad)
    np.savez_compressed(os.path.join(GOLD, "orientation_loss_grads.npz"), pred=npf(Rp), target=npf(Rt), **ol)

    tot = sum(os.path.getsize(os.path.join(GOLD, f)) for f in os.listdir(GOLD))
    print(f"wrote {len(os.listdir(GOLD))} fixtures, {tot/1024:.0f} KiB, under {GOLD}")


if __name__ == "__main__":
    main()
