"""Autograd through the module forwards, and the module-level functions of the hot path, against the REAL reference.

Upstream, Denoiser.forward (diffab_pytorch.py:558-607), InvariantPointAttentionLayer.forward (:389-465), OrientationLoss (:610-625)
and euclidean_transform / inverse_euclidean_transform (:315-336) are ordinary differentiable torch code: a caller may put a loss of
their own on model.denoise().  Here each is a taped HIP forward plus a HIP backward from arbitrary cotangents
(diffab_denoise_step_fwd_taped / _bwd, diffab_ipa_layer_fwd_taped / _bwd, diffab_orientation_loss_bwd, diffab_frames_*).  Goldens:
oracle/gen_golden.py, autograd of the unmodified reference with seeded random cotangents (tests/golden/module_autograd_*.npz,
orientation_loss_grads.npz, frames.npz).
"""
import numpy as np
import pytest
import torch

from conftest import maxrel
from diffab_pytorch import _hip, synthetic as syn

pytestmark = pytest.mark.gpu
T = torch.from_numpy
GTOL = 2e-4  # gradients: the bar of the training-step goldens (tests/test_gpu_parity.py)


@pytest.fixture(scope="module")
def hip():
    lib = _hip.lib()
    assert lib.diffab_device_ok() == 1
    return lib


def check_grad(name, got, g, tol=GTOL):
    """full gradient ("grad/<name>") or norm + strided subsample ("info/", "sub/") as written by gen_golden.py"""
    got = got.detach().float().cpu()
    if "grad/" + name in g:
        want = T(g["grad/" + name])
        assert got.shape == want.shape, name
        assert maxrel(got, want) < tol, (name, maxrel(got, want))
        return
    n, stride, off, norm, amax = g["info/" + name]
    flat = got.reshape(-1)
    assert flat.numel() == int(n), name
    sub = flat[int(off)::int(stride)][:512]
    want = T(g["sub/" + name])
    assert float((sub.double() - want.double()).abs().max()) < tol * amax, (name, float((sub - want).abs().max()), amax)
    assert abs(float(flat.double().norm()) - norm) < tol * max(norm, 1e-30) * 10, (name, float(flat.double().norm()), norm)


def test_frames_and_angular_encoding_vs_reference_goldens(hip, golden):
    from diffab_pytorch.diffab_pytorch import AngularEncoding, euclidean_transform, inverse_euclidean_transform

    g = golden("frames")
    x, R, t = T(g["x"]).cuda(), T(g["R"]).cuda(), T(g["t"]).cuda()
    assert maxrel(euclidean_transform(x, R, t), g["fwd"]) < 1e-6
    assert maxrel(inverse_euclidean_transform(x, R, t), g["inv"]) < 1e-6
    assert maxrel(inverse_euclidean_transform(euclidean_transform(x, R, t), R, t), g["x"]) < 2e-6
    # CPU tensors in, CPU tensors out (reference callers never move tensors themselves)
    assert not euclidean_transform(x.cpu(), R.cpu(), t.cpu()).is_cuda
    # gradients with respect to the points
    for fn, key in ((euclidean_transform, "grad_fwd"), (inverse_euclidean_transform, "grad_inv")):
        xg = x.clone().requires_grad_(True)
        (fn(xg, R, t) * T(g["cot"]).cuda()).sum().backward()
        assert maxrel(xg.grad, g[key]) < 1e-6, key
    # ... and with respect to the frames (reference: the einsums of :324 / :336 are differentiable in r and t)
    for fn, key in ((euclidean_transform, "fwd"), (inverse_euclidean_transform, "inv")):
        Rg, tg = R.clone().requires_grad_(True), t.clone().requires_grad_(True)
        (fn(x, Rg, tg) * T(g["cot"]).cuda()).sum().backward()
        assert maxrel(Rg.grad, g[f"grad_{key}_R"]) < 2e-6 and maxrel(tg.grad, g[f"grad_{key}_t"]) < 2e-6, key
    # reference tests/test_modules.py:16-26
    enc = AngularEncoding(num_funcs=3)
    assert enc.get_output_dimension(3) == 3 * (3 * 2 * 2 + 1)
    assert enc(torch.rand(32, 16, 3)).shape == (32, 16, 39)
    out = enc(T(g["xa"]).cuda())
    assert maxrel(out, g["enc"]) < 1e-6 and torch.equal(out[..., 0::13].cpu(), T(g["xa"]))  # the x column is copied, not recomputed
    assert torch.equal(enc.freq_bands, torch.tensor([1.0, 2.0, 3.0, 1.0, 0.5, 1.0 / 3.0]))
    # differentiable in x like the reference's torch expression (diffab_pytorch.py:41-52): HIP backward against torch autograd of the
    # same three terms on the CPU
    xa = T(g["xa"])
    cot = torch.randn(*xa.shape[:-1], xa.shape[-1] * 13, generator=torch.Generator().manual_seed(3))
    xg = xa.clone().cuda().requires_grad_(True)
    (enc(xg) * cot.cuda()).sum().backward()
    xr = xa.clone().requires_grad_(True)
    xe = xr.unsqueeze(-1)
    ref = torch.cat([xe, torch.sin(enc.freq_bands * xe), torch.cos(enc.freq_bands * xe)], dim=-1).flatten(-2)
    (ref * cot).sum().backward()
    assert maxrel(xg.grad, xr.grad) < 1e-6


def test_orientation_loss_autograd_vs_reference_goldens(hip, golden):
    from diffab_pytorch.diffab_pytorch import OrientationLoss

    g = golden("orientation_loss_grads")
    for red in ("mean", "sum", "none"):
        p, t = T(g["pred"]).cuda().requires_grad_(True), T(g["target"]).cuda().requires_grad_(True)
        val = OrientationLoss(reduction=red)(p, t)
        assert val.grad_fn is not None
        assert maxrel(val, g[f"{red}/value"]) < 1e-6
        (val * T(g[f"{red}/cot"]).cuda()).sum().backward()
        assert maxrel(p.grad, g[f"{red}/d_pred"]) < 1e-5, red
        assert maxrel(t.grad, g[f"{red}/d_target"]) < 1e-5, red
    # reference tests/test_loss.py:17-21
    R = T(g["target"]).cuda()
    assert float(OrientationLoss()(R, R)) < 1e-10


@pytest.mark.parametrize("tag", ["unit", "bench"])
def test_denoiser_and_ipa_layer_autograd_vs_reference_goldens(hip, golden, tag):
    from diffab_pytorch.diffab_pytorch import Denoiser

    g = golden("module_autograd_" + tag)
    B, K, seed, D, C, NL, DS, H, PQ, PV = [int(v) for v in g["meta"]]
    dims = dict(D=D, C=C, NL=NL, DS=DS, H=H, PQ=PQ, PV=PV, V=21)
    den = Denoiser(D, C, NL, DS, PQ, PV, H, 21)
    den.load_state_dict(syn.denoiser_state_dict(dims, seed=seed, prefix=""), strict=True)
    den = den.cuda().train()
    inp = {k: v.cuda() for k, v in syn.patches(B, K, dims, seed=seed, coord_sigma=float(g["coord_sigma"])).items()}
    rc = inp["res_context_emb"].clone().requires_grad_(True)
    pc = inp["pair_context_emb"].clone().requires_grad_(True)
    x_t = inp["translations"].clone().requires_grad_(True)  # the frames are differentiable inputs as well (:315-336, :594-596)
    O_t = inp["orientations"].clone().requires_grad_(True)
    out = den(inp["seq_idx"], x_t, O_t, rc, pc, T(g["beta"]).cuda(), None, None)
    for k in ("translations_eps", "orientations_t0", "seq_posterior"):
        assert out[k].grad_fn is not None, k  # a caller's own loss on these reaches the parameters
    assert maxrel(out["translations_eps"], g["out_eps"]) < 1e-4 and maxrel(out["seq_posterior"], g["out_post"]) < 1e-4
    loss = (out["translations_eps"] * T(g["c_eps"]).cuda()).sum() + (out["orientations_t0"] * T(g["c_O0"]).cuda()).sum() + \
        (out["seq_posterior"] * T(g["c_post"]).cuda()).sum()
    loss.backward()
    check_grad("res_ctx", rc.grad, g)
    check_grad("pair_ctx", pc.grad, g)
    check_grad("x_t", x_t.grad, g)
    check_grad("O_t", O_t.grad, g)
    for n_, p_ in den.named_parameters():
        assert p_.grad is not None, n_
        check_grad(n_, p_.grad, g)
    # a single cotangent (the others None inside autograd): only the translation head and the trunk receive gradients
    den.zero_grad(set_to_none=True)
    out = den(inp["seq_idx"], inp["translations"], inp["orientations"], inp["res_context_emb"], inp["pair_context_emb"], T(g["beta"]).cuda())
    out["translations_eps"].square().sum().backward()
    assert den.coordinate_denoising[4].weight.grad.abs().max() > 0
    assert float(den.sequence_denoising[4].weight.grad.abs().max()) == 0.0
    # ---- one IPA layer: d y -> d x, d e, parameter gradients
    layer = den.ipa.layers[0]
    layer.zero_grad(set_to_none=True)
    x = inp["res_context_emb"].clone().requires_grad_(True)
    e = inp["pair_context_emb"].clone().requires_grad_(True)
    Rl, tl = inp["orientations"].clone().requires_grad_(True), inp["translations"].clone().requires_grad_(True)
    y = layer(x, e, Rl, tl)
    assert y.grad_fn is not None and maxrel(y, g["layer/y"]) < 1e-4
    (y * T(g["layer/c_y"]).cuda()).sum().backward()
    check_grad("layer/x", x.grad, g)
    check_grad("layer/e", e.grad, g)
    check_grad("layer/R", Rl.grad, g)
    check_grad("layer/t", tl.grad, g)
    for n_, p_ in layer.named_parameters():
        check_grad("layer/" + n_, p_.grad, g)
    # DIFFAB_FLAG_FORCE_GENERIC under autograd: the taped forwards ignore it (their backward reads the tape the MFMA path
    # writes) - same output, same gradients, at unit dims (generic kernels on both sides anyway) and at the benchmark geometry
    layer.zero_grad(set_to_none=True)
    x2, e2 = x.detach().clone().requires_grad_(True), e.detach().clone().requires_grad_(True)
    y2 = layer(x2, e2, inp["orientations"], inp["translations"], flags=_hip.FLAG_FORCE_GENERIC)
    assert maxrel(y2, g["layer/y"]) < 1e-4
    (y2 * T(g["layer/c_y"]).cuda()).sum().backward()
    check_grad("layer/x", x2.grad, g)
    check_grad("layer/e", e2.grad, g)
    for n_, p_ in layer.named_parameters():
        check_grad("layer/" + n_, p_.grad, g)
    # under no_grad the same call is the inference kernel: detached output, same numbers
    with torch.no_grad():
        y0 = layer(x, e, inp["orientations"], inp["translations"])
    assert y0.grad_fn is None and maxrel(y0, g["layer/y"]) < 1e-4


@pytest.mark.parametrize("tag", ["unit", "bench"])
def test_ipa_layer_without_pair_bias_vs_reference_goldens(hip, golden, tag):
    """InvariantPointAttentionLayer(use_pair_bias=False) (reference :348-385, :422-459: two independent logits, no pair terms in to_out):
    forward and every gradient (x, the frames, the nine parameters) against the real reference's autograd.  Weights: the same seeded
    construction as the generator (creation order of the reference, then the gamma draw)."""
    from diffab_pytorch.diffab_pytorch import InvariantPointAttentionLayer

    g = golden("ipa_layer_no_pair_bias_" + tag)
    B, K, seed, D, C, DS, H, PQ, PV = [int(v) for v in g["meta"]]
    dims = dict(D=D, C=C, NL=1, DS=DS, H=H, PQ=PQ, PV=PV, V=21)
    torch.manual_seed(seed)
    layer = InvariantPointAttentionLayer(D, C, DS, PQ, PV, H, use_pair_bias=False)
    with torch.no_grad():
        layer.gamma.copy_(torch.rand(H) + 0.2)
    assert not hasattr(layer, "to_pair_bias") and layer.to_out.in_features == H * DS + H * PV * 3 + H * PV
    layer = layer.cuda()
    inp = {k: v.cuda() for k, v in syn.patches(B, K, dims, seed=seed, coord_sigma=float(g["coord_sigma"])).items()}
    with torch.no_grad():
        assert maxrel(layer(inp["res_context_emb"], inp["pair_context_emb"], inp["orientations"], inp["translations"]), g["y"]) < 1e-4
    x = inp["res_context_emb"].clone().requires_grad_(True)
    Rl, tl = inp["orientations"].clone().requires_grad_(True), inp["translations"].clone().requires_grad_(True)
    y = layer(x, inp["pair_context_emb"], Rl, tl)
    assert maxrel(y, g["y"]) < 1e-4
    (y * T(g["c_y"]).cuda()).sum().backward()
    check_grad("x", x.grad, g)
    check_grad("R", Rl.grad, g)
    check_grad("t", tl.grad, g)
    for n_, p_ in layer.named_parameters():
        check_grad(n_, p_.grad, g)


def test_backbone_from_sampled_frames_on_the_device(hip):
    """Output side: backbone atoms of sampled frames through the HIP frame kernel against the host expression
    (float64), the exact inverse on (N, CA, C), and a PDB written from device tensors."""
    import os
    import tempfile

    from diffab_pytorch import io as dio

    d = syn.UNIT_DIMS
    inp = syn.patches(3, 24, d, seed=8, coord_sigma=12.0)
    x, O = inp["translations"].cuda(), inp["orientations"].cuda()
    bb = dio.backbone_from_frames(x, O)
    assert bb.is_cuda and bb.shape == (3, 24, 5, 3)
    local = torch.tensor([dio.IDEAL_BACKBONE[a] for a in dio.BACKBONE_ATOMS], dtype=torch.float64)
    want = torch.einsum("ak,...kc->...ac", local, inp["orientations"].double()) + inp["translations"].double().unsqueeze(-2)
    assert maxrel(bb, want) < 1e-6
    assert maxrel(bb, dio.backbone_from_frames(inp["translations"], inp["orientations"])) < 1e-6  # host path, same numbers
    t2, R2 = dio.frames_from_backbone(bb[..., 0, :], bb[..., 1, :], bb[..., 2, :])
    assert maxrel(t2, x) < 1e-6 and maxrel(R2, O) < 1e-5
    with tempfile.TemporaryDirectory() as tmp:
        n = dio.write_pdb(os.path.join(tmp, "p.pdb"), inp["seq_idx"][0].cuda(), x[0], O[0])
        assert n == 24 * 4


def test_empty_and_odd_inputs_through_the_elementwise_entries():
    """Edge cases of the entries that take free-form tensors: an EMPTY batch (torch's empty tensors carry null pointers: the C ABI returns
    before it looks at them), non-contiguous and float64 inputs (made contiguous fp32 by the wrappers), and the without-replacement bin
    draw with a table whose bin count is not a power of two, K = n_bins (a permutation of all bins) and K > n_bins (an error)."""
    import diffab_oracle as orc
    from diffab_pytorch import _hip, so3 as S
    from diffab_pytorch.diffab_pytorch import AngularEncoding, euclidean_transform, inverse_euclidean_transform
    from diffab_pytorch.diffusion import CoordinateDiffuser, OrientationDiffuser, SequenceDiffuser

    g = torch.Generator().manual_seed(0)
    R = torch.linalg.qr(torch.randn(2, 5, 3, 3, generator=g)).Q.cuda()
    t = torch.randn(2, 5, 3, generator=g).cuda()
    x = torch.randn(2, 4, 5, 3, 3, generator=g).cuda()
    y = euclidean_transform(x, R, t)
    assert euclidean_transform(x[:0], R[:0], t[:0]).shape == (0, 4, 5, 3, 3)
    assert inverse_euclidean_transform(y[:0], R[:0], t[:0]).shape == (0, 4, 5, 3, 3)
    xn = torch.randn(2, 4, 5, 3, 6, generator=g).cuda()[..., ::2]
    assert torch.equal(euclidean_transform(xn, R, t), euclidean_transform(xn.contiguous(), R, t))
    assert torch.equal(euclidean_transform(x.double(), R.double(), t.double()).float(), y)
    assert float((inverse_euclidean_transform(y, R, t) - x).abs().max()) < 1e-5
    ae = AngularEncoding()
    a = torch.randn(3, 7, 2, generator=g).cuda()
    assert ae(a[:0]).shape == (0, 7, ae(a).shape[-1])
    e_seq, e_m, e_t = torch.zeros(0, 16, dtype=torch.long).cuda(), torch.zeros(0, 16, dtype=torch.bool).cuda(), torch.zeros(0, dtype=torch.long).cuda()
    first = lambda r: r[0] if isinstance(r, tuple) else r
    assert first(SequenceDiffuser(100, 0.01, 0.999, 21).diffuse_from_t0(e_seq, e_t, e_m)).shape == (0, 16)
    assert first(CoordinateDiffuser(100, 0.01, 0.999).diffuse_from_t0(torch.zeros(0, 16, 3).cuda(), e_t, e_m)).shape == (0, 16, 3)
    assert first(OrientationDiffuser(100, 0.01, 0.999).diffuse_from_t0(torch.zeros(0, 16, 3, 3).cuda(), e_m, e_t)).shape == (0, 16, 3, 3)
    assert S.log_rotmat(torch.zeros(0, 3, 3).cuda()).shape == (0, 3, 3)
    so = S.SO3(torch.linspace(0.02, 1.5, 16), n_bins=1000, num_iters=256)
    rows = torch.tensor([0, 3, 15])
    race = -torch.log(torch.rand(3, 1000, generator=g).clamp_min(1e-30))
    b = so.draw_bins_without_replacement(rows, 1000, race=race).cpu().long()
    assert torch.equal(b, orc.igso3_bins_without_replacement(so.histograms[rows].cpu(), race, 1000))
    assert all(sorted(r.tolist()) == list(range(1000)) for r in b)
    assert so.draw_bins_without_replacement(rows[:0], 8, race=race[:0]).shape == (0, 8)
    with pytest.raises(_hip.DiffabHipError):
        so.draw_bins_without_replacement(rows, 1001, race=race)
    # an empty batch through the layer and the denoiser: empty outputs, as the reference's einsums give
    from diffab_pytorch.diffab_pytorch import Denoiser, InvariantPointAttentionLayer

    d = syn.BENCH_DIMS
    layer = InvariantPointAttentionLayer(d["D"], d["C"], d["DS"], d["PQ"], d["PV"], d["H"]).cuda()
    z = lambda *sh: torch.zeros(*sh, device="cuda")
    assert layer(z(0, 64, d["D"]), z(0, 64, 64, d["C"]), z(0, 64, 3, 3), z(0, 64, 3)).shape == (0, 64, d["D"])
    den = Denoiser(d["D"], d["C"], 1, d["DS"], d["PQ"], d["PV"], d["H"], 21).cuda()
    out = den(torch.zeros(0, 64, dtype=torch.long, device="cuda"), z(0, 64, 3), z(0, 64, 3, 3), z(0, 64, d["D"]), z(0, 64, 64, d["C"]), z(0))
    assert out["translations_eps"].shape == (0, 64, 3) and out["orientations_t0"].shape == (0, 64, 3, 3) and out["seq_posterior"].shape == (0, 64, 21)
