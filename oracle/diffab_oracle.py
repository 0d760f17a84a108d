"""CPU oracle for the DiffAb diffusion / denoise hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import this module, and there only as the checker (or as the
timed CPU baseline), never as the thing shipped.  The product path
(``diffab-pytorch_amd/``) never imports it and fails loudly without its HIP
library.

What it is: a from-scratch torch-CPU restatement of the reference algorithm
(dohlee/diffab-pytorch), written from the math in SURVEY.md Appendix A, in the
same *formulation* as the reference (including the materialised
(b, h, K, K, p, 3) point-difference tensor, so that timing it is
representative of the reference's CPU path).  Every function cites the
reference file:line it restates.

Pinning: ``oracle/gen_golden.py`` imports the real reference in the build
container, checks every function below against it, and writes the golden
vectors under ``tests/golden/``; ``tests/test_oracle_golden.py`` re-checks the
oracle against those vectors wherever the tests run (the reference itself
never travels).  Parts with no counterpart in the reference (the reverse
sampler, the Philox stream, the inverse-CDF angle draw) are marked
BUILD-DEFINED.

All functions are dtype-generic: feed float64 tensors to get a float64 "truth".
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import numpy as np
import torch

# --------------------------------------------------------------------------
# A.1 variance schedule                       reference: diffusion.py:11-35
# --------------------------------------------------------------------------


def cosine_variance_schedule(T: int, s: float = 8e-3, beta_max: float = 0.999) -> Dict[str, torch.Tensor]:
    """f_t = cos^2(((t/T)+s)/(1+s) * pi/2); abar = f_t/f_0; beta_0 = 0,
    beta_t = clip(1 - abar_t/abar_{t-1}, 1e-5, beta_max).  (diffusion.py:11-35)

    Note abar is NOT the cumulative product of the clipped alphas
    (diffusion.py:19-26) - reproduce, do not "fix".
    """
    steps = torch.arange(T + 1)
    f = torch.cos((steps / T + s) / (1 + s) * math.pi / 2.0) ** 2
    abar = f / f[0]
    ratio = 1 - abar[1:] / abar[:-1]
    beta = torch.cat([torch.zeros(1), ratio.clamp(min=1e-5, max=beta_max)])
    return {
        "alpha": 1 - beta,
        "alpha_bar": abar,
        "alpha_bar_sqrt": abar.sqrt(),
        "one_minus_alpha_bar_sqrt": (1 - abar).sqrt(),
        "beta": beta,
    }


# --------------------------------------------------------------------------
# A.2 sequence (multinomial) diffusion        reference: diffusion.py:44-192
# --------------------------------------------------------------------------

V_AA = 21  # diffusion.py:47 hard-codes 21 and ignores the ctor argument


def _onehot(idx: torch.Tensor, dtype=torch.float32) -> torch.Tensor:
    return torch.nn.functional.one_hot(idx, V_AA).to(dtype)


def weighted_multinomial(p1, p2, w1, w2):
    """w1 p1 + w2 p2, weights (B,) broadcast over (K, V).  (diffusion.py:38-41)"""
    return w1[:, None, None] * p1 + w2[:, None, None] * p2


def seq_forward_prob_single_step(seq, t, mask, sched, dtype=torch.float32):
    """q(s_t | s_{t-1}) = (1-beta_t) onehot + beta_t/21; exact one-hot where
    the residue is not generated.  (diffusion.py:49-79)"""
    oh = _onehot(seq, dtype)
    b = sched["beta"][t].to(dtype)[:, None, None]
    noised = (1 - b) * oh + b * (torch.ones_like(oh) / V_AA)
    return torch.where(mask[..., None], noised, oh)


def seq_forward_prob_from_t0(seq0, t, mask, sched, dtype=torch.float32):
    """q(s_t | s_0) = abar_t onehot + (1-abar_t)/21.  (diffusion.py:105-135)"""
    oh = _onehot(seq0, dtype)
    ab = sched["alpha_bar"][t].to(dtype)[:, None, None]
    noised = ab * oh + (1 - ab) * (torch.ones_like(oh) / V_AA)
    return torch.where(mask[..., None], noised, oh)


def seq_posterior_single_step(seq_t, seq0, t, mask, sched, dtype=torch.float32):
    """q(s_{t-1} | s_t, s_0) prop. to q(s_t|.)[centre s_t] * q(s_{t-1}|s_0),
    normalised over the vocabulary.  (diffusion.py:168-192)"""
    p = seq_forward_prob_single_step(seq_t, t, mask, sched, dtype) * seq_forward_prob_from_t0(
        seq0, t - 1, mask, sched, dtype
    )
    return p / p.sum(dim=-1, keepdim=True)


def categorical_from_uniform(p: torch.Tensor, u: torch.Tensor) -> torch.Tensor:
    """BUILD-DEFINED draw: smallest v with cumsum(p)[v] > u * sum(p), sequential
    fp32 prefix sum (same order as the device kernel).  Replaces
    torch.multinomial (diffusion.py:156-158), whose mt19937 stream cannot be
    reproduced on a GPU."""
    acc = torch.zeros_like(u)
    tot = torch.zeros_like(u)
    for v in range(p.shape[-1]):
        tot = tot + p[..., v]
    thr = u * tot
    out = torch.full(u.shape, p.shape[-1] - 1, dtype=torch.int64)
    done = torch.zeros(u.shape, dtype=torch.bool)
    for v in range(p.shape[-1]):
        acc = acc + p[..., v]
        hit = (acc > thr) & ~done
        out = torch.where(hit, torch.full_like(out, v), out)
        done = done | hit
    return out


# --------------------------------------------------------------------------
# A.3 translation diffusion                   reference: diffusion.py:199-236
# --------------------------------------------------------------------------


def coord_diffuse_from_t0(x0, t, mask, eps, sched):
    """x_t = sqrt(abar_t) x0 + sqrt(1-abar_t) eps, where(mask, x_t, x0).
    eps is returned unmasked by the reference (diffusion.py:225-234)."""
    a = sched["alpha_bar_sqrt"][t].to(x0.dtype)[:, None, None]
    b = sched["one_minus_alpha_bar_sqrt"][t].to(x0.dtype)[:, None, None]
    xt = a * x0 + b * eps
    return torch.where(mask[..., None], xt, x0)


# --------------------------------------------------------------------------
# A.4 SO(3)                                   reference: so3.py:142-259
# --------------------------------------------------------------------------


def hat(v):
    """vector -> skew-symmetric matrix [[0,-z,y],[z,0,-x],[-y,x,0]].  (so3.py:185-204)"""
    x, y, z = v[..., 0], v[..., 1], v[..., 2]
    o = torch.zeros_like(x)
    return torch.stack(
        [torch.stack([o, -z, y], -1), torch.stack([z, o, -x], -1), torch.stack([-y, x, o], -1)], -2
    )


def vee(S):
    """(S21, S02, S10).  (so3.py:165-170)"""
    return torch.stack([S[..., 2, 1], S[..., 0, 2], S[..., 1, 0]], -1)


def exp_so3(S):
    """I + S sin(n)/n + S^2 (1-cos n)/n^2, n = |vee S|; NaN at n = 0 like the
    reference.  (so3.py:219-237)"""
    n = vee(S).norm(dim=-1)[..., None, None]
    eye = torch.eye(3, dtype=S.dtype).expand_as(S)
    return eye + S * torch.sin(n) / n + S @ S * (1 - torch.cos(n)) / n**2


def log_so3(R):
    """theta/(2 sin theta) (R - R^T), theta = acos((tr R - 1)/2); NaN at
    theta = 0 like the reference.  (so3.py:146-162)"""
    tr = R.diagonal(dim1=-2, dim2=-1).sum(-1)
    th = torch.acos((tr - 1) / 2)[..., None, None]
    return th / (2 * torch.sin(th)) * (R - R.transpose(-1, -2))


def rotvec_to_matrix(v):
    """so3.py:207-216"""
    return exp_so3(hat(v))


def matrix_to_rotvec(R):
    """so3.py:173-182"""
    return vee(log_so3(R))


def scale_rot(R, k):
    """exp(k log R), k broadcast from the left.  (so3.py:240-259)"""
    if k.ndim > R.ndim:
        raise ValueError("k has more dimensions than R")
    while k.ndim < R.ndim:
        k = k.unsqueeze(-1)
    return exp_so3(k * log_so3(R))


def igso3_angular_pdf(theta, sigma, num_iters=1024):
    """f(theta) = (1-cos theta)/pi * sum_{l<L} (2l+1) exp(-l(l+1) sigma^2)
    sin((l+1/2) theta)/sin(theta/2).  (so3.py:65-72)"""
    l = torch.arange(num_iters).view(-1, 1)
    a = (1 - torch.cos(theta)) / torch.pi
    b = (2 * l + 1) * torch.exp(-l * (l + 1) * sigma**2)
    c = torch.sin((l + 0.5) * theta) / torch.sin(theta / 2.0)
    return (a * b * c).sum(dim=0)


def igso3_table(sigmas, n_bins=8192, num_iters=1024):
    """One un-normalised histogram row per sigma, density at the bin centres,
    NaN -> 0 and negatives -> 0.  (so3.py:37-63)"""
    width = torch.pi / n_bins
    centres = torch.arange(0, torch.pi, width) + width / 2.0
    rows = []
    for sg in sigmas:
        rows.append(torch.nan_to_num(igso3_angular_pdf(centres, sg, num_iters)).clamp_min(0.0))
    return torch.stack(rows)


def igso3_cdf_table(pdf_table: torch.Tensor) -> torch.Tensor:
    """BUILD-DEFINED: normalised inclusive prefix sum of each histogram row
    (float64 accumulate, stored float32, last entry forced to 1)."""
    c = pdf_table.double().cumsum(dim=-1)
    c = c / c[:, -1:]
    c[:, -1] = 1.0
    return c.float()


def igso3_theta_from_hist(bin_idx, u, n_bins=8192):
    """theta = bin_start[m] + width * U(0,1).  (so3.py:74-84)"""
    width = torch.pi / n_bins
    starts = torch.arange(0, torch.pi, width)
    return starts[bin_idx] + width * u


def igso3_bin_from_cdf(cdf_rows: torch.Tensor, u: torch.Tensor) -> torch.Tensor:
    """BUILD-DEFINED inverse-CDF bin draw (with replacement): first m with
    cdf[m] > u.  cdf_rows (..., n_bins) matches u (...,) on the leading dims.
    The reference draws K bins per row WITHOUT replacement with
    torch.multinomial (so3.py:78); see DESIGN.md."""
    cdf_rows = cdf_rows.contiguous()
    if cdf_rows.ndim == u.ndim + 1:  # one table row per draw
        idx = torch.searchsorted(cdf_rows, u[..., None].contiguous(), right=True)[..., 0]
    else:  # one table row per leading index, many draws per row
        idx = torch.searchsorted(cdf_rows, u.contiguous(), right=True)
    return idx.clamp_max(cdf_rows.shape[-1] - 1)


def igso3_bins_without_replacement(pdf_rows: torch.Tensor, race: torch.Tensor, num_samples: int) -> torch.Tensor:
    """The K bins of each row drawn WITHOUT replacement, in draw order - the joint distribution of
    `torch.multinomial(probs, num_samples)` (so3.py:78; replacement=False is its default) - as the exponential race:
    with race ~ Exp(1) i.i.d., the bins ordered by probs / race, largest first (ties: lower bin), are such a draw.
    pdf_rows, race: (n, n_bins) float32; returns (n, num_samples) int64."""
    import numpy as np

    # IEEE float32 division, as the HIP kernel; zero-mass bins have key 0 and a draw of exactly 0 counts as the smallest positive float
    # (csrc/diffusion_kernels.hip igso3_race_kernel: the keys are NaN-free for any race)
    p32, e32 = pdf_rows.float().numpy(), np.maximum(race.float().numpy(), np.float32(1.17549435e-38))
    with np.errstate(divide="ignore", invalid="ignore"):
        key = np.where(p32 > 0, (p32 / e32).astype(np.float32), np.float32(0)).astype(np.float32)
    n, nb = key.shape
    out = np.empty((n, num_samples), dtype=np.int64)
    ar = np.arange(nb)
    for r in range(n):
        order = np.lexsort((ar, -key[r].astype(np.float64)))  # primary: key descending, secondary: bin ascending
        out[r] = order[:num_samples]
    return torch.from_numpy(out)


def igso3_theta_from_gaussian(sigma, z):
    """(2 sigma + sigma z) mod pi, floor-mod.  (so3.py:86-96)"""
    return (2.0 * sigma + sigma * z) % torch.pi


def igso3_rotvec(axis_raw, theta_hist, theta_gauss, sigma, sigma_threshold=0.1):
    """u = normalize(randn) ; theta = hist if sigma < thr else gaussian.
    (so3.py:98-126)"""
    u = torch.nn.functional.normalize(axis_raw, dim=-1)
    use_hist = (sigma < sigma_threshold)[..., None].expand_as(theta_hist)
    theta = torch.where(use_hist, theta_hist, theta_gauss)
    return u * theta[..., None]


def orient_diffuse_from_t0(O0, mask, t, rotvec, sched):
    """O_t = scale_rot(O_0, sqrt(abar_t)) @ exp(hat(rotvec)); where(mask, O_t, O_0).
    (diffusion.py:262-294)"""
    mean = scale_rot(O0, sched["alpha_bar_sqrt"][t].to(O0.dtype))
    Ot = torch.einsum("bnij,bnjk->bnik", mean, rotvec_to_matrix(rotvec))
    return torch.where(mask[..., None, None], Ot, O0)


# --------------------------------------------------------------------------
# A.5 invariant point attention               reference: diffab_pytorch.py:315-498
# --------------------------------------------------------------------------


def to_global(p, R, t):
    """row-vector convention: global = p @ R + t, R/t broadcast over heads.
    p (b,h,l,P,3).  (diffab_pytorch.py:315-324)"""
    return torch.einsum("bhlpk,blkc->bhlpc", p, R) + t[:, None, :, None, :]


def to_local(p, R, t):
    """local = (p - t) @ R^T.  (diffab_pytorch.py:327-336)"""
    return torch.einsum("bhlpk,blck->bhlpc", p - t[:, None, :, None, :], R)


def ipa_layer(x, e, R, t, sd, prefix, H, return_attn=False, use_pair_bias=True):
    """One InvariantPointAttentionLayer.forward (diffab_pytorch.py:389-465):
    no LayerNorm / residual / transition, raw gamma (no softplus), unmasked.
    use_pair_bias=False (:348-385): no to_pair_bias, two independent logits (scale 2^-1/2), no pair block in to_out's input."""
    g = lambda name: sd[prefix + name].to(x.dtype)
    B, K, D = x.shape
    Wqs, Wks, Wvs = g("to_q_scalar.weight"), g("to_k_scalar.weight"), g("to_v_scalar.weight")
    Wqp, Wkp, Wvp = g("to_q_point.weight"), g("to_k_point.weight"), g("to_v_point.weight")
    Wb, gamma = (g("to_pair_bias.weight") if use_pair_bias else None), g("gamma")
    Wo, bo = g("to_out.weight"), g("to_out.bias")
    ds = Wqs.shape[0] // H
    Pq = Wqp.shape[0] // (3 * H)
    Pv = Wvp.shape[0] // (3 * H)

    def heads(y):  # "(h d)" head-major split, :396
        return y.view(B, K, H, -1).permute(0, 2, 1, 3)

    qs, ks, vs = heads(x @ Wqs.T), heads(x @ Wks.T), heads(x @ Wvs.T)

    def points(y, P):  # "(h p c)" split, :406
        return y.view(B, K, H, P, 3).permute(0, 2, 1, 3, 4)

    qp = to_global(points(x @ Wqp.T, Pq), R, t)
    kp = to_global(points(x @ Wkp.T, Pq), R, t)
    vp = to_global(points(x @ Wvp.T, Pv), R, t)

    logit_s = torch.einsum("bhid,bhjd->bhij", qs, ks) * ds**-0.5  # :416-419
    bias = (e @ Wb.T).permute(0, 3, 1, 2) if use_pair_bias else 0.0  # :423
    diff = qp[:, :, :, None] - kp[:, :, None, :]  # (b,h,i,j,p,3)   :426-428
    logit_p = -0.5 * (4.5 * Pq) ** -0.5 * gamma.view(1, H, 1, 1) * (diff**2).sum(-1).sum(-1)  # :431-436
    n_logits = 3 if use_pair_bias else 2  # :385
    attn = ((n_logits**-0.5) * ((logit_s + bias + logit_p) if use_pair_bias else (logit_s + logit_p))).softmax(dim=-1)  # :439-443

    o_s = torch.einsum("bhij,bhjd->bhid", attn, vs).permute(0, 2, 1, 3).reshape(B, K, H * ds)  # :445-446
    o_e = torch.einsum("bhij,bijc->bhic", attn, e).permute(0, 2, 1, 3).reshape(B, K, -1)  # :449-450
    o_g = torch.einsum("bhij,bhjpc->bhipc", attn, vp)  # :452
    o_l = to_local(o_g, R, t)  # :453
    o_n = o_l.norm(dim=-1)  # :454
    o_l = o_l.permute(0, 2, 1, 3, 4).reshape(B, K, H * Pv * 3)  # "(h p c)"
    o_n = o_n.permute(0, 2, 1, 3).reshape(B, K, H * Pv)  # "(h p)"
    feat = torch.cat([o_s, o_e, o_l, o_n] if use_pair_bias else [o_s, o_l, o_n], dim=-1)  # :460
    out = feat @ Wo.T + bo  # :464
    if return_attn:
        return out, attn, feat
    return out


def ipa_module(x, e, R, t, sd, prefix, n_layers, H):
    """x <- layer_l(x, e, R, t), same e/R/t every layer.  (diffab_pytorch.py:494-498)"""
    for l in range(n_layers):
        x = ipa_layer(x, e, R, t, sd, f"{prefix}layers.{l}.", H)
    return x


# --------------------------------------------------------------------------
# A.6 denoiser                                reference: diffab_pytorch.py:501-607
# --------------------------------------------------------------------------


def _mlp3(z, sd, prefix):
    g = lambda n: sd[prefix + n].to(z.dtype)
    h = torch.relu(z @ g("0.weight").T + g("0.bias"))
    h = torch.relu(h @ g("2.weight").T + g("2.bias"))
    return h @ g("4.weight").T + g("4.bias")


def denoiser(sd, seq_t, x_t, O_t, res_ctx, pair_ctx, beta, n_layers, H, prefix="denoiser."):
    """Denoiser.forward (diffab_pytorch.py:558-607).  Masks are accepted by the
    reference but never read (:566-567), so they are not parameters here.
    Returns the three public outputs plus the pre-softmax aa-type logits."""
    g = lambda n: sd[prefix + n].to(res_ctx.dtype)
    B, K = seq_t.shape
    s_emb = g("sequence_embedding.weight")[seq_t]  # :572
    h = torch.cat([res_ctx, s_emb], dim=-1)  # :573
    h = torch.relu(h @ g("to_res_emb.0.weight").T + g("to_res_emb.0.bias"))
    h = h @ g("to_res_emb.2.weight").T + g("to_res_emb.2.bias")  # :574
    h = ipa_module(h, pair_ctx, O_t, x_t, sd, prefix + "ipa.", n_layers, H)  # :579-581
    beta = beta.to(h.dtype)
    temb = torch.stack([beta, torch.sin(beta), torch.cos(beta)], dim=-1)[:, None, :].expand(B, K, 3)  # :584-585
    z = torch.cat([h, temb], dim=-1)  # :588
    eps = _mlp3(z, sd, prefix + "coordinate_denoising.")  # :591
    v = _mlp3(z, sd, prefix + "orientation_denoising.")  # :594
    O0 = O_t @ rotvec_to_matrix(v)  # :595-596
    logits = _mlp3(z, sd, prefix + "sequence_denoising.")
    return {
        "translations_eps": eps,
        "orientations_t0": O0,
        "seq_posterior": logits.softmax(dim=-1),  # :555,:599
        "aa_logits": logits,
        "rotvec": v,
        "res_emb": h,
    }


# --------------------------------------------------------------------------
# A.7 losses                                  reference: diffab_pytorch.py:610-625, 856-880
# --------------------------------------------------------------------------


def orientation_loss_elems(pred, target):
    """(pred^T target - I)^2 element-wise, reduction none.  (diffab_pytorch.py:610-625)"""
    d = torch.einsum("blij,blik->bljk", pred, target)
    return (d - torch.eye(3, dtype=d.dtype).expand_as(d)) ** 2


def hotpath_losses(den, noised_posterior, eps_true, O0_true, gen_mask, res_mask):
    """KL(q_post || p_hat) with p_hat.log() (0 where q = 0), MSE on eps, and the
    orientation discrepancy; each masked by gen & residue, summed, divided by
    the number of masked RESIDUES.  (diffab_pytorch.py:856-880)"""
    m = (gen_mask & res_mask)
    denom = m.sum()
    kl = torch.nn.functional.kl_div(den["seq_posterior"].log(), noised_posterior, reduction="none")
    mse = (den["translations_eps"] - eps_true) ** 2
    ol = orientation_loss_elems(den["orientations_t0"], O0_true)
    mf = m.to(kl.dtype)
    return (
        (kl * mf[..., None]).sum() / denom,
        (mse * mf[..., None]).sum() / denom,
        (ol * mf[..., None, None]).sum() / denom,
    )


# --------------------------------------------------------------------------
# A.8 reverse sampler - BUILD-DEFINED (reference: stub at diffab_pytorch.py:770-776)
# --------------------------------------------------------------------------


def reverse_update(t: int, seq_t, x_t, O_t, den, mask, sched, z, rotvec, u_seq):
    """One reverse step t -> t-1 from the denoiser outputs, with injected noise:
      x_{t-1} = (x_t - beta_t/sqrt(1-abar_t) eps_hat)/sqrt(alpha_t) + sqrt(beta_t) z [t>1]
      O_{t-1} = O0_hat @ exp(hat(rotvec)) [t>1 else O0_hat]
      s_{t-1} ~ Categorical(seq_posterior) via u_seq
    each followed by where(mask, new, old)."""
    dt = x_t.dtype
    beta = sched["beta"][t].to(dt)
    alpha = sched["alpha"][t].to(dt)
    c = beta / sched["one_minus_alpha_bar_sqrt"][t].to(dt)
    x_new = (x_t - c * den["translations_eps"]) / alpha.sqrt()
    O_new = den["orientations_t0"]
    if t > 1:
        x_new = x_new + beta.sqrt() * z
        O_new = O_new @ rotvec_to_matrix(rotvec)
    s_new = categorical_from_uniform(den["seq_posterior"], u_seq)
    return (
        torch.where(mask, s_new, seq_t),
        torch.where(mask[..., None], x_new, x_t),
        torch.where(mask[..., None, None], O_new, O_t),
    )


# --------------------------------------------------------------------------
# Philox4x32-10 counter RNG - BUILD-DEFINED (bit-identical to csrc/philox.h)
# --------------------------------------------------------------------------

_PHILOX_M0 = np.uint64(0xD2511F53)
_PHILOX_M1 = np.uint64(0xCD9E8D57)
_PHILOX_W0 = np.uint32(0x9E3779B9)
_PHILOX_W1 = np.uint32(0xBB67AE85)

STREAM_SEQ, STREAM_TRANS, STREAM_AXIS, STREAM_ANGLE, STREAM_INIT_X, STREAM_INIT_O, STREAM_INIT_S = range(7)


def philox4x32(c0, c1, c2, c3, k0, k1):
    """Vectorised Philox4x32-10.  Counters: arrays/ints (broadcast), key: ints.
    Returns 4 uint32 arrays."""
    c0, c1, c2, c3 = np.broadcast_arrays(*(np.asarray(c, dtype=np.uint32) for c in (c0, c1, c2, c3)))
    c0, c1, c2, c3 = (c.copy() for c in (c0, c1, c2, c3))
    k0 = np.uint32(k0)
    k1 = np.uint32(k1)
    with np.errstate(over="ignore"):
        for _ in range(10):
            p0 = c0.astype(np.uint64) * _PHILOX_M0
            p1 = c2.astype(np.uint64) * _PHILOX_M1
            hi0, lo0 = (p0 >> np.uint64(32)).astype(np.uint32), p0.astype(np.uint32)
            hi1, lo1 = (p1 >> np.uint64(32)).astype(np.uint32), p1.astype(np.uint32)
            c0, c1, c2, c3 = hi1 ^ c1 ^ k0, lo1, hi0 ^ c3 ^ k1, lo0
            k0 = np.uint32(k0 + _PHILOX_W0)
            k1 = np.uint32(k1 + _PHILOX_W1)
    return c0, c1, c2, c3


def u32_to_unit(x):
    """(0,1) float32 from the top 24 bits: (x>>8)*2^-24 + 2^-25, the largest value (which rounds to 1.0 in fp32) clamped to
    1 - 2^-24 (csrc/philox.h)."""
    u = ((x >> np.uint32(8)).astype(np.float32) * np.float32(2.0**-24) + np.float32(2.0**-25)).astype(np.float32)
    return np.minimum(u, np.float32(0.99999994)).astype(np.float32)


def philox_uniform4(seed, patch, residue, step, stream):
    """4 uniforms per (patch, residue, step, stream): counter = (residue, patch, step, stream),
    key = (seed lo, seed hi)."""
    r = philox4x32(residue, patch, step, stream, seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    return tuple(u32_to_unit(x) for x in r)


def philox_normal4(seed, patch, residue, step, stream):
    """4 standard normals by Box-Muller on (u0,u1) and (u2,u3), float32 math."""
    u0, u1, u2, u3 = philox_uniform4(seed, patch, residue, step, stream)
    two_pi = np.float32(6.283185307179586)
    r0 = np.sqrt(np.float32(-2.0) * np.log(u0)).astype(np.float32)
    r1 = np.sqrt(np.float32(-2.0) * np.log(u2)).astype(np.float32)
    return (
        (r0 * np.cos(two_pi * u1)).astype(np.float32),
        (r0 * np.sin(two_pi * u1)).astype(np.float32),
        (r1 * np.cos(two_pi * u3)).astype(np.float32),
        (r1 * np.sin(two_pi * u3)).astype(np.float32),
    )


def uniform_rotation_from_normals(n4):
    """BUILD-DEFINED uniform SO(3) draw: unit quaternion from 4 normals ->
    rotation matrix (w, x, y, z order)."""
    q = torch.nn.functional.normalize(n4, dim=-1)
    w, x, y, z = q.unbind(-1)
    return torch.stack(
        [
            torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)], -1),
            torch.stack([2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)], -1),
            torch.stack([2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)], -1),
        ],
        -2,
    )


# --------------------------------------------------------------------------
# encode_context               reference: diffab_pytorch.py:20-312, 680-724
# --------------------------------------------------------------------------

AA_UNK = 20  # protstruc.general.AA.UNK (reference :115, :273)


def angular_encoding(x, num_funcs):
    """[x, sin(f x), cos(f x)] per input dim, f in [1..n, 1/1..1/n], flattened "(d1 d2)".  (diffab_pytorch.py:20-54)"""
    freq = torch.tensor([i + 1.0 for i in range(num_funcs)] + [1.0 / (i + 1.0) for i in range(num_funcs)]).float().to(x.dtype)
    xe = x.unsqueeze(-1)
    enc = torch.cat([xe, torch.sin(freq * xe), torch.cos(freq * xe)], dim=-1)
    return enc.flatten(-2)


def _mlp(x, sd, prefix, idxs):
    for n, i in enumerate(idxs):
        x = x @ sd[f"{prefix}{i}.weight"].to(x.dtype).T + sd[f"{prefix}{i}.bias"].to(x.dtype)
        if n + 1 < len(idxs):
            x = torch.relu(x)
    return x


def residue_embedding(sd, seq_idx, xyz, orientation, dihedrals, chain_idx, atom_mask, structure_context_mask=None,
                      sequence_context_mask=None, prefix="residue_context_embedding."):
    """ResidueEmbedding.forward  (diffab_pytorch.py:81-183)"""
    B, L, A, _ = xyz.shape
    if sequence_context_mask is not None:
        seq_idx = torch.where(sequence_context_mask.bool(), seq_idx, torch.full_like(seq_idx, AA_UNK))
    aa = sd[prefix + "amino_acid_type_embedding.weight"][seq_idx]
    rel = xyz - xyz[:, :, 1:2, :]
    local = torch.einsum("blji,blaj->blai", orientation, rel) * atom_mask[:, :, :, None]  # O^T (x - x_CA)
    onehot = torch.nn.functional.one_hot(seq_idx, 21).to(xyz.dtype)  # (B,L,21)
    coord = (onehot[:, :, :, None, None] * local[:, :, None, :, :]).reshape(B, L, 21 * A * 3)
    if structure_context_mask is not None:
        coord = coord * structure_context_mask[:, :, None]
    dih = angular_encoding(dihedrals, 3)
    if structure_context_mask is not None:
        dmask = torch.stack([torch.roll(structure_context_mask, shifts=s, dims=1) for s in range(-1, 1)]).all(dim=0)
        dih = dih * dmask[:, :, None]
    chain = sd[prefix + "chain_embedding.weight"][chain_idx]
    x = torch.cat([aa, coord, dih, chain], dim=-1)
    return _mlp(x, sd, prefix + "mlp.", (0, 2, 4, 6))


def pair_embedding(sd, seq_idx, distmat, dihedrals, residue_idx, chain_idx, atom_mask, structure_context_mask=None,
                   sequence_context_mask=None, max_dist=32, prefix="pair_context_embedding."):
    """PairEmbedding.forward  (diffab_pytorch.py:220-312).  The structure-context mask is multiplied into `distmat` only
    after its last use (:295-301), so it does not reach the output: accepted and ignored, like the reference's effect."""
    B, L = seq_idx.shape
    A = atom_mask.shape[-1]
    am_pair = (atom_mask[:, :, None, :, None] * atom_mask[:, None, :, None, :]).reshape(B, L, L, A * A)
    rmask = atom_mask[:, :, 1]
    rmask_pair = rmask[:, :, None] * rmask[:, None, :]
    if sequence_context_mask is not None:
        seq_idx = torch.where(sequence_context_mask.bool(), seq_idx, torch.full_like(seq_idx, AA_UNK))
    sp = seq_idx[:, :, None] * 21 + seq_idx[:, None, :]
    sp_feat = sd[prefix + "aa_pair_type_embedding.weight"][sp]
    same_chain = chain_idx[:, :, None] * chain_idx[:, None, :]  # a product (:279)
    rel = (residue_idx[:, :, None] - residue_idx[:, None, :]).clamp(-max_dist, max_dist)
    rel_feat = sd[prefix + "relpos_embedding.weight"][rel + max_dist] * same_chain[:, :, :, None]
    coef = torch.nn.functional.softplus(sd[prefix + "pair2distcoef.weight"][sp])
    dm = torch.exp(-1 * coef * distmat.reshape(B, L, L, A * A) ** 2)
    dist_feat = torch.relu(_mlp(dm * am_pair, sd, prefix + "distance_embedding.", (0, 2)))
    dih = angular_encoding(dihedrals, 2)
    x = torch.cat([sp_feat, rel_feat.expand(B, L, L, -1), dist_feat, dih], dim=-1)
    return _mlp(x, sd, prefix + "mlp.", (0, 2, 4)) * rmask_pair[:, :, :, None]


def encode_context(sd, batch, generate_structure=True, generate_sequence=True, max_dist=32):
    """DiffAb.encode_context  (diffab_pytorch.py:680-724)"""
    ctx = batch["residue_mask"].bool() & (~batch["generation_mask"].bool())
    sm = ctx if generate_structure else None
    qm = ctx if generate_sequence else None
    res = residue_embedding(sd, batch["seq_idx"], batch["xyz"], batch["orientations"], batch["backbone_dihedrals"], batch["chain_idx"],
                            batch["atom_mask"], sm, qm)
    pair = pair_embedding(sd, batch["seq_idx"], batch["distmat"], batch["pairwise_dihedrals"], batch["residue_idx"], batch["chain_idx"],
                          batch["atom_mask"], sm, qm, max_dist)
    return res, pair


# --------------------------------------------------------------------------
# featurisation from coordinates        reference: data.py:75-82, preprocess_pdb.py:60-65 (protstruc calls)
# --------------------------------------------------------------------------
# protstruc is not in the reference tree: these are the geometric definitions the HIP kernels implement (include/diffab_hip.h,
# diffab_featurize_xyz), restated in float64.  PARITY UNPINNED against protstruc's own conventions.

def dihedral(p0, p1, p2, p3):
    """IUPAC dihedral of four points (..., 3): atan2(|b1| b0.(b1 x b2), (b0 x b1).(b1 x b2))."""
    b0, b1, b2 = p1 - p0, p2 - p1, p3 - p2
    n1, n2 = torch.cross(b0, b1, dim=-1), torch.cross(b1, b2, dim=-1)
    y = b1.norm(dim=-1) * (b0 * n2).sum(-1)
    x = (n1 * n2).sum(-1)
    return torch.atan2(y, x)


def featurize_xyz(xyz, chain_idx=None, residue_mask=None):
    """xyz (B,K,A,3), atoms N, CA, C in slots 0..2 -> orientations (rows = local axes, Gram-Schmidt on CA->C then CA->N), backbone
    (phi, psi, omega) + mask, pairwise (phi_ij, psi_ij) = ((C_i,N_j,CA_j,C_j), (N_i,CA_i,C_i,N_j))."""
    x = xyz.double()
    B, K = x.shape[:2]
    n, ca, c = x[:, :, 0], x[:, :, 1], x[:, :, 2]
    e1 = torch.nn.functional.normalize(c - ca, dim=-1)
    u = (n - ca) - ((n - ca) * e1).sum(-1, keepdim=True) * e1
    e2 = torch.nn.functional.normalize(u, dim=-1)
    e3 = torch.cross(e1, e2, dim=-1)
    O = torch.stack([e1, e2, e3], dim=-2)
    rm = torch.ones(B, K, dtype=torch.bool) if residue_mask is None else residue_mask.bool()
    ch = torch.zeros(B, K, dtype=torch.long) if chain_idx is None else chain_idx
    link = rm[:, :-1] & rm[:, 1:] & (ch[:, :-1] == ch[:, 1:])  # residues l and l+1 are consecutive members of one chain
    prev_ok = torch.cat([torch.zeros(B, 1, dtype=torch.bool), link], 1)
    next_ok = torch.cat([link, torch.zeros(B, 1, dtype=torch.bool)], 1)
    z3 = torch.zeros(B, 1, 3, dtype=x.dtype)
    c_prev = torch.cat([z3, c[:, :-1]], 1)
    n_next = torch.cat([n[:, 1:], z3], 1)
    ca_next = torch.cat([ca[:, 1:], z3], 1)
    phi = torch.where(prev_ok, dihedral(c_prev, n, ca, c), torch.zeros(B, K, dtype=x.dtype))
    psi = torch.where(next_ok, dihedral(n, ca, c, n_next), torch.zeros(B, K, dtype=x.dtype))
    omg = torch.where(next_ok, dihedral(ca, c, n_next, ca_next), torch.zeros(B, K, dtype=x.dtype))
    pphi = dihedral(c[:, :, None], n[:, None, :], ca[:, None, :], c[:, None, :])
    ppsi = dihedral(n[:, :, None], ca[:, :, None], c[:, :, None], n[:, None, :])
    return {"orientations": O, "backbone_dihedrals": torch.stack([phi, psi, omg], -1),
            "backbone_dihedrals_mask": torch.stack([prev_ok, next_ok, next_ok], -1),
            "pairwise_dihedrals": torch.stack([pphi.expand(B, K, K), ppsi.expand(B, K, K)], -1)}
