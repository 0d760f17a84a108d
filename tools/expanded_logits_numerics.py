"""CPU numerics study for the attention kernel's phase 1 on the bf16 matrix cores (round 3).

Question: can the point-distance logits of InvariantPointAttentionLayer.forward (reference diffab_pytorch.py:426-436) run as a
split-precision bilinear form without losing the 1e-4 bar?  Form studied (gq_p = t_i + a_p, gk_p = t_j + b_p, D = t_i - t_j,
t' = t - patch centroid; terms that depend on (i, h) only are dropped: softmax over j is invariant to them):

    sum_p |gq_p - gk_p|^2  ~  P |D|^2  [direct differences, fp32 VALU, shared by the heads]
                             + beta_j + 2 t'_j . w_j                      [per key, rides in a slot against a 1.0]
                             - 2 (u_i . t'_j + t'_i . w_j + sum_p a_p . b_p)   [30-dim dot product -> one 16x16x32 k-step]

with u = sum_p a_p, w = sum_p b_p, beta = sum_p |b_p|^2; every slot of both operands split into three bf16 planes, six partial
products, fp32 accumulation.  The script compares, against the float64 oracle, (a) the fp32 oracle itself, (b) this form, (c) the
plain expanded form |gq'|^2 + |gk'|^2 - 2 gq'.gk' with the centroid subtracted - on the benchmark-geometry inputs, with a 150 A
offset, and on a drifted two-cluster geometry.  Run: python tools/expanded_logits_numerics.py
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "diffab-pytorch_amd"))
import diffab_oracle as orc  # noqa: E402
from diffab_pytorch import synthetic as syn  # noqa: E402


def split3(x):
    h = x.to(torch.bfloat16).to(torch.float32)
    m = (x - h).to(torch.bfloat16).to(torch.float32)
    l = (x - h - m).to(torch.bfloat16).to(torch.float32)
    return h, m, l


def dot6(A, Bm):
    """sum_k A[..., i, k] B[..., j, k] as six bf16 partial products with fp32 accumulation (smallest terms first)."""
    a = split3(A)
    b = split3(Bm)
    acc = torch.zeros(A.shape[:-1] + (Bm.shape[-2],), dtype=torch.float32)
    for ta, tb in ((1, 1), (2, 0), (0, 2), (1, 0), (0, 1), (0, 0)):
        acc = acc + torch.einsum("...ik,...jk->...ij", a[ta], b[tb])
    return acc


def ipa_layer_variant(x, e, R, t, sd, prefix, H, mode):
    g = lambda name: sd[prefix + name].to(x.dtype)
    B, K, D = x.shape
    Wqs, Wks, Wvs = g("to_q_scalar.weight"), g("to_k_scalar.weight"), g("to_v_scalar.weight")
    Wqp, Wkp, Wvp = g("to_q_point.weight"), g("to_k_point.weight"), g("to_v_point.weight")
    Wb, gamma = g("to_pair_bias.weight"), g("gamma")
    Wo, bo = g("to_out.weight"), g("to_out.bias")
    ds = Wqs.shape[0] // H
    Pq = Wqp.shape[0] // (3 * H)
    heads = lambda y: y.view(B, K, H, -1).permute(0, 2, 1, 3)
    qs, ks, vs = heads(x @ Wqs.T), heads(x @ Wks.T), heads(x @ Wvs.T)
    points = lambda y, P: y.view(B, K, H, P, 3).permute(0, 2, 1, 3, 4)
    rot = lambda p: torch.einsum("bhlpk,blkc->bhlpc", p, R)  # local offsets in the global orientation (no translation)
    a, b = rot(points(x @ Wqp.T, Pq)), rot(points(x @ Wkp.T, Pq))
    vp = orc.to_global(points(x @ Wvp.T, Pq), R, t)
    scale_s = ds ** -0.5
    coef = (-0.5 * (4.5 * Pq) ** -0.5 * gamma).view(1, H, 1, 1)
    tc = t - t.mean(dim=1, keepdim=True)  # (B, K, 3)
    if mode == "direct":
        qp, kp = a + t[:, None, :, None, :], b + t[:, None, :, None, :]
        diff = qp[:, :, :, None] - kp[:, :, None, :]
        lg = torch.einsum("bhid,bhjd->bhij", qs, ks) * scale_s + coef * (diff ** 2).sum(-1).sum(-1)
    elif mode == "bilinear":
        Dij = t[:, :, None, :] - t[:, None, :, :]
        d2 = (Dij ** 2).sum(-1)[:, None]  # (B, 1, K, K) direct
        u, w = a.sum(3), b.sum(3)  # (B, H, K, 3)
        beta = (b ** 2).sum(-1).sum(-1)  # (B, H, K)
        tcb = tc[:, None].expand(B, H, K, 3)
        ck = coef[..., 0] * (beta + 2.0 * (tcb * w).sum(-1))  # (B, H, K)
        c2 = -2.0 * coef[..., 0:1]  # (1, H, 1, 1)
        one = torch.ones(B, H, K, 1)
        QA = torch.cat([qs * scale_s, c2 * u, one, c2 * tcb, c2 * a.reshape(B, H, K, -1)], dim=-1)
        KB = torch.cat([ks, tcb, ck[..., None], w, b.reshape(B, H, K, -1)], dim=-1)
        lg = dot6(QA, KB) + coef * (Pq * d2)
    elif mode == "expanded":
        qp, kp = (a + tc[:, None, :, None, :]).reshape(B, H, K, -1), (b + tc[:, None, :, None, :]).reshape(B, H, K, -1)
        ck = coef[..., 0] * (kp ** 2).sum(-1)
        c2 = -2.0 * coef[..., 0:1]
        one = torch.ones(B, H, K, 1)
        QA = torch.cat([qs * scale_s, one, c2 * qp], dim=-1)
        KB = torch.cat([ks, ck[..., None], kp], dim=-1)
        lg = dot6(QA, KB)
    else:
        raise ValueError(mode)
    bias = (e @ Wb.T).permute(0, 3, 1, 2)
    attn = ((3 ** -0.5) * (lg + bias)).softmax(dim=-1)
    o_s = torch.einsum("bhij,bhjd->bhid", attn, vs).permute(0, 2, 1, 3).reshape(B, K, -1)
    o_e = torch.einsum("bhij,bijc->bhic", attn, e).permute(0, 2, 1, 3).reshape(B, K, -1)
    o_l = orc.to_local(torch.einsum("bhij,bhjpc->bhipc", attn, vp), R, t)
    o_n = o_l.norm(dim=-1)
    feat = torch.cat([o_s, o_e, o_l.permute(0, 2, 1, 3, 4).reshape(B, K, -1), o_n.permute(0, 2, 1, 3).reshape(B, K, -1)], dim=-1)
    return feat @ Wo.T + bo


def denoise(sd, p, beta, dims, mode, dtype):
    cast = lambda v: v.to(dtype) if v.is_floating_point() else v
    sdd = {k: cast(v) for k, v in sd.items()}
    if mode == "oracle":
        return orc.denoiser(sdd, p["seq_idx"], cast(p["translations"]), cast(p["orientations"]), cast(p["res_context_emb"]),
                            cast(p["pair_context_emb"]), beta.to(dtype), dims["NL"], dims["H"])
    saved = orc.ipa_layer
    orc.ipa_layer = lambda x, e, R, t, sd_, prefix, H, return_attn=False: ipa_layer_variant(x, e, R, t, sd_, prefix, H, mode)
    try:
        return orc.denoiser(sdd, p["seq_idx"], cast(p["translations"]), cast(p["orientations"]), cast(p["res_context_emb"]),
                            cast(p["pair_context_emb"]), beta.to(dtype), dims["NL"], dims["H"])
    finally:
        orc.ipa_layer = saved


def maxrel(a, b):
    return float((a.double() - b.double()).abs().max() / b.double().abs().max())


def main():
    torch.manual_seed(0)
    dims = syn.BENCH_DIMS
    sd = syn.denoiser_state_dict(dims, seed=0)
    cases = []
    for name, sigma, off in (("wide", 10.0, 0.0), ("tight", 2.0, 0.0), ("offset150", 10.0, 150.0)):
        p = syn.patches(2, 128, dims, seed=3, coord_sigma=sigma)
        p["translations"] = p["translations"] + off
        cases.append((name, p))
    p = syn.patches(2, 128, dims, seed=5, coord_sigma=6.0)  # two clusters 5 000 A apart (an untrained sampler's drift)
    p["translations"][:, 64:] += 5000.0
    cases.append(("clusters5000", p))
    beta = torch.tensor([0.03, 0.4])
    print(f"{'case':14s} {'variant':10s} {'aa_logits':>10s} {'eps':>10s} {'res_emb':>10s}   (max-rel vs float64 oracle)")
    for name, p in cases:
        ref = denoise(sd, p, beta, dims, "oracle", torch.float64)
        for mode in ("oracle", "direct", "bilinear", "expanded"):
            out = denoise(sd, p, beta, dims, mode, torch.float32)
            print(f"{name:14s} {mode:10s} {maxrel(out['aa_logits'], ref['aa_logits']):10.2e} "
                  f"{maxrel(out['translations_eps'], ref['translations_eps']):10.2e} {maxrel(out['res_emb'], ref['res_emb']):10.2e}")


if __name__ == "__main__":
    main()
