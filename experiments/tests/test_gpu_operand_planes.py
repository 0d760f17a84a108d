"""The operand-plane form of the IPA logits (csrc/proj_planes.hip + the B6L phase 1 of the attention kernel).

The projection kernel writes the query / key sides of InvariantPointAttentionLayer.forward (reference diffab_pytorch.py:391-436)
as three-plane bf16 MFMA operands of ONE 64-slot dot product per (head, query, key).  These tests decode the planes on the host
and check them slot by slot against a float64 restatement of the same algebra, and check that the dot product of the decoded
operands plus the direct |t_i - t_j|^2 term reproduces the oracle's logits up to the per-row constants softmax does not see.
"""
import ctypes as C
import os

import pytest
import torch

import diffab_oracle as orc
from conftest import maxrel
from diffab_pytorch import _hip, synthetic as syn

pytestmark = pytest.mark.gpu

EXP_LIB = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "experiments", "build", "libdiffab_hip.so")


@pytest.fixture(scope="module")
def hip():
    """The EXPERIMENTAL build of the library (experiments/build.sh): the operand-plane kernels are
    a measured, unadopted variant and are not part of the product library."""
    assert _hip.lib().diffab_device_ok() == 1
    if not os.path.exists(EXP_LIB):
        pytest.skip("experimental library not built (bash experiments/build.sh)")
    lib = C.CDLL(EXP_LIB)
    _PD, _fp, _sz = C.POINTER(_hip.Dims), C.c_void_p, C.c_size_t
    exp_symbols = {  # experiments/include/diffab_hip_experimental.h
        "diffab_debug_proj_planes_scratch_bytes": (_sz, [_PD]),
        "diffab_debug_proj_planes": (C.c_int, [_PD, C.POINTER(_hip.IpaLayerWeights), _fp, _fp, _fp, _fp, _fp, _fp, _sz, _fp]),
    }
    for name, (res, args) in exp_symbols.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    return lib


def decode_planes(qk, B, K):
    """(rows x 1536) floats -> float64 slots [side][B][H][K][64] (sum of the three bf16 planes) and the planes themselves."""
    H, nt = 8, K // 16
    raw = qk.view(torch.bfloat16).view(2, B, H, nt, 2, 3, 64, 8).double().cpu()  # side, b, h, tile, k-step, plane, lane, element
    v = raw.sum(dim=5)  # side, b, h, tile, ks, lane, el
    v = v.view(2, B, H, nt, 2, 4, 16, 8)  # lane = l15 + 16 g -> (g, l15)
    v = v.permute(0, 1, 2, 3, 6, 4, 5, 7)  # side, b, h, tile, l15, ks, g, el
    return v.reshape(2, B, H, K, 64), raw  # slot = 32 ks + 8 g + el


def expected_slots(sd, x, R, t, H=8, P=8, DS=32):
    """float64 restatement of the slot table in the header of csrc/proj_planes.hip."""
    g = lambda n: sd[n].double()
    B, K, _ = x.shape
    x, R, t = x.double(), R.double(), t.double()
    heads = lambda y: y.view(B, K, H, -1).permute(0, 2, 1, 3)
    qs, ks = heads(x @ g("to_q_scalar.weight").T), heads(x @ g("to_k_scalar.weight").T)
    pts = lambda y: y.view(B, K, H, P, 3).permute(0, 2, 1, 3, 4)
    rot = lambda p: torch.einsum("bhlpk,blkc->bhlpc", p, R)
    a, b = rot(pts(x @ g("to_q_point.weight").T)), rot(pts(x @ g("to_k_point.weight").T))
    gam = g("gamma").view(1, H, 1)
    coef = -0.5 * (4.5 * P) ** -0.5 * gam
    c2 = (-2.0 * coef)[..., None]
    tc = (t - t.mean(dim=1, keepdim=True))[:, None].expand(B, H, K, 3)
    u, w = a.sum(3), b.sum(3)
    ck = coef * ((b ** 2).sum(-1).sum(-1) + 2.0 * (tc * w).sum(-1))
    one, zero = torch.ones(B, H, K, 1, dtype=torch.float64), torch.zeros(B, H, K, 1, dtype=torch.float64)

    def geometry(pts_, e1, e2):  # lane quarter g: point g (3), E1_g | point 4 + g (3), E2_g
        out = []
        for gq in range(4):
            out += [pts_[:, :, :, gq], e1[..., gq:gq + 1], pts_[:, :, :, 4 + gq], e2[..., gq:gq + 1]]
        return torch.cat(out, dim=-1)

    QA = torch.cat([qs * DS ** -0.5, geometry(c2[..., None] * a, torch.cat([c2 * u, one], -1), torch.cat([c2 * tc, zero], -1))], dim=-1)
    KB = torch.cat([ks, geometry(b, torch.cat([tc, ck[..., None]], -1), torch.cat([w, zero], -1))], dim=-1)
    return QA, KB, coef


@pytest.mark.parametrize("B,K,sigma,offset", [(2, 128, 8.0, 0.0), (3, 64, 5.0, 0.0), (1, 256, 10.0, 150.0)])
def test_operand_planes_vs_float64(hip, B, K, sigma, offset):
    from diffab_pytorch.diffab_pytorch import InvariantPointAttentionLayer, _named

    d = syn.BENCH_DIMS
    torch.manual_seed(1)
    layer = InvariantPointAttentionLayer(d["D"], d["C"], d["DS"], d["PQ"], d["PV"], d["H"])
    with torch.no_grad():
        layer.gamma.copy_(layer.gamma + 0.1 * torch.randn(8))
    layer = layer.cuda().requires_grad_(False)
    sd = {k: v.detach().cpu() for k, v in layer.state_dict().items()}
    inp = syn.patches(B, K, d, seed=11, coord_sigma=sigma)
    x, R, t = inp["res_context_emb"], inp["orientations"], inp["translations"] + offset
    rows = B * K
    dims = _hip.make_dims(B, K, d["D"], d["C"], d["H"], d["DS"], d["PQ"], d["PV"], 1)
    keep = []
    w = _hip.ipa_layer_weights(_named(layer), keep)
    scratch = _hip.workspace(hip.diffab_debug_proj_planes_scratch_bytes(C.byref(dims)))
    qk = torch.zeros(rows * 1536, dtype=torch.float32, device="cuda")
    proj = torch.full((rows, 1344), float("nan"), dtype=torch.float32, device="cuda")
    xd, Rd, td = _hip.dev_f32(x), _hip.dev_f32(R), _hip.dev_f32(t)
    assert 0 == hip.diffab_debug_proj_planes(C.byref(dims), C.byref(w), _hip.ptr(xd), _hip.ptr(Rd), _hip.ptr(td), _hip.ptr(qk), _hip.ptr(proj),
                                            _hip.ptr(scratch), scratch.numel(), _hip.stream_ptr())
    torch.cuda.synchronize()
    got, raw = decode_planes(qk, B, K)
    QA, KB, coef = expected_slots(sd, x, R, t)
    # the three planes of a slot are an exact split: hi = bf16(v), so |mid| <= ulp(hi) / 2 etc. - spot-check the magnitudes
    assert torch.isfinite(got).all()
    assert (raw[:, :, :, :, :, 1].abs() <= raw[:, :, :, :, :, 0].abs() * 2.0 ** -7 + 1e-30).all()
    # slot by slot: scalar slots (one fp32 product + a scale), geometry slots (rotation + sums in fp32)
    for side, want, name in ((0, QA, "query"), (1, KB, "key")):
        err = (got[side] - want).abs()
        scale = want.abs().amax(dim=(0, 2), keepdim=True).clamp_min(1e-30)  # per (head, slot) maximum
        assert float((err / scale).max()) < 2e-6, (name, float((err / scale).max()), int((err / scale).argmax()))
    # the dot product of the decoded operands + the direct distance term = the oracle's logits up to a constant per (b, h, i)
    Dij = (t.double()[:, :, None] - t.double()[:, None]).pow(2).sum(-1)[:, None]
    L = torch.einsum("bhik,bhjk->bhij", got[0], got[1]) + 8.0 * coef[..., None] * Dij
    x64, R64, t64 = x.double(), R.double(), t.double()
    g64 = lambda n: sd[n].double()
    heads = lambda y: y.view(B, K, 8, -1).permute(0, 2, 1, 3)
    pts = lambda y: orc.to_global(y.view(B, K, 8, 8, 3).permute(0, 2, 1, 3, 4), R64, t64)
    qp, kp = pts(x64 @ g64("to_q_point.weight").T), pts(x64 @ g64("to_k_point.weight").T)
    ref = torch.einsum("bhid,bhjd->bhij", heads(x64 @ g64("to_q_scalar.weight").T), heads(x64 @ g64("to_k_scalar.weight").T)) * 32 ** -0.5
    ref = ref + coef[..., None] * ((qp[:, :, :, None] - kp[:, :, None]) ** 2).sum(-1).sum(-1)
    dlt = L - ref
    dlt = dlt - dlt.mean(dim=-1, keepdim=True)
    assert float(dlt.abs().max()) < 2e-4 * max(1.0, offset / 10.0), float(dlt.abs().max())  # logits span thousands; softmax sees differences
    # value side: the fp32 projection buffer's v_s and global value-point columns
    want_vs = x64 @ g64("to_v_scalar.weight").T
    want_gv = orc.to_global((x64 @ g64("to_v_point.weight").T).view(B, K, 8, 8, 3).permute(0, 2, 1, 3, 4), R64, t64).permute(0, 2, 1, 3, 4).reshape(B, K, 192)
    pc = proj.view(B, K, 1344).cpu()
    assert maxrel(pc[..., 512:768], want_vs) < 2e-6
    assert maxrel(pc[..., 1152:1344], want_gv) < 2e-6
    assert torch.isnan(pc[..., :512]).all() and torch.isnan(pc[..., 768:1152]).all()  # nothing else is written


def test_queue_driven_attention_and_to_out_is_bitwise_the_default(hip):
    """EXPERIMENTAL build, DIFFAB_ATTN_QUEUE=1: attention + to_out of a layer as ONE persistent launch over per-XCD work queues (the QUEUE
    form of ipa_attn_fast_kernel, profiles/r03_lockstep.md section 5).  Same arithmetic per work item, so the layer output must be
    bitwise the default two launches.  The switch is read from the environment once per process: tools/queue_check.py runs the variants in
    child processes and compares the saved outputs."""
    import subprocess
    import sys

    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(repo, "tools", "queue_check.py"), "16"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "variant 1 bitwise equal to the default: True" in out.stdout and "variant 2 bitwise equal to the default: True" in out.stdout, out.stdout
