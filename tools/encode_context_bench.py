"""Timing of DiffAb.encode_context at the benchmark model: B patches of K residues, A=15 atoms;
with the materialised distance tensor (reference signature) and with distances taken from xyz inside the pair kernel.
usage: encode_context_bench.py [B] [K] [--backward] [--variant=N]
(--backward: forward + backward of both encoders from random cotangents, xyz form; --variant: diffab_debug_set_attn_variant, e.g. 64 =
the backward's 64-wide tail as separate launches)"""
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "diffab-pytorch_amd"))
import torch  # noqa: E402

from diffab_pytorch import DiffAb, _hip, synthetic as syn  # noqa: E402

BACKWARD = "--backward" in sys.argv
argv = [a for a in sys.argv[1:] if not a.startswith("--")]
B = int(argv[0]) if len(argv) > 0 else 32
K = int(argv[1]) if len(argv) > 1 else 128
for a_ in sys.argv[1:]:
    if a_.startswith("--variant="):
        _hip.lib().diffab_debug_set_attn_variant(int(a_.split("=")[1]))
d = syn.BENCH_DIMS
model = DiffAb(d["D"], d["C"], 1, d["DS"], d["PQ"], d["PV"], d["H"]).cuda()
model.load_state_dict(syn.context_state_dict(d["D"], d["C"], 15, 32, seed=1), strict=False)
cb = {k: v.cuda() for k, v in syn.context_batch(B, K, 15, seed=1).items()}
args = lambda dm: (cb["seq_idx"], cb["xyz"], cb["orientations"], cb["backbone_dihedrals"], dm, cb["pairwise_dihedrals"], cb["atom_mask"],
                   cb["chain_idx"], cb["residue_idx"], cb["generation_mask"], cb["residue_mask"])
for name, dm in (("distmat", cb["distmat"]), ("xyz", None)):  # (inference form: no graph, so no tape is written)
    with torch.no_grad():
        for _ in range(2):
            model.encode_context(*args(dm))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 5
    with torch.no_grad():
        for _ in range(n):
            model.encode_context(*args(dm))
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(f"encode_context[{name}] B={B} K={K}: {dt*1e3:.2f} ms = {dt*1e3/B:.4f} ms/patch; distmat stream {B*K*K*225*4/dt/1e9:.0f} GB/s-equivalent")

if BACKWARD:
    g = torch.Generator(device="cuda").manual_seed(3)
    c_res = torch.randn(B, K, d["D"], device="cuda", generator=g)
    c_pair = torch.randn(B, K, K, d["C"], device="cuda", generator=g)

    def fb():
        for p_ in model.parameters():
            p_.grad = None
        r_, p_ = model.encode_context(*args(None))
        torch.autograd.backward([r_, p_], [c_res, c_pair])

    for _ in range(2):
        fb()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 5
    for _ in range(n):
        fb()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(f"encode_context[xyz] forward + backward B={B} K={K}: {dt*1e3:.2f} ms = {dt*1e3/B:.4f} ms/patch")
