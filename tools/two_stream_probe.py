"""Experiment: the reverse sampler of 256 patches as ONE stream of launches vs TWO half-batch pipelines on two streams (the
attention kernel of one half overlapping the dense kernels of the other: do the CUs' stream phases desynchronise, are the
launch boundaries hidden?).  Prints ms per step of both forms.  usage: two_stream_probe.py [steps] [nsplit]"""
import ctypes as C
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "diffab-pytorch_amd"))
import torch  # noqa: E402

from diffab_pytorch import DiffAb, _hip, synthetic as syn  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
NS = int(sys.argv[2]) if len(sys.argv) > 2 else 2
lib = _hip.lib()
dims = dict(syn.BENCH_DIMS)
B, K = 256, 128
torch.manual_seed(0)
model = DiffAb(dims["D"], dims["C"], dims["NL"], dims["DS"], dims["PQ"], dims["PV"], dims["H"]).cuda()
inp = syn.patches(B, K, dims, seed=0, coord_sigma=10.0)
dev = {k: v.cuda() for k, v in inp.items()}
w = model.denoiser.hip_weights()
sd_dev = model._sched_on_device()
tab = model._reverse_so3().struct()
seed = 2024


def make(lo, hi):
    n = hi - lo
    hd = model.denoiser.hip_dims(n, K)
    ws = _hip.workspace(lib.diffab_sample_workspace_bytes(C.byref(hd)))
    st = {k: dev[k][lo:hi].clone() for k in ("seq_idx", "translations", "orientations")}
    ctx = {k: dev[k][lo:hi].contiguous() for k in ("generation_mask", "res_context_emb", "pair_context_emb")}
    return dict(hd=hd, ws=ws, st=st, ctx=ctx, lo=lo, n=n)


def run(p, t_hi, t_lo, stream):
    with torch.cuda.stream(stream):
        _hip.check(lib.diffab_sample_loop(C.byref(p["hd"]), C.byref(w.struct), C.byref(sd_dev.struct), C.byref(tab), _hip.ptr(p["st"]["seq_idx"]),
                                          _hip.ptr(p["st"]["translations"]), _hip.ptr(p["st"]["orientations"]), _hip.ptr(p["ctx"]["res_context_emb"]),
                                          _hip.ptr(p["ctx"]["pair_context_emb"]), _hip.ptr(p["ctx"]["generation_mask"]), seed, p["lo"], t_hi, t_lo,
                                          _hip.ptr(p["ws"]), p["ws"].numel(), 0, _hip.stream_ptr()), "sample_loop")


whole = make(0, B)
parts = [make(i * B // NS, (i + 1) * B // NS) for i in range(NS)]
streams = [torch.cuda.Stream() for _ in range(NS)]
s0 = torch.cuda.Stream()
torch.cuda.synchronize()  # the clones above ran on the default stream
for label, work in (("one stream, 256 patches", [(whole, s0)]), (f"{NS} streams, {B // NS} patches each", list(zip(parts, streams)))):
    for p, s in work:
        run(p, 100, 95, s)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for p, s in work:
        run(p, 95, 95 - steps, s)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{label}: {dt / steps * 1e3:.3f} ms per step of {B} patches ({B * K * steps / dt / 1e6:.2f} M residue-steps/s)")
a = torch.cat([p["st"]["translations"] for p in parts])
print("bitwise equal to the single-stream trajectory:", bool((a == whole["st"]["translations"]).all()))
seq_parts = [make(i * B // NS, (i + 1) * B // NS) for i in range(NS)]
torch.cuda.synchronize()
for p in seq_parts:  # the same shards one after the other on ONE stream
    run(p, 100, 95, s0)
    run(p, 95, 95 - steps, s0)
torch.cuda.synchronize()
b = torch.cat([p["st"]["translations"] for p in seq_parts])
print("shards run one after the other == whole batch:", bool((b == whole["st"]["translations"]).all()), "| == concurrent shards:", bool((a == b).all()),
      "| max |diff| concurrent vs sequential:", float((a - b).abs().max()))
