"""Two reverse-sampling calls on two streams at the same time vs the same calls one after the other (must be bitwise equal)."""
import ctypes as C
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "diffab-pytorch_amd"))
import torch  # noqa: E402

from diffab_pytorch import DiffAb, _hip, synthetic as syn  # noqa: E402

lib = _hip.lib()
dims = dict(syn.BENCH_DIMS)
B, K, steps = int(sys.argv[1]) if len(sys.argv) > 1 else 64, 128, int(sys.argv[2]) if len(sys.argv) > 2 else 3
torch.manual_seed(0)
model = DiffAb(dims["D"], dims["C"], dims["NL"], dims["DS"], dims["PQ"], dims["PV"], dims["H"]).cuda()
inp = syn.patches(2 * B, K, dims, seed=0, coord_sigma=10.0)
dev = {k: v.cuda() for k, v in inp.items()}
w = model.denoiser.hip_weights()
sd_dev = model._sched_on_device()
tab = model._reverse_so3().struct()


def make(lo, hi):
    hd = model.denoiser.hip_dims(hi - lo, K)
    ws = _hip.workspace(lib.diffab_sample_workspace_bytes(C.byref(hd)))
    st = {k: dev[k][lo:hi].clone() for k in ("seq_idx", "translations", "orientations")}
    ctx = {k: dev[k][lo:hi].clone() for k in ("generation_mask", "res_context_emb", "pair_context_emb")}
    return dict(hd=hd, ws=ws, st=st, ctx=ctx, lo=lo)


def run(p, stream, flags):
    with torch.cuda.stream(stream):
        _hip.check(lib.diffab_sample_loop(C.byref(p["hd"]), C.byref(w.struct), C.byref(sd_dev.struct), C.byref(tab), _hip.ptr(p["st"]["seq_idx"]),
                                          _hip.ptr(p["st"]["translations"]), _hip.ptr(p["st"]["orientations"]), _hip.ptr(p["ctx"]["res_context_emb"]),
                                          _hip.ptr(p["ctx"]["pair_context_emb"]), _hip.ptr(p["ctx"]["generation_mask"]), 7, p["lo"], 100, 100 - steps,
                                          _hip.ptr(p["ws"]), p["ws"].numel(), flags, _hip.stream_ptr()), "sample_loop")


s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
for name, flags in (("default", 0), ("default again", 0), ("pair f32", _hip.FLAG_PAIR_F32), ("fp32 gemm", _hip.FLAG_FP32_GEMM),
                    ("pair f32 + fp32 gemm", _hip.FLAG_PAIR_F32 | _hip.FLAG_FP32_GEMM)):
    seq_ = [make(0, B), make(B, 2 * B)]
    con_ = [make(0, B), make(B, 2 * B)]
    torch.cuda.synchronize()
    for p in seq_:
        run(p, s1, flags)
        torch.cuda.synchronize()
    run(con_[0], s1, flags)
    run(con_[1], s2, flags)
    torch.cuda.synchronize()
    msg = []
    for k in ("seq_idx", "translations", "orientations"):
        a = torch.cat([p["st"][k] for p in seq_]).double()
        b = torch.cat([p["st"][k] for p in con_]).double()
        msg.append(f"{k}: {int((a != b).sum())} of {a.numel()} differ (max {float((a - b).abs().max()):.3g})")
    print(f"{name:22s} " + " | ".join(msg))
