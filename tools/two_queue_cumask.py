"""does the two-queue wrong result (profiles/r04_two_queue.md) need the two pipelines to SHARE compute units?
Two sampler pipelines (B patches each, guard off) run at the same time on two streams created with hipExtStreamCreateWithCUMask:
  plain      two ordinary streams (the failing case of round 4)
  full       both streams with an all-ones CU mask (control for the Ext API itself)
  halves     stream 1 on CUs [0, n/2), stream 2 on CUs [n/2, n)      - disjoint
  interleave stream 1 on even mask bits, stream 2 on odd mask bits   - disjoint, different physical pattern
  victim16   stream 1 on mask bits [0, 16), stream 2 on the other bits - disjoint, small victim set
  blocksG    blocks of G consecutive mask bits alternate between the streams (G = 2 .. 32): the granularity of what must not be shared
Every concurrent repetition is compared bit for bit with the same two calls run one after the other.
usage: two_queue_cumask.py [B] [steps] [repetitions]"""
import ctypes as C
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "diffab-pytorch_amd"))
import torch  # noqa: E402

from diffab_pytorch import DiffAb, _hip, synthetic as syn  # noqa: E402

lib = _hip.lib()
hip = C.CDLL("libamdhip64.so")
dims = dict(syn.BENCH_DIMS)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 12
K = 128
torch.manual_seed(0)
model = DiffAb(dims["D"], dims["C"], dims["NL"], dims["DS"], dims["PQ"], dims["PV"], dims["H"]).cuda()
inp = syn.patches(2 * B, K, dims, seed=0, coord_sigma=10.0)
dev = {k: v.cuda() for k, v in inp.items()}
w = model.denoiser.hip_weights()
sd_dev = model._sched_on_device()
tab = model._reverse_so3().struct()
ncu = torch.cuda.get_device_properties(0).multi_processor_count
nwords = (ncu + 31) // 32


def mask_stream(bits):
    """a HIP stream restricted to the CUs whose mask bit is set (None: an ordinary non-blocking stream)"""
    st = C.c_void_p()
    if bits is None:
        rc = hip.hipStreamCreateWithFlags(C.byref(st), 1)  # hipStreamNonBlocking
    else:
        words = (C.c_uint32 * nwords)(*[sum(1 << b for b in range(32) if (32 * wi + b) in bits) for wi in range(nwords)])
        rc = hip.hipExtStreamCreateWithCUMask(C.byref(st), nwords, words)
    assert rc == 0, rc
    return st


def make(lo, hi):
    hd = model.denoiser.hip_dims(hi - lo, K)
    ws = _hip.workspace(lib.diffab_sample_workspace_bytes(C.byref(hd)))
    st = {k: dev[k][lo:hi].clone() for k in ("seq_idx", "translations", "orientations")}
    ctx = {k: dev[k][lo:hi].clone() for k in ("generation_mask", "res_context_emb", "pair_context_emb")}
    return dict(hd=hd, ws=ws, st=st, ctx=ctx, lo=lo)


def run(p, stream):
    _hip.check(lib.diffab_sample_loop(C.byref(p["hd"]), C.byref(w.struct), C.byref(sd_dev.struct), C.byref(tab), _hip.ptr(p["st"]["seq_idx"]),
                                      _hip.ptr(p["st"]["translations"]), _hip.ptr(p["st"]["orientations"]), _hip.ptr(p["ctx"]["res_context_emb"]),
                                      _hip.ptr(p["ctx"]["pair_context_emb"]), _hip.ptr(p["ctx"]["generation_mask"]), 7, p["lo"], 100, 100 - steps,
                                      _hip.ptr(p["ws"]), p["ws"].numel(), 0, stream), "sample_loop")


def sync(st):
    assert hip.hipStreamSynchronize(st) == 0


allb = set(range(ncu))
configs = [
    ("plain", None, None),
    ("full", allb, allb),
    ("halves", set(range(ncu // 2)), set(range(ncu // 2, ncu))),
    ("interleave", set(range(0, ncu, 2)), set(range(1, ncu, 2))),
    ("victim16", set(range(16)), set(range(16, ncu))),
    # granularity of the sharing: blocks of g consecutive mask bits alternate between the two streams
    ("blocks2", {b for b in range(ncu) if (b // 2) % 2 == 0}, {b for b in range(ncu) if (b // 2) % 2 == 1}),
    ("blocks4", {b for b in range(ncu) if (b // 4) % 2 == 0}, {b for b in range(ncu) if (b // 4) % 2 == 1}),
    ("blocks8", {b for b in range(ncu) if (b // 8) % 2 == 0}, {b for b in range(ncu) if (b // 8) % 2 == 1}),
    ("blocks16", {b for b in range(ncu) if (b // 16) % 2 == 0}, {b for b in range(ncu) if (b // 16) % 2 == 1}),
    ("blocks32", {b for b in range(ncu) if (b // 32) % 2 == 0}, {b for b in range(ncu) if (b // 32) % 2 == 1}),
]
if os.environ.get("CONFIGS"):
    configs = [c for c in configs if c[0] in os.environ["CONFIGS"].split(",")]
lib.diffab_set_stream_guard(0)
ref = [make(0, B), make(B, 2 * B)]
torch.cuda.synchronize()
s0 = mask_stream(None)
for p in ref:  # the sequential reference: one pipeline at a time
    run(p, s0)
    sync(s0)
print(f"{ncu} CUs, {nwords} mask words; B = {B} patches per pipeline, {steps} steps, {reps} concurrent repetitions per configuration", flush=True)
for name, m1, m2 in configs:
    s1, s2 = mask_stream(m1), mask_stream(m2)
    bad_reps, bad_elems, bad_rows = 0, 0, 0
    for _ in range(reps):
        con = [make(0, B), make(B, 2 * B)]
        torch.cuda.synchronize()
        run(con[0], s1)
        run(con[1], s2)
        sync(s1)
        sync(s2)
        nbad = 0
        for k in ("seq_idx", "translations", "orientations"):
            a = torch.cat([p["st"][k] for p in ref])
            b = torch.cat([p["st"][k] for p in con])
            df = a != b
            nbad += int(df.sum())
            if k == "orientations":
                bad_rows += int(df.view(2 * B * K, -1).any(1).sum())
        bad_reps += nbad > 0
        bad_elems += nbad
    print(f"{name:11s}: {bad_reps} of {reps} repetitions differ from the sequential runs ({bad_elems} elements, {bad_rows} orientation rows)", flush=True)
    hip.hipStreamDestroy(s1)
    hip.hipStreamDestroy(s2)
lib.diffab_set_stream_guard(1)
