"""Debug: one reverse step both ways (patch-resident module kernel | per-layer launches) from the same state, the whole workspace
compared word for word - do the forms differ, and in which clusters of words?  (How the packed-fp32 hazard of
profiles/r05_pk_opsel_hazard.md was found: rows 32 rw + 15 of the projection buffer, x component, block 0.)
usage: persistent_diff_ws.py [trials] ; env NL = IPA layers of the model (default 1: the first differing buffer is the culprit)"""
import ctypes as C
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "diffab-pytorch_amd"))
import torch  # noqa: E402

from diffab_pytorch import DiffAb, _hip, synthetic as syn  # noqa: E402

TRIALS = int(sys.argv[1]) if len(sys.argv) > 1 else 8
B, K = 256, 128
lib = _hip.lib()
dims = dict(syn.BENCH_DIMS)
torch.manual_seed(0)
NLAYERS = int(os.environ.get("NL", "1"))
model = DiffAb(dims["D"], dims["C"], NLAYERS, dims["DS"], dims["PQ"], dims["PV"], dims["H"]).cuda()
inp = {k: v.cuda() for k, v in syn.patches(B, K, dims, seed=0, coord_sigma=10.0).items()}
hd, w = model.denoiser.hip_dims(B, K), model.denoiser.hip_weights()
sd_dev, tab = model._sched_on_device(), model._reverse_so3().struct()
ws = _hip.workspace(lib.diffab_sample_workspace_bytes(C.byref(hd)))
gm, rc, pc = inp["generation_mask"], inp["res_context_emb"], inp["pair_context_emb"]
if os.environ.get("STAGGER"):
    tk, cl = (int(v) for v in os.environ["STAGGER"].split(","))
    lib.diffab_debug_set_module_stagger(tk, cl)


def run(flags):
    seq, x, O = inp["seq_idx"].clone(), inp["translations"].clone(), inp["orientations"].clone()
    _hip.check(lib.diffab_sample_init(_hip.ptr(seq), _hip.ptr(x), _hip.ptr(O), _hip.ptr(gm), 2024, 0, B, K, model.T, _hip.stream_ptr()), "init")
    ws.zero_()
    _hip.check(lib.diffab_sample_loop(C.byref(hd), C.byref(w.struct), C.byref(sd_dev.struct), C.byref(tab), _hip.ptr(seq), _hip.ptr(x),
                                      _hip.ptr(O), _hip.ptr(rc), _hip.ptr(pc), _hip.ptr(gm), 2024, 0, model.T, model.T - 1, _hip.ptr(ws),
                                      ws.numel(), flags, _hip.stream_ptr()), "sample_loop")
    torch.cuda.synchronize()
    return ws[: ws.numel() // 4 * 4].view(torch.int32).clone()


ref = run(_hip.FLAG_MULTI_LAUNCH)
print(f"workspace {ref.numel()} words; rows {B * K}: a projection block is {B * K * 1344} words, a feature block {B * K * 1024}", flush=True)
nbad = 0
for trial in range(TRIALS):
    got = run(_hip.FLAG_PERSISTENT_MODULE)
    idx = (got != ref).nonzero().flatten()
    if idx.numel() == 0:
        print(f"trial {trial}: identical", flush=True)
        continue
    nbad += 1
    idx = idx.cpu()
    cuts = ((idx[1:] - idx[:-1]) > 20000).nonzero().flatten() + 1  # clusters: gaps wider than a few rows
    starts = [0] + cuts.tolist()
    ends = cuts.tolist() + [idx.numel()]
    print(f"trial {trial}: {idx.numel()} differing words in {len(starts)} clusters", flush=True)
    for s, e in list(zip(starts, ends))[:40]:
        a, b = int(idx[s]), int(idx[e - 1])
        gf = got.view(torch.float32)[idx[s:e].cuda()][:4].tolist()
        rf = ref.view(torch.float32)[idx[s:e].cuda()][:4].tolist()
        print(f"   words {a}..{b} ({e - s} differ); first: persistent {gf} per-layer {rf}", flush=True)
print(f"{nbad} of {TRIALS} trials differ")
