"""Which part of a reverse step differs under a busy second queue?  Repeats one piece of work with identical inputs on stream 1 while stream 2
runs the same kind of work on other patches; every repetition must be bitwise the solo result.
usage: sampler_race_check.py MODE [B] [reps]     MODE: denoise | denoise_planes | denoise_fp32gemm | sample1 | sample1_pairf32 | sample1_fp32gemm | sample5"""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "diffab-pytorch_amd"))
import torch  # noqa: E402

from diffab_pytorch import DiffAb, _hip, synthetic as syn  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "sample1"
B, K = int(sys.argv[2]) if len(sys.argv) > 2 else 64, 128
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 200
d = syn.BENCH_DIMS
torch.manual_seed(0)
model = DiffAb(d["D"], d["C"], d["NL"], d["DS"], d["PQ"], d["PV"], d["H"]).cuda().requires_grad_(False)
inp = {k: v.cuda() for k, v in syn.patches(2 * B, K, d, seed=3, coord_sigma=10.0).items()}
half = lambda k, i: inp[k][i * B:(i + 1) * B].contiguous()
beta = torch.full((B,), 0.01, device="cuda")
rm = torch.ones(B, K, dtype=torch.bool, device="cuda")
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
fl = {"denoise": 0, "denoise_planes": _hip.FLAG_PAIR_PLANES, "denoise_fp32gemm": _hip.FLAG_FP32_GEMM, "sample1": 0, "sample5": 0,
      "sample1_pairf32": _hip.FLAG_PAIR_F32, "sample1_fp32gemm": _hip.FLAG_FP32_GEMM}[mode]


def call(i):
    with torch.no_grad():
        if mode.startswith("denoise"):
            return model.denoiser(half("seq_idx", i), half("translations", i), half("orientations", i), half("res_context_emb", i),
                                 half("pair_context_emb", i), beta, half("generation_mask", i), rm, flags=fl)
        n = 5 if mode == "sample5" else 1
        return model.sample(half("seq_idx", i), half("translations", i), half("orientations", i), res_context_emb=half("res_context_emb", i),
                            pair_context_emb=half("pair_context_emb", i), generation_mask=half("generation_mask", i), seed=3, first_patch=i * B,
                            t_start=50, t_stop=50 - n, init=False, flags=fl)


dump = None
if os.environ.get("HF_DUMP"):  # ablation build -DHF_DUMP: heads_finish_dump_kernel writes 40 floats per row of stream 1's AND stream 2's calls
    dump = torch.zeros(B * K, 48, device="cuda")
    os.environ["DIFFAB_HF_DUMP"] = hex(dump.data_ptr())
    os.environ["DIFFAB_HF_DUMP_STREAM"] = hex(s1.cuda_stream)
torch.cuda.synchronize()
with torch.cuda.stream(s1):
    ref = {k: v.clone() for k, v in call(0).items()}
torch.cuda.synchronize()
ref_dump = dump.clone() if dump is not None else None
torch.cuda.synchronize()
keys = list(ref)
bad = torch.zeros(reps, len(keys), dtype=torch.int64, device="cuda")
keep = []
for r in range(reps):
    with torch.cuda.stream(s2):
        call(1)
    with torch.cuda.stream(s1):
        y = call(0)
        for j, k in enumerate(keys):
            bad[r, j] = (y[k] != ref[k]).sum()
        if r < 60:
            keep.append(y)
        if dump is not None:
            y["_dump"] = dump.clone()
torch.cuda.synchronize()
nb = bad.cpu()
print(f"mode {mode} B {B} reps {reps}")
for j, k in enumerate(keys):
    print(f"  {k:20s}: {int((nb[:, j] > 0).sum())} of {reps} repetitions differ (worst {int(nb[:, j].max())} of {ref[k].numel()} elements)")
shown = 0
for r, y in enumerate(keep):
    for k in keys:
        df = (y[k] != ref[k])
        if df.dim() > 2:
            df = df.flatten(2).any(-1)
        if df.any() and shown < 8:
            idx = df.nonzero()
            rows = (idx[:, 0] * K + idx[:, 1]).tolist()
            print(f"  rep {r} {k}: {len(rows)} rows; patches {sorted(set(x // K for x in rows))[:8]}; rows in patch {sorted(set(x % K for x in rows))[:24]}; "
                  f"max |diff| {float((y[k].double() - ref[k].double()).abs().max()):.3g}")
            shown += 1

if dump is not None:
    names = ["vx", "vy", "vz"] + [f"o{k}" for k in range(9)] + [f"ex{k}" for k in range(9)] + [f"res{k}" for k in range(9)] + ["tid", "xcc", "hwid"]
    for r, y in enumerate(keep):
        dd = y["_dump"]
        df = (dd[:, :30] != ref_dump[:, :30]).any(-1)
        if df.any():
            rows = df.nonzero().flatten().tolist()
            cols = (dd[:, :30] != ref_dump[:, :30]).any(0).nonzero().flatten().tolist()
            r0 = rows[0]
            print(f"  rep {r}: dump differs in {len(rows)} rows {rows[:4]}..{rows[-1]}; columns {[names[c] for c in cols]}")
            print(f"     row {r0}: got v {dd[r0, :3].tolist()} ref {ref_dump[r0, :3].tolist()}")
            print(f"     row {r0}: got o {dd[r0, 3:12].tolist()}\n              ref o {ref_dump[r0, 3:12].tolist()}")
            late, early = dd[:, :12], dd[:, 33:45]
            ne = (late != early).any(-1)
            print(f"     rows where the early copy of a loaded register != its late value: {int(ne.sum())} {ne.nonzero().flatten().tolist()[:20]}; "
                  f"columns {[names[c] for c in (late != early).any(0).nonzero().flatten().tolist()]}")
            print(f"     xcc got {dd[r0, 31].item()} ref {ref_dump[r0, 31].item()}  hwid got {dd[r0, 32].view(torch.int32).item():#x} ref {ref_dump[r0, 32].view(torch.int32).item():#x}")
