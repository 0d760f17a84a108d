"""Diagnostic: where does the fused IPA attention kernel spend its cycles?  Runs one IPA layer at the benchmark
geometry with s_memtime stamps enabled (never enabled in production) and prints per-phase shader-cycle averages."""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "diffab-pytorch_amd"))
import torch  # noqa: E402

from diffab_pytorch import _hip, synthetic as syn  # noqa: E402
from diffab_pytorch.diffab_pytorch import InvariantPointAttentionLayer  # noqa: E402

B, K = int(sys.argv[1]) if len(sys.argv) > 1 else 256, int(sys.argv[2]) if len(sys.argv) > 2 else 128
lib = _hip.lib()
d = syn.BENCH_DIMS
torch.manual_seed(0)
layer = InvariantPointAttentionLayer(d["D"], d["C"], d["DS"], d["PQ"], d["PV"], d["H"]).cuda().requires_grad_(False)  # inference kernels
g = torch.Generator(device="cuda").manual_seed(0)
x = torch.randn(B, K, d["D"], device="cuda", generator=g)
e = torch.randn(B, K, K, d["C"], device="cuda", generator=g)
t = 10 * torch.randn(B, K, 3, device="cuda", generator=g)
q = torch.nn.functional.normalize(torch.randn(B, K, 4, device="cuda", generator=g), dim=-1)
w_, x_, y_, z_ = q.unbind(-1)
R = torch.stack([1 - 2 * (y_ * y_ + z_ * z_), 2 * (x_ * y_ - z_ * w_), 2 * (x_ * z_ + y_ * w_), 2 * (x_ * y_ + z_ * w_),
                 1 - 2 * (x_ * x_ + z_ * z_), 2 * (y_ * z_ - x_ * w_), 2 * (x_ * z_ - y_ * w_), 2 * (y_ * z_ + x_ * w_),
                 1 - 2 * (x_ * x_ + y_ * y_)], -1).view(B, K, 3, 3).contiguous()
FLAGS = int(os.environ.get("ATTN_FLAGS", "0"))  # e.g. 32 = _hip.FLAG_PAIR_PLANES
for _ in range(3):
    layer(x, e, R, t, flags=FLAGS)
nwg = B * (K // 16)
stamps = torch.zeros(nwg * 8 * 8, dtype=torch.int64, device="cuda")
lib.diffab_debug_set_attn_stamps(_hip.ptr(stamps))
layer(x, e, R, t, flags=FLAGS)
torch.cuda.synchronize()
lib.diffab_debug_set_attn_stamps(None)
s = stamps.view(nwg, 8, 8).cpu().double()
if (os.environ.get("DIFFAB_ATTN_FLASH", "0") != "0" or os.environ.get("DIFFAB_ATTN_PIPE", "0") != "0") and K in (64, 128):
    # key-tile pipeline (attention_flash.hip): stamps 0 start | 1 prologue barrier | 2 step 0 | 3 step 1 | 4 step 4 | 6 step NT | 7 step NT+1 | 5 end
    tot = s[..., 5] - s[..., 0]
    print(f"B={B} K={K}: {nwg} work-groups; per-wave lifetime mean {tot.mean():.0f} (min {tot.min():.0f}, max {tot.max():.0f})")
    for nm, a, b_, div in (("prologue (to first barrier)", 0, 1, 1), ("step 0", 1, 2, 1), ("step 1", 2, 3, 1), ("steps 2..4 (per step)", 3, 4, 3),
                           (f"steps 5..NT (per step)", 4, 6, K // 16 - 4), ("step NT+1", 6, 7, 1), ("epilogue", 7, 5, 1)):
        dlt = (s[..., b_] - s[..., a]) / div
        print(f"  {nm:32s} mean {dlt.mean():9.0f}  min {dlt.min():9.0f}  max {dlt.max():9.0f}")
    sys.exit(0)
names = ["phase1 (S: q.k MFMA + point VALU)", "phase2 prologue + barrier", "phase2 (bias, softmax, o_e)", "phase3 prologue + barrier",
         "phase3 (o_s, o_p, epilogue)"]
tot = (s[..., 5] - s[..., 0])
print(f"B={B} K={K}: {nwg} work-groups; per-wave lifetime mean {tot.mean():.0f} cycles (min {tot.min():.0f}, max {tot.max():.0f})")
for k, n in enumerate(names):
    dlt = s[..., k + 1] - s[..., k]
    print(f"  {n:42s} mean {dlt.mean():9.0f}  min {dlt.min():9.0f}  max {dlt.max():9.0f}  ({100 * dlt.mean() / tot.mean():.1f} %)")
if (s[..., 6] > 0).all():
    for nm, a, b_ in (("  P1 start -> end of key tile 0", 0, 6), ("  P1 key tiles 1..3", 6, 7), ("  P1 key tiles 4..7", 7, 1)):
        dlt = s[..., b_] - s[..., a]
        print(f"{nm:44s} mean {dlt.mean():9.0f}  min {dlt.min():9.0f}  max {dlt.max():9.0f}")
e1 = s[..., 1] - s[..., 0].min(dim=1, keepdim=True).values  # end of phase 1 relative to the work-group's first stamp
print(f"  within a work-group: phase-1 end of the slowest wave - mean over its waves: mean {(e1.max(dim=1).values - e1.mean(dim=1)).mean():.0f} cycles; "
      f"start skew (last wave's first stamp - first wave's) mean {(s[..., 0].max(dim=1).values - s[..., 0].min(dim=1).values).mean():.0f}")
for wvi in range(8):
    print(f"    wave {wvi}: phase 1 mean {(s[:, wvi, 1] - s[:, wvi, 0]).mean():.0f}")
span = s[..., 5].max() - s[..., 0].min()
print(f"  kernel span (first stamp to last stamp): {span:.0f} cycles; 100 MHz-based? memtime ticks are shader cycles")
d06 = (s[..., 6] - s[..., 0])
if (s[..., 6] > 0).all():
    print(f"  stamp 0 -> 6: first 256 work-groups mean {d06[:256].mean():.0f}, the rest mean {d06[256:].mean():.0f}; "
          f"percentiles 10/50/90 of all: {d06.flatten().quantile(torch.tensor([0.1, 0.5, 0.9], dtype=torch.float64)).tolist()}")
# time series (wave 0 of every work-group): how many work-groups are inside phase 2 at a time, and how long phase 2 takes for the
# work-groups that enter it in each slice of the launch - are the CUs' stream phases synchronised?
if os.environ.get("ATTN_TIMESERIES", "0") != "0":
    w0 = s[:, 0, :].clone()  # needs a build with -DAT_STAMP_REALTIME (100 MHz ticks, comparable between CUs; s_memtime is not)
    w0 -= w0[:, 0].min()
    a, b_ = w0[:, 2], w0[:, 3]
    nb = 48
    edges = torch.linspace(0, float(w0[:, 5].max()), nb + 1, dtype=torch.float64)
    print("  slice of the launch [k cycles] | work-groups inside phase 2 (mean over the slice) | phase-2 duration of those entering it [k cycles]")
    for k in range(nb):
        lo, hi = edges[k], edges[k + 1]
        inside = ((torch.minimum(b_, hi) - torch.maximum(a, lo)).clamp_min(0)).sum() / (hi - lo)
        ent = (a >= lo) & (a < hi)
        dur = (b_ - a)[ent].mean() / 1e3 if ent.any() else float("nan")
        print(f"   {lo / 1e3:7.0f} .. {hi / 1e3:7.0f} | {inside:6.1f} | {dur:6.1f}  (n = {int(ent.sum())})")
p2 = s[..., 3] - s[..., 2]
print("  phase 2 by wave: " + "  ".join(f"w{w}: {p2[:, w].mean():.0f}" for w in range(8)))
print(f"  phase 2 within a work-group: slowest wave - mean wave = {(p2.max(dim=1).values - p2.mean(dim=1)).mean():.0f}, "
      f"slowest - fastest = {(p2.max(dim=1).values - p2.min(dim=1).values).mean():.0f}; which wave is slowest: "
      + " ".join(f"{int((p2.argmax(dim=1) == w).sum())}" for w in range(8)))
