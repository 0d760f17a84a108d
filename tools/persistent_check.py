"""The patch-resident module kernel (DIFFAB_FLAG_PERSISTENT_MODULE) against the multi-launch path, on the GPU box:
   1. bitwise: the same reverse steps from the same state, both ways (seq, x, O compared bit for bit)
   2. timing: N-step sampler blocks, alternating the two forms (ms per step, min / median of the blocks)
   3. optional (STAMPS=1): the phase time series of one step - how many CUs are inside the pair stream (phase 2) at a time
usage: persistent_check.py [B] [steps] ; env STAGGER="ticks,classes" (10 ns ticks) ; STAMPS=1
"""
import ctypes as C
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "diffab-pytorch_amd"))
import torch  # noqa: E402

from diffab_pytorch import DiffAb, _hip, synthetic as syn  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 20
K = 128
lib = _hip.lib()
dims = dict(syn.BENCH_DIMS)
torch.manual_seed(0)
model = DiffAb(dims["D"], dims["C"], dims["NL"], dims["DS"], dims["PQ"], dims["PV"], dims["H"]).cuda()
inp = {k: v.cuda() for k, v in syn.patches(B, K, dims, seed=0, coord_sigma=10.0).items()}
hd, w = model.denoiser.hip_dims(B, K), model.denoiser.hip_weights()
sd_dev, tab = model._sched_on_device(), model._reverse_so3().struct()
ws = _hip.workspace(lib.diffab_sample_workspace_bytes(C.byref(hd)))
gm, rc, pc = inp["generation_mask"], inp["res_context_emb"], inp["pair_context_emb"]
if os.environ.get("VARIANT"):
    lib.diffab_debug_set_attn_variant(int(os.environ["VARIANT"]))  # 2: no operands carried between the items of the persistent kernel
if os.environ.get("STAGGER"):
    tk, cl = (int(v) for v in os.environ["STAGGER"].split(","))
    lib.diffab_debug_set_module_stagger(tk, cl)


def fresh():
    seq, x, O = inp["seq_idx"].clone(), inp["translations"].clone(), inp["orientations"].clone()
    _hip.check(lib.diffab_sample_init(_hip.ptr(seq), _hip.ptr(x), _hip.ptr(O), _hip.ptr(gm), 2024, 0, B, K, model.T, _hip.stream_ptr()), "init")
    return seq, x, O


def loop(state, t_hi, n, flags):
    seq, x, O = state
    _hip.check(lib.diffab_sample_loop(C.byref(hd), C.byref(w.struct), C.byref(sd_dev.struct), C.byref(tab), _hip.ptr(seq), _hip.ptr(x),
                                      _hip.ptr(O), _hip.ptr(rc), _hip.ptr(pc), _hip.ptr(gm), 2024, 0, t_hi, t_hi - n, _hip.ptr(ws), ws.numel(),
                                      flags, _hip.stream_ptr()), "sample_loop")


P, ML = _hip.FLAG_PERSISTENT_MODULE, _hip.FLAG_MULTI_LAUNCH  # (flags 0: the loop picks the persistent form itself at B >= #CUs)
# 1. bitwise: each form twice (run-to-run determinism), then one against the other
runs = []
for fl in (ML, ML, P, P, P):
    s_ = fresh()
    loop(s_, model.T, 4, fl)
    torch.cuda.synchronize()
    runs.append(s_)
eq = lambda u, v: all(torch.equal(p_, q_) for p_, q_ in zip(u, v))
print(f"B={B}: 4 reverse steps; multi-launch twice equal {eq(runs[0], runs[1])}; persistent runs 1=2 {eq(runs[2], runs[3])} 2=3 {eq(runs[3], runs[4])}; "
      f"persistent = multi-launch {eq(runs[0], runs[2])}", flush=True)
a, b = runs[0], runs[2]
same = eq(a, b)
if not same:
    for nm, u, v in zip(("seq", "x", "O"), a, b):
        df = (u != v)
        print(f"  {nm}: {int(df.sum())} differing elements of {u.numel()}; patches touched: {int(df.view(B, -1).any(1).sum())}; "
              f"max |diff| {float((u.double() - v.double()).abs().max()):.3e}")
# 2. timing
st = fresh()
res = {ML: [], P: []}
for rep in range(6):
    for fl in (ML, P):
        loop(st, model.T, 3, fl)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        loop(st, model.T - 3, STEPS, fl)
        torch.cuda.synchronize()
        res[fl].append((time.perf_counter() - t0) / STEPS * 1e3)
for fl, nm in ((ML, "multi-launch"), (P, "persistent  ")):
    r = sorted(res[fl][1:])
    print(f"  {nm}: ms per step min {r[0]:.4f} median {r[len(r) // 2]:.4f} max {r[-1]:.4f}   blocks {['%.3f' % v for v in res[fl]]}", flush=True)
# 3. phase time series
if os.environ.get("STAMPS", "0") != "0":
    NL = dims["NL"]
    n_att = B * NL * 8 * 64
    stamps = torch.zeros(n_att + B * NL * 4, dtype=torch.int64, device="cuda")
    lib.diffab_debug_set_module_stamps(_hip.ptr(stamps))
    loop(st, model.T - 30, 1, P)
    torch.cuda.synchronize()
    lib.diffab_debug_set_module_stamps(None)
    s = stamps[:n_att].view(B, NL, 8, 8, 8).cpu().double()  # [patch][layer][tile][wave][stamp]
    ph = stamps[n_att:].view(B, NL, 4).cpu().double()
    t00 = ph[:, 0, 0].min()
    tick = 0.01  # us
    print(f"  module span {(ph[:, -1, 3].max() - t00) * tick:.1f} us; per patch-layer: proj {((ph[..., 1] - ph[..., 0]).mean()) * tick:.1f} us, "
          f"attention {((ph[..., 2] - ph[..., 1]).mean()) * tick:.1f} us, to_out {((ph[..., 3] - ph[..., 2]).mean()) * tick:.1f} us; "
          f"last work-group done at {(ph[:, -1, 3].max() - t00) * tick:.1f}, first at {(ph[:, -1, 3].min() - t00) * tick:.1f}")
    dur = (ph[:, -1, 3] - ph[:, 0, 0]) * tick  # a work-group's own module time (first projection stamp -> last to_out stamp)
    start = (ph[:, 0, 0] - t00) * tick
    print(f"  per work-group: own time min {dur.min():.0f} mean {dur.mean():.0f} max {dur.max():.0f} us; start min {start.min():.0f} max {start.max():.0f} us")
    print("  own time by XCD (work-group % 8): " + " ".join(f"{float(dur[x::8].mean()):.0f}" for x in range(8)))
    if os.environ.get("STAMPS") == "2":  # is a work-group's lateness systematic (the same CUs every launch) or drawn anew per launch?
        durs, fins = [dur], [(ph[:, -1, 3] - t00) * tick]
        for rep in range(3):
            stamps.zero_()
            lib.diffab_debug_set_module_stamps(_hip.ptr(stamps))
            loop(st, model.T - 40 - rep, 1, P)
            torch.cuda.synchronize()
            lib.diffab_debug_set_module_stamps(None)
            ph2 = stamps[n_att:].view(B, NL, 4).cpu().double()
            durs.append((ph2[:, -1, 3] - ph2[:, 0, 0]) * tick)
            fins.append((ph2[:, -1, 3] - ph2[:, 0, 0].min()) * tick)
        D = torch.stack(durs)
        print("  four stamped launches: own time mean " + " ".join(f"{float(d.mean()):.0f}" for d in durs) + " | max " +
              " ".join(f"{float(d.max()):.0f}" for d in durs) + " | launch end (last finish) " + " ".join(f"{float(f_.max()):.0f}" for f_ in fins))
        cc = torch.corrcoef(D)
        print("  correlation of the per-work-group own times between launches: " + " ".join(f"{float(cc[0, j]):.2f}" for j in range(1, 4)) +
              f" | mean over launches per work-group: min {float(D.mean(0).min()):.0f} max {float(D.mean(0).max()):.0f} (systematic spread) vs "
              f"sum of the four launch maxima {float(D.max(1).values.sum()):.0f} against max of the four-launch sums {float(D.sum(0).max()):.0f} "
              f"(what a trajectory-resident work-group would pay)")
    print("  own time by stagger class ((work-group / 8) % 8): " + " ".join(f"{float(dur[[i for i in range(B) if (i // 8) % 8 == c]].mean()):.0f}" for c in range(8)))
    print("  finish time by stagger class: " + " ".join(f"{float(((ph[:, -1, 3] - t00) * tick)[[i for i in range(B) if (i // 8) % 8 == c]].mean()):.0f}" for c in range(8)))
    w0 = s[:, :, :, 0, :]  # wave 0
    tile_life = (w0[..., 5] - w0[..., 0]) * tick
    p1, p2, p3 = (w0[..., 2] - w0[..., 0]) * tick, (w0[..., 3] - w0[..., 2]) * tick, (w0[..., 5] - w0[..., 3]) * tick
    print(f"  attention tile: lifetime {tile_life.mean():.2f} us (phase 1 + barrier {p1.mean():.2f}, phase 2 {p2.mean():.2f}, barrier + phase 3 {p3.mean():.2f})")
    # all eight waves: spans of the phases inside a tile (first wave in -> last wave out)
    sa = s  # [B][NL][tile][wave][stamp]
    t_in = sa[..., 0].min(-1).values
    p1_first, p1_last = sa[..., 1].min(-1).values, sa[..., 1].max(-1).values
    b1 = sa[..., 2].min(-1).values  # barrier released
    p2_first, p2_last = sa[..., 3].min(-1).values, sa[..., 3].max(-1).values
    b2_ = sa[..., 4].min(-1).values
    t_out = sa[..., 5].max(-1).values
    f = lambda v: f"{float(v.mean()) * tick:.2f}"
    print(f"  tile, all waves [us]: start -> first wave ends P1 {f(p1_first - t_in)} | -> last wave ends P1 {f(p1_last - p1_first)} | barrier release "
          f"{f(b1 - p1_last)} | -> first wave ends P2 {f(p2_first - b1)} | -> last wave ends P2 {f(p2_last - p2_first)} | barrier {f(b2_ - p2_last)} | "
          f"P3 -> last wave out {f(t_out - b2_)} | total {f(t_out - t_in)}")
    print("  P1 by wave [us]: " + " ".join(f"{float((sa[..., w_, 1] - sa[..., w_, 0]).mean()) * tick:.2f}" for w_ in range(8))
          + " | key tile 0 done: " + " ".join(f"{float((sa[..., w_, 6] - sa[..., w_, 0]).mean()) * tick:.2f}" for w_ in range(8)))
    print("  P2 by wave [us]: " + " ".join(f"{float((sa[..., w_, 3] - sa[..., w_, 2]).mean()) * tick:.2f}" for w_ in range(8)))
    print("  P3 by wave [us]: " + " ".join(f"{float((sa[..., w_, 5] - sa[..., w_, 4]).mean()) * tick:.2f}" for w_ in range(8)))
    print("  tile lifetime by tile index [us]: " + " ".join(f"{float((t_out - t_in)[:, :, ti].mean()) * tick:.2f}" for ti in range(8)))
    print("  tile lifetime by layer [us]: " + " ".join(f"{float((t_out - t_in)[:, li].mean()) * tick:.2f}" for li in range(NL)))
    a2, b2 = (w0[..., 2] - t00).flatten(), (w0[..., 3] - t00).flatten()
    end = float((ph[:, -1, 3].max() - t00))
    nb = 60
    edges = torch.linspace(0, end, nb + 1, dtype=torch.float64)
    inside = []
    for k in range(nb):
        lo, hi = edges[k], edges[k + 1]
        inside.append(float(((torch.minimum(b2, hi) - torch.maximum(a2, lo)).clamp_min(0)).sum() / (hi - lo)))
    print("  CUs inside phase 2 (pair stream), mean over each 1/60 of the launch:")
    print("   " + " ".join(f"{v:.0f}" for v in inside))
    mid = inside[3:-3]
    print(f"  min / mean / max over the middle 54 slices: {min(mid):.0f} / {sum(mid) / len(mid):.0f} / {max(mid):.0f}")
