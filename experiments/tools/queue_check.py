"""The queue-driven attention + to_out launch (EXPERIMENTAL build, DIFFAB_ATTN_QUEUE=1) against the default two launches:
bitwise equality of one IPA layer's output, then the time of the sampler step.  usage: queue_check.py [B]  (parent spawns two children)"""
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXP = os.path.join(REPO, "experiments", "build", "libdiffab_hip.so")

if len(sys.argv) > 2 and sys.argv[2] == "child":
    sys.path.insert(0, os.path.join(REPO, "diffab-pytorch_amd"))
    import torch

    from diffab_pytorch import _hip, synthetic as syn
    from diffab_pytorch.diffab_pytorch import InvariantPointAttentionLayer

    B, K = int(sys.argv[1]), 128
    d = syn.BENCH_DIMS
    torch.manual_seed(0)
    layer = InvariantPointAttentionLayer(d["D"], d["C"], d["DS"], d["PQ"], d["PV"], d["H"]).cuda().requires_grad_(False)
    inp = {k: v.cuda() for k, v in syn.patches(B, K, d, seed=3, coord_sigma=10.0).items()}
    args = (inp["res_context_emb"], inp["pair_context_emb"], inp["orientations"], inp["translations"])
    y = layer(*args, flags=_hip.FLAG_PAIR_PLANES)
    for _ in range(3):
        y2 = layer(*args, flags=_hip.FLAG_PAIR_PLANES)
    torch.cuda.synchronize()
    assert torch.equal(y, y2), "not reproducible"
    torch.save(y.cpu(), sys.argv[3])
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(20):
        layer(*args, flags=_hip.FLAG_PAIR_PLANES)
    t1.record()
    torch.cuda.synchronize()
    print(f"  layer (incl. the per-call pair split): {t0.elapsed_time(t1) / 20:.4f} ms", flush=True)
    sys.exit(0)

B = sys.argv[1] if len(sys.argv) > 1 else "256"
outs = []
for name, env in (("default", {}), ("queue", {"DIFFAB_ATTN_QUEUE": "1"}), ("queue, no stagger", {"DIFFAB_ATTN_QUEUE": "1", "DIFFAB_ATTN_QUEUE_STAGGER": "0"})):
    out = f"/tmp/queue_check_{len(outs)}.pt"
    e = dict(os.environ, DIFFAB_HIP_LIB=EXP, **env)
    print(name, flush=True)
    subprocess.run([sys.executable, __file__, B, "child", out], env=e, check=True)
    outs.append(out)
import torch  # noqa: E402

ys = [torch.load(o) for o in outs]
for i in (1, 2):
    same = torch.equal(ys[0], ys[i])
    print(f"variant {i} bitwise equal to the default: {same}" + ("" if same else f" (max |diff| {float((ys[0] - ys[i]).abs().max()):.3g}, {int((ys[0] != ys[i]).sum())} elements)"))
