"""Training-step timing (BASELINE config 4 shape per GPU: B=128 synthetic K=128 patches, benchmark model):
forward-noise + taped denoise forward + 3 losses + full backward, one process.  Prints ms/step and residue-steps/s."""
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "diffab-pytorch_amd"))
import torch  # noqa: E402

from diffab_pytorch import DiffAb, synthetic as syn  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
K, steps = 128, 5
d = syn.BENCH_DIMS
torch.manual_seed(0)
model = DiffAb(d["D"], d["C"], d["NL"], d["DS"], d["PQ"], d["PV"], d["H"]).cuda()
g = torch.Generator(device="cuda").manual_seed(0)
inp = syn.patches(8, K, d, seed=0)
rep = B // 8
batch = {"seq_idx": inp["seq_idx"].repeat(rep, 1).cuda(), "xyz": inp["translations"].repeat(rep, 1, 1).cuda(),
         "orientations": inp["orientations"].repeat(rep, 1, 1, 1).cuda(), "generation_mask": inp["generation_mask"].repeat(rep, 1).cuda(),
         "residue_mask": inp["residue_mask"].repeat(rep, 1).cuda(),
         "res_context_emb": torch.randn(B, K, d["D"], device="cuda", generator=g),
         "pair_context_emb": torch.randn(B, K, K, d["C"], device="cuda", generator=g)}
opt = model.configure_optimizers()
for it in range(2 + steps):
    if it == 2:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
    opt.zero_grad()
    loss = model.training_step(batch, it)
    loss.backward()
    opt.step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
print(f"train step B={B} K={K}: {dt*1e3:.1f} ms/step, {B*K/dt:.0f} residue-steps/s, loss {float(loss):.4f}")
