"""Which dense kernel is sensitive to a second stream?  One IPA layer (projections + attention + to_out) and one full denoise
step, repeated on stream 1 with identical inputs while stream 2 runs the same kind of work: every repetition must be bitwise
the solo result.  Prints the number of repetitions that differ, per flag set."""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "diffab-pytorch_amd"))
import torch  # noqa: E402

from diffab_pytorch import _hip, synthetic as syn  # noqa: E402
from diffab_pytorch.diffab_pytorch import InvariantPointAttentionLayer  # noqa: E402

B, K, reps = int(sys.argv[1]) if len(sys.argv) > 1 else 128, 128, int(sys.argv[2]) if len(sys.argv) > 2 else 200
d = syn.BENCH_DIMS
torch.manual_seed(0)
layer = InvariantPointAttentionLayer(d["D"], d["C"], d["DS"], d["PQ"], d["PV"], d["H"]).cuda().requires_grad_(False)
inp = {k: v.cuda() for k, v in syn.patches(2 * B, K, d, seed=3, coord_sigma=10.0).items()}
half = lambda k, i: inp[k][i * B:(i + 1) * B].contiguous()
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def layer_call(i, flags):
    return layer(half("res_context_emb", i), half("pair_context_emb", i), half("orientations", i), half("translations", i), flags=flags)


for what, call in (("ipa layer", layer_call),):
    for name, flags in (("default", 0), ("fp32 gemm", _hip.FLAG_FP32_GEMM), ("pair planes", _hip.FLAG_PAIR_PLANES)):
        torch.cuda.synchronize()
        ref = call(0, flags).clone()
        torch.cuda.synchronize()
        bad = torch.zeros(reps, dtype=torch.int64, device="cuda")
        for r in range(reps):
            with torch.cuda.stream(s2):
                call(1, flags)
            with torch.cuda.stream(s1):
                y = call(0, flags)
                bad[r] = (y != ref).sum()
        torch.cuda.synchronize()
        nb = bad.cpu()
        print(f"{what:13s} {name:10s}: {int((nb > 0).sum())} of {reps} repetitions differ from the solo result (worst: {int(nb.max())} elements)")
