"""One IPA layer on the first patches of a batch at several batch sizes: every launch form (k parts, 64-row groups, 128-row groups,
column-split projections) must give the same bits for the same patch.  usage: shard_forms_check.py"""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "diffab-pytorch_amd"))
import torch  # noqa: E402

from diffab_pytorch import _hip, synthetic as syn  # noqa: E402
from diffab_pytorch.diffab_pytorch import InvariantPointAttentionLayer  # noqa: E402

K = 128
d = dict(syn.BENCH_DIMS)
torch.manual_seed(0)
layer = InvariantPointAttentionLayer(d["D"], d["C"], d["DS"], d["PQ"], d["PV"], d["H"]).cuda().requires_grad_(False)
g = torch.Generator(device="cuda").manual_seed(0)
BMAX = 256
x = torch.randn(BMAX, K, d["D"], device="cuda", generator=g)
e = torch.randn(8, K, K, d["C"], device="cuda", generator=g)
t = 10 * torch.randn(BMAX, K, 3, device="cuda", generator=g)
qn = torch.nn.functional.normalize(torch.randn(BMAX, K, 4, device="cuda", generator=g), dim=-1)
w_, x_, y_, z_ = qn.unbind(-1)
R = torch.stack([1 - 2 * (y_ * y_ + z_ * z_), 2 * (x_ * y_ - z_ * w_), 2 * (x_ * z_ + y_ * w_), 2 * (x_ * y_ + z_ * w_),
                 1 - 2 * (x_ * x_ + z_ * z_), 2 * (y_ * z_ - x_ * w_), 2 * (x_ * z_ - y_ * w_), 2 * (y_ * z_ + x_ * w_),
                 1 - 2 * (x_ * x_ + y_ * y_)], -1).view(BMAX, K, 3, 3).contiguous()
outs = {}
for B in (1, 8, 16, 40, 128, 256):
    eb = e.repeat((B + 7) // 8, 1, 1, 1, 1)[:B].contiguous()
    outs[B] = layer(x[:B].contiguous(), eb, R[:B].contiguous(), t[:B].contiguous(), flags=_hip.FLAG_PAIR_PLANES)[: min(B, 8)].clone()
torch.cuda.synchronize()
ref = outs[256]
for B, o in outs.items():
    n = o.shape[0]
    df = (o != ref[:n])
    print(f"B={B:4d}: first {n} patches vs the B=256 run: {int(df.sum())} differing words; max |diff| {float((o - ref[:n]).abs().max()):.3e}; "
          f"rows touched {int(df.any(-1).sum())}; columns touched {int(df.any(0).any(0).sum())}", flush=True)
