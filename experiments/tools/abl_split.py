"""Timing helper for the ablation builds of attention_split.hip: one IPA layer with DIFFAB_FLAG_SPLIT_ATTENTION, 6 calls."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "diffab-pytorch_amd"))
import torch
from diffab_pytorch import _hip, synthetic as syn
from diffab_pytorch.diffab_pytorch import InvariantPointAttentionLayer
B, K = 256, 128
d = syn.BENCH_DIMS
torch.manual_seed(0)
layer = InvariantPointAttentionLayer(d["D"], d["C"], d["DS"], d["PQ"], d["PV"], d["H"]).cuda()
g = torch.Generator(device="cuda").manual_seed(0)
x = torch.randn(B, K, d["D"], device="cuda", generator=g)
e = torch.randn(B, K, K, d["C"], device="cuda", generator=g)
t = 10 * torch.randn(B, K, 3, device="cuda", generator=g)
R = torch.linalg.qr(torch.randn(B, K, 3, 3, device="cuda", generator=g))[0].contiguous()
for _ in range(6):
    layer(x, e, R, t, flags=_hip.FLAG_SPLIT_ATTENTION)
torch.cuda.synchronize()
