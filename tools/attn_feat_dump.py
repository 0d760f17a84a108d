"""Diagnostic: run ONE IPA layer at the benchmark geometry through diffab_ipa_layer_fwd and dump the attention feature rows
(feat = [o_s 256 | o_pair 512 | o_pts 192 | norms 64], taken from the workspace) to a .npy file.  usage: attn_feat_dump.py OUT FLAGS"""
import ctypes as C
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "diffab-pytorch_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from diffab_pytorch import _hip, synthetic as syn  # noqa: E402
from diffab_pytorch.diffab_pytorch import InvariantPointAttentionLayer, _named  # noqa: E402

out, flags = sys.argv[1], int(sys.argv[2])
B, K = 2, 128
lib = _hip.lib()
d = syn.BENCH_DIMS
torch.manual_seed(0)
layer = InvariantPointAttentionLayer(d["D"], d["C"], d["DS"], d["PQ"], d["PV"], d["H"]).cuda().requires_grad_(False)  # inference kernels
inp = {k: v.cuda() for k, v in syn.patches(B, K, d, seed=3, coord_sigma=8.0).items()}
x, e, R, t = inp["res_context_emb"], inp["pair_context_emb"], inp["orientations"], inp["translations"]
dims = _hip.make_dims(B, K, d["D"], d["C"], d["H"], d["DS"], d["PQ"], d["PV"], 1)
keep = []
w = _hip.ipa_layer_weights(_named(layer), keep)
ws = _hip.workspace(lib.diffab_denoise_workspace_bytes(C.byref(dims)))
y = torch.empty_like(x)
_hip.check(lib.diffab_ipa_layer_fwd(C.byref(dims), C.byref(w), _hip.ptr(x), _hip.ptr(e), _hip.ptr(R), _hip.ptr(t), _hip.ptr(y), _hip.ptr(ws),
                                    ws.numel(), flags, _hip.stream_ptr()), "ipa")
torch.cuda.synchronize()
rows, D, V = B * K, d["D"], 21
off = 0
def take(nbytes):
    global off
    off = (off + 255) // 256 * 256
    r = off
    off += nbytes
    return r
for n in (rows * 2 * D, rows * D, rows * D, rows * D, rows * (D + 3), rows * D, rows * D, rows * 3, rows * V, 25 * D, 3 * B * D):
    take(4 * n)
ipa = take(0)
feat = ws.view(torch.uint8)[ipa + rows * 1344 * 4: ipa + rows * (1344 + 1024) * 4].view(torch.float32).view(rows, 1024).cpu().numpy()
np.save(out, np.concatenate([feat, y.view(rows, D).cpu().numpy()], axis=1))
print("dumped", out, feat.shape, float(np.abs(feat).max()))
