"""A/B of the plane attention kernel's forms on the GPU box (diffab_debug_set_attn_variant): one IPA layer at the benchmark geometry,
output difference between the forms, then sampler steps timed with each.  usage: attn_variant_check.py [B] [steps]"""
import ctypes as C
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "diffab-pytorch_amd"))
import torch  # noqa: E402

from diffab_pytorch import DiffAb, _hip, synthetic as syn  # noqa: E402
from diffab_pytorch.diffab_pytorch import InvariantPointAttentionLayer  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 20
K = 128
lib = _hip.lib()
d = dict(syn.BENCH_DIMS)
torch.manual_seed(0)
layer = InvariantPointAttentionLayer(d["D"], d["C"], d["DS"], d["PQ"], d["PV"], d["H"]).cuda().requires_grad_(False)
g = torch.Generator(device="cuda").manual_seed(0)
Bs = min(B, 16)
x = torch.randn(Bs, K, d["D"], device="cuda", generator=g)
e = torch.randn(Bs, K, K, d["C"], device="cuda", generator=g)
t = 10 * torch.randn(Bs, K, 3, device="cuda", generator=g)
qn = torch.nn.functional.normalize(torch.randn(Bs, K, 4, device="cuda", generator=g), dim=-1)
w_, x_, y_, z_ = qn.unbind(-1)
R = torch.stack([1 - 2 * (y_ * y_ + z_ * z_), 2 * (x_ * y_ - z_ * w_), 2 * (x_ * z_ + y_ * w_), 2 * (x_ * y_ + z_ * w_),
                 1 - 2 * (x_ * x_ + z_ * z_), 2 * (y_ * z_ - x_ * w_), 2 * (x_ * z_ - y_ * w_), 2 * (y_ * z_ + x_ * w_),
                 1 - 2 * (x_ * x_ + y_ * y_)], -1).view(Bs, K, 3, 3).contiguous()
outs = {}
VARIANTS = tuple(int(v) for v in os.environ.get("VARIANTS", "0,1").split(","))  # diffab_debug_set_attn_variant values to compare
for v in VARIANTS:
    lib.diffab_debug_set_attn_variant(v)
    outs[v] = layer(x, e, R, t, flags=_hip.FLAG_PAIR_PLANES).clone()
lib.diffab_debug_set_attn_variant(0)
ref = layer(x, e, R, t, flags=0)  # fp32 pair stream
torch.cuda.synchronize()
den = float(ref.abs().max())
v0, v1 = VARIANTS[0], VARIANTS[-1]
print(f"one IPA layer, {Bs} patches: |variant{v1} - variant{v0}| / max = {float((outs[v1] - outs[v0]).abs().max()) / den:.3e} "
      f"(bitwise equal: {torch.equal(outs[v1], outs[v0])}); variant{v0} vs fp32-pair kernel {float((outs[v0] - ref).abs().max()) / den:.3e}; "
      f"variant{v1} vs fp32-pair kernel {float((outs[v1] - ref).abs().max()) / den:.3e}; finite {bool(torch.isfinite(outs[v1]).all())}", flush=True)

dims = d
model = DiffAb(dims["D"], dims["C"], dims["NL"], dims["DS"], dims["PQ"], dims["PV"], dims["H"]).cuda()
inp = {k: v.cuda() for k, v in syn.patches(B, K, dims, seed=0, coord_sigma=10.0).items()}
hd, w = model.denoiser.hip_dims(B, K), model.denoiser.hip_weights()
sd_dev, tab = model._sched_on_device(), model._reverse_so3().struct()
ws = _hip.workspace(lib.diffab_sample_workspace_bytes(C.byref(hd)))
gm, rc, pc = inp["generation_mask"], inp["res_context_emb"], inp["pair_context_emb"]
seq, xx, O = inp["seq_idx"].clone(), inp["translations"].clone(), inp["orientations"].clone()
_hip.check(lib.diffab_sample_init(_hip.ptr(seq), _hip.ptr(xx), _hip.ptr(O), _hip.ptr(gm), 2024, 0, B, K, model.T, _hip.stream_ptr()), "init")


def loop(t_hi, n):
    _hip.check(lib.diffab_sample_loop(C.byref(hd), C.byref(w.struct), C.byref(sd_dev.struct), C.byref(tab), _hip.ptr(seq), _hip.ptr(xx),
                                      _hip.ptr(O), _hip.ptr(rc), _hip.ptr(pc), _hip.ptr(gm), 2024, 0, t_hi, t_hi - n, _hip.ptr(ws), ws.numel(),
                                      _hip.FLAG_MULTI_LAUNCH, _hip.stream_ptr()), "sample_loop")


res = {v: [] for v in VARIANTS}
for rep in range(5):
    for v in VARIANTS:
        lib.diffab_debug_set_attn_variant(v)
        loop(model.T, 3)
        torch.cuda.synchronize()
        lib.diffab_kernel_timer_enable(1)
        t0 = time.perf_counter()
        loop(model.T - 3, STEPS)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / STEPS * 1e3
        n_, ms_ = C.c_int64(0), C.c_double(0.0)
        lib.diffab_kernel_timer_read(C.byref(n_), C.byref(ms_))
        lib.diffab_kernel_timer_enable(0)
        res[v].append((dt, ms_.value / max(n_.value, 1)))
lib.diffab_debug_set_attn_variant(0)
for v in VARIANTS:
    r = sorted(res[v][1:])
    print(f"  variant {v}: ms per step median {r[len(r) // 2][0]:.4f} (min {r[0][0]:.4f}); attention launch median {r[len(r) // 2][1] * 1e3:.1f} us; "
          f"blocks {['%.3f' % a for a, _ in res[v]]}; finite {bool(torch.isfinite(xx).all())}", flush=True)
