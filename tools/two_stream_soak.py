"""Round 6: the body of tests/test_gpu_configs.py::test_two_sampler_pipelines_on_two_streams_... with the library's
cross-stream guard OFF, many repetitions: two reverse-sampling pipelines (T = 200 steps, B patches each) enqueued on two streams at the same
time against the same two calls run one after the other, compared bit for bit; then the single denoise step repeated on one stream while the
other stream runs the same work.  Rounds 3-5 saw wrong orientation rows here (lanes 48-63 of heads_finish_kernel / reverse_update_philox_kernel);
round 6 removed the v_pk_*_f32 op_sel:[0,1] form from every kernel of the library (profiles/r06_lanes_48_63.md).
usage: two_stream_soak.py [repetitions] [B] [steps]"""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "diffab-pytorch_amd"))
import torch  # noqa: E402

from diffab_pytorch import DiffAb, _hip, synthetic as syn  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
T = int(sys.argv[3]) if len(sys.argv) > 3 else 200
lib = _hip.lib()
dims = dict(syn.BENCH_DIMS)
torch.manual_seed(0)
model = DiffAb(dims["D"], dims["C"], dims["NL"], dims["DS"], dims["PQ"], dims["PV"], dims["H"], T=T).cuda()
model.denoiser.load_state_dict(syn.denoiser_state_dict(dims, seed=1, prefix=""))
inp = {k: v.cuda() for k, v in syn.patches(2 * B, 128, dims, seed=21, coord_sigma=10.0).items()}
halves = [slice(0, B), slice(B, 2 * B)]


def run(sl, lo):
    return model.sample(inp["seq_idx"][sl], inp["translations"][sl], inp["orientations"][sl], seed=5, first_patch=lo,
                        res_context_emb=inp["res_context_emb"][sl], pair_context_emb=inp["pair_context_emb"][sl],
                        generation_mask=inp["generation_mask"][sl])


guard = int(os.environ.get("GUARD", "0"))
assert lib.diffab_set_stream_guard(guard) == 0
torch.cuda.synchronize()
seq = []
for i, sl in enumerate(halves):
    seq.append(run(sl, i * B))
    torch.cuda.synchronize()
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
bad_reps, bad_elems = 0, 0
for rep in range(reps):
    con = [None, None]
    with torch.cuda.stream(s1):
        con[0] = run(halves[0], 0)
    with torch.cuda.stream(s2):
        con[1] = run(halves[1], B)
    torch.cuda.synchronize()
    n = sum(int((con[i][k] != seq[i][k]).sum()) for i in range(2) for k in seq[i])
    bad_reps += n > 0
    bad_elems += n
print(f"guard {guard}: two {T}-step pipelines (B = {B} each) on two streams: {bad_reps} of {reps} repetitions differ from the sequential runs "
      f"({bad_elems} elements)", flush=True)
# the single denoise step on stream 1 while stream 2 runs the same work on the other half
args = lambda sl: (inp["seq_idx"][sl], inp["translations"][sl], inp["orientations"][sl], inp["res_context_emb"][sl], inp["pair_context_emb"][sl],
                   torch.full((B,), 0.3, device="cuda"), inp["generation_mask"][sl], torch.ones(B, 128, dtype=torch.bool, device="cuda"))
with torch.no_grad():
    solo = model.denoise(*args(halves[0]))
    torch.cuda.synchronize()
    bad_steps, bad_step_elems = 0, 0
    for rep in range(4 * reps):
        with torch.cuda.stream(s2):
            for _ in range(3):
                model.denoise(*args(halves[1]))
        with torch.cuda.stream(s1):
            out = model.denoise(*args(halves[0]))
        torch.cuda.synchronize()
        n = sum(int((out[k] != solo[k]).sum()) for k in solo)
        bad_steps += n > 0
        bad_step_elems += n
print(f"guard {guard}: denoise step beside a busy second stream: {bad_steps} of {4 * reps} repetitions differ from the solo step ({bad_step_elems} elements)",
      flush=True)
lib.diffab_set_stream_guard(1)
sys.exit(1 if (bad_reps or bad_steps) else 0)
