#!/usr/bin/env python3
"""bench.py - CDR-residue denoise-steps/sec on synthetic K=128 patches (BASELINE.json metric).

A "step" is one pass of the hot path over one batch: one reverse-diffusion step (Denoiser forward + Philox noise +
IGSO3 draw + state update) for B patches of K residues on every rank.  residue-steps = n_gpus * B * K * steps.
Workload at N=1: BASELINE.json configs[1] - batch=256 synthetic K=128 patches, 100-step sampling, benchmark model
(reference train.py:62-70).  N>1: the same 256 patches PER GPU (weak scaling), no data-path collective; one RCCL
all-gather of the sampled structures at the end of the timed region.

    python bench.py --gpus 1 --steps 100 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
    ... bench.py --train [--gpus N]        BASELINE config 4 (training step, 128 patches per GPU, gradient all-reduce) instead

Prints ONE JSON line on rank 0 (contract: task prompt section 4).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(REPO, "diffab-pytorch_amd"))

import torch  # noqa: E402

HBM_PEAK_GBPS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def algorithmic_bytes_per_attention_launch(B, K, D, C):
    """compulsory traffic of one IPA layer's attention kernel with perfect on-chip reuse:
    the pair embedding e[b] streamed once (k4 + k8 fused), fp32.  Per patch K*K*C*4 bytes."""
    return B * K * K * C * 4


def algorithmic_flops_per_residue_step(K, dims):
    """FLOPs of one denoise call per residue, counted on the reference's formulation (2 per multiply-add):
    per IPA layer and residue the six projections 2 D (3 H ds + 3 H P 3), logits 2 K H (ds + 3 P) + bias 2 K C H, the three
    attention-weighted sums 2 K H (ds + C + 3 P), to_out 2 D (H ds + H C + 4 H P); plus the embedding and head MLPs."""
    D, C, H, ds, P, NL = dims["D"], dims["C"], dims["H"], dims["DS"], dims["PQ"], dims["NL"]
    layer = 2 * D * (3 * H * ds + 9 * H * P) + 2 * K * H * (ds + 3 * P) + 2 * K * C * H + 2 * K * H * (ds + C + 3 * P) \
        + 2 * D * (H * ds + H * C + 4 * H * P)
    mlps = 2 * (2 * D * D + D * D) + 3 * 2 * ((D + 3) * D + D * D) + 2 * D * (3 + 3 + 21)
    return NL * layer + mlps


def algorithmic_bytes_per_residue_step(K, D, C, NL, V=21):
    """NL*K^2*C*4 (pair stream) + K*D*4 (res ctx) + K*(8+12+36) in + K*(12+36+4V) out, per residue."""
    per_patch = NL * K * K * C * 4 + K * D * 4 + K * (8 + 12 + 36) + K * (12 + 36 + 4 * V)
    return per_patch / K


def cpu_baseline(dims, sd, K, seed, budget_s=30.0):
    """The oracle (torch-CPU restatement of the reference's formulation incl. the materialised point-difference tensor, checked
    against the real reference by the golden vectors) timed on this host's cores, as SURVEY 8(d) / BASELINE.md section 3 specify:
    B = 1 and B = 8 patches of K residues; (i) reverse-sampling steps, (ii) the hot-path training step (forward + backward);
    all cores and one thread.  A bounded sample: each case runs one warm-up and as many timed iterations (1..5) as fit its share of
    `budget_s`.  `value` is the best all-core sampling case; every case is listed under `cases`."""
    sys.path.insert(0, os.path.join(REPO, "oracle"))
    import numpy as np

    import diffab_oracle as orc
    from diffab_pytorch import synthetic as syn

    sched = orc.cosine_variance_schedule(100, s=0.01, beta_max=0.999)
    sdo = {"denoiser." + k: v for k, v in sd.items()}
    default_threads = torch.get_num_threads()
    thread_sets = sorted({default_threads, min(16, default_threads)}, reverse=True)  # the 1-GPU box's CPU share is 16 of the host's cores

    def make(B):
        inp = syn.patches(B, K, dims, seed=seed, coord_sigma=10.0)
        patch = np.arange(B)[:, None] + np.zeros((B, K), dtype=np.int64)
        res = np.zeros((B, K), dtype=np.int64) + np.arange(K)[None, :]
        return inp, patch, res

    def sampling_case(B):
        inp, patch, res = make(B)
        state = [inp["seq_idx"], inp["translations"], inp["orientations"]]

        def it(i):
            t = 100 - (i % 100)
            with torch.no_grad():
                den = orc.denoiser(sdo, state[0], state[1], state[2], inp["res_context_emb"], inp["pair_context_emb"],
                                   sched["beta"][t].expand(B), dims["NL"], dims["H"])
                z = torch.from_numpy(np.stack(orc.philox_normal4(seed, patch, res, t, 1)[:3], -1))
                rv = 0.1 * torch.from_numpy(np.stack(orc.philox_normal4(seed, patch, res, t, 2)[:3], -1))
                u = torch.from_numpy(orc.philox_uniform4(seed, patch, res, t, 0)[0])
                state[0], state[1], state[2] = orc.reverse_update(t, state[0], state[1], state[2], den, inp["generation_mask"], sched, z, rv, u)
        return it

    def training_case(B):
        inp, patch, res = make(B)
        params = {k: v.clone().requires_grad_(True) for k, v in sdo.items()}
        t = torch.full((B,), 40)
        eps = torch.from_numpy(np.stack(orc.philox_normal4(seed, patch, res, 40, 1)[:3], -1))
        x_t = orc.coord_diffuse_from_t0(inp["translations"], t, inp["generation_mask"], eps, sched)
        post = orc.seq_posterior_single_step(inp["seq_idx"], inp["seq_idx"], t, inp["generation_mask"], sched)

        def it(i):
            for p in params.values():
                p.grad = None
            den = orc.denoiser(params, inp["seq_idx"], x_t, inp["orientations"], inp["res_context_emb"], inp["pair_context_emb"],
                               sched["beta"][t], dims["NL"], dims["H"])
            ls = orc.hotpath_losses(den, post, eps, inp["orientations"], inp["generation_mask"], inp["residue_mask"])
            (ls[0] + ls[1] + ls[2]).backward()
        return it

    plan = []  # (name, B, threads, factory): all-core cases first so the single-thread ones take what is left of the budget
    for threads in thread_sets + [1]:
        for B in (1, 8):
            plan.append(("sampling", B, threads, sampling_case))
            plan.append(("training", B, threads, training_case))
    cases, t_start = [], time.perf_counter()
    for idx, (name, B, threads, factory) in enumerate(plan):
        left = budget_s - (time.perf_counter() - t_start)
        if left <= 0.5:
            cases.append({"what": name, "patches": B, "threads": threads, "skipped": "budget spent"})
            continue
        share = left / (len(plan) - idx) * (2.0 if threads == 1 else 1.0)
        torch.set_num_threads(threads)
        it = factory(B)
        t0 = time.perf_counter()
        it(0)  # warm-up (its time bounds how many timed iterations fit)
        warm = time.perf_counter() - t0
        n = int(max(1, min(5, share / max(warm, 1e-3) - 1)))
        t0 = time.perf_counter()
        for i in range(n):
            it(1 + i)
        dt = (time.perf_counter() - t0) / n
        cases.append({"what": name, "patches": B, "threads": threads, "iters": n, "ms_per_step": dt * 1e3,
                      "residue_steps_per_s": B * K / dt})
    torch.set_num_threads(default_threads)
    best = max((c for c in cases if c["what"] == "sampling" and c["threads"] > 1 and "ms_per_step" in c),
               key=lambda c: c["residue_steps_per_s"])
    return {
        "value": best["residue_steps_per_s"],
        "unit": "residue-steps/s",
        "cores": best["threads"],  # (the contract's field name: the torch intra-op threads of the case that won, NOT the host's core count)
        "threads": best["threads"],
        "host_logical_cpus": os.cpu_count(),
        "kind": "port",
        "sample": f"oracle/diffab_oracle.py (torch CPU fp32, reference formulation) on {os.cpu_count()} logical CPUs: B = 1 and 8 patches x "
                  f"K={K}; reverse-sampling steps and hot-path training steps (forward + backward); {thread_sets} threads and 1 thread; one "
                  f"warm-up + 1..5 timed iterations per case, {time.perf_counter() - t_start:.0f} s in all; value = sampling, "
                  f"B={best['patches']}, {best['threads']} threads",
        "cases": cases,
    }


def gpu_clocks():
    """sclk / mclk / power / power cap of GPU 0 as rocm-smi reports them (a child process; None where the tool or a field is missing).
    Read before and after the timed blocks so that a slow box can be told from a regression."""
    import shutil
    import subprocess

    if "rocprof" in os.environ.get("LD_PRELOAD", "").lower() or any(k.startswith(("ROCPROF", "ROCP_TOOL")) for k in os.environ):
        # under rocprofv3 the profiler's preloaded library has initialised the GPU in this process before main(): starting rocm-smi
        # (a python script: fork + exec) from here is the exec the GPU boxes refuse
        return {"skipped": "under rocprofv3"}
    exe = shutil.which("rocm-smi") or "/opt/rocm/bin/rocm-smi"
    try:
        out = subprocess.run([exe, "-d", "0", "--showclocks", "--showpower", "--showmaxpower", "--showperflevel", "--json"], capture_output=True,
                             text=True, timeout=20).stdout
        card = next(iter(json.loads(out).values()))
    except Exception as ex:  # noqa: BLE001
        return {"error": f"{type(ex).__name__}: {ex}"[:120]}
    keep = {}
    for k, v in card.items():
        kl = k.lower()
        if any(t in kl for t in ("sclk", "mclk", "fclk", "power", "performance level")):
            keep[k] = v
    return keep


def train_bytes_per_patch(K, dims, dpair):
    """Algorithmic HBM bytes of one training step per patch: the pair tensor read once by the forward and once by the
    backward of each layer, the attention tape (probabilities written and read once: 2 x H K^2 4 B per layer),
    and - only when d pair_ctx is asked for - its read-modify-write (two more passes over the pair tensor per layer): 56.6 MB and
    107 MB per patch-step at K = 128, NL = 6."""
    pair = dims["NL"] * K * K * dims["C"] * 4
    tape = dims["NL"] * 2 * dims["H"] * K * K * 4  # probabilities written + read (the squared distances the tape also keeps are a choice)
    return 2 * pair + tape + (2 * pair if dpair else 0)


def other_configs(model, dims, flags):
    """Short secondary measurements on one GPU, reported next to the headline line (never part of `value`):
    BASELINE config 5 (K=256 long-CDR stress, 128 patches: per-GPU share of 512 on 4 GPUs) and config 4 (training step,
    128 patches per GPU: forward + HIP backward + Adam).  Failures are reported as strings, they never break the headline."""
    import gc

    from diffab_pytorch import _hip, synthetic as syn

    res = {}
    lib = _hip.lib()
    try:
        B, K, steps = 128, 256, 10
        inp = {k: v.cuda() for k, v in syn.patches(B, K, dims, seed=1).items()}
        seq, x, O = inp["seq_idx"].clone(), inp["translations"].clone(), inp["orientations"].clone()
        hd, w = model.denoiser.hip_dims(B, K), model.denoiser.hip_weights()
        sd_dev, tab = model._sched_on_device(), model._reverse_so3().struct()
        ws = _hip.workspace(lib.diffab_sample_workspace_bytes(C.byref(hd)))
        _hip.check(lib.diffab_sample_init(_hip.ptr(seq), _hip.ptr(x), _hip.ptr(O), _hip.ptr(inp["generation_mask"]), 7, 0, B, K, model.T,
                                          _hip.stream_ptr()), "sample_init")

        def loop(t_hi, n):
            _hip.check(lib.diffab_sample_loop(C.byref(hd), C.byref(w.struct), C.byref(sd_dev.struct), C.byref(tab), _hip.ptr(seq), _hip.ptr(x),
                                              _hip.ptr(O), _hip.ptr(inp["res_context_emb"]), _hip.ptr(inp["pair_context_emb"]),
                                              _hip.ptr(inp["generation_mask"]), 7, 0, t_hi, t_hi - n, _hip.ptr(ws), ws.numel(), flags,
                                              _hip.stream_ptr()), "sample_loop")

        loop(model.T, 2)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        loop(model.T - 2, steps)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        res["k256_sampling"] = {"patches": B, "K": K, "steps": steps, "ms_per_step": dt / steps * 1e3,
                                "residue_steps_per_s": B * K * steps / dt, "finite": bool(torch.isfinite(x).all())}
        del inp, seq, x, O, ws
    except Exception as ex:  # noqa: BLE001
        res["k256_sampling"] = f"failed: {type(ex).__name__}: {ex}"
    gc.collect()
    torch.cuda.empty_cache()
    try:  # BASELINE config 5 at its own batch: 512 patches of K = 256 on one GPU (two patches per CU: the patch-resident module launch)
        B, K, steps = 512, 256, 10
        g = torch.Generator(device="cuda").manual_seed(5)
        small = {k: v.cuda() for k, v in syn.patches(8, K, dims, seed=5).items()}  # states / masks / frames: host generator, tiled
        rep = lambda v: v.repeat((B // 8,) + (1,) * (v.dim() - 1)).contiguous()
        seq, x, O, gmask = rep(small["seq_idx"]), rep(small["translations"]), rep(small["orientations"]), rep(small["generation_mask"])
        res_c = torch.randn(B, K, dims["D"], device="cuda", generator=g)
        pair_c = torch.randn(B, K, K, dims["C"], device="cuda", generator=g)  # 8.6 GB: generated on the device
        hd, w = model.denoiser.hip_dims(B, K), model.denoiser.hip_weights()
        sd_dev, tab = model._sched_on_device(), model._reverse_so3().struct()
        ws = _hip.workspace(lib.diffab_sample_workspace_bytes(C.byref(hd)))
        _hip.check(lib.diffab_sample_init(_hip.ptr(seq), _hip.ptr(x), _hip.ptr(O), _hip.ptr(gmask), 7, 0, B, K, model.T, _hip.stream_ptr()),
                   "sample_init")
        forms = {}
        for name, fl in (("module_launch", flags | _hip.FLAG_PERSISTENT_MODULE), ("per_layer_launches", flags | _hip.FLAG_MULTI_LAUNCH)):
            for n_ in (2, steps):  # warm-up, then timed
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                _hip.check(lib.diffab_sample_loop(C.byref(hd), C.byref(w.struct), C.byref(sd_dev.struct), C.byref(tab), _hip.ptr(seq), _hip.ptr(x),
                                                  _hip.ptr(O), _hip.ptr(res_c), _hip.ptr(pair_c), _hip.ptr(gmask), 7, 0, model.T, model.T - n_,
                                                  _hip.ptr(ws), ws.numel(), fl, _hip.stream_ptr()), "sample_loop")
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
            forms[name] = dt / steps * 1e3  # (includes the once-per-call plane split of the 8.6 GB pair context)
        res["k256_sampling_b512"] = {"patches": B, "K": K, "steps": steps, "ms_per_step": forms["per_layer_launches"],
                                     "ms_per_step_module_launch": forms["module_launch"],
                                     "residue_steps_per_s": B * K / (forms["per_layer_launches"] * 1e-3), "finite": bool(torch.isfinite(x).all()),
                                     "what": "BASELINE config 5's batch on one GPU; per-layer launches (the sampler's choice at K = 256) and the "
                                             "patch-resident module launch (DIFFAB_FLAG_PERSISTENT_MODULE); both include the per-call pair-plane split"}
        del small, seq, x, O, gmask, res_c, pair_c, ws
    except Exception as ex:  # noqa: BLE001
        res["k256_sampling_b512"] = f"failed: {type(ex).__name__}: {ex}"
    gc.collect()
    torch.cuda.empty_cache()
    try:  # the headline shape with DIFFAB_FLAG_SKIP_UNUSED_ROWS: same samples bit for bit, the last layer's attention only for the row
        # tiles that hold a generated residue.  NOT the headline: the work skipped depends on the mask (synthetic: one CDR-like segment
        # of 5..20 residues per patch), the headline runs every row of every layer as the reference does.
        B, K, steps = 256, 128, 20
        inp = {k: v.cuda() for k, v in syn.patches(B, K, dims, seed=0, coord_sigma=10.0).items()}
        seq, x, O = inp["seq_idx"].clone(), inp["translations"].clone(), inp["orientations"].clone()
        hd, w = model.denoiser.hip_dims(B, K), model.denoiser.hip_weights()
        sd_dev, tab = model._sched_on_device(), model._reverse_so3().struct()
        ws = _hip.workspace(lib.diffab_sample_workspace_bytes(C.byref(hd)))
        _hip.check(lib.diffab_sample_init(_hip.ptr(seq), _hip.ptr(x), _hip.ptr(O), _hip.ptr(inp["generation_mask"]), 2024, 0, B, K, model.T,
                                          _hip.stream_ptr()), "sample_init")
        lean = {}
        for name, fl in (("all_rows", flags), ("skip_unused_rows", flags | _hip.FLAG_SKIP_UNUSED_ROWS), ("fp32_gemm", flags | _hip.FLAG_FP32_GEMM)):
            for n_ in (3, steps):  # warm-up, then timed
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                _hip.check(lib.diffab_sample_loop(C.byref(hd), C.byref(w.struct), C.byref(sd_dev.struct), C.byref(tab), _hip.ptr(seq), _hip.ptr(x),
                                                  _hip.ptr(O), _hip.ptr(inp["res_context_emb"]), _hip.ptr(inp["pair_context_emb"]),
                                                  _hip.ptr(inp["generation_mask"]), 2024, 0, model.T, model.T - n_, _hip.ptr(ws), ws.numel(), fl,
                                                  _hip.stream_ptr()), "sample_loop")
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
            lean[name] = dt / steps * 1e3  # (includes the once-per-call plane split, 0.5 ms over 20 steps, in both)
        tiles = inp["generation_mask"].view(B, K // 16, 16).any(-1).float().mean().item()
        res["skip_unused_rows"] = {"patches": B, "K": K, "steps": steps, "ms_per_step_all_rows": lean["all_rows"],
                                   "ms_per_step": lean["skip_unused_rows"], "last_layer_row_tiles_run": tiles,
                                   "what": "DIFFAB_FLAG_SKIP_UNUSED_ROWS (opt-in): bitwise the same samples; not the headline"}
        # the price of exact fp32 OPERANDS in the dense products (DIFFAB_FLAG_FP32_GEMM: f32-input MFMAs for the projections, to_out and the
        # MLPs instead of the three-term fp16 / six-term bf16 split products; per-layer launches - the module launch holds the fp16 tiles only)
        res["fp32_gemm_operands"] = {"patches": B, "K": K, "steps": steps, "ms_per_step": lean["fp32_gemm"], "ms_per_step_default": lean["all_rows"],
                                     "what": "DIFFAB_FLAG_FP32_GEMM (opt-in): exact fp32 operands in every dense product; not the headline"}
        del inp, seq, x, O, ws
    except Exception as ex:  # noqa: BLE001
        res["skip_unused_rows"] = f"failed: {type(ex).__name__}: {ex}"
    gc.collect()
    torch.cuda.empty_cache()
    try:  # BASELINE config 1: ONE K=128 patch through the whole 100-step reverse loop (eager launches, and one captured step replayed)
        B, K = 1, 128
        inp = {k: v.cuda() for k, v in syn.patches(B, K, dims, seed=3).items()}
        hd, w = model.denoiser.hip_dims(B, K), model.denoiser.hip_weights()
        sd_dev, tab = model._sched_on_device(), model._reverse_so3().struct()
        ws = _hip.workspace(lib.diffab_sample_workspace_bytes(C.byref(hd)))
        c1 = {"patches": B, "K": K, "steps": model.T}
        for name, fl in (("ms_per_trajectory", flags), ("ms_per_trajectory_graph", flags | _hip.FLAG_GRAPH_SAMPLER)):
            best = None
            for rep in range(3):
                seq, x, O = inp["seq_idx"].clone(), inp["translations"].clone(), inp["orientations"].clone()
                _hip.check(lib.diffab_sample_init(_hip.ptr(seq), _hip.ptr(x), _hip.ptr(O), _hip.ptr(inp["generation_mask"]), 9, 0, B, K, model.T,
                                                  _hip.stream_ptr()), "sample_init")
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                _hip.check(lib.diffab_sample_loop(C.byref(hd), C.byref(w.struct), C.byref(sd_dev.struct), C.byref(tab), _hip.ptr(seq), _hip.ptr(x),
                                                  _hip.ptr(O), _hip.ptr(inp["res_context_emb"]), _hip.ptr(inp["pair_context_emb"]),
                                                  _hip.ptr(inp["generation_mask"]), 9, 0, model.T, 0, _hip.ptr(ws), ws.numel(), fl,
                                                  _hip.stream_ptr()), "sample_loop")
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
                best = dt if best is None or dt < best else best
            c1[name] = best * 1e3
        c1["finite"] = bool(torch.isfinite(x).all())
        res["config1_one_patch"] = c1
        del inp, ws
    except Exception as ex:  # noqa: BLE001
        res["config1_one_patch"] = f"failed: {type(ex).__name__}: {ex}"
    # BASELINE config 4 at its per-GPU share (128 patches): three definitions of "a training step", each with its own algorithmic bytes
    #   contexts_given        contexts are constant inputs (no d pair_ctx): pair stream read by forward and backward + P / d2 tape
    #   contexts_given_dpair  SURVEY 8(d)'s definition: + the read-modify-write of d pair_ctx (2 more passes over the pair tensor)
    #   full_step_raw_batch   the reference-shaped step (:808-880): raw batch -> encode_context forward + backward -> all 2 538 468 parameters
    def train_leg(batch, steps=8, warm=4, dpair=False):
        opt = model.configure_optimizers()
        loss = None
        for it in range(warm + steps):
            if it == warm:
                torch.cuda.synchronize()
                t0 = time.perf_counter()
            opt.zero_grad(set_to_none=True)
            if dpair:
                batch["pair_context_emb"].grad = None
            loss = model.training_step(batch, it)
            loss.backward()
            opt.step()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps, bool(torch.isfinite(loss.detach()).item())

    try:
        B, K = 128, 128
        inp = syn.patches(B, K, dims, seed=2)
        batch = {"seq_idx": inp["seq_idx"].cuda(), "xyz": inp["translations"].cuda(), "orientations": inp["orientations"].cuda(),
                 "generation_mask": inp["generation_mask"].cuda(), "residue_mask": inp["residue_mask"].cuda(),
                 "res_context_emb": inp["res_context_emb"].cuda(), "pair_context_emb": inp["pair_context_emb"].cuda()}
        legs = {}
        for name, dpair in (("contexts_given", False), ("contexts_given_dpair", True)):
            batch["pair_context_emb"].requires_grad_(dpair)
            dt, fin = train_leg(batch, dpair=dpair)
            mb = train_bytes_per_patch(K, dims, dpair) / 1e6
            legs[name] = {"ms_per_step": dt * 1e3, "residue_steps_per_s": B * K / dt, "loss_finite": fin,
                          "algorithmic_MB_per_patch_step": mb, "hbm_frac": B * mb * 1e6 / dt / 1e9 / HBM_PEAK_GBPS}
        batch["pair_context_emb"].requires_grad_(False)
        del batch, inp
        gc.collect()
        torch.cuda.empty_cache()
        raw = {k: v.cuda() for k, v in syn.context_batch(B, K, n_atoms=15, seed=5, with_distmat=False).items()}
        raw.pop("distmat")  # (the reference's collate_fn does not produce it either, data.py:94-95: distances are taken from xyz)
        dt, fin = train_leg(raw)
        legs["full_step_raw_batch"] = {"ms_per_step": dt * 1e3, "residue_steps_per_s": B * K / dt, "loss_finite": fin,
                                       "what": "raw batch (15 atoms, chain ids) -> featurisation of xyz -> encode_context forward + backward -> "
                                               "noise + taped denoise forward + 3 losses + HIP backward -> Adam on all 2 538 468 parameters"}
        res["training_step"] = dict(patches=B, K=K, steps=8, **legs)
    except Exception as ex:  # noqa: BLE001
        res["training_step"] = f"failed: {type(ex).__name__}: {ex}"
    gc.collect()
    torch.cuda.empty_cache()
    return res


def train_bench(args, model, dims, rank, world, dist):
    """BASELINE config 4: B patches per GPU (1024 = 8 x 128), one training step = forward noise + taped denoise forward + three
    losses + HIP backward + all-reduce of the gradient buckets (in place, RCCL) + Adam.  The measured step (`value`) follows SURVEY
    8(d)'s definition: the contexts are grad-requiring inputs, so the backward also produces d pair_ctx (107 MB of algorithmic
    traffic per patch-step).  The same step with constant contexts (no d pair_ctx: 56.6 MB per patch-step) is timed behind it and
    reported as `contexts_constant` with its own roofline fraction."""
    from diffab_pytorch import _hip, distributed as D, synthetic as syn

    B, K = args.batch, args.k
    if args.attn_variant:  # A/B switches (tools/ab_train.sh): 8 = the six-term bf16 form of the dense products
        _hip.lib().diffab_debug_set_attn_variant(args.attn_variant)
    inp = syn.patches(B, K, dims, seed=2, first_patch=rank * B)
    batch = {"seq_idx": inp["seq_idx"].cuda(), "xyz": inp["translations"].cuda(), "orientations": inp["orientations"].cuda(),
             "generation_mask": inp["generation_mask"].cuda(), "residue_mask": inp["residue_mask"].cuda(),
             "res_context_emb": inp["res_context_emb"].cuda(), "pair_context_emb": inp["pair_context_emb"].cuda()}
    torch.manual_seed(1000 + rank)  # per-rank noise / timestep stream (the parameters above are identical on every rank)
    opt = model.configure_optimizers()
    loss = None

    def step(i):
        nonlocal loss
        opt.zero_grad(set_to_none=True)
        batch["pair_context_emb"].grad = None
        loss = model.training_step(batch, i)
        loss.backward()
        D.allreduce_gradients(model.parameters(), dist, flats=model.gradient_buckets())
        opt.step()

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(dpair):
        batch["pair_context_emb"].requires_grad_(dpair)
        for i in range(args.warmup):
            step(i)
        barrier()
        t0 = time.perf_counter()
        for i in range(args.steps):
            step(args.warmup + i)
        barrier()
        el = time.perf_counter() - t0
        if dist is not None:
            tmax = torch.tensor([el], device="cuda", dtype=torch.float64)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            el = float(tmax.item())
        return el

    elapsed = timed(True)
    had_dpair = batch["pair_context_emb"].grad is not None
    elapsed_const = timed(False)
    in_place = all(p.grad is None or any(f is not None and p.grad.untyped_storage().data_ptr() == f.untyped_storage().data_ptr()
                                         for f in model.gradient_buckets()) for p in model.denoiser.parameters())
    if rank == 0:
        value = world * B * K * args.steps / elapsed
        by, by_c = train_bytes_per_patch(K, dims, True), train_bytes_per_patch(K, dims, False)
        achieved = value / world / K * by / 1e9
        value_c = world * B * K * args.steps / elapsed_const
        print(json.dumps({
            "metric": "CDR-residue training-steps/sec (K=128 patch, forward + backward + Adam)", "value": value, "unit": "residue-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32 (fp16x3 / bf16x6 split-precision dense products on the 16-bit matrix cores, fp32 accumulate; fp32-accurate, DESIGN section 4)", "data": "synthetic",
            "config": {"workload": f"BASELINE config 4: training step, batch={B}/GPU synthetic K={K} patches, benchmark model NL=6; noise + taped "
                                   "forward + 3 losses + HIP backward incl. d pair_ctx + gradient all-reduce (RCCL, in-place buckets) + Adam; "
                                   "contexts given as grad-requiring inputs (SURVEY 8d)",
                       "patches_per_gpu": B, "K": K, "global_batch": world * B, "parallelism": f"data-parallel x{world}"},
            "roofline": {"kernel": "whole training step (no single dominant kernel: profiles/r04_train_kernel_stats.csv)", "bound": "hbm",
                         "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS, "traffic": None,
                         "algorithmic_bytes_per_patch_step": by},
            "contexts_constant": {"what": "the same step with constant contexts (no d pair_ctx)", "ms_per_step": elapsed_const / args.steps * 1e3,
                                  "residue_steps_per_s": value_c, "algorithmic_bytes_per_patch_step": by_c,
                                  "hbm_frac": value_c / world / K * by_c / 1e9 / HBM_PEAK_GBPS},
            "d_pair_ctx_computed": bool(had_dpair),
            "loss_finite": bool(torch.isfinite(loss.detach()).item()), "gradients_reduced_in_place": bool(in_place),
        }), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=256, help="patches per GPU")
    ap.add_argument("--k", type=int, default=128, help="residues per patch")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget", type=float, default=25.0, help="seconds of host time for the CPU baseline cases")
    ap.add_argument("--generic", action="store_true", help="force the generic (non-MFMA) kernels")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the short BASELINE config 4 / 5 measurements (N=1 only)")
    ap.add_argument("--train", action="store_true",
                    help="BASELINE config 4 instead of the sampling headline: training steps (noise + taped forward + 3 losses + HIP "
                         "backward + gradient all-reduce over RCCL + Adam), 128 patches per GPU unless --batch is given")
    ap.add_argument("--repeats", type=int, default=5,
                    help="the timed block of --steps steps is run this many times back to back (each bracketed by barrier + synchronize, "
                         "max over ranks); ms_per_step / value are the MEDIAN block, every block is listed in ms_per_step_runs")
    ap.add_argument("--multi-launch", action="store_true",
                    help="DIFFAB_FLAG_MULTI_LAUNCH: one launch per kernel of an IPA layer instead of the patch-resident module launch the "
                         "sampler chooses at this batch size (bitwise the same samples; for profiles of the separate kernels)")
    ap.add_argument("--attn-variant", type=int, default=0,
                    help="developer A/B switch: diffab_debug_set_attn_variant(v) before anything runs (16 = value planes, 8 = bf16x6 dense tiles)")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="N > 1 rehearsal on a 1-GPU box: every rank on cuda:0, gloo instead of RCCL (tests/test_gpu_two_ranks.py); "
                         "exercises the launch contract, sharding, gather and max-over-ranks timing - NOT a scaling measurement")
    args = ap.parse_args()
    if args.train and "--batch" not in sys.argv:
        args.batch = 128

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # a plain `python bench.py --gpus N`: start the N ranks ourselves, as a CHILD process (nothing above has touched the GPU:
        # `import torch` alone does not initialise HIP), forward its output and leave with its exit code - never exec
        import socket
        import subprocess

        with socket.socket() as s_:
            s_.bind(("127.0.0.1", 0))
            port = s_.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.run(cmd).returncode)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if args.rehearse_on_one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist

        if args.rehearse_on_one_gpu:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))  # nccl == RCCL on ROCm

    from diffab_pytorch import DiffAb, _hip, synthetic as syn
    from diffab_pytorch.distributed import gather_samples

    lib = _hip.lib()
    dims = dict(syn.BENCH_DIMS)
    B, K = args.batch, args.k
    torch.manual_seed(0)  # default init of the boundary module, identical on every rank
    model = DiffAb(dims["D"], dims["C"], dims["NL"], dims["DS"], dims["PQ"], dims["PV"], dims["H"]).cuda()
    first_patch = rank * B  # global patch ids: rank r owns [r*B, (r+1)*B)
    if args.train:
        train_bench(args, model, dims, rank, world, dist)
        if dist is not None:
            dist.destroy_process_group()
        return
    t_gen = time.perf_counter()
    inp = syn.patches(B, K, dims, seed=0, coord_sigma=10.0, first_patch=first_patch)  # CPU-generated, then copied
    dev = {k: v.cuda() for k, v in inp.items()}
    t_gen = time.perf_counter() - t_gen
    seq, x, O = dev["seq_idx"].clone(), dev["translations"].clone(), dev["orientations"].clone()
    gm, rc, pc = dev["generation_mask"], dev["res_context_emb"], dev["pair_context_emb"]

    hd = model.denoiser.hip_dims(B, K)
    w = model.denoiser.hip_weights()
    sd_dev = model._sched_on_device()
    tab = model._reverse_so3().struct()
    ws = _hip.workspace(lib.diffab_sample_workspace_bytes(C.byref(hd)))
    flags = _hip.FLAG_FORCE_GENERIC if args.generic else (_hip.FLAG_MULTI_LAUNCH if args.multi_launch else 0)
    if args.attn_variant:
        lib.diffab_debug_set_attn_variant(args.attn_variant)
    seed = 2024
    _hip.check(lib.diffab_sample_init(_hip.ptr(seq), _hip.ptr(x), _hip.ptr(O), _hip.ptr(gm), seed, first_patch, B, K, model.T,
                                      _hip.stream_ptr()), "sample_init")

    def run_steps(n, t_hi):
        """n reverse steps starting at timestep t_hi, wrapping T..1; every launch enqueued by the C-ABI loop."""
        t = t_hi
        while n > 0:
            m = min(n, t)
            _hip.check(lib.diffab_sample_loop(C.byref(hd), C.byref(w.struct), C.byref(sd_dev.struct), C.byref(tab), _hip.ptr(seq),
                                              _hip.ptr(x), _hip.ptr(O), _hip.ptr(rc), _hip.ptr(pc), _hip.ptr(gm), seed, first_patch, t, t - m,
                                              _hip.ptr(ws), ws.numel(), flags, _hip.stream_ptr()), "sample_loop")
            n -= m
            t = t - m if t - m > 0 else model.T
        return t

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    t_next = run_steps(args.warmup, model.T)
    if dist is not None:  # untimed: RCCL sets up its channels / buffers on the first collective of each kind
        gather_samples({"seq_idx": seq, "translations": x, "orientations": O}, dist)
    clocks_before = gpu_clocks() if rank == 0 else None
    lib.diffab_kernel_timer_enable(1)
    # The timed block (EXACTLY --steps steps + the all-gather, barrier + synchronize on both sides, max over ranks) is run --repeats
    # times back to back in this process: one number per round cannot tell a 3 % change from the box-to-box spread.
    runs = []
    for _rep in range(max(1, args.repeats)):
        barrier()
        t0 = time.perf_counter()
        t_next = run_steps(args.steps, t_next)
        samples = gather_samples({"seq_idx": seq, "translations": x, "orientations": O}, dist)  # RCCL all-gather when N > 1
        barrier()
        el = time.perf_counter() - t0
        if dist is not None:
            tmax = torch.tensor([el], device="cuda", dtype=torch.float64)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            el = float(tmax.item())
        runs.append(el)
    elapsed = sorted(runs)[len(runs) // 2]  # median block
    launches, total_ms = C.c_int64(0), C.c_double(0.0)
    _hip.check(lib.diffab_kernel_timer_read(C.byref(launches), C.byref(total_ms)), "kernel_timer_read")
    lib.diffab_kernel_timer_enable(0)
    clocks_after = gpu_clocks() if rank == 0 else None
    assert samples["translations"].shape[0] == world * B
    finite = bool(torch.isfinite(x).all() and torch.isfinite(O).all())
    # Which kernel the timer bracketed: the sampler runs the IPA module as ONE patch-resident launch per step when the batch fills the
    # chip (csrc/ipa_persistent.hip; bitwise the per-layer launches), else one attention launch per layer.  In the first case the
    # attention tile body is also timed as its own launch (DIFFAB_FLAG_MULTI_LAUNCH, 20 untimed-for-`value` steps) for comparison.
    total_steps = args.steps * len(runs)
    module_form = launches.value == total_steps
    attn_alone = None
    if module_form and rank == 0 and not args.generic:
        fl2 = flags | _hip.FLAG_MULTI_LAUNCH
        t_ = model.T
        for n_, timed in ((5, False), (20, True)):
            if timed:
                lib.diffab_kernel_timer_enable(1)
                torch.cuda.synchronize()
                t0_ = time.perf_counter()
            _hip.check(lib.diffab_sample_loop(C.byref(hd), C.byref(w.struct), C.byref(sd_dev.struct), C.byref(tab), _hip.ptr(seq), _hip.ptr(x),
                                              _hip.ptr(O), _hip.ptr(rc), _hip.ptr(pc), _hip.ptr(gm), seed, first_patch, t_, t_ - n_,
                                              _hip.ptr(ws), ws.numel(), fl2, _hip.stream_ptr()), "sample_loop")
            t_ -= n_
        torch.cuda.synchronize()
        ms_step_multi = (time.perf_counter() - t0_) / 20 * 1e3
        l2, ms2 = C.c_int64(0), C.c_double(0.0)
        _hip.check(lib.diffab_kernel_timer_read(C.byref(l2), C.byref(ms2)), "kernel_timer_read")
        lib.diffab_kernel_timer_enable(0)
        attn_alone = (l2.value, ms2.value / max(l2.value, 1), ms_step_multi)

    if rank == 0:
        value = world * B * K * args.steps / elapsed
        avg_ms = total_ms.value / max(launches.value, 1)
        alg = algorithmic_bytes_per_attention_launch(B, K, dims["D"], dims["C"]) * (dims["NL"] if module_form else 1)
        achieved = alg / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        traffic = None
        tpath = os.path.join(REPO, "profiles", "roofline_traffic.json")
        if os.path.exists(tpath):  # HBM bytes per launch from the committed rocprofv3 --pmc passes of this command
            try:
                tj = json.load(open(tpath))
                if tj.get("B") == B and tj.get("K") == K and ("ipa_module_persistent" in tj.get("kernel", "")) == module_form:
                    traffic = tj.get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "CDR-residue denoise-steps/sec (K=128 patch)",
            "value": value,
            "unit": "residue-steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            # every timed block of this process (each: --steps steps + gather, bracketed); ms_per_step / value = the median block
            "ms_per_step_runs": [r / args.steps * 1e3 for r in runs],
            "ms_per_step_min": min(runs) / args.steps * 1e3,
            "ms_per_step_median": elapsed / args.steps * 1e3,
            "gpu_clocks": {"before": clocks_before, "after": clocks_after},
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32 (bf16x6 / fp16x3 split-precision products on the 16-bit matrix cores, fp32 accumulate; fp32-accurate, DESIGN section 4)",
            "data": "synthetic",
            "config": {
                "workload": f"batch={B}/GPU synthetic K={K} patches, reverse sampling steps (T=100 schedule), benchmark model "
                            "D=128 C=64 NL=6 H=8 ds=32 P=8 (reference train.py:62-70), random-init weights",
                "patches_per_gpu": B, "K": K, "global_batch": world * B,
                "parallelism": f"patch-sharded x{world}" + (" (REHEARSAL: all ranks on one GPU over gloo, not a scaling measurement)" if args.rehearse_on_one_gpu else ""),
                "path": "generic" if args.generic else "mfma",
            },
            "residue_steps_per_s_per_gpu": value / world,
            # SURVEY 8(d) secondary figures: only the residues being generated (masks do not prune compute: every residue of a patch
            # costs the same), and the same rate against the fp32 compute roof (7.1 MFLOP per residue-step at K = 128; 157 TFLOP/s
            # vendor fp32 MFMA / VALU peak) - which lies BELOW the HBM roof for this path (22 M vs 40.5 M residue-steps/s per GPU)
            "generated_residue_steps_per_s": value * float(gm.float().mean()),
            "fp32_compute_frac": value / world * algorithmic_flops_per_residue_step(K, dims) / 157e12,
            "whole_path_hbm_frac": value / world * algorithmic_bytes_per_residue_step(K, dims["D"], dims["C"], dims["NL"]) / 1e9
                                   / HBM_PEAK_GBPS,
            "roofline": {
                "kernel": ("ipa_module_persistent_kernel: one patch-resident launch per denoiser forward - the embedding MLP, the NL = %d "
                           "layers of the IPA module (per layer: six projections, eight attention row tiles, to_out) and the three heads "
                           "of each patch; algorithmic bytes = NL x the pair-embedding stream" % dims["NL"]) if module_form else
                          "ipa_attention (pair-embedding stream: bias + softmax + attn-weighted pair/scalar/point sums)",
                "bound": "hbm",
                "achieved": achieved,
                "peak": HBM_PEAK_GBPS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBPS,
                "traffic": traffic,  # the PMC file under profiles/ is for the fused kernel
                # not measured in this run: rocprofv3 --pmc passes of this same command (FETCH_SIZE x 2 + WRITE_SIZE per launch,
                # tools/profile_round.sh), committed as profiles/roofline_traffic.json
                "traffic_source": None if traffic is None else "profiles/roofline_traffic.json (committed rocprofv3 --pmc passes, not this run)",
                "launches": launches.value,
                "avg_launch_ms": avg_ms,
                "algorithmic_bytes_per_launch": alg,
            },
            "outputs_finite": finite,
            "input_gen_s": t_gen,
        }
        if attn_alone is not None:
            a1 = algorithmic_bytes_per_attention_launch(B, K, dims["D"], dims["C"])
            out["roofline_attention_launch"] = {
                "what": "the attention tile body of the module kernel as its own launch per layer (DIFFAB_FLAG_MULTI_LAUNCH; bitwise the "
                        "same samples), 20 steps outside the timed blocks: the pair-stream kernel alone, as rounds 1-4 reported it",
                "bound": "hbm", "achieved": a1 / (attn_alone[1] * 1e-3) / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": a1 / (attn_alone[1] * 1e-3) / 1e9 / HBM_PEAK_GBPS, "launches": attn_alone[0], "avg_launch_ms": attn_alone[1],
                "algorithmic_bytes_per_launch": a1, "ms_per_step_of_this_form": attn_alone[2],
            }
        if world == 1 and not args.no_other_configs and not args.generic:
            out["other_configs"] = other_configs(model, dims, flags)
        if world == 1 and not args.no_cpu_baseline:
            torch.manual_seed(0)
            sd = {k: v.detach().cpu() for k, v in model.denoiser.state_dict().items()}
            out["cpu_baseline"] = cpu_baseline(dims, sd, K, seed=seed, budget_s=args.cpu_budget)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
