"""Two ranks with REAL compute on one GPU (the N > 1 code paths that the gloo tests on CPU exercise with fake tensors only).

Both ranks run on cuda:0 with the gloo backend (`--rehearse-on-one-gpu` in bench.py / train.py): this box has one GPU and RCCL
wants one device per rank.  What is covered is everything except the RCCL transport itself - patch sharding by global id, Philox
noise keyed by the global patch id, gather_samples, the in-place all-reduce of the HIP backward's flat gradient buckets, the
harness' rank seeding / sharded loader, and bench.py's launch contract under torch.distributed.run.  NOT a scaling measurement.
"""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch

from diffab_pytorch import synthetic as syn

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _small_model(seed=0, NL=2):
    from diffab_pytorch import DiffAb

    d = dict(syn.BENCH_DIMS, NL=NL)
    torch.manual_seed(seed)
    model = DiffAb(d["D"], d["C"], d["NL"], d["DS"], d["PQ"], d["PV"], d["H"]).cuda()
    model.denoiser.load_state_dict(syn.denoiser_state_dict(d, seed=seed, prefix=""))
    return d, model


def _train_batch(dims, B, K, seed):
    inp = syn.patches(B, K, dims, seed=seed, coord_sigma=6.0)
    return {"seq_idx": inp["seq_idx"], "xyz": inp["translations"], "orientations": inp["orientations"],
            "generation_mask": inp["generation_mask"], "residue_mask": inp["residue_mask"],
            "res_context_emb": inp["res_context_emb"], "pair_context_emb": inp["pair_context_emb"]}


def _worker(rank, world, port, mode, out_dir):
    """One rank (spawned): gloo rendezvous on 127.0.0.1, device cuda:0."""
    sys.path.insert(0, os.path.join(REPO, "diffab-pytorch_amd"))
    import torch.distributed as dist

    from diffab_pytorch import distributed as D

    torch.cuda.set_device(0)
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    dims, model = _small_model()
    if mode == "sample":
        B, K = 8, 128
        inp = syn.patches(B, K, dims, seed=5, coord_sigma=8.0)
        lo, hi = D.shard_range(B, rank, world)
        sl = slice(lo, hi)
        local = model.sample(inp["seq_idx"][sl].cuda(), inp["translations"][sl].cuda(), inp["orientations"][sl].cuda(),
                             res_context_emb=inp["res_context_emb"][sl].cuda(), pair_context_emb=inp["pair_context_emb"][sl].cuda(),
                             generation_mask=inp["generation_mask"][sl].cuda(), seed=77, first_patch=lo, t_start=100, t_stop=90)
        full = D.gather_samples(local, dist)
        if rank == 0:
            torch.save({k: v.cpu() for k, v in full.items()}, os.path.join(out_dir, "sample.pt"))
    else:  # two data-parallel training steps: local backward, in-place all-reduce of the flat buckets, Adam
        B, K = 4, 64
        batch = _train_batch(dims, B, K, seed=9)
        lo, hi = D.shard_range(B, rank, world)
        mine = {k: v[lo:hi].cuda() for k, v in batch.items()}
        opt = model.configure_optimizers()
        for step in range(2):
            torch.manual_seed(1000 * step + rank)  # timesteps + forward-noise seeds of THIS rank's shard
            opt.zero_grad(set_to_none=True)
            loss = model.training_step(mine, step)
            loss.backward()
            D.allreduce_gradients(model.parameters(), dist, flats=model.gradient_buckets())
            opt.step()
        torch.save({n: p.detach().cpu() for n, p in model.named_parameters()}, os.path.join(out_dir, f"params_rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def _spawn(mode, out_dir):
    import torch.multiprocessing as mp

    port = _free_port()
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_worker, args=(r, 2, port, mode, str(out_dir))) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(600)
        assert p.exitcode == 0, (mode, p.exitcode)


def test_two_ranks_sampling_shards_equal_the_single_process_result(tmp_path):
    _spawn("sample", tmp_path)
    got = torch.load(os.path.join(tmp_path, "sample.pt"))
    dims, model = _small_model()
    inp = syn.patches(8, 128, dims, seed=5, coord_sigma=8.0)
    want = model.sample(inp["seq_idx"].cuda(), inp["translations"].cuda(), inp["orientations"].cuda(),
                        res_context_emb=inp["res_context_emb"].cuda(), pair_context_emb=inp["pair_context_emb"].cuda(),
                        generation_mask=inp["generation_mask"].cuda(), seed=77, first_patch=0, t_start=100, t_stop=90)
    for k in ("seq_idx", "translations", "orientations"):
        assert torch.equal(got[k], want[k].cpu()), k  # bitwise: noise is keyed by the global patch id, patches are independent


def test_two_ranks_training_steps_equal_the_averaged_single_process_run(tmp_path):
    _spawn("train", tmp_path)
    p0 = torch.load(os.path.join(tmp_path, "params_rank0.pt"))
    p1 = torch.load(os.path.join(tmp_path, "params_rank1.pt"))
    for n in p0:
        assert torch.equal(p0[n], p1[n]), n  # every rank applied the same all-reduced gradient
    # the same two steps in ONE process: each shard's backward with that rank's seeds, gradients averaged by hand
    dims, model = _small_model()
    batch = _train_batch(dims, 4, 64, seed=9)
    opt = model.configure_optimizers()
    params = list(model.parameters())
    for step in range(2):
        acc = [torch.zeros_like(p) for p in params]
        for r in range(2):
            torch.manual_seed(1000 * step + r)
            opt.zero_grad(set_to_none=True)
            model.training_step({k: v[2 * r:2 * r + 2].cuda() for k, v in batch.items()}, step).backward()
            for a, p in zip(acc, params):
                if p.grad is not None:
                    a += p.grad
        opt.zero_grad(set_to_none=True)
        for a, p in zip(acc, params):
            p.grad = a / 2
        opt.step()
    worst = 0.0
    for n, p in model.named_parameters():
        if not n.startswith("denoiser."):
            continue  # contexts are given: the encoders receive no gradient in this batch
        err = float((p.detach().cpu() - p0[n]).abs().max() / p0[n].abs().max().clamp_min(1e-30))
        worst = max(worst, err)
        assert err < 1e-5, (n, err)  # float atomics in the weight-gradient kernels: not bitwise run to run
    print(f"two ranks vs one process after two Adam steps: worst parameter difference {worst:.2e} (relative to the parameter maximum)")


def test_bench_two_ranks_launch_contract(tmp_path):
    """bench.py exactly as the driver launches it for N = 2 (torch.distributed.run, one JSON line on rank 0), rehearsed on one GPU."""
    port = _free_port()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--batch", "8",
           "--rehearse-on-one-gpu"]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    res = subprocess.run(cmd, cwd=str(tmp_path), env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]  # ONE JSON line, from rank 0
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["warmup"] == 1 and out["scaling"] == "weak"
    assert out["config"]["global_batch"] == 16 and out["config"]["patches_per_gpu"] == 8 and "REHEARSAL" in out["config"]["parallelism"]
    assert abs(out["value"] - 2 * 8 * 128 * 3 / (out["ms_per_step"] * 3e-3)) < 1e-6 * out["value"]  # whole-job aggregate over both ranks
    assert out["outputs_finite"] and "roofline" in out and "cpu_baseline" not in out  # (the CPU baseline is an N = 1 leg)
    assert len(out["ms_per_step_runs"]) == 5 and out["ms_per_step_min"] <= out["ms_per_step"] == out["ms_per_step_median"]
    # the plain form (no launcher, no WORLD_SIZE in the environment): bench.py starts the ranks itself as a child process
    env_plain = {k: v for k, v in env.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    plain = [sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--batch", "8", "--repeats", "2",
             "--rehearse-on-one-gpu"]
    res = subprocess.run(plain, cwd=str(tmp_path), env=env_plain, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["global_batch"] == 16 and len(out["ms_per_step_runs"]) == 2
    # the training leg under the same launcher
    cmd_t = cmd[:-1] + ["--train", "--rehearse-on-one-gpu"]
    cmd_t[cmd_t.index("--batch") + 1] = "4"
    res = subprocess.run(cmd_t, cwd=str(tmp_path), env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["global_batch"] == 8
