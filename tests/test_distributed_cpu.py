"""CPU, world_size 2 over gloo: the N>1 path of the sampler - shard ranges, bit-exact packing, and the single
all-gather that collects sampled structures in global patch order (ragged shards included)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from diffab_pytorch.distributed import gather_samples, pack_samples, shard_range, unpack_samples


def test_shard_range_partitions():
    for n in (0, 1, 7, 256, 2048, 2049):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
    assert shard_range(2048, 3, 8) == (768, 1024)


def _fake(lo, hi, K=16):
    g = torch.Generator().manual_seed(1234)
    seq = torch.randint(0, 21, (10, K), generator=g)
    x = torch.randn(10, K, 3, generator=g)
    O = torch.randn(10, K, 3, 3, generator=g)
    x[0, 0, 0] = float("nan")  # packing must be a bit copy, NaN payloads included
    return {"seq_idx": seq[lo:hi], "translations": x[lo:hi], "orientations": O[lo:hi]}


def test_pack_roundtrip_bit_exact():
    s = _fake(0, 10)
    r = unpack_samples(pack_samples(s))
    assert torch.equal(r["seq_idx"], s["seq_idx"])
    assert torch.equal(r["translations"].view(torch.int32), s["translations"].view(torch.int32))
    assert torch.equal(r["orientations"].view(torch.int32), s["orientations"].view(torch.int32))
    assert pack_samples(s).shape == (10, 16, 14) and pack_samples(s).element_size() * 14 * 128 == 7168  # 7 168 B / K=128 patch


def _worker(rank, world, port, n, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lo, hi = shard_range(n, rank, world)
        sizes = [shard_range(n, r, world)[1] - shard_range(n, r, world)[0] for r in range(world)]
        out = gather_samples(_fake(lo, hi), dist, sizes=sizes)
        full = _fake(0, n)
        ok = all(torch.equal(out[k].view(torch.int32) if out[k].dtype.is_floating_point else out[k],
                             full[k].view(torch.int32) if full[k].dtype.is_floating_point else full[k]) for k in full)
        q.put((rank, ok, tuple(out["translations"].shape)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n", [10, 7])
def test_gather_world2_gloo(n):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res), res
    assert all(shape[0] == n for _, _, shape in res)


def _grad_worker(rank, world, port, q):
    from diffab_pytorch.distributed import allreduce_gradients

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(0)
        ps = [torch.nn.Parameter(torch.randn(5, 3)), torch.nn.Parameter(torch.randn(7)), torch.nn.Parameter(torch.randn(2, 2))]
        ps[2].grad = None  # parameters without a gradient are skipped
        g = torch.Generator().manual_seed(100 + rank)
        for p in ps[:2]:
            p.grad = torch.randn(p.shape, generator=g)
        allreduce_gradients(ps, dist)
        want = []
        for i, p in enumerate(ps[:2]):
            acc = torch.zeros_like(p)
            for r in range(world):
                gr = torch.Generator().manual_seed(100 + r)
                vals = [torch.randn(q_.shape, generator=gr) for q_ in ps[:2]]
                acc += vals[i]
            want.append(acc / world)
        ok = all(torch.allclose(p.grad, w_, atol=1e-6) for p, w_ in zip(ps[:2], want)) and ps[2].grad is None
        # flat buckets (what the HIP backward hands out): gradients that are views of a bucket are reduced IN PLACE, on the bucket,
        # with no packed copy; a gradient outside every bucket still goes through the packed path in the same call
        g = torch.Generator().manual_seed(200 + rank)
        flat = torch.randn(64 + 64 + 64, generator=g)          # 3 aligned slots, the last one unused padding
        loose = torch.randn(4, generator=g)
        qs = [torch.nn.Parameter(torch.zeros(5, 3)), torch.nn.Parameter(torch.zeros(7)), torch.nn.Parameter(torch.zeros(4))]
        qs[0].grad, qs[1].grad, qs[2].grad = flat[0:15].view(5, 3), flat[64:71], loose.clone()
        ptrs = [p_.grad.data_ptr() for p_ in qs]
        allreduce_gradients(qs, dist, flats=[flat, None])
        tot_flat, tot_loose = torch.zeros(192), torch.zeros(4)
        for r in range(world):
            gr = torch.Generator().manual_seed(200 + r)
            tot_flat += torch.randn(192, generator=gr)
            tot_loose += torch.randn(4, generator=gr)
        ok = ok and torch.allclose(flat, tot_flat / world, atol=1e-6) and torch.allclose(qs[2].grad, tot_loose / world, atol=1e-6)
        ok = ok and torch.allclose(qs[0].grad, (tot_flat / world)[0:15].view(5, 3), atol=1e-6)
        ok = ok and ptrs == [p_.grad.data_ptr() for p_ in qs] and qs[0].grad.untyped_storage().data_ptr() == flat.untyped_storage().data_ptr()
        q.put((rank, ok))
    finally:
        dist.destroy_process_group()


def test_gradient_allreduce_world2_gloo():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_grad_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok in res), res
