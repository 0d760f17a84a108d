"""Patch sharding over ranks (one process per GPU, torch.distributed; backend "nccl" is RCCL on ROCm).

Patches are independent everywhere on the hot path (reference: every einsum carries the batch index,
diffab_pytorch.py:416-457), so the path shards with NO data-path collective: rank r owns a contiguous range of
global patch ids, noise is keyed by the global id (csrc/philox.h), and the only exchange is ONE all-gather of the
sampled structures at the end - 7 168 B per K=128 patch (s int64 + x 3 f32 + O 9 f32 per residue), a single
fixed-size buffer per rank so RCCL moves it in one collective over xGMI.
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import torch


def shard_range(n_patches: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous, even split of [0, n_patches): the first n % world ranks get one extra patch."""
    q, r = divmod(n_patches, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def pack_samples(s: Dict[str, torch.Tensor]) -> torch.Tensor:
    """(B,K) int64 + (B,K,3) f32 + (B,K,3,3) f32 -> one (B,K,14) int32 buffer (bit-exact round trip)."""
    seq = s["seq_idx"].contiguous().view(torch.int32).view(*s["seq_idx"].shape, 2)
    x = s["translations"].contiguous().view(torch.int32)
    O = s["orientations"].contiguous().view(torch.int32).flatten(-2)
    return torch.cat([seq, x, O], dim=-1).contiguous()


def unpack_samples(buf: torch.Tensor) -> Dict[str, torch.Tensor]:
    B, K = buf.shape[:2]
    return {
        "seq_idx": buf[..., 0:2].contiguous().view(torch.int64).view(B, K),
        "translations": buf[..., 2:5].contiguous().view(torch.float32),
        "orientations": buf[..., 5:14].contiguous().view(torch.float32).view(B, K, 3, 3),
    }


def gather_samples(local: Dict[str, torch.Tensor], dist=None, sizes: Optional[list] = None) -> Dict[str, torch.Tensor]:
    """All ranks' samples in global patch order.  `dist` is torch.distributed (initialised) or None for one process.
    Equal shard sizes use one all_gather_into_tensor; ragged shards (sizes = patches per rank) pad to the maximum."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return local
    world = dist.get_world_size()
    buf = pack_samples(local)
    B = buf.shape[0]
    if sizes is None:
        sizes = [B] * world
    Bmax = max(sizes)
    if B < Bmax:
        buf = torch.cat([buf, buf.new_zeros(Bmax - B, *buf.shape[1:])])
    dev = buf.device
    if buf.is_cuda and dist.get_backend() == "gloo":  # gloo has no device all-gather: stage through the host (CPU rehearsals of the N > 1 path)
        buf = buf.cpu()
    out = buf.new_empty(world * Bmax, *buf.shape[1:])
    dist.all_gather_into_tensor(out, buf.contiguous())
    out = out.to(dev)
    parts = [out[r * Bmax: r * Bmax + sizes[r]] for r in range(world)]
    return unpack_samples(torch.cat(parts))


def allreduce_gradients(params, dist=None, average: bool = True, flats=None) -> None:
    """Data-parallel gradient exchange for the training step: all-reduce (sum) of flat fp32 gradient buckets, then
    / world.  Each rank normalises its losses by its own number of masked residues (reference diffab_pytorch.py:868-878 on the
    local shard), so this is the usual DDP mean of per-rank means.

    `flats`: the flat buffers the HIP backward passes wrote their gradients into (one per module: DiffAb.gradient_buckets()).
    After `zero_grad(set_to_none=True)` every `p.grad` is a view of one of them, so they are reduced IN PLACE - three collectives
    (7.9 MB denoiser + 1.5 MB + 0.7 MB encoders), no gather / scatter copies.  Gradients that are not views of a bucket (another
    producer, accumulated grads) go through one packed bucket as before.  On the 8-GPU xGMI mesh RCCL moves each bucket over all 7
    links of a GPU; no per-tensor collectives."""
    params = [p for p in params if p.grad is not None]
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1 or not params:
        return
    world = dist.get_world_size()
    covered = set()
    for flat in (flats or []):
        if flat is None:
            continue
        base = flat.untyped_storage().data_ptr()
        lo, hi = flat.data_ptr(), flat.data_ptr() + flat.numel() * flat.element_size()
        mine = [p for p in params if p.grad.untyped_storage().data_ptr() == base and lo <= p.grad.data_ptr() and
                p.grad.data_ptr() + p.grad.numel() * p.grad.element_size() <= hi and p.grad.is_contiguous()]
        if not mine:
            continue
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        if average:
            flat /= world
        covered.update(id(p) for p in mine)
    rest = [p for p in params if id(p) not in covered]
    if not rest:
        return
    packed = torch.cat([p.grad.reshape(-1).float() for p in rest])
    dist.all_reduce(packed, op=dist.ReduceOp.SUM)
    if average:
        packed /= world
    off = 0
    for p in rest:
        n = p.grad.numel()
        p.grad.copy_(packed[off:off + n].view_as(p.grad))
        off += n
