"""Trainer harness for the MI355X engine: the caller of the hot path that replaces the reference's Lightning script
(reference diffab_pytorch/train.py:14-113).

Same command-line flags and the same model / optimiser construction as the reference:

* flags of ``parse_argument`` (train.py:14-43): ``--meta --data-dir --val-pct --cdrs -b/--bsz -e/--epochs
  -l/--learning-rate -s/--seed --no-wandb``; plus ``--gradient-clip-val`` (the reference passes ``args.gradient_clip_val``
  to its Trainer at train.py:102 without ever defining the flag - an AttributeError upstream) and the three data sources
  this build can actually serve: ``--patch-dir`` (``.pt`` patches in the format ``preprocess_pdb.py:67-80`` writes),
  ``--synthetic N`` (seeded synthetic patches, ``synthetic.context_batch``) and nothing else: the reference's
  ``--meta/--data-dir`` route parses PDB files through ``protstruc`` (data.py:67-98), which is not part of this path.
* model hyper-parameters of train.py:62-80; Adam from ``DiffAb.configure_optimizers`` (diffab_pytorch.py:925-931);
* the loss keys logged by ``training_step`` / ``validation_step`` (diffab_pytorch.py:889-921), written as JSON lines;
* checkpoints ``{"state_dict": ..., "optimizer": ..., "epoch": ..., "global_step": ...}`` - the Lightning layout, with the
  parameter names of SURVEY Appendix B.3, so a reference checkpoint loads here and vice versa.

Multi-GPU: one process per GPU under ``python -m torch.distributed.run`` (backend "nccl" = RCCL).  Patches are independent, so
every rank runs forward + backward on its contiguous shard of each batch and the only exchange is the all-reduce of the flat
gradient buckets the HIP backward wrote (denoiser + the two encoders, reduced in place: ``distributed.allreduce_gradients``);
parameters stay identical because every rank applies the same averaged gradient with the same optimiser state.
"""
from __future__ import annotations

import argparse
import glob
import json
import os
import time
from typing import Dict, Iterator, List, Optional

import torch

MODEL_HPARAMS = dict(d_residue_emb=128, d_pair_emb=64, n_ipa_layers=6, d_scalar_per_head=32, n_query_point_per_head=8,
                     n_value_point_per_head=8, n_head=8)  # reference train.py:62-70

# keys of one preprocessed patch (reference preprocess_pdb.py:67-80); distmat is commented out upstream ("171M"): the pair kernel
# takes the (plain Euclidean atom-atom) distances from xyz instead
PATCH_KEYS = ("xyz", "orientations", "backbone_dihedrals", "backbone_dihedrals_mask", "pairwise_dihedrals", "atom_mask", "seq_idx",
              "chain_idx", "residue_idx", "residue_mask")


def parse_argument(argv: Optional[List[str]] = None) -> argparse.Namespace:
    parser = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    parser.add_argument("--meta", default=None, help="Metadata file for train/validation data (reference flag; needs protstruc)")
    parser.add_argument("--data-dir", default=None, help="Directory containing PDB files (reference flag; needs protstruc)")
    parser.add_argument("--val-pct", type=float, default=0.1, help="Proportion of validation data to use.")
    parser.add_argument("--cdrs", nargs="+", type=str, default=["H3"])
    parser.add_argument("-b", "--bsz", type=int, default=128, help="Batch size (global, split over ranks)")
    parser.add_argument("-e", "--epochs", type=int, default=60, help="Number of epochs to train")
    parser.add_argument("-l", "--learning-rate", type=float, default=0.01,
                        help="Learning rate.  As upstream the default is NOT applied: train.py never passes it to DiffAb, whose own "
                             "default 1e-4 (diffab_pytorch.py:637) is what Adam gets; give the flag explicitly to override")
    parser.add_argument("-s", "--seed", type=int, default=42, help="Random seed")
    parser.add_argument("--no-wandb", action="store_true", default=False, help="Don't use wandb for logging (never used here)")
    # --- beyond the reference
    parser.add_argument("--gradient-clip-val", type=float, default=None, help="Global gradient-norm clip (train.py:102 intended it)")
    parser.add_argument("--patch-dir", default=None, help="Directory of .pt patches written by preprocess_pdb.py")
    parser.add_argument("--synthetic", type=int, default=0, help="Train on this many seeded synthetic patches")
    parser.add_argument("--k", type=int, default=128, help="Residues per synthetic patch")
    parser.add_argument("--ckpt-dir", default=None, help="Write checkpoints here (one per epoch + last.ckpt)")
    parser.add_argument("--resume", default=None, help="Checkpoint to resume from")
    parser.add_argument("--log", default=None, help="JSON-lines log file (default: stdout on rank 0)")
    parser.add_argument("--max-steps", type=int, default=0, help="Stop after this many optimiser steps (0 = run all epochs)")
    parser.add_argument("--cdr-mask-key", default="generation_mask",
                        help="Key of the boolean (1, K) CDR mask inside each patch file.  preprocess_pdb.py:67-80 stores none (upstream "
                             "derives it from the PDB numbering with protstruc, data.py:92), so patches must be augmented with one")
    parser.add_argument("--rehearse-on-one-gpu", action="store_true",
                        help="N > 1 rehearsal on a 1-GPU box: every rank on cuda:0, gloo instead of RCCL (tests/test_gpu_two_ranks.py)")
    return parser.parse_args(argv)


def pairwise_atom_distances(xyz: torch.Tensor) -> torch.Tensor:
    """(B,K,A,3) -> (B,K,K,A,A) Euclidean distances: the `distmat` the reference computes with protstruc (data.py:76) and then
    leaves out of its batches; PairEmbedding.forward (diffab_pytorch.py:220-312) needs it.  Host-side statement of what
    diffab_pair_embedding_xyz_fwd computes in the kernel (used by the tests; the training path never builds this tensor)."""
    d = xyz[:, :, None, :, None, :] - xyz[:, None, :, None, :, :]
    return d.square().sum(-1).sqrt()


def load_patch(path: str, cdr_mask_key: str = "generation_mask") -> Dict[str, torch.Tensor]:
    """One patch file of preprocess_pdb.py:67-80 -> tensors with a leading batch dimension of 1."""
    data = torch.load(path, map_location="cpu")
    missing = [k for k in PATCH_KEYS if k not in data]
    if missing:
        raise KeyError(f"{path}: patch file lacks {missing} (expected the keys of reference preprocess_pdb.py:67-80)")
    out = {k: data[k] for k in PATCH_KEYS}
    if cdr_mask_key not in data:
        # preprocess_pdb.py:67-80 writes no CDR mask and it cannot be rebuilt from the patch (upstream takes it from the PDB
        # numbering through protstruc, data.py:92: get_cdr_mask(subset=cdrs)).  An all-false mask would make every loss 0/0.
        raise KeyError(f"{path}: no '{cdr_mask_key}' entry - add a boolean (1, K) mask of the residues to generate to the patch "
                       f"file (or name its key with --cdr-mask-key); the reference's patch format has none")
    mask = data[cdr_mask_key].bool()
    if mask.shape != data["residue_mask"].shape:
        raise ValueError(f"{path}: '{cdr_mask_key}' has shape {tuple(mask.shape)}, expected {tuple(data['residue_mask'].shape)}")
    if not bool((mask & data["residue_mask"].bool()).any()):
        raise ValueError(f"{path}: '{cdr_mask_key}' selects no valid residue (the three losses divide by the number of masked residues)")
    out["generation_mask"] = mask
    return out


def collate(patches: List[Dict[str, torch.Tensor]]) -> Dict[str, torch.Tensor]:
    """Stack single-patch dicts (each tensor has a leading 1) into a batch dict with the reference's keys."""
    batch = {}
    for k in patches[0]:
        if k == "residue_idx":
            batch[k] = patches[0][k]
        else:
            batch[k] = torch.cat([p[k] for p in patches], dim=0)
    return batch  # no distmat: DiffAb.encode_context takes the distances from xyz inside the pair kernel


class PatchSource:
    """Batches of the reference's batch dict from one of the supported sources, sharded over ranks by contiguous ranges."""

    def __init__(self, args: argparse.Namespace, rank: int, world: int, split: str):
        from . import distributed as D, synthetic as syn

        self.args, self.rank, self.world, self.split = args, rank, world, split
        self._shard = D.shard_range
        self._syn = syn
        if args.patch_dir:
            files = sorted(glob.glob(os.path.join(args.patch_dir, "*.pt")))
            if not files:
                raise FileNotFoundError(f"no .pt patches under {args.patch_dir}")
            g = torch.Generator().manual_seed(args.seed)
            order = torch.randperm(len(files), generator=g).tolist()  # reference: meta.sample(frac=1, random_state=seed), train.py:82
            n_val = int(len(files) * args.val_pct)
            keep = order[len(order) - n_val:] if split == "val" else order[: len(order) - n_val]
            self.files = [files[i] for i in keep]
            self.n = len(self.files)
        elif args.synthetic > 0:
            n_val = int(args.synthetic * args.val_pct)
            self.n = n_val if split == "val" else args.synthetic - n_val
            self.offset = args.synthetic - n_val if split == "val" else 0
        else:
            raise SystemExit("no usable data source: give --patch-dir DIR or --synthetic N.  The reference's --meta/--data-dir route "
                             "parses PDB files with protstruc (data.py:67-98), which this build does not include.")

    def __len__(self) -> int:
        return (self.n + self.args.bsz - 1) // self.args.bsz

    def batches(self, epoch: int) -> Iterator[Dict[str, torch.Tensor]]:
        bsz = self.args.bsz
        g = torch.Generator().manual_seed(self.args.seed + 1000 * epoch)
        order = torch.randperm(self.n, generator=g).tolist() if self.split == "train" else list(range(self.n))
        for b0 in range(0, self.n, bsz):
            ids = order[b0: b0 + bsz]
            if len(ids) < self.world:
                continue  # a tail smaller than the world would leave a rank without data (and the all-reduce without a peer): dropped on EVERY rank
            lo, hi = self._shard(len(ids), self.rank, self.world)
            mine = ids[lo:hi]
            if self.args.patch_dir:
                yield collate([load_patch(self.files[i], self.args.cdr_mask_key) for i in mine])
            else:
                parts = [self._syn.context_batch(1, self.args.k, seed=self.args.seed + self.offset + i, with_distmat=False) for i in mine]
                for p_ in parts:
                    p_.pop("distmat")  # as in the reference's batches (data.py:93-94): distances are taken from xyz on the device
                yield collate(parts)


def save_checkpoint(path: str, model, optimizer, epoch: int, global_step: int) -> None:
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    torch.save({"state_dict": model.state_dict(), "optimizer": optimizer.state_dict(), "epoch": epoch, "global_step": global_step,
                "hyper_parameters": dict(MODEL_HPARAMS)}, path)


def load_checkpoint(path: str, model, optimizer=None) -> Dict[str, int]:
    ck = torch.load(path, map_location="cpu")
    model.load_state_dict(ck["state_dict"])
    if optimizer is not None and "optimizer" in ck:
        optimizer.load_state_dict(ck["optimizer"])
    return {"epoch": int(ck.get("epoch", 0)), "global_step": int(ck.get("global_step", 0))}


def main(argv: Optional[List[str]] = None) -> int:
    args = parse_argument(argv)
    import torch.distributed as dist

    from . import DiffAb
    from . import distributed as D

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("diffab_pytorch.train needs a gfx950 device: the hot path has no CPU fallback")
    if args.rehearse_on_one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1 and args.rehearse_on_one_gpu:
        dist.init_process_group("gloo")
    elif world > 1:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    torch.manual_seed(args.seed)  # reference: pl.seed_everything(args.seed), train.py:50

    model = DiffAb(*MODEL_HPARAMS.values()).cuda()
    if any(a in ("-l", "--learning-rate") or a.startswith("--learning-rate=") for a in (argv if argv is not None else os.sys.argv[1:])):
        model.lr = args.learning_rate
    optimizer = model.configure_optimizers()
    start = {"epoch": 0, "global_step": 0}
    if args.resume:
        start = load_checkpoint(args.resume, model, optimizer)
    if world > 1:  # identical parameters on every rank
        for p in model.parameters():
            dist.broadcast(p.data, src=0)
    # Timesteps and the forward-noise Philox seeds are drawn from torch's default generator: after the (identical) model
    # construction give every rank its own stream, otherwise rank r's local patch b would get exactly the t, eps, categorical
    # and IGSO3 draws of rank 0's local patch b (B / world distinct draws per global batch).
    torch.manual_seed(args.seed + 7919 * (rank + 1) + 104729 * start["global_step"])

    train_src, val_src = PatchSource(args, rank, world, "train"), PatchSource(args, rank, world, "val")
    log_f = open(args.log, "a") if (args.log and rank == 0) else None

    def emit(rec: Dict) -> None:
        if rank != 0:
            return
        line = json.dumps(rec)
        if log_f:
            log_f.write(line + "\n")
            log_f.flush()
        else:
            print(line, flush=True)

    captured: Dict[str, float] = {}
    model.log_dict = lambda d, **kw: captured.update({k: float(v.detach()) if torch.is_tensor(v) else float(v) for k, v in d.items()})  # the LightningModule hook

    def to_dev(batch):
        return {k: v.cuda(non_blocking=True) for k, v in batch.items()}

    step = start["global_step"]
    for epoch in range(start["epoch"], args.epochs):
        model.train()
        t0 = time.perf_counter()
        for i, batch in enumerate(train_src.batches(epoch)):
            optimizer.zero_grad(set_to_none=True)  # the backward hands out views of its flat buckets as p.grad
            loss = model.training_step(to_dev(batch), i)
            if not bool(torch.isfinite(loss.detach())):
                raise SystemExit(f"rank {rank}: non-finite loss {float(loss)} at epoch {epoch}, batch {i} (captured: {captured}) - aborting")
            loss.backward()
            D.allreduce_gradients(model.parameters(), dist if world > 1 else None, flats=model.gradient_buckets())
            if args.gradient_clip_val:
                torch.nn.utils.clip_grad_norm_(model.parameters(), args.gradient_clip_val)
            optimizer.step()
            step += 1
            emit({"epoch": epoch, "step": step, **captured, "lr": optimizer.param_groups[0]["lr"]})
            if args.max_steps and step >= args.max_steps:
                break
        model.eval()
        sums: Dict[str, float] = {}
        nval = 0
        for i, batch in enumerate(val_src.batches(epoch)):
            model.validation_step(to_dev(batch), i)
            for k, v in captured.items():
                if k.startswith("val/"):
                    sums[k] = sums.get(k, 0.0) + v
            nval += 1
        if nval:
            emit({"epoch": epoch, "step": step, **{k: v / nval for k, v in sums.items()}, "epoch_s": time.perf_counter() - t0})
        if args.ckpt_dir and rank == 0:
            save_checkpoint(os.path.join(args.ckpt_dir, f"epoch={epoch}-step={step}.ckpt"), model, optimizer, epoch + 1, step)
            save_checkpoint(os.path.join(args.ckpt_dir, "last.ckpt"), model, optimizer, epoch + 1, step)
        if args.max_steps and step >= args.max_steps:
            break
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if log_f:
        log_f.close()
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
