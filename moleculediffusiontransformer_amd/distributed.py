"""Batch-sharded sampling over the GPUs of one node: one process per GPU, weights replicated, no
collective inside the step loop, ONE all-gather of the generated samples per call (RCCL over xGMI;
backend "nccl" is RCCL on ROCm).  No cross-sample op exists on the path, so sharding
the batch axis is exact: with the counter-based noise keyed by the GLOBAL sample index an N-rank run
returns the same samples as a 1-rank run -- bit for bit as long as both run the same kernels.  The one
batch-dependent kernel choice of the package (the form of the 256-channel transformers,
generative._QMBase._wide) is therefore PINNED for a sharded call: sample_sharded(..., model=m) resolves it
once from the largest shard, on every rank alike, and a later 1-rank run of any sub-batch on the same model
reproduces the rows exactly (m.kernel_choice stays pinned until pin_kernel_choice(None)).
"""
from __future__ import annotations

from typing import Callable, Optional, Tuple

import torch
import torch.distributed as dist

Tensor = torch.Tensor


def shard_bounds(total: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous [start, stop) of `total` items owned by `rank` (first `total % world` ranks get one more)."""
    q, r = divmod(total, world)
    start = rank * q + min(rank, r)
    return start, start + q + (1 if rank < r else 0)


def all_gather_samples(local: Tensor, total: int, group=None, force_collective: bool = False) -> Tensor:
    """All-gather per-rank (b_r, C, L) results into the global (total, C, L) tensor on every rank.  A one-rank group returns
    ``local`` as is unless ``force_collective`` (the RCCL smoke test on a one-GPU box: the same all_gather_into_tensor call)."""
    world = dist.get_world_size(group)
    if world == 1 and not force_collective:
        return local
    rank = dist.get_rank(group)
    sizes = [shard_bounds(total, world, r)[1] - shard_bounds(total, world, r)[0] for r in range(world)]
    assert local.shape[0] == sizes[rank], (local.shape, sizes, rank)
    mx = max(sizes)
    pad = local
    if local.shape[0] < mx:
        pad = torch.cat([local, local.new_zeros((mx - local.shape[0],) + tuple(local.shape[1:]))])
    out = local.new_empty((world * mx,) + tuple(local.shape[1:]))
    dist.all_gather_into_tensor(out, pad.contiguous(), group=group)
    if all(s == mx for s in sizes):
        return out
    return torch.cat([out[r * mx: r * mx + s] for r, s in enumerate(sizes)])


def pin_for_shards(model, total: int, world: int, guided: bool = False) -> Optional[str]:
    """Pin ``model``'s batch-dependent kernel choice from the LARGEST shard of ``total`` samples over ``world`` ranks (the
    same value on every rank), unless it is pinned already.  Returns the choice in force (None without a model)."""
    if model is None:
        return None
    if getattr(model, "kernel_choice", "auto") == "auto":
        lo, hi = shard_bounds(total, world, 0)
        model.pin_kernel_choice((hi - lo) * (2 if guided else 1))
    return model.kernel_choice


def sample_sharded(local_sample: Callable[[Tensor, int], Tensor], sequences: Tensor, group=None, model=None,
                   guided: bool = False) -> Tensor:
    """Every rank passes the same global `sequences` (B, n); rank r generates samples for its contiguous
    slice via ``local_sample(seq_slice, first_global_index)`` and all ranks receive the full result.  ``model``: the
    QMDiffusion* object ``local_sample`` calls, so that its kernel choice is pinned shard-independently (module docstring)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    pin_for_shards(model, sequences.shape[0], world, guided)
    lo, hi = shard_bounds(sequences.shape[0], world, rank)
    local = local_sample(sequences[lo:hi], lo)
    if world == 1:
        return local
    return all_gather_samples(local, sequences.shape[0], group)


def all_gather_tokens(local: Tensor, total: int, vocab: int, group=None, force_collective: bool = False) -> Tensor:
    """All-gather decoded token ids (b_r, L) -> (total, L) int64 on every rank.  Ids below 256 travel as ONE byte each
    (L bytes per molecule instead of the 4 * C * L bytes of the fp32 sample: 64 B instead of 4 KB for BASELINE configs[1])."""
    wire = local.to(torch.uint8 if vocab <= 256 else torch.int32)
    return all_gather_samples(wire, total, group, force_collective).long()


def sample_tokens_sharded(local_sample_tokens: Callable[[Tensor, int], Tensor], sequences: Tensor, vocab: int,
                          group=None, model=None, guided: bool = False) -> Tensor:
    """As sample_sharded, for ``local_sample_tokens(seq_slice, first_global_index) -> (b_r, L)`` token ids."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    pin_for_shards(model, sequences.shape[0], world, guided)
    lo, hi = shard_bounds(sequences.shape[0], world, rank)
    local = local_sample_tokens(sequences[lo:hi], lo)
    if world == 1:
        return local.long()
    return all_gather_tokens(local, sequences.shape[0], vocab, group)
