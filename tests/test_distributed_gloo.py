"""Gloo tests (CPU, world sizes 2 and 8) of the batch-shard + all-gather path used for N > 1 GPUs.

The local sampler is a stand-in (a deterministic function of the GLOBAL sample index, as the counter-based
noise of the HIP path is); what is tested is the sharding arithmetic and the single all-gather per call:
an N-rank run must return, on every rank, exactly the 1-rank result."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from moleculediffusiontransformer_amd.distributed import (all_gather_samples, sample_sharded, sample_tokens_sharded,
                                                          shard_bounds)


def _fake_local_sample(seq, first):
    b = seq.shape[0]
    idx = torch.arange(first, first + b, dtype=torch.float32).view(b, 1, 1)
    return seq.sum(dim=1).view(b, 1, 1) + idx * torch.ones(b, 3, 8) + torch.arange(8.0).view(1, 1, 8) * 0.25


def _fake_local_tokens(seq, first):
    """decoded ids (b, 8) in [0, 16): a function of the global sample index only"""
    b = seq.shape[0]
    idx = torch.arange(first, first + b).view(b, 1)
    return (idx * 5 + torch.arange(8).view(1, 8) * 3) % 16


def _worker(rank, world, port, total, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        seq = torch.arange(total * 4, dtype=torch.float32).view(total, 4) * 0.01
        out = sample_sharded(_fake_local_sample, seq)
        lo, hi = shard_bounds(total, world, rank)
        again = all_gather_samples(_fake_local_sample(seq[lo:hi], lo), total)
        tok = sample_tokens_sharded(_fake_local_tokens, seq, vocab=16)     # one byte per id on the wire
        assert tok.dtype == torch.int64
        q.put((rank, out.numpy(), again.numpy(), tok.numpy()))    # by value: a tensor would travel as a shared-memory handle that
                                                      # dies with this process if the parent is slow to open it
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("total", [8, 7])
def test_two_rank_result_equals_single_rank(total):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, total, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    seq = torch.arange(total * 4, dtype=torch.float32).view(total, 4) * 0.01
    want = _fake_local_sample(seq, 0)
    for rank, out, again, tok in results:
        assert torch.equal(torch.from_numpy(out), want) and torch.equal(torch.from_numpy(again), want), rank
        assert torch.equal(torch.from_numpy(tok), _fake_local_tokens(seq, 0)), rank


def test_shard_bounds_cover_the_batch():
    for total in (1, 7, 8, 1024, 65536):
        for world in (1, 2, 3, 8):
            spans = [shard_bounds(total, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


class _FakeModel:
    """What sample_sharded(model=) needs of a QMDiffusion*: the batch-dependent kernel choice and its pin (generative.py)."""
    def __init__(self):
        self.kernel_choice = "auto"

    def pin_kernel_choice(self, batch):
        self.kernel_choice = None if batch is None else ("wide" if batch > 1024 else "narrow")


def _worker8(rank, world, port, total, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        seq = (torch.arange(total * 4, dtype=torch.float32).view(total, 4) % 977) * 0.01
        m = _FakeModel()
        seen = []

        def local(sl, first):
            seen.append((first, sl.shape[0], m.kernel_choice))      # the choice is pinned BEFORE the local sampler runs
            return _fake_local_sample(sl, first)
        out = sample_sharded(local, seq, model=m)
        tok = sample_tokens_sharded(_fake_local_tokens, seq, vocab=16, model=m)
        lo, hi = shard_bounds(total, world, rank)
        ok = torch.equal(out, _fake_local_sample(seq, 0)) and torch.equal(tok, _fake_local_tokens(seq, 0))
        q.put((rank, bool(ok), seen[0], tuple(out.shape), tuple(tok.shape), (lo, hi)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("total", [65536, 1027])
def test_eight_rank_result_equals_single_rank_with_uneven_shards(total):
    """world size 8 with BASELINE configs[3]'s global batch (65,536: 8,192 per rank) and a total that does not
    divide (1,027: three ranks hold 129 rows, five hold 128 -- the padded all_gather_into_tensor + re-slice path) for samples AND
    token ids; every rank gets the 1-rank result and has the same kernel choice pinned from the LARGEST shard."""
    world = 8
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker8, args=(r, world, port, total, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    largest = shard_bounds(total, world, 0)[1]
    want_choice = "wide" if largest > 1024 else "narrow"
    for rank, ok, (first, n, choice), oshape, tshape, (lo, hi) in results:
        assert ok, rank
        assert (first, n) == (lo, hi - lo) and choice == want_choice, (rank, first, n, choice)
        assert oshape == (total, 3, 8) and tshape == (total, 8)
    assert [r[5] for r in results] == [shard_bounds(total, world, r) for r in range(world)]
