"""world_size-2 gloo tests of the utterance-sharding path (CPU; the N>1 bench path on the
GPU uses the same functions over RCCL)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from myrtlespeech_amd import parallel as P
from oracle import ds_oracle as O


def test_shard_bounds_cover_the_batch():
    for batch in (1, 7, 32, 33, 256):
        for world in (1, 2, 3, 8):
            spans = [P.shard_bounds(batch, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == batch
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [e - b for b, e in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        P.shard_bounds(4, 2, 2)
    with pytest.raises(ValueError):
        P.shard_batch(torch.zeros(3, 2), torch.tensor([1, 3, 2]), 2, 0)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


class _StubModel:
    """Stands in for the encoder on CPU: 'logits' are a fixed function of the shard, cut to
    the shard's own longest utterance (so ranks hold different T, as the real encoder would)."""

    def __call__(self, x):
        feats, lens = x
        t_r = int(lens.max())
        return (feats[:, :t_r].transpose(0, 1).contiguous(), lens), None


class _OracleGreedy:
    def __call__(self, logits, lens):
        return O.ctc_greedy_decode(logits.numpy(), lens.numpy(), 4)


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rng = np.random.default_rng(0)
        n, t, v = 7, 20, 5
        lens = torch.tensor(sorted(rng.integers(3, t + 1, size=n).tolist(), reverse=True))
        feats = torch.from_numpy((rng.normal(size=(n, t, v)) * 2).round().astype(np.float32))
        for ni in range(n):
            feats[ni, int(lens[ni]):] = 0
        xs, ls = P.shard_batch(feats, lens, world, rank)
        (logits, out_lens), _ = _StubModel()((xs, ls))
        full, full_lens = P.gather_logits(logits, out_lens)
        want = feats.transpose(0, 1)[: int(lens.max())]
        ok_gather = bool(torch.equal(full, want)) and bool(torch.equal(full_lens, lens))
        a = P.sharded_forward_decode(_StubModel(), _OracleGreedy(), feats, lens, batched_decode=True)
        b = P.sharded_forward_decode(_StubModel(), _OracleGreedy(), feats, lens, batched_decode=False)
        ref = O.ctc_greedy_decode(want.numpy(), lens.numpy(), 4)
        q.put((rank, ok_gather, a == ref, b == ref))
    finally:
        dist.destroy_process_group()


def test_two_rank_gather_and_decode_gloo():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(r[0] for r in results) == [0, 1]
    for _, ok_gather, ok_batched, ok_local in results:
        assert ok_gather and ok_batched and ok_local


def _single_rank_worker(port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["MS_FORCE_COLLECTIVE"] = "1"
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        logits = torch.arange(5 * 3 * 4, dtype=torch.float32).view(5, 3, 4)
        lens = torch.tensor([5, 4, 2])
        full, full_lens = P.gather_logits(logits, lens)
        q.put((bool(torch.equal(full, logits)), bool(torch.equal(full_lens, lens)), full.data_ptr() != logits.data_ptr()))
    finally:
        dist.destroy_process_group()


def test_forced_collective_path_at_world_size_one_gloo():
    """MS_FORCE_COLLECTIVE=1 sends a one-rank group through the all-gather (the switch the one-GPU RCCL test uses)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_single_rank_worker, args=(_free_port(), q))
    p.start()
    same, same_lens, went_through_collective = q.get(timeout=120)
    p.join(timeout=60)
    assert p.exitcode == 0
    assert same and same_lens and went_through_collective
