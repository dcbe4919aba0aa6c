"""CPU, world_size 2, gloo: the N>1 plumbing of the inference path (clip sharding, MAX time reduce, logits gather)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_clips, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from synchformer_amd.dist import allreduce_mean_, gather_logits, max_over_ranks, shard_range
        s, e = shard_range(n_clips, rank, world)
        # stand-in for the per-rank forward: logits row i = clip index i (so the gather order is checkable)
        local = torch.arange(s, e, dtype=torch.float32).unsqueeze(1).repeat(1, 21)
        t = max_over_ranks(0.5 + rank)
        flat = allreduce_mean_(torch.full((1000,), float(rank + 1)))          # DDP gradient averaging on the flat bucket
        assert torch.all(flat == 1.5)
        full = gather_logits(local, n_clips)
        dist.barrier()
        q.put((rank, (s, e), t, None if full is None else full[:, 0].tolist()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('n_clips', [5, 8])
def test_two_rank_sharding_and_gather(n_clips):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_clips, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, span0, t0, full0), (r1, span1, t1, full1) = res
    assert span0[0] == 0 and span0[1] == span1[0] and span1[1] == n_clips
    assert t0 == t1 == 1.5                                   # MAX over ranks
    assert full0 == [float(i) for i in range(n_clips)] and full1 is None


def _gather_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from synchformer_amd.dist import all_gather_rows
        local = torch.full((3, 768), float(rank)) + torch.arange(3.).unsqueeze(1)      # row i of rank r = r + i
        full = all_gather_rows(local)
        # both modalities in one message, and the matching backward (sum over ranks of the gradient rows this rank owns)
        from synchformer_amd.dist import all_gather_pair, reduce_scatter_pair
        va, aa = all_gather_pair(local, local + 100.0)
        ok_pair = torch.equal(va, full) and torch.equal(aa, full + 100.0)
        gv = torch.arange(6 * 768, dtype=torch.float32).view(6, 768) * (rank + 1)        # this rank's gradient w.r.t. ALL gathered rows
        ga = -gv
        dv, da = reduce_scatter_pair(gv, ga, 3)
        want = torch.arange(6 * 768, dtype=torch.float32).view(6, 768)[rank * 3:(rank + 1) * 3] * 3.0       # (1 + 2) x the rows of this rank
        ok_rs = torch.equal(dv, want) and torch.equal(da, -want)
        q.put((rank, tuple(full.shape), full[:, 0].tolist(), ok_pair, ok_rs))
    finally:
        dist.destroy_process_group()


def test_two_rank_embedding_all_gather():
    """AVCLIP gather_for_loss (open_clip/model.py:489-491): rank-ordered concatenation of the per-rank embedding blocks."""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gather_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for _, shape, col, ok_pair, ok_rs in res:
        assert shape == (6, 768) and col == [0., 1., 2., 1., 2., 3.]
        assert ok_pair and ok_rs
