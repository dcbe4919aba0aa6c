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


def _eight_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from synchformer_amd.dist import (BucketedAllReduce, all_gather_pair, allreduce_mean_, gather_logits, max_over_ranks, reduce_scatter_pair, scaled_lr,
                                          scaled_warmup, shard_range)
        out = {'rank': rank}
        # (1) clip sharding of a 16-clip and of a ragged 13-clip batch, and the logits gather in shard order
        out['spans'] = [shard_range(n, rank, world) for n in (16, 13)]
        s, e = out['spans'][1]
        full = gather_logits(torch.arange(s, e, dtype=torch.float32).unsqueeze(1).repeat(1, 21), 13)
        out['gathered'] = None if full is None else full[:, 0].tolist()
        out['tmax'] = max_over_ranks(0.25 * (rank + 1))
        # (2) Stage-1: both modalities' (28, 768) embeddings in one all-gather; row (r, i) of modality m carries 1000 m + 100 r + i
        n = 28
        a = (100.0 * rank + torch.arange(n, dtype=torch.float32)).unsqueeze(1).repeat(1, 768)
        b = a + 1000.0
        ga, gb = all_gather_pair(a, b)
        want = (100.0 * torch.arange(world).view(world, 1) + torch.arange(n).view(1, n)).reshape(-1)
        out['gather_ok'] = bool(torch.equal(ga[:, 0], want) and torch.equal(gb[:, 5], want + 1000.0) and ga.shape == (world * n, 768))
        # ... and its backward: every rank holds gradients for ALL gathered rows; rank r receives the sum over ranks of rows [r n, (r + 1) n)
        da = torch.arange(world * n, dtype=torch.float32).unsqueeze(1).repeat(1, 768) * (rank + 1)
        db = -da
        ra, rb = reduce_scatter_pair(da, db, n)
        tot = world * (world + 1) / 2
        mine = torch.arange(rank * n, (rank + 1) * n, dtype=torch.float32) * tot
        out['rs_ok'] = bool(torch.equal(ra[:, 0], mine) and torch.equal(rb[:, 767], -mine) and ra.shape == (n, 768))
        # (3) the gradient buffer in 7 buckets launched out of order (audio tower first, then the visual groups back to front) == ONE mean all-reduce
        g = torch.Generator().manual_seed(100 + rank)
        flat = torch.randint(-64, 64, (100003,), generator=g).float()          # integer-valued: the sums are exact whatever the reduction order
        one = allreduce_mean_(flat.clone())
        cuts = [0, 9001, 20011, 37003, 54321, 70001, 88007, 100003]
        red = BucketedAllReduce(flat)
        for i in (6, 5, 4, 3, 2, 1, 0):
            red.launch(cuts[i], cuts[i + 1])
        out['covered'] = red.covered()
        out['buckets_ok'] = bool(torch.equal(red.finish(), one))
        hole = BucketedAllReduce(torch.zeros(10))
        hole.world = 1
        hole.launch(0, 4); hole.launch(5, 10)
        out['hole_detected'] = not hole.covered()
        # (4) the reference's world-size rules: base learning rate x world (scripts/train_utils.py:218), warm-up / world (training/train_clip.py:312)
        from synchformer_amd.stage1 import cosine_lr
        from synchformer_amd.train import constant_with_warmup_lr
        out['lr'] = (scaled_lr(2e-6, world), scaled_warmup(1000, world), cosine_lr(scaled_warmup(1000, world) - 1, scaled_lr(1e-4, world), scaled_warmup(1000, world), 10000),
                     constant_with_warmup_lr(1000, scaled_lr(2e-6, world)))
        dist.barrier()
        q.put(out)
    finally:
        dist.destroy_process_group()


def test_eight_rank_plumbing():
    """World size 8 on gloo (the size of the node the driver's scaling bench uses; nothing above 2 ranks was exercised before): clip sharding + logits gather + MAX
    time reduce of the inference path, Stage-1's paired embedding all-gather and its reduce-scatter backward, the 7-bucket gradient all-reduce against one mean
    all-reduce, and the learning-rate / warm-up scaling rules."""
    world = 8
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_eight_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=240) for _ in procs), key=lambda o: o['rank'])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for k, n in enumerate((16, 13)):
        spans = [o['spans'][k] for o in res]
        assert spans[0][0] == 0 and spans[-1][1] == n and all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
        sizes = [e - s for s, e in spans]
        assert max(sizes) - min(sizes) <= 1 and sizes == sorted(sizes, reverse=True)
    assert res[0]['gathered'] == [float(i) for i in range(13)] and all(o['gathered'] is None for o in res[1:])
    assert all(o['tmax'] == 2.0 for o in res)
    assert all(o['gather_ok'] and o['rs_ok'] and o['covered'] and o['buckets_ok'] and o['hole_detected'] for o in res), res
    lr, warm, lr_end_warm, lr_const = res[0]['lr']
    assert lr == 1.6e-5 and warm == 125 and abs(lr_end_warm - 8e-4) < 1e-12 and abs(lr_const - 1.6e-5) < 1e-18
