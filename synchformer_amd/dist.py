"""Multi-GPU plumbing for the inference path: one process per GPU, clips sharded by rank, NO data-path collective.

Offset-prediction inference is embarrassingly parallel over clips ("replicas only"): every rank holds a full
replica of the 475 MB bf16 weights and processes its own contiguous slice of the batch.  The only communication is
control-plane: a barrier + MAX-reduce of the step time for honest throughput accounting, and an optional gather of the
(B, 21) logits to rank 0.  Backend 'nccl' (= RCCL over xGMI on ROCm) on GPUs, 'gloo' in the CPU unit tests.
"""
from typing import List, Optional, Tuple

import torch
import torch.distributed as dist


def shard_range(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous, balanced split of `n_items` clips: the first (n % world) ranks get one extra."""
    if world <= 0 or not 0 <= rank < world:
        raise ValueError(f'bad rank/world {rank}/{world}')
    base, extra = divmod(n_items, world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def max_over_ranks(value: float, device: Optional[torch.device] = None) -> float:
    """MAX-reduce a python float (step time) over the default process group; identity when not initialised."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=device or 'cpu')
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def allreduce_mean_(flat: torch.Tensor) -> torch.Tensor:
    """In-place mean all-reduce of a flat gradient bucket (DDP semantics: sum / world); identity without a process group."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        flat.div_(dist.get_world_size())
    return flat


def gather_logits(local: torch.Tensor, n_total: int) -> Optional[torch.Tensor]:
    """Gather per-rank logits (ragged first dim, shard_range order) to rank 0 -> (n_total, C); None on other ranks."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return local
    world, rank = dist.get_world_size(), dist.get_rank()
    sizes = [shard_range(n_total, r, world) for r in range(world)]
    pad = max(e - s for s, e in sizes)
    buf = torch.zeros(pad, local.shape[1], dtype=local.dtype, device=local.device)
    buf[:local.shape[0]] = local
    out: List[torch.Tensor] = [torch.empty_like(buf) for _ in range(world)] if rank == 0 else None
    dist.gather(buf, out, dst=0)
    if rank != 0:
        return None
    return torch.cat([o[:e - s] for o, (s, e) in zip(out, sizes)], 0)


def all_gather_rows(local: torch.Tensor) -> torch.Tensor:
    """Concatenate equally-shaped per-rank (n, D) embedding blocks in rank order -> (world * n, D): the forward half of
    `torch.cat(torch.distributed.nn.all_gather(x))` in AVCLIP.forward (open_clip/model.py:489-491).  One all-gather of
    n * 768 fp32 per modality per step (RCCL over xGMI on GPUs); identity without a process group."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return local
    out = torch.empty(dist.get_world_size() * local.shape[0], *local.shape[1:], dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, local.contiguous())
    return out


def all_gather_pair(a: torch.Tensor, b: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """Both modalities' (n, D) embeddings in ONE all-gather: the message is latency-bound (2 x 28 x 768 fp32 = 172 KB per rank at the
    configured 2 clips x 14 segments), so two separate collectives cost two latencies for nothing.  Returns the two
    (world * n, D) matrices in rank order, as `torch.cat(torch.distributed.nn.all_gather(x))` would (open_clip/model.py:489-491)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return a, b
    world, n = dist.get_world_size(), a.shape[0]
    assert a.shape == b.shape and a.dtype == b.dtype
    local = torch.cat([a, b], 0).contiguous()                                             # (2 n, D)
    out = torch.empty(world, 2 * n, *a.shape[1:], dtype=a.dtype, device=a.device)
    dist.all_gather_into_tensor(out.view(world * 2 * n, *a.shape[1:]), local)
    return out[:, :n].reshape(world * n, *a.shape[1:]), out[:, n:].reshape(world * n, *a.shape[1:])


def reduce_scatter_pair(da_all: torch.Tensor, db_all: torch.Tensor, n: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """Backward of `all_gather_pair`: every rank holds gradients w.r.t. ALL gathered rows, (world * n, D) per modality; rank r needs the sum
    over ranks of rows [r n, (r + 1) n).  One reduce-scatter of a (world, 2 n, D) buffer on RCCL (each rank receives 2 n rows instead of
    the world * 2 n an all-reduce would deliver); gloo has no reduce-scatter, so the CPU / shared-GPU tests take one all-reduce + slice -
    the same sums."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return da_all, db_all
    world, rank = dist.get_world_size(), dist.get_rank()
    packed = torch.stack([da_all.view(world, n, -1), db_all.view(world, n, -1)], 1).contiguous()        # (world, 2, n, D)
    if dist.get_backend() == 'nccl':
        out = torch.empty_like(packed[0])
        dist.reduce_scatter_tensor(out.view(-1), packed.view(-1), op=dist.ReduceOp.SUM)
    else:
        dist.all_reduce(packed, op=dist.ReduceOp.SUM)
        out = packed[rank]
    return out[0], out[1]
