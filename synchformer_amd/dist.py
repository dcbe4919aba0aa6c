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


def scaled_lr(base_lr: float, world: int) -> float:
    """STAGE 2 ONLY: the reference scales the base learning rate by the number of GPUs (scripts/train_utils.py:218 `base_learning_rate * num_gpus`).  Stage 1 hands
    cfg.training.learning_rate to its optimizer and scheduler unscaled (train_clip_src/training/train_clip.py:276,314) and only divides the warm-up (`scaled_warmup`)."""
    return base_lr * max(int(world), 1)


def scaled_warmup(warmup: int, world: int) -> int:
    """... and divides the warm-up length of the Stage-1 schedule by the world size (train_clip_src/training/train_clip.py:312)."""
    return int(warmup // max(int(world), 1))


class BucketedAllReduce:
    """Mean all-reduce of a flat gradient buffer in BUCKETS that leave while the backward is still producing the rest (Stage-1: 7 buckets of the 857 MB
    buffer): `launch(lo, hi)` starts an asynchronous SUM all-reduce of flat[lo:hi] as soon as that range is final (on RCCL it runs on the
    collective's own stream behind an event on the compute stream), `finish()` waits for all of them and divides by the world size - element for element the
    sums of ONE all-reduce over the whole buffer (DDP semantics).  Without a process group both calls do nothing."""

    def __init__(self, flat: torch.Tensor):
        self.flat = flat
        self.world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        self.handles = []
        self.spans = []

    def launch(self, lo: int, hi: int):
        if hi <= lo:
            return
        self.spans.append((lo, hi))
        if self.world > 1:
            self.handles.append(dist.all_reduce(self.flat[lo:hi], op=dist.ReduceOp.SUM, async_op=True))

    def wait(self):
        for h in self.handles:
            h.wait()
        self.handles = []

    def finish(self) -> torch.Tensor:
        self.wait()
        if self.world > 1:
            self.flat.div_(self.world)
        return self.flat

    def covered(self) -> bool:
        """True when the launched buckets tile the buffer exactly once (no gap, no overlap)."""
        pos = 0
        for lo, hi in sorted(self.spans):
            if lo != pos:
                return False
            pos = hi
        return pos == self.flat.numel()


def numa_cpus_of_gpu(index: int):
    """The CPUs of the NUMA node a GPU hangs off (sysfs; None when the platform does not say).  bench.py pins each rank's launcher thread there: eight Python
    launchers on a 256-cpu host otherwise wander across sockets, and every hipLaunchKernel of a rank crosses the fabric to its GPU."""
    try:
        pr = torch.cuda.get_device_properties(index)
        bdf = f'{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0'
        node = int(open(f'/sys/bus/pci/devices/{bdf}/numa_node').read().strip())
        if node < 0:
            return None
        cpus = set()
        for part in open(f'/sys/devices/system/node/node{node}/cpulist').read().strip().split(','):
            a, _, b = part.partition('-')
            cpus.update(range(int(a), int(b or a) + 1))
        return node, sorted(cpus)
    except Exception:                                                       # noqa: BLE001 - advisory only
        return None
