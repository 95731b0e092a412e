"""Multi-GPU sharding of the FM channel batch: one process per GPU, static block partition of the
channel ids, NO collective on the data path (FM channels are independent: SURVEY 8e).  Collectives
(RCCL on GPUs, gloo in the CPU tests) are used only for the fan-out/gather either side of the path and
for the max-over-ranks timing of bench.py."""
import os

import torch
import torch.distributed as dist


def shard_channels(total, world, rank):
    """Static block partition: returns (first_channel, count) of `rank`; the first total % world ranks get one extra."""
    if not (0 <= rank < world) or total < 0:
        raise ValueError("bad shard request")
    base, rem = divmod(total, world)
    count = base + (1 if rank < rem else 0)
    first = rank * base + min(rank, rem)
    return first, count


def init(backend=None):
    """Initialise torch.distributed from the torchrun environment (RANK/WORLD_SIZE/MASTER_*).  Returns (rank, world)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world


def _collectives_on():
    """Collectives run when a process group exists and has more than one rank -- or when FMX_SHARD_FORCE_COLLECTIVES is set: a
    one-rank group still goes through the backend (communicator set-up, the collective's kernels), which is how the RCCL path is
    exercised on a one-GPU box (tests/test_gpu_round3.py)."""
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size() > 1 or bool(os.environ.get("FMX_SHARD_FORCE_COLLECTIVES"))


def max_over_ranks(value, device="cpu"):
    """bench.py's timing rule: the slowest rank defines the step time."""
    if not _collectives_on():
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_pcm(pcm_local, total_channels, dst=0):
    """Gather per-rank PCM [channels_local, frames, 2] onto `dst` as [total_channels, frames, 2]
    (384 kB/s per channel: far below one xGMI link, so a plain gather is the right collective).
    Returns the full tensor on dst, None elsewhere."""
    if not _collectives_on():
        return pcm_local
    world, rank = dist.get_world_size(), dist.get_rank()
    base = (total_channels + world - 1) // world                # pad every shard to the largest one
    frames = pcm_local.shape[1]
    padded = torch.zeros((base, frames, 2), dtype=pcm_local.dtype, device=pcm_local.device)
    padded[: pcm_local.shape[0]] = pcm_local
    bufs = [torch.empty_like(padded) for _ in range(world)] if rank == dst else None
    dist.gather(padded, bufs, dst=dst)
    if rank != dst:
        return None
    parts = []
    for r in range(world):
        _, cnt = shard_channels(total_channels, world, r)
        parts.append(bufs[r][:cnt])
    return torch.cat(parts, dim=0)


def broadcast_stream(iq, src=0):
    """Fan-out of a shared wide-band IQ stream (BASELINE configs[2]): every rank demodulates its own
    carriers out of the same samples.  18.4 MB/s per stream."""
    if _collectives_on():
        dist.broadcast(iq, src=src)
    return iq
