"""Multi-GPU sharding of the hot path: frames are independent, so they are dealt round-robin to ranks
(frame i -> rank i mod world, BASELINE.json configs[3]) and every rank runs the same single-GPU pipeline
on its shard.  No data-path collective: torch.distributed (RCCL on GPUs, gloo on CPU tests) is used only
to line ranks up for timing, to take the max of their elapsed times and to gather the 81-byte results."""
import os

import torch
import torch.distributed as dist


def env_rank_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def init(backend=None, device=None):
    """Initialises the default process group from the torchrun environment when WORLD_SIZE > 1."""
    rank, local_rank, world = env_rank_world()
    if world > 1 and not dist.is_initialized():
        backend = backend or ("nccl" if torch.cuda.is_available() else "gloo")
        kwargs = {}
        if backend == "nccl":
            kwargs["device_id"] = torch.device("cuda", local_rank) if device is None else device
        dist.init_process_group(backend, **kwargs)
    return rank, local_rank, world


def shard_indices(n_total: int, rank: int, world: int):
    """Indices of the frames rank `rank` owns: i with i % world == rank."""
    return list(range(rank, n_total, world))


def barrier(device=None):
    if device is not None and device.type == "cuda":
        torch.cuda.synchronize(device)
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()
    if device is not None and device.type == "cuda":
        torch.cuda.synchronize(device)


def max_over_ranks(value: float, device=None) -> float:
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_digits(local_digits: torch.Tensor, n_total: int, rank: int, world: int):
    """Reassembles per-rank digit tensors [n_local,81] into frame order [n_total,81] on every rank (host-sized data)."""
    if not (dist.is_initialized() and world > 1):
        return local_digits
    n_max = (n_total + world - 1) // world
    pad = torch.zeros((n_max, 81), dtype=local_digits.dtype, device=local_digits.device)
    pad[:local_digits.shape[0]] = local_digits
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad)
    out = torch.empty((n_total, 81), dtype=local_digits.dtype, device=local_digits.device)
    for r in range(world):
        idx = shard_indices(n_total, r, world)
        out[idx] = parts[r][:len(idx)]
    return out
