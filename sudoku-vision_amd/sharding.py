"""Multi-GPU sharding of the hot path: frames are independent, so they are dealt round-robin to ranks
(frame i -> rank i mod world, BASELINE.json configs[3]) and every rank runs the same single-GPU pipeline
on its shard.  No data-path collective and no RCCL dependency: a host-side gloo group is used only
to line ranks up for timing, to take the max of their elapsed times and to gather the 81-byte results."""
import os
import socket
import subprocess
import sys
import time

import torch
import torch.distributed as dist


def env_rank_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def launch_local_ranks(n_ranks, argv, rank0_stdout=None, extra_env=None):
    """Starts `n_ranks` fresh interpreters running `python argv...`, one per GPU of this node, with the torchrun environment
    (RANK, LOCAL_RANK, WORLD_SIZE, MASTER_ADDR=127.0.0.1, a free MASTER_PORT).  For callers that were started without a
    launcher: the caller must not have touched the GPU (it only waits).  Rank 0's stdout goes to `rank0_stdout` (default: this
    process's), the other ranks' stdout is dropped, stderr is shared.  A rank that fails takes the others down with it (they
    would wait at a barrier forever).  Returns the first non-zero exit code, else 0."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n_ranks):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n_ranks), LOCAL_WORLD_SIZE=str(n_ranks),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        env.update(extra_env or {})
        procs.append(subprocess.Popen([sys.executable] + list(argv), env=env, stdout=rank0_stdout if r == 0 else subprocess.DEVNULL))
    rc = 0
    while any(p.poll() is None for p in procs):
        bad = [p.returncode for p in procs if p.poll() not in (None, 0)]
        if bad:
            rc = bad[0]
            for p in procs:
                if p.poll() is None:
                    p.kill()
            break
        time.sleep(0.05)
    for p in procs:
        p.wait()
        rc = rc or p.returncode
    return rc


def init(backend="gloo"):
    """Initialises the default process group from the torchrun environment when WORLD_SIZE > 1.  gloo: the group carries
    a barrier, one float64 MAX and the 81-byte digit gather, all host-sized -- rank start-up does not wait on RCCL."""
    rank, local_rank, world = env_rank_world()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if os.environ["MASTER_ADDR"] in ("127.0.0.1", "localhost"):
            os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")      # one node: the container's hostname may not resolve
        dist.init_process_group(backend)
    return rank, local_rank, world


def shutdown():
    if dist.is_initialized():
        dist.destroy_process_group()


def shard_indices(n_total: int, rank: int, world: int):
    """Indices of the frames rank `rank` owns: i with i % world == rank."""
    return list(range(rank, n_total, world))


def barrier(device=None):
    """Device work of this rank done -> all ranks here -> return.  `device`: the CUDA device to drain first (optional)."""
    if device is not None and device.type == "cuda":
        torch.cuda.synchronize(device)
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()


def max_over_ranks(value: float) -> float:
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return float(value)
    t = torch.tensor([value], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_objects(obj):
    """-> [rank 0's obj, rank 1's, ...] on every rank (a list of one without a group): small host-side records, e.g. which GPU a rank sits on."""
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return [obj]
    out = [None] * dist.get_world_size()
    dist.all_gather_object(out, obj)
    return out


def gather_digits(local_digits: torch.Tensor, n_total: int, rank: int, world: int):
    """Reassembles per-rank digit tensors [n_local,81] into frame order [n_total,81] on every rank (host-sized data; device
    tensors are gathered through the host, the group being gloo)."""
    if not (dist.is_initialized() and world > 1):
        return local_digits
    dev = local_digits.device
    n_max = (n_total + world - 1) // world
    pad = torch.zeros((n_max, 81), dtype=local_digits.dtype)
    pad[:local_digits.shape[0]] = local_digits.cpu()
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad)
    out = torch.empty((n_total, 81), dtype=local_digits.dtype)
    for r in range(world):
        idx = shard_indices(n_total, r, world)
        out[idx] = parts[r][:len(idx)]
    return out.to(dev)
