"""Multi-GPU driver: fields shard embarrassingly across ranks (one process per GPU); the only exchange
is the final variable-length gather of the peak tables to rank 0 (RCCL over xGMI when the backend is
"nccl"; the same code runs on "gloo" for the CPU tests).

The reference's counterpart is pflib.parallel_image_batch (pflib.py:1000-1111): a multiprocessing.Pool
over image partitions whose "gather" is the file system."""
import os

import numpy as np


def init_from_env(backend=None):
    """torch.distributed init from RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (torchrun). Returns (rank, world, local)."""
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:       # FSQ_DIST_BACKEND=gloo: rehearse the multi-rank path on a box with fewer GPUs than ranks
            backend = os.environ.get("FSQ_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def shard_fields(n_fields, rank, world):
    """Field indices of `rank` under the static round-robin partition (field i -> rank i mod world)."""
    return list(range(rank, n_fields, world))


def lpt_partition(weights, n_parts):
    """Longest-processing-time-first partition, the balancing rule of pflib.parallel_image_batch
    (pflib.py:1056-1069): sort by descending candidate count, give each item to the emptiest partition."""
    order = sorted(range(len(weights)), key=lambda i: -weights[i])
    parts = [[] for _ in range(n_parts)]
    load = [0] * n_parts
    for i in order:
        k = min(range(n_parts), key=lambda p: load[p])
        parts[k].append(i)
        load[k] += weights[i]
    return parts


def gather_tables(local_rows, dst=0):
    """Variable-length gather of row tables (2-D uint8/any dtype tensors, same trailing shape) to `dst`.

    all_gather of the per-rank row counts, then direct point-to-point receives into slices of one
    buffer on dst (no ring: each rank's table crosses exactly one xGMI link).  Returns (table, counts)
    on dst and (None, counts) elsewhere.  With world size 1 it returns its input."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return local_rows, [int(local_rows.shape[0])]
    world, rank = dist.get_world_size(), dist.get_rank()
    n = torch.tensor([local_rows.shape[0]], dtype=torch.int64, device=local_rows.device)
    counts = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(counts, n)
    counts = [int(c.item()) for c in counts]
    if rank == dst:
        out = torch.empty((sum(counts),) + tuple(local_rows.shape[1:]), dtype=local_rows.dtype, device=local_rows.device)
        offs = np.concatenate([[0], np.cumsum(counts)])
        out[offs[dst]:offs[dst + 1]].copy_(local_rows)
        ops = [dist.P2POp(dist.irecv, out[offs[r]:offs[r + 1]], r) for r in range(world) if r != dst and counts[r] > 0]
        if ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()
        return out, counts
    if counts[rank] > 0:
        for w in dist.batch_isend_irecv([dist.P2POp(dist.isend, local_rows.contiguous(), dst)]):
            w.wait()
    return None, counts
