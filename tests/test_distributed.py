"""CPU tests (gloo, world size 2) of the multi-GPU layer: field sharding and the variable-length gather
of peak tables (fluorosequencingimageanalysis_amd/distributed.py).  Same code path as RCCL on the GPUs."""
import os
import socket
import sys

import numpy as np
import pytest

from _util import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import torch
    import torch.distributed as dist
    from fluorosequencingimageanalysis_amd import distributed as D
    r, w, l = D.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    # every rank owns the fields i with i % world == rank and makes a fake 128-byte-row table per field
    fields = D.shard_fields(10, rank, world)
    rows = []
    for f in fields:
        n = 3 + f          # ragged: a different row count per field
        t = np.zeros((n, 128), np.uint8)
        t[:, 0] = f
        t[:, 1] = np.arange(n)
        rows.append(t)
    local = torch.from_numpy(np.concatenate(rows) if rows else np.zeros((0, 128), np.uint8))
    table, counts = D.gather_tables(local, dst=0)
    if rank == 0:
        q.put((table.numpy().copy(), counts))
    else:
        assert table is None
    # an empty contribution also works
    t2, c2 = D.gather_tables(local[:0] if rank == 1 else local, dst=0)
    if rank == 0:
        q.put(c2)
    dist.barrier()
    dist.destroy_process_group()


def test_gather_tables_world2():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    table, counts = q.get(timeout=120)
    c2 = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    exp_counts = [sum(3 + f for f in range(0, 10, 2)), sum(3 + f for f in range(1, 10, 2))]
    assert counts == exp_counts and c2 == [exp_counts[0], 0]
    assert table.shape == (sum(exp_counts), 128)
    # rank-major, per-rank order preserved
    fields_seen = table[:, 0].tolist()
    assert fields_seen[:exp_counts[0]] == [f for f in range(0, 10, 2) for _ in range(3 + f)]
    assert fields_seen[exp_counts[0]:] == [f for f in range(1, 10, 2) for _ in range(3 + f)]


def test_partitions():
    sys.path.insert(0, ROOT)
    from fluorosequencingimageanalysis_amd import distributed as D
    assert D.shard_fields(7, 1, 3) == [1, 4]
    parts = D.lpt_partition([5, 9, 1, 7, 3, 3], 2)       # pflib.py:1056-1069: descending, emptiest partition first
    assert sorted(sum(parts, [])) == list(range(6))
    loads = [sum([5, 9, 1, 7, 3, 3][i] for i in p) for p in parts]
    assert abs(loads[0] - loads[1]) <= 2
    assert parts[0][0] == 1 and parts[1][0] == 3


def test_gather_single_process():
    import torch
    from fluorosequencingimageanalysis_amd import distributed as D
    t = torch.zeros((5, 128), dtype=torch.uint8)
    out, counts = D.gather_tables(t)
    assert out is t and counts == [5]
