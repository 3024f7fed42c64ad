"""Multi-GPU driver: fields shard embarrassingly across ranks (one process per GPU); the only exchange
is the final variable-length gather of the peak tables to rank 0 (RCCL over xGMI when the backend is
"nccl"; the same code runs on "gloo" for the CPU tests).

The reference's counterpart is pflib.parallel_image_batch (pflib.py:1000-1111): a multiprocessing.Pool
over image partitions whose "gather" is the file system."""
import os
import sys

import numpy as np


def init_from_env(backend=None):
    """torch.distributed init from RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (torchrun). Returns (rank, world, local)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world <= 1:
        return rank, world, local       # (a single process has no group to join, and does not wait for `import torch` here)
    import torch
    import torch.distributed as dist
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


def world_size():
    dist = sys.modules.get("torch.distributed")     # never imported: no process group can exist
    try:
        return dist.get_world_size() if (dist is not None and dist.is_available() and dist.is_initialized()) else 1
    except AttributeError:                          # (another thread is importing it right now: no group yet)
        return 1


def get_rank():
    dist = sys.modules.get("torch.distributed")
    try:
        return dist.get_rank() if (dist is not None and dist.is_available() and dist.is_initialized()) else 0
    except AttributeError:
        return 0


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


def gather_tables(local_rows, dst=0, _force=False):
    """Variable-length gather of row tables (2-D uint8/any dtype tensors, same trailing shape) to `dst`.

    all_gather of the per-rank row counts, then direct point-to-point receives into slices of one
    buffer on dst (no ring: each rank's table crosses exactly one xGMI link).  Returns (table, counts)
    on dst and (None, counts) elsewhere.  With world size 1 it returns its input."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size() == 1 and not _force):
        return local_rows, [int(local_rows.shape[0])]
    world, rank = dist.get_world_size(), dist.get_rank()
    if dist.get_backend() == "gloo" and local_rows.is_cuda:      # (CPU rehearsal of the N > 1 path: gloo moves host memory)
        local_rows = local_rows.cpu()
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


def _partition(weights_or_n, world, partition):
    if partition == "round_robin":
        return [shard_fields(weights_or_n, r, world) for r in range(world)]
    if partition == "lpt":
        return [sorted(p) for p in lpt_partition(weights_or_n, world)]
    raise ValueError("partition must be 'lpt' or 'round_robin'")


def _all_ranks_ok(local_error):
    """Collective error check: every rank reports whether its part of the work raised; if any did, EVERY rank raises (the
    failing one its own exception, the others a RuntimeError naming it) instead of leaving its peers waiting in the next
    collective until the backend times out."""
    import torch.distributed as dist
    reports = [None] * dist.get_world_size()
    dist.all_gather_object(reports, None if local_error is None else "%s: %s" % (type(local_error).__name__, local_error))
    if local_error is not None:
        raise local_error
    bad = [(r, m) for r, m in enumerate(reports) if m is not None]
    if bad:
        raise RuntimeError("rank %d failed: %s" % bad[0])


def find_peptides_sharded(images, partition="lpt", dst=0, n_fields=None, output="dicts", **find_peptides_parameters):
    """pflib.find_peptides over n same-shaped fields sharded over the ranks of the process group (one process per GPU).
    Every rank calls it with the same arguments.

    images: a stack uint16[n, H, W] every rank holds, or a LOADER - a callable taking a list of field indices and returning
    their stack - together with n_fields, so that a rank only ever holds the fields it works on.
    partition='lpt' balances the fields by candidate count with the reference's longest-processing-time rule
    (pflib.parallel_image_batch, pflib.py:1043-1069: candidates are counted first - every rank counts a round-robin share in
    one detection pass, 2 % of the work of a fit pass - then the fields are dealt out); 'round_robin' gives field i to rank
    i mod world.  Each rank streams its share through the same pipeline as pflib.find_peptides_batch; its peak records stay
    in HBM from the kernels that write them to the send (no host copy in between when the backend is RCCL).

    output - what comes back, and what it costs (the reference's pool returns nothing at all: its workers write files,
    pflib.py:1082-1106):
      'records'  on `dst`: (records uint8[k, engine.PEAK_RECORD_BYTES] of ALL fields in field order, int32[n] peaks per field
                 (-1: the re-key assertion of pflib.py:518 fired), pixel format) - pflib.find_peptides_records' form, None on
                 the other ranks.  The only exchange is the point-to-point gather of the records (gather_tables); no Python
                 object per peak exists anywhere, so this form scales with the ranks.
      'local'    on EVERY rank: {global field index: dict} for the fields the rank fitted - the dicts are built where the
                 fields were fitted (every rank's interpreter works on its own share) and nothing travels.
      'dicts'    on `dst`: the list of n dicts pflib.find_peptides_batch returns on one GPU (built by dst's interpreter from the
                 gathered records: about 4 500 fields/s whatever the number of ranks - the compatibility form).
    A rank that fails makes every rank raise."""
    import torch
    import torch.distributed as dist
    from . import engine as E
    from . import pflib
    if output not in ("dicts", "records", "local"):
        raise ValueError("output must be 'dicts', 'records' or 'local'")
    world, rank = world_size(), get_rank()
    if callable(images):
        if n_fields is None:
            raise ValueError("a loader needs n_fields")
        loader, n = images, int(n_fields)
    else:
        stack = np.asarray(images)
        if stack.ndim != 3:
            raise ValueError("images must have shape (n, H, W)")
        loader, n = (lambda idx: stack[idx]), len(stack)
    if world == 1 and not find_peptides_parameters.get("_force_collectives"):
        everything = loader(list(range(n)))
        if output == "records":
            return pflib.find_peptides_records(everything, **find_peptides_parameters)
        dicts = pflib.find_peptides_batch(everything, **find_peptides_parameters)
        return dict(enumerate(dicts)) if output == "local" else dicts
    fp = dict(find_peptides_parameters)
    fp.pop("_force_collectives", None)          # (tests: run the exchange code with a process group of one rank)
    if fp.get("consolidation_radius", 4) < 2:
        raise ValueError("consolidation_radius must be at least 2")
    if fp.get("fit_type", "gauss") != "gauss":
        raise NotImplementedError("fit_type='monte_carlo' is not reproduced (pflib.py:117-177)")
    dev = torch.device("cuda", torch.cuda.current_device())
    gloo = dist.get_backend() == "gloo"
    held = {}                                   # fields this rank has loaded: index -> array

    def load(idx):
        need = [i for i in idx if i not in held]
        if need:
            for i, a in zip(need, loader(need)):
                held[i] = a
        return np.stack([held[i] for i in idx]) if idx else None

    err, parts, rec, counts, fmt, local = None, None, None, None, 0, None
    try:
        if partition == "lpt":
            mine = shard_fields(n, rank, world)
            w = torch.zeros(n, dtype=torch.int64)
            if mine:
                w[mine] = torch.from_numpy(pflib.count_candidates(load(mine), **fp))
    except Exception as e:      # noqa: BLE001 - reported to every rank below
        err = e
    _all_ranks_ok(err)
    if partition == "lpt":
        w = w if gloo else w.to(dev)
        dist.all_reduce(w)
        parts = _partition([int(x) for x in w.cpu()], world, "lpt")
    else:
        parts = _partition(n, world, partition)
    mine = parts[rank]
    # one record format for the whole job: a share with a pixel value beyond 16 bits makes every rank work uint32 words
    # (428-byte records), so that the gathered table has one row width
    share, wide_local = None, 0
    try:
        for i in [i for i in held if i not in mine]:
            del held[i]
        if mine:
            share = load(mine)
            wide_local = int(share.dtype == np.uint32 or (share.dtype != np.float16 and share.size > 0 and float(share.max()) > 65535))
        held.clear()
    except Exception as e:      # noqa: BLE001
        err = e
    _all_ranks_ok(err)
    t_wide = torch.tensor([wide_local], dtype=torch.int32)
    t_wide = t_wide if gloo else t_wide.to(dev)
    dist.all_reduce(t_wide, op=dist.ReduceOp.MAX)
    wide = bool(int(t_wide.item()))
    try:
        if output == "local":
            local = dict(zip(mine, pflib.find_peptides_batch(share, errors='return', **fp))) if mine else {}
        elif mine:
            rec, counts, fmt = pflib.find_peptides_records(share, device=True, wide=wide, **fp)
        else:
            rec = torch.zeros((0, E.PEAK_RECORD_BYTES_U32 if wide else E.PEAK_RECORD_BYTES), dtype=torch.uint8, device=dev)
            counts = np.zeros(0, np.int32)
        share = None
    except Exception as e:      # noqa: BLE001
        err = e
    _all_ranks_ok(err)
    if output == "local":
        for d in local.values():
            if isinstance(d, Exception):
                raise d
        return local
    # the exchange: the records as they sit in HBM (gloo, the CPU rehearsal, moves host memory: gather_tables copies them)
    t_cnt = torch.from_numpy(np.ascontiguousarray(counts, dtype=np.int32).reshape(-1, 1))
    if not gloo:
        t_cnt = t_cnt.to(dev)
    table, _ = gather_tables(rec, dst, _force=True)
    fields, _ = gather_tables(t_cnt, dst, _force=True)
    fmts = [None] * world
    dist.all_gather_object(fmts, int(fmt) if mine else None)
    if rank != dst:
        return None
    fmt = next((f for f in fmts if f is not None), 0)
    order = [i for p in parts for i in p]                  # global field index of every gathered per-field entry
    by_rank = fields.cpu().numpy().reshape(-1).astype(np.int32)
    if output == "records":
        # gathered order (rank by rank) -> global field order: one gather of record rows on the device
        offs = np.concatenate([[0], np.cumsum(np.maximum(by_rank, 0))])
        where = np.empty(n, np.int64)
        where[order] = np.arange(n)
        idx = (np.concatenate([np.arange(offs[where[i]], offs[where[i] + 1]) for i in range(n)])
               if n else np.zeros(0, np.int64))
        same = bool(len(idx) == 0 or (np.diff(idx) == 1).all())
        if not same:
            table = table[torch.from_numpy(idx).to(table.device)]
        return table.cpu().numpy(), by_rank[where], fmt
    dicts = pflib.records_to_dicts(table.cpu().numpy(), by_rank, fmt)
    out = [None] * n
    for k, i in enumerate(order):
        out[i] = dicts[k]
    for d in out:
        if isinstance(d, Exception):
            raise d
    return out


def image_batch_sharded(image_paths, find_peptides_parameters=None, timestamp_epoch=None):
    """pflib.parallel_image_batch with the ranks of the process group as its workers (pflib.py:1000-1111): the images
    are read and their candidates counted (each rank takes a round-robin share of that), dealt out by the reference's
    longest-processing-time rule, every rank runs pflib.image_batch on its share (writing its own pickle / CSV files:
    the file system is the reference's gather too) and the per-rank result dicts are merged on all ranks.
    Keys of the result are the ORIGINAL image paths, as the reference's docstring and its single-process branch have them
    (pflib.py:1013-1016, 949-996); its multi-process branch keys by the converted .png paths instead (pflib.py:1046-1054
    hands image_batch the converted paths), an inconsistency of the reference that is not reproduced.
    A rank that fails (as opposed to an image that fails, which is logged and skipped) makes every rank raise."""
    import torch.distributed as dist
    from . import pflib
    world, rank = world_size(), get_rank()
    if timestamp_epoch is None:             # one timestamp for the whole job, as the reference's parent process takes
        box = [pflib._py2_round(__import__("time").time())]
        if world > 1:
            dist.broadcast_object_list(box, src=0)
        timestamp_epoch = box[0]
    paths = list(dict.fromkeys(os.path.abspath(p) for p in image_paths))
    if world == 1:
        return pflib.image_batch(paths, find_peptides_parameters, timestamp_epoch)
    fp = dict(find_peptides_parameters or {})
    det = {k: fp[k] for k in ("median_filter_size", "correlation_matrix", "c_std") if k in fp}
    mine = shard_fields(len(paths), rank, world)
    err, local = None, []
    try:
        local = pflib._candidate_counts([paths[i] for i in mine], det)      # unreadable images count as None
    except Exception as e:      # noqa: BLE001 - reported to every rank below
        err = e
    _all_ranks_ok(err)
    gathered = [None] * world
    dist.all_gather_object(gathered, list(zip(mine, local)))
    counts = dict(kv for part in gathered for kv in part)
    usable = [i for i in range(len(paths)) if counts.get(i) is not None]
    parts = lpt_partition([counts[i] for i in usable], world)
    my_paths = [paths[usable[k]] for k in sorted(parts[rank])]
    res = {}
    try:
        res = pflib.image_batch(my_paths, find_peptides_parameters, timestamp_epoch)
    except Exception as e:      # noqa: BLE001
        err = e
    _all_ranks_ok(err)
    merged = [None] * world
    dist.all_gather_object(merged, res)
    out = {}
    for part in merged:
        for k, v in part.items():
            out.setdefault(k, v)
    return {p: out[p] for p in paths if p in out}


def lpt_assignment(image_paths, counts, world):
    """Which rank image_batch_sharded gives every usable image of a list to (for tests / logs): {path: rank}."""
    paths = list(dict.fromkeys(os.path.abspath(p) for p in image_paths))
    usable = [i for i in range(len(paths)) if counts[i] is not None]
    parts = lpt_partition([counts[i] for i in usable], world)
    return {paths[usable[k]]: r for r in range(world) for k in parts[r]}
