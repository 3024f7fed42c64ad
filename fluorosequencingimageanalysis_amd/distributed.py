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


def world_size():
    import torch.distributed as dist
    return dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1


def get_rank():
    import torch.distributed as dist
    return dist.get_rank() if (dist.is_available() and dist.is_initialized()) else 0


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


def find_peptides_sharded(images, partition="lpt", dst=0, **find_peptides_parameters):
    """pflib.find_peptides over a stack uint16[n, H, W] with the fields sharded over the ranks of the process group
    (one process per GPU).  Every rank calls it with the same stack and parameters.

    partition='lpt' balances the fields by candidate count with the reference's longest-processing-time rule
    (pflib.parallel_image_batch, pflib.py:1043-1069: candidates are counted first, then the fields are dealt out);
    'round_robin' gives field i to rank i mod world.  Each rank runs detect -> fit -> consolidate on its share; the
    only exchange is the gather of the peak records to `dst` (RCCL p2p, gather_tables).  Returns the list of n dicts
    (identical to pflib.find_peptides_batch on one GPU) on `dst` and None on the other ranks."""
    import torch
    import torch.distributed as dist
    from . import _native as N
    from . import engine as E
    from . import pflib
    world, rank = world_size(), get_rank()
    if world == 1:
        return pflib.find_peptides_batch(images, **find_peptides_parameters)
    imgs, fmt = E.as_pixel_fields(images)
    if imgs.ndim != 3:
        raise ValueError("images must have shape (n, H, W)")
    n, H, W = imgs.shape
    fp = dict(find_peptides_parameters)
    radius = fp.get("consolidation_radius", 4)
    if radius < 2:
        raise ValueError("consolidation_radius must be at least 2")
    if fp.get("fit_type", "gauss") != "gauss":
        raise NotImplementedError("fit_type='monte_carlo' is not reproduced (pflib.py:117-177)")
    prm = E.detect_params(fp.get("median_filter_size", 5), fp.get("correlation_matrix", pflib.default_correlation_matrix),
                          fp.get("c_std", 2), fmt)
    dev = torch.device("cuda", torch.cuda.current_device())

    def run_share(idx, fit):
        if not idx:
            return None, None
        eng = E.Engine(len(idx), H, W, device=dev)
        d_img = E.to_device_u16(imgs[idx], dev)
        if fit:
            eng.run(d_img, prm, fp.get("r_2_threshold", 0.7), radius, N.MODE_REF, pflib.PY2_ROUND)
        else:
            eng.detect(d_img, prm)
        return eng, d_img

    if partition == "lpt":
        # count candidates on a round-robin share, exchange the counts, deal the fields out
        mine = shard_fields(n, rank, world)
        eng, _ = run_share(mine, fit=False)
        w = torch.zeros(n, dtype=torch.int64)
        if eng is not None:
            w[mine] = eng.counts[:len(mine)].cpu().long()
        w = w.to(dev) if dist.get_backend() != "gloo" else w
        dist.all_reduce(w)
        parts = _partition([int(x) for x in w.cpu()], world, "lpt")
        del eng
    else:
        parts = _partition(n, world, partition)
    mine = parts[rank]
    eng, d_img = run_share(mine, fit=True)
    if eng is not None:
        rec, offs = eng.peak_records(d_img)
        per_field = (offs[1:] - offs[:-1]).to(torch.int32)
        per_field = torch.where(eng.nkeep[:len(mine)] < 0, eng.nkeep[:len(mine)], per_field).reshape(-1, 1)
    else:
        rec = torch.empty((0, E.PEAK_RECORD_BYTES), dtype=torch.uint8, device=dev)
        per_field = torch.empty((0, 1), dtype=torch.int32, device=dev)
    table, _ = gather_tables(rec, dst)
    fields, _ = gather_tables(per_field.contiguous(), dst)
    if rank != dst:
        return None
    rows, fit, sub = E.split_peak_records(table.cpu().numpy(), fmt)
    nk = fields.cpu().numpy().reshape(-1)
    order = [i for p in parts for i in p]                  # global field index of every gathered per-field entry
    failed = set(int(k) for k in np.nonzero(nk < 0)[0])
    offs = np.concatenate([[0], np.cumsum(np.maximum(nk, 0))])
    dicts = pflib._records_to_dicts(rows, fit, sub, offs, failed)
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
    the file system is the reference's gather too) and the per-rank result dicts are merged on all ranks."""
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
    local = pflib._candidate_counts([paths[i] for i in mine], det)          # unreadable images count as None
    gathered = [None] * world
    dist.all_gather_object(gathered, list(zip(mine, local)))
    counts = dict(kv for part in gathered for kv in part)
    usable = [i for i in range(len(paths)) if counts.get(i) is not None]
    parts = lpt_partition([counts[i] for i in usable], world)
    my_paths = [paths[usable[k]] for k in sorted(parts[rank])]
    res = pflib.image_batch(my_paths, find_peptides_parameters, timestamp_epoch)
    merged = [None] * world
    dist.all_gather_object(merged, res)
    out = {}
    for part in merged:
        for k, v in part.items():
            out.setdefault(k, v)
    return out
