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


# ---- the parallel file path: pflib.parallel_image_batch with ranks as workers (pflib.py:1043-1111) -------------------------
def _fake_gpu(pflib, fail_rank=None, rank=0):
    """Stand-ins for the two GPU entry points the file layer uses (as tests/test_batch_io.py does): every image 'finds' the
    golden PSFs; its candidate count is its top-left pixel."""
    import numpy as np
    from test_batch_io import golden_psfs, records_of
    from fluorosequencingimageanalysis_amd import _native
    psfs, _ = golden_psfs()
    one = records_of(psfs)

    def fake_fit(images, **kw):                 # (image_batch's GPU call: byte tables, find_peptides_records)
        if fail_rank == rank:
            raise RuntimeError("device lost")
        return np.concatenate([one for _ in images]), np.full(len(images), len(one), np.int32), _native.PIXELS_U16

    def fake_counts(paths, detect_parameters=None):
        out = []
        for p in paths:
            try:
                out.append(int(pflib.read_image(p)[1][0, 0]))
            except Exception:       # noqa: BLE001
                out.append(None)
        return out
    pflib.find_peptides_records = fake_fit
    pflib._candidate_counts = fake_counts


def _tiffs(d):
    """8 ragged TIFFs (two shapes, 'candidate counts' 5..90 in the top-left pixel) + one unreadable file."""
    from PIL import Image
    import numpy as np
    os.makedirs(d, exist_ok=True)
    paths = []
    for i, cnt in enumerate((50, 5, 90, 20, 35, 60, 10, 75)):
        a = np.full((64, 80) if i % 3 == 0 else (96, 96), 100, np.uint16)
        a[0, 0] = cnt
        p = os.path.join(d, "im%d.tif" % i)
        Image.fromarray(a).save(p, format="TIFF")
        paths.append(p)
    bad = os.path.join(d, "corrupt.tif")
    open(bad, "wb").write(b"not an image")
    return paths[:4] + [bad] + paths[4:]


def _file_worker(rank, world, port, q, d, fail_rank):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    from fluorosequencingimageanalysis_amd import distributed as D, pflib
    D.init_from_env(backend="gloo")
    _fake_gpu(pflib, fail_rank, rank)
    paths = _tiffs(d) if rank == 0 else None
    box = [paths]
    dist.broadcast_object_list(box, src=0)
    try:
        res = pflib.parallel_image_batch(box[0], {"c_std": 3}, 1450000000.4, num_processes=2)
        q.put((rank, "ok", res))
    except Exception as e:      # noqa: BLE001
        q.put((rank, "error", "%s: %s" % (type(e).__name__, e)))
    dist.barrier()
    dist.destroy_process_group()


def _run_file_ranks(tmp_path, fail_rank=None):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_file_worker, args=(r, 2, port, q, str(tmp_path / "imgs"), fail_rank)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict((r, (s, v)) for r, s, v in (q.get(timeout=180) for _ in range(2)))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    return got


def test_parallel_image_batch_world2(tmp_path):
    """Two ranks (gloo) share a list of ragged TIFFs: candidates counted on a round-robin share, the images dealt out by the
    longest-processing-time rule, each rank fits and saves its own, the result dicts merged on every rank - equal to what one
    rank returns, with every image's files on disk and the unreadable file logged and skipped."""
    sys.path.insert(0, ROOT)
    from fluorosequencingimageanalysis_amd import distributed as D, pflib
    got = _run_file_ranks(tmp_path)
    assert got[0][0] == got[1][0] == "ok" and got[0][1] == got[1][1]
    res = got[0][1]
    paths = _tiffs(str(tmp_path / "imgs"))
    good = [p for p in paths if "corrupt" not in p]
    assert list(res) == good                                    # original paths, list order, the corrupt file missing
    for p in good:
        conv, pkl, tab, png = res[p]
        assert conv == p + ".png" and pkl == conv + "_psfs_nzaj5s.pkl" and tab == conv + "_psfs_nzaj5s.csv" and png == conv + "_psfs_nzaj5s.png"
        assert os.path.exists(pkl) and os.path.exists(tab) and os.path.exists(png)
    # the assignment the ranks used: LPT over the counts (top-left pixels), balanced within the largest weight
    counts = [None if "corrupt" in p else int(os.path.basename(p)[2]) for p in paths]
    weights = {p: (50, 5, 90, 20, 35, 60, 10, 75)[int(os.path.basename(p)[2])] for p in good}
    assign = D.lpt_assignment(paths, [weights.get(p) for p in paths], 2)
    loads = [sum(weights[p] for p in good if assign[p] == r) for r in range(2)]
    assert sorted(assign) == sorted(good) and abs(loads[0] - loads[1]) <= 90 and min(loads) > 0
    # single rank, same stand-ins: the same dict
    _fake_gpu(pflib)
    try:
        one = pflib.image_batch(paths, {"c_std": 3}, 1450000000.4)
    finally:
        import importlib
        importlib.reload(pflib)
    assert one == res


def test_parallel_image_batch_world2_failing_rank(tmp_path):
    """A rank whose GPU work fails outright (not an image that fails) makes EVERY rank raise instead of leaving its peer in a
    collective: per-image failures are logged and skipped inside image_batch, so the stand-in fails in the counting pass."""
    sys.path.insert(0, ROOT)

    got = _run_file_ranks_counting_failure(tmp_path)
    assert got[0][0] == got[1][0] == "error"
    assert "device lost" in got[1][1] and "rank 1 failed" in got[0][1]


def _file_worker_countfail(rank, world, port, q, d):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    from fluorosequencingimageanalysis_amd import distributed as D, pflib
    D.init_from_env(backend="gloo")
    _fake_gpu(pflib)
    if rank == 1:
        def broken(paths, detect_parameters=None):
            raise RuntimeError("device lost")
        pflib._candidate_counts = broken
    paths = _tiffs(d) if rank == 0 else None
    box = [paths]
    dist.broadcast_object_list(box, src=0)
    try:
        res = pflib.parallel_image_batch(box[0], None, 36)
        q.put((rank, "ok", res))
    except Exception as e:      # noqa: BLE001
        q.put((rank, "error", "%s: %s" % (type(e).__name__, e)))
    dist.barrier()
    dist.destroy_process_group()


def _run_file_ranks_counting_failure(tmp_path):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_file_worker_countfail, args=(r, 2, port, q, str(tmp_path / "imgs"))) for r in range(2)]
    for p in procs:
        p.start()
    got = dict((r, (s, v)) for r, s, v in (q.get(timeout=180) for _ in range(2)))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    return got
