"""GPU tests (-m gpu) of the multi-GPU product path, rehearsed with two ranks on the ONE GPU of the test box
(gloo for the exchange; on a multi-GPU node the same code runs over RCCL): fields sharded over the ranks must gather
to the table one rank computes ("same fields sharded 1/2/4/8 ways must gather to the identical table").
The ranks are fresh child processes (spawn), never a re-exec of the pytest process."""
import os
import socket
import sys

import numpy as np
import pytest

from _util import ROOT, bits_equal

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _fields():
    from fluorosequencingimageanalysis_amd import synth
    # ragged candidate counts so that the LPT partition differs from round robin
    return np.stack([synth.make_field(300 + i, (128, 128), (5, 60, 10, 40, 25, 8, 50, 30, 15, 35)[i]) for i in range(10)])


def _fields_wide():
    """The same fields as uint32 with ONE of them scaled beyond 16 bits: whichever rank gets it, both must ship 428-byte records."""
    a = _fields().astype(np.uint32)
    a[3] *= 100
    return a


def _worker(rank, world, port, q, partition):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch
    import torch.distributed as dist
    from fluorosequencingimageanalysis_amd import distributed as D
    torch.cuda.set_device(0)
    D.init_from_env(backend="gloo")
    if partition == "records_wide":
        q.put((rank, D.find_peptides_sharded(_fields_wide(), partition="lpt", output="records", c_std=2)))
        dist.barrier()
        dist.destroy_process_group()
        return
    if partition in ("records", "local"):      # the forms that scale: byte tables on rank 0 / dicts where the fields were fitted
        from fluorosequencingimageanalysis_amd import pflib
        built = []
        real = pflib._records_to_dicts
        pflib._records_to_dicts = lambda *a, **k: built.append(1) or real(*a, **k)
        out = D.find_peptides_sharded(_fields(), partition="lpt", output=partition, c_std=2)
        if partition == "records":
            assert not built, "output='records' must not create a Python object per peak on any rank"
            q.put((rank, out))
        else:
            q.put((rank, out))
        dist.barrier()
        dist.destroy_process_group()
        return
    if partition == "loader":                  # a per-rank loader: the rank only ever holds the fields it asks for
        stack, asked = _fields(), set()

        def loader(idx):
            asked.update(idx)
            return stack[list(idx)]
        out = D.find_peptides_sharded(loader, n_fields=len(stack), partition="lpt", c_std=2)
        assert 0 < len(asked) < len(stack), sorted(asked)      # its counting share + its LPT share, never the whole list
    else:
        out = D.find_peptides_sharded(_fields(), partition=partition, c_std=2)
    if rank == 0:
        q.put(out)
    else:
        assert out is None
    dist.barrier()
    dist.destroy_process_group()


def _same_dicts(a, b):
    assert len(a) == len(b)
    for da, db in zip(a, b):
        assert list(da.keys()) == list(db.keys())
        for k in da:
            va, vb = da[k], db[k]
            assert bits_equal(np.array([float(x) for x in va[:7]]), np.array([float(x) for x in vb[:7]])).all()
            assert np.array_equal(va[7], vb[7]) and va[7].dtype == vb[7].dtype
            assert bits_equal(va[8], vb[8]).all()
            assert bits_equal(np.array([float(va[9]), float(va[10]), float(va[11])]),
                              np.array([float(vb[9]), float(vb[10]), float(vb[11])])).all()


@pytest.mark.parametrize("partition", ["lpt", "round_robin", "loader"])
def test_sharded_equals_single_rank(partition):
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    import torch.multiprocessing as mp
    from fluorosequencingimageanalysis_amd import pflib
    single = pflib.find_peptides_batch(_fields())
    assert sum(len(d) for d in single) > 50
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, partition)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=300)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    _same_dicts(got, single)


def _same_records(a, b, fmt=0):
    """Peak records equal byte for byte, except FsqRow.field - the field's number within the chunk of the rank that fitted it."""
    from fluorosequencingimageanalysis_amd import engine as E
    va, vb = E.peak_record_view(a, fmt), E.peak_record_view(b, fmt)
    assert len(va) == len(vb)
    for name in E.RECORD_DTYPE.names:
        if name != "field":
            assert np.array_equal(np.ascontiguousarray(va[name]).view(np.uint8), np.ascontiguousarray(vb[name]).view(np.uint8)), name


def _run_two(partition):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, partition)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=300) for _ in range(2))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    return got


def test_sharded_records_output_gathers_bytes_only():
    """output='records': rank 0 receives the byte table of all fields in field order + the per-field counts - equal to
    find_peptides_records on one GPU, byte for byte -, rank 1 gets None, and no rank builds a dict."""
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from fluorosequencingimageanalysis_amd import pflib
    rec1, counts1, fmt1 = pflib.find_peptides_records(_fields(), c_std=2)
    got = _run_two("records")
    assert got[1] is None
    rec, counts, fmt = got[0]
    assert fmt == fmt1 and np.array_equal(counts, counts1) and rec.dtype == np.uint8
    assert rec.shape == rec1.shape
    _same_records(rec, rec1)
    _same_dicts(pflib.records_to_dicts(rec, counts, fmt), pflib.find_peptides_batch(_fields(), c_std=2))


def test_sharded_records_of_wide_pixels_agree_on_one_format():
    """One field of the stack has pixel values beyond 16 bits: the ranks agree (an all-reduce) to work uint32 words, and rank 0
    receives the 428-byte records of all fields - equal to find_peptides_records on one GPU."""
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from fluorosequencingimageanalysis_amd import _native as N, engine as E, pflib
    rec1, counts1, fmt1 = pflib.find_peptides_records(_fields_wide(), c_std=2)
    assert fmt1 == N.PIXELS_U32 and rec1.shape[1] == E.PEAK_RECORD_BYTES_U32
    got = _run_two("records_wide")
    assert got[1] is None
    rec, counts, fmt = got[0]
    assert fmt == fmt1 and np.array_equal(counts, counts1) and rec.shape == rec1.shape
    _same_records(rec, rec1, fmt)
    d = pflib.records_to_dicts(rec, counts, fmt)
    assert int(max(v[7].max() for v in d[3].values())) > 65535 and all(len(x) > 0 for x in d)


def test_sharded_local_output_builds_dicts_where_the_fields_were_fitted():
    """output='local': every rank returns {global field index: dict} for its own share; together they are the single-rank
    list, and the shares are the LPT partition."""
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from fluorosequencingimageanalysis_amd import distributed as D, pflib
    single = pflib.find_peptides_batch(_fields(), c_std=2)
    got = _run_two("local")
    assert set(got[0]) and set(got[1]) and not (set(got[0]) & set(got[1])) and set(got[0]) | set(got[1]) == set(range(10))
    parts = D._partition([int(x) for x in pflib.count_candidates(_fields(), c_std=2)], 2, "lpt")
    assert sorted(got[0]) == parts[0] and sorted(got[1]) == parts[1]
    merged = {**got[0], **got[1]}
    _same_dicts([merged[i] for i in range(10)], single)


def _rccl_worker(q):
    sys.path.insert(0, ROOT)
    port = _free_port()
    os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch
    import torch.distributed as dist
    from fluorosequencingimageanalysis_amd import distributed as D
    torch.cuda.set_device(0)
    dist.init_process_group(backend="nccl", rank=0, world_size=1)
    assert dist.get_backend() == "nccl"
    rec = D.find_peptides_sharded(_fields(), partition="lpt", output="records", c_std=2, _force_collectives=True)
    dicts = D.find_peptides_sharded(_fields(), partition="round_robin", output="dicts", c_std=2, _force_collectives=True)
    # the variable-length gather itself on device tensors (counts all_gather on the GPU; one rank: no peer to receive from)
    t = torch.arange(12, dtype=torch.uint8, device="cuda").reshape(3, 4)
    table, counts = D.gather_tables(t, 0, _force=True)
    assert counts == [3] and torch.equal(table, t) and table.is_cuda
    torch.cuda.synchronize()
    q.put((rec, dicts))
    dist.barrier()
    dist.destroy_process_group()


def test_rccl_backend_runs_the_exchange_code_on_device_tensors():
    """The `nccl` (= RCCL) branches of the sharded path - the candidate-count all_reduce and the counts all_gather on DEVICE
    tensors, the records handed to the gather as they sit in HBM - executed for real: a process group of ONE rank over RCCL on
    the test box's one GPU (two RCCL ranks cannot share a device), the exchange code forced on.  What a one-rank group
    cannot exercise is the point-to-point send itself; that is covered over gloo by the two-rank tests above."""
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    import torch.multiprocessing as mp
    from fluorosequencingimageanalysis_amd import pflib
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_worker, args=(q,))
    p.start()
    (rec, counts, fmt), dicts = q.get(timeout=300)
    p.join(timeout=120)
    assert p.exitcode == 0
    rec1, counts1, fmt1 = pflib.find_peptides_records(_fields(), c_std=2)
    _same_records(rec, rec1)
    assert np.array_equal(counts, counts1) and fmt == fmt1
    _same_dicts(dicts, pflib.find_peptides_batch(_fields(), c_std=2))


def test_sharded_world1_is_batch():
    """Without a process group find_peptides_sharded is find_peptides_batch."""
    from fluorosequencingimageanalysis_amd import distributed as D, pflib
    imgs = _fields()[:3]
    _same_dicts(D.find_peptides_sharded(imgs), pflib.find_peptides_batch(imgs))


# ---- the parallel FILE path under two ranks: basic_image_script / pflib.parallel_image_batch (pflib.py:1043-1111) ----------
def _make_tiffs(d):
    """10 ragged 16-bit TIFFs of two shapes (5 .. 60 spots: very different candidate counts) and one unreadable file."""
    from PIL import Image
    from fluorosequencingimageanalysis_amd import synth
    os.makedirs(os.path.join(d, "sub"), exist_ok=True)
    paths = []
    for i, spots in enumerate((5, 60, 10, 40, 25, 8, 50, 30, 15, 35)):
        img = synth.make_field(500 + i, (96, 96) if i % 3 else (64, 80), spots)
        p = os.path.join(d, "sub" if i % 4 == 0 else "", "im%02d.tif" % i)
        Image.fromarray(img).save(p, format="TIFF")
        paths.append(p)
    open(os.path.join(d, "broken.tif"), "wb").write(b"II*\x00garbage")
    return paths


def _cli_worker(rank, world, port, q, d):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0", FSQ_DIST_BACKEND="gloo")
    import torch
    torch.cuda.set_device(0)
    from fluorosequencingimageanalysis_amd import basic_image_script as cli, pflib
    seen = {}
    real = pflib.image_batch

    def spy(paths, *a, **k):            # which images this rank was dealt
        seen["mine"] = list(paths)
        return real(paths, *a, **k)
    pflib.image_batch = spy
    res = cli.main(["--parameters", "{'c_std': 2}", "-n", "2", "-L", os.path.join(d, "log.txt"), d])
    q.put((rank, res, seen.get("mine", [])))


def test_cli_two_ranks_equal_one_rank(tmp_path):
    """python -m ... basic_image_script under two ranks (fresh processes on the one GPU, gloo): the merged result dict, every
    image's pickle / CSV and the longest-processing-time assignment equal what a single rank produces; the unreadable file is
    logged and skipped; both ranks return the same dict."""
    import pickle
    import warnings
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    import torch.multiprocessing as mp
    from fluorosequencingimageanalysis_amd import basic_image_script as cli, distributed as D, pflib
    d1, d2 = str(tmp_path / "one"), str(tmp_path / "two")
    _make_tiffs(d1)
    paths2 = _make_tiffs(d2)
    one = cli.main(["--parameters", "{'c_std': 2}", "-L", os.path.join(d1, "log.txt"), d1])
    assert len(one) == 10
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_cli_worker, args=(r, 2, port, q, d2)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict((r, (res, mine)) for r, res, mine in (q.get(timeout=600) for _ in range(2)))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert got[0][0] == got[1][0]
    two = got[0][0]
    assert sorted(os.path.relpath(p, d2) for p in two) == sorted(os.path.relpath(p, d1) for p in one)
    # every rank got a share, the shares are disjoint, cover the readable images and follow the LPT rule on the counts
    mine0, mine1 = set(got[0][1]), set(got[1][1])
    assert mine0 and mine1 and not (mine0 & mine1) and (mine0 | mine1) == set(two)
    listed = cli.find_target_images([d2])
    counts = pflib._candidate_counts(listed)
    assign = D.lpt_assignment(listed, counts, 2)
    assert {p for p, r in assign.items() if r == 0} == mine0 and {p for p, r in assign.items() if r == 1} == mine1
    assert "broken.tif" in open(os.path.join(d2, "log.txt")).read() + open(os.path.join(d2, "log.txt.rank1")).read()
    for p2, v2 in two.items():
        v1 = one[os.path.join(d1, os.path.relpath(p2, d2))]
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            a, b = pickle.load(open(v1[1], "rb")), pickle.load(open(v2[1], "rb"))
        _same_dicts([a], [b])
        t1, t2 = open(v1[2], newline="").read().split("\r\n"), open(v2[2], newline="").read().split("\r\n")
        assert [x.split("\t")[1:] for x in t1] == [x.split("\t")[1:] for x in t2] and len(t1) == len(a) + 2
