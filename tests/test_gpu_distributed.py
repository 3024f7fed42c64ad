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


def _worker(rank, world, port, q, partition):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch
    import torch.distributed as dist
    from fluorosequencingimageanalysis_amd import distributed as D
    torch.cuda.set_device(0)
    D.init_from_env(backend="gloo")
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


@pytest.mark.parametrize("partition", ["lpt", "round_robin"])
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


def test_sharded_world1_is_batch():
    """Without a process group find_peptides_sharded is find_peptides_batch."""
    from fluorosequencingimageanalysis_amd import distributed as D, pflib
    imgs = _fields()[:3]
    _same_dicts(D.find_peptides_sharded(imgs), pflib.find_peptides_batch(imgs))
