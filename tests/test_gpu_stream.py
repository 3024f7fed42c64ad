"""GPU tests (-m gpu) of continuous batching: engine.FitQueue / engine.StreamPipeline (fsq_fitq_* of include/fsq.h).

Fits are independent, so a batch that shares the round launches with other batches must come out exactly as when it is
fitted alone: every test compares the pipeline's rows / kept peaks with Engine.run on the same fields, byte for byte
(Engine.run itself is pinned to the reference's goldens by test_gpu_fit.py / test_gpu_pipeline.py)."""
import numpy as np
import pytest

from _util import load_field

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from fluorosequencingimageanalysis_amd import _native, engine, pflib, synth
    return torch, _native, engine, pflib, synth


def _single(E, d_img, prm, n, H, W):
    eng = E.Engine(n, H, W)
    total = eng.run(d_img, prm)
    return _snapshot(eng, total)


def _snapshot(eng, total):
    nk = eng.nkeep.cpu().numpy().copy()
    off = eng.offsets.cpu().numpy().copy()
    keep = eng.keep[:max(total, 1)].cpu().numpy().copy()
    kept = [keep[off[f]:off[f] + max(int(nk[f]), 0)].copy() for f in range(eng.n_fields)]
    return total, eng.rows[:total].cpu().numpy().tobytes(), nk, kept, eng.counts.cpu().numpy().copy()


def _same(a, b):
    assert a[0] == b[0]
    assert a[1] == b[1], "rows differ"
    assert np.array_equal(a[2], b[2]) and np.array_equal(a[4], b[4])
    for x, y in zip(a[3], b[3]):
        assert np.array_equal(x, y)


@pytest.mark.parametrize("inject_below,depth", [(1 << 40, 8), (0, 2), (2000, 3)])
def test_stream_pipeline_equals_single_engine(env, inject_below, depth):
    """Seven different batches through the pipeline - everything in flight at once, strictly one after the other,
    and in between - each equal to its stand-alone run.  The pool is small enough that slots are re-used in ring
    order while older batches are still in flight."""
    torch, N, E, pflib, synth = env
    n, H, W = 4, 128, 128
    prm = E.detect_params(5, pflib.default_correlation_matrix, 2)
    batches = [np.stack([synth.make_field(900 + 10 * b + i, (H, W), 20 + 7 * b) for i in range(n)]) for b in range(7)]
    batches[3] = np.full((n, H, W), 100, np.uint16)             # flat frames: every interior pixel is a candidate
    d_imgs = [E.to_device_u16(x) for x in batches]
    ref = [_single(E, d, prm, n, H, W) for d in d_imgs]
    assert ref[3][0] == n * (H - 4) * (W - 4) and ref[0][0] > 0
    per = max(r[0] for r in ref) + 64
    pipe = E.StreamPipeline(n, H, W, depth=depth, cand_per_batch=per, inject_below=inject_below)
    got = {}

    def on_done(j, eng, total):
        got[j] = _snapshot(eng, total)

    for rep in range(2):                                         # a second pass re-uses the queue and its tickets
        got.clear()
        totals = pipe.run([(d, prm) for d in d_imgs], on_done)
        assert totals == [r[0] for r in ref]
        assert sorted(got) == list(range(7))
        for j in range(7):
            _same(got[j], ref[j])
    assert pipe.queue.alive == 0
    pipe.close()


def test_pipeline_group_equals_single_engine(env):
    """engine.StreamPipelineGroup: the fields of every batch split over two fit queues on their own streams / threads -
    every share equals the corresponding fields of the stand-alone run."""
    torch, N, E, pflib, synth = env
    n, H, W = 6, 128, 128
    prm = E.detect_params(5, pflib.default_correlation_matrix, 2)
    batches = [np.stack([synth.make_field(700 + 10 * b + i, (H, W), 15 + 9 * b) for i in range(n)]) for b in range(4)]
    d_imgs = [E.to_device_u16(x) for x in batches]
    group = E.StreamPipelineGroup(n, H, W, queues=2, depth=3, inject_below=1500)
    assert group.cut == [0, 3, 6]
    got = {}

    def on_done(j, k, eng, total):
        got[(j, k)] = (eng.all_rows(total), eng.counts.cpu().numpy().copy(), eng.kept_table()[0].cpu().numpy().copy())

    totals = group.run([(d, prm) for d in d_imgs], on_done)
    group.close()
    for j, d in enumerate(d_imgs):
        one = E.Engine(n, H, W)
        total = one.run(d, prm)
        assert totals[j] == total
        rows = one.all_rows(total)
        counts = one.counts.cpu().numpy()
        table = one.kept_table()[0].cpu().numpy().view(N.ROW_DTYPE).reshape(-1)
        lo = 0
        klo = 0
        for k in range(2):
            r, c, t = got[(j, k)]
            nk = int(c[:3].sum())
            a, b = r.copy(), rows[lo:lo + nk].copy()
            for x in (a, b):
                x["key_h"] = x["key_w"] = -1
            a["field"] += 3 * k                         # rows carry the field index inside their share
            assert a.tobytes() == b.tobytes()
            t = t.view(N.ROW_DTYPE).reshape(-1).copy()
            t["field"] += 3 * k
            assert t.tobytes() == table[klo:klo + len(t)].tobytes()
            lo += nk
            klo += len(t)
        assert lo == total and klo == len(table)


def test_stream_pipeline_golden_field(env):
    """The reference's own table for golden field f1 comes out of the pipeline while other batches share its rounds."""
    torch, N, E, pflib, synth = env
    g, img = load_field("f1_cfg2_512_500")
    other = synth.make_field(4242, (512, 512), 300)
    prm = E.detect_params(5, pflib.default_correlation_matrix, 2)
    jobs = [(E.to_device_u16(np.stack([other, img])), prm), (E.to_device_u16(np.stack([img, other])), prm),
            (E.to_device_u16(np.stack([other, other])), prm)]
    pipe = E.StreamPipeline(2, 512, 512, depth=4, inject_below=1 << 40)
    out = {}

    def on_done(j, eng, total):
        rows = eng.all_rows(total)
        counts = eng.counts.cpu().numpy()
        out[j] = (rows, counts, eng.kept_tables(total))

    pipe.run(jobs, on_done)
    for j, f in ((0, 1), (1, 0)):
        rows, counts, tables = out[j]
        lo = int(counts[:f].sum())
        r = rows[lo:lo + counts[f]]
        assert len(r) == len(g["candidates"])
        p = np.stack([r[k] for k in ("H", "A", "p2", "p3", "sigma_h", "sigma_w", "theta")], axis=-1)
        assert (p.view(np.uint64) == np.ascontiguousarray(g["params"]).view(np.uint64)).all()
        assert (r["status"] == g["status"]).all() and (r["niter"] == g["niter"]).all()
        kept_rows, _fit = tables[f]
        assert [(int(a), int(b)) for a, b in zip(kept_rows["key_h"], kept_rows["key_w"])] == \
               [tuple(int(v) for v in k) for k in g["table_keys"]]
    pipe.close()


def test_fit_queue_capacity_and_errors(env):
    """submit refuses (returns None / FSQ_EAGAIN) instead of overrunning the pool or the queues, bad arguments give
    ValueError, and a refused batch goes through once room has been made."""
    torch, N, E, pflib, synth = env
    n, H, W = 2, 96, 96
    prm = E.detect_params(5, pflib.default_correlation_matrix, 2)
    d = E.to_device_u16(np.stack([synth.make_field(70 + i, (H, W), 15) for i in range(n)]))
    total, ref_rows = _single(E, d, prm, n, H, W)[:2]

    def no_keys(b):                                              # (consolidation writes the dict keys into kept rows)
        r = np.frombuffer(b, dtype=N.ROW_DTYPE).copy()
        r["key_h"] = r["key_w"] = -1
        return r.tobytes()
    engs = [E.Engine(n, H, W, fit_workspace=False) for _ in range(3)]
    for e in engs:
        assert e.detect(d, prm) == total
    with pytest.raises(RuntimeError):
        engs[0].fit(d, total)                                    # no fit workspace: fails loudly
    torch.cuda.synchronize()
    q = E.FitQueue(pool_slots=2 * total, queue_cap=total + total // 2)
    with pytest.raises(ValueError):
        E.FitQueue(pool_slots=0, queue_cap=10)
    t0 = q.submit(d, n, H, W, engs[0].cand, total, engs[0].rows)
    assert t0 is not None and q.alive == total
    assert q.submit(d, n, H, W, engs[1].cand, total, engs[1].rows) is None        # queue_cap: 2 x total alive is too many
    q.advance(2, 0)
    while q.alive > total // 2:
        q.advance(1, 0)
    t1 = q.submit(d, n, H, W, engs[1].cand, total, engs[1].rows)                  # pool: exactly two batches fit
    assert t1 is not None
    assert q.submit(d, n, H, W, engs[2].cand, total + 1, engs[2].rows) is None    # no room while t1 is in flight
    cur = torch.cuda.current_stream()
    done = set()
    while len(done) < 2:
        q.advance(0, 0)
        for t in (t0, t1):
            if t not in done and q.take(t, cur):
                done.add(t)
    with pytest.raises(ValueError):
        q.take(t0, cur)                                          # released tickets are invalid
    t_empty = q.submit(d, n, H, W, engs[2].cand, 0, engs[2].rows)                 # a batch without a single candidate
    assert t_empty is not None and q.take(t_empty, cur)
    t2 = q.submit(d, n, H, W, engs[2].cand, total, engs[2].rows)
    assert t2 is not None
    while not q.take(t2, cur):
        q.advance(0, 0)
    torch.cuda.synchronize()
    for e in engs:
        assert no_keys(e.rows[:total].cpu().numpy().tobytes()) == no_keys(ref_rows)
    q.close()


@pytest.mark.parametrize("force_slow", [0, 1])
def test_fit_queue_filled_exactly_to_its_capacity(env, monkeypatch, force_slow):
    """A queue of 256 positions holding exactly 256 live fits, submitted as 100 + 156 (neither a multiple of 64): every list
    becomes exactly full - in round 1 every fit sits in B lo - and the lanes that append nothing must not trip the capacity
    guard (they were handed the counter value behind the wave's reservation, == cap, and reported FSQ_EINTERNAL).
    force_slow: every fit through the plain-division kernel in every round (its appends go through the same reservation)."""
    torch, N, E, pflib, synth = env
    if force_slow:
        monkeypatch.setenv("FSQ_DEBUG_FORCE_SLOW", "1")
    H = W = 96
    prm = E.detect_params(5, pflib.default_correlation_matrix, 2)
    d = E.to_device_u16(np.stack([synth.make_field(170 + i, (H, W), 25) for i in range(2)]))
    eng = E.Engine(2, H, W)
    total = eng.run(d, prm)
    assert total >= 256
    ref = eng.rows[:256].cpu().numpy().view(N.ROW_DTYPE).reshape(-1).copy()
    e1, e2 = E.Engine(2, H, W, fit_workspace=False), E.Engine(2, H, W, fit_workspace=False)
    for e in (e1, e2):
        assert e.detect(d, prm) == total
    torch.cuda.synchronize()
    q = E.FitQueue(pool_slots=512, queue_cap=256)
    assert q.queue_cap == 256
    t1 = q.submit(d, 2, H, W, e1.cand, 100, e1.rows)
    t2 = q.submit(d, 2, H, W, e2.cand[100:], 156, e2.rows)
    assert t1 is not None and t2 is not None and q.alive == 256
    assert q.submit(d, 2, H, W, e2.cand, 1, e2.rows) is None          # not one more
    cur = torch.cuda.current_stream()
    done = set()
    while len(done) < 2:
        q.advance(0, 0)                                                # (raises RuntimeError on FSQ_EINTERNAL)
        for t in (t1, t2):
            if t not in done and q.take(t, cur):
                done.add(t)
    torch.cuda.synchronize()
    got = np.concatenate([e1.rows[:100].cpu().numpy().view(N.ROW_DTYPE).reshape(-1), e2.rows[:156].cpu().numpy().view(N.ROW_DTYPE).reshape(-1)])
    for x in (got, ref):
        x["key_h"] = x["key_w"] = -1
    assert got.tobytes() == ref.tobytes()
    q.close()
