"""GPU parity tests (-m gpu) of the whole per-field path through the drop-in surface
(fluorosequencingimageanalysis_amd.pflib) and the C ABI, against the oracle and the reference's goldens."""
import numpy as np
import pytest

from _util import FIELD_NAMES, PARAM_NAMES, TINY_NAMES, bits_equal, golden_params, load_field

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    import torch
    assert torch.cuda.is_available()
    import oracle as O
    O.build()
    from fluorosequencingimageanalysis_amd import pflib, engine
    return torch, pflib, engine, O


@pytest.mark.parametrize("name", FIELD_NAMES)
def test_candidates_exact(env, name):
    torch, pflib, engine, O = env
    g, img = load_field(name)
    got = pflib._psf_candidates(img)
    assert got == [tuple(int(v) for v in hw) for hw in g["candidates"]]
    assert all(isinstance(h, int) and isinstance(w, int) for h, w in got[:5])


@pytest.mark.parametrize("name", FIELD_NAMES)
def test_threshold_and_counts(env, name):
    torch, pflib, engine, O = env
    g, img = load_field(name)
    H, W = img.shape
    eng = engine.Engine(1, H, W)
    total = eng.detect(engine.to_device_u16(img[None]), engine.detect_params(5, pflib.default_correlation_matrix, 2))
    _, _, thr = O.candidates(img, return_cm=True)
    assert eng.thr.cpu().numpy()[0] == thr             # numpy.mean + 2*numpy.std, bit for bit
    assert total == len(g["candidates"])


@pytest.mark.parametrize("name", FIELD_NAMES)
def test_find_peptides_equals_reference(env, name):
    """dict keys, order, the 12-tuples (incl. sub_img and fit_img) == the reference's own output."""
    torch, pflib, engine, O = env
    g, img = load_field(name)
    d = pflib.find_peptides(img)
    keys = np.array(list(d.keys()), dtype=np.int32).reshape(-1, 2)
    assert np.array_equal(keys, g["table_keys"])
    vals = list(d.values())
    got7 = np.array([[float(x) for x in v[:7]] for v in vals]).reshape(-1, 7)
    assert bits_equal(got7, g["table7"]).all()
    gotm = np.array([[float(v[9]), float(v[10]), float(v[11])] for v in vals]).reshape(-1, 3)
    assert bits_equal(gotm, g["table_metrics"]).all()
    assert np.array_equal(np.array([v[7] for v in vals]).reshape(-1, 5, 5), g["table_sub"])
    assert bits_equal(np.array([v[8] for v in vals]).reshape(-1, 5, 5), g["table_fit"]).all()
    v = vals[0]
    assert v[7].dtype == np.int64 and v[8].dtype == np.float64 and isinstance(v[9], float)


def test_batch_equals_single(env):
    torch, pflib, engine, O = env
    imgs = np.stack([load_field("f0_cfg1_512_200")[1], load_field("f1_cfg2_512_500")[1]])
    out = pflib.find_peptides_batch(imgs)
    for f, name in enumerate(("f0_cfg1_512_200", "f1_cfg2_512_500")):
        g, _ = load_field(name)
        assert np.array_equal(np.array(list(out[f].keys()), dtype=np.int32).reshape(-1, 2), g["table_keys"])


def test_nondefault_parameters(env):
    """other median size / kernel / c_std / radius / threshold vs the oracle."""
    torch, pflib, engine, O = env
    g, img = load_field("f3_hard_256")
    K = np.array([[-1, -2, -1], [-2, 13, -2], [-1, -2, -2]])
    for med, KK, c_std, r2, rad in ((3, K, 1.5, 0.5, 2), (4, pflib.default_correlation_matrix, 2.5, 0.8, 6),
                                    (7, np.array([[5]]), 3, 0.7, 3)):
        cand = pflib._psf_candidates(img, median_filter_size=med, correlation_matrix=KK, c_std=c_std)
        ref = O.candidates(img, med_size=med, K=KK, c_std=c_std)
        assert cand == [tuple(int(v) for v in hw) for hw in ref]
        d = pflib.find_peptides(img, median_filter_size=med, correlation_matrix=KK, c_std=c_std, r_2_threshold=r2,
                                consolidation_radius=rad)
        rows, fits, keep, key = O.find_peptides(img, med_size=med, K=KK, c_std=c_std, r2_thr=r2, radius=rad, n_threads=8)
        assert np.array_equal(np.array(list(d.keys()), dtype=np.int32).reshape(-1, 2), key)
        got = np.array([[float(x) for x in v[:7]] for v in d.values()]).reshape(-1, 7)
        r = rows[keep]
        exp = np.stack([r[k] for k in ("h0", "w0", "H", "A", "sigma_h", "sigma_w", "theta")], axis=1)
        assert bits_equal(got, exp).all()


@pytest.mark.parametrize("name", ["params_" + n for n in PARAM_NAMES] + ["tiny_" + n for n in TINY_NAMES])
def test_nondefault_keywords_equal_reference(env, name):
    """find_peptides with non-default median windows / correlation matrices / c_std / r_2 threshold / consolidation radius
    against the reference's own recorded tables (tests/golden/params_*.npz, oracle/gen_golden.py --only params), and on frames
    of one 5 x 5 neighbourhood or little more, smaller than their windows (tiny_*.npz, --only tiny)."""
    torch, pflib, engine, O = env
    prefix, name = name.split("_", 1)
    g, img = load_field(name, prefix=prefix + "_")
    prm = golden_params(g)
    det = {k: prm[k] for k in ("median_filter_size", "correlation_matrix", "c_std") if k in prm}
    assert pflib._psf_candidates(img, **det) == [tuple(int(v) for v in hw) for hw in g["candidates"]]
    d = pflib.find_peptides(img, **prm)
    assert np.array_equal(np.array(list(d.keys()), dtype=np.int32).reshape(-1, 2), g["table_keys"].reshape(-1, 2))
    vals = list(d.values())
    assert bits_equal(np.array([[float(x) for x in v[:7]] for v in vals]).reshape(-1, 7), g["table7"].reshape(-1, 7)).all()
    assert np.array_equal(np.array([v[7] for v in vals]).reshape(-1, 5, 5), g["table_sub"])
    assert bits_equal(np.array([v[8] for v in vals]).reshape(-1, 5, 5), g["table_fit"]).all()
    assert bits_equal(np.array([[float(v[9]), float(v[10]), float(v[11])] for v in vals]).reshape(-1, 3),
                      g["table_metrics"].reshape(-1, 3)).all()


def test_large_windows(env):
    """Median windows and correlation matrices beyond 9 x 9 (the limit until round 4; now FSQ_MAX_KSIZE = 15): candidates equal to
    the oracle's, and the size beyond the limit is refused with a message that names it."""
    torch, pflib, engine, O = env
    g, img = load_field("f5_small_96")
    rng = np.random.default_rng(15)
    for med, ks in ((11, 13), (15, 15), (5, 11), (12, 3)):
        KK = rng.integers(-5, 6, (ks, ks)).astype(np.int64)
        KK[ks // 2, ks // 2] += 40
        cand = pflib._psf_candidates(img, median_filter_size=med, correlation_matrix=KK, c_std=1.5)
        ref = O.candidates(img, med_size=med, K=KK, c_std=1.5)
        assert len(ref) > 0 and cand == [tuple(int(v) for v in hw) for hw in ref]
    with pytest.raises(NotImplementedError, match="15"):
        pflib._psf_candidates(img, median_filter_size=5, correlation_matrix=np.ones((17, 17), np.int64))
    with pytest.raises(NotImplementedError, match="15"):
        pflib._psf_candidates(img, median_filter_size=16)


def test_edge_cases(env):
    torch, pflib, engine, O = env
    # empty: a flat image has no candidates above threshold... (cm all zero -> thr 0 -> every interior pixel passes)
    flat = np.full((16, 16), 100, np.uint16)
    assert pflib._psf_candidates(flat) == [tuple(int(v) for v in hw) for hw in O.candidates(flat)]
    tiny = np.full((5, 5), 7, np.uint16)
    tiny[2, 2] = 900
    assert pflib._psf_candidates(tiny) == [tuple(int(v) for v in hw) for hw in O.candidates(tiny)]
    sat = np.full((32, 32), 65535, np.uint16)
    assert pflib._psf_candidates(sat) == [tuple(int(v) for v in hw) for hw in O.candidates(sat)]
    with pytest.raises(ValueError):
        pflib.find_peptides(flat, consolidation_radius=1)
    with pytest.raises(ValueError):
        pflib._psf_candidates(flat, correlation_matrix=np.ones((4, 4), int))
    with pytest.raises(NotImplementedError):
        pflib._fit_2d_gaussian(np.zeros((5, 5), int), implementation='scipy')
    with pytest.raises(AssertionError):
        pflib._fit_2d_gaussian(np.zeros((4, 5), int))
    with pytest.raises(NotImplementedError):
        pflib.find_peptides(flat, fit_type='monte_carlo')


def test_fit_2d_gaussian_surface(env):
    torch, pflib, engine, O = env
    g, img = load_field("f5_small_96")
    h, w = g["candidates"][3]
    sub = img[h - 2:h + 3, w - 2:w + 3].astype(np.int64)
    r = pflib._fit_2d_gaussian(sub)
    p = g["params"][3]
    assert [float(x) for x in r[:7]] == [p[2], p[3], p[0], p[1], p[4], p[5], p[6]]
    assert bits_equal(r[7], O.model(p)).all()
    assert pflib.illumina_s_n(sub) == O.illumina_s_n(sub)


def test_concurrent_calls_and_cache_eviction():
    """Three threads call find_peptides_batch on five stack shapes in different orders (more shapes than the resource cache
    holds, so runners are evicted - closed - while other threads are mid-call or about to call): every result equals the
    single-threaded one."""
    import threading
    from fluorosequencingimageanalysis_amd import pflib, synth
    shapes = [(48, 48), (64, 40), (40, 72), (56, 56), (80, 48)]
    stacks = [np.stack([synth.make_field(700 + 10 * k + i, s, 6 + i) for i in range(3)]) for k, s in enumerate(shapes)]
    want = [pflib.find_peptides_batch(st) for st in stacks]
    errs = []

    def same(a, b):
        return len(a) == len(b) and all(list(x) == list(y) and all(np.array_equal(np.asarray(u), np.asarray(v), equal_nan=True)
                                                                    for k in x for u, v in zip(x[k], y[k])) for x, y in zip(a, b))

    def work(order):
        try:
            for rep in range(3):
                for k in order:
                    if not same(pflib.find_peptides_batch(stacks[k]), want[k]):
                        errs.append("shape %s differs" % (shapes[k],))
        except Exception as e:      # noqa: BLE001 - reported below
            errs.append(repr(e))

    ths = [threading.Thread(target=work, args=(o,)) for o in ([0, 1, 2, 3, 4], [4, 3, 2, 1, 0], [2, 0, 4, 1, 3])]
    for t in ths:
        t.start()
    for t in ths:
        t.join(timeout=600)
        assert not t.is_alive()
    assert not errs, errs[:3]


def test_whole_path_entry_point_equals_the_staged_calls():
    """fsq_find_peptides (one call: images in HBM -> peak records) gives byte for byte the records the Python layer assembles
    from fsq_detect / fsq_fit_candidates / fsq_consolidate / fsq_kept_rows / fsq_fit_images, also when its buffers start too
    small (FSQ_ERANGE -> grown) and for float16 pixels."""
    import torch
    from fluorosequencingimageanalysis_amd import _native as N, engine as E, pflib, synth
    imgs = np.stack([synth.make_field(40 + i, (96, 128), 10 + 7 * i) for i in range(5)])
    for stack in (imgs, imgs.astype(np.float16)):
        words, fmt = E.as_pixel_fields(stack)
        prm = E.detect_params(5, pflib.default_correlation_matrix, 2, fmt)
        d = E.to_device_u16(words)
        eng = E.Engine(5, 96, 128)
        eng.run(d, prm, 0.7, 4, N.MODE_REF, True)
        rec0, offs0 = eng.peak_records(d)
        for caps in ((None, None), (64, 8)):
            pr = E.PathRunner(5, 96, 128, cand_cap=caps[0], record_cap=caps[1])
            rec, offs, nk, ncand = pr.run(d, prm, 0.7, 4, N.MODE_REF, True)
            torch.cuda.synchronize()
            assert ncand == int(eng.offsets[5].item()) and len(rec) == len(rec0) > 20
            assert torch.equal(rec, rec0) and torch.equal(offs, offs0) and torch.equal(nk, eng.nkeep)


def test_engine_buffers_may_hold_anything():
    """detect -> fit -> consolidate with every workspace / output buffer of the Engine pre-filled with random bits (recycled
    device memory is not zero): the tables still equal the oracle's."""
    import torch
    import oracle as O
    from fluorosequencingimageanalysis_amd import _native as N, engine as E, pflib, synth
    O.build()
    imgs = np.stack([synth.make_field(60 + i, (80, 112), 8 + 5 * i) for i in range(4)])
    eng = E.Engine(4, 80, 112)
    gen = torch.Generator(device="cuda").manual_seed(3)
    for t in (eng.ws, eng.fit_ws, eng.rows, eng.cand, eng.keep, eng.counts, eng.offsets, eng.nkeep, eng.thr):
        raw = t.view(torch.uint8)
        raw.copy_(torch.randint(0, 256, (raw.numel(),), dtype=torch.uint8, device="cuda", generator=gen).reshape(raw.shape))
    d = E.to_device_u16(imgs)
    prm = E.detect_params(5, pflib.default_correlation_matrix, 2)
    eng.run(d, prm, 0.7, 4, N.MODE_REF, True)
    dicts = pflib._engine_dicts(eng, d)
    for f in range(4):
        rows, fits, keep, key = O.find_peptides(imgs[f])
        assert [tuple(k) for k in key.tolist()] == list(dicts[f])
        exp = np.stack([rows[keep][k] for k in ("h0", "w0", "H", "A", "sigma_h", "sigma_w", "theta")], axis=1)
        val = np.array([[float(x) for x in v[:7]] for v in dicts[f].values()]).reshape(-1, 7)
        assert np.array_equal(val.view(np.uint64), exp.view(np.uint64))


def test_fresh_device_memory_may_hold_anything(monkeypatch):
    """Every device buffer the Python layer allocates with torch.empty is handed out pre-filled with random bits (what a
    long-lived process gets from the caching allocator): registration, both trackers, photometry and the batch surface still
    equal the oracle / the undisturbed results."""
    import torch
    import oracle as O
    from fluorosequencingimageanalysis_amd import flexlibrary as fl, pflib, phase_correlate as pc, photometry, synth
    O.build()
    rng = np.random.default_rng(9)
    imgs = np.stack([synth.make_field(80 + i, (96, 96), 10 + 3 * i) for i in range(4)])
    want_batch = pflib.find_peptides_batch(imgs)
    ref, reg = imgs[0], np.roll(imgs[0], (3, -2), (0, 1))
    want_pc = pc.phase_correlate(ref, reg, upsample_factor=20)
    hw = [np.array(sorted(d), np.int32).reshape(-1, 2) for d in want_batch]
    want_tr = fl.track_fields([hw], [[(0.0, 0.0)] * 4], (96, 96), 2, 0.0)
    want_ph = photometry.mexican_hat_photometry_metric(imgs[:1], np.concatenate([np.zeros((len(hw[0]), 1), np.int32), hw[0]], 1))
    pflib.release_gpu_resources()
    real_empty = torch.empty
    gen = torch.Generator(device="cuda").manual_seed(1)

    def dirty_empty(*a, **k):
        t = real_empty(*a, **k)
        if t.is_cuda and t.numel():
            raw = t.view(torch.uint8) if t.is_contiguous() else None
            if raw is not None:
                raw.copy_(torch.randint(0, 256, (raw.numel(),), dtype=torch.uint8, device=t.device, generator=gen).reshape(raw.shape))
                torch.cuda.current_stream(t.device).synchronize()  # (the fill must not race with the buffer's first use on another stream)
        return t
    monkeypatch.setattr(torch, "empty", dirty_empty)
    got_batch = pflib.find_peptides_batch(imgs)
    got_pc = pc.phase_correlate(ref, reg, upsample_factor=20)
    got_tr = fl.track_fields([hw], [[(0.0, 0.0)] * 4], (96, 96), 2, 0.0)
    got_ph = photometry.mexican_hat_photometry_metric(imgs[:1], np.concatenate([np.zeros((len(hw[0]), 1), np.int32), hw[0]], 1))
    monkeypatch.setattr(torch, "empty", real_empty)
    pflib.release_gpu_resources()
    assert [list(d) for d in got_batch] == [list(d) for d in want_batch]
    assert all(np.array_equal(np.asarray(x), np.asarray(y), equal_nan=True) for a, b in zip(got_batch, want_batch) for k in a for x, y in zip(a[k], b[k]))
    assert tuple(float(x) for x in got_pc) == tuple(float(x) for x in want_pc)
    assert all(np.array_equal(x, y) if isinstance(x, np.ndarray) else x == y for a, b in zip(got_tr, want_tr) for x, y in zip(a, b))
    assert np.array_equal(np.asarray(got_ph), np.asarray(want_ph), equal_nan=True)
