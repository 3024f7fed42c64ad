"""GPU tests (-m gpu) of the tracking kernel (csrc/fsq_track.hip): trace membership and order equal to
the reference's Experiment.greedy_particle_tracking as recorded in tests/golden/tracking.npz, and to the oracle on
random layouts; the integer restatement of the x87 dnrm2 distance against the oracle's long-double form."""
import numpy as np
import pytest

from test_tracking import load_cases, load_centroid_cases

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from fluorosequencingimageanalysis_amd import _native, flexlibrary
    import oracle as O
    O.build()
    return torch, _native, flexlibrary, O


@pytest.mark.parametrize("case", list(load_cases()), ids=lambda c: c[0])
def test_tracking_equals_reference(env, case):
    torch, N, fl, O = env
    name, frame_hw, offsets, shape, radius, spot_radius, traces, discarded = case
    got, nd, prev, nxt, kept = fl.track_fields([frame_hw], [offsets], shape, radius, spot_radius)[0]
    assert nd == discarded and got.shape == traces.shape and np.array_equal(got, traces)
    o_tr, o_nd, o_prev, o_next, o_kept = O.greedy_tracking(frame_hw, offsets, shape, radius, spot_radius)
    assert np.array_equal(prev, o_prev) and np.array_equal(nxt, o_next) and np.array_equal(kept, o_kept)


@pytest.mark.parametrize("case", list(load_cases("tracking_long.npz")), ids=lambda c: c[0])
def test_long_series_and_large_fields_equal_reference(env, case):
    """90 / 70 frames and 34 000 spots (tests/golden/tracking_long.npz: the reference's own traces): the frame tables in the
    workspace instead of LDS, the pairing facts read off the links instead of LDS bitmaps."""
    torch, N, fl, O = env
    name, frame_hw, offsets, shape, radius, spot_radius, traces, discarded = case
    got, nd, prev, nxt, kept = fl.track_fields([frame_hw], [offsets], shape, radius, spot_radius)[0]
    assert nd == discarded and got.shape == traces.shape and np.array_equal(got, traces)


@pytest.mark.parametrize("case", list(load_centroid_cases("centroid_tracking_wide.npz")), ids=lambda c: c[0])
def test_centroid_tracking_of_wide_frames_equals_reference(env, case):
    """uint32 frames with pixel values beyond 16 bits (fsq_centroid_tracking_u32) against the reference's recorded tracks
    (tests/golden/centroid_tracking_wide.npz) and the oracle."""
    torch, N, fl, O = env
    name, frames, init, offsets, sr, cut, hw = case
    got, present = fl.centroid_track_fields(frames, init, None, sr, cut, None if offsets is None else offsets[None])
    assert np.array_equal(got, hw) and np.array_equal(present, hw[:, :, 0] >= 0)
    o_hw, o_pr = O.centroid_tracking(frames, init, sr, cut, offsets)
    assert np.array_equal(got, o_hw) and np.array_equal(present, o_pr)


def test_tracking_batch_and_object_surface(env):
    """Many fields per launch (random layouts vs the oracle), and the Experiment.greedy_particle_tracking surface on
    Spot-like objects."""
    torch, N, fl, O = env
    rng = np.random.default_rng(5)
    fields, offs = [], []
    F, shape = 6, (96, 128)
    for k in range(40):
        base = rng.integers(4, 90, (60, 2)).astype(np.int64)
        base[:, 1] = rng.integers(4, 122, 60)
        base = base[np.unique(base[:, 0] * 1000 + base[:, 1], return_index=True)[1]]
        frames, o, cum = [], [(0, 0)], np.zeros(2)
        for f in range(F):
            if f:
                step = np.round(rng.uniform(-2, 2, 2) * 20) / 20
                o.append((float(step[0]), float(step[1])))
                cum = cum + step
            pts = np.rint(base - cum).astype(np.int64) + rng.integers(-1, 2, base.shape)
            pts = pts[rng.uniform(size=len(pts)) > 0.2]
            pts = pts[np.unique(pts[:, 0] * 1000 + pts[:, 1], return_index=True)[1]]
            frames.append(pts)
        fields.append(frames)
        offs.append(o)
    res = fl.track_fields(fields, offs, shape)
    n_checked = 0
    for frames, o, r in zip(fields, offs, res):
        try:
            exp = O.greedy_tracking(frames, o, shape)
        except AssertionError:
            continue
        assert np.array_equal(r[0], exp[0]) and r[1] == exp[1]
        assert np.array_equal(r[2], exp[2]) and np.array_equal(r[3], exp[3]) and np.array_equal(r[4], exp[4])
        n_checked += 1
    assert n_checked >= 30

    class Spot(object):
        def __init__(self, h, w):
            self.h, self.w = int(h), int(w)
    name, frame_hw, offsets, shape, radius, spot_radius, traces, discarded = next(c for c in load_cases() if c[0] == "stack256")
    frame_spots = [[Spot(h, w) for h, w in hw] for hw in frame_hw]
    tr, nd = fl.Experiment.greedy_particle_tracking(frame_spots, shape, offsets=offsets)
    flat = [s for fr in frame_spots for s in fr]
    assert nd == discarded and len(tr) == len(traces)
    for row, exp in zip(tr, traces):
        assert [(-1 if s is None else flat.index(s)) for s in row] == list(exp)
    assert fl.Experiment.accumulate_offsets(offsets)[3] == (sum(o[0] for o in offsets[:4]), sum(o[1] for o in offsets[:4]))


def test_tracking_errors(env):
    torch, N, fl, O = env
    with pytest.raises(ValueError):                     # flexlibrary.py:581-583
        fl.track_fields([[np.array([[5, 5]])]], [[(1, 0)]], (12, 12))
    with pytest.raises(AssertionError):                 # flexlibrary.py:851-856
        fl.track_fields([[np.array([[5, 5], [5, 5]])]], [[(0, 0)]], (12, 12))
    with pytest.raises(ValueError):
        fl.Experiment.accumulate_offsets([(0, 1), (0, 0)])
    with pytest.raises(TypeError):
        fl.Experiment.greedy_particle_tracking([[]], (8, 8))


def test_large_fields_equal_the_oracle(env):
    """More than 32 768 spots in one field (the LDS pairing bitmaps end there; round 4 reads the pairing facts off the links): a
    lattice of 182 x 182 = 33 124 spots per frame moving by one pixel over three frames (99 372 spots) against the oracle tracker."""
    torch, N, fl, O = env
    lattice = np.array([(h, w) for h in range(8, 372, 2) for w in range(8, 372, 2)], np.int32)          # 33 124 spots, 2 px apart
    frames = [lattice, lattice + np.array([0, 1], np.int32), lattice + np.array([1, 1], np.int32)]
    offsets = [(0, 0), (0, 0), (0, 0)]
    got, nd, prev, nxt, kept = fl.track_fields([frames], [offsets], (384, 384), candidate_radius=2)[0]
    o_tr, o_nd, o_prev, o_next, o_kept = O.greedy_tracking(frames, offsets, (384, 384), 2, 0)
    assert len(lattice) > 32768 and nd == o_nd and np.array_equal(kept, o_kept)
    assert np.array_equal(prev, o_prev) and np.array_equal(nxt, o_next) and np.array_equal(got, o_tr)


def test_long_time_series_equal_the_oracle(env):
    """More than 64 frames per field (the kernel's LDS frame tables end there; round 4 keeps the tables of longer series in the
    workspace): 150 frames of drifting, blinking spots with accumulating sub-pixel offsets in two fields of one launch, and a
    65-frame series of empty frames, against the oracle tracker (itself pinned to the reference's traces, tests/test_tracking.py)."""
    torch, N, fl, O = env
    rng = np.random.default_rng(404)
    H = W = 96
    fields, offsets = [], []
    for fld in range(2):
        F = 150 if fld == 0 else 97
        base = np.stack([rng.integers(8, H - 8, 40), rng.integers(8, W - 8, 40)], axis=1)
        drift = np.cumsum(rng.uniform(-0.4, 0.4, (F, 2)), axis=0)
        frames, offs = [], [(0.0, 0.0)]
        for f in range(F):
            on = rng.random(len(base)) > 0.2                               # a fifth of the spots is dark in any frame
            hw = np.unique(np.rint(base[on] + drift[f] + rng.integers(-1, 2, (int(on.sum()), 2))).astype(np.int32), axis=0)
            frames.append(hw)
            if f:
                offs.append((float(-(drift[f] - drift[f - 1])[0]), float(-(drift[f] - drift[f - 1])[1])))
        fields.append(frames)
        offsets.append(offs)
    # (fields of one launch share a frame count: the shorter series is padded with empty frames)
    F = max(len(x) for x in fields)
    for frames, offs in zip(fields, offsets):
        while len(frames) < F:
            frames.append(np.zeros((0, 2), np.int32))
            offs.append((0.0, 0.0))
    got = fl.track_fields(fields, offsets, (H, W), candidate_radius=3)
    for (tr, nd, prev, nxt, kept), frames, offs in zip(got, fields, offsets):
        o_tr, o_nd, o_prev, o_next, o_kept = O.greedy_tracking(frames, offs, (H, W), 3, 0)
        assert nd == o_nd and np.array_equal(kept, o_kept)
        assert np.array_equal(prev, o_prev) and np.array_equal(nxt, o_next) and np.array_equal(tr, o_tr)
        assert tr.shape[1] == F and len(tr) > 40 and (np.sum(tr >= 0, axis=1) > 20).any()      # long traces exist
    empty = fl.track_fields([[np.zeros((0, 2), np.int32)] * 65], [[(0, 0)] * 65], (12, 12))[0]
    assert empty[0].shape == (0, 65) and empty[1] == 0


def test_pair_list_grows_on_demand(env):
    """A wide candidate radius on a dense lattice produces far more candidate pairs than the first guess of the pair list
    (8 x the largest frame): the call is repeated with a longer list instead of failing, and equals the oracle."""
    torch, N, fl, O = env
    base = np.array([(h, w) for h in range(4, 60, 2) for w in range(4, 60, 2)], np.int32)   # 784 spots, 2 px apart
    frames = [base, base + np.array([0, 1], np.int32), base]
    offsets = [(0, 0), (0, 0), (0, 0)]
    got, nd, prev, nxt, kept = fl.track_fields([frames], [offsets], (64, 64), candidate_radius=7)[0]
    o_tr, o_nd, o_prev, o_next, o_kept = O.greedy_tracking(frames, offsets, (64, 64), 7, 0)
    assert nd == o_nd and np.array_equal(got, o_tr) and np.array_equal(prev, o_prev) and np.array_equal(nxt, o_next)


def test_x87_dnrm2_restatement(env):
    """fsq_dnrm2_2 (integer arithmetic on 64-bit significands) == the oracle's long-double chain on 2 M vectors:
    the 1/20-pixel grid of registration offsets, arbitrary doubles, wide exponent ranges, zeros."""
    torch, N, fl, O = env
    rng = np.random.default_rng(11)
    n = 500000
    dh = np.concatenate([rng.integers(-60, 61, n) / 20.0, rng.uniform(-3, 3, n), rng.uniform(-1000, 1000, n),
                         np.ldexp(rng.uniform(-2, 2, n), rng.integers(-200, 200, n)), [0.0, 0.0, 3.0, -0.0]])
    dw = np.concatenate([rng.integers(-60, 61, n) / 20.0, rng.uniform(-3, 3, n), rng.uniform(-3, 3, n) * 1e-3,
                         np.ldexp(rng.uniform(-2, 2, n), rng.integers(-200, 200, n)), [0.0, 1.5, 4.0, 2.0]])
    d_h, d_w = torch.from_numpy(dh).cuda(), torch.from_numpy(dw).cuda()
    out = torch.empty_like(d_h)
    N.check(N.lib().fsq_selftest_dnrm2(d_h.data_ptr(), d_w.data_ptr(), len(dh), out.data_ptr(),
                                       torch.cuda.current_stream().cuda_stream), "fsq_selftest_dnrm2")
    got = out.cpu().numpy()
    import ctypes
    L = O.lib()
    L.fsq_o_euclid2.restype = ctypes.c_double
    L.fsq_o_euclid2.argtypes = [ctypes.c_double, ctypes.c_double]
    exp = np.array([L.fsq_o_euclid2(a, b) for a, b in zip(dh, dw)])
    assert np.array_equal(got.view(np.uint64), exp.view(np.uint64))
    assert (exp != np.sqrt(dh * dh + dw * dw)).mean() > 0.05           # ...and plain double arithmetic is not the same


@pytest.mark.parametrize("case", list(load_centroid_cases()), ids=lambda c: c[0])
def test_centroid_tracking_equals_reference(env, case):
    """N4: luminosity-centroid tracking == the reference's recorded tracks (tests/golden/centroid_tracking.npz)."""
    torch, N, fl, O = env
    name, frames, init, offsets, sr, cut, hw = case
    got, present = fl.centroid_track_fields(frames, init, None, sr, cut, None if offsets is None else offsets[None])
    assert np.array_equal(got, hw) and np.array_equal(present, hw[:, :, 0] >= 0)


def test_centroid_tracking_batch_and_objects(env):
    torch, N, fl, O = env
    cases = {c[0]: c for c in load_centroid_cases()}
    name, frames, init, offsets, sr, cut, hw = cases["stack160_registered"]
    rng = np.random.default_rng(3)
    stack = np.stack([frames, frames[::-1].copy(), np.roll(frames, 5, axis=1)])        # three different fields
    offs = np.stack([offsets, rng.integers(-3, 4, offsets.shape), np.zeros_like(offsets)])
    offs[:, 0] = 0
    pts = np.concatenate([init, init, init])
    fld = np.repeat(np.arange(3), len(init)).astype(np.int32)
    got, present = fl.centroid_track_fields(stack, pts, fld, sr, cut, offs)
    for k in range(3):
        exp, ep = O.centroid_tracking(stack[k], init, sr, cut, offs[k])
        assert np.array_equal(got[fld == k], exp) and np.array_equal(present[fld == k], ep)
    # the same fields as uint32 frames with values up to 2^31 - 1 (fsq_centroid_tracking_u32), other radii and cut-offs
    wide = np.minimum(stack.astype(np.int64) * np.array([41, 9000, 300000])[:, None, None, None], 2 ** 31 - 1).astype(np.uint32)
    for sr2, cut2 in ((3, 3.0), (1, 0.5), (6, 8.0)):
        got, present = fl.centroid_track_fields(wide, pts, fld, sr2, cut2, offs)
        for k in range(3):
            exp, ep = O.centroid_tracking(wide[k], init, sr2, cut2, offs[k])
            assert np.array_equal(got[fld == k], exp) and np.array_equal(present[fld == k], ep), (sr2, k)
    with pytest.raises(NotImplementedError):
        fl.centroid_track_fields(wide, pts, fld, 513, cut, offs)

    class Img(object):
        def __init__(self, a):
            self.image = a
    imgs = [Img(f) for f in frames]
    spots = [fl.Spot(imgs[0], int(h), int(w), 5) for h, w in init]
    tracks = fl.Experiment.luminosity_centroid_particle_tracking(imgs, spots, offsets=[tuple(int(x) for x in o) for o in offsets])
    assert len(tracks) == len(init)
    got_hw = np.array([[(-1, -1) if s is None else (s.h, s.w) for s in tr] for tr in tracks])
    assert np.array_equal(got_hw, hw)
    with pytest.raises(ValueError):
        fl.Experiment.luminosity_centroid_particle_tracking(imgs[1:], spots)
    with pytest.raises(TypeError):
        fl.centroid_track_fields(frames, init, None, 3, 3.0, offsets[None] + 0.5)
    with pytest.raises(ValueError):
        fl.centroid_track_fields(np.zeros((2, 16, 16), np.uint16), [(8, 8)])
    with pytest.raises(AttributeError):
        fl.Spot(imgs[0], 1, 5, 5)


def test_image_and_spot_surface(env):
    """flexlibrary.Image.find_gaussian_psfs (flexlibrary.py:426-455) and the Spot metrics on the path: Spots carry the
    reference's dict keys and tuples; photometry equals the reference's recorded values (tests/golden/photometry.npz);
    a stack of Images in one pass gives the same Spots; Images feed both trackers."""
    import os
    from _util import GOLD, load_field
    torch, N, fl, O = env
    g, img = load_field("f5_small_96")
    im = fl.Image(image=img)
    assert im.find_gaussian_psfs() == len(g["table_keys"]) == len(im.spots)
    assert [(s.h, s.w) for s in im.spots] == [tuple(int(v) for v in k) for k in g["table_keys"]]
    assert all(s.size == 5 and s.parent_Image is im for s in im.spots)
    got7 = np.array([[float(x) for x in s.gaussian_fit[:7]] for s in im.spots])
    assert np.array_equal(got7.view(np.uint64), np.ascontiguousarray(g["table7"]).view(np.uint64))
    ph = np.load(os.path.join(GOLD, "photometry.npz"))
    n = len(im.spots)
    assert [float(s.mexican_hat_photometry_metric()) for s in im.spots] == list(ph["mexican_hat_b6_r9_f5_small_96"][:n])
    assert [float(s.mexican_hat_photometry_metric(brim_size=2, radius=4)) for s in im.spots] == list(ph["mexican_hat_b2_r4_f5_small_96"][:n])
    assert [float(s.gaussian_volume_photometry_metric()) for s in im.spots] == list(ph["gaussian_volume_f5_small_96"][:n])
    s0 = im.spots[0]
    assert np.array_equal(s0.image_slice(), img[s0.h - 2:s0.h + 3, s0.w - 2:s0.w + 3]) and s0.valid_slice() and not s0.valid_slice(radius=90)
    assert s0.simple_photometry_metric() == img[s0.h - 2:s0.h + 3, s0.w - 2:s0.w + 3].sum()
    assert s0.illumina_s_n() == O.illumina_s_n(img[s0.h - 2:s0.h + 3, s0.w - 2:s0.w + 3].astype(np.int64))
    assert im.find_gaussian_psfs(spots_append=True) == n and len(im.spots) == 2 * n
    assert im.find_gaussian_psfs(pflib_args={"r_2_threshold": 0.99}, spots_append=False) == len(im.spots) < n
    with pytest.raises(AttributeError):
        fl.Image()
    # a cycle stack as Images: one batched pass, then the greedy tracker on the Spots
    name, frame_hw, offsets, shape, radius, spot_radius, traces, discarded = next(c for c in load_cases() if c[0] == "stack256")
    from fluorosequencingimageanalysis_amd import synth
    frames, _ = synth.make_cycle_stack(30, n_cycles=8, shape=(256, 256), n_spots=150)
    imgs = [fl.Image(image=f) for f in frames]
    counts = fl.find_gaussian_psfs_batch(imgs)
    assert counts == [len(x) for x in frame_hw]
    for im_, hw in zip(imgs, frame_hw):
        assert [(s.h, s.w) for s in im_.spots] == [tuple(int(v) for v in k) for k in hw]
    tr, nd = fl.Experiment.greedy_particle_tracking([im_.spots for im_ in imgs], imgs[0].image.shape, offsets=offsets)
    flat = [s for im_ in imgs for s in im_.spots]
    assert nd == discarded and [[(-1 if s is None else flat.index(s)) for s in row] for row in tr] == traces.tolist()
    # SequenceExperiment: registration (== the reference's phase_correlate tuples behind the golden offsets) + tracking
    ex = fl.SequenceExperiment(peptide_frames=imgs, alignment_frames=imgs)
    assert [(float(a), float(b)) for a, b in ex.offsets_from_frames(upsample_factor=20)] == [(float(a), float(b)) for a, b in offsets]
    tr2 = ex.trace_existing_spots()
    assert ex.num_discarded_spots == discarded and [[(-1 if s is None else flat.index(s)) for s in row] for row in tr2] == traces.tolist()
    with pytest.raises(AttributeError):
        fl.SequenceExperiment(peptide_frames=imgs).offsets_from_frames()


def test_cli_then_loader_then_tracker_reproduces_reference_traces(env, tmp_path):
    """The chain of the experiment scripts (basic_experiment_script.py:241-257, 377-399, 424-471): the frames of a cycle stack
    are fitted by the command line into per-image pickles, loaded back with Experiment.easy_load_processed_image (the resume
    path: nothing is fitted again), registered and tracked - traces, discarded count and offsets equal the reference's
    (tests/golden/tracking.npz, case stack256)."""
    from PIL import Image as PILImage
    torch, N, fl, O = env
    from fluorosequencingimageanalysis_amd import basic_image_script as cli, synth
    name, frame_hw, offsets, shape, radius, spot_radius, traces, discarded = next(c for c in load_cases() if c[0] == "stack256")
    frames, _ = synth.make_cycle_stack(30, n_cycles=8, shape=(256, 256), n_spots=150)
    d = tmp_path / "field0"
    d.mkdir()
    for f, fr in enumerate(frames):
        PILImage.fromarray(fr).save(str(d / ("cycle%02d.tif" % f)), format="TIFF")
    res = cli.main(["-L", str(tmp_path / "log.txt"), str(d)])
    assert len(res) == len(frames)
    imgs, lost = [], 0
    for f in range(len(frames)):
        converted = res[str(d / ("cycle%02d.tif" % f))][0]
        im, dropped = fl.Experiment.easy_load_processed_image(converted)      # finds <converted>_psfs_<hash>.pkl
        imgs.append(im)
        lost += dropped
        assert np.array_equal(im.image, frames[f])
    assert lost == 0
    for im, hw in zip(imgs, frame_hw):
        assert [(s.h, s.w) for s in im.spots] == [tuple(int(v) for v in k) for k in hw]
        assert all(s.size == 5 and s.gaussian_fit[8].shape == (5, 5) for s in im.spots)
    ex = fl.SequenceExperiment(peptide_frames=imgs, alignment_frames=imgs)
    assert [(float(a), float(b)) for a, b in ex.offsets_from_frames(upsample_factor=20)] == [(float(a), float(b)) for a, b in offsets]
    tr = ex.trace_existing_spots()
    flat = [s for im in imgs for s in im.spots]
    assert ex.num_discarded_spots == discarded
    assert [[(-1 if s is None else flat.index(s)) for s in row] for row in tr] == traces.tolist()
