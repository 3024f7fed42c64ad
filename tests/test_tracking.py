"""CPU tests (-m "not gpu") of the tracking oracle (oracle/fsq_track_oracle.c) against outputs of the
reference's own Experiment.greedy_particle_tracking (flexlibrary.py:680-1027), recorded by
oracle/gen_golden.py --only track into tests/golden/tracking.npz."""
import os

import numpy as np
import pytest

import oracle as O
from _util import GOLD


@pytest.fixture(scope="module", autouse=True)
def _build():
    O.build()


def load_cases(file="tracking.npz"):
    g = np.load(os.path.join(GOLD, file))
    for name in g["names"]:
        name = str(name)
        counts = g[name + "_counts"]
        hw = g[name + "_hw"]
        cuts = np.concatenate([[0], np.cumsum(counts)])
        frame_hw = [hw[cuts[f]:cuts[f + 1]] for f in range(len(counts))]
        yield (name, frame_hw, [tuple(o) for o in g[name + "_offsets"]], tuple(int(x) for x in g[name + "_shape"]),
               int(g[name + "_radius"][0]), float(g[name + "_radius"][1]), g[name + "_traces"], int(g[name + "_discarded"]))


def test_tracking_goldens_cover_the_cases():
    names = [c[0] for c in load_cases()]
    assert {"stack256", "stack512", "ties_int", "drift_dropout", "drift_radius3_edge2", "compete", "empty_frames"} <= set(names)


@pytest.mark.parametrize("case", list(load_cases()), ids=lambda c: c[0])
def test_oracle_tracking_equals_reference(case):
    """Trace membership and order, and the number of spots discarded for drifting out of the field, exactly."""
    name, frame_hw, offsets, shape, radius, spot_radius, traces, discarded = case
    got, nd, prev, nxt, kept = O.greedy_tracking(frame_hw, offsets, shape, radius, spot_radius)
    assert nd == discarded
    assert got.shape == traces.shape and np.array_equal(got, traces)
    assert int((~kept).sum()) == discarded


@pytest.mark.parametrize("case", list(load_cases("tracking_long.npz")), ids=lambda c: c[0])
def test_oracle_tracking_equals_reference_on_long_series_and_large_fields(case):
    """A 90-frame series (drift, drop-outs, re-appearing spots, the stage returning to its start) and a field of 34 000 spots
    through the reference itself (oracle/gen_golden.py --only track_long): beyond the 64 frames / 32 768 spots the GPU tracker
    was limited to until round 4."""
    name, frame_hw, offsets, shape, radius, spot_radius, traces, discarded = case
    assert len(frame_hw) > 64 or sum(len(x) for x in frame_hw) > 32768
    got, nd, prev, nxt, kept = O.greedy_tracking(frame_hw, offsets, shape, radius, spot_radius)
    assert nd == discarded and got.shape == traces.shape and np.array_equal(got, traces)


def test_euclid_is_dnrm2_of_this_scipy():
    """The pair distance: scipy.spatial.distance.euclidean (flexlibrary.py:927) = OpenBLAS dnrm2, x87 extended precision -
    the oracle's restatement equals it on every one of 50 000 displacement vectors (plain double arithmetic does not)."""
    from scipy.spatial.distance import euclidean
    rng = np.random.default_rng(1)
    a = np.round(rng.uniform(-3, 3, (50000, 2)) * 20) / 20 + rng.integers(0, 500, (50000, 2))
    b = a + np.round(rng.uniform(-2.5, 2.5, (50000, 2)) * 20) / 20
    b[:1000] = a[:1000] + rng.uniform(-3, 3, (1000, 2))
    plain = 0
    for u, v in zip(a, b):
        d = u - v
        assert euclidean(u, v) == O.euclid2(d[0], d[1])
        plain += euclidean(u, v) != np.sqrt(d[0] * d[0] + d[1] * d[1])
    assert plain > 1000


def test_tracking_errors():
    with pytest.raises(ValueError):                     # flexlibrary.py:581-583
        O.greedy_tracking([np.array([[5, 5]])], [(1, 0)], (12, 12))
    with pytest.raises(AssertionError):                 # two spots of one frame in one bin, flexlibrary.py:851-856
        O.greedy_tracking([np.array([[5, 5], [5, 5]])], [(0, 0)], (12, 12))


def test_x87_header_on_host(tmp_path):
    """csrc/fsq_x87.h (what the GPU kernel computes the pair distance with) compiled for the host: equal to the
    long-double chain on 2 M vectors."""
    import subprocess
    from _util import ROOT
    exe = str(tmp_path / "x87_check")
    subprocess.check_call(["g++", "-O2", "-o", exe, os.path.join(ROOT, "tests", "x87_check.cpp")])
    out = subprocess.check_output([exe]).decode()
    assert "bad=0" in out, out


def load_centroid_cases(file="centroid_tracking.npz"):
    g = np.load(os.path.join(GOLD, file))
    for name in g["names"]:
        name = str(name)
        off = g[name + "_offsets"]
        yield (name, g[name + "_frames"], g[name + "_init"], None if len(off) == 0 else off, int(g[name + "_params"][0]),
               float(g[name + "_params"][1]), g[name + "_hw"])


@pytest.mark.parametrize("case", list(load_centroid_cases()), ids=lambda c: c[0])
def test_oracle_centroid_tracking_equals_reference(case):
    """N4: Experiment.luminosity_centroid_particle_tracking (flexlibrary.py:1262-1317) as recorded from the reference
    (oracle/gen_golden.py --only centroid): every spot's position, or None, in every frame."""
    name, frames, init, offsets, sr, cut, hw = case
    got, present = O.centroid_tracking(frames, init, sr, cut, offsets)
    assert np.array_equal(got, hw)
    assert np.array_equal(present, hw[:, :, 0] >= 0)


@pytest.mark.parametrize("case", list(load_centroid_cases("centroid_tracking_wide.npz")), ids=lambda c: c[0])
def test_oracle_centroid_tracking_of_wide_frames_equals_reference(case):
    """uint32 frames with pixel values beyond 16 bits (oracle/gen_golden.py --only centroid_wide)."""
    name, frames, init, offsets, sr, cut, hw = case
    assert frames.dtype == np.uint32 and int(frames.max()) > 65535
    got, present = O.centroid_tracking(frames, init, sr, cut, offsets)
    assert np.array_equal(got, hw) and np.array_equal(present, hw[:, :, 0] >= 0)


def test_centroid_goldens_cover_the_cases():
    cases = {c[0]: c for c in load_centroid_cases()}
    assert (cases["borders"][6][:, 1:, 0] < 0).any() and (cases["borders"][6][:, 1:, 0] >= 0).any()
    a, b = cases["stack160_registered"][6], cases["stack160_strict"][6]
    assert not np.array_equal(a, b)                     # the s_n cut-off branch (same coordinates as the prior spot) is taken
    with pytest.raises(ValueError):                     # all-zero search window: centre of mass is NaN
        O.centroid_tracking(np.zeros((2, 16, 16), np.uint16), [(8, 8)])


def test_flexlibrary_host_helpers_match_reference_goldens():
    """The host-side helpers of the flexlibrary mirror (accumulate_offsets, discard_dropouts, round_coordinates,
    offset arithmetic: flexlibrary.py:567-678) against the recorded reference runs: the number of spots
    discard_dropouts drops per case equals the reference's, and the cumulative offsets are Python's left-to-right sums."""
    from fluorosequencingimageanalysis_amd import flexlibrary as fl

    class S(object):
        def __init__(self, h, w):
            self.h, self.w = int(h), int(w)
    for name, frame_hw, offsets, shape, radius, spot_radius, traces, discarded in load_cases():
        cum = fl.Experiment.accumulate_offsets(offsets)
        assert cum[0] == (0, 0) and len(cum) == len(offsets)
        for f in range(len(offsets)):
            assert cum[f] == (sum([o[0] for o in offsets[:f + 1]]), sum([o[1] for o in offsets[:f + 1]]))
            assert fl.Experiment.get_cumulative_offset(offsets, f) == cum[f]
        dropped = 0
        for f, hw in enumerate(frame_hw):
            spots = [S(h, w) for h, w in hw]
            kept, nd = fl.Experiment.discard_dropouts(spots, [cum[f]] * len(spots), cum, shape, spot_radius)
            assert len(kept) + nd == len(spots)
            dropped += nd
        assert dropped == discarded, name
    assert fl.Experiment.round_coordinates(2.5, -2.5) == (3, -3)                   # Python-2 round
    assert fl.Experiment.unapply_offset(fl.Experiment.apply_offset((3, 4), (1.5, -2)), (1.5, -2)) == (3.0, 4)
    assert fl.Experiment.offset_frame_coordinates([(0, 0), (1, 2), (3, 4)], (10, 10), 2, 1) == (13, 14)
    with pytest.raises(ValueError):
        fl.Experiment.accumulate_offsets([(1, 0)])
