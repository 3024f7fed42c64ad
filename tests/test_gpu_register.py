"""GPU parity tests (-m gpu) of the registration path against the reference's own outputs
(tests/golden/registration.npz, produced by the reference's phase_correlate on numpy.fft).
Shifts must agree exactly on the 1/upsample grid; error / diffphase to FFT rounding level (1e-9 abs,
different FFT factorizations round differently - SURVEY.md 8c)."""
import os

import numpy as np
import pytest

from _util import GOLD

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pc():
    import torch
    assert torch.cuda.is_available()
    from fluorosequencingimageanalysis_amd import phase_correlate as m
    return m


def test_golden_pairs(pc):
    g = np.load(os.path.join(GOLD, "registration.npz"))
    for name in g["names"]:
        a, b = g["ref_" + name], g["reg_" + name]
        for uf in (1, 20, 100):
            r = pc.phase_correlate(a, b, upsample_factor=uf)
            e = g["out_%s_uf%d" % (name, uf)]
            assert float(r[0]) == e[0] and float(r[1]) == e[1], (name, uf, r, e)
            assert abs(float(r[2]) - e[2]) < 1e-9 and abs(float(r[3]) - e[3]) < 1e-9, (name, uf, r, e)
    r = pc.phase_correlate(g["ref_odd0_63x65"], g["reg_odd0_63x65"], 1)
    assert isinstance(r[0], np.int64) and isinstance(r[2], np.float64)


def test_batch_equals_single_and_known_shift(pc):
    from fluorosequencingimageanalysis_amd import synth
    frames, off = synth.make_cycle_stack(9, n_cycles=4, shape=(256, 256), n_spots=200)
    out = pc.phase_correlate_batch(frames[:-1].astype(np.float64), frames[1:].astype(np.float64), 20)
    for k in range(3):
        single = pc.phase_correlate(frames[k], frames[k + 1], 20)
        assert float(single[0]) == out[k][0] and float(single[1]) == out[k][1]
        assert abs(float(single[2]) - out[k][2]) < 1e-12 and abs(float(single[3]) - out[k][3]) < 1e-12
        true = off[k + 1] - off[k]          # content moved by +true  => reported shift is -true
        assert abs(out[k][0] + true[0]) < 0.15 and abs(out[k][1] + true[1]) < 0.15


def test_errors(pc):
    with pytest.raises(ValueError):
        pc.phase_correlate(np.zeros((8, 8)), np.zeros((8, 9)))
    with pytest.raises(ValueError):
        pc.phase_correlate(np.zeros((2, 8, 8)), np.zeros((2, 8, 8)))
